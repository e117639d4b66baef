"""TPC-H-shaped benchmark support (the counterpart of the reference's benchmarks/src/bin/tpch.rs and benchmarks/queries/): the
device-side generator wrappers (SURVEY.md section 8d synthetic tables) and the physical plans of q1 / q3 / q5 written with the
reference's operator names.  Used by bench.py, bench_extras.py and the test-suite (tests/tpch_util.py adds the oracle side)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


SEED_LINEITEM, SEED_ORDERS, SEED_CUSTOMER, SEED_SUPPLIER = 0x1, 0x2, 0x3, 0x4


D152 = {"Decimal128": [15, 2]}


Q1_SHIPDATE_MAX = 10471          # date '1998-09-02' as days (the reference plan folds the date, planner.rs:489)


LINEITEM_ROWS = {1: 6_001_215, 10: 59_986_052, 100: 600_037_902}


# ------------------------------------------------------------------ device generator
def gen_lineitem_device(tc, n, seed=SEED_LINEITEM, seed_orders=SEED_ORDERS, row0=0, n_supp=10_000,
                        columns=("l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate")):
    """Device-resident lineitem columns (Arrow physical layout) produced by the HIP generator."""
    import torch
    import arrow_ballista_amd as g
    from arrow_ballista_amd import binding as B
    dev = tc.device
    bufs, cs = {}, B.gpuq_lineitem_cols()
    cols = []

    def alloc(nbytes):
        return torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
    for name in columns:
        if name in ("l_orderkey", "l_suppkey"):
            t = alloc(8 * n); setattr(cs, name, t.data_ptr()); cols.append(g.DeviceColumn(name, "Int64", t, n, nullable=False))
        elif name in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"):
            t = alloc(16 * n); setattr(cs, name, t.data_ptr()); cols.append(g.DeviceColumn(name, D152, t, n, nullable=False))
        elif name == "l_shipdate":
            t = alloc(4 * n); setattr(cs, name, t.data_ptr()); cols.append(g.DeviceColumn(name, "Date32", t, n, nullable=False))
        elif name in ("l_returnflag", "l_linestatus"):
            t = alloc(n); o = torch.empty(n + 4, dtype=torch.int32, device=dev)
            setattr(cs, name, t.data_ptr()); setattr(cs, name + "_off", o.data_ptr())
            cols.append(g.DeviceColumn(name, "Utf8", t, n, offsets=o, nullable=False))
        else:
            raise KeyError(name)
    tc.ctx.check(tc.ctx.L.gpuq_gen_lineitem(tc.ctx.h, tc.stream_ptr(), seed, seed_orders, row0, n, n_supp, C.byref(cs)))
    tc.sync()
    return g.DeviceTable(cols, n)


# ------------------------------------------------------------------ q1 plan (reference benchmarks/queries/q1.sql)
def q1_plan(source, two_phase=True, strategy="auto"):
    """Physical plan in the shape DataFusion produces for q1 (stage trees: scheduler/src/planner.rs:376-392):
       SortExec <- ProjectionExec <- AggregateExec(FinalPartitioned) <- AggregateExec(Partial)
                <- ProjectionExec <- CoalesceBatchesExec <- FilterExec <- source"""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
    s = source.schema()
    one = lit(1, ("Decimal128", 20, 0))     # Int64(1) coerced to Decimal128(20,0) by the planner
    filt = g.FilterExec(binary(col("l_shipdate", s), Op.LtEq, lit(Q1_SHIPDATE_MAX, "Date32")), source)
    cb = g.CoalesceBatchesExec(filt, 8192)
    disc_price = binary(col("l_extendedprice", s), Op.Multiply, binary(one, Op.Minus, col("l_discount", s)))
    proj = g.ProjectionExec([
        (disc_price, "__common_expr_1"), (col("l_quantity", s), "l_quantity"), (col("l_extendedprice", s), "l_extendedprice"),
        (col("l_discount", s), "l_discount"), (col("l_tax", s), "l_tax"),
        (col("l_returnflag", s), "l_returnflag"), (col("l_linestatus", s), "l_linestatus")], cb)
    ps = [{"name": n} for n in ("__common_expr_1", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus")]
    c = lambda n: col(n, ps)
    aggs = [
        {"fn": "SUM", "expr": c("l_quantity"), "name": "SUM(lineitem.l_quantity)"},
        {"fn": "SUM", "expr": c("l_extendedprice"), "name": "SUM(lineitem.l_extendedprice)"},
        {"fn": "SUM", "expr": c("__common_expr_1"), "name": "SUM(lineitem.l_extendedprice * (Int64(1) - lineitem.l_discount))"},
        {"fn": "SUM", "expr": binary(c("__common_expr_1"), Op.Multiply, binary(one, Op.Plus, c("l_tax"))),
         "name": "SUM(lineitem.l_extendedprice * (Int64(1) - lineitem.l_discount) * (Int64(1) + lineitem.l_tax))"},
        {"fn": "AVG", "expr": c("l_quantity"), "name": "AVG(lineitem.l_quantity)"},
        {"fn": "AVG", "expr": c("l_extendedprice"), "name": "AVG(lineitem.l_extendedprice)"},
        {"fn": "AVG", "expr": c("l_discount"), "name": "AVG(lineitem.l_discount)"},
        {"fn": "COUNT", "expr": lit(1), "name": "COUNT(*)"},
    ]
    groups = [(c("l_returnflag"), "l_returnflag"), (c("l_linestatus"), "l_linestatus")]
    if two_phase:
        partial = g.AggregateExec("Partial", groups, aggs, proj, strategy=strategy)
        fs = partial.schema()
        final = g.AggregateExec("FinalPartitioned", [(col("l_returnflag", fs), "l_returnflag"), (col("l_linestatus", fs), "l_linestatus")],
                                [dict(a, expr=None) for a in aggs], g.CoalesceBatchesExec(partial, 8192))
    else:
        final = g.AggregateExec("Single", groups, aggs, proj, strategy=strategy)
    os_ = final.schema()
    names = ["l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order"]
    out = g.ProjectionExec([(col(f["name"], os_), n) for f, n in zip(os_, names)], final)
    so = out.schema()
    return g.SortExec([{"expr": col("l_returnflag", so), "asc": True, "nulls_first": False},
                       {"expr": col("l_linestatus", so), "asc": True, "nulls_first": False}], out)


def q1_split_plan(lineitem, state_capacity=64):
    """q1 as bench.py drives it on N ranks: (partial, full, final_src).  partial = the per-rank stage (fused filter +
    projection + partial aggregate) whose result is a fixed-layout record of `state_capacity` rows; full = the rest of the
    plan (final aggregate, projection, sort) reading its input from final_src.partitions[0]."""
    import arrow_ballista_amd as g
    full = q1_plan(g.MemoryExec([lineitem]), two_phase=True)
    node = full
    chain = []
    while True:
        chain.append(node)
        if isinstance(node, g.AggregateExec) and node.mode == "Partial":
            break
        node = node.children()[0]
    partial = node
    partial.output_capacity = state_capacity
    final_agg = next(c for c in chain if isinstance(c, g.AggregateExec) and c.mode == "FinalPartitioned")
    final_src = g.MemoryExec([None], schema=partial.schema())
    final_agg.input = final_src
    return partial, full, final_src


def q1_dist_plan(lineitem):
    """q1 across the ranks of a node: the partial aggregate over the rank's lineitem shard, the partial states of all ranks
    gathered (BroadcastExec: 4 groups x 8 state columns per rank -- no row exchange), final aggregate + projection + sort on
    every rank."""
    import arrow_ballista_amd as g
    full = q1_plan(g.MemoryExec([lineitem]), two_phase=True)
    node = full
    while not (isinstance(node, g.AggregateExec) and node.mode == "FinalPartitioned"):
        node = node.children()[0]
    partial = node.input
    while not (isinstance(partial, g.AggregateExec) and partial.mode == "Partial"):
        partial = partial.children()[0]
    node.input = g.BroadcastExec(partial)
    return full


def run_q1(tc, lineitem, two_phase=True, strategy="auto"):
    import arrow_ballista_amd as g
    plan = q1_plan(g.MemoryExec([lineitem]), two_phase, strategy)
    return g.plan.materialize(tc, plan.execute(0, tc))


def table_to_rows(tc, table):
    """Materialised DeviceTable -> list of tuples; decimals as unscaled ints, dates as days."""
    import pyarrow as pa
    t = table.to_arrow(tc.ctx)
    cols = []
    for f, c in zip(t.schema, t.columns):
        if pa.types.is_decimal128(f.type):
            cols.append([None if v is None else int(v.scaleb(f.type.scale)) for v in c.to_pylist()])
        elif pa.types.is_date32(f.type):
            cols.append(c.cast(pa.int32()).to_pylist())
        else:
            cols.append(c.to_pylist())
    return list(zip(*cols)) if cols else []


def q1_result_to_rows(tc, table):
    return [tuple(r) for r in table_to_rows(tc, table)]


# ------------------------------------------------------------------ other tables (device generator + oracle restatement)
NATIONS = [("ALGERIA", 0), ("ARGENTINA", 1), ("BRAZIL", 1), ("CANADA", 1), ("EGYPT", 4), ("ETHIOPIA", 0), ("FRANCE", 3), ("GERMANY", 3), ("INDIA", 2),
           ("INDONESIA", 2), ("IRAN", 4), ("IRAQ", 4), ("JAPAN", 2), ("JORDAN", 4), ("KENYA", 0), ("MOROCCO", 0), ("MOZAMBIQUE", 0), ("PERU", 1),
           ("CHINA", 2), ("ROMANIA", 3), ("SAUDI ARABIA", 4), ("VIETNAM", 2), ("RUSSIA", 3), ("UNITED KINGDOM", 3), ("UNITED STATES", 1)]


REGIONS = ["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"]      # ids as in ballista/scheduler/testdata/region/region.tbl


def nation_region_arrow():
    import pyarrow as pa
    nation = pa.table({"n_nationkey": pa.array(range(25), type=pa.int64()), "n_name": pa.array([n for n, _ in NATIONS]),
                       "n_regionkey": pa.array([r for _, r in NATIONS], type=pa.int64())})
    region = pa.table({"r_regionkey": pa.array(range(5), type=pa.int64()), "r_name": pa.array(REGIONS)})
    nation = nation.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in nation.schema]))
    region = region.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in region.schema]))
    return nation, region


def _dev_cols(tc, spec, n):
    """spec: [(name, type, bytes_per_row | ('utf8', data_bytes))] -> (columns, {name: ptr})"""
    import torch
    import arrow_ballista_amd as g
    cols, ptrs = [], {}
    for name, ty, w in spec:
        if isinstance(w, tuple):
            t = torch.empty(w[1] + 16, dtype=torch.uint8, device=tc.device)
            o = torch.empty(n + 4, dtype=torch.int32, device=tc.device)
            ptrs[name], ptrs[name + "_off"] = t.data_ptr(), o.data_ptr()
            cols.append(g.DeviceColumn(name, ty, t, n, offsets=o, nullable=False))
        else:
            t = torch.empty(w * n + 16, dtype=torch.uint8, device=tc.device)
            ptrs[name] = t.data_ptr()
            cols.append(g.DeviceColumn(name, ty, t, n, nullable=False))
    return cols, ptrs


def gen_orders_device(tc, n, n_cust, seed=SEED_ORDERS, row0=0):
    import arrow_ballista_amd as g
    from arrow_ballista_amd import binding as B
    cols, p = _dev_cols(tc, [("o_orderkey", "Int64", 8), ("o_custkey", "Int64", 8), ("o_orderdate", "Date32", 4), ("o_shippriority", "Int32", 4)], n)
    cs = B.gpuq_orders_cols(**p)
    tc.ctx.check(tc.ctx.L.gpuq_gen_orders(tc.ctx.h, tc.stream_ptr(), seed, row0, n, n_cust, C.byref(cs)))
    tc.sync()
    return g.DeviceTable(cols, n)


def gen_customer_device(tc, n, seed=SEED_CUSTOMER, row0=0):
    import arrow_ballista_amd as g
    from arrow_ballista_amd import binding as B
    assert n % 5 == 0 and row0 % 5 == 0
    cols, p = _dev_cols(tc, [("c_custkey", "Int64", 8), ("c_nationkey", "Int64", 8), ("c_mktsegment", "Utf8", ("utf8", n * 9))], n)
    cs = B.gpuq_customer_cols(**p)
    tc.ctx.check(tc.ctx.L.gpuq_gen_customer(tc.ctx.h, tc.stream_ptr(), seed, row0, n, C.byref(cs)))
    tc.sync()
    return g.DeviceTable(cols, n)


def gen_supplier_device(tc, n, seed=SEED_SUPPLIER, row0=0):
    import arrow_ballista_amd as g
    from arrow_ballista_amd import binding as B
    cols, p = _dev_cols(tc, [("s_suppkey", "Int64", 8), ("s_nationkey", "Int64", 8)], n)
    cs = B.gpuq_supplier_cols(**p)
    tc.ctx.check(tc.ctx.L.gpuq_gen_supplier(tc.ctx.h, tc.stream_ptr(), seed, row0, n, C.byref(cs)))
    tc.sync()
    return g.DeviceTable(cols, n)


Q3_DATE = 9204       # date '1995-03-15'


Q5_DATE_LO, Q5_DATE_HI = 8766, 9131      # 1994-01-01, 1995-01-01 (the folded ints of planner.rs:489)


def q3_plan(customer, orders, lineitem):
    """reference benchmarks/queries/q3.sql as the physical plan DataFusion builds: build sides on the LEFT."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
    cs, os_, ls = customer.schema(), orders.schema(), lineitem.schema()
    c = g.FilterExec(binary(col("c_mktsegment", cs), Op.Eq, lit("BUILDING")), customer)
    o = g.FilterExec(binary(col("o_orderdate", os_), Op.Lt, lit(Q3_DATE, "Date32")), orders)
    j1 = g.HashJoinExec(g.CoalesceBatchesExec(c), g.CoalesceBatchesExec(o), [(col("c_custkey", cs), col("o_custkey", os_))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    l = g.FilterExec(binary(col("l_shipdate", ls), Op.Gt, lit(Q3_DATE, "Date32")), lineitem)
    j2 = g.HashJoinExec(j1, g.CoalesceBatchesExec(l), [(col("o_orderkey", j1s), col("l_orderkey", ls))], None, "Inner", "CollectLeft", False)
    j2s = j2.schema()
    rev = binary(col("l_extendedprice", j2s), Op.Multiply, binary(lit(1, ("Decimal128", 20, 0)), Op.Minus, col("l_discount", j2s)))
    agg = g.AggregateExec("Single", [(col("l_orderkey", j2s), "l_orderkey"), (col("o_orderdate", j2s), "o_orderdate"), (col("o_shippriority", j2s), "o_shippriority")],
                          [{"fn": "SUM", "expr": rev, "name": "revenue"}], j2)
    as_ = agg.schema()
    proj = g.ProjectionExec([(col("l_orderkey", as_), "l_orderkey"), (col("revenue", as_), "revenue"), (col("o_orderdate", as_), "o_orderdate"),
                             (col("o_shippriority", as_), "o_shippriority")], agg)
    ps = proj.schema()
    return g.SortExec([{"expr": col("revenue", ps), "asc": False, "nulls_first": True}, {"expr": col("o_orderdate", ps), "asc": True, "nulls_first": False}], proj)


def q3_dist_plan(customer, orders, lineitem, world, mode="partitioned"):
    """q3 across the ranks of a node, every rank holding a shard of the three tables (the stages the reference's planner cuts
    at RepartitionExec(Hash), planner.rs:137-151, run as one native plan per rank with the exchanges inside):
      customer |> filter |> c_custkey            -- BroadcastExec (3 M keys per SF100: far below the probe side)
      orders   |> filter |x| customers           -- local CollectLeft join against the broadcast keys
      mode "partitioned": both sides of orders |x| lineitem hash-repartitioned on the order key (RepartitionExec + exchange),
                          HashJoinExec(Partitioned), AggregateExec(Single): a group's rows all meet on one rank
      mode "broadcast":   the joined orders are broadcast instead, lineitem stays where it is; AggregateExec(Partial) ->
                          exchange on l_orderkey -> AggregateExec(FinalPartitioned)
      every rank sorts its groups; the sorted runs are gathered and merged (SortPreservingMergeExec) on every rank."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
    cs, os_, ls = customer.schema(), orders.schema(), lineitem.schema()
    c = g.FilterExec(binary(col("c_mktsegment", cs), Op.Eq, lit("BUILDING")), customer)
    cb = g.BroadcastExec(g.ProjectionExec([(col("c_custkey", cs), "c_custkey")], c))
    cbs = cb.schema()
    o = g.FilterExec(binary(col("o_orderdate", os_), Op.Lt, lit(Q3_DATE, "Date32")), orders)
    j1 = g.HashJoinExec(cb, g.CoalesceBatchesExec(o), [(col("c_custkey", cbs), col("o_custkey", os_))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    j1p = g.ProjectionExec([(col(n, j1s), n) for n in ("o_orderkey", "o_orderdate", "o_shippriority")], j1)
    ps1 = j1p.schema()
    l = g.FilterExec(binary(col("l_shipdate", ls), Op.Gt, lit(Q3_DATE, "Date32")), lineitem)
    lp = g.ProjectionExec([(col(n, ls), n) for n in ("l_orderkey", "l_extendedprice", "l_discount")], l)
    ps2 = lp.schema()
    if mode == "partitioned":
        left = g.RepartitionExchangeExec(j1p, [col("o_orderkey", ps1)], world)
        right = g.RepartitionExchangeExec(lp, [col("l_orderkey", ps2)], world)
        j2 = g.HashJoinExec(left, right, [(col("o_orderkey", ps1), col("l_orderkey", ps2))], None, "Inner", "Partitioned", False)
    else:
        j2 = g.HashJoinExec(g.BroadcastExec(j1p), lp, [(col("o_orderkey", ps1), col("l_orderkey", ps2))], None, "Inner", "CollectLeft", False)
    j2s = j2.schema()
    rev = binary(col("l_extendedprice", j2s), Op.Multiply, binary(lit(1, ("Decimal128", 20, 0)), Op.Minus, col("l_discount", j2s)))
    groups = [(col("l_orderkey", j2s), "l_orderkey"), (col("o_orderdate", j2s), "o_orderdate"), (col("o_shippriority", j2s), "o_shippriority")]
    aggs = [{"fn": "SUM", "expr": rev, "name": "revenue"}]
    if mode == "partitioned":
        agg = g.AggregateExec("Single", groups, aggs, j2)
    else:
        part = g.AggregateExec("Partial", groups, aggs, j2)
        fs = part.schema()
        ex = g.RepartitionExchangeExec(part, [col("l_orderkey", fs)], world)
        agg = g.AggregateExec("FinalPartitioned", [(col(n, fs), n) for n in ("l_orderkey", "o_orderdate", "o_shippriority")], [dict(a, expr=None) for a in aggs], ex)
    as_ = agg.schema()
    proj = g.ProjectionExec([(col("l_orderkey", as_), "l_orderkey"), (col("revenue", as_), "revenue"), (col("o_orderdate", as_), "o_orderdate"),
                             (col("o_shippriority", as_), "o_shippriority")], agg)
    ps = proj.schema()
    order = [{"expr": col("revenue", ps), "asc": False, "nulls_first": True}, {"expr": col("o_orderdate", ps), "asc": True, "nulls_first": False}]
    local = g.SortExec(order, proj)
    return g.SortPreservingMergeExec(order, g.BroadcastExec(local))


def q5_plan(customer, orders, lineitem, supplier, nation, region):
    """reference benchmarks/queries/q5.sql: region |x| nation |x| customer |x| orders |x| lineitem |x| supplier (2-column key)."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, Operator as Op
    rs, ns, cs, os_, ls, ss = region.schema(), nation.schema(), customer.schema(), orders.schema(), lineitem.schema(), supplier.schema()
    r = g.FilterExec(binary(col("r_name", rs), Op.Eq, lit("ASIA")), region)
    j1 = g.HashJoinExec(r, nation, [(col("r_regionkey", rs), col("n_regionkey", ns))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    j2 = g.HashJoinExec(j1, customer, [(col("n_nationkey", j1s), col("c_nationkey", cs))], None, "Inner", "CollectLeft", False)
    j2s = j2.schema()
    o = g.FilterExec(and_(binary(col("o_orderdate", os_), Op.GtEq, lit(Q5_DATE_LO, "Date32")), binary(col("o_orderdate", os_), Op.Lt, lit(Q5_DATE_HI, "Date32"))), orders)
    j3 = g.HashJoinExec(j2, o, [(col("c_custkey", j2s), col("o_custkey", os_))], None, "Inner", "CollectLeft", False)
    j3s = j3.schema()
    j4 = g.HashJoinExec(j3, lineitem, [(col("o_orderkey", j3s), col("l_orderkey", ls))], None, "Inner", "CollectLeft", False)
    j4s = j4.schema()
    j5 = g.HashJoinExec(supplier, j4, [(col("s_suppkey", ss), col("l_suppkey", j4s)), (col("s_nationkey", ss), col("c_nationkey", j4s))], None, "Inner", "CollectLeft", False)
    j5s = j5.schema()
    rev = binary(col("l_extendedprice", j5s), Op.Multiply, binary(lit(1, ("Decimal128", 20, 0)), Op.Minus, col("l_discount", j5s)))
    agg = g.AggregateExec("Single", [(col("n_name", j5s), "n_name")], [{"fn": "SUM", "expr": rev, "name": "revenue"}], j5)
    as_ = agg.schema()
    return g.SortExec([{"expr": col("revenue", as_), "asc": False, "nulls_first": True}], agg)


def q5_dist_plan(customer, orders, lineitem, supplier, nation, region, world):
    """q5 (BASELINE configs[3]: "6-way join, hash-partitioned across the GPUs") as one native plan per rank.  customer, orders,
    lineitem and supplier are sharded; nation and region are replicated (dimension tables far below the reference's broadcast
    threshold, config.rs:198-200).
      region |> filter |x| nation |x| customer shard     -- local CollectLeft joins; the ASIA customers' (key, nation, name)
                                                            are then BROADCAST (a fifth of customer)
      orders shard |> filter |x| customers               -- local CollectLeft join against the broadcast rows
      both sides of orders |x| lineitem hash-repartitioned on the order key, HashJoinExec(Partitioned)
      supplier shards BROADCAST, |x| on (suppkey, nationkey)
      AggregateExec(Partial) by n_name -> partial states gathered on every rank -> AggregateExec(Final) -> SortExec."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, Operator as Op
    rs, ns, cs, os_, ls, ss = region.schema(), nation.schema(), customer.schema(), orders.schema(), lineitem.schema(), supplier.schema()
    r = g.FilterExec(binary(col("r_name", rs), Op.Eq, lit("ASIA")), region)
    j1 = g.HashJoinExec(r, nation, [(col("r_regionkey", rs), col("n_regionkey", ns))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    j2 = g.HashJoinExec(j1, customer, [(col("n_nationkey", j1s), col("c_nationkey", cs))], None, "Inner", "CollectLeft", False)
    j2s = j2.schema()
    cb = g.BroadcastExec(g.ProjectionExec([(col(n, j2s), n) for n in ("c_custkey", "c_nationkey", "n_name")], j2))
    cbs = cb.schema()
    o = g.FilterExec(and_(binary(col("o_orderdate", os_), Op.GtEq, lit(Q5_DATE_LO, "Date32")), binary(col("o_orderdate", os_), Op.Lt, lit(Q5_DATE_HI, "Date32"))), orders)
    j3 = g.HashJoinExec(cb, g.CoalesceBatchesExec(o), [(col("c_custkey", cbs), col("o_custkey", os_))], None, "Inner", "CollectLeft", False)
    j3s = j3.schema()
    j3p = g.ProjectionExec([(col(n, j3s), n) for n in ("o_orderkey", "c_nationkey", "n_name")], j3)
    p3 = j3p.schema()
    lp = g.ProjectionExec([(col(n, ls), n) for n in ("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount")], lineitem)
    pl = lp.schema()
    left = g.RepartitionExchangeExec(j3p, [col("o_orderkey", p3)], world)
    right = g.RepartitionExchangeExec(lp, [col("l_orderkey", pl)], world)
    j4 = g.HashJoinExec(left, right, [(col("o_orderkey", p3), col("l_orderkey", pl))], None, "Inner", "Partitioned", False)
    j4s = j4.schema()
    sb = g.BroadcastExec(supplier)
    j5 = g.HashJoinExec(sb, j4, [(col("s_suppkey", ss), col("l_suppkey", j4s)), (col("s_nationkey", ss), col("c_nationkey", j4s))], None, "Inner", "CollectLeft", False)
    j5s = j5.schema()
    rev = binary(col("l_extendedprice", j5s), Op.Multiply, binary(lit(1, ("Decimal128", 20, 0)), Op.Minus, col("l_discount", j5s)))
    aggs = [{"fn": "SUM", "expr": rev, "name": "revenue"}]
    part = g.AggregateExec("Partial", [(col("n_name", j5s), "n_name")], aggs, j5)
    fs = part.schema()
    fin = g.AggregateExec("Final", [(col("n_name", fs), "n_name")], [dict(a, expr=None) for a in aggs], g.BroadcastExec(part))
    as_ = fin.schema()
    return g.SortExec([{"expr": col("revenue", as_), "asc": False, "nulls_first": True}], fin)
