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
# benchmarks/libgpuq_tpchgen.so (benchmarks/tpchgen/: built by arrow-ballista_amd/build.py next to libgpuq.so, but a library of its own:
# the product carries no test scaffolding).  Runs on the current device (torch's, = the TaskContext's) on the TaskContext's stream.
class gpuq_lineitem_cols(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("l_orderkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax",
                                          "l_shipdate", "l_returnflag", "l_returnflag_off", "l_linestatus", "l_linestatus_off")]


class gpuq_orders_cols(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("o_orderkey", "o_custkey", "o_orderdate", "o_shippriority")]


class gpuq_customer_cols(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("c_custkey", "c_nationkey", "c_mktsegment", "c_mktsegment_off")]


class gpuq_supplier_cols(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("s_suppkey", "s_nationkey")]


_GEN = None


def _gen():
    global _GEN
    if _GEN is None:
        p = os.path.join(ROOT, "benchmarks", "libgpuq_tpchgen.so")
        if not os.path.exists(p):
            raise RuntimeError("benchmarks/libgpuq_tpchgen.so is not built: run `python arrow-ballista_amd/build.py`")
        import arrow_ballista_amd as g
        g.lib()                          # (one HIP runtime per process: torch's, loaded by the binding first)
        L = C.CDLL(p)
        vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
        L.gpuq_tpchgen_lineitem.argtypes = [vp, u64, u64, i64, i64, i64, C.POINTER(gpuq_lineitem_cols)]
        L.gpuq_tpchgen_orders.argtypes = [vp, u64, i64, i64, i64, C.POINTER(gpuq_orders_cols)]
        L.gpuq_tpchgen_customer.argtypes = [vp, u64, i64, i64, C.POINTER(gpuq_customer_cols)]
        L.gpuq_tpchgen_supplier.argtypes = [vp, u64, i64, i64, C.POINTER(gpuq_supplier_cols)]
        L.gpuq_tpchgen_last_error.restype = C.c_char_p
        for f in (L.gpuq_tpchgen_lineitem, L.gpuq_tpchgen_orders, L.gpuq_tpchgen_customer, L.gpuq_tpchgen_supplier):
            f.restype = i32
        _GEN = L
    return _GEN


def _gen_check(tc, rc):
    if rc != 0:
        raise RuntimeError("tpchgen: " + (_gen().gpuq_tpchgen_last_error() or b"").decode())


def _on_device(tc):
    import torch
    return torch.cuda.device(tc.device)
def gen_lineitem_device(tc, n, seed=SEED_LINEITEM, seed_orders=SEED_ORDERS, row0=0, n_supp=10_000,
                        columns=("l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate")):
    """Device-resident lineitem columns (Arrow physical layout) produced by the HIP generator."""
    import torch
    import arrow_ballista_amd as g
    from arrow_ballista_amd import binding as B
    dev = tc.device
    bufs, cs = {}, gpuq_lineitem_cols()
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
    with _on_device(tc):
        _gen_check(tc, _gen().gpuq_tpchgen_lineitem(tc.stream_ptr(), seed, seed_orders, row0, n, n_supp, C.byref(cs)))
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
    cs = gpuq_orders_cols(**p)
    with _on_device(tc):
        _gen_check(tc, _gen().gpuq_tpchgen_orders(tc.stream_ptr(), seed, row0, n, n_cust, C.byref(cs)))
    tc.sync()
    return g.DeviceTable(cols, n)


def gen_customer_device(tc, n, seed=SEED_CUSTOMER, row0=0):
    import arrow_ballista_amd as g
    from arrow_ballista_amd import binding as B
    assert n % 5 == 0 and row0 % 5 == 0
    cols, p = _dev_cols(tc, [("c_custkey", "Int64", 8), ("c_nationkey", "Int64", 8), ("c_mktsegment", "Utf8", ("utf8", n * 9))], n)
    cs = gpuq_customer_cols(**p)
    with _on_device(tc):
        _gen_check(tc, _gen().gpuq_tpchgen_customer(tc.stream_ptr(), seed, row0, n, C.byref(cs)))
    tc.sync()
    return g.DeviceTable(cols, n)


def gen_supplier_device(tc, n, seed=SEED_SUPPLIER, row0=0):
    import arrow_ballista_amd as g
    from arrow_ballista_amd import binding as B
    cols, p = _dev_cols(tc, [("s_suppkey", "Int64", 8), ("s_nationkey", "Int64", 8)], n)
    cs = gpuq_supplier_cols(**p)
    with _on_device(tc):
        _gen_check(tc, _gen().gpuq_tpchgen_supplier(tc.stream_ptr(), seed, row0, n, C.byref(cs)))
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


# ------------------------------------------------------------------ more of the harness's queries (benchmarks/queries/q{6,7,12,14,16,19,22}.sql)
# Physical plans in the shape DataFusion builds (build sides on the LEFT, filters below the joins, computed aggregate arguments in a
# ProjectionExec below the AggregateExec).  Sources are MemoryExec nodes over tables with the reference's column names and types
# (benchmarks/src/bin/tpch.rs:864-957); tests/test_gpu_tpch_more.py runs them over small generated tables against plain Python.
D_1994, D_1995, D_1995_09, D_1995_10, D_1996_12_31, D_1995_01 = 8766, 9131, 9374, 9404, 9861, 9131


def _dec_lit(unscaled, p=15, s=2):
    from arrow_ballista_amd.expr import lit
    return lit(unscaled, ("Decimal128", p, s))


def _revenue(schema):
    from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
    return binary(col("l_extendedprice", schema), Op.Multiply, binary(lit(1, ("Decimal128", 20, 0)), Op.Minus, col("l_discount", schema)))


def q6_plan(lineitem):
    """q6.sql: one filter, SUM(l_extendedprice * l_discount)."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, Operator as Op
    s = lineitem.schema()
    pred = and_(binary(col("l_shipdate", s), Op.GtEq, lit(D_1994, "Date32")), binary(col("l_shipdate", s), Op.Lt, lit(D_1995, "Date32")),
                binary(col("l_discount", s), Op.GtEq, _dec_lit(5)), binary(col("l_discount", s), Op.LtEq, _dec_lit(7)),
                binary(col("l_quantity", s), Op.Lt, _dec_lit(2400)))
    f = g.FilterExec(pred, lineitem)
    return g.AggregateExec("Single", [], [{"fn": "SUM", "expr": binary(col("l_extendedprice", s), Op.Multiply, col("l_discount", s)), "name": "revenue"}], g.CoalesceBatchesExec(f))


def q12_plan(orders, lineitem):
    """q12.sql: lineitem |x| orders, two SUM(CASE ..) by l_shipmode, ORDER BY l_shipmode."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, or_, in_list, case, Operator as Op
    os_, ls = orders.schema(), lineitem.schema()
    pred = and_(in_list(col("l_shipmode", ls), [lit("MAIL"), lit("SHIP")]), binary(col("l_commitdate", ls), Op.Lt, col("l_receiptdate", ls)),
                binary(col("l_shipdate", ls), Op.Lt, col("l_commitdate", ls)), binary(col("l_receiptdate", ls), Op.GtEq, lit(D_1994, "Date32")),
                binary(col("l_receiptdate", ls), Op.Lt, lit(D_1995, "Date32")))
    l = g.FilterExec(pred, lineitem)
    j = g.HashJoinExec(orders, g.CoalesceBatchesExec(l), [(col("o_orderkey", os_), col("l_orderkey", ls))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    urgent = or_(binary(col("o_orderpriority", js), Op.Eq, lit("1-URGENT")), binary(col("o_orderpriority", js), Op.Eq, lit("2-HIGH")))
    not_urgent = and_(binary(col("o_orderpriority", js), Op.NotEq, lit("1-URGENT")), binary(col("o_orderpriority", js), Op.NotEq, lit("2-HIGH")))
    proj = g.ProjectionExec([(col("l_shipmode", js), "l_shipmode"), (case([(urgent, lit(1))], lit(0)), "hi"), (case([(not_urgent, lit(1))], lit(0)), "lo")], j)
    ps = proj.schema()
    agg = g.AggregateExec("Single", [(col("l_shipmode", ps), "l_shipmode")],
                          [{"fn": "SUM", "expr": col("hi", ps), "name": "high_line_count"}, {"fn": "SUM", "expr": col("lo", ps), "name": "low_line_count"}], proj)
    as_ = agg.schema()
    return g.SortExec([{"expr": col("l_shipmode", as_), "asc": True, "nulls_first": False}], agg)


def q14_plan(part, lineitem):
    """q14.sql: lineitem |x| part, the two sums (promo revenue and revenue) and 100.00 * promo / total."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, like, case, Operator as Op
    ps_, ls = part.schema(), lineitem.schema()
    l = g.FilterExec(and_(binary(col("l_shipdate", ls), Op.GtEq, lit(D_1995_09, "Date32")), binary(col("l_shipdate", ls), Op.Lt, lit(D_1995_10, "Date32"))), lineitem)
    j = g.HashJoinExec(part, g.CoalesceBatchesExec(l), [(col("p_partkey", ps_), col("l_partkey", ls))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    rev = _revenue(js)
    proj = g.ProjectionExec([(case([(like(col("p_type", js), "PROMO%"), rev)], lit(0, ("Decimal128", 38, 4))), "promo"), (rev, "rev")], j)
    ps = proj.schema()
    agg = g.AggregateExec("Single", [], [{"fn": "SUM", "expr": col("promo", ps), "name": "promo_sum"}, {"fn": "SUM", "expr": col("rev", ps), "name": "rev_sum"}], proj)
    as_ = agg.schema()
    ratio = binary(binary(lit(10000, ("Decimal128", 5, 2)), Op.Multiply, col("promo_sum", as_)), Op.Divide, col("rev_sum", as_))
    return g.ProjectionExec([(ratio, "promo_revenue"), (col("promo_sum", as_), "promo_sum"), (col("rev_sum", as_), "rev_sum")], agg)


def q19_plan(part, lineitem):
    """q19.sql: lineitem |x| part on the key; the three OR-ed groups are the join's filter (JoinFilter over both sides' columns);
    l_shipinstruct = 'DELIVER IN PERSON' is a literal beyond 15 bytes (runs as LIKE without wildcards)."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, or_, in_list, Operator as Op
    ps_, ls = part.schema(), lineitem.schema()
    # what all three groups share goes below the join (DataFusion pushes it there)
    l = g.FilterExec(and_(in_list(col("l_shipmode", ls), [lit("AIR"), lit("AIR REG")]), binary(col("l_shipinstruct", ls), Op.Eq, lit("DELIVER IN PERSON"))), lineitem)
    j0 = g.HashJoinExec(part, g.CoalesceBatchesExec(l), [(col("p_partkey", ps_), col("l_partkey", ls))], None, "Inner", "CollectLeft", False)
    js = j0.schema()

    def grp(brand, containers, qlo, size_hi):
        return and_(binary(col("p_brand", js), Op.Eq, lit(brand)), in_list(col("p_container", js), [lit(c) for c in containers]),
                    binary(col("l_quantity", js), Op.GtEq, _dec_lit(qlo * 100)), binary(col("l_quantity", js), Op.LtEq, _dec_lit((qlo + 10) * 100)),
                    binary(col("p_size", js), Op.GtEq, lit(1, "Int32")), binary(col("p_size", js), Op.LtEq, lit(size_hi, "Int32")))
    filt = or_(grp("Brand#12", ("SM CASE", "SM BOX", "SM PACK", "SM PKG"), 1, 5), grp("Brand#23", ("MED BAG", "MED BOX", "MED PKG", "MED PACK"), 10, 10),
               grp("Brand#34", ("LG CASE", "LG BOX", "LG PACK", "LG PKG"), 20, 15))
    j = g.HashJoinExec(part, g.CoalesceBatchesExec(l), [(col("p_partkey", ps_), col("l_partkey", ls))], filt, "Inner", "CollectLeft", False)
    return g.AggregateExec("Single", [], [{"fn": "SUM", "expr": _revenue(js), "name": "revenue"}], j)


def q16_plan(supplier, part, partsupp):
    """q16.sql: partsupp |x| part, NOT IN (complaining suppliers) as a RightAnti join, COUNT(DISTINCT ps_suppkey) by (p_brand, p_type,
    p_size), ORDER BY supplier_cnt DESC, p_brand, p_type, p_size.  p_type is longer than 15 bytes: dictionary-coded group keys."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, in_list, like, Operator as Op
    ss, ps_, pss = supplier.schema(), part.schema(), partsupp.schema()
    p = g.FilterExec(and_(binary(col("p_brand", ps_), Op.NotEq, lit("Brand#45")), like(col("p_type", ps_), "MEDIUM POLISHED%", negated=True),
                          in_list(col("p_size", ps_), [lit(v, "Int32") for v in (49, 14, 23, 45, 19, 3, 36, 9)])), part)
    j1 = g.HashJoinExec(g.CoalesceBatchesExec(p), partsupp, [(col("p_partkey", ps_), col("ps_partkey", pss))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    bad = g.ProjectionExec([(col("s_suppkey", ss), "s_suppkey")], g.FilterExec(like(col("s_comment", ss), "%Customer%Complaints%"), supplier))
    bs = bad.schema()
    j2 = g.HashJoinExec(bad, j1, [(col("s_suppkey", bs), col("ps_suppkey", j1s))], None, "RightAnti", "CollectLeft", False)
    j2s = j2.schema()
    agg = g.AggregateExec("Single", [(col("p_brand", j2s), "p_brand"), (col("p_type", j2s), "p_type"), (col("p_size", j2s), "p_size")],
                          [{"fn": "COUNT", "expr": col("ps_suppkey", j2s), "name": "supplier_cnt", "distinct": True}], j2)
    as_ = agg.schema()
    return g.SortExec([{"expr": col("supplier_cnt", as_), "asc": False, "nulls_first": True}, {"expr": col("p_brand", as_), "asc": True, "nulls_first": False},
                       {"expr": col("p_type", as_), "asc": True, "nulls_first": False}, {"expr": col("p_size", as_), "asc": True, "nulls_first": False}], agg)


Q22_CODES = ("13", "31", "23", "29", "30", "18", "17")


def q22_avg_plan(customer):
    """q22.sql's scalar subquery: AVG(c_acctbal) over the positive balances of the seven country codes (run first; its value is a
    literal of the main plan -- the reference's planner turns the uncorrelated scalar subquery into a join with a one-row side)."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, in_list, substr, Operator as Op
    cs = customer.schema()
    f = g.FilterExec(and_(binary(col("c_acctbal", cs), Op.Gt, _dec_lit(0)), in_list(substr(col("c_phone", cs), 1, 2), [lit(c) for c in Q22_CODES])), customer)
    return g.AggregateExec("Single", [], [{"fn": "AVG", "expr": col("c_acctbal", cs), "name": "avg_bal"}], g.CoalesceBatchesExec(f))


def q22_plan(orders, customer, avg_unscaled):
    """q22.sql with the subquery's value in (Decimal128(19,6) unscaled): customers of the seven country codes with an above-average
    balance and NO order (RightAnti against orders' customer keys), COUNT(*) and SUM(c_acctbal) by country code."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, in_list, substr, cast, Operator as Op
    os_, cs = orders.schema(), customer.schema()
    code = substr(col("c_phone", cs), 1, 2)
    f = g.FilterExec(and_(in_list(code, [lit(c) for c in Q22_CODES]),
                          binary(cast(col("c_acctbal", cs), ("Decimal128", 19, 6)), Op.Gt, lit(avg_unscaled, ("Decimal128", 19, 6)))), customer)
    okeys = g.ProjectionExec([(col("o_custkey", os_), "o_custkey")], orders)
    ks = okeys.schema()
    j = g.HashJoinExec(okeys, g.CoalesceBatchesExec(f), [(col("o_custkey", ks), col("c_custkey", cs))], None, "RightAnti", "CollectLeft", False)
    js = j.schema()
    proj = g.ProjectionExec([(substr(col("c_phone", js), 1, 2), "cntrycode"), (col("c_acctbal", js), "c_acctbal")], j)
    ps = proj.schema()
    agg = g.AggregateExec("Single", [(col("cntrycode", ps), "cntrycode")], [{"fn": "COUNT", "expr": lit(1), "name": "numcust"}, {"fn": "SUM", "expr": col("c_acctbal", ps), "name": "totacctbal"}], proj)
    as_ = agg.schema()
    return g.SortExec([{"expr": col("cntrycode", as_), "asc": True, "nulls_first": False}], agg)


def q7_plan(supplier, lineitem, orders, customer, nation):
    """q7.sql: supplier |x| lineitem |x| orders |x| customer |x| nation n1 |x| nation n2, the (FRANCE, GERMANY) pair either way round,
    extract(year from l_shipdate) as a group key (date_part -> Float64), SUM(volume), ORDER BY the three keys."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, or_, in_list, date_part, Operator as Op
    ss, ls, os_, cs, ns = supplier.schema(), lineitem.schema(), orders.schema(), customer.schema(), nation.schema()
    two = in_list(col("n_name", ns), [lit("FRANCE"), lit("GERMANY")])
    n1 = g.ProjectionExec([(col("n_nationkey", ns), "n1_key"), (col("n_name", ns), "supp_nation")], g.FilterExec(two, nation))
    n2 = g.ProjectionExec([(col("n_nationkey", ns), "n2_key"), (col("n_name", ns), "cust_nation")], g.FilterExec(two, nation))
    n1s, n2s = n1.schema(), n2.schema()
    sj = g.HashJoinExec(n1, supplier, [(col("n1_key", n1s), col("s_nationkey", ss))], None, "Inner", "CollectLeft", False)          # suppliers of the two nations
    sjs = sj.schema()
    l = g.FilterExec(and_(binary(col("l_shipdate", ls), Op.GtEq, lit(D_1995_01, "Date32")), binary(col("l_shipdate", ls), Op.LtEq, lit(D_1996_12_31, "Date32"))), lineitem)
    lj = g.HashJoinExec(sj, g.CoalesceBatchesExec(l), [(col("s_suppkey", sjs), col("l_suppkey", ls))], None, "Inner", "CollectLeft", False)
    ljs = lj.schema()
    cj = g.HashJoinExec(n2, customer, [(col("n2_key", n2s), col("c_nationkey", cs))], None, "Inner", "CollectLeft", False)          # customers of the two nations
    cjs = cj.schema()
    oj = g.HashJoinExec(cj, orders, [(col("c_custkey", cjs), col("o_custkey", os_))], None, "Inner", "CollectLeft", False)
    ojs = oj.schema()
    pair = or_(and_(binary(col("supp_nation", ljs), Op.Eq, lit("FRANCE")), binary(col("cust_nation", ojs), Op.Eq, lit("GERMANY"))),
               and_(binary(col("supp_nation", ljs), Op.Eq, lit("GERMANY")), binary(col("cust_nation", ojs), Op.Eq, lit("FRANCE"))))
    j = g.HashJoinExec(oj, lj, [(col("o_orderkey", ojs), col("l_orderkey", ljs))], pair, "Inner", "CollectLeft", False)
    js = j.schema()
    proj = g.ProjectionExec([(col("supp_nation", js), "supp_nation"), (col("cust_nation", js), "cust_nation"), (date_part("YEAR", col("l_shipdate", js)), "l_year"),
                             (_revenue(js), "volume")], j)
    ps = proj.schema()
    agg = g.AggregateExec("Single", [(col(n, ps), n) for n in ("supp_nation", "cust_nation", "l_year")], [{"fn": "SUM", "expr": col("volume", ps), "name": "revenue"}], proj)
    as_ = agg.schema()
    return g.SortExec([{"expr": col(n, as_), "asc": True, "nulls_first": False} for n in ("supp_nation", "cust_nation", "l_year")], agg)


D_1993_07, D_1993_10 = 8582, 8674


def q4_plan(orders, lineitem):
    """q4.sql: orders of one quarter that have a late lineitem (EXISTS -> semi join: lineitem is the build side, the surviving ORDERS
    rows are the probe side: RightSemi), COUNT(*) by o_orderpriority, ORDER BY o_orderpriority."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, Operator as Op
    os_, ls = orders.schema(), lineitem.schema()
    o = g.FilterExec(and_(binary(col("o_orderdate", os_), Op.GtEq, lit(D_1993_07, "Date32")), binary(col("o_orderdate", os_), Op.Lt, lit(D_1993_10, "Date32"))), orders)
    l = g.ProjectionExec([(col("l_orderkey", ls), "l_orderkey")], g.FilterExec(binary(col("l_commitdate", ls), Op.Lt, col("l_receiptdate", ls)), lineitem))
    lks = l.schema()
    j = g.HashJoinExec(l, g.CoalesceBatchesExec(o), [(col("l_orderkey", lks), col("o_orderkey", os_))], None, "RightSemi", "CollectLeft", False)
    js = j.schema()
    agg = g.AggregateExec("Single", [(col("o_orderpriority", js), "o_orderpriority")], [{"fn": "COUNT", "expr": lit(1), "name": "order_count"}], j)
    as_ = agg.schema()
    return g.SortExec([{"expr": col("o_orderpriority", as_), "asc": True, "nulls_first": False}], agg)


def q13_plan(customer, orders):
    """q13.sql: customer LEFT JOIN orders (the ON clause's NOT LIKE filters the orders side), COUNT(o_orderkey) per customer, then the
    distribution of those counts; ORDER BY custdist DESC, c_count DESC."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, like
    cs, os_ = customer.schema(), orders.schema()
    o = g.FilterExec(like(col("o_comment", os_), "%special%requests%", negated=True), orders)
    ck = g.ProjectionExec([(col("c_custkey", cs), "c_custkey")], customer)
    cks = ck.schema()
    j = g.HashJoinExec(ck, g.CoalesceBatchesExec(o), [(col("c_custkey", cks), col("o_custkey", os_))], None, "Left", "CollectLeft", False)
    js = j.schema()
    per = g.AggregateExec("Single", [(col("c_custkey", js), "c_custkey")], [{"fn": "COUNT", "expr": col("o_orderkey", js), "name": "c_count"}], j)
    ps = per.schema()
    dist = g.AggregateExec("Single", [(col("c_count", ps), "c_count")], [{"fn": "COUNT", "expr": lit(1), "name": "custdist"}], per)
    ds = dist.schema()
    return g.SortExec([{"expr": col("custdist", ds), "asc": False, "nulls_first": True}, {"expr": col("c_count", ds), "asc": False, "nulls_first": True}], dist)


def q9_plan(part, supplier, lineitem, partsupp, orders, nation):
    """q9.sql: the green parts' lineitems joined to supplier / nation, partsupp (two-column key) and orders; profit =
    l_extendedprice * (1 - l_discount) - ps_supplycost * l_quantity by (nation, extract(year from o_orderdate)); ORDER BY nation, o_year DESC."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, binary, like, date_part, Operator as Op
    ps_, ss, ls, pss, os_, ns = part.schema(), supplier.schema(), lineitem.schema(), partsupp.schema(), orders.schema(), nation.schema()
    p = g.ProjectionExec([(col("p_partkey", ps_), "p_partkey")], g.FilterExec(like(col("p_name", ps_), "%green%"), part))
    pks = p.schema()
    j1 = g.HashJoinExec(p, lineitem, [(col("p_partkey", pks), col("l_partkey", ls))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    sn = g.HashJoinExec(nation, supplier, [(col("n_nationkey", ns), col("s_nationkey", ss))], None, "Inner", "CollectLeft", False)
    sns = sn.schema()
    j2 = g.HashJoinExec(sn, j1, [(col("s_suppkey", sns), col("l_suppkey", j1s))], None, "Inner", "CollectLeft", False)
    j2s = j2.schema()
    j3 = g.HashJoinExec(partsupp, j2, [(col("ps_suppkey", pss), col("l_suppkey", j2s)), (col("ps_partkey", pss), col("l_partkey", j2s))], None, "Inner", "CollectLeft", False)
    j3s = j3.schema()
    ok = g.ProjectionExec([(col("o_orderkey", os_), "o_orderkey"), (col("o_orderdate", os_), "o_orderdate")], orders)
    oks = ok.schema()
    j4 = g.HashJoinExec(ok, j3, [(col("o_orderkey", oks), col("l_orderkey", j3s))], None, "Inner", "CollectLeft", False)
    j4s = j4.schema()
    amount = binary(_revenue(j4s), Op.Minus, binary(col("ps_supplycost", j4s), Op.Multiply, col("l_quantity", j4s)))
    proj = g.ProjectionExec([(col("n_name", j4s), "nation"), (date_part("YEAR", col("o_orderdate", j4s)), "o_year"), (amount, "amount")], j4)
    prs = proj.schema()
    agg = g.AggregateExec("Single", [(col("nation", prs), "nation"), (col("o_year", prs), "o_year")], [{"fn": "SUM", "expr": col("amount", prs), "name": "sum_profit"}], proj)
    as_ = agg.schema()
    return g.SortExec([{"expr": col("nation", as_), "asc": True, "nulls_first": False}, {"expr": col("o_year", as_), "asc": False, "nulls_first": True}], agg)


D_1993_10_01, D_1994_01_01 = 8674, 8766


def q10_plan(customer, orders, lineitem, nation):
    """q10.sql: returned items of one quarter: customer |x| orders |x| lineitem |x| nation, SUM(revenue) grouped by SEVEN columns (c_custkey,
    c_name, c_acctbal, c_phone, n_name, c_address, c_comment: more than the aggregate's table holds keys -- the narrow ones are packed,
    the strings travel as dictionary codes), ORDER BY revenue DESC."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, Operator as Op
    cs, os_, ls, ns = customer.schema(), orders.schema(), lineitem.schema(), nation.schema()
    o = g.FilterExec(and_(binary(col("o_orderdate", os_), Op.GtEq, lit(D_1993_10_01, "Date32")), binary(col("o_orderdate", os_), Op.Lt, lit(D_1994_01_01, "Date32"))), orders)
    l = g.FilterExec(binary(col("l_returnflag", ls), Op.Eq, lit("R")), lineitem)
    nc = g.HashJoinExec(nation, customer, [(col("n_nationkey", ns), col("c_nationkey", cs))], None, "Inner", "CollectLeft", False)
    ncs = nc.schema()
    co = g.HashJoinExec(nc, g.CoalesceBatchesExec(o), [(col("c_custkey", ncs), col("o_custkey", os_))], None, "Inner", "CollectLeft", False)
    cos = co.schema()
    j = g.HashJoinExec(co, g.CoalesceBatchesExec(l), [(col("o_orderkey", cos), col("l_orderkey", ls))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    keys = ("c_custkey", "c_name", "c_acctbal", "c_phone", "n_name", "c_address", "c_comment")
    proj = g.ProjectionExec([(col(k, js), k) for k in keys] + [(_revenue(js), "rev")], j)
    ps = proj.schema()
    agg = g.AggregateExec("Single", [(col(k, ps), k) for k in keys], [{"fn": "SUM", "expr": col("rev", ps), "name": "revenue"}], proj)
    as_ = agg.schema()
    out = g.ProjectionExec([(col(k, as_), k) for k in ("c_custkey", "c_name", "revenue", "c_acctbal", "n_name", "c_address", "c_phone", "c_comment")], agg)
    return g.SortExec([{"expr": col("revenue", out.schema()), "asc": False, "nulls_first": True}], out)


def q18_plan(customer, orders, lineitem, quantity_unscaled=30000):
    """q18.sql: large-volume customers: o_orderkey IN (SELECT l_orderkey .. GROUP BY l_orderkey HAVING SUM(l_quantity) > 300) is a semi join with the
    filtered aggregate as its build side; then customer |x| orders |x| lineitem, SUM(l_quantity) grouped by FIVE columns, ORDER BY
    o_totalprice DESC, o_orderdate."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
    cs, os_, ls = customer.schema(), orders.schema(), lineitem.schema()
    per = g.AggregateExec("Single", [(col("l_orderkey", ls), "l_orderkey")], [{"fn": "SUM", "expr": col("l_quantity", ls), "name": "q"}], lineitem)
    pers = per.schema()
    big = g.ProjectionExec([(col("l_orderkey", pers), "big_orderkey")], g.FilterExec(binary(col("q", pers), Op.Gt, lit(quantity_unscaled, ("Decimal128", 25, 2))), per))
    bs = big.schema()
    o = g.HashJoinExec(big, orders, [(col("big_orderkey", bs), col("o_orderkey", os_))], None, "RightSemi", "CollectLeft", False)
    oss = o.schema()
    co = g.HashJoinExec(customer, o, [(col("c_custkey", cs), col("o_custkey", oss))], None, "Inner", "CollectLeft", False)
    cos = co.schema()
    j = g.HashJoinExec(co, lineitem, [(col("o_orderkey", cos), col("l_orderkey", ls))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    keys = ("c_name", "c_custkey", "o_orderkey", "o_orderdate", "o_totalprice")
    agg = g.AggregateExec("Single", [(col(k, js), k) for k in keys], [{"fn": "SUM", "expr": col("l_quantity", js), "name": "sum_qty"}], j)
    as_ = agg.schema()
    return g.SortExec([{"expr": col("o_totalprice", as_), "asc": False, "nulls_first": True}, {"expr": col("o_orderdate", as_), "asc": True, "nulls_first": False}], agg)


def q8_plan(part, supplier, lineitem, orders, customer, nation, region):
    """q8.sql: market share of BRAZIL within AMERICA for one part type, by extract(year from o_orderdate): eight tables,
    SUM(CASE WHEN nation = 'BRAZIL' THEN volume ELSE 0 END) / SUM(volume) (decimal division), ORDER BY o_year."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, case, date_part, Operator as Op
    ps_, ss, ls, os_, cs, ns, rs = part.schema(), supplier.schema(), lineitem.schema(), orders.schema(), customer.schema(), nation.schema(), region.schema()
    p = g.ProjectionExec([(col("p_partkey", ps_), "p_partkey")], g.FilterExec(binary(col("p_type", ps_), Op.Eq, lit("ECONOMY ANODIZED STEEL")), part))
    pl = g.HashJoinExec(p, lineitem, [(col("p_partkey", p.schema()), col("l_partkey", ls))], None, "Inner", "CollectLeft", False)
    pls = pl.schema()
    n2 = g.ProjectionExec([(col("n_nationkey", ns), "n2_key"), (col("n_name", ns), "nation")], nation)
    sn = g.HashJoinExec(n2, supplier, [(col("n2_key", n2.schema()), col("s_nationkey", ss))], None, "Inner", "CollectLeft", False)
    sns = sn.schema()
    sl = g.HashJoinExec(sn, pl, [(col("s_suppkey", sns), col("l_suppkey", pls))], None, "Inner", "CollectLeft", False)
    sls = sl.schema()
    r = g.FilterExec(binary(col("r_name", rs), Op.Eq, lit("AMERICA")), region)
    n1 = g.HashJoinExec(r, nation, [(col("r_regionkey", rs), col("n_regionkey", ns))], None, "Inner", "CollectLeft", False)
    n1s = n1.schema()
    n1k = g.ProjectionExec([(col("n_nationkey", n1s), "n1_key")], n1)
    c = g.HashJoinExec(n1k, customer, [(col("n1_key", n1k.schema()), col("c_nationkey", cs))], None, "RightSemi", "CollectLeft", False)      # customers of AMERICA
    cks = c.schema()
    o = g.FilterExec(and_(binary(col("o_orderdate", os_), Op.GtEq, lit(D_1995_01, "Date32")), binary(col("o_orderdate", os_), Op.LtEq, lit(D_1996_12_31, "Date32"))), orders)
    co = g.HashJoinExec(g.ProjectionExec([(col("c_custkey", cks), "c_custkey")], c), g.CoalesceBatchesExec(o), [(col("c_custkey", cks), col("o_custkey", os_))], None, "RightSemi", "CollectLeft", False)
    cos = co.schema()
    j = g.HashJoinExec(co, sl, [(col("o_orderkey", cos), col("l_orderkey", sls))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    proj = g.ProjectionExec([(date_part("YEAR", col("o_orderdate", js)), "o_year"), (_revenue(js), "volume"), (col("nation", js), "nation")], j)
    pj = proj.schema()
    zero = lit(0, ("Decimal128", 38, 4))
    brazil = case([(binary(col("nation", pj), Op.Eq, lit("BRAZIL")), col("volume", pj))], zero)
    agg = g.AggregateExec("Single", [(col("o_year", pj), "o_year")], [{"fn": "SUM", "expr": brazil, "name": "brazil"}, {"fn": "SUM", "expr": col("volume", pj), "name": "total"}], proj)
    as_ = agg.schema()
    share = g.ProjectionExec([(col("o_year", as_), "o_year"), (binary(col("brazil", as_), Op.Divide, col("total", as_)), "mkt_share"), (col("brazil", as_), "brazil"), (col("total", as_), "total")], agg)
    return g.SortExec([{"expr": col("o_year", share.schema()), "asc": True, "nulls_first": False}], share)


def _europe_suppliers(supplier, nation, region, region_name):
    """region(r_name) |x| nation |x| supplier: the suppliers of one region with their nation's name"""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
    ss, ns, rs = supplier.schema(), nation.schema(), region.schema()
    r = g.ProjectionExec([(col("r_regionkey", rs), "r_regionkey")], g.FilterExec(binary(col("r_name", rs), Op.Eq, lit(region_name)), region))
    rn = g.HashJoinExec(r, nation, [(col("r_regionkey", r.schema()), col("n_regionkey", ns))], None, "Inner", "CollectLeft", False)
    rns = rn.schema()
    rnk = g.ProjectionExec([(col("n_nationkey", rns), "n_nationkey"), (col("n_name", rns), "n_name")], rn)
    return g.HashJoinExec(rnk, supplier, [(col("n_nationkey", rnk.schema()), col("s_nationkey", ss))], None, "Inner", "CollectLeft", False)


def q2_plan(part, supplier, partsupp, nation, region, size=15, type_suffix="BRASS", region_name="EUROPE"):
    """q2.sql: minimum-cost supplier.  The correlated scalar subquery (min(ps_supplycost) over the region's suppliers of THIS part) is what
    DataFusion's decorrelation makes of it: MIN grouped by ps_partkey, joined back on (p_partkey, ps_supplycost = min)."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, like, Operator as Op
    ps_, pss = part.schema(), partsupp.schema()
    es = _europe_suppliers(supplier, nation, region, region_name)
    ess = es.schema()
    pse = g.HashJoinExec(es, partsupp, [(col("s_suppkey", ess), col("ps_suppkey", pss))], None, "Inner", "CollectLeft", False)      # the region's offers
    pses = pse.schema()
    minc = g.AggregateExec("Single", [(col("ps_partkey", pses), "min_partkey")], [{"fn": "MIN", "expr": col("ps_supplycost", pses), "name": "min_cost"}], pse)
    ms = minc.schema()
    p = g.FilterExec(and_(binary(col("p_size", ps_), Op.Eq, lit(size, "Int32")), like(col("p_type", ps_), "%" + type_suffix)), part)
    pk = g.ProjectionExec([(col("p_partkey", ps_), "p_partkey"), (col("p_mfgr", ps_), "p_mfgr")], p)
    j1 = g.HashJoinExec(pk, pse, [(col("p_partkey", pk.schema()), col("ps_partkey", pses))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    j2 = g.HashJoinExec(minc, j1, [(col("min_partkey", ms), col("p_partkey", j1s)), (col("min_cost", ms), col("ps_supplycost", j1s))], None, "Inner", "CollectLeft", False)
    j2s = j2.schema()
    out = g.ProjectionExec([(col(n, j2s), n) for n in ("s_acctbal", "s_name", "n_name", "p_partkey", "p_mfgr", "s_address", "s_phone", "s_comment")], j2)
    os_ = out.schema()
    return g.SortExec([{"expr": col("s_acctbal", os_), "asc": False, "nulls_first": True}, {"expr": col("n_name", os_), "asc": True, "nulls_first": False},
                       {"expr": col("s_name", os_), "asc": True, "nulls_first": False}, {"expr": col("p_partkey", os_), "asc": True, "nulls_first": False}], out)


def q11_plan(partsupp, supplier, nation, nation_name="GERMANY", fraction_unscaled=1):
    """q11.sql: important stock: SUM(ps_supplycost * ps_availqty) per part of one nation's suppliers HAVING it above 0.0001 of the nation's total --
    the uncorrelated scalar subquery is a one-row aggregate, CROSS JOINed to the groups and filtered (DataFusion's plan shape).  The
    fraction is a decimal literal here (0.0001 = 1 at scale 4), so the comparison is exact."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
    pss, ss, ns = partsupp.schema(), supplier.schema(), nation.schema()
    n = g.ProjectionExec([(col("n_nationkey", ns), "n_nationkey")], g.FilterExec(binary(col("n_name", ns), Op.Eq, lit(nation_name)), nation))
    sn = g.HashJoinExec(n, supplier, [(col("n_nationkey", n.schema()), col("s_nationkey", ss))], None, "Inner", "CollectLeft", False)
    sk = g.ProjectionExec([(col("s_suppkey", sn.schema()), "s_suppkey")], sn)
    j = g.HashJoinExec(sk, partsupp, [(col("s_suppkey", sk.schema()), col("ps_suppkey", pss))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    v = g.ProjectionExec([(col("ps_partkey", js), "ps_partkey"), (binary(col("ps_supplycost", js), Op.Multiply, col("ps_availqty", js)), "v")], j)
    vs = v.schema()
    per = g.AggregateExec("Single", [(col("ps_partkey", vs), "ps_partkey")], [{"fn": "SUM", "expr": col("v", vs), "name": "value"}], v)
    tot = g.AggregateExec("Single", [], [{"fn": "SUM", "expr": col("v", vs), "name": "total"}], v)
    thr = g.ProjectionExec([(binary(col("total", tot.schema()), Op.Multiply, lit(fraction_unscaled, ("Decimal128", 5, 4))), "threshold")], tot)
    cj = g.CrossJoinExec(thr, per)
    cs = cj.schema()
    f = g.FilterExec(binary(col("value", cs), Op.Gt, col("threshold", cs)), cj)
    out = g.ProjectionExec([(col("ps_partkey", cs), "ps_partkey"), (col("value", cs), "value")], f)
    return g.SortExec([{"expr": col("value", out.schema()), "asc": False, "nulls_first": True}], out)


D_1996_01, D_1996_04 = 9496, 9587


def q15_plan(supplier, lineitem):
    """q15.sql: top supplier: the view revenue0 (SUM(revenue) by l_suppkey over one quarter) is computed, its MAX is the scalar subquery,
    `total_revenue = (select max ..)` an inner join on that one row; ORDER BY s_suppkey."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, Operator as Op
    ss, ls = supplier.schema(), lineitem.schema()
    l = g.FilterExec(and_(binary(col("l_shipdate", ls), Op.GtEq, lit(D_1996_01, "Date32")), binary(col("l_shipdate", ls), Op.Lt, lit(D_1996_04, "Date32"))), lineitem)
    lp = g.ProjectionExec([(col("l_suppkey", ls), "supplier_no"), (_revenue(ls), "rev")], l)
    rev = g.AggregateExec("Single", [(col("supplier_no", lp.schema()), "supplier_no")], [{"fn": "SUM", "expr": col("rev", lp.schema()), "name": "total_revenue"}], lp)
    rs = rev.schema()
    mx = g.AggregateExec("Single", [], [{"fn": "MAX", "expr": col("total_revenue", rs), "name": "max_revenue"}], rev)
    top = g.HashJoinExec(mx, rev, [(col("max_revenue", mx.schema()), col("total_revenue", rs))], None, "Inner", "CollectLeft", False)
    ts = top.schema()
    tp = g.ProjectionExec([(col("supplier_no", ts), "supplier_no"), (col("total_revenue", ts), "total_revenue")], top)
    j = g.HashJoinExec(tp, supplier, [(col("supplier_no", tp.schema()), col("s_suppkey", ss))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    out = g.ProjectionExec([(col(n, js), n) for n in ("s_suppkey", "s_name", "s_address", "s_phone", "total_revenue")], j)
    return g.SortExec([{"expr": col("s_suppkey", out.schema()), "asc": True, "nulls_first": False}], out)


def q17_plan(lineitem, part, brand="Brand#23", container="MED BOX"):
    """q17.sql: small-quantity-order revenue: the correlated subquery 0.2 * avg(l_quantity) per part is AVG grouped by l_partkey joined back
    on the part key, `l_quantity < 0.2 * avg` the join's filter; SUM(l_extendedprice) / 7.0."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, Operator as Op
    ls, ps_ = lineitem.schema(), part.schema()
    avgq = g.AggregateExec("Single", [(col("l_partkey", ls), "avg_partkey")], [{"fn": "AVG", "expr": col("l_quantity", ls), "name": "avg_qty"}], lineitem)
    as_ = avgq.schema()
    p = g.ProjectionExec([(col("p_partkey", ps_), "p_partkey")], g.FilterExec(and_(binary(col("p_brand", ps_), Op.Eq, lit(brand)), binary(col("p_container", ps_), Op.Eq, lit(container))), part))
    j1 = g.HashJoinExec(p, lineitem, [(col("p_partkey", p.schema()), col("l_partkey", ls))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    small = binary(col("l_quantity", j1s), Op.Lt, binary(lit(2, ("Decimal128", 2, 1)), Op.Multiply, col("avg_qty", as_)))
    j2 = g.HashJoinExec(avgq, j1, [(col("avg_partkey", as_), col("p_partkey", j1s))], small, "Inner", "CollectLeft", False)
    agg = g.AggregateExec("Single", [], [{"fn": "SUM", "expr": col("l_extendedprice", j2.schema()), "name": "s"}], j2)
    return g.ProjectionExec([(binary(col("s", agg.schema()), Op.Divide, lit(70, ("Decimal128", 2, 1))), "avg_yearly"), (col("s", agg.schema()), "s")], agg)


def q20_plan(supplier, nation, partsupp, part, lineitem, prefix="forest", nation_name="CANADA"):
    """q20.sql: potential part promotion: nested IN subqueries are semi joins, the correlated 0.5 * sum(l_quantity) over (part, supplier) of one
    year an aggregate joined on both keys with `ps_availqty > 0.5 * sum` as the join filter; ORDER BY s_name."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, and_, like, Operator as Op
    ss, ns, pss, ps_, ls = supplier.schema(), nation.schema(), partsupp.schema(), part.schema(), lineitem.schema()
    p = g.ProjectionExec([(col("p_partkey", ps_), "p_partkey")], g.FilterExec(like(col("p_name", ps_), prefix + "%"), part))
    ps1 = g.HashJoinExec(p, partsupp, [(col("p_partkey", p.schema()), col("ps_partkey", pss))], None, "RightSemi", "CollectLeft", False)
    p1s = ps1.schema()
    l = g.FilterExec(and_(binary(col("l_shipdate", ls), Op.GtEq, lit(D_1994, "Date32")), binary(col("l_shipdate", ls), Op.Lt, lit(D_1995, "Date32"))), lineitem)
    lq = g.AggregateExec("Single", [(col("l_partkey", ls), "q_partkey"), (col("l_suppkey", ls), "q_suppkey")], [{"fn": "SUM", "expr": col("l_quantity", ls), "name": "q"}], l)
    qs = lq.schema()
    enough = binary(col("ps_availqty", p1s), Op.Gt, binary(lit(5, ("Decimal128", 2, 1)), Op.Multiply, col("q", qs)))
    ps2 = g.HashJoinExec(lq, ps1, [(col("q_partkey", qs), col("ps_partkey", p1s)), (col("q_suppkey", qs), col("ps_suppkey", p1s))], enough, "Inner", "CollectLeft", False)
    sk = g.ProjectionExec([(col("ps_suppkey", ps2.schema()), "ok_suppkey")], ps2)
    n = g.ProjectionExec([(col("n_nationkey", ns), "n_nationkey")], g.FilterExec(binary(col("n_name", ns), Op.Eq, lit(nation_name)), nation))
    sn = g.HashJoinExec(n, supplier, [(col("n_nationkey", n.schema()), col("s_nationkey", ss))], None, "Inner", "CollectLeft", False)
    sns = sn.schema()
    r = g.HashJoinExec(sk, sn, [(col("ok_suppkey", sk.schema()), col("s_suppkey", sns))], None, "RightSemi", "CollectLeft", False)
    out = g.ProjectionExec([(col("s_name", r.schema()), "s_name"), (col("s_address", r.schema()), "s_address")], r)
    return g.SortExec([{"expr": col("s_name", out.schema()), "asc": True, "nulls_first": False}], out)


def q21_plan(supplier, lineitem, orders, nation, nation_name="SAUDI ARABIA"):
    """q21.sql: suppliers who kept orders waiting: EXISTS (another supplier's lineitem in the order) is a semi join, NOT EXISTS (another supplier's
    LATE lineitem) an anti join, both on l_orderkey with `l_suppkey <> l1.l_suppkey` as the join filter; COUNT(*) by s_name, ORDER BY
    numwait DESC, s_name."""
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
    ss, ls, os_, ns = supplier.schema(), lineitem.schema(), orders.schema(), nation.schema()
    late = binary(col("l_receiptdate", ls), Op.Gt, col("l_commitdate", ls))
    n = g.ProjectionExec([(col("n_nationkey", ns), "n_nationkey")], g.FilterExec(binary(col("n_name", ns), Op.Eq, lit(nation_name)), nation))
    sn = g.HashJoinExec(n, supplier, [(col("n_nationkey", n.schema()), col("s_nationkey", ss))], None, "Inner", "CollectLeft", False)
    s = g.ProjectionExec([(col("s_suppkey", sn.schema()), "s_suppkey"), (col("s_name", sn.schema()), "s_name")], sn)
    l1 = g.ProjectionExec([(col("l_orderkey", ls), "l_orderkey"), (col("l_suppkey", ls), "l_suppkey")], g.FilterExec(late, lineitem))
    j = g.HashJoinExec(s, l1, [(col("s_suppkey", s.schema()), col("l_suppkey", l1.schema()))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    o = g.ProjectionExec([(col("o_orderkey", os_), "o_orderkey")], g.FilterExec(binary(col("o_orderstatus", os_), Op.Eq, lit("F")), orders))
    jo = g.HashJoinExec(o, j, [(col("o_orderkey", o.schema()), col("l_orderkey", js))], None, "RightSemi", "CollectLeft", False)
    jos = jo.schema()
    l2 = g.ProjectionExec([(col("l_orderkey", ls), "l2_orderkey"), (col("l_suppkey", ls), "l2_suppkey")], lineitem)
    l2s = l2.schema()
    ex = g.HashJoinExec(l2, jo, [(col("l2_orderkey", l2s), col("l_orderkey", jos))], binary(col("l2_suppkey", l2s), Op.NotEq, col("l_suppkey", jos)), "RightSemi", "CollectLeft", False)
    l3 = g.ProjectionExec([(col("l_orderkey", ls), "l3_orderkey"), (col("l_suppkey", ls), "l3_suppkey")], g.FilterExec(late, lineitem))
    l3s = l3.schema()
    nx = g.HashJoinExec(l3, ex, [(col("l3_orderkey", l3s), col("l_orderkey", ex.schema()))], binary(col("l3_suppkey", l3s), Op.NotEq, col("l_suppkey", ex.schema())), "RightAnti", "CollectLeft", False)
    agg = g.AggregateExec("Single", [(col("s_name", nx.schema()), "s_name")], [{"fn": "COUNT", "expr": lit(1), "name": "numwait"}], nx)
    as_ = agg.schema()
    return g.SortExec([{"expr": col("numwait", as_), "asc": False, "nulls_first": True}, {"expr": col("s_name", as_), "asc": True, "nulls_first": False}], agg)
