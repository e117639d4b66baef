"""Shared helpers for tests, __graft_entry__.smoke() and bench.py: synthetic TPC-H-shaped inputs
(device generator + the oracle's CPU restatement of it), the q1/q3/q5 operator plans written with
the reference's operator names, and result normalisation.  The oracle is used here only as the checker."""
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


# ------------------------------------------------------------------ oracle (C) loader
_ORACLE = None


def oracle_lib():
    global _ORACLE
    if _ORACLE is None:
        p = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(p):
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
        L = C.CDLL(p)
        L.oracle_num_threads.restype = C.c_int
        L.oracle_q1.restype = C.c_int
        L.oracle_join_build.restype = C.c_void_p
        L.oracle_join_build.argtypes = [C.c_void_p, C.c_int64]
        L.oracle_join_free.argtypes = [C.c_void_p]
        L.oracle_join_probe.restype = C.c_int64
        L.oracle_join_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        vp, i64, u64, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int32
        L.oracle_sort_u64.argtypes = [vp, i64, vp]
        L.oracle_partition_ids_i64.argtypes = [vp, i64, C.c_uint32, vp]
        L.oracle_gen_lineitem.argtypes = [u64, u64, i64, i64, i64] + [vp] * 11
        L.oracle_gen_orders.argtypes = [u64, i64, i64, i64] + [vp] * 4
        L.oracle_gen_customer.argtypes = [u64, i64, i64] + [vp] * 4
        L.oracle_gen_supplier.argtypes = [u64, i64, i64] + [vp] * 2
        L.oracle_q1.argtypes = [i64] + [vp] * 9 + [i32] + [vp] * 3
        _ORACLE = L
    return _ORACLE


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def gen_lineitem_host(n, seed=SEED_LINEITEM, seed_orders=SEED_ORDERS, row0=0, n_supp=10_000):
    """Oracle-side (CPU) restatement of the device generator: dict of numpy arrays in Arrow physical layout."""
    L = oracle_lib()
    d = dict(l_orderkey=np.empty(n, np.int64), l_suppkey=np.empty(n, np.int64),
             l_quantity=np.empty(2 * n, np.uint64), l_extendedprice=np.empty(2 * n, np.uint64),
             l_discount=np.empty(2 * n, np.uint64), l_tax=np.empty(2 * n, np.uint64),
             l_shipdate=np.empty(n, np.int32),
             l_returnflag=np.empty(max(n, 1), np.uint8), l_returnflag_off=np.empty(n + 1, np.int32),
             l_linestatus=np.empty(max(n, 1), np.uint8), l_linestatus_off=np.empty(n + 1, np.int32))
    L.oracle_gen_lineitem(C.c_uint64(seed), C.c_uint64(seed_orders), C.c_int64(row0), C.c_int64(n), C.c_int64(n_supp),
                          _p(d["l_orderkey"]), _p(d["l_suppkey"]), _p(d["l_quantity"]), _p(d["l_extendedprice"]), _p(d["l_discount"]),
                          _p(d["l_tax"]), _p(d["l_shipdate"]), _p(d["l_returnflag"]), _p(d["l_returnflag_off"]), _p(d["l_linestatus"]),
                          _p(d["l_linestatus_off"]))
    return d


def lineitem_host_to_arrow(d, n):
    """numpy generator output -> pyarrow Table with the reference schema (tpch.rs:923-940)."""
    import pyarrow as pa

    def decimal(a):
        return pa.Array.from_buffers(pa.decimal128(15, 2), n, [None, pa.py_buffer(a.tobytes())])

    def utf8(data, off):
        return pa.Array.from_buffers(pa.string(), n, [None, pa.py_buffer(off.tobytes()), pa.py_buffer(data[:n].tobytes())])
    return pa.table({
        "l_orderkey": pa.array(d["l_orderkey"]), "l_suppkey": pa.array(d["l_suppkey"]),
        "l_quantity": decimal(d["l_quantity"]), "l_extendedprice": decimal(d["l_extendedprice"]),
        "l_discount": decimal(d["l_discount"]), "l_tax": decimal(d["l_tax"]),
        "l_returnflag": utf8(d["l_returnflag"], d["l_returnflag_off"]), "l_linestatus": utf8(d["l_linestatus"], d["l_linestatus_off"]),
        "l_shipdate": pa.Array.from_buffers(pa.date32(), n, [None, pa.py_buffer(d["l_shipdate"].tobytes())]),
    })


def q1_oracle_raw(n, seed=SEED_LINEITEM, seed_orders=SEED_ORDERS, row0=0, host=None):
    """C oracle q1 over generated rows -> list of (rf, ls, [5 sums], count)."""
    L = oracle_lib()
    d = host if host is not None else gen_lineitem_host(n, seed, seed_orders, row0)
    keys = np.zeros(16, np.uint8); sums = np.zeros(8 * 5 * 2, np.uint64); cnts = np.zeros(8, np.int64)
    ng = L.oracle_q1(C.c_int64(n), _p(d["l_quantity"]), _p(d["l_extendedprice"]), _p(d["l_discount"]), _p(d["l_tax"]), _p(d["l_shipdate"]),
                     _p(d["l_returnflag"]), _p(d["l_returnflag_off"]), _p(d["l_linestatus"]), _p(d["l_linestatus_off"]),
                     C.c_int32(Q1_SHIPDATE_MAX), _p(keys), _p(sums), _p(cnts))
    assert ng <= 8
    out = []
    for g in range(ng):
        vals = []
        for a in range(5):
            lo, hi = int(sums[(g * 5 + a) * 2]), int(sums[(g * 5 + a) * 2 + 1])
            v = (hi << 64) | lo
            vals.append(v - (1 << 128) if v >> 127 else v)
        out.append((chr(keys[2 * g]), chr(keys[2 * g + 1]), vals, int(cnts[g])))
    return out


def _tdiv(a, b):
    q = abs(a) // abs(b)
    return -q if (a < 0) != (b < 0) else q


def q1_rows_from_raw(raw):
    """(rf, ls, sum_qty, sum_base, sum_disc_price, sum_charge, avg_qty, avg_price, avg_disc, count) as unscaled ints,
    ordered by (rf, ls) -- the q1 ORDER BY."""
    rows = []
    for rf, ls, (s_qty, s_base, s_dp, s_ch, s_disc), cnt in raw:
        rows.append((rf, ls, s_qty, s_base, s_dp, s_ch, _tdiv(s_qty * 10**4, cnt), _tdiv(s_base * 10**4, cnt), _tdiv(s_disc * 10**4, cnt), cnt))
    return sorted(rows)


def q1_oracle_rows(n, seed=SEED_LINEITEM, seed_orders=SEED_ORDERS, row0=0):
    return q1_rows_from_raw(q1_oracle_raw(n, seed, seed_orders, row0))


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
