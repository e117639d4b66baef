"""Shared helpers for tests, __graft_entry__.smoke() and bench.py: synthetic TPC-H-shaped inputs
(device generator + the oracle's CPU restatement of it), the q1/q3/q5 operator plans written with
the reference's operator names, and result normalisation.  The oracle is used here only as the checker."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.tpch import *  # noqa: E402,F401,F403  (generators and plans: shared with bench.py)
from benchmarks.tpch import _dev_cols  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
        L.oracle_q3.restype = i64
        L.oracle_q3.argtypes = [i64, vp, vp, vp, C.c_char_p, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, vp]
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


def gen_other_tables_host(n_orders, n_cust, n_supp):
    """Oracle-side restatement of orders / customer / supplier as pyarrow tables."""
    import pyarrow as pa
    L = oracle_lib()
    ok, oc, od, osp = np.empty(n_orders, np.int64), np.empty(n_orders, np.int64), np.empty(n_orders, np.int32), np.empty(n_orders, np.int32)
    L.oracle_gen_orders(SEED_ORDERS, 0, n_orders, n_cust, _p(ok), _p(oc), _p(od), _p(osp))
    ck, cn, cm, co = np.empty(n_cust, np.int64), np.empty(n_cust, np.int64), np.empty(n_cust * 9 + 16, np.uint8), np.empty(n_cust + 1, np.int32)
    L.oracle_gen_customer(SEED_CUSTOMER, 0, n_cust, _p(ck), _p(cn), _p(cm), _p(co))
    sk, sn = np.empty(n_supp, np.int64), np.empty(n_supp, np.int64)
    L.oracle_gen_supplier(SEED_SUPPLIER, 0, n_supp, _p(sk), _p(sn))
    orders = pa.table({"o_orderkey": ok, "o_custkey": oc, "o_orderdate": pa.array(od).cast(pa.date32()), "o_shippriority": osp})
    seg = pa.Array.from_buffers(pa.string(), n_cust, [None, pa.py_buffer(co.tobytes()), pa.py_buffer(cm.tobytes())])
    customer = pa.table({"c_custkey": ck, "c_nationkey": cn, "c_mktsegment": seg})
    supplier = pa.table({"s_suppkey": sk, "s_nationkey": sn})
    return orders, customer, supplier


def q3_oracle(customer, orders, lineitem):
    """numpy restatement of q3 over pyarrow tables -> rows ordered by (revenue desc, o_orderdate)."""
    ck = customer["c_custkey"].to_numpy(); seg = np.array(customer["c_mktsegment"].to_pylist())
    good_c = set(ck[seg == "BUILDING"].tolist())
    ok, oc, od, osp = (orders[c].to_numpy() if c != "o_orderdate" else orders[c].cast("int32").to_numpy() for c in ("o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"))
    omap = {int(k): (int(d), int(s)) for k, c, d, s in zip(ok, oc, od, osp) if d < Q3_DATE and int(c) in good_c}
    lk = lineitem["l_orderkey"].to_numpy(); ls = lineitem["l_shipdate"].cast("int32").to_numpy()
    ext = [int(x.scaleb(2)) for x in lineitem["l_extendedprice"].to_pylist()]; disc = [int(x.scaleb(2)) for x in lineitem["l_discount"].to_pylist()]
    groups = {}
    for i in range(len(lk)):
        if ls[i] > Q3_DATE and int(lk[i]) in omap:
            d, s = omap[int(lk[i])]
            k = (int(lk[i]), d, s)
            groups[k] = groups.get(k, 0) + ext[i] * (100 - disc[i])
    return sorted(((k[0], v, k[1], k[2]) for k, v in groups.items()), key=lambda r: (-r[1], r[2]))


def q5_oracle(customer, orders, lineitem, supplier):
    asia = {i for i, (_, r) in enumerate(NATIONS) if r == 2}
    cn = dict(zip(customer["c_custkey"].to_pylist(), customer["c_nationkey"].to_pylist()))
    sn = dict(zip(supplier["s_suppkey"].to_pylist(), supplier["s_nationkey"].to_pylist()))
    od = orders["o_orderdate"].cast("int32").to_numpy()
    omap = {int(k): cn[int(c)] for k, c, d in zip(orders["o_orderkey"].to_numpy(), orders["o_custkey"].to_numpy(), od)
            if Q5_DATE_LO <= d < Q5_DATE_HI and cn[int(c)] in asia}
    lk = lineitem["l_orderkey"].to_numpy(); lsup = lineitem["l_suppkey"].to_numpy()
    ext = [int(x.scaleb(2)) for x in lineitem["l_extendedprice"].to_pylist()]; disc = [int(x.scaleb(2)) for x in lineitem["l_discount"].to_pylist()]
    groups = {}
    for i in range(len(lk)):
        nk = omap.get(int(lk[i]))
        if nk is not None and sn[int(lsup[i])] == nk:
            groups[NATIONS[nk][0]] = groups.get(NATIONS[nk][0], 0) + ext[i] * (100 - disc[i])
    return sorted(groups.items(), key=lambda r: -r[1])


def gen_q3_tables_host(n_li, n_cust):
    """numpy columns (Arrow physical layout) of the three q3 tables from the oracle's generator: the inputs of q3_oracle_c."""
    L = oracle_lib()
    n_orders = (n_li + 3) // 4
    li = gen_lineitem_host(n_li)
    ok, oc, od, osp = np.empty(n_orders, np.int64), np.empty(n_orders, np.int64), np.empty(n_orders, np.int32), np.empty(n_orders, np.int32)
    L.oracle_gen_orders(SEED_ORDERS, 0, n_orders, n_cust, _p(ok), _p(oc), _p(od), _p(osp))
    ck, cn, cm, co = np.empty(n_cust, np.int64), np.empty(n_cust, np.int64), np.empty(n_cust * 9 + 16, np.uint8), np.empty(n_cust + 1, np.int32)
    L.oracle_gen_customer(SEED_CUSTOMER, 0, n_cust, _p(ck), _p(cn), _p(cm), _p(co))
    return dict(n_li=n_li, n_orders=n_orders, n_cust=n_cust, l_orderkey=li["l_orderkey"], l_extendedprice=li["l_extendedprice"], l_discount=li["l_discount"],
                l_shipdate=li["l_shipdate"], o_orderkey=ok, o_custkey=oc, o_orderdate=od, o_shippriority=osp, c_custkey=ck, c_mktsegment=cm, c_mktsegment_off=co)


def gen_q5_tables_host(n_li, n_cust, n_supp):
    """numpy columns of the q5 tables from the oracle's generator (lineitem, orders, customer, supplier): the inputs of q5_oracle_c."""
    L = oracle_lib()
    h = gen_q3_tables_host(n_li, n_cust)
    li = gen_lineitem_host(n_li, n_supp=n_supp)
    h["l_suppkey"] = li["l_suppkey"]; h["l_orderkey"] = li["l_orderkey"]; h["l_extendedprice"] = li["l_extendedprice"]; h["l_discount"] = li["l_discount"]
    cn = np.empty(n_cust, np.int64); ck = np.empty(n_cust, np.int64); cm = np.empty(n_cust * 9 + 16, np.uint8); co = np.empty(n_cust + 1, np.int32)
    L.oracle_gen_customer(SEED_CUSTOMER, 0, n_cust, _p(ck), _p(cn), _p(cm), _p(co))
    sk, sn = np.empty(n_supp, np.int64), np.empty(n_supp, np.int64)
    L.oracle_gen_supplier(SEED_SUPPLIER, 0, n_supp, _p(sk), _p(sn))
    h.update(c_nationkey=cn, s_suppkey=sk, s_nationkey=sn, n_supp=n_supp)
    return h


def q5_oracle_c(h):
    """C oracle q5 over gen_q5_tables_host columns -> (rows [(n_name, revenue)] ordered by revenue desc, stats dict)."""
    L = oracle_lib()
    L.oracle_q5.restype = C.c_int64
    nat_reg = np.array([r for _, r in NATIONS], np.int64)
    out_n, out_r, st = np.zeros(64, np.int32), np.zeros(128, np.uint64), np.zeros(3, np.int64)
    ng = L.oracle_q5(C.c_int64(len(NATIONS)), _p(nat_reg), C.c_int64(2),
                     C.c_int64(h["n_cust"]), _p(h["c_custkey"]), _p(h["c_nationkey"]),
                     C.c_int64(h["n_orders"]), _p(h["o_orderkey"]), _p(h["o_custkey"]), _p(h["o_orderdate"]), C.c_int32(Q5_DATE_LO), C.c_int32(Q5_DATE_HI),
                     C.c_int64(h["n_li"]), _p(h["l_orderkey"]), _p(h["l_suppkey"]), _p(h["l_extendedprice"]), _p(h["l_discount"]),
                     C.c_int64(h["n_supp"]), _p(h["s_suppkey"]), _p(h["s_nationkey"]), _p(out_n), _p(out_r), _p(st))
    rows = []
    for g_ in range(int(ng)):
        v = (int(out_r[2 * g_ + 1]) << 64) | int(out_r[2 * g_])
        rows.append((NATIONS[int(out_n[g_])][0], v - (1 << 128) if v >> 127 else v))
    return rows, dict(customers=int(st[0]), orders=int(st[1]), pairs=int(st[2]))


def q3_oracle_c(h, cap=None):
    """C oracle q3 over gen_q3_tables_host columns -> (rows [(l_orderkey, revenue, o_orderdate, o_shippriority)] ordered by
    (revenue desc, o_orderdate, l_orderkey), stats dict)."""
    L = oracle_lib()
    cap = cap if cap is not None else h["n_orders"]
    ok, rev, od, sp = np.empty(max(cap, 1), np.int64), np.empty(2 * max(cap, 1), np.uint64), np.empty(max(cap, 1), np.int32), np.empty(max(cap, 1), np.int32)
    st = np.zeros(4, np.int64)
    ng = L.oracle_q3(h["n_cust"], _p(h["c_custkey"]), _p(h["c_mktsegment"]), _p(h["c_mktsegment_off"]), b"BUILDING",
                     h["n_orders"], _p(h["o_orderkey"]), _p(h["o_custkey"]), _p(h["o_orderdate"]), _p(h["o_shippriority"]),
                     h["n_li"], _p(h["l_orderkey"]), _p(h["l_extendedprice"]), _p(h["l_discount"]), _p(h["l_shipdate"]), Q3_DATE,
                     cap, _p(ok), _p(rev), _p(od), _p(sp), _p(st))
    k = min(ng, cap)
    rows = []
    for g_ in range(k):
        v = (int(rev[2 * g_ + 1]) << 64) | int(rev[2 * g_])
        rows.append((int(ok[g_]), v - (1 << 128) if v >> 127 else v, int(od[g_]), int(sp[g_])))
    return rows, dict(groups=int(ng), j1_build_rows=int(st[0]), j1_output_rows=int(st[1]), j2_probe_rows=int(st[2]), j2_matches=int(st[3]))
