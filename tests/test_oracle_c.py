"""CPU: the C restatement (oracle/oracle.c, also bench.py's cpu_baseline) against the Python oracle."""
import ctypes as C

import numpy as np

import tpch_util as T
from oracle import oracle_np as O


def test_q1_c_equals_python_oracle():
    n = 20_000
    host = T.gen_lineitem_host(n, seed=3)
    t = O.Table.from_arrow(T.lineitem_host_to_arrow(host, n))
    c = lambda x: {"column": {"name": x}}
    one = {"literal": {"type": {"Decimal128": [20, 0]}, "value": "1"}}
    dp = {"binary_expr": {"l": c("l_extendedprice"), "r": {"binary_expr": {"l": one, "r": c("l_discount"), "op": "-"}}, "op": "*"}}
    ch = {"binary_expr": {"l": dp, "r": {"binary_expr": {"l": one, "r": c("l_tax"), "op": "+"}}, "op": "*"}}
    aggs = [{"fn": "SUM", "expr": c("l_quantity"), "name": "a"}, {"fn": "SUM", "expr": c("l_extendedprice"), "name": "b"},
            {"fn": "SUM", "expr": dp, "name": "c"}, {"fn": "SUM", "expr": ch, "name": "d"}, {"fn": "AVG", "expr": c("l_quantity"), "name": "e"},
            {"fn": "AVG", "expr": c("l_extendedprice"), "name": "f"}, {"fn": "AVG", "expr": c("l_discount"), "name": "g"},
            {"fn": "COUNT", "expr": {"literal": {"type": "Int64", "value": 1}}, "name": "h"}]
    pred = {"binary_expr": {"l": c("l_shipdate"), "r": {"literal": {"type": "Date32", "value": T.Q1_SHIPDATE_MAX}}, "op": "<="}}
    out = O.aggregate(t, [(c("l_returnflag"), "rf"), (c("l_linestatus"), "ls")], aggs, "Single", predicate=pred)
    assert sorted(out.rows()) == T.q1_rows_from_raw(T.q1_oracle_raw(n, host=host))
    assert out.types[2:] == [O.dec(25, 2), O.dec(25, 2), O.dec(38, 4), O.dec(38, 6), O.dec(19, 6), O.dec(19, 6), O.dec(19, 6), "Int64"]


def test_join_sort_partition_c_equals_python_oracle():
    L = T.oracle_lib()
    r = np.random.default_rng(1)
    build = r.integers(0, 500, 2000).astype(np.int64)
    probe = r.integers(0, 700, 5000).astype(np.int64)
    tbl = L.oracle_join_build(build.ctypes.data, len(build))
    ob, op = np.zeros(100000, np.uint32), np.zeros(100000, np.uint32)
    cs = C.c_uint64(0)
    k = L.oracle_join_probe(tbl, probe.ctypes.data, len(probe), ob.ctypes.data, op.ctypes.data, len(ob), C.byref(cs))
    k2 = L.oracle_join_probe(tbl, probe.ctypes.data, len(probe), None, None, 0, C.byref(cs))
    L.oracle_join_free(tbl)
    c = lambda x: {"column": {"name": x}}
    lt, rt = O.Table(["k"], ["Int64"], [build.tolist()]), O.Table(["k"], ["Int64"], [probe.tolist()])
    assert k == k2 and sorted(zip(ob[:k].tolist(), op[:k].tolist())) == sorted(O.hash_join(lt, rt, [(c("k"), c("k"))], "Inner"))
    keys = r.integers(0, 1 << 40, 10000).astype(np.uint64)
    perm = np.zeros(len(keys), np.uint32)
    L.oracle_sort_u64(keys.ctypes.data, C.c_int64(len(keys)), perm.ctypes.data)
    assert perm.tolist() == np.argsort(keys, kind="stable").tolist()
    pid = np.zeros(len(build), np.uint32)
    L.oracle_partition_ids_i64(build.ctypes.data, C.c_int64(len(build)), C.c_uint32(16), pid.ctypes.data)
    assert pid.tolist() == O.hash_partition(lt, [c("k")], 16)


def test_q3_c_equals_python_oracle():
    """oracle_q3 (bench.py's cpu_baseline at N=1) against the numpy restatement of q3 on the same generated tables."""
    import pyarrow as pa
    n_li, n_cust = 60_000, 1500
    h = T.gen_q3_tables_host(n_li, n_cust)
    rows, st = T.q3_oracle_c(h)
    hl = T.lineitem_host_to_arrow(T.gen_lineitem_host(n_li), n_li)
    ho, hc, _hs = T.gen_other_tables_host(h["n_orders"], n_cust, 100)
    exp = T.q3_oracle(hc, ho, hl)
    assert st["groups"] == len(exp) > 0 and st["j2_matches"] >= len(exp)
    assert [(r[1], r[2]) for r in rows] == [(r[1], r[2]) for r in exp]          # ORDER BY revenue desc, o_orderdate
    assert sorted(rows) == sorted(exp)


def test_c_q5_equals_the_python_restatement():
    """oracle_q5 (C, OpenMP: the CPU baseline of bench.py) against q5_oracle (plain Python dictionaries) on the same generated tables."""
    import tpch_util as T
    n_li, n_cust, n_supp = 60_000, 1500, 100
    h = T.gen_q5_tables_host(n_li, n_cust, n_supp)
    rows, st = T.q5_oracle_c(h)
    hl = T.lineitem_host_to_arrow(T.gen_lineitem_host(n_li, n_supp=n_supp), n_li)
    ho, hc, hs = T.gen_other_tables_host((n_li + 3) // 4, n_cust, n_supp)
    exp = T.q5_oracle(hc, ho, hl, hs)
    assert [tuple(r) for r in rows] == [tuple(r) for r in exp] and len(rows) == 5 and st["pairs"] > 0
