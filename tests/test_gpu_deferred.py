"""Deferred execution of native plans (include/gpuq.h "Deferred execution", csrc/plan_exec.cpp): from its second execution on a plan
queues every operator without reading anything back -- build-side key ranges, table sizes, group counts, sort-key layouts are the
ones the operators remember from their synchronous run, row counts travel between operators as device words -- and ONE host round
trip settles it.  What must hold: the same rows as the synchronous execution, and, when the input changes under the plan so that what
was remembered no longer fits (more matches than the pair vectors hold, keys outside the remembered range, a duplicate build key, more
groups, values outside the sort key's field), the execution is redone synchronously and still returns the right rows."""
import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
import tpch_util as T
from arrow_ballista_amd.expr import col, lit
from test_gpu_native_plan import arrow_rows

pytestmark = pytest.mark.gpu


def _q3_tables(tc, n_li, n_cust, seed_shift=0):
    cols = ("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate")
    li = T.gen_lineitem_device(tc, n_li, n_supp=100, columns=cols)
    od = T.gen_orders_device(tc, (n_li + 3) // 4, n_cust)
    cu = T.gen_customer_device(tc, n_cust)
    hl = T.lineitem_host_to_arrow(T.gen_lineitem_host(n_li, n_supp=100), n_li)
    ho, hc, _ = T.gen_other_tables_host((n_li + 3) // 4, n_cust, 100)
    return (li, od, cu), (hl, ho, hc)


def _check_q3(got, exp):
    assert len(got) == len(exp) and len(exp) > 0
    assert [(r[1], r[2]) for r in got] == [(r[1], r[2]) for r in exp]
    assert sorted(got) == sorted(exp)


@pytest.mark.parametrize("n_li", [4000, 300_000, 5_000_000])
def test_q3_second_execution_is_deferred_and_equal(tc, n_li):
    (li, od, cu), (hl, ho, hc) = _q3_tables(tc, n_li, 1500 if n_li < 1_000_000 else 60_000)
    p = g.NativePlan(T.q3_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li])), tc)
    first = arrow_rows(p.execute(0).to_arrow())
    s0 = p.exec_stats()
    assert not s0["deferred"] and s0["host_syncs"] > 4
    if n_li <= 300_000:
        _check_q3(first, T.q3_oracle(hc, ho, hl))
    for _ in range(3):
        again = arrow_rows(p.execute(0).to_arrow())
        st = p.exec_stats()
        assert again == first
        assert st["deferred"] and st["settles"] == 1 and st["host_syncs"] == 0 and st["retries"] == 0, st
    m = {x["node"]: x for x in p.metrics()}
    assert m["SortExec"]["output_rows"] == 4 * len(first)          # metrics booked with a bound are corrected at the settle


def test_q3_input_changes_under_the_plan(tc):
    """The plan learns on a small input, then meets a larger one (more pairs than its vectors hold, order keys beyond the remembered
    range, more groups) and a smaller one: each time the answer is the oracle's for THAT input."""
    small, hs = _q3_tables(tc, 4000, 1500)
    big, hb = _q3_tables(tc, 120_000, 1500)
    p = g.NativePlan(T.q3_plan(g.MemoryExec([small[2]]), g.MemoryExec([small[1]]), g.MemoryExec([small[0]])), tc)
    _check_q3(arrow_rows(p.execute(0).to_arrow()), T.q3_oracle(hs[2], hs[1], hs[0]))
    _check_q3(arrow_rows(p.execute(0).to_arrow()), T.q3_oracle(hs[2], hs[1], hs[0]))
    assert p.exec_stats()["deferred"]
    # slots follow MemoryExec creation order inside q3_plan: customer, orders, lineitem
    for slot, t in zip(range(3), (big[2], big[1], big[0])):
        p.set_input(slot, t)
    _check_q3(arrow_rows(p.execute(0).to_arrow()), T.q3_oracle(hb[2], hb[1], hb[0]))
    st = p.exec_stats()
    assert not st["deferred"] and st["retries"] == 1, st
    _check_q3(arrow_rows(p.execute(0).to_arrow()), T.q3_oracle(hb[2], hb[1], hb[0]))
    assert p.exec_stats()["deferred"]
    for slot, t in zip(range(3), (small[2], small[1], small[0])):
        p.set_input(slot, t)
    _check_q3(arrow_rows(p.execute(0).to_arrow()), T.q3_oracle(hs[2], hs[1], hs[0]))      # smaller fits what is remembered: held or redone, right either way


def _kv(keys, vals):
    return pa.table({"k": pa.array(keys, pa.int64()), "v": pa.array(vals, pa.int64())})


def test_join_build_side_changes_under_the_plan(tc):
    """Inner join whose build side (a) gets a key outside the remembered dense range, (b) gets a duplicate key; probe side with more
    matches than before.  Checked against the oracle's join each time."""
    rng = np.random.default_rng(5)

    def run(p, lt, rt):
        rows = arrow_rows(p.execute(0).to_arrow())
        import collections
        by_key = collections.defaultdict(list)
        for k, v in zip(lt["k"].to_pylist(), lt["v"].to_pylist()):
            by_key[k].append(v)
        exp = [(k, v, k, rv) for k, rv in zip(rt["rk"].to_pylist(), rt["rv"].to_pylist()) for v in by_key.get(k, ())]
        assert sorted(rows) == sorted(exp) and len(exp) > 0
    lt = _kv(rng.permutation(5000)[:3000] + 100, np.arange(3000))
    rt = pa.table({"rk": pa.array(rng.integers(0, 6000, 40_000), pa.int64()), "rv": pa.array(np.arange(40_000), pa.int64())})
    L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
    p = g.NativePlan(g.HashJoinExec(L, R, [(col("k", L.schema()), col("rk", R.schema()))], None, "Inner", "CollectLeft", False), tc)
    run(p, lt, rt); run(p, lt, rt)
    assert p.exec_stats()["deferred"]
    # (a) a key far outside the remembered range
    lt2 = _kv(np.concatenate([lt["k"].to_numpy(), [10_000_000]]), np.arange(3001))
    rt2 = pa.table({"rk": pa.array(np.concatenate([rt["rk"].to_numpy()[:-5], [10_000_000] * 5]), pa.int64()), "rv": rt["rv"]})
    p.set_input(0, g.DeviceTable.from_arrow(lt2, tc.device)); p.set_input(1, g.DeviceTable.from_arrow(rt2, tc.device))
    run(p, lt2, rt2)
    assert p.exec_stats()["retries"] == 1
    run(p, lt2, rt2)
    # (b) a duplicate build key (unique keys were remembered)
    k3 = lt["k"].to_numpy().copy(); k3[7] = k3[8]
    lt3 = _kv(k3, np.arange(3000))
    p.set_input(0, g.DeviceTable.from_arrow(lt3, tc.device)); p.set_input(1, g.DeviceTable.from_arrow(rt, tc.device))
    run(p, lt3, rt)
    run(p, lt3, rt)
    # (c) every probe row matches now: more pairs than the vectors sized from the last run
    rt4 = pa.table({"rk": pa.array(rng.choice(k3, 40_000), pa.int64()), "rv": pa.array(np.arange(40_000), pa.int64())})
    p.set_input(1, g.DeviceTable.from_arrow(rt4, tc.device))
    run(p, lt3, rt4)
    run(p, lt3, rt4)


def test_aggregate_and_sort_change_under_the_plan(tc):
    """GROUP BY + ORDER BY: first few groups (LDS dictionary), then thousands (hash table), then values outside the sort key's field."""
    rng = np.random.default_rng(11)

    def table(n, ngroups, vmax):
        return pa.table({"k": pa.array(rng.integers(0, ngroups, n), pa.int64()), "v": pa.array(rng.integers(-vmax, vmax, n), pa.int64())})

    def plan_of(src):
        s = src.schema()
        agg = g.AggregateExec("Single", [(col("k", s), "k")], [{"fn": "SUM", "expr": col("v", s), "name": "s"}, {"fn": "COUNT", "expr": lit(1), "name": "c"}], src)
        a = agg.schema()
        return g.SortExec([{"expr": col("s", a), "asc": False, "nulls_first": True}, {"expr": col("k", a), "asc": True, "nulls_first": False}], agg)

    def expect(t):
        import collections
        acc = collections.defaultdict(lambda: [0, 0])
        for k, v in zip(t["k"].to_pylist(), t["v"].to_pylist()):
            acc[k][0] += v; acc[k][1] += 1
        return sorted(((k, s, c) for k, (s, c) in acc.items()), key=lambda r: (-r[1], r[0]))

    t1 = table(100_000, 7, 1000)
    src = g.MemoryExec([t1])
    p = g.NativePlan(plan_of(src), tc)
    for t, want_retry in ((t1, False), (t1, False), (table(100_000, 5000, 1000), True), (None, False), (table(100_000, 5000, 10**12), True), (None, False), (table(3_000_000, 200_000, 50), True), (None, False)):
        if t is None:
            t = last
        else:
            p.set_input(0, g.DeviceTable.from_arrow(t, tc.device))
        last = t
        before = p.exec_stats()["retries"]
        got = [tuple(r) for r in arrow_rows(p.execute(0).to_arrow())]
        assert got == expect(t)
        if want_retry:
            assert p.exec_stats()["retries"] == before + 1


def test_filter_view_then_consumers(tc):
    """A standalone FilterExec (LIKE keeps it from being fused) hands a device-side count to projection, join and aggregate."""
    n = 50_000
    rng = np.random.default_rng(3)
    words = np.array(["alpha", "beta", "gamma special", "delta", "special requests"])
    t = pa.table({"s": pa.array(words[rng.integers(0, 5, n)]), "k": pa.array(rng.integers(0, 300, n), pa.int64()), "v": pa.array(rng.integers(0, 1000, n), pa.int64())})
    d = _kv(np.arange(0, 300, 2), np.arange(150))
    from arrow_ballista_amd.expr import like
    src, dim = g.MemoryExec([t]), g.MemoryExec([d])
    s, ds = src.schema(), dim.schema()
    f = g.FilterExec(like(col("s", s), "%special%"), src)
    j = g.HashJoinExec(dim, f, [(col("k", ds), col("k", s))], None, "Inner", "CollectLeft", False)
    p = g.NativePlan(j, tc)
    rows = [arrow_rows(p.execute(0).to_arrow()) for _ in range(3)]
    assert rows[0] == rows[1] == rows[2] and len(rows[0]) > 0
    exp = [(int(k), int(k) // 2, sv, int(k), int(v)) for sv, k, v in zip(t["s"].to_pylist(), t["k"].to_pylist(), t["v"].to_pylist()) if "special" in sv and k % 2 == 0]
    assert sorted(rows[0]) == sorted(exp)
    assert p.exec_stats()["deferred"]
