"""Chain fusion in the native executor (csrc/plan_exec.cpp HashJoinExec::chain_build, include/gpuq.h gpuq_join_build_run_semi):
(A |x| B) |x| C with A's keys unique builds the outer join's table straight from B's rows that find their key in A's table.  The
rows must be those of the two-step form -- checked against a plain dictionary join on the host and against the Python mirror
(plan.py executes the two joins one after the other) -- for every outer join type the fused form takes, with filters fused into each
side, NULL keys, A's columns used above the join, duplicate keys in A (falls back), and again when executed deferred."""
import collections

import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd.expr import Operator as Op
from arrow_ballista_amd.expr import binary, col, lit
from test_gpu_native_plan import arrow_rows
from test_gpu_operators import dev_rows, norm

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _node_execute_is_the_mirror(mirror_layer):
    """In this module `plan.execute(0, tc)` is the second opinion the native executor's result (NativePlan / native_rows) is compared with."""



def _tables(seed, na, nb, nc, dup_a=False, nulls=0.1):
    r = np.random.default_rng(seed)
    ak = r.permutation(na * 2)[:na].astype(np.int64)
    if dup_a:
        ak[na // 2:] = ak[: na - na // 2]
    A = pa.table({"ak": pa.array(ak), "av": pa.array(r.integers(0, 100, na), pa.int64())})
    bka = r.integers(0, na * 2, nb).astype(object)
    bkc = (r.permutation(nb * 3)[:nb] + 7).astype(object)
    for arr in (bka, bkc):
        arr[r.random(nb) < nulls] = None
    B = pa.table({"bka": pa.array(list(bka), pa.int64()), "bkc": pa.array(list(bkc), pa.int64()), "bv": pa.array(r.integers(0, 1000, nb), pa.int64())})
    ck = r.integers(0, nb * 3 + 7, nc).astype(object)
    ck[r.random(nc) < nulls] = None
    C = pa.table({"ck": pa.array(list(ck), pa.int64()), "cv": pa.array(np.arange(nc), pa.int64())})
    return A, B, C


def _expected(A, B, C, jt, a_pred=None, b_pred=None, c_pred=None):
    a_rows = [r for r in zip(*[A[c].to_pylist() for c in A.column_names]) if a_pred is None or a_pred(r)]
    b_rows = [r for r in zip(*[B[c].to_pylist() for c in B.column_names]) if b_pred is None or b_pred(r)]
    c_rows = [r for r in zip(*[C[c].to_pylist() for c in C.column_names]) if c_pred is None or c_pred(r)]
    by_a = collections.defaultdict(list)
    for r in a_rows:
        by_a[r[0]].append(r)
    ab = [a + b for b in b_rows if b[0] is not None for a in by_a.get(b[0], ())]
    by_bkc = collections.defaultdict(list)
    for r in ab:
        if r[3] is not None:
            by_bkc[r[3]].append(r)
    out = []
    for c in c_rows:
        ms = by_bkc.get(c[0], ()) if c[0] is not None else ()
        if jt == "Inner":
            out += [m + c for m in ms]
        elif jt == "Right":
            out += [m + c for m in ms] if ms else [(None,) * 5 + c]
        elif jt == "RightSemi":
            out += [c] if ms else []
        elif jt == "RightAnti":
            out += [] if ms else [c]
    return out


def _plan(A, B, C, jt, filters=False):
    a, b, c = g.MemoryExec([A]), g.MemoryExec([B]), g.MemoryExec([C])
    as_, bs, cs = a.schema(), b.schema(), c.schema()
    la = g.FilterExec(binary(col("av", as_), Op.Lt, lit(80)), a) if filters else a
    lb = g.FilterExec(binary(col("bv", bs), Op.GtEq, lit(100)), b) if filters else b
    lc = g.FilterExec(binary(col("cv", cs), Op.NotEq, lit(3)), c) if filters else c
    j1 = g.HashJoinExec(la, g.CoalesceBatchesExec(lb), [(col("ak", as_), col("bka", bs))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    return g.HashJoinExec(g.CoalesceBatchesExec(j1), lc, [(col("bkc", j1s), col("ck", cs))], None, jt, "CollectLeft", False)


@pytest.mark.parametrize("jt", ["Inner", "Right", "RightSemi", "RightAnti"])
@pytest.mark.parametrize("filters", [False, True])
@pytest.mark.parametrize("dup_a", [False, True])
def test_chain_equals_two_step(tc, jt, filters, dup_a):
    A, B, C = _tables(17 + dup_a, 700, 5000, 9000, dup_a=dup_a)
    plan = _plan(A, B, C, jt, filters)
    exp = _expected(A, B, C, jt, *( (lambda r: r[1] < 80, lambda r: r[2] >= 100, lambda r: r[1] != 3) if filters else (None, None, None)))
    p = g.NativePlan(plan, tc)
    for run in range(3):          # synchronous, then deferred twice
        got = arrow_rows(p.execute(0).to_arrow())
        assert norm(got) == norm(exp), (run, p.exec_stats())
        assert len(exp) > 0
    assert p.exec_stats()["deferred"]
    assert norm(dev_rows(tc, plan.execute(0, tc))) == norm(exp)          # the mirror's two-step execution agrees
    m = [x for x in p.metrics() if x["node"] == "HashJoinExec"]
    inner_rows = len(_expected(A, B, pa.table({"ck": pa.array([], pa.int64()), "cv": pa.array([], pa.int64())}), "Inner")) if False else None
    assert len(m) == 2 and all(x["output_rows"] >= 0 for x in m)


def test_chain_inner_metrics_and_large(tc):
    """2 M x 6 M rows, A's columns read above the join (through the hit vector), output rows of BOTH joins reported."""
    r = np.random.default_rng(2)
    na, nb, nc = 200_000, 2_000_000, 6_000_000
    A = pa.table({"ak": pa.array(r.permutation(na * 3)[:na].astype(np.int64)), "av": pa.array(r.integers(0, 100, na), pa.int64())})
    B = pa.table({"bka": pa.array(r.integers(0, na * 3, nb), pa.int64()), "bkc": pa.array(np.arange(nb, dtype=np.int64) * 4 + 1), "bv": pa.array(r.integers(0, 1000, nb), pa.int64())})
    C = pa.table({"ck": pa.array(np.repeat(np.arange(nb, dtype=np.int64) * 4 + 1, 3)[:nc]), "cv": pa.array(np.arange(nc), pa.int64())})
    plan = _plan(A, B, C, "Inner", filters=True)
    js = plan.schema()
    agg = g.AggregateExec("Single", [(col("av", js), "av")], [{"fn": "SUM", "expr": col("cv", js), "name": "s"}, {"fn": "COUNT", "expr": lit(1), "name": "c"}], plan)
    p = g.NativePlan(agg, tc)
    # host-side truth with numpy
    a_ok = A["av"].to_numpy() < 80
    amap = dict(zip(A["ak"].to_numpy()[a_ok].tolist(), A["av"].to_numpy()[a_ok].tolist()))
    bka, bv = B["bka"].to_numpy(), B["bv"].to_numpy()
    b_av = np.array([amap.get(k, -1) for k in bka.tolist()])
    b_live = (b_av >= 0) & (bv >= 100)
    c_b = (C["ck"].to_numpy() - 1) // 4
    cv = C["cv"].to_numpy()
    live = b_live[c_b] & (cv != 3)
    acc = collections.defaultdict(lambda: [0, 0])
    for av, v in zip(b_av[c_b][live].tolist(), cv[live].tolist()):
        acc[av][0] += v; acc[av][1] += 1
    exp = sorted((k, s, c) for k, (s, c) in acc.items())
    for run in range(3):
        got = sorted(tuple(r) for r in arrow_rows(p.execute(0).to_arrow()))
        assert got == exp, run
    assert p.exec_stats() == {"deferred": True, "settles": 1, "host_syncs": 0, "retries": 0}
    m = [x for x in p.metrics() if x["node"] == "HashJoinExec"]
    assert sorted(x["output_rows"] for x in m) == sorted([3 * int(b_live.sum()), 3 * int(live.sum())])


def test_chain_without_the_inner_build_sides_columns(tc):
    """Nothing above the inner join reads a column of A: the fused build does not form the hit vector and A's columns are not part of
    what flows up (the required-columns walk at plan creation decides); a projection of B's and C's columns must still be right."""
    A, B, C = _tables(5, 900, 20_000, 30_000)
    plan = _plan(A, B, C, "Inner", filters=True)
    js = plan.schema()
    proj = g.ProjectionExec([(col("cv", js), "cv"), (binary(col("bv", js), Op.Plus, col("ck", js)), "x")], plan)
    exp = [(r[6], r[4] + r[5]) for r in _expected(A, B, C, "Inner", lambda r: r[1] < 80, lambda r: r[2] >= 100, lambda r: r[1] != 3)]
    p = g.NativePlan(proj, tc)
    for _ in range(3):
        assert norm(arrow_rows(p.execute(0).to_arrow())) == norm(exp) and len(exp) > 0
    assert p.exec_stats()["deferred"]
