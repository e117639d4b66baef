"""The hiprtc-specialised front-end (JIT "force") must give the same answers as the oracle -- the same
test bodies as the interpreter path, re-run with every sink kernel compiled from the generated source."""
import pytest

import tpch_util as T

pytestmark = pytest.mark.gpu


@pytest.fixture()
def jit(tc):
    st = tc.ctx.jit_stats()
    if not st["available"]:
        pytest.skip("hiprtc not available")
    tc.ctx.set_jit("force")
    before = tc.ctx.jit_stats()["launches"]
    yield tc
    tc.ctx.set_jit("auto")
    assert tc.ctx.jit_stats()["launches"] > before, "no JIT launch happened"


def test_q1_jit(jit):
    for n in (65, 200_000):
        li = T.gen_lineitem_device(jit, n, seed=7)
        for two_phase in (True, False):
            assert T.q1_result_to_rows(jit, T.run_q1(jit, li, two_phase=two_phase)) == T.q1_oracle_rows(n, seed=7)
    li = T.gen_lineitem_device(jit, 300_000, seed=11)
    assert T.q1_result_to_rows(jit, T.run_q1(jit, li, strategy="hash")) == T.q1_oracle_rows(300_000, seed=11)


def test_operators_jit(jit):
    import test_gpu_operators as M
    M.test_filter_exec(jit, 1000, 0.2)
    M.test_filter_exec(jit, 70_001, 0.0)
    M.test_filter_of_filter_and_projection(jit)
    M.test_projection_decimal_types_and_cast(jit)
    M.test_aggregate_single(jit, 500, 0.15)
    M.test_aggregate_single(jit, 40_000, 0.0)
    M.test_aggregate_partial_final_and_strategies(jit, "hash")
    M.test_aggregate_kat_alltypes_plain(jit)


def test_joins_sort_partition_jit(jit, tmp_path):
    import test_gpu_operators as M
    for jt in ("Inner", "Left", "Full", "RightAnti", "LeftSemi"):
        M.test_hash_join_types(jit, jt, 0.2)
    M.test_hash_join_null_equals_null_and_fused_filters(jit)
    M.test_join_then_aggregate_then_sort_pipeline(jit)
    M.test_sort_exec(jit, 10_000, 0.2)
    M.test_sort_fetch(jit)
    M.test_hash_partition(jit, 16)
    M.test_shuffle_writer_round_trip(jit, tmp_path)


def test_tpch_jit(jit):
    import test_gpu_tpch as M
    M.test_q3(jit, 120_000)
    M.test_q5(jit, 120_000)


@pytest.mark.parametrize("jt", ["Inner", "Left", "Right", "Full", "RightSemi", "RightAnti"])
def test_clustered_probe_keys_jit(jit, jt):
    """Foreign keys arrive clustered (the lines of one order are neighbours): the specialised unique-key probe looks a run
    of equal keys up once and hands the answer to the run's other lanes.  Runs of 1..7 equal keys, NULL keys, runs that
    cross wave boundaries, a fused probe-side filter, unique build keys (PK side), against the oracle."""
    import numpy as np
    import pyarrow as pa
    import arrow_ballista_amd as g
    from arrow_ballista_amd.expr import col, binary, lit, Operator as Op
    from oracle import oracle_np as O
    import test_gpu_operators as M
    r = np.random.default_rng(5)
    nb = 5000
    bkeys = r.permutation(np.arange(0, 3 * nb, 3))[:nb].astype(np.int64)          # unique, multiples of 3
    build = pa.table({"bk": pa.array(bkeys), "bv": pa.array(np.arange(nb, dtype=np.int64))})
    runs = r.integers(1, 8, 9000)
    base = r.integers(0, 3 * nb, len(runs)).astype(np.int64)                        # ~1/3 of the runs hit
    pk = np.repeat(base, runs)
    mask = np.repeat(r.random(len(runs)) < 0.1, runs)                               # whole runs of NULL keys
    flt = r.integers(0, 10, len(pk)).astype(np.int32)
    probe = pa.table({"pk": pa.array(pk, mask=mask), "pf": pa.array(flt), "pid": pa.array(np.arange(len(pk), dtype=np.int64))})
    L, R0 = g.MemoryExec([build]), g.MemoryExec([probe])
    rs0 = R0.schema()
    R = g.FilterExec(binary(col("pf", rs0), Op.Lt, lit(8, "Int32")), R0)            # breaks some runs
    ls, rs = L.schema(), R.schema()
    plan = g.HashJoinExec(L, R, [(col("bk", ls), col("pk", rs))], None, jt, "CollectLeft", False)
    got = M.norm(M.dev_rows(jit, plan.execute(0, jit)))
    ol = O.Table.from_arrow(build)
    orr = O.Table.from_arrow(probe.filter(pa.compute.less(probe["pf"], 8)))
    pairs = O.hash_join(ol, orr, [({"column": {"name": "bk"}}, {"column": {"name": "pk"}})], jt)
    lrows, rrows = [tuple(x) for x in ol.rows()], [tuple(x) for x in orr.rows()]
    if jt in ("RightSemi", "RightAnti"):
        exp = [rrows[j] for _, j in pairs]
    else:
        exp = [(lrows[i] if i is not None else (None,) * 2) + (rrows[j] if j is not None else (None,) * 3) for i, j in pairs]
    assert got == M.norm(exp)


def test_background_tier_switches_without_changing_results(tc):
    """auto mode: a plan that keeps running on a small input is specialised by the worker thread from its third run on; the
    rows before, while and after the switch are the oracle's, and the specialised kernels do get launched afterwards."""
    st = tc.ctx.jit_stats()
    if not st["available"]:
        pytest.skip("hiprtc not available")
    import arrow_ballista_amd as g
    from test_gpu_native_plan import arrow_rows

    def rows(res):
        return [tuple(r) for r in arrow_rows(res.to_arrow())]
    tc.ctx.set_jit("auto")
    n = 3000
    li = T.gen_lineitem_device(tc, n, seed=23)
    exp = T.q1_oracle_rows(n, seed=23)
    plan = g.NativePlan(T.q1_plan(g.MemoryExec([li]), two_phase=True), tc)
    before = tc.ctx.jit_stats()["launches"]
    for _ in range(4):
        assert rows(plan.execute(0)) == exp
    tc.ctx.jit_wait()
    mid = tc.ctx.jit_stats()["launches"]
    for _ in range(3):
        assert rows(plan.execute(0)) == exp
    assert tc.ctx.jit_stats()["launches"] > mid >= before
    # switched off: the counters stand still and the interpreter kernels give the same rows
    tc.ctx.set_jit("off")
    off = tc.ctx.jit_stats()["launches"]
    assert rows(plan.execute(0)) == exp and tc.ctx.jit_stats()["launches"] == off
    tc.ctx.set_jit("auto")


def test_code_objects_are_cached_on_disk(tmp_path):
    """A second PROCESS does not pay hiprtc for a pipeline the first one compiled: code objects are kept under
    $GPUQ_JIT_CACHE_DIR.  Same q1 rows from both processes; the first compiles and fills the directory, the second only loads."""
    import json
    import os
    import subprocess
    import sys
    prog = r"""
import ctypes as C, json, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import tpch_util as T, arrow_ballista_amd as g
tc = g.TaskContext(device=0)
if not tc.ctx.jit_stats()["available"]:
    print(json.dumps({"skip": True})); sys.exit(0)
tc.ctx.set_jit("force")
li = T.gen_lineitem_device(tc, 50_000, seed=3)
rows = T.q1_result_to_rows(tc, T.run_q1(tc, li))
h, c = C.c_int(0), C.c_int(0)
tc.ctx.L.gpuq_jit_cache_stats(C.byref(h), C.byref(c))
print(json.dumps({"ok": rows == T.q1_oracle_rows(50_000, seed=3), "disk_hits": h.value, "compiles": c.value}))
"""
    env = dict(os.environ, GPUQ_JIT_CACHE_DIR=str(tmp_path / "jit"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", prog], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    if outs[0].get("skip"):
        pytest.skip("hiprtc not available")
    assert outs[0]["ok"] and outs[1]["ok"]
    assert outs[0]["compiles"] > 0 and outs[0]["disk_hits"] == 0
    assert outs[1]["compiles"] == 0 and outs[1]["disk_hits"] == outs[0]["compiles"]
    assert len([f for f in os.listdir(tmp_path / "jit") if f.endswith(".co")]) == outs[0]["compiles"]
