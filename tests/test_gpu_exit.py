"""Process exit with a background specialisation in flight (DESIGN.md section 7 "Exit order"): hiprtc / comgr are dlopen()ed on the first
compile, so their static destructors run before this library's at exit; a worker thread still inside hiprtc at that moment crashed
one benchmark run after it had printed its result.  The binding registers gpuq_jit_quiesce with atexit (a C host calls it before
returning from main, the Rust shim from Drop).  This test does not try to reproduce the crash: it checks the fix's contract -- a child
process that queues background compiles and exits AT ONCE must exit with status 0, and so must one that is told to skip Python's atexit
handlers but calls gpuq_jit_quiesce itself."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, pyarrow as pa
import arrow_ballista_amd as g
from arrow_ballista_amd.expr import col, lit, binary, Operator as Op
tc = g.TaskContext(device=0)
t = pa.table({"k": pa.array(np.arange(1000) %% 7, pa.int64()), "v": pa.array(np.arange(1000), pa.int64())})
src = g.MemoryExec([t]); s = src.schema()
plans = []
for j in range(6):          # six distinct small pipelines: each becomes "hot" on its third run and is queued for the worker thread
    agg = g.AggregateExec("Single", [(col("k", s), "k")], [{"fn": "SUM", "expr": binary(col("v", s), Op.Plus, lit(j)), "name": "s"}], src)
    plans.append(g.NativePlan(agg, tc))
for p in plans:
    for _ in range(4):
        p.execute(0)
avail, launches, err = tc.ctx.jit_stats() if hasattr(tc.ctx, "jit_stats") else (1, 0, "")
print("queued", flush=True)
%(tail)s
'''


def _run(tail):
    env = dict(os.environ, GPUQ_JIT_CACHE_DIR="off")          # every compile really runs: nothing comes from the on-disk cache
    return subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tail": tail}], capture_output=True, text=True, timeout=300, env=env)


def test_exit_right_after_queueing_background_compiles():
    r = _run("sys.exit(0)")
    assert "queued" in r.stdout, r.stderr[-2000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])


def test_exit_without_atexit_after_an_explicit_quiesce():
    r = _run("g.lib().gpuq_jit_quiesce(); sys.stdout.flush(); os._exit(0)")
    assert "queued" in r.stdout, r.stderr[-2000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
