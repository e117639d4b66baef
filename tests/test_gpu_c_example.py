"""examples/q1_native.c -- a C99 program that drives TPC-H q1 through the C ABI alone (device buffers, generator, native
plan executor, result read-back): the boundary exercised exactly as a cgo / JNI / Rust FFI binding would, no Python or torch
in the process.  Its output must be the oracle's q1 result."""
import os
import subprocess

import pytest

import tpch_util as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [1000, 300_000])
def test_q1_through_the_c_abi_only(n):
    exe = os.path.join(ROOT, "examples", "q1_native")
    if not os.path.exists(exe):
        import __graft_entry__ as E
        E._build_c_example()
    r = subprocess.run([exe, os.path.join(ROOT, "examples", "q1_plan.json"), str(n), "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = []
    for line in r.stdout.strip().splitlines():
        f = line.split("|")
        got.append((f[0], f[1]) + tuple(int(x) for x in f[2:]))
    assert got == T.q1_oracle_rows(n, seed=1)
    assert "AggregateExec" in r.stderr         # per-node metrics came back
