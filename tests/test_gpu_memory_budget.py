"""The memory budget (include/gpuq.h gpuq_memory_limit; the reference's operators run under DataFusion's MemoryPool and fail with
ResourcesExhausted when it is used up): every device byte the library holds is counted; the allocation that would cross the limit
fails the task with GPUQ_ERR_CAPACITY "Resources exhausted", the process and the next task go on."""
import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd.expr import col

pytestmark = pytest.mark.gpu


def test_a_sort_and_a_join_build_fail_loudly_under_a_budget_and_run_without(tc):
    r = np.random.default_rng(1)
    n = 2_000_000
    t = pa.table({"k": pa.array(r.integers(0, 1 << 40, n), pa.int64()), "v": pa.array(r.integers(0, 100, n), pa.int64())})
    src = g.MemoryExec([t]); s = src.schema()
    sort = g.SortExec([{"expr": col("k", s), "asc": True, "nulls_first": False}], src)
    small = g.MemoryExec([t.slice(0, 1000)])
    join = g.HashJoinExec(src, small, [(col("k", s), col("k", small.schema()))], None, "Inner", "CollectLeft", False)      # 2 M build rows
    base = g.memory_stats(reset_peak=True)
    assert base["limit"] == 0
    want = np.sort(t["k"].to_numpy())
    out = g.NativePlan(sort, tc).execute(0).to_arrow()
    assert np.array_equal(out["k"].to_numpy(), want)
    used = g.memory_stats()["peak"] - base["in_use"]
    assert used > 16 * n          # the sort holds at least its 16-byte records
    try:
        g.memory_limit(base["in_use"] + used // 4)
        for plan in (sort, join):
            with pytest.raises(g.GpuqError) as e:
                g.NativePlan(plan, tc).execute(0)
            assert e.value.status == 4 and "Resources exhausted" in str(e.value) and "memory limit" in str(e.value)
        # nothing leaked by the failed tasks: what is in use is what was in use (plus the plans' cached operators, a few KB)
        assert g.memory_stats()["in_use"] - base["in_use"] < (8 << 20)
    finally:
        g.memory_limit(0)
    out = g.NativePlan(sort, tc).execute(0).to_arrow()
    assert np.array_equal(out["k"].to_numpy(), want)
    assert g.NativePlan(join, tc).execute(0).num_rows == 1000
