"""Which accumulator makes h2o q4 (avg(v1), avg(v2), avg(v3) by id4, 100 groups) slow on the block-local path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, pyarrow as pa, torch
import arrow_ballista_amd as g
from arrow_ballista_amd.expr import col
tc = g.TaskContext(device=0)
n, k = 10_000_000, 100
r = np.random.default_rng(11)
x = pa.table({"id4": pa.array(r.integers(1, k + 1, n).astype(np.int32), pa.int32()), "v1": pa.array(r.integers(1, 6, n).astype(np.int32), pa.int32()),
              "v2": pa.array(r.integers(1, 16, n).astype(np.int32), pa.int32()), "v3": pa.array(np.round(r.random(n) * 100, 6), pa.float64())})
x = x.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in x.schema]))
dx = g.MemoryExec([g.DeviceTable.from_arrow(x, tc.device)])
s = dx.schema()
A = lambda fn, c: {"fn": fn, "expr": col(c, s), "name": fn + c}
for name, aggs in (("sum v1", [A("SUM", "v1")]), ("avg v1", [A("AVG", "v1")]), ("sum v3", [A("SUM", "v3")]), ("avg v3", [A("AVG", "v3")]),
                   ("avg v1 v2", [A("AVG", "v1"), A("AVG", "v2")]), ("avg v1 v2 v3", [A("AVG", "v1"), A("AVG", "v2"), A("AVG", "v3")]),
                   ("sum v1 v2 v3", [A("SUM", "v1"), A("SUM", "v2"), A("SUM", "v3")])):
    plan = g.NativePlan(g.AggregateExec("Single", [(col("id4", s), "id4")], aggs, dx), tc)
    for _ in range(3): plan.execute(0)
    plan.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): plan.execute(0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    ms, launches, desc = plan.profile(False)
    print("%-14s wall %.3f ms   dominant kernel %.3f ms x %d" % (name, dt, ms / max(1, launches), launches))
