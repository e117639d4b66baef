#!/bin/bash
# round 3 step p: SF100 q6 (pure scan) and a memory-stats reading of the SF100 q3 step
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 600 python - > $O/q6.json 2> $O/q6.err <<'PY' || { tail -20 $O/q6.err; exit 1; }
import json, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import arrow_ballista_amd as g
import tpch_util as T
import bench_extras
tc = g.TaskContext(device=0)
tc.ctx.set_jit("wait")
out = {"q6": bench_extras.q6_pipeline(tc, T, g, 100)}
g.memory_stats(reset_peak=True)
tp = bench_extras.tpch_pipelines(tc, T, g, 100)
out["q3"], out["q5"] = tp["q3"], tp["q5"]
out["memory_after_q3_q5"] = g.memory_stats()
print(json.dumps(out))
PY
cat $O/q6.json
