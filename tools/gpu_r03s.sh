#!/bin/bash
# round 3 step s: SF300 sort shard with the key column rebuilt from the sorted records
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
timeout -k 10 600 python - > $O/sort.json <<'PY'
import json, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import arrow_ballista_amd as g
import tpch_util as T
import bench_extras
tc = g.TaskContext(device=0)
tc.ctx.set_jit("wait")
print(json.dumps({"sort_sf300_shard": bench_extras.sort_shard(tc, T, g), "sort_micro": bench_extras.sort_micro(tc, T, g)}, indent=1))
PY
cat $O/sort.json
