#!/bin/bash
# round 3 step r: LIKE kernel with the wave's strings staged in LDS: tests + micro
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "like or Like or LIKE or q13 or q16 or q9 or q2_ or q20 or long_literal or strings" 2>&1 | tail -5 || exit 1
timeout -k 10 300 python - > $O/like.json <<'PY'
import json, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import arrow_ballista_amd as g
import tpch_util as T
import bench_extras
tc = g.TaskContext(device=0)
print(json.dumps(bench_extras.like_micro(tc, T, g), indent=1))
PY
cat $O/like.json
