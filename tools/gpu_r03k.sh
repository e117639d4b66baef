#!/bin/bash
# round 3 step k: word-wise string loads: the whole GPU suite, then the q3 step
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03k/bench.json").read().strip().splitlines()[-1])
print("q3 ms_per_step %.3f" % d["ms_per_step"], "probe %.3f" % d["roofline"]["avg_launch_ms"], d["host"], d["check"]["sum_revenue_matches"], [(o["label"] or o["op"], round(o["op_ms_per_step"], 3)) for o in d["operators"]])
PY
