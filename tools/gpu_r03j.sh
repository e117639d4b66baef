#!/bin/bash
# round 3 step j: column pruning at joins / exchanges + chain fusion over a view build side: parity suites, q5 and q3 timings
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_join_chain.py tests/test_gpu_deferred.py tests/test_gpu_native_plan.py tests/test_gpu_tpch.py tests/test_gpu_tpch_more.py tests/test_gpu_vs_acero.py tests/test_gpu_distributed.py tests/test_gpu_long_string_keys.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 300 python tools/q5_step.py 100 > $O/q5.txt 2>&1 || { tail -20 $O/q5.txt; exit 1; }
cat $O/q5.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03j/bench.json").read().strip().splitlines()[-1])
print("q3 ms_per_step %.3f" % d["ms_per_step"], "probe %.3f" % d["roofline"]["avg_launch_ms"], d["host"], d["check"]["sum_revenue_matches"], [(o["label"] or o["op"], round(o["op_ms_per_step"], 3)) for o in d["operators"]])
PY
