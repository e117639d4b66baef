#!/bin/bash
# round 3 step c: chain fusion -- parity, then the q3 step with / without it and a trace of the fused step
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_join_chain.py tests/test_gpu_deferred.py tests/test_gpu_native_plan.py tests/test_gpu_tpch.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -5 $O/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_chain.json 2> $O/bench_chain.err || { tail -20 $O/bench_chain.err; exit 1; }
GPUQ_JOIN_CHAIN=0 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_nochain.json 2> $O/bench_nochain.err || { tail -20 $O/bench_nochain.err; exit 1; }
python3 - <<'PY'
import json
for n in ("chain", "nochain"):
    d = json.loads(open("gpurun_out/r03c/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "ms_per_step %.3f" % d["ms_per_step"], "probe %.3f" % d["roofline"]["avg_launch_ms"], "groups", d["config"]["result_groups"], [(o["op"], round(o["kernel_ms_per_step"], 3)) for o in d["operators"]])
PY
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q3_trace -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras > $O/q3_trace.log 2>&1 || { tail -20 $O/q3_trace.log; exit 1; }
f=$(find $O/q3_trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_step.py "$f" 0 2 > $O/step.txt
cat $O/step.txt
