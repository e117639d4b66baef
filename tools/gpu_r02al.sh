#!/bin/bash
# round 2 step al: join build with the key range guessed from a row sample (no measuring pass): parity + SF100 q3 / q5 with the guess on / off
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02al; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_tpch.py tests/test_gpu_native_plan.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py -q -m gpu -x > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<PY
import json; d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print("guess on ", d["ms_per_step"], d["roofline"]["avg_launch_ms"], [ (o["op"], round(o["kernel_ms_per_step"],3)) for o in d["operators"]])
PY
GPUQ_JOIN_SPECULATE=0 timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_exact.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<PY
import json; d=json.loads(open("$O/bench_exact.json").read().strip().splitlines()[-1]); print("guess off", d["ms_per_step"], d["roofline"]["avg_launch_ms"], [ (o["op"], round(o["kernel_ms_per_step"],3)) for o in d["operators"]])
PY
timeout -k 10 900 python bench_extras.py --sf100 > $O/sf100.json 2> $O/sf100.err || { tail -20 $O/sf100.err; exit 1; }
cat $O/sf100.json | tr -d "\n " | cut -c1-1500; echo
