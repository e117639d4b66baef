#!/bin/bash
# round 2 step ad: the radix-partitioned aggregate on single-read passes: aggregate / sort / partition suites + micro
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ad; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_sort.py tests/test_gpu_fullsize.py tests/test_gpu_jit.py tests/test_gpu_native_plan.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/micro_ops.py > $O/micro.log 2>&1 || { tail -20 $O/micro.log; exit 1; }
grep -E "partition|sort|agg" $O/micro.log | head -12
