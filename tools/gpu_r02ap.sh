#!/bin/bash
# round 2 step ap: which table a guessed / measured join build ends up with (GPUQ_TRACE_JOIN_BUILD), join suites after the descriptor reset
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ap; mkdir -p $O
GPUQ_TRACE_JOIN_BUILD=1 timeout -k 10 600 python -m pytest tests/test_gpu_join_tables.py -q -m gpu -x -s -k guessed > $O/trace.log 2>&1 || { tail -40 $O/trace.log; exit 1; }
grep -E "gpuq\]|passed|failed" $O/trace.log | sort | uniq -c
timeout -k 10 900 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_tpch.py tests/test_gpu_operators.py tests/test_gpu_fullsize.py -q -m gpu -x > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
