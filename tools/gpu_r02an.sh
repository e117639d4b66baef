#!/bin/bash
# round 2 step an: ORDER BY Utf8 keys longer than 15 bytes (native SortExec, stable passes over 14-byte pieces); sort refuses instead of truncating
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02an; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_long_string_keys.py tests/test_gpu_operators.py tests/test_gpu_native_plan.py tests/test_gpu_tpch.py tests/test_gpu_jit.py tests/test_gpu_q1.py -q -m gpu -x > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -60; tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
