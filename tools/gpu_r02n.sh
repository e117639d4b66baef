#!/bin/bash
# round 2 step n: ordered fan-in (merge-path) parity + the suites that use sort / coalesce / distributed sort
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02n; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sort.py tests/test_gpu_operators.py tests/test_gpu_native_plan.py tests/test_gpu_distributed.py tests/test_gpu_tpch.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
