#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_distributed.py tests/test_gpu_jit.py tests/test_gpu_shuffle_codec.py tests/test_gpu_native_plan.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
