#!/bin/bash
# round 2 step ac: hash repartition as a packed single-read pass: parity (partition, shuffle, exchange, distributed) + micro
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ac; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_fullsize.py tests/test_gpu_exchange.py tests/test_gpu_distributed.py tests/test_gpu_native_plan.py tests/test_gpu_shuffle_codec.py -q -m gpu -k "partition or shuffle or exchange or distributed or stage or repartition" > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/micro_ops.py > $O/micro.log 2>&1 || { tail -20 $O/micro.log; exit 1; }
grep -E "partition|sort|agg" $O/micro.log | head -12
