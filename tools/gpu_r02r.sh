#!/bin/bash
# round 2 step r: Snappy pages, multi-partition CollectLeft in the native executor
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scan_decode.py tests/test_gpu_native_plan.py tests/test_gpu_shuffle_codec.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python bench_extras.py --scan > $O/scan.json 2> $O/scan.err || { tail -30 $O/scan.err; exit 1; }
grep -A3 '"parquet' $O/scan.json | head -30
