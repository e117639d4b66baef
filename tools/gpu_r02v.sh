#!/bin/bash
# round 2 step v: long Utf8 keys through dictionary codes
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02v; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scan_decode.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -15 $O/tests.log; exit 1; }
tail -2 $O/tests.log
