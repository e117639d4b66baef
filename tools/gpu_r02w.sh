#!/bin/bash
# round 2 step w: LIKE with LDS-staged strings: parity + micro
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02w; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -k "like or Like or LIKE" > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -15 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python bench_extras.py --like > $O/like.json 2> $O/like.err || { tail -20 $O/like.err; exit 1; }
cat $O/like.json
