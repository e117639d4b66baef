#!/bin/bash
# round 3 step i: where SF100 q5 goes
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 300 python tools/q5_step.py 100 > $O/q5.txt 2>&1 || { tail -20 $O/q5.txt; exit 1; }
cat $O/q5.txt
