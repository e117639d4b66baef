#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03v; rm -rf $O; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/snappy_prof.py > $O/log.txt 2>&1 || { tail -20 $O/log.txt; exit 1; }
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-140
