#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/${1:-trace}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/q3_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/q3_trace.log 2>&1
f=$(find $O/q3_trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_last_run.py "$f" > $O/last_run.txt
cat $O/last_run.txt
