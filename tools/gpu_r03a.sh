#!/bin/bash
# round 3 step a: where the q3 step goes before any change: every dispatch of one step with gaps
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q3_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/q3_trace.log 2>&1 || { tail -20 $O/q3_trace.log; exit 1; }
f=$(find $O/q3_trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_step.py "$f" > $O/step.txt
tail -40 $O/step.txt
tail -c 1200 $O/q3_trace.log
