#!/bin/bash
# round 3 step f: evidence for the headline: bench line, kernel trace + stats of the same command, PMC traffic passes for the probe kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03f; rm -rf $O/q3_trace $O/pmc_fetch $O/pmc_write; mkdir -p $O
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "bench done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q3_trace -- python3 bench.py --no-cpu-baseline --no-extras > $O/q3_trace.log 2>&1 || { tail -20 $O/q3_trace.log; exit 1; }
echo "trace done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
echo "fetch done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1 || { tail -20 $O/pmc_write.log; exit 1; }
echo "write done"
python3 tools/make_traffic.py $O/pmc_fetch $O/pmc_write $O/bench.json $O/traffic.json | tail -12
f=$(find $O/q3_trace -name "*kernel_stats.csv" | head -1); head -14 "$f"
t=$(find $O/q3_trace -name "*kernel_trace.csv" | head -1); python3 tools/trace_step.py "$t" 20 2 > $O/step.txt; tail -22 $O/step.txt
rm -rf $O/pmc_fetch/*/*.db $O/pmc_write/*/*.db 2>/dev/null; du -sh $O
