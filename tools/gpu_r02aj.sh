#!/bin/bash
# round 2 step aj: SortExec with the key layout guessed from a sample (no full min/max pass): parity + 2^27 timings, guess on / off
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02aj; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sort.py tests/test_gpu_native_plan.py tests/test_gpu_operators.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -q -m gpu -k "sort or order or guessed or q3 or distributed" > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 600 python bench_extras.py --sort > $O/sort.json 2> $O/sort.err || { tail -20 $O/sort.err; exit 1; }
cat $O/sort.json | tr -d "\n "; echo
GPUQ_SORT_SPECULATE=0 timeout -k 10 600 python bench_extras.py --sort > $O/sort_exact.json 2> $O/sort.err || { tail -20 $O/sort.err; exit 1; }
cat $O/sort_exact.json | tr -d "\n "; echo
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o sort -- python3 $GRAFT_REPO_ROOT/bench_extras.py --sort > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof.err || { tail -5 $GRAFT_REPO_ROOT/$O/prof.err; exit 1; }
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/sort_kernel_stats.csv; head -12 $O/sort_kernel_stats.csv | cut -c1-150
