#!/bin/bash
# round 2 step ak: sort min/max + pack kernels with U words of rows in flight per wave (U = 1 / 2 / 4), on top of the guessed key layout
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ak; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sort.py tests/test_gpu_native_plan.py tests/test_gpu_operators.py -q -m gpu -k "sort or order or guessed or stable" > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for u in 1 2 4; do
  GPUQ_JIT_DEFINES="GPUQ_SORT_ROWS_U=$u" timeout -k 10 600 python bench_extras.py --sort > $O/sort_u$u.json 2> $O/sort.err || { tail -20 $O/sort.err; exit 1; }
  echo "U=$u $(cat $O/sort_u$u.json | tr -d '\n ')"
  GPUQ_SORT_SPECULATE=0 GPUQ_JIT_DEFINES="GPUQ_SORT_ROWS_U=$u" timeout -k 10 600 python bench_extras.py --sort > $O/sort_exact_u$u.json 2> $O/sort.err || { tail -20 $O/sort.err; exit 1; }
  echo "U=$u exact $(cat $O/sort_exact_u$u.json | tr -d '\n ')"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench_extras.py --sort 27 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/sort_kernel_stats.csv; head -12 $O/sort_kernel_stats.csv | cut -c1-170
rm -rf $O/trace
