#!/bin/bash
# round 2 step m: single-read radix passes: sort parity (new + existing sort tests), sort micro at 2^27, kernel trace
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_operators.py tests/test_gpu_fullsize.py -q -m gpu -k "sort or Sort" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python bench_extras.py --sort 27 > $O/sort.json 2> $O/sort.err || { tail -30 $O/sort.err; exit 1; }
cat $O/sort.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench_extras.py --sort 27 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -r head -16
