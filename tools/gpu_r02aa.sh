#!/bin/bash
# round 2 step aa: tile size of the packed single-read pass (8192 vs 6144 records), then the sort suites
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02aa; mkdir -p $O
for v in 32 24 32 24; do
  GPUQ_SORT_ROUNDS=$v timeout -k 10 300 python bench_extras.py --sort 27 > $O/sort_$v.json 2> $O/sort_$v.err || { tail -20 $O/sort_$v.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/sort_$v.json')); print('rounds=$v', round(d['by_extendedprice']['ms'],3), round(d['by_orderkey_desc_shipdate']['ms'],3))"
done
timeout -k 10 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_operators.py tests/test_gpu_fullsize.py tests/test_gpu_tpch.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
