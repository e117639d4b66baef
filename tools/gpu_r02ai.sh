#!/bin/bash
# round 2 step ai: merge-path rounds with the tile boundaries found by a partition kernel: parity + round time (rounds forced)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ai; mkdir -p $O
GPUQ_MERGE_ROUNDS=force timeout -k 10 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_native_plan.py tests/test_gpu_operators.py -q -m gpu -k "merge or coalesce or fan_in" > $O/tests_forced.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests_forced.log | head -40; tail -5 $O/tests_forced.log; exit 1; }
tail -1 $O/tests_forced.log
timeout -k 10 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_native_plan.py tests/test_gpu_operators.py tests/test_gpu_distributed.py -q -m gpu -k "merge or coalesce or fan_in or distributed" > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
GPUQ_MERGE_ROUNDS=force timeout -k 10 600 python bench_extras.py --merge 27 > $O/merge_forced.json 2> $O/merge.err || { tail -20 $O/merge.err; exit 1; }
cat $O/merge_forced.json | tr -d "\n "; echo
timeout -k 10 600 python bench_extras.py --merge 27 > $O/merge.json 2> $O/merge.err || { tail -20 $O/merge.err; exit 1; }
cat $O/merge.json | tr -d "\n "; echo
