#!/bin/bash
# round 2 step ab: rows in flight in the hash aggregate and the generic unique probe: parity + SF100 q1/q3/q5 + A/B (GPUQ_ROWS_U)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ab; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_join_tables.py tests/test_gpu_tpch.py tests/test_gpu_native_plan.py tests/test_gpu_q1.py tests/test_gpu_long_string_keys.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for v in 1 2 1 2; do
  GPUQ_JIT_DEFINES="GPUQ_ROWS_U=$v" timeout -k 10 300 python bench_extras.py --sf100 > $O/sf100_$v.json 2> $O/sf100_$v.err || { tail -20 $O/sf100_$v.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/sf100_$v.json')); print('rows_u=$v', {k: round(v['wall_ms_best'],2) for k,v in d.items()})"
done
