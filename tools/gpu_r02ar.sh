#!/bin/bash
# round 2 step ar: bitmap-mode join build with a non-returning atomic (duplicates found by counting bits afterwards): parity + SF100 q3
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ar; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_tpch.py tests/test_gpu_operators.py tests/test_gpu_fullsize.py tests/test_gpu_native_plan.py tests/test_gpu_distributed.py -q -m gpu -x > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/b_$i.json 2> $O/b_$i.err || { tail -20 $O/b_$i.err; exit 1; }
  python3 -c "
import json
l=json.loads([x for x in open('$O/b_$i.json') if x.startswith('{')][-1])
print('run $i: step', round(l['ms_per_step'],3), 'probe', round(l['roofline']['avg_launch_ms'],3), [(o['op'], round(o['kernel_ms_per_step'],3)) for o in l['operators'][1:6]])"
done
timeout -k 10 900 python bench_extras.py --sf100 > $O/sf100.json 2> $O/sf100.err || { tail -20 $O/sf100.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/sf100.json')); print({k:round(d[k]['wall_ms_best'],3) for k in ('q1','q3','q5')})"
