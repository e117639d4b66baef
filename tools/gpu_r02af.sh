#!/bin/bash
# round 2 step af: builds without the `present` bitmap for joins that never ask for build-side rows: join suites + q3 bench
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02af; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_operators.py tests/test_gpu_tpch.py tests/test_gpu_native_plan.py tests/test_gpu_long_string_keys.py tests/test_gpu_distributed.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$i.json 2> $O/bench_$i.err || { tail -20 $O/bench_$i.err; exit 1; }
python3 -c "
import json
l=json.loads([x for x in open('$O/bench_$i.json') if x.startswith('{')][-1])
print('ms_per_step', round(l['ms_per_step'],3), [(o['op'], round(o['kernel_ms_per_step'],3)) for o in l['operators'][:6]])"
done
