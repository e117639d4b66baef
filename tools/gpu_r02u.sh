#!/bin/bash
# round 2 step u: rows in flight in the build / key-range kernels (GPUQ_ROWS_U), A/B on one box + join parity
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02u; mkdir -p $O
show() { python3 -c "
import json,sys
l=json.loads([x for x in open('$1') if x.startswith('{')][-1])
print('$2', 'ms_per_step', round(l['ms_per_step'],3), [(o['op'], round(o['kernel_ms_per_step'],3)) for o in l['operators'][:6]])"; }
for v in 1 4 2 1 4; do
  GPUQ_JIT_DEFINES="GPUQ_ROWS_U=$v" timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
  show $O/bench_$v.json rows_u=$v
done
timeout -k 10 600 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_tpch.py tests/test_gpu_operators.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
