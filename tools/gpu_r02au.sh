#!/bin/bash
# round 2 step au: rows in flight per wave in the build / aggregate kernels (GPUQ_ROWS_U = 2 default, 4, 8): the build waits 90 % of its cycles
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02au; mkdir -p $O
for u in 2 4 8 2 4 8; do
  GPUQ_JIT_DEFINES="GPUQ_ROWS_U=$u" timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/b_$u.json 2> $O/b_$u.err || { tail -20 $O/b_$u.err; exit 1; }
  python3 -c "
import json
l=json.loads([x for x in open('$O/b_$u.json') if x.startswith('{')][-1])
print('U=$u: step', round(l['ms_per_step'],3), [(o['op'], round(o['kernel_ms_per_step'],3)) for o in l['operators'][:6]])"
done
