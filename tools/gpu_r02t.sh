#!/bin/bash
# round 2 step t: rows in flight per lane of the staged probe (GPUQ_PROBE_ROWS), A/B on one box
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02t; mkdir -p $O
show() { python3 -c "
import json,sys
l=json.loads([x for x in open('$1') if x.startswith('{')][-1])
print('$2', 'ms_per_step', round(l['ms_per_step'],3), 'probe_ms', round(l['roofline']['avg_launch_ms'],3), 'frac', round(l['roofline']['frac'],3), [(o['op'], round(o['kernel_ms_per_step'],3)) for o in l['operators'][:5]])"; }
for v in 4 8 2 6 4 8; do
  GPUQ_JIT_DEFINES="GPUQ_PROBE_ROWS=$v" timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
  show $O/bench_$v.json rows=$v
done
