#!/bin/bash
# round 2 step z: lane-major rows in the staged probe (16-byte loads), A/B on one box + join parity
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02z; mkdir -p $O
show() { python3 -c "
import json,sys
l=json.loads([x for x in open('$1') if x.startswith('{')][-1])
print('$2', 'ms_per_step', round(l['ms_per_step'],3), 'probe', round(l['roofline']['avg_launch_ms'],3), 'frac', round(l['roofline']['frac'],3), [(o['op'], round(o['kernel_ms_per_step'],3)) for o in l['operators'][:4]])"; }
for v in 0 1 0 1; do
  GPUQ_JIT_DEFINES="GPUQ_PROBE_LANE_MAJOR=$v" timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
  show $O/bench_$v.json lane_major=$v
done
for v in 0 1; do
  GPUQ_JIT_DEFINES="GPUQ_PROBE_LANE_MAJOR=$v" timeout -k 10 300 python bench_extras.py --probe-micro 20 24 27 > $O/micro_$v.json 2> $O/micro_$v.err || { tail -20 $O/micro_$v.err; exit 1; }
  python3 -c "
import json
print('lane_major=$v', [(p['build_rows'], round(p['probe_ms'],2), round(p['probe_rows_per_s']/1e9,1), p['pairs_valid']) for p in json.load(open('$O/micro_$v.json'))])"
done
timeout -k 10 600 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_tpch.py tests/test_gpu_fullsize.py tests/test_gpu_native_plan.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
