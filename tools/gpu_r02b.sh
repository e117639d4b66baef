#!/bin/bash
# GPU session r02b: parity of the direct-addressed join table + segmented pair emission, then the headline again
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_operators.py tests/test_gpu_jit.py tests/test_gpu_tpch.py tests/test_gpu_native_plan.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python bench.py --steps 10 --warmup 2 --no-extras > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02b/bench.json'))
print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
for o in d['operators']: print(o)
PY
python bench_extras.py --probe-micro 20 24 27 > $O/micro.json 2>&1
grep -E "build_rows|probe_ms|probe_rows_per_s|build_ms" $O/micro.json
