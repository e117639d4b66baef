#!/bin/bash
# round 2 step p: the whole -m gpu suite + smoke, then cold vs warm start of the q3 plan (on-disk code-object cache)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python -c "import __graft_entry__ as e; e.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
export GPUQ_JIT_CACHE_DIR=/tmp/gpuq_jit_cache_$$
for run in cold warm; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/start_$run.json 2> $O/start_$run.err || { tail -20 $O/start_$run.err; exit 1; }
  python3 -c "import json,sys; l=json.loads([x for x in open('$O/start_$run.json') if x.startswith('{')][-1]); print('$run', 'first_run_ms', l['config']['first_run_ms_cold_jit'], 'ms_per_step', l['ms_per_step'])"
done
