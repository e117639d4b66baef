#!/bin/bash
# round 3 step b: deferred execution -- parity of the new tests and the native plan suite, then the q3 step with and without it
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_native_plan.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -5 $O/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_defer.json 2> $O/bench_defer.err || { tail -20 $O/bench_defer.err; exit 1; }
GPUQ_DEFER=0 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_sync.json 2> $O/bench_sync.err || { tail -20 $O/bench_sync.err; exit 1; }
python3 - <<'PY'
import json
for n in ("defer", "sync"):
    d = json.loads(open("gpurun_out/r03b/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "ms_per_step %.3f" % d["ms_per_step"], "probe %.3f" % d["roofline"]["avg_launch_ms"], "groups", d["config"]["result_groups"])
PY
