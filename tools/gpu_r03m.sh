#!/bin/bash
# round 3 step m: hit queue in the unique probe: join parity suites, then the q3 step three times
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_join_chain.py tests/test_gpu_vs_acero.py tests/test_gpu_tpch.py tests/test_gpu_deferred.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$i.json 2> $O/bench_$i.err || { tail -20 $O/bench_$i.err; exit 1; }
done
python3 - <<'PY'
import json
for i in (1, 2, 3):
    d = json.loads(open("gpurun_out/r03m/bench_%d.json" % i).read().strip().splitlines()[-1])
    print("q3 ms_per_step %.3f" % d["ms_per_step"], "probe %.3f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], d["check"]["sum_revenue_matches"], [(o["label"] or o["op"], round(o["op_ms_per_step"], 3)) for o in d["operators"]])
PY
