#!/bin/bash
# round 3 step h: non-temporal streaming loads, alternating A/B in six fresh processes (the probe's time depends on the process's memory layout)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
for i in 1 2 3; do for v in base nt; do
  if [ $v = nt ]; then export GPUQ_JIT_DEFINES="GPUQ_NT_STREAM=1"; else unset GPUQ_JIT_DEFINES; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err || { tail -20 $O/bench_${v}_$i.err; exit 1; }
done; done
python3 - <<'PY'
import json
for n in ("base", "nt"):
    for i in (1, 2, 3):
        d = json.loads(open("gpurun_out/r03h/bench_%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
        print(n, i, "ms_per_step %.3f" % d["ms_per_step"], "probe %.3f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], [(o["label"] or o["op"], round(o["op_ms_per_step"], 3)) for o in d["operators"][:4]])
PY
