#!/bin/bash
# round 3 step g: the random-gather ceiling of the part next to the uniform-key probe grid; non-temporal streaming loads A/B on the q3 step
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 300 tools/gather_ceiling > $O/gather_ceiling.txt 2>&1 || { tail $O/gather_ceiling.txt; exit 1; }
cat $O/gather_ceiling.txt
for v in base nt; do
  if [ $v = nt ]; then export GPUQ_JIT_DEFINES="GPUQ_NT_STREAM=1"; else unset GPUQ_JIT_DEFINES; fi
  GPUQ_JIT_CACHE_DIR=off timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
done
python3 - <<'PY'
import json
for n in ("base", "nt"):
    d = json.loads(open("gpurun_out/r03g/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "ms_per_step %.3f" % d["ms_per_step"], "probe %.3f" % d["roofline"]["avg_launch_ms"], [(o["label"] or o["op"], round(o["op_ms_per_step"], 3)) for o in d["operators"]])
PY
