#!/bin/bash
# round 3 step t: hash-aggregate table load factor on SF100 q3 (GPUQ_AGG_SLOT_PCT: slots per estimated group, percent, before the power-of-two rounding)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
for v in 200 130 110; do
  export GPUQ_AGG_SLOT_PCT=$v
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r03t/bench.json").read().strip().splitlines()[-1])
print("slot_pct %-5s ms_per_step %.3f" % (sys.argv[1], d["ms_per_step"]), "check", d["check"]["sum_revenue_matches"], [(o["label"] or o["op"], round(o["op_ms_per_step"], 3)) for o in d["operators"][:4]])
PY
done
