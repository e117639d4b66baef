#!/bin/bash
# round 3 step n: what bounds the fused orders-side build: rows in flight, non-returning bitmap atomic (experiment switch)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
for v in base "GPUQ_SEMI_ROWS=8" "GPUQ_SEMI_ROWS=2" "GPUQ_EXP_NORETURN_OR=1" "GPUQ_EXP_NORETURN_OR=1;GPUQ_SEMI_ROWS=8"; do
  if [ "$v" = base ]; then unset GPUQ_JIT_DEFINES; else export GPUQ_JIT_DEFINES="$v"; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r03n/bench.json").read().strip().splitlines()[-1])
print("%-40s ms_per_step %.3f" % (sys.argv[1], d["ms_per_step"]), "check", d["check"]["sum_revenue_matches"], [(o["label"] or o["op"], round(o["op_ms_per_step"], 3)) for o in d["operators"][:3]])
PY
done
