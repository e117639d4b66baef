#!/bin/bash
# the final build's bench in six fresh processes on one box: step, probe launch, first run
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03s2; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/b$i.json 2> $O/b$i.err || { tail -20 $O/b$i.err; exit 1; }
  python3 - $O/b$i.json $i >> $O/spread.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print(json.dumps({"run": int(sys.argv[2]), "ms_per_step": round(d["ms_per_step"], 4), "probe_ms": round(r["avg_launch_ms"], 4), "frac": round(r["frac"], 4), "first_run_ms": round(d["config"]["first_run_ms"], 1), "check": d["check"]["sum_revenue_matches"] and d["check"]["groups_match"]}))
PY
  tail -1 $O/spread.txt
done
