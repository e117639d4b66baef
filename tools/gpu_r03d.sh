#!/bin/bash
# round 3 step d: the full bench line (new fields) and the whole GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
( time timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time || { tail -30 $O/bench.err; exit 1; }
tail -3 $O/bench.time
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03d/bench.json").read().strip().splitlines()[-1])
print("ms_per_step %.3f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], d["roofline"]["kernel_name"], "host", d["host"], "check", d["check"])
print("operators sum %.3f" % d["operators_ms_per_step"], [(o["label"] or o["op"], round(o["kernel_ms_per_step"], 3)) for o in d["operators"]])
cb = d["cpu_baseline"]; print("cpu", cb["host"], {k: (round(cb[k]["wall_ms_min"], 1), round(cb[k]["wall_ms_mean"], 1)) for k in ("q1", "q3", "q5")}, cb.get("proxy_acero", {}).get("q3"), cb.get("proxy_acero", {}).get("q5"))
for e in d["extra"]["join_probe"]: print("probe", e["build_rows"], e["hit_rate"], e["probe_keys"], "%.2f ms" % e["probe_ms"], "%.1f G/s" % (e["probe_rows_per_s"] / 1e9), "frac %.3f" % e["frac_hbm_peak"])
print({k: round(v["wall_ms_best"], 2) for k, v in d["extra"].items() if k.startswith("sf100")})
PY
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
