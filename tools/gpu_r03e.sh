#!/bin/bash
# round 3 step e: the N>1 bench path rehearsed with 2 ranks sharing the GPU (host-staged transport), then the N=1 line
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
GPUQ_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 5 --warmup 2 --sf 4 > $O/bench_n2.json 2> $O/bench_n2.err || { tail -30 $O/bench_n2.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03e/bench_n2.json").read().strip().splitlines()[-1])
print("N=2 (gloo rehearsal) ms_per_step %.3f" % d["ms_per_step"], d["config"]["parallelism"], "host", d.get("host"), "groups", d["config"]["result_groups"])
print({k: (round(v["ms_per_step"], 2), v.get("result_groups", v.get("rows_on_rank0"))) for k, v in d["extra"].items()})
PY
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03e/bench.json").read().strip().splitlines()[-1])
print("N=1 ms_per_step %.3f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], "operators share %.3f" % d["operators_share_of_step"], [(o["label"] or o["op"], round(o["op_ms_per_step"], 3)) for o in d["operators"]])
print("sort shard", d["extra"]["sort_sf300_shard"])
PY
