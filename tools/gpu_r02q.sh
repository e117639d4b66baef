#!/bin/bash
# round 2 step q: distributed q5 (2-rank native test) + N=2 rehearsal of bench.py over gloo with the q5 leg
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
GPUQ_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 3 --warmup 1 --sf 2 > $O/bench_n2.json 2> $O/bench_n2.err || { tail -30 $O/bench_n2.err; exit 1; }
python3 -c "
import json
l=json.loads([x for x in open('$O/bench_n2.json') if x.startswith('{')][-1])
print(l['n_gpus'], l['scaling'], round(l['ms_per_step'],2), {k:(round(v['ms_per_step'],2), v['result_groups']) for k,v in l['extra'].items()})"
