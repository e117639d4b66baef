#!/bin/bash
# round 2 step ao: the N > 1 bench path rehearsed on one GPU (2 ranks share it, gloo + host-staged transport) at a size where the guessed key ranges are live
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ao; mkdir -p $O
GPUQ_BENCH_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 3 --warmup 1 --sf 40 > $O/bench_n2.json 2> $O/bench_n2.err || { tail -30 $O/bench_n2.err; exit 1; }
tail -c 2500 $O/bench_n2.json
