#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
# N=2 rehearsal on the one GPU (host-staged transport), small scale; then the N=1 headline
GPUQ_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 3 --warmup 1 --sf 2 > $O/bench_n2.json 2> $O/bench_n2.err || { tail -30 $O/bench_n2.err; exit 1; }
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r02g/bench_n2.json') if l.startswith('{')][-1]); print('N2', d['ms_per_step'], d['value'], d['config']['parallelism'], d['extra'])"
python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02g/bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['first_run_ms_cold_jit'])
for o in d['operators']: print(o)
print(d['cpu_baseline'])
e=d['extra']
for j in e['join_probe']: print(j['build_rows'], j['probe_ms'], j['probe_rows_per_s'], j['build_ms'])
print(e['sf100_q1'], e['sf100_q3'], e['sf100_q5'])
PY
