#!/bin/bash
# round 2 step as: the bench line with the final build (three processes)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02as; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/b_$i.json 2> $O/b_$i.err || { tail -20 $O/b_$i.err; exit 1; }
  python3 -c "
import json
l=json.loads([x for x in open('$O/b_$i.json') if x.startswith('{')][-1])
print('run $i: step', round(l['ms_per_step'],3), 'probe', round(l['roofline']['avg_launch_ms'],3), 'frac', round(l['roofline']['frac'],3))"
done
