#!/bin/bash
# round 2 step av: the driver's command as it will be run (default flags: cpu_baseline + extras), final build
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02av; mkdir -p $O
SECONDS=0; timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "wall seconds: $SECONDS"
python3 -c "
import json
l=json.loads([x for x in open('$O/bench.json') if x.startswith('{')][-1])
print('step', round(l['ms_per_step'],3), 'value', l['value'], 'frac', round(l['roofline']['frac'],3), 'cpu', round(l['cpu_baseline']['value']/1e9,3), 'G rows/s;', {k:(round(v['wall_ms_best'],2) if isinstance(v,dict) and 'wall_ms_best' in v else '...') for k,v in l['extra'].items()})"
