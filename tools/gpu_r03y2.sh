#!/bin/bash
# after force-inlining the ZSTD decoder: parity, SF1 timing; then the bench's first-run time in two fresh processes
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03y; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scan_decode.py -m gpu -x -q > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
CODEC_PER_COLUMN=1 timeout -k 10 300 python tools/codec_sf.py 1 ZSTD > $O/zstd_sf1.json 2> $O/zstd_sf1.err || { tail -20 $O/zstd_sf1.err; exit 1; }
cat $O/zstd_sf1.json
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err || { tail -20 $O/bench$i.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/bench$i.json') if l.startswith('{')][-1]); print('bench $i', d['ms_per_step'], d['config'].get('first_run_ms'))"
done
