#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scan_decode.py -q -x 2>&1 | tail -3 || exit 1
timeout -k 10 300 python tools/snappy_sf.py 1 | tee $O/sf1.json
timeout -k 10 900 python tools/snappy_sf.py 10 | tee $O/sf10.json
GPUQ_SNAPPY_PJ=0 timeout -k 10 900 python tools/snappy_sf.py 10 | tee $O/sf10_serial.json
