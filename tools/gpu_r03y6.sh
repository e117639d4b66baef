#!/bin/bash
# ZSTD kernel without the 32 KB output ring (more pages in flight per CU): parity, SF1 and SF10
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03y6; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scan_decode.py -m gpu -x -q > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
timeout -k 10 300 python tools/codec_sf.py 1 ZSTD > $O/zstd_sf1.json 2> $O/zstd_sf1.err || { tail -20 $O/zstd_sf1.err; exit 1; }
cat $O/zstd_sf1.json
timeout -k 10 500 python tools/codec_sf.py 10 ZSTD > $O/zstd_sf10.json 2> $O/zstd_sf10.err || { tail -20 $O/zstd_sf10.err; exit 1; }
cat $O/zstd_sf10.json
