#!/bin/bash
# ZSTD Parquet pages: parity tests, then lineitem SF1 under ZSTD (the reference's convert default) beside Snappy, 
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03y; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scan_decode.py -m gpu -x -q > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
CODEC_PER_COLUMN=1 timeout -k 10 300 python tools/codec_sf.py 1 ZSTD > $O/zstd_sf1.json 2> $O/zstd_sf1.err || { tail -20 $O/zstd_sf1.err; exit 1; }
cat $O/zstd_sf1.json
timeout -k 10 300 python tools/codec_sf.py 1 SNAPPY > $O/snappy_sf1.json 2> $O/snappy_sf1.err || { tail -20 $O/snappy_sf1.err; exit 1; }
cat $O/snappy_sf1.json
