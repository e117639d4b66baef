#!/bin/bash
# round 2 step l: scan decode after the LDS-staged CSV parser: tests, throughput, kernel trace of the throughput run
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_scan_decode.py -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 500 python bench_extras.py --scan > $O/scan.json 2> $O/scan.err || { tail -30 $O/scan.err; exit 1; }
cat $O/scan.json
