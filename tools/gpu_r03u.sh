#!/bin/bash
# round 3 step u: Snappy pages by pointer jumping: tests + the SF1 scan micro, against the serial decoder (GPUQ_SNAPPY_PJ=0)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_scan_decode.py tests/test_gpu_types2.py -q -x 2>&1 | tail -5 || exit 1
for v in 1 0; do
GPUQ_SNAPPY_PJ=$v timeout -k 10 500 python - > $O/scan_$v.json <<'PY'
import json, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import arrow_ballista_amd as g
import tpch_util as T
import bench_extras
tc = g.TaskContext(device=0)
print(json.dumps(bench_extras.scan_decode(tc, T, g, sf=1), indent=1))
PY
echo "GPUQ_SNAPPY_PJ=$v"; grep -i "snappy\|ms\b\|_ms" $O/scan_$v.json | head -20
done
