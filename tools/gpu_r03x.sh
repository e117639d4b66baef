#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_shuffle_codec.py -q -x 2>&1 | tail -4 || exit 1
timeout -k 10 300 python tools/lz4_linked.py | tee $O/linked.json
