#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
timeout -k 10 900 python bench_extras.py --ingest --sf10 > $O/ingest.json 2> $O/ingest.err || { tail -30 $O/ingest.err; exit 1; }
cat $O/ingest.json
