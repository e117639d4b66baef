#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_join_tables.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for cfg in "off 18" "force 16" "force 18" "force 20" "auto 18"; do
  set -- $cfg
  python bench_extras.py --probe-micro 24 27 --radix $1 --slice $2 > $O/micro_$1_$2.json 2>&1
  echo "== radix $1 slice $2"; grep -E "build_rows\"|probe_ms|pairs_valid|\"matches\"" $O/micro_$1_$2.json
done
