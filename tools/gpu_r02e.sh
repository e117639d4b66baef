#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_join_tables.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 bench_extras.py --probe-micro 24 27 --radix force --slice 18 > $O/log.txt 2>&1
f=$(find $O/tr -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-60s calls=%s avg_us=%.1f min=%.1f max=%.1f" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
