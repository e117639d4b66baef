#!/bin/bash
# round 2 step ae: direct-addressed tables for sparser domains (ratio 4096): join suites, SF100 q3/q5 node times, probe micro
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ae; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_join_tables.py tests/test_gpu_operators.py tests/test_gpu_tpch.py tests/test_gpu_native_plan.py tests/test_gpu_fullsize.py tests/test_gpu_long_string_keys.py -q -m gpu > $O/tests.log 2>&1 || { grep -E "^E  |^FAILED|Error" $O/tests.log | head -40; tail -5 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python tools/sf100_metrics.py 2>&1 | grep -E "^q|HashJoin|Aggregate" > $O/metrics.log; cat $O/metrics.log
timeout -k 10 300 python bench_extras.py --probe-micro 20 24 27 > $O/micro.json 2> $O/micro.err || { tail -20 $O/micro.err; exit 1; }
python3 -c "
import json
print([(p['build_rows'], round(p['probe_ms'],2), round(p['probe_rows_per_s']/1e9,1), round(p['build_ms'],2)) for p in json.load(open('$O/micro.json'))])"
