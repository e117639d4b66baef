#!/bin/bash
# the abort at process exit seen twice after a Snappy decode + pq.read_table: which library calls std::terminate?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03z; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5; do
  LD_PRELOAD=$PWD/tools/diag/term_trace.so timeout -k 10 200 python -X faulthandler tools/codec_sf.py 1 SNAPPY > $O/run$i.json 2> $O/run$i.err
  echo "run $i rc $?" | tee -a $O/rc.txt
done
grep -l "std::terminate" $O/*.err | head
