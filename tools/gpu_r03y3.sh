#!/bin/bash
# kernel trace of three device decodes of the SF1 ZSTD file
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03y3; rm -rf $O; mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/codec_prof.py ZSTD > $O/log.txt 2>&1 || { tail -20 $O/log.txt; exit 1; }
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160
rm -f $O/trace/*/*kernel_trace.csv $O/trace/*/*.db 2>/dev/null; du -sh $O
