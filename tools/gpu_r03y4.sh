#!/bin/bash
# SQ counters of the ZSTD page kernel (one counter pass; no traces alongside): what are a sequence's ~2,400 cycles made of?
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03y4; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc -- python3 tools/codec_prof.py ZSTD > $O/log.txt 2>&1 || { tail -20 $O/log.txt; exit 1; }
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "zstd" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
for k, d in acc.items():
    print(k, "dispatches", calls[k]); [print("  %-20s %.4g per dispatch" % (c, v / max(calls[k], 1))) for c, v in sorted(d.items())]
PY
rm -f $O/pmc/*/*.db 2>/dev/null; du -sh $O
