#!/bin/bash
# round 2 step y: SQ counters of the lineitem probe (is it ALU- or memory-bound now?)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02y; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc1 -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc1.log 2>&1 || { tail -20 $O/pmc1.log; exit 1; }
python3 tools/pmc_summary.py $O/pmc1 "join_probe_unique"
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc2 -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc2.log 2>&1 || { tail -20 $O/pmc2.log; exit 1; }
python3 tools/pmc_summary.py $O/pmc2 "join_probe_unique"
