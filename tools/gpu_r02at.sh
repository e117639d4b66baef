#!/bin/bash
# round 2 step at: SQ counters + HBM traffic of the join build kernel (what bounds 0.7 ms for 14.7 M rows?)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02at; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc1 -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc1.log 2>&1 || { tail -20 $O/pmc1.log; exit 1; }
python3 tools/pmc_summary.py $O/pmc1 "join_build"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc2 -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc2.log 2>&1 || { tail -20 $O/pmc2.log; exit 1; }
python3 tools/pmc_summary.py $O/pmc2 "join_build"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc3 -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc3.log 2>&1 || { tail -20 $O/pmc3.log; exit 1; }
python3 tools/pmc_summary.py $O/pmc3 "join_build"
