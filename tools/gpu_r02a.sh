#!/bin/bash
# GPU session r02a: baseline of the new headline (SF100 q3) + counters for the direct probe, before the radix join exists.
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
echo "bench done"; tail -c 600 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/q3_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/q3_trace.log 2>&1
echo "trace done"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-20)
  rocprofv3 --pmc $C --output-format csv -d $O/q3_pmc_$T -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > $O/q3_pmc_$T.log 2>&1
  echo "q3 pmc $T done"
  rocprofv3 --pmc $C --output-format csv -d $O/micro_pmc_$T -- python3 bench_extras.py --probe-micro 24 27 > $O/micro_pmc_$T.log 2>&1
  echo "micro pmc $T done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/micro_trace -- python3 bench_extras.py --probe-micro 20 24 27 > $O/micro_trace.log 2>&1
echo "all done"
