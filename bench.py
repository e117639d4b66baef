#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the gpuq operator engine on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched by
torch.distributed.run, one rank per GPU over RCCL.  Prints ONE JSON line on rank 0.

Workload at N=1 (BASELINE.json configs[1]): TPC-H SF10 q1 -- FilterExec -> ProjectionExec ->
AggregateExec(Partial) -> AggregateExec(FinalPartitioned) -> ProjectionExec -> SortExec over the 7
Arrow-physical lineitem columns q1 reads (78 B/row, 59,986,052 rows = 4.68 GB), inputs resident in HBM
when the timed region starts (synthetic TPC-H-shaped data produced on the device).
A "step" = one full q1 over the rank's rows.  N>1: weak scaling, every rank owns its own SF10 shard,
partial states are merged with one tiny all-gather (the path has no row exchange for q1).
value = rows processed by all ranks / wall time (max over ranks).

roofline: dominant kernel k_agg_tiny (fused filter+projection+partial aggregate); algorithmic bytes
= 78 B/row x rows per launch; duration = HIP events around the launch on the stream it runs on.
cpu_baseline: the C oracle's q1 ("port", OpenMP over all host cores) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

Q1_BYTES_PER_ROW = 78            # 4 x Decimal128 (64) + Date32 (4) + 2 x Utf8 (4 B offset + 1 B data)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: SF10 lineitem = 59,986,052)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=16_000_000)
    ap.add_argument("--extras", action="store_true", help="also time the join-probe / sort / partition micro-workloads")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import tpch_util as T
    import arrow_ballista_amd as g
    from arrow_ballista_amd import parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GPUQ_BENCH_BACKEND=gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices, the
    # collectives are staged through host memory); the measured configuration is always nccl = RCCL, one rank per GPU.
    backend = os.environ.get("GPUQ_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    tc = g.TaskContext(device=local_rank)

    n = args.rows or T.LINEITEM_ROWS[10]
    lineitem = T.gen_lineitem_device(tc, n, seed=T.SEED_LINEITEM, seed_orders=T.SEED_ORDERS, row0=rank * n)

    # ---- plans (built once).  Both stages run in the native plan executor (csrc/plan_exec.cpp): one library call per stage,
    # no Python between operators.
    STATE_CAP = 64      # rows of partial-aggregate state a rank ships; the same on every rank (fixes the record layout)
    partial_py, full_py, final_src = T.q1_split_plan(lineitem, STATE_CAP)
    partial = g.NativePlan(partial_py, tc)            # fused filter + projection + partial aggregate over the rank's rows

    def gather(res):
        states = res.to_device_table(tc.device)
        return parallel.allgather_table(states, cap=STATE_CAP)

    res0 = partial.execute(0)
    final_src.partitions[0] = gather(res0) if world > 1 else res0.to_device_table(tc.device)
    final = g.NativePlan(full_py, tc)                 # final aggregate + projection + sort over the (gathered) states

    def step():
        res = partial.execute(0)
        if world > 1:
            final.set_input(0, gather(res))            # one all-gather of fixed-layout records, read in place through a view
        else:
            final.set_input_result(0, res)
        return final.execute(0), res

    for _ in range(max(args.warmup, 3)):
        out, _r = step()
    # programs that keep running on small inputs (the final stage's) are specialised by a background thread from their third
    # run on: let those compiles finish inside the warm-up, as any JIT's would
    tc.ctx.jit_wait()
    out, _r = step()
    partial.profile(True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Python's cyclic GC walks the whole torch/pyarrow heap (~40 ms per full collection): keep it out of the timed region
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, _r = step()
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    kernel_ms, launches, _desc = partial.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=tc.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    rows_total = n * world * args.steps
    value = rows_total / dt
    result_rows = out.to_arrow().to_pylist()

    line = {
        "metric": "tpch_q1_operator_rows_per_sec", "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "i128", "data": "synthetic",
        "config": {"workload": "TPC-H SF10 q1 hash-aggregate (BASELINE configs[1]): filter+projection+partial/final aggregate+sort",
                   "rows_per_gpu": n, "bytes_per_row": Q1_BYTES_PER_ROW, "input": "Arrow-physical columns resident in HBM (64k-row batches concatenated at ingest)",
                   "groups": len(result_rows), "parallelism": "partition-per-gpu x%d, all-gather of partial states" % world},
    }
    if launches > 0:
        avg_ms = kernel_ms / launches
        achieved = Q1_BYTES_PER_ROW * n / (avg_ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes over this same
        # command, gfx950 correction applied; provenance in profiles/r01_traffic.json).  Only valid for the default workload.
        traffic = None
        tp = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if n == T.LINEITEM_ROWS[10] and os.path.exists(tp):
            traffic = json.load(open(tp)).get("traffic_bytes_per_launch")
        line["roofline"] = {"bound": "hbm", "kernel": "k_agg_tiny (hiprtc-specialised: gpuq_jit_agg_tiny)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_ms, "launches": launches,
                            "algorithmic_bytes_per_launch": Q1_BYTES_PER_ROW * n}

    if rank == 0 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(T, min(n, args.cpu_sample_rows))
    if args.extras and rank == 0:
        import bench_extras
        line["extra"] = bench_extras.run(tc, T, g)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(T, sample_rows):
    """Oracle (C restatement, OpenMP) q1 on the host cores over a bounded sample of the same workload."""
    host = T.gen_lineitem_host(sample_rows, seed=T.SEED_LINEITEM, seed_orders=T.SEED_ORDERS)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        T.q1_oracle_raw(sample_rows, host=host)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return {"value": sample_rows / best, "unit": "rows/s", "cores": T.oracle_lib().oracle_num_threads(), "kind": "port",
            "sample": "C oracle q1 (filter+project+group-by, int128 sums) over the first %d synthetic lineitem rows, best of 3, data in host memory" % sample_rows}


if __name__ == "__main__":
    main()
