#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the gpuq operator engine on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched by
torch.distributed.run, one rank per GPU over RCCL.  Prints ONE JSON line on rank 0.

Workload at N=1 (BASELINE.json configs[2], the largest single-GPU configuration of the metric "rows/sec hash-join probe +
TPC-H SF100 q1/q3/q5 wall-time"): TPC-H SF100 q3 -- FilterExec x3 -> HashJoinExec(customer |x| orders) ->
HashJoinExec(.. |x| lineitem) -> AggregateExec -> ProjectionExec -> SortExec (reference benchmarks/queries/q3.sql,
benchmarks/src/bin/tpch.rs:286-351) over synthetic TPC-H-shaped tables resident in HBM (lineitem 600,037,902 rows, orders
150,009,476, customer 15,000,000; SURVEY.md section 8d generator).  A "step" = one full q3 through the native plan executor.
value = lineitem rows through the query per second.
roofline: the hash-join probe of lineitem (the operator with the most kernel time); algorithmic bytes = SURVEY.md section 8d's
24 B per probe row + 12 B per emitted pair x the counts measured on this input; duration = HIP events around the probe
operator's kernels on the stream they run on.
cpu_baseline: the C oracle's q3 ("port", OpenMP over all host cores) on an SF10 sample of the same generator.
"extra": the join-probe micro-grid (2^28 probes x {2^20, 2^24, 2^27} build keys) and SF100 q1 / q5 wall times.

Workload at N>1 (until the exchange moves under the C ABI): the round-1 q1 weak-scaling leg, below.
---- q1 leg (BASELINE.json configs[1]): TPC-H SF10 q1 -- FilterExec -> ProjectionExec ->
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


def main_q1(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: SF10 lineitem = 59,986,052)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=16_000_000)
    ap.add_argument("--extras", action="store_true", help="also time the join-probe / sort / partition micro-workloads")
    args, _unknown = ap.parse_known_args(argv)

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


Q3_SLOT_BYTES = 16              # one 16-byte slot touch per probe (key + row id), SURVEY.md section 8d
Q3_PROBE_KEY_BYTES = 8
Q3_PAIR_BYTES = 12              # u64 build idx + u32 probe idx per emitted pair in section 8d's formula


def main_q3():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sf", type=float, default=100.0, help="scale factor (default 100 = BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-sf", type=float, default=10.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the probe micro-grid and the SF100 q1 / q5 wall times")
    args = ap.parse_args()

    import torch
    import tpch_util as T
    import arrow_ballista_amd as g

    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"
    torch.cuda.set_device(0)
    tc = g.TaskContext(device=0)
    sf = args.sf
    n_li = T.LINEITEM_ROWS.get(int(sf), int(6_000_000 * sf)) if sf == int(sf) else int(6_000_000 * sf)
    n_orders, n_cust, n_supp = (n_li + 3) // 4, int(150_000 * sf), int(10_000 * sf)
    li = T.gen_lineitem_device(tc, n_li, n_supp=n_supp, columns=("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate"))
    od = T.gen_orders_device(tc, n_orders, n_cust)
    cu = T.gen_customer_device(tc, n_cust)
    plan = g.NativePlan(T.q3_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li])), tc)

    t0 = time.perf_counter()
    res = plan.execute(0)
    tc.sync()
    first_ms = (time.perf_counter() - t0) * 1e3          # cold: includes the hiprtc specialisation of every pipeline
    for _ in range(max(args.warmup, 2)):
        res = plan.execute(0)
    tc.ctx.jit_wait()
    m0 = plan.metrics()
    res = plan.execute(0)
    tc.sync()
    m1 = plan.metrics()
    # pairs emitted by the lineitem probe in one run = the larger of the two joins' output rows
    matches = max([int(b["output_rows"]) - int(a["output_rows"]) for a, b in zip(m0, m1) if b["node"] == "HashJoinExec"] or [0])
    plan.profile(True)

    import gc
    gc.collect(); gc.freeze(); gc.disable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = plan.execute(0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    ops = plan.profile_all()
    plan.profile(False)
    groups = res.num_rows

    # measured counts of the lineitem probe on this input (outside the timed region): rows that pass the fused filter = probes
    shipdate = li.columns[[c.name for c in li.columns].index("l_shipdate")].data[: 4 * n_li].view(torch.int32)
    probes = int((shipdate > T.Q3_DATE).sum().item())

    line = {
        "metric": "tpch_sf100_q3_lineitem_rows_per_sec", "value": n_li * args.steps / dt, "unit": "rows/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "i128", "data": "synthetic",
        "config": {"workload": "TPC-H SF%g q3 (BASELINE configs[2]): filter x3 + hash join customer|x|orders + hash join |x| lineitem + aggregate + sort, one task on 1 x MI355X" % sf,
                   "lineitem_rows": n_li, "orders_rows": n_orders, "customer_rows": n_cust, "result_groups": groups,
                   "input": "Arrow-physical columns resident in HBM", "first_run_ms_cold_jit": first_ms, "parallelism": "single partition"},
    }
    probe_ops = sorted((o for o in ops if o["op"] == "join_probe"), key=lambda o: -o["kernel_ms"])
    if probe_ops and probe_ops[0]["launches"] > 0:
        po = probe_ops[0]
        avg_ms = po["kernel_ms"] / po["launches"]
        alg = (Q3_PROBE_KEY_BYTES + Q3_SLOT_BYTES) * probes + Q3_PAIR_BYTES * matches
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "r02_traffic.json")
        if sf == 100 and os.path.exists(tp):
            traffic = json.load(open(tp)).get("traffic_bytes_per_launch")
        line["roofline"] = {"bound": "hbm", "kernel": "HashJoinExec probe of lineitem (fused filter l_shipdate > date + key lookup + pair emit)",
                            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                            "avg_launch_ms": avg_ms, "launches": po["launches"], "algorithmic_bytes_per_launch": alg,
                            "probe_rows": probes, "matches": matches, "rows_scanned": n_li,
                            "probe_rows_per_s": probes / (avg_ms * 1e-3), "rows_scanned_per_s": n_li / (avg_ms * 1e-3),
                            "achieved_incl_fused_filter_column": (alg + 4 * n_li + Q3_PROBE_KEY_BYTES * (n_li - probes)) / (avg_ms * 1e-3) / 1e9}
    line["operators"] = [{"op": o["op"], "kernel_ms_per_step": o["kernel_ms"] / max(1, args.steps), "launches": o["launches"]} for o in sorted(ops, key=lambda o: -o["kernel_ms"])[:8]]
    del plan, res
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_q3(T, args.cpu_sample_sf)
    if not args.no_extras:
        import bench_extras
        del li, od, cu
        torch.cuda.empty_cache()
        extra = {"join_probe": [bench_extras.join_probe_micro(tc, g, b, 28, 1.0) for b in (20, 24, 27)]}
        torch.cuda.empty_cache()
        extra["sf100_q1"] = bench_extras.q1_pipeline(tc, T, g, 100)
        tp = bench_extras.tpch_pipelines(tc, T, g, 100)
        extra["sf100_q3"], extra["sf100_q5"] = tp["q3"], tp["q5"]
        line["extra"] = extra
    print(json.dumps(line))


def cpu_baseline_q3(T, sample_sf):
    """Oracle (C restatement, OpenMP) q3 on the host cores over an SF`sample_sf` sample of the same generator."""
    n_li = T.LINEITEM_ROWS.get(int(sample_sf), int(6_000_000 * sample_sf))
    h = T.gen_q3_tables_host(n_li, int(150_000 * sample_sf))
    best, st = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        _rows, st = T.q3_oracle_c(h, cap=0)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return {"value": n_li / best, "unit": "rows/s", "cores": T.oracle_lib().oracle_num_threads(), "kind": "port", "wall_ms": best * 1e3,
            "sample": "C oracle q3 (3 filters, 2 chained-hash-table joins, per-thread hash aggregate, sort) over synthetic SF%g tables (%d lineitem rows, %d groups), best of 3, data in host memory"
                      % (sample_sf, n_li, st["groups"])}


if __name__ == "__main__":
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        main_q1()
    else:
        main_q3()
