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

Workload at N>1 (strong scaling: the job stays SF100, every rank holds 1/N of each table): ONE native plan per rank with the
exchanges inside (csrc/plan_exec.cpp RepartitionExec / BroadcastExec over csrc/exchange.cpp).  The headline plan is the one a
cost-based planner picks for q3 -- the build side of orders |x| lineitem is 20 x smaller than the probe side, so it is
BROADCAST and lineitem stays where it is: customer keys broadcast, orders joined locally, the joined orders (14.6 M rows at
SF100) all-gathered (one grouped ncclSend / ncclRecv round per column buffer), CollectLeft join + partial aggregate per rank,
partial states hash-repartitioned on the group key, final aggregate, sorted runs gathered and merged.  value = lineitem rows of
the whole job per second, time = MAX over ranks between barriers.  "extra": the BASELINE configs[3] shape ("hash-partitioned RCCL
all-to-all": BOTH sides of orders |x| lineitem repartitioned on the order key -- it moves 13 GB of lineitem columns where the
broadcast moves 0.2 GB, and is reported for that reason) on q3 and on q5, and q1 (partial states gathered).
GPUQ_BENCH_BACKEND=gloo rehearses the N>1 path with ranks sharing a GPU (host-staged transport).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling


Q3_SLOT_BYTES = 16              # one 16-byte slot touch per probe (key + row id), SURVEY.md section 8d
Q3_PROBE_KEY_BYTES = 8
Q3_PAIR_BYTES = 12              # u64 build idx + u32 probe idx per emitted pair in section 8d's formula


def main_q3():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sf", type=float, default=100.0, help="scale factor of the WHOLE job (default 100 = BASELINE configs[2]); N ranks hold 1/N of every table each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-sf", type=float, default=10.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (N=1: probe micro-grid, SF100 q1 / q5; N>1: q3 with the build side broadcast, distributed q5, q1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import tpch_util as T
    import arrow_ballista_amd as g
    from arrow_ballista_amd import parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GPUQ_BENCH_BACKEND=gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices, the exchange
    # runs over the host-staged transport); the measured configuration is always nccl = RCCL, one rank per GPU.
    backend = os.environ.get("GPUQ_BENCH_BACKEND", "nccl")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    tc = g.TaskContext(device=local_rank)
    comm = parallel.Comm(tc) if world > 1 else None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(fn, steps):
        """K steps between fences, MAX over ranks."""
        import gc
        gc.collect(); gc.freeze(); gc.disable()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        fence()
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=tc.device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    sf = args.sf
    n_total = T.LINEITEM_ROWS.get(int(sf), int(6_000_000 * sf)) if sf == int(sf) else int(6_000_000 * sf)
    # strong scaling: the job is SF`sf`; rank r holds rows [r * n_li, (r + 1) * n_li) of lineitem and the matching ranges of orders / customer
    n_li = n_total if world == 1 else n_total // world // 4 * 4
    n_orders = (n_li + 3) // 4
    n_cust_total, n_supp = int(150_000 * sf), int(10_000 * sf)
    n_cust = n_cust_total if world == 1 else n_cust_total // world // 5 * 5
    li = T.gen_lineitem_device(tc, n_li, n_supp=n_supp, columns=("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate"), row0=rank * n_li)
    od = T.gen_orders_device(tc, n_orders, n_cust_total, row0=rank * n_orders)
    cu = T.gen_customer_device(tc, n_cust, row0=rank * n_cust)
    if world == 1:
        plan = g.NativePlan(T.q3_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li])), tc)
    else:
        plan = g.NativePlan(T.q3_dist_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), world, "broadcast"), tc)
        plan.set_comm(comm)

    fence()
    t0 = time.perf_counter()
    res = plan.execute(0)
    tc.sync()
    first_ms = (time.perf_counter() - t0) * 1e3          # cold: hiprtc specialisation of every pipeline (or its load from the on-disk cache)
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
    dt, res = timed_steps(lambda: plan.execute(0), args.steps)
    ops = plan.profile_all()
    plan.profile(False)
    groups = res.num_rows

    # measured counts of the lineitem probe on this input (outside the timed region): rows that pass the fused filter = probes
    shipdate = li.columns[[c.name for c in li.columns].index("l_shipdate")].data[: 4 * n_li].view(torch.int32)
    probes = int((shipdate > T.Q3_DATE).sum().item())

    rows_job = n_li * world
    line = {
        "metric": "tpch_sf100_q3_lineitem_rows_per_sec", "value": rows_job * args.steps / dt, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "i128", "data": "synthetic",
        "config": {"workload": ("TPC-H SF%g q3 (BASELINE configs[2]): filter x3 + hash join customer|x|orders + hash join |x| lineitem + aggregate + sort, " % sf)
                               + ("one task on 1 x MI355X" if world == 1 else
                                  "%d ranks x 1/%d of every table: customer keys and the joined orders broadcast over RCCL (grouped send/recv per column buffer), lineitem stays in place, "
                                  "join + partial aggregate per rank, partial states repartitioned on the group key, final aggregate, sorted runs gathered and merged" % (world, world)),
                   "lineitem_rows": rows_job, "lineitem_rows_per_gpu": n_li, "orders_rows_per_gpu": n_orders, "customer_rows_per_gpu": n_cust, "result_groups": groups,
                   "input": "Arrow-physical columns resident in HBM", "first_run_ms_cold_jit": first_ms,
                   "parallelism": "single partition" if world == 1 else "partition-per-gpu x%d, exchange inside the native plan (%s)" % (world, comm.transport)},
    }
    probe_ops = sorted((o for o in ops if o["op"] == "join_probe"), key=lambda o: -o["kernel_ms"])
    if probe_ops and probe_ops[0]["launches"] > 0:
        po = probe_ops[0]
        avg_ms = po["kernel_ms"] / po["launches"]
        alg = (Q3_PROBE_KEY_BYTES + Q3_SLOT_BYTES) * probes + Q3_PAIR_BYTES * matches
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "r02_traffic.json")
        if sf == 100 and world == 1 and os.path.exists(tp):
            traffic = json.load(open(tp)).get("traffic_bytes_per_launch")
        line["roofline"] = {"bound": "hbm", "kernel": "HashJoinExec probe of lineitem (fused filter l_shipdate > date + key lookup + pair emit; rank 0's launch)",
                            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                            "avg_launch_ms": avg_ms, "launches": po["launches"], "algorithmic_bytes_per_launch": alg,
                            "probe_rows": probes, "matches": matches, "rows_scanned": n_li,
                            "probe_rows_per_s": probes / (avg_ms * 1e-3), "rows_scanned_per_s": n_li / (avg_ms * 1e-3),
                            "achieved_incl_fused_filter_column": (alg + 4 * n_li + Q3_PROBE_KEY_BYTES * (n_li - probes)) / (avg_ms * 1e-3) / 1e9}
    line["operators"] = [{"op": o["op"], "kernel_ms_per_step": o["kernel_ms"] / max(1, args.steps), "launches": o["launches"]} for o in sorted(ops, key=lambda o: -o["kernel_ms"])[:8]]
    del plan, res
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_q3(T, args.cpu_sample_sf)
    if not args.no_extras:
        import bench_extras
        if world == 1:
            del li, od, cu
            torch.cuda.empty_cache()
            extra = {"join_probe": [bench_extras.join_probe_micro(tc, g, b, 28, 1.0) for b in (20, 24, 27)]}
            torch.cuda.empty_cache()
            extra["sf100_q1"] = bench_extras.q1_pipeline(tc, T, g, 100)
            tp = bench_extras.tpch_pipelines(tc, T, g, 100)
            extra["sf100_q3"], extra["sf100_q5"] = tp["q3"], tp["q5"]
        else:
            # second legs, every rank takes part: q3 with BOTH sides of orders |x| lineitem repartitioned (the BASELINE configs[3]
            # shape: 60 x the bytes of the broadcast plan above), distributed q5, and q1 (partial states gathered)
            extra = {}
            k = max(3, args.steps // 4)
            p2 = g.NativePlan(T.q3_dist_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), world, "partitioned"), tc)
            p2.set_comm(comm)
            for _ in range(3):
                p2.execute(0)
            d2, r2 = timed_steps(lambda: p2.execute(0), k)
            extra["q3_both_sides_repartitioned"] = {"ms_per_step": d2 / k * 1e3, "lineitem_rows_per_s": rows_job * k / d2, "result_groups": r2.num_rows}
            # BASELINE configs[3]: q5, the 6-way join with orders |x| lineitem hash-partitioned across the ranks (T.q5_dist_plan)
            sper = n_supp // world
            su = T.gen_supplier_device(tc, sper if rank < world - 1 else n_supp - sper * (world - 1), row0=rank * sper)
            nation, region = T.nation_region_arrow()
            p5 = g.NativePlan(T.q5_dist_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), g.MemoryExec([su]), g.MemoryExec([nation]), g.MemoryExec([region]), world), tc)
            p5.set_comm(comm)
            for _ in range(3):
                p5.execute(0)
            tc.ctx.jit_wait()
            d5, r5 = timed_steps(lambda: p5.execute(0), k)
            extra["q5_partitioned_join"] = {"ms_per_step": d5 / k * 1e3, "lineitem_rows_per_s": rows_job * k / d5, "result_groups": r5.num_rows}
            del p2, r2, p5, r5, su, li, od, cu
            torch.cuda.empty_cache()
            l1 = T.gen_lineitem_device(tc, n_li, row0=rank * n_li)
            p3 = g.NativePlan(T.q1_dist_plan(l1), tc)
            p3.set_comm(comm)
            for _ in range(3):
                p3.execute(0)
            tc.ctx.jit_wait()
            d3, r3 = timed_steps(lambda: p3.execute(0), k)
            extra["q1_partial_states_gathered"] = {"ms_per_step": d3 / k * 1e3, "lineitem_rows_per_s": rows_job * k / d3, "result_groups": r3.num_rows}
        line["extra"] = extra
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


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
    main_q3()
