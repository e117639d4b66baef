"""One rank of the 2-process GPU test of the NATIVE exchange (tests/test_gpu_distributed.py): both ranks share cuda:0, so the
exchange entry points of the C ABI run over the host-staged transport (torch.distributed gloo moves the staged bytes; RCCL
refuses two ranks on one device), everything else -- partitioning, packing, bitmap / offset handling, the distributed q3
plans in the native executor -- is the code the RCCL configuration runs.  Usage: python dist_worker_native.py RANK WORLD PORT OUT"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import arrow_ballista_amd as g
    from arrow_ballista_amd import parallel
    from arrow_ballista_amd.expr import col
    import tpch_util as T
    from test_gpu_operators import rand_table
    from test_gpu_native_plan import arrow_rows
    import pyarrow as pa
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    try:
        tc = g.TaskContext(device=0)
        comm = parallel.Comm(tc)
        assert comm.transport == "host" and comm.world == world
        rows = lambda t: [list(r) for r in T.table_to_rows(tc, g.plan.materialize(tc, t))]
        lt = rand_table(1000 + rank, 3000 + 100 * rank, 0.2)
        lt = lt.append_column(pa.field("long_s", pa.string()), pa.array([None if i % 11 == 0 else "rank%d-row%d-%s" % (rank, i, "y" * (i % 33)) for i in range(lt.num_rows)]))
        rt = rand_table(2000 + rank, 5000, 0.2)
        rt = rt.rename_columns(["r_" + c for c in rt.schema.names])
        L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
        ls, rs = L.schema(), R.schema()
        ltab, rtab = L.execute(0, tc), R.execute(0, tc)
        res["exchange_rows"] = rows(parallel.repartition_exchange(tc, ltab, [col("k64", ls), col("flag", ls)], comm=comm))
        for jt in ("Inner", "Left"):
            res["join_" + jt] = rows(parallel.partitioned_hash_join(tc, ltab, rtab, [(col("k64", ls), col("r_k64", rs))], jt, comm=comm))
        res["bcast_rows"] = rows(parallel.broadcast_table(tc, g.plan.slice_table(tc, ltab, 0, 10 + rank), comm=comm))
        # distributed SortExec through the native RangeRepartitionExec (config #5's shape): sample -> splitters -> one range exchange ->
        # ordered fan-in; twice more deferred.  Sorted by (flag asc nulls last, dec desc nulls first), and by one Decimal128 key.
        st = rand_table(1000 + rank, 3000 + 100 * rank, 0.2)
        S = g.MemoryExec([st]); ss = S.schema()
        for name, order in (("sort2", [{"expr": col("flag", ss), "asc": True, "nulls_first": False}, {"expr": col("dec", ss), "asc": False, "nulls_first": True}]),
                            ("sort1", [{"expr": col("dec", ss), "asc": True, "nulls_first": False}])):
            sp = g.NativePlan(g.RangeRepartitionExec(S, order, world, samples=64), tc)
            sp.set_comm(comm)
            first = [list(r) for r in arrow_rows(sp.execute(0).to_arrow())]
            res["native_" + name] = first
            again = [list(r) for r in arrow_rows(sp.execute(0).to_arrow())]
            res["native_%s_again" % name] = [[int(again == first)]]
        # distributed q3 through the native executor: shards of the three generated tables
        n_li, n_cust = 60_000, 1500            # per rank / in total
        cols = ("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate")
        li = T.gen_lineitem_device(tc, n_li, n_supp=100, columns=cols, row0=rank * n_li)
        od = T.gen_orders_device(tc, n_li // 4, n_cust, row0=rank * (n_li // 4))
        per = n_cust // world // 5 * 5
        cu = T.gen_customer_device(tc, per if rank < world - 1 else n_cust - per * (world - 1), row0=rank * per)
        for mode in ("partitioned", "broadcast"):
            plan = g.NativePlan(T.q3_dist_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), world, mode), tc)
            plan.set_comm(comm)
            res["q3_" + mode] = [list(r) for r in arrow_rows(plan.execute(0).to_arrow())]
            # twice more: deferred on every rank, closed by the agreement round (no rank redoes anything alone)
            for k in range(2):
                again = [list(r) for r in arrow_rows(plan.execute(0).to_arrow())]
                st = plan.exec_stats()
                res["q3_%s_again%d_same" % (mode, k)] = [[int(again == res["q3_" + mode]), int(st["deferred"]), st["retries"]]]
            if mode == "partitioned":
                # rank 1 alone gets a bigger orders shard under the plan: what it remembered no longer fits, so BOTH ranks redo the
                # execution synchronously (the status word of the exchanges' meta round) -- and neither hangs
                od2 = T.gen_orders_device(tc, n_li // 4 + (4000 if rank == 1 else 0), n_cust, row0=rank * (n_li // 4))
                plan.set_input(1, od2)
                r1 = [list(r) for r in arrow_rows(plan.execute(0).to_arrow())]
                st = plan.exec_stats()
                res["q3_changed_input"] = r1
                res["q3_changed_stats"] = [[int(st["deferred"]), st["retries"]]]
                plan.set_input(1, od)
        # one rank fails below an exchange (rank 0 takes a substring that reaches beyond byte 15 of a long string, which the device refuses
        # at run time -- an ordering comparison of long strings no longer fails: the executor lowers it): the other rank must come back with
        # an error naming it, not wait for data that never comes
        from arrow_ballista_amd.expr import binary, lit, substr, Operator as Op
        pred = binary(substr(col("long_s", ls), 14, 5), Op.Eq, lit("zzz")) if rank == 0 else binary(col("k32", ls), Op.Gt, lit(0, "Int32"))
        try:
            p = g.NativePlan(g.BroadcastExec(g.FilterExec(pred, L)), tc)
            p.set_comm(comm)
            p.execute(0)
            res["peer_failure"] = [["no error", ""]]
        except Exception as e:          # noqa: BLE001
            res["peer_failure"] = [[type(e).__name__, str(e)[:300]]]
        # distributed q5 (BASELINE configs[3]): sharded customer / orders / lineitem / supplier, replicated nation / region
        n_supp = 100
        sper = n_supp // world
        su = T.gen_supplier_device(tc, sper if rank < world - 1 else n_supp - sper * (world - 1), row0=rank * sper)
        nation, region = T.nation_region_arrow()
        plan = g.NativePlan(T.q5_dist_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), g.MemoryExec([su]), g.MemoryExec([nation]), g.MemoryExec([region]), world), tc)
        plan.set_comm(comm)
        res["q5"] = [list(r) for r in arrow_rows(plan.execute(0).to_arrow())]
        comm.close()
    finally:
        dist.destroy_process_group()

    def enc(x):
        import decimal
        if isinstance(x, decimal.Decimal):
            return {"d": str(x)}
        if isinstance(x, float):
            return {"f": x.hex()}
        return x
    json.dump({k: [[enc(x) for x in r] for r in v] for k, v in res.items()}, open(out, "w"))


if __name__ == "__main__":
    main()
