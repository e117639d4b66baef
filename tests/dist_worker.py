"""One rank of the 2-process GPU test (tests/test_gpu_distributed.py).  Both ranks share cuda:0; the collectives run over
gloo (parallel.py stages device buffers through host memory for that backend), every operator runs in libgpuq on the GPU.
Usage: python dist_worker.py RANK WORLD PORT OUT_JSON"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import pyarrow as pa
    import numpy as np
    import torch
    import torch.distributed as dist
    import arrow_ballista_amd as g
    from arrow_ballista_amd import parallel
    from arrow_ballista_amd.expr import col
    import tpch_util as T
    from test_gpu_operators import rand_table
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    try:
        tc = g.TaskContext(device=0)
        rows = lambda t: [list(r) for r in T.table_to_rows(tc, g.plan.materialize(tc, t))]
        # shards: left 3000+rank*100 rows, right 5000 rows, 20 % nulls
        lt = rand_table(1000 + rank, 3000 + 100 * rank, 0.2)
        rt = rand_table(2000 + rank, 5000, 0.2)
        rt = rt.rename_columns(["r_" + c for c in rt.schema.names])
        L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
        ls, rs = L.schema(), R.schema()
        ltab, rtab = L.execute(0, tc), R.execute(0, tc)
        # 1. repartition + exchange (nullable columns, Utf8 keys packed)
        mine = parallel.repartition_exchange(tc, ltab, [col("k64", ls), col("flag", ls)])
        res["exchange_rows"] = rows(mine)
        # 2. partitioned hash join on a nullable int key
        for jt in ("Inner", "Left"):
            jv = parallel.partitioned_hash_join(tc, ltab, rtab, [(col("k64", ls), col("r_k64", rs))], jt)
            res["join_" + jt] = rows(jv)
        # 3. distributed sort by (flag asc nulls last, dec desc nulls first)
        order = [{"expr": col("flag", ls), "asc": True, "nulls_first": False}, {"expr": col("dec", ls), "asc": False, "nulls_first": True}]
        sv = parallel.distributed_sort(tc, ltab, order, samples_per_rank=64)
        res["sort_rows"] = rows(sv)
        # 4. broadcast (CollectLeft build side)
        small = g.plan.slice_table(tc, ltab, 0, 10 + rank)
        res["bcast_rows"] = rows(parallel.broadcast_table(tc, small))
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
