"""The exchange entry points of the C ABI (include/gpuq.h "exchange", csrc/exchange.cpp) on one GPU: a world of ONE rank over
RCCL (every send is a send to self: the grouped ncclSend / ncclRecv path, the counts exchange, bitmap cutting / re-joining
and Utf8 offset rebasing all run), and the host-staged transport with the same inputs.  Several ranks: test_gpu_distributed.py
(2 processes, host transport) and bench.py --gpus N (RCCL, one rank per GPU)."""
import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd import parallel
from arrow_ballista_amd.expr import col
from oracle import oracle_np as O
import test_gpu_operators as M
import tpch_util as T

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["rccl", "host"])
def comm(tc, request, monkeypatch):
    monkeypatch.setenv("GPUQ_COMM_TRANSPORT", request.param)
    try:
        c = parallel.Comm(tc)
    except g.GpuqError as e:
        if request.param == "rccl" and "not available" in str(e):
            pytest.skip("RCCL not available")
        raise
    assert c.transport == request.param and (c.rank, c.world) == (0, 1)
    yield c
    c.close()


@pytest.mark.parametrize("nulls", [0.0, 0.2])
def test_self_exchange_and_allgather_round_trip_every_column_kind(tc, comm, nulls):
    """Int64 / Int32 / Date32 / Decimal128 / Float64 / Boolean / Utf8 (Arrow layout, strings longer than 15 bytes included)
    with and without NULLs: what a rank sends to itself comes back row for row, regrouped by partition (the oracle's
    partition function), and an all-gather of one rank is the table itself."""
    n = 5000
    t = M.rand_table(77, n, nulls)
    long_s = pa.array(["row-%d-%s" % (i, "x" * (i % 40)) if (nulls == 0 or i % 7) else None for i in range(n)], type=pa.string())
    t = t.append_column(pa.field("long_s", pa.string(), nullable=nulls > 0), long_s)
    src = g.MemoryExec([t])
    s = src.schema()
    tab = src.execute(0, tc)
    out = parallel.repartition_exchange(tc, tab, [col("k64", s)], comm=comm)      # world 1: everything hashes to rank 0
    got = M.dev_rows(tc, out)
    exp = M.ora_rows(O.Table.from_arrow(t))
    M.close_rows(got, exp)                                                           # one partition keeps input order
    bc = parallel.broadcast_table(tc, g.plan.slice_table(tc, tab, 100, 1234), comm=comm)
    M.close_rows(M.dev_rows(tc, bc), exp[100:1334])
    empty = parallel.broadcast_table(tc, g.plan.slice_table(tc, tab, 0, 0), comm=comm)
    assert empty.num_rows == 0


def test_distributed_q3_plan_on_one_rank(tc, comm):
    """The distributed q3 plans (both sides of the big join exchanged / joined orders broadcast) through the native executor
    with the ranks attached: on a world of one they must give the single-GPU answer -- the oracle's."""
    n_li, n_cust = 120_000, 1500
    cols = ("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate")
    li = T.gen_lineitem_device(tc, n_li, n_supp=100, columns=cols)
    od = T.gen_orders_device(tc, (n_li + 3) // 4, n_cust)
    cu = T.gen_customer_device(tc, n_cust)
    exp, _st = T.q3_oracle_c(T.gen_q3_tables_host(n_li, n_cust))
    from test_gpu_native_plan import arrow_rows
    for mode in ("partitioned", "broadcast"):
        plan = g.NativePlan(T.q3_dist_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), 1, mode), tc)
        with pytest.raises(g.GpuqError, match="ranks of the node"):
            plan.execute(0)
        plan.set_comm(comm)
        for _ in range(2):
            rows = [tuple(r) for r in arrow_rows(plan.execute(0).to_arrow())]
            assert [(r[1], r[2]) for r in rows] == [(r[1], r[2]) for r in exp]
            assert sorted(rows) == sorted(exp)


def test_exchange_moves_the_narrow_and_temporal_types(tc, comm):
    """Int8 / Int16 / UInt8 / UInt16 / Float32 / Timestamp / Date64 columns (1-, 2-, 4- and 8-byte values, with NULLs) through the native
    RepartitionExec / BroadcastExec over RCCL on a world of one, and through a hash-partitioned join on an Int16 key: the exchange moves
    bytes by width, a column narrower than four bytes must come back row for row like any other."""
    import test_gpu_types2 as T2
    t = T2.table(seed=5, n=7001, nulls=0.15)
    src = g.MemoryExec([t])
    s = src.schema()
    plan = g.NativePlan(g.RepartitionExchangeExec(src, [col("i16", s), col("u8", s)], 1), tc)
    plan.set_comm(comm)
    out = plan.execute(0).to_arrow()
    for name in t.column_names:
        T2.same_column(out[name], t[name], name)
    bc = g.NativePlan(g.BroadcastExec(src), tc)
    bc.set_comm(comm)
    out = bc.execute(0).to_arrow()
    for name in t.column_names:
        T2.same_column(out[name], t[name], name)
