"""TPC-H q1 on the device (FilterExec -> ProjectionExec -> AggregateExec Partial/Final -> SortExec),
bit-exact against the C oracle on the same seeded synthetic lineitem rows."""
import pytest

import tpch_util as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 200_000, 3_000_001])
@pytest.mark.parametrize("two_phase", [True, False])
def test_q1_matches_oracle(tc, n, two_phase):
    li = T.gen_lineitem_device(tc, n, seed=7)
    got = T.q1_result_to_rows(tc, T.run_q1(tc, li, two_phase=two_phase))
    assert got == T.q1_oracle_rows(n, seed=7)


def test_q1_hash_strategy_matches_tiny(tc):
    n = 500_000
    li = T.gen_lineitem_device(tc, n, seed=11)
    a = T.q1_result_to_rows(tc, T.run_q1(tc, li, strategy="tiny"))
    b = T.q1_result_to_rows(tc, T.run_q1(tc, li, strategy="hash"))
    assert a == b == T.q1_oracle_rows(n, seed=11)


def test_generator_matches_oracle_restatement(tc):
    import numpy as np
    n = 100_003
    cols = ("l_orderkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate")
    dev = T.gen_lineitem_device(tc, n, seed=5, row0=12345 * 4, columns=cols)
    host = T.gen_lineitem_host(n, seed=5, row0=12345 * 4)
    for c in dev.columns:
        d = c.data.cpu().numpy()
        h = host[c.name]
        nb = n * {'Int64': 8, 'Date32': 4, 'Utf8': 1}.get(c.type if isinstance(c.type, str) else '', 16)
        assert d[:nb].tobytes() == h.view(np.uint8)[:nb].tobytes(), c.name
        if c.offsets is not None:
            assert (c.offsets.cpu().numpy()[: n + 1] == host[c.name + "_off"]).all()


def test_q1_rank_records_merge_as_in_bench(tc):
    """bench.py's N>1 step replayed on one GPU: the partial-aggregate results of two shards are fixed-layout records; laid
    back to back (what RCCL's all-gather delivers) they are read in place by the final aggregate through a view."""
    import torch
    import arrow_ballista_amd as g
    from arrow_ballista_amd import parallel
    n = 150_000
    recs, counts, cols0 = [], [], None
    for r in range(2):
        li = T.gen_lineitem_device(tc, n, seed=T.SEED_LINEITEM, seed_orders=T.SEED_ORDERS, row0=r * n)
        partial, full, final_src = T.q1_split_plan(li, 64)
        st = partial.execute(0, tc)
        buf, cap = st._record
        assert cap == 64 and not st.is_view()
        buf[:8] = torch.tensor([st.num_rows], dtype=torch.int64).view(torch.uint8).to(buf.device)    # header word written by allgather_table
        recs.append(buf); counts.append(st.num_rows); cols0 = st.columns
    recv = torch.cat(recs)
    assert recv.view(2, -1)[:, :8].contiguous().view(torch.int64).flatten().tolist() == counts
    merged = parallel.unpack_records(cols0, recv, counts, 64)
    assert merged.is_view() and merged.num_rows == sum(counts)
    final_src.partitions[0] = merged
    got = T.q1_result_to_rows(tc, g.plan.materialize(tc, full.execute(0, tc)))
    assert got == T.q1_oracle_rows(2 * n)
