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


def test_q1_rank_records_merge_as_in_bench(tc, mirror_layer):
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


def test_q1_on_the_reference_testdata_two_partitions(tc):
    """The rows of ballista/scheduler/testdata/lineitem/partition{0,1}.tbl (dbgen output the reference's planner tests read,
    planner.rs:376-392 -- the reference asserts the plan SHAPE over them, not values): q1 as that two-stage plan (Partial per
    partition, FinalPartitioned over both) on the device, against plain `decimal` arithmetic on the .tbl text."""
    import datetime
    import decimal
    import os
    import pyarrow as pa
    import arrow_ballista_amd as g
    D = decimal.Decimal
    gold = os.path.join(os.path.dirname(__file__), "golden", "tpch10")
    parts, rows = [], []
    for fn in ("lineitem.partition0.tbl", "lineitem.partition1.tbl"):
        rs = [l.rstrip("\n").split("|") for l in open(os.path.join(gold, fn))]
        rows += rs
        day = lambda s: (datetime.date.fromisoformat(s) - datetime.date(1970, 1, 1)).days
        parts.append(pa.table({
            "l_quantity": pa.array([D(r[4]).quantize(D("0.01")) for r in rs], pa.decimal128(15, 2)),
            "l_extendedprice": pa.array([D(r[5]) for r in rs], pa.decimal128(15, 2)),
            "l_discount": pa.array([D(r[6]) for r in rs], pa.decimal128(15, 2)),
            "l_tax": pa.array([D(r[7]) for r in rs], pa.decimal128(15, 2)),
            "l_returnflag": pa.array([r[8] for r in rs]), "l_linestatus": pa.array([r[9] for r in rs]),
            "l_shipdate": pa.array([day(r[10]) for r in rs], pa.int32()).cast(pa.date32())},
            schema=pa.schema([pa.field(n, t, nullable=False) for n, t in [("l_quantity", pa.decimal128(15, 2)), ("l_extendedprice", pa.decimal128(15, 2)),
                              ("l_discount", pa.decimal128(15, 2)), ("l_tax", pa.decimal128(15, 2)), ("l_returnflag", pa.string()), ("l_linestatus", pa.string()),
                              ("l_shipdate", pa.date32())]])))
    src = g.MemoryExec(parts)
    plan = T.q1_plan(src, two_phase=True)
    # the Final stage reads BOTH partial partitions: coalesce them below it, as the reference's stage boundary does
    node = plan
    while not (isinstance(node, g.AggregateExec) and node.mode == "FinalPartitioned"):
        node = node.children()[0]
    node.input = g.CoalescePartitionsExec(node.input)
    got = T.q1_result_to_rows(tc, g.plan.materialize(tc, plan.execute(0, tc)))
    native = [tuple(r.values()) for r in g.NativePlan(plan, tc).execute(0).to_arrow().to_pylist()]
    exp = {}
    cutoff = datetime.date(1998, 9, 2)
    for r in rows:
        if datetime.date.fromisoformat(r[10]) > cutoff:
            continue
        q, e, d, t = D(r[4]), D(r[5]), D(r[6]), D(r[7])
        a = exp.setdefault((r[8], r[9]), [D(0), D(0), D(0), D(0), D(0), 0])
        a[0] += q; a[1] += e; a[2] += e * (1 - d); a[3] += e * (1 - d) * (1 + t); a[4] += d; a[5] += 1
    want = []
    for k in sorted(exp):
        q, e, dp, ch, ds, c = exp[k]
        trunc = lambda x, s: int((x.scaleb(s)).to_integral_value(rounding=decimal.ROUND_DOWN))
        want.append(k + (trunc(q, 2), trunc(e, 2), trunc(dp, 4), trunc(ch, 6), trunc(q / c, 6), trunc(e / c, 6), trunc(ds / c, 6), c))
    assert got == want and len(want) >= 2
    assert [tuple(int(x.scaleb(-x.as_tuple().exponent)) if isinstance(x, D) else x for x in r) for r in native] == want
