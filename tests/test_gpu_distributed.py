"""GPU, 2 processes sharing cuda:0: the multi-GPU legs of the path end to end -- hash repartition +
exchange, partitioned hash join, range-partitioned distributed sort, broadcast -- with every operator running in libgpuq
and the collectives over gloo (the one-GPU box cannot host two RCCL ranks).  Checked against the oracle on the union of
the shards.  The RCCL transport itself is exercised by bench.py --gpus N on the driver's 8-GPU node."""
import json
import os
import subprocess
import sys
import tempfile

import pyarrow as pa
import pytest

from oracle import oracle_np as O
from test_gpu_operators import norm, rand_table
from arrow_ballista_amd.expr import col

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _dec(x):
    if isinstance(x, dict):
        return int(__import__("decimal").Decimal(x["d"]).scaleb(2)) if "d" in x else float.fromhex(x["f"])
    return x


def _run(world=2, worker="dist_worker.py", port_base=29700):
    port = str(port_base + os.getpid() % 1500)
    with tempfile.TemporaryDirectory() as d:
        outs = [os.path.join(d, "r%d.json" % r) for r in range(world)]
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, worker), str(r), str(world), port, outs[r]],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
        logs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=420)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
            logs.append(o.decode(errors="replace")[-3000:])
        assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
        return [{k: [tuple(_dec(x) for x in r) for r in v] for k, v in json.load(open(f)).items()} for f in outs]


@pytest.fixture(scope="module")
def ranks():
    return _run(2)


def _shards():
    lts = [rand_table(1000 + r, 3000 + 100 * r, 0.2) for r in range(2)]
    rts = [rand_table(2000 + r, 5000, 0.2) for r in range(2)]
    rts = [t.rename_columns(["r_" + c for c in t.schema.names]) for t in rts]
    return lts, rts


def _orows(t):
    return [tuple(r) for r in t.rows()]


def test_repartition_exchange_is_a_partition_of_the_union(ranks):
    lts, _ = _shards()
    allrows = _orows(O.Table.from_arrow(pa.concat_tables(lts)))
    got = [r for rk in ranks for r in rk["exchange_rows"]]
    assert norm(got) == norm(allrows)
    # both ranks applied the same function of (k64, flag): no key value appears on two ranks
    keys = [set((r[0], r[6]) for r in rk["exchange_rows"]) for rk in ranks]
    assert not (keys[0] & keys[1])
    assert all(len(k) > 0 for k in keys)


@pytest.mark.parametrize("jt", ["Inner", "Left"])
def test_partitioned_hash_join_equals_oracle_join_of_union(ranks, jt):
    lts, rts = _shards()
    ol, orr = O.Table.from_arrow(pa.concat_tables(lts)), O.Table.from_arrow(pa.concat_tables(rts))
    s = [{"name": n} for n in ol.names]
    pairs = O.hash_join(ol, orr, [({"column": {"name": "k64"}}, {"column": {"name": "r_k64"}})], jt)
    lrows, rrows = _orows(ol), _orows(orr)
    exp = [(lrows[i] if i is not None else (None,) * len(ol.names)) + (rrows[j] if j is not None else (None,) * len(orr.names)) for i, j in pairs]
    got = [r for rk in ranks for r in rk["join_" + jt]]
    assert len(got) == len(exp)
    assert norm(got) == norm(exp)


def test_distributed_sort_is_globally_ordered(ranks):
    lts, _ = _shards()
    ot = O.Table.from_arrow(pa.concat_tables(lts))
    order = [{"expr": {"column": {"name": "flag"}}, "asc": True, "nulls_first": False}, {"expr": {"column": {"name": "dec"}}, "asc": False, "nulls_first": True}]
    got = ranks[0]["sort_rows"] + ranks[1]["sort_rows"]            # rank order = global order
    assert norm(got) == norm(_orows(ot))                            # a permutation of the union
    gt = O.Table(ot.names, ot.types, [[r[c] for r in got] for c in range(len(ot.names))])
    keys = O.sort_keys(gt, order)
    assert all(keys[i] <= keys[i + 1] for i in range(len(keys) - 1))
    assert len(ranks[0]["sort_rows"]) > 0 and len(ranks[1]["sort_rows"]) > 0     # both ranges are populated


def test_broadcast_gives_every_rank_all_rows(ranks):
    assert ranks[0]["bcast_rows"] == ranks[1]["bcast_rows"]
    assert len(ranks[0]["bcast_rows"]) == 10 + 11


# ---------------------------------------------------------------- the same legs through the C ABI's exchange (csrc/exchange.cpp)
@pytest.fixture(scope="module")
def native_ranks():
    return _run(2, worker="dist_worker_native.py", port_base=31300)


def _native_left_shards():
    out = []
    for r in range(2):
        t = rand_table(1000 + r, 3000 + 100 * r, 0.2)
        out.append(t.append_column(pa.field("long_s", pa.string()),
                                   pa.array([None if i % 11 == 0 else "rank%d-row%d-%s" % (r, i, "y" * (i % 33)) for i in range(t.num_rows)])))
    return out


def test_native_exchange_is_a_partition_of_the_union(native_ranks):
    """gpuq_exchange_partitions over the host-staged transport, two ranks: nullable columns, Boolean bitmaps, Utf8 of any
    length in Arrow layout (offsets rebased per received piece).  The union of what the ranks hold afterwards is the union of
    what they held before, and no key lives on two ranks."""
    lts = _native_left_shards()
    allrows = _orows(O.Table.from_arrow(pa.concat_tables(lts)))
    got = [r for rk in native_ranks for r in rk["exchange_rows"]]
    assert norm(got) == norm(allrows)
    keys = [set((r[0], r[6]) for r in rk["exchange_rows"]) for rk in native_ranks]
    assert not (keys[0] & keys[1]) and all(len(k) > 0 for k in keys)


@pytest.mark.parametrize("jt", ["Inner", "Left"])
def test_native_partitioned_join_equals_oracle_join_of_union(native_ranks, jt):
    lts, (_, rts) = _native_left_shards(), _shards()
    ol, orr = O.Table.from_arrow(pa.concat_tables(lts)), O.Table.from_arrow(pa.concat_tables(rts))
    pairs = O.hash_join(ol, orr, [({"column": {"name": "k64"}}, {"column": {"name": "r_k64"}})], jt)
    lrows, rrows = _orows(ol), _orows(orr)
    exp = [(lrows[i] if i is not None else (None,) * len(ol.names)) + (rrows[j] if j is not None else (None,) * len(orr.names)) for i, j in pairs]
    got = [r for rk in native_ranks for r in rk["join_" + jt]]
    assert len(got) == len(exp) and norm(got) == norm(exp)


def test_native_broadcast_gives_every_rank_all_rows(native_ranks):
    assert native_ranks[0]["bcast_rows"] == native_ranks[1]["bcast_rows"] and len(native_ranks[0]["bcast_rows"]) == 10 + 11
    lts = _native_left_shards()
    exp = _orows(O.Table.from_arrow(pa.concat_tables([lts[0].slice(0, 10), lts[1].slice(0, 11)])))
    assert [tuple(r) for r in native_ranks[0]["bcast_rows"]] == exp            # rank order, row order


@pytest.mark.parametrize("mode", ["partitioned", "broadcast"])
def test_native_distributed_q3_equals_the_oracle_on_the_union(native_ranks, mode):
    """Distributed q3 (BASELINE configs[2] shape across ranks; T.q3_dist_plan) as ONE native plan per rank with the exchanges
    inside (RepartitionExec / BroadcastExec nodes): every rank ends with the full, globally ordered result, equal to the
    oracle's q3 over the union of the shards."""
    import tpch_util as T
    exp, _st = T.q3_oracle_c(T.gen_q3_tables_host(120_000, 1500))
    for rk in native_ranks:
        rows = [tuple(r) for r in rk["q3_" + mode]]
        assert [(r[1], r[2]) for r in rows] == [(r[1], r[2]) for r in exp]
        assert sorted(rows) == sorted(exp)


def test_native_distributed_q5_equals_the_oracle_on_the_union(native_ranks):
    """Distributed q5 (BASELINE configs[3]: 6-way join, the big join hash-partitioned across ranks; T.q5_dist_plan): broadcast of the
    filtered customers and of supplier, both sides of orders |x| lineitem exchanged, partial aggregate states gathered; every
    rank ends with the oracle's q5 over the union of the shards."""
    import tpch_util as T
    hl = T.lineitem_host_to_arrow(T.gen_lineitem_host(120_000, n_supp=100), 120_000)
    ho, hc, hs = T.gen_other_tables_host(30_000, 1500, 100)
    exp = [tuple(r) for r in T.q5_oracle(hc, ho, hl, hs)]
    assert len(exp) > 0
    for rk in native_ranks:
        got = [(r[0], int(r[1]["d"].replace(".", "")) if isinstance(r[1], dict) else r[1]) for r in rk["q5"]]
        assert got == exp


def test_native_distributed_plans_run_deferred_and_fail_together(native_ranks):
    """The same distributed plan executed again runs deferred on both ranks (identical rows, no retries); when ONE rank's input
    changes under the plan both redo the execution synchronously (the status word of the exchange's meta round) and agree on the new
    answer; when one rank fails below an exchange the other returns an error that names it.  None of this may hang: the worker
    processes are under a timeout."""
    for r in native_ranks:
        for mode in ("partitioned", "broadcast"):
            for k in range(2):
                same, deferred, retries = r["q3_%s_again%d_same" % (mode, k)][0]
                assert same == 1 and deferred == 1 and retries == 0, (mode, k, r["q3_%s_again%d_same" % (mode, k)])
        deferred, retries = r["q3_changed_stats"][0]
        assert deferred == 0 and retries == 1, r["q3_changed_stats"]
    assert native_ranks[0]["q3_changed_input"] == native_ranks[1]["q3_changed_input"] and len(native_ranks[0]["q3_changed_input"]) > 0
    e0, e1 = native_ranks[0]["peer_failure"][0], native_ranks[1]["peer_failure"][0]
    assert e0[0] == "GpuqError" and "15 bytes" in e0[1], e0
    assert e1[0] == "GpuqError" and "rank 0 failed" in e1[1], e1


@pytest.mark.parametrize("name,order", [
    ("sort2", [{"expr": {"column": {"name": "flag"}}, "asc": True, "nulls_first": False}, {"expr": {"column": {"name": "dec"}}, "asc": False, "nulls_first": True}]),
    ("sort1", [{"expr": {"column": {"name": "dec"}}, "asc": True, "nulls_first": False}])])
def test_native_range_partition_sort_is_the_sort_of_the_union(native_ranks, name, order):
    """RangeRepartitionExec in the native executor (csrc/plan_exec.cpp; planner.rs:120-136's single-partition merge stage as a range
    exchange): rank 0's rows followed by rank 1's are a permutation of the union in the oracle's key order, both ranges are populated,
    and a second (deferred) execution returns the same rows."""
    lts, _ = _shards()
    ot = O.Table.from_arrow(pa.concat_tables(lts))
    got = native_ranks[0]["native_" + name] + native_ranks[1]["native_" + name]
    assert norm(got) == norm(_orows(ot))
    gt = O.Table(ot.names, ot.types, [[r[c] for r in got] for c in range(len(ot.names))])
    keys = O.sort_keys(gt, order)
    assert all(keys[i] <= keys[i + 1] for i in range(len(keys) - 1))
    assert all(len(r["native_" + name]) > 200 for r in native_ranks)
    assert all(r["native_%s_again" % name] == [(1,)] for r in native_ranks)
