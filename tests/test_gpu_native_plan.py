"""The native (C++) plan executor, csrc/plan_exec.cpp, driven through gpuq_plan_create / gpuq_plan_execute: the same plans as
the operator tests, executed without any Python between operators, against the oracle (and against the Python mirror in
plan.py, which must agree row for row)."""
import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
import tpch_util as T
from arrow_ballista_amd.expr import Operator as Op
from arrow_ballista_amd.expr import and_, binary, col, is_not_null, like, lit, or_
from oracle import oracle_np as O
from test_gpu_operators import JOIN_TYPES, agg_cases, close_rows, dev_rows, norm, ora_rows, rand_table

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _node_execute_is_the_mirror(mirror_layer):
    """In this module `plan.execute(0, tc)` is the second opinion the native executor's result (NativePlan / native_rows) is compared with."""



def arrow_rows(t):
    cols = []
    for f, c in zip(t.schema, t.columns):
        if pa.types.is_decimal128(f.type):
            cols.append([None if v is None else int(v.scaleb(f.type.scale)) for v in c.to_pylist()])
        elif pa.types.is_date32(f.type):
            cols.append(c.cast(pa.int32()).to_pylist())
        else:
            cols.append(c.to_pylist())
    return list(zip(*cols)) if cols else []


def native_rows(tc, plan, partition=0):
    """(rows, NativePlan) of a plan executed by the native executor.  Every plan that goes through here also checks
    gpuq_plan_schema: the schema announced BEFORE execution (QueryStageExecutor::schema(), execution_engine.rs:59) is the schema
    of what execution returns: names and types, column by column (the nullable flag of an executed column only says whether a
    validity bitmap was materialised -- the declared nullability is the plan's)."""
    from arrow_ballista_amd.table import type_json
    np_ = g.NativePlan(plan, tc)
    announced = np_.schema()
    r = np_.execute(partition)
    _cols, fields = r.columns_c()
    got = [(fields[i].name.decode(), type_json(fields[i].type, fields[i].precision, fields[i].scale), bool(fields[i].nullable)) for i in range(r.num_columns)]
    assert [(n, t) for n, t, _ in announced] == [(n, t) for n, t, _ in got], (announced, got)
    rows = arrow_rows(r.to_arrow())
    # Every plan that goes through here is executed twice more: from its second execution on a plan runs DEFERRED (operators keep
    # what their synchronous run learned, row counts travel as device words, one host round trip settles everything; DESIGN.md
    # section 2) and must return the same rows in the same order.  Plans that write files are left alone.
    if RERUN and "ShuffleWriterExec" not in np_.json:
        ordered = np_.json.startswith('{"SortExec"') or np_.json.startswith('{"SortPreservingMergeExec"')
        key = (lambda r: r) if ordered else norm      # without an ORDER BY at the root the row order is free (hash tables)
        for _ in range(2):
            again = arrow_rows(np_.execute(partition).to_arrow())
            close_rows(key(again), key(rows))      # (float sums: atomics add in any order, 1e-9 relative as everywhere)
    return rows, np_


RERUN = True


@pytest.mark.parametrize("n", [1, 65, 200_000])
@pytest.mark.parametrize("two_phase", [True, False])
def test_native_q1(tc, n, two_phase):
    li = T.gen_lineitem_device(tc, n, seed=T.SEED_LINEITEM, seed_orders=T.SEED_ORDERS)
    plan = T.q1_plan(g.MemoryExec([li]), two_phase=two_phase)
    got, np_ = native_rows(tc, plan)
    assert [tuple(r) for r in got] == T.q1_oracle_rows(n)
    m = np_.metrics()
    assert m[0]["node"] == "SortExec" and m[0]["output_rows"] == 3 * len(got) and any(x["node"] == "AggregateExec" for x in m)


def test_native_q3_q5(tc):
    import test_gpu_tpch as M
    (li, od, cu, su), (hl, ho, hc, hs) = M._tables(tc, 120_000, 1500, 100)
    q3 = T.q3_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]))
    got, _ = native_rows(tc, q3)
    exp = T.q3_oracle(hc, ho, hl)
    assert [(r[1], r[2]) for r in got] == [(r[1], r[2]) for r in exp] and sorted(got) == sorted(exp) and len(exp) > 0
    nation, region = T.nation_region_arrow()
    q5 = T.q5_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li]), g.MemoryExec([su]), g.MemoryExec([nation]), g.MemoryExec([region]))
    got, _ = native_rows(tc, q5)
    assert [tuple(r) for r in got] == [tuple(r) for r in T.q5_oracle(hc, ho, hl, hs)]


@pytest.mark.parametrize("nulls", [0.0, 0.15])
def test_native_filter_projection_aggregate(tc, nulls):
    t = rand_table(900, 30_000, nulls)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    pred = and_(binary(col("d", s), Op.LtEq, lit(9800, "Date32")), binary(col("k32", s), Op.Gt, lit(-20, "Int32")))
    f = g.FilterExec(pred, src)
    exprs = [(binary(col("dec", s), Op.Multiply, col("dec", s)), "sq"), (col("flag", s), "flag"), (col("k64", s), "k64"), (col("f", s), "f")]
    p = g.ProjectionExec(exprs, g.CoalesceBatchesExec(f))
    got, _ = native_rows(tc, p)
    exp = ora_rows(O.project(ot.take(O.filter_rows(ot, pred)), [e for e, _ in exprs], [n for _, n in exprs]))
    close_rows(got, exp, rel=0.0)
    assert got == dev_rows(tc, p.execute(0, tc))
    ps = p.schema()
    for groups, aggs in agg_cases(s):
        a = g.AggregateExec("Single", groups, aggs, f)
        got, _ = native_rows(tc, a)
        close_rows(norm(got), norm(ora_rows(O.aggregate(ot, groups, aggs, "Single", predicate=pred))))
    # two-phase, the projection fused below the partial aggregate
    aggs = [{"fn": "SUM", "expr": col("sq", ps), "name": "s"}, {"fn": "COUNT", "expr": lit(1), "name": "c"}, {"fn": "AVG", "expr": col("f", ps), "name": "a"}]
    part = g.AggregateExec("Partial", [(col("flag", ps), "flag")], aggs, p)
    fs = part.schema()
    fin = g.AggregateExec("FinalPartitioned", [(col("flag", fs), "flag")], [dict(x, expr=None) for x in aggs], part)
    got, _ = native_rows(tc, fin)
    exp = O.aggregate(O.project(ot.take(O.filter_rows(ot, pred)), [e for e, _ in exprs], [n for _, n in exprs]),
                      [(col("flag", ps), "flag")], aggs, "Single")
    close_rows(norm(got), norm(ora_rows(exp)))


@pytest.mark.parametrize("jt", JOIN_TYPES)
def test_native_hash_join_types(tc, jt):
    nl, nr = 3000, 7000
    lt = rand_table(21, nl, 0.2).append_column("lid", pa.array(np.arange(nl, dtype=np.int64)))
    rt = rand_table(22, nr, 0.2).append_column("rid", pa.array(np.arange(nr, dtype=np.int64)))
    rt = rt.rename_columns([c if c == "rid" else "r_" + c for c in rt.schema.names])
    L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
    ls, rs = L.schema(), R.schema()
    for on in ([(col("k64", ls), col("r_k64", rs))], [(col("k32", ls), col("r_k32", rs)), (col("flag", ls), col("r_flag", rs))]):
        lf = g.FilterExec(is_not_null(col("d", ls)), L)        # fused into the build
        rf = g.FilterExec(binary(col("r_k32", rs), Op.Lt, lit(40, "Int32")), R)   # fused into the probe
        plan = g.HashJoinExec(lf, rf, on, None, jt, "CollectLeft", False)
        got, _ = native_rows(tc, plan)
        assert norm(got) == norm(dev_rows(tc, plan.execute(0, tc)))


@pytest.mark.parametrize("jt", ["Inner", "Left", "Full", "LeftAnti"])
def test_native_collect_left_gathers_every_build_partition(tc, jt):
    """CollectLeft with a build side of several partitions (each with its own fused filter): the native executor collects them
    all into one build table, as DataFusion's collect_left_input does; every probe partition's task sees the whole build side.
    Checked against the oracle's join of the concatenated build partitions, by row ids."""
    lparts, lid0 = [], 0
    for i in range(3):
        t = rand_table(31 + i, 1500 + 300 * i, 0.2)
        lparts.append(t.append_column("lid", pa.array(np.arange(lid0, lid0 + t.num_rows, dtype=np.int64)))); lid0 += t.num_rows
    rparts = []
    for i in range(2):
        t = rand_table(41 + i, 4000, 0.2).append_column("rid", pa.array(np.arange(4000, dtype=np.int64)))
        rparts.append(t.rename_columns([c if c == "rid" else "r_" + c for c in t.schema.names]))
    L, R = g.MemoryExec(lparts), g.MemoryExec(rparts)
    ls, rs = L.schema(), R.schema()
    lpred = is_not_null(col("d", ls))
    on = [(col("k64", ls), col("r_k64", rs))]
    plan = g.HashJoinExec(g.FilterExec(lpred, L), R, on, None, jt, "CollectLeft", False)
    js = plan.schema()
    proj = g.ProjectionExec([(col("lid", js), "lid")] + ([] if jt == "LeftAnti" else [(col("rid", js), "rid")]), plan)
    lo = O.Table.from_arrow(pa.concat_tables(lparts))
    native = g.NativePlan(proj, tc)
    for q in range(2):
        got = sorted(tuple(-1 if v is None else v for v in r) for r in arrow_rows(native.execute(q).to_arrow()))
        ro = O.Table.from_arrow(rparts[q])
        pairs = O.hash_join(lo, ro, on, jt, left_pred=lpred)
        lid, rid = lo.col("lid"), ro.col("rid")
        exp = sorted((-1 if i is None else lid[i],) if jt == "LeftAnti" else (-1 if i is None else lid[i], -1 if j is None else rid[j]) for i, j in pairs)
        assert got == exp and len(exp) > 0


def test_native_sort_fetch_and_limit(tc):
    t = rand_table(77, 50_000, 0.1)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    spec = [{"expr": col("flag", s), "asc": True, "nulls_first": False}, {"expr": col("dec", s), "asc": False, "nulls_first": True},
            {"expr": col("k64", s), "asc": True, "nulls_first": True}]
    got, _ = native_rows(tc, g.SortExec(spec, src, fetch=100))
    exp = [tuple(c[i] for c in ot.cols) for i in O.sort_perm(ot, spec)[:100]]
    close_rows(got, exp, rel=0.0)
    got, _ = native_rows(tc, g.LocalLimitExec(g.SortExec(spec[:1], g.FilterExec(is_not_null(col("flag", s)), src)), 7))
    assert len(got) == 7 and all(r[6] == got[0][6] for r in got)


@pytest.mark.parametrize("nulls", [0.0, 0.2])
def test_native_fan_in_merge_limits_union(tc, nulls):
    parts = [rand_table(500 + i, n, nulls) for i, n in enumerate([1000, 0, 77, 4099, 1])]
    src = g.MemoryExec(parts)
    s = src.schema()
    for plan in (g.CoalesceTasksExec(src, [0, 1, 2, 3, 4]), g.CoalesceTasksExec(src, [3, 2]), g.CoalesceTasksExec(src, [2]), g.CoalescePartitionsExec(src),
                 g.CoalescePartitionsExec(g.UnionExec([src, g.MemoryExec(parts[:2])]))):
        got, _ = native_rows(tc, plan)
        assert got == dev_rows(tc, plan.execute(0, tc))
    order = [{"expr": col("flag", s), "asc": True, "nulls_first": False}, {"expr": col("dec", s), "asc": False, "nulls_first": True}]
    sorted_src = g.SortExec(order, src, preserve_partitioning=True)
    for plan in (g.CoalesceTasksExec(sorted_src, [0, 3], order_by=order), g.SortPreservingMergeExec(order, sorted_src, fetch=10)):
        got, _ = native_rows(tc, plan)
        assert got == dev_rows(tc, plan.execute(0, tc))
    one = g.CoalescePartitionsExec(src)
    f = g.FilterExec(binary(col("k32", s), Op.Gt, lit(0, "Int32")), src)
    for plan in (g.GlobalLimitExec(one, skip=998, fetch=13), g.GlobalLimitExec(one, skip=3, fetch=None), g.GlobalLimitExec(one, skip=99999, fetch=4),
                 g.GlobalLimitExec(g.CoalesceTasksExec(f, [0]), skip=5, fetch=70), g.GlobalLimitExec(g.CoalesceTasksExec(src, [3]), skip=0, fetch=9)):
        got, _ = native_rows(tc, plan)
        assert got == dev_rows(tc, plan.execute(0, tc))
    # context.rs:691-733 through the native executor: UNION ALL keeps both rows, UNION (aggregate over all columns) one
    a = pa.table({"NUMBER": pa.array([1], pa.int64())})
    u = g.UnionExec([g.MemoryExec([a]), g.MemoryExec([a])])
    assert native_rows(tc, g.CoalescePartitionsExec(u))[0] == [(1,), (1,)]
    us = u.schema()
    assert native_rows(tc, g.AggregateExec("Single", [(col("NUMBER", us), "NUMBER")], [], g.CoalescePartitionsExec(u)))[0] == [(1,)]


def test_native_unsupported_node_fails_loudly(tc):
    t = rand_table(1, 10, 0.0)
    with pytest.raises(g.GpuqError):
        g.NativePlan(g.RepartitionExec(g.MemoryExec([t]), [col("k64", g.MemoryExec([t]).schema())], 4), tc)


def test_long_strings_travel_as_payload(tc):
    """Utf8 values of any length pass through filters, joins, sorts and materialisation as payload (gpuq_take_utf8); only
    comparisons / keys are limited to 15 bytes."""
    n = 3000
    r = np.random.default_rng(8)
    words = ["", "x", "exactly15bytes!", "a comment that is clearly longer than fifteen bytes", "δοκιμή utf-8 ✓ multibyte text that is long"]
    comment = pa.array([words[i] if r.random() > 0.1 else None for i in r.integers(0, len(words), n)])
    lt = pa.table({"k": pa.array(r.integers(0, 500, n), pa.int64()), "comment": comment, "v": pa.array(np.arange(n, dtype=np.int64))})
    rt = pa.table({"rk": pa.array(np.arange(500, dtype=np.int64)), "name": pa.array(["supplier#%09d of somewhere far away" % i for i in range(500)])})
    L, R = g.MemoryExec([rt]), g.MemoryExec([lt])
    ls, rs = L.schema(), R.schema()
    f = g.FilterExec(binary(col("v", rs), Op.Gt, lit(99)), R)
    j = g.HashJoinExec(L, f, [(col("rk", ls), col("k", rs))], None, "Inner", "CollectLeft", False)
    js = j.schema()
    plan = g.SortExec([{"expr": col("v", js), "asc": False, "nulls_first": False}], j, fetch=200)
    exp = []
    lk, lc, lv = lt["k"].to_pylist(), lt["comment"].to_pylist(), lt["v"].to_pylist()
    names = rt["name"].to_pylist()
    for i in sorted(range(n), key=lambda i: -lv[i]):
        if lv[i] > 99:
            exp.append((lk[i], names[lk[i]], lk[i], lc[i], lv[i]))
    exp = exp[:200]
    assert dev_rows(tc, plan.execute(0, tc)) == exp
    got, _ = native_rows(tc, plan)
    assert got == exp
    # fan-in of partitions that carry long strings (filtered views and plain tables mixed)
    fan = g.CoalescePartitionsExec(g.UnionExec([g.MemoryExec([lt.slice(0, 700), lt.slice(700, 1)]), g.FilterExec(binary(col("v", rs), Op.Lt, lit(50)), g.MemoryExec([lt]))]))
    want = [tuple(r.values()) for r in lt.slice(0, 701).to_pylist()] + [tuple(r.values()) for r in lt.slice(0, 50).to_pylist()]
    assert dev_rows(tc, fan.execute(0, tc)) == want
    assert native_rows(tc, fan)[0] == want


def test_native_stage_driver_writes_and_reads_shuffle_files(tc, tmp_path):
    """The stage driver in the native executor: ShuffleWriterExec root (hash repartition + device LZ4 + IPC files in the
    reference's path layout, result batch = partition / path / num_rows / num_batches / num_bytes as shuffle_writer.rs:470-520)
    and a ShuffleReaderExec leaf feeding a reduce-side aggregate.  Files are also read by Arrow C++; rows against the oracle."""
    import os
    parts = [rand_table(700 + i, 5000, 0.1) for i in range(2)]
    src = g.MemoryExec(parts)
    s = src.schema()
    pred = binary(col("k32", s), Op.Gt, lit(-30, "Int32"))
    nred = 3
    writer = g.ShuffleWriterExec("jobN", 2, g.FilterExec(pred, src), str(tmp_path), ([col("k64", s)], nred), partitions=[0, 1])
    wp = g.NativePlan(writer, tc)
    files = []
    for p in range(2):
        files += arrow_rows(wp.execute(p).to_arrow())
    ot = O.Table.from_arrow(pa.concat_tables(parts))
    keep = O.filter_rows(ot, pred)
    assert sum(f[2] for f in files) == len(keep)
    seen = []
    for pid, path, rows, batches, nbytes in files:
        assert path.startswith(os.path.join(str(tmp_path), "jobN", "2", str(pid)) + os.sep) and path.endswith(".arrow")
        assert os.path.getsize(path) == nbytes and batches == 1 and rows > 0
        rd = pa.ipc.open_stream(path).read_all()
        assert rd.num_rows == rows
        assert set(O.hash_partition(O.Table.from_arrow(rd), [col("k64", s)], nred)) == {pid}
        seen += ora_rows(O.Table.from_arrow(rd))
    close_rows(norm(seen), norm(ora_rows(ot.take(keep))))
    mets = {m["node"]: m for m in wp.metrics()}
    assert mets["ShuffleWriterExec"]["output_rows"] == len(keep) and mets["ShuffleWriterExec"]["write_time"] > 0 and mets["ShuffleWriterExec"]["repart_time"] > 0
    assert mets["ShuffleWriterExec"]["input_rows"] == len(keep)
    assert mets["ShuffleWriterExec"]["partitions"] == [0, 1]      # the stage partitions of the task (shuffle_writer.rs:118-119), echoed for ShuffleWritePartition
    # reduce side
    reader = g.ShuffleReaderExec([[{"path": f[1]} for f in files if f[0] == q] for q in range(nred)], s)
    rs = reader.schema()
    aggs = [{"fn": "SUM", "expr": col("dec", rs), "name": "sd"}, {"fn": "COUNT", "expr": lit(1), "name": "c"}, {"fn": "MIN", "expr": col("d", rs), "name": "md"}]
    rp = g.NativePlan(g.AggregateExec("Single", [(col("k64", rs), "k64")], aggs, reader), tc)
    got = []
    for q in range(nred):
        got += arrow_rows(rp.execute(q).to_arrow())
    exp = ora_rows(O.aggregate(ot.take(keep), [(col("k64", s), "k64")], aggs, "Single"))
    assert norm(got) == norm(exp)
    # a hash-partitioned file names its partition function; a partition whose files disagree is refused (a stage whose map tasks
    # ran partly on a CPU executor -- ahash, no marker -- and partly here would otherwise mis-route equal keys silently)
    first = next(f for f in files if f[0] == 0)
    assert pa.ipc.open_stream(first[1]).schema.metadata == {b"gpuq.partition_fn": b"gpuq-mix64-v1"}
    foreign = str(tmp_path / "cpu_written.arrow")
    with pa.OSFile(foreign, "wb") as fo, pa.ipc.new_stream(fo, pa.ipc.open_stream(first[1]).schema.remove_metadata(), options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
        w.write_table(pa.ipc.open_stream(first[1]).read_all().replace_schema_metadata(None))
    mixed = g.NativePlan(g.ShuffleReaderExec([[{"path": first[1]}, {"path": foreign}]], s), tc)
    with pytest.raises(g.GpuqError, match="same engine"):
        mixed.execute(0)
    assert g.NativePlan(g.ShuffleReaderExec([[{"path": foreign}, {"path": foreign}]], s), tc).execute(0).num_rows == 2 * first[2]
    # unpartitioned stage: one data.arrow per task; strings longer than 15 bytes travel; missing file = FetchFailed
    w2 = g.NativePlan(g.ShuffleWriterExec("jobN", 3, src, str(tmp_path)), tc)
    (pid, path, rows, batches, nbytes), = arrow_rows(w2.execute(1).to_arrow())
    assert pid == 1 and path.endswith("data.arrow") and rows == 5000
    back = pa.ipc.open_stream(path).read_all()
    assert not pa.ipc.open_stream(path).schema.metadata          # no partition function involved
    assert norm(ora_rows(O.Table.from_arrow(back))) == norm(ora_rows(O.Table.from_arrow(parts[1])))
    bad = g.NativePlan(g.ShuffleReaderExec([[{"path": str(tmp_path / "nope.arrow")}]], s), tc)
    with pytest.raises(g.GpuqError, match="FetchFailed"):
        bad.execute(0)
    assert g.NativePlan(g.ShuffleReaderExec([[]], s), tc).execute(0).num_rows == 0


def test_native_like(tc):
    """LIKE in the native executor: FilterExec (alone, over a view, under an aggregate that fuses its input) and ProjectionExec,
    row for row against the oracle."""
    from test_gpu_operators import LIKE_PATTERNS, like_table
    t = like_table(78, 4000, 0.15)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    for pat in LIKE_PATTERNS[::2]:
        e = like(col("s", s), pat, negated=(len(pat) % 2 == 1))
        got, _ = native_rows(tc, g.FilterExec(e, src))
        assert got == ora_rows(ot.take(O.filter_rows(ot, e))), pat
    inner = g.FilterExec(binary(col("k", s), Op.Gt, lit(10)), src)
    pred = or_(like(col("s", s), "%ING"), like(col("s", s), "%special%requests%"))
    outer = g.FilterExec(pred, inner)
    keep = O.filter_rows(ot, and_(binary(col("k", s), Op.Gt, lit(10)), pred))
    assert native_rows(tc, outer)[0] == ora_rows(ot.take(keep))
    aggs = [{"fn": "COUNT", "expr": lit(1), "name": "c"}, {"fn": "SUM", "expr": col("k", s), "name": "sk"}]
    got, _ = native_rows(tc, g.AggregateExec("Single", [(col("k", s), "k")], aggs, outer))
    assert norm(got) == norm(ora_rows(O.aggregate(ot.take(keep), [(col("k", s), "k")], aggs, "Single")))
    exprs = [(like(col("s", s), "B%G"), "l"), (binary(col("k", s), Op.Multiply, lit(2)), "k2")]
    got, _ = native_rows(tc, g.ProjectionExec(exprs, inner))
    sub = ot.take(O.filter_rows(ot, binary(col("k", s), Op.Gt, lit(10))))
    assert got == ora_rows(O.project(sub, [e for e, _ in exprs], [n for _, n in exprs]))


def test_async_execute_and_cancel(tc):
    """gpuq_plan_execute_async / gpuq_task_cancel: the reference cancels a task by dropping its future (executor.rs:201-240).
    A q3 over 3 M lineitem rows is started on a worker thread and cancelled at once; the wait reports CANCELLED (or, if the
    task won the race, its complete result); the same plan, stream and context then run the query again, synchronously and
    asynchronously, with the oracle's rows; dropping a running task (gpuq_task_free) is a cancel too."""
    n_li, n_cust = 3_000_000, 30_000
    cols = ("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate")
    li = T.gen_lineitem_device(tc, n_li, n_supp=100, columns=cols)
    od = T.gen_orders_device(tc, n_li // 4, n_cust)
    cu = T.gen_customer_device(tc, n_cust)
    plan = g.NativePlan(T.q3_plan(g.MemoryExec([cu]), g.MemoryExec([od]), g.MemoryExec([li])), tc)
    exp, _st = T.q3_oracle_c(T.gen_q3_tables_host(n_li, n_cust))
    want = sorted(exp)
    assert sorted(tuple(r) for r in arrow_rows(plan.execute(0).to_arrow())) == want      # warm (JIT compiled)
    cancelled = 0
    for it in range(6):
        task = plan.execute_async(0)
        if it % 2 == 0:
            task.cancel()
        try:
            res = task.wait()
            assert sorted(tuple(r) for r in arrow_rows(res.to_arrow())) == want
        except g.GpuqError as e:
            assert e.status == 6 and it % 2 == 0, e
            cancelled += 1
        assert task.done()
        task.close()
        # the plan, the stream and the pool are intact: the next run is right
        assert sorted(tuple(r) for r in arrow_rows(plan.execute(0).to_arrow())) == want
    assert cancelled >= 1, "no cancel ever arrived before the task finished"
    t2 = plan.execute_async(0)
    t2.close()                                              # free while running = cancel + join
    assert sorted(tuple(r) for r in arrow_rows(plan.execute(0).to_arrow())) == want
