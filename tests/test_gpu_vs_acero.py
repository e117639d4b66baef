"""The HIP path against pyarrow ACERO directly -- no oracle in the loop.  VERDICT r2: for hash-join output, sort order, grouped
aggregation and filter output the reference asserts nothing, so every other parity test compares the device with a restatement
written by the same author; a shared misreading of the semantics would be invisible.  Acero (Arrow C++'s hash join / group-by /
sort / filter) is an independent implementation of the same relational semantics: these tests run plans through the native executor
(synchronously, then deferred) and compare the rows with Acero's on raw integers -- Acero's decimal TYPING differs from DataFusion's
(SURVEY.md section 8c), so decimals are compared as unscaled integers and the typing rules stay with the pinned tests.
Covered: all 8 join types x {one key, two keys} with duplicate and NULL keys on both sides and filters fused into each side;
grouped SUM / COUNT / MIN / MAX (+ COUNT(*)) with NULL keys and NULL values; multi-key ORDER BY with NULLS FIRST / LAST; FilterExec
with AND / OR / NOT over nullable columns (a NULL predicate drops the row)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd.expr import Operator as Op
from arrow_ballista_amd.expr import and_, binary, col, is_not_null, lit, not_, or_
from test_gpu_native_plan import native_rows

pytestmark = pytest.mark.gpu


def _table(seed, n, nkeys, nulls=0.15, prefix=""):
    r = np.random.default_rng(seed)

    def mask():
        return r.random(n) < nulls
    t = pa.table({
        "k": pa.array(r.integers(0, nkeys, n), pa.int64(), mask=mask()),
        "g": pa.array(r.integers(0, 4, n).astype(np.int32), pa.int32(), mask=mask()),
        "v": pa.array(r.integers(-10**6, 10**6, n), pa.int64(), mask=mask()),
        "d": pa.array(r.integers(-10**9, 10**9, n), pa.int64(), mask=mask()),       # cast to Decimal128(15,2) on the device side below
        "id": pa.array(np.arange(n), pa.int64()),
    })
    return t.rename_columns([prefix + c for c in t.column_names])


def _key(rows):
    return sorted(rows, key=lambda r: tuple((x is None, 0 if x is None else x) for x in r))


JOINS = [("Inner", "inner"), ("Left", "left outer"), ("Right", "right outer"), ("Full", "full outer"),
         ("LeftSemi", "left semi"), ("LeftAnti", "left anti"), ("RightSemi", "right semi"), ("RightAnti", "right anti")]


@pytest.mark.parametrize("jt,how", JOINS)
@pytest.mark.parametrize("two_keys", [False, True])
def test_join_rows_equal_aceros(tc, jt, how, two_keys):
    lt, rt = _table(11, 1500, 600), _table(12, 4000, 600, prefix="r_")
    L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
    ls, rs = L.schema(), R.schema()
    # a filter fused into each side (build and probe)
    lf = g.FilterExec(binary(col("v", ls), Op.Gt, lit(-900_000)), L)
    rf = g.FilterExec(binary(col("r_v", rs), Op.Lt, lit(900_000)), R)
    on = [(col("k", ls), col("r_k", rs))] + ([(col("g", ls), col("r_g", rs))] if two_keys else [])
    got, _ = native_rows(tc, g.HashJoinExec(lf, rf, on, None, jt, "CollectLeft", False))
    la = lt.filter(pc.greater(lt["v"], -900_000))            # (a NULL predicate drops the row in Acero's filter too)
    ra = rt.filter(pc.less(rt["r_v"], 900_000))
    a = la.join(ra, keys=["k"] + (["g"] if two_keys else []), right_keys=["r_k"] + (["r_g"] if two_keys else []), join_type=how, coalesce_keys=False)
    if "Semi" in jt or "Anti" in jt:
        names = lt.column_names if jt.startswith("Left") else rt.column_names
    else:
        names = lt.column_names + rt.column_names
    exp = list(zip(*[a[c].to_pylist() for c in names]))
    assert len(exp) > 0
    assert _key(got) == _key(exp)


def test_grouped_aggregates_equal_aceros(tc):
    t = _table(21, 200_000, 300)
    src = g.MemoryExec([t])
    s = src.schema()
    aggs = [{"fn": "SUM", "expr": col("v", s), "name": "s"}, {"fn": "COUNT", "expr": col("v", s), "name": "c"}, {"fn": "MIN", "expr": col("v", s), "name": "mn"},
            {"fn": "MAX", "expr": col("v", s), "name": "mx"}, {"fn": "COUNT", "expr": lit(1), "name": "n"}, {"fn": "SUM", "expr": col("d", s), "name": "sd"}]
    for keys in (["k"], ["k", "g"], ["g"]):
        for strategy in ("auto", "hash"):
            got, _ = native_rows(tc, g.AggregateExec("Single", [(col(k, s), k) for k in keys], aggs, src, strategy=strategy))
            a = t.group_by(keys, use_threads=False).aggregate([("v", "sum"), ("v", "count"), ("v", "min"), ("v", "max"), ([], "count_all"), ("d", "sum")])
            exp = list(zip(*[a[c].to_pylist() for c in keys + ["v_sum", "v_count", "v_min", "v_max", "count_all", "d_sum"]]))
            assert _key(got) == _key(exp), (keys, strategy)
    # two-phase (Partial -> FinalPartitioned) must give the same groups
    part = g.AggregateExec("Partial", [(col("k", s), "k")], aggs, src)
    fs = part.schema()
    fin = g.AggregateExec("FinalPartitioned", [(col("k", fs), "k")], [dict(x, expr=None) for x in aggs], part)
    got, _ = native_rows(tc, fin)
    a = t.group_by(["k"], use_threads=False).aggregate([("v", "sum"), ("v", "count"), ("v", "min"), ("v", "max"), ([], "count_all"), ("d", "sum")])
    assert _key(got) == _key(list(zip(*[a[c].to_pylist() for c in ["k", "v_sum", "v_count", "v_min", "v_max", "count_all", "d_sum"]])))


@pytest.mark.parametrize("nulls_first", [True, False])
@pytest.mark.parametrize("n", [300, 5000, 300_000])
def test_multi_key_sort_order_equals_aceros(tc, nulls_first, n):
    t = _table(31, n, 40)
    src = g.MemoryExec([t])
    s = src.schema()
    spec = [{"expr": col("g", s), "asc": True, "nulls_first": nulls_first}, {"expr": col("k", s), "asc": False, "nulls_first": nulls_first},
            {"expr": col("v", s), "asc": True, "nulls_first": nulls_first}]
    got, _ = native_rows(tc, g.SortExec(spec, src))
    idx = pc.sort_indices(t, sort_keys=[("g", "ascending"), ("k", "descending"), ("v", "ascending")], null_placement="at_start" if nulls_first else "at_end")
    exp = list(zip(*[t.take(idx)[c].to_pylist() for c in t.column_names]))
    assert [r[:3] for r in got] == [r[:3] for r in exp]          # the key columns, position by position
    assert got == exp                                            # and, both sorts being stable, every row (ties keep input order)


def test_filter_output_equals_aceros(tc):
    t = _table(41, 100_000, 1000)
    src = g.MemoryExec([t])
    s = src.schema()
    cases = [
        (binary(col("v", s), Op.Gt, lit(0)), pc.greater(t["v"], 0)),
        (and_(binary(col("v", s), Op.Gt, lit(0)), binary(col("g", s), Op.NotEq, lit(2, "Int32"))), pc.and_kleene(pc.greater(t["v"], 0), pc.not_equal(t["g"], 2))),
        (or_(binary(col("k", s), Op.Lt, lit(100)), binary(col("d", s), Op.GtEq, lit(0))), pc.or_kleene(pc.less(t["k"], 100), pc.greater_equal(t["d"], 0))),
        (not_(binary(col("k", s), Op.Eq, col("v", s))), pc.invert(pc.equal(t["k"], t["v"]))),
        (and_(is_not_null(col("k", s)), binary(binary(col("v", s), Op.Plus, col("d", s)), Op.Lt, lit(0))), pc.and_kleene(pc.is_valid(t["k"]), pc.less(pc.add(t["v"], t["d"]), 0))),
    ]
    for pred, mask in cases:
        got, _ = native_rows(tc, g.FilterExec(pred, src))
        a = t.filter(mask, null_selection_behavior="drop")
        assert got == list(zip(*[a[c].to_pylist() for c in t.column_names])) and a.num_rows > 0


def test_more_than_four_group_columns_equal_aceros(tc):
    """GROUP BY over six and seven columns (the aggregate's table holds four keys: narrow keys are packed into 126-bit composites --
    biased, with a NULL bit -- and unpacked over the groups; strings take a slot of their own or travel as dictionary codes): nullable
    keys, negative values, narrow integer types, a date, a short and a long string, two-phase as well."""
    r = np.random.default_rng(51)
    n = 120_000

    def mask(p=0.15):
        return r.random(n) < p
    words = np.array(["", "a", "BUILDING", "exactly15bytes!", "a key that is longer than fifteen bytes", "another long key, same length.........."])
    t = pa.table({
        "a": pa.array(r.integers(-3, 4, n).astype(np.int8), pa.int8(), mask=mask()),
        "b": pa.array(r.integers(-300, 300, n).astype(np.int16) // 100, pa.int16()),
        "c": pa.array(r.integers(-2**31, 2**31 - 1, n).astype(np.int32) // 2**29, pa.int32(), mask=mask()),
        "d": pa.array(r.integers(-2**62, 2**62, n) // 2**60, pa.int64(), mask=mask()),
        "e": pa.array(r.integers(9000, 9004, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "s": pa.array(words[r.integers(0, 4, n)], pa.string(), mask=mask()),
        "l": pa.array(words[r.integers(0, 6, n)], pa.string(), mask=mask(0.05)),
        "v": pa.array(r.integers(-10**6, 10**6, n), pa.int64(), mask=mask()),
    })
    src = g.MemoryExec([t])
    s = src.schema()
    aggs = [{"fn": "SUM", "expr": col("v", s), "name": "sv"}, {"fn": "COUNT", "expr": lit(1), "name": "n"}, {"fn": "MIN", "expr": col("v", s), "name": "mn"}]
    for keys in (["a", "b", "c", "d", "e", "s"], ["a", "b", "c", "d", "e", "s", "l"], ["l", "d", "c", "b", "a"]):
        got, _ = native_rows(tc, g.AggregateExec("Single", [(col(k, s), k) for k in keys], aggs, src))
        a = t.group_by(keys, use_threads=False).aggregate([("v", "sum"), ([], "count_all"), ("v", "min")])
        cols = [a[k].cast(pa.int32()).to_pylist() if k == "e" else a[k].to_pylist() for k in keys] + [a["v_sum"].to_pylist(), a["count_all"].to_pylist(), a["v_min"].to_pylist()]
        exp = list(zip(*cols))
        key = lambda rows: sorted(rows, key=lambda r_: tuple((x is None, "" if x is None else (x if isinstance(x, str) else 0), 0 if x is None or isinstance(x, str) else x) for x in r_))
        assert key(got) == key(exp) and len(exp) > 500, keys
    keys = ["a", "b", "c", "d", "e"]
    part = g.AggregateExec("Partial", [(col(k, s), k) for k in keys], aggs, src)
    fs = part.schema()
    fin = g.AggregateExec("FinalPartitioned", [(col(k, fs), k) for k in keys], [dict(x, expr=None) for x in aggs], part)
    got, _ = native_rows(tc, fin)
    a = t.group_by(keys, use_threads=False).aggregate([("v", "sum"), ([], "count_all"), ("v", "min")])
    exp = list(zip(*([a[k].cast(pa.int32()).to_pylist() if k == "e" else a[k].to_pylist() for k in keys] + [a["v_sum"].to_pylist(), a["count_all"].to_pylist(), a["v_min"].to_pylist()])))
    assert _key(got) == _key(exp)


@pytest.mark.parametrize("asc,nulls_first", [(True, False), (False, True), (True, True), (False, False)])
def test_single_key_sort_hands_the_key_column_back_in_order(tc, asc, nulls_first):
    """ORDER BY one plain integer-like column over more rows than one block sorts: the sorted key column is rebuilt from the sort's own
    records (gpuq_sort_run_keys) instead of gathered -- values, NULLs, and every other column (through the permutation) must be what a stable
    sort gives, for every key type the decode takes, ascending and descending, NULLS FIRST and LAST; then once more deferred."""
    import decimal
    r = np.random.default_rng(61)
    n = (1 << 20) + 12345

    def m():
        return r.random(n) < 0.05
    t = pa.table({
        "i64": pa.array(r.integers(-10**12, 10**12, n), pa.int64(), mask=m()),
        "i32": pa.array(r.integers(-2**31, 2**31 - 1, n).astype(np.int32), pa.int32(), mask=m()),
        "i16": pa.array(r.integers(-2**15, 2**15, n).astype(np.int16), pa.int16()),
        "u8": pa.array(r.integers(0, 256, n).astype(np.uint8), pa.uint8(), mask=m()),
        "d32": pa.array(r.integers(-5000, 20000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "ts": pa.array(r.integers(0, 10**15, n), pa.timestamp("us"), mask=m()),
        "dec": pa.array(r.integers(-10**11, 10**11, n), pa.int64()).cast(pa.decimal128(19, 0)).cast(pa.decimal128(21, 2)),
        "id": pa.array(np.arange(n), pa.int64()),
        "s": pa.array(np.array(["a", "bb", "a string longer than fifteen bytes"])[r.integers(0, 3, n)]),
    })
    src = g.MemoryExec([t])
    s = src.schema()
    for key in ("i64", "i32", "i16", "u8", "d32", "ts", "dec"):
        plan = g.NativePlan(g.SortExec([{"expr": col(key, s), "asc": asc, "nulls_first": nulls_first}], src), tc)
        idx = pc.sort_indices(t, sort_keys=[(key, "ascending" if asc else "descending")], null_placement="at_start" if nulls_first else "at_end")
        want = t.take(idx)
        for _ in range(2):      # synchronous, then deferred
            out = plan.execute(0).to_arrow()
            for name in t.column_names:
                a, b = out[name].combine_chunks(), want[name].combine_chunks()
                if pa.types.is_timestamp(b.type):
                    a, b = a.cast(pa.int64()), b.cast(pa.int64())
                assert a.equals(b), (key, name)
