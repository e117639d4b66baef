"""SortExec on the single-read radix passes (kernels_sort.hip, decoupled look-back): the permutation must be THE stable order --
equal keys keep their input order -- at sizes around the 4096-key tiles, for packed (<= 32 key bits), one-word and two-word
composite keys, with the duplicates that make instability visible.  Checker: numpy's stable argsort / lexsort on the same arrays."""
import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd.expr import col

pytestmark = pytest.mark.gpu


def sorted_row_ids(tc, t, spec_cols):
    src = g.MemoryExec([t])
    s = src.schema()
    spec = [{"expr": col(c, s), "asc": asc, "nulls_first": False} for c, asc in spec_cols]
    out = g.plan.materialize(tc, g.SortExec(spec, g.ProjectionExec([(col("rid", s), "rid")] + [(col(c, s), c) for c, _ in spec_cols], src)).execute(0, tc))
    return np.asarray(out.to_arrow(tc.ctx).column("rid"))


@pytest.mark.parametrize("n", [2049, 4096, 4097, 8191, 100_003, (1 << 20) + 5])
def test_stable_order_for_every_key_width(tc, n):
    r = np.random.default_rng(n)
    rid = np.arange(n, dtype=np.int64)
    k20 = r.integers(0, 1 << 20, n).astype(np.int32)                       # 3 packed passes, the last one 4 bits wide
    few = r.integers(-3, 4, n).astype(np.int64)                            # 3 bits: one pass, ~n/7 duplicates per value
    k40 = (r.integers(0, 1 << 40, n) - (1 << 39)).astype(np.int64)         # one u64 word, 5 passes
    wide_a = r.integers(-2**62, 2**62, n).astype(np.int64) >> r.integers(0, 60, n)      # all magnitudes
    wide_b = r.integers(-2**62, 2**62, n).astype(np.int64)
    dup_a = wide_a[r.integers(0, max(1, n // 50), n)]                      # two-word key whose first field repeats
    t = pa.table({"rid": rid, "k20": k20, "few": few, "k40": k40, "dup_a": dup_a, "wide_b": wide_b})
    t = t.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in t.schema]))
    assert np.array_equal(sorted_row_ids(tc, t, [("k20", True)]), np.argsort(k20, kind="stable"))
    assert np.array_equal(sorted_row_ids(tc, t, [("few", True)]), np.argsort(few, kind="stable"))
    assert np.array_equal(sorted_row_ids(tc, t, [("few", False)]), np.argsort(-few, kind="stable"))
    assert np.array_equal(sorted_row_ids(tc, t, [("k40", True)]), np.argsort(k40, kind="stable"))
    assert np.array_equal(sorted_row_ids(tc, t, [("few", True), ("k20", False)]), np.lexsort((-k20.astype(np.int64), few)))
    assert np.array_equal(sorted_row_ids(tc, t, [("dup_a", True), ("wide_b", False)]), np.lexsort((-wide_b, dup_a)))      # |wide_b| < 2^62: no overflow


def test_all_keys_equal_and_sorted_inputs(tc):
    n = 50_000
    t = pa.table({"rid": np.arange(n, dtype=np.int64), "c": np.full(n, 7, np.int64), "up": np.arange(n, dtype=np.int64), "down": np.arange(n, dtype=np.int64)[::-1].copy()})
    t = t.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in t.schema]))
    assert np.array_equal(sorted_row_ids(tc, t, [("c", True)]), np.arange(n))
    assert np.array_equal(sorted_row_ids(tc, t, [("up", True)]), np.arange(n))
    assert np.array_equal(sorted_row_ids(tc, t, [("down", True)]), np.arange(n)[::-1])
    assert np.array_equal(sorted_row_ids(tc, t, [("up", False)]), np.arange(n)[::-1])


# ------------------------------------------------------------------ large inputs: key layout guessed from a sample, verified by the pack kernel
def test_guessed_key_layout_holds_or_falls_back(tc):
    """From 2^22 rows on gpuq_sort_run sizes the composite key from the min/max of every 16th 64-row word (here) and the pack kernel
    checks each row against it.  The permutation must be the stable order whether the guess holds (uniform keys), or does not:
    an outlier / a NULL in a word the sample skips, keys far outside a clustered sample."""
    n = (1 << 22) + 77
    r = np.random.default_rng(22)
    rid = np.arange(n, dtype=np.int64)
    price = r.integers(90_000, 10_494_951, n).astype(np.int64)              # 24 bits: packed records, 3 passes
    date = r.integers(8000, 10_600, n).astype(np.int32)
    low = price.copy(); low[100] = -5_000_000_000                           # row 100 is in word 1: not sampled
    high = price.copy(); high[n - 3] = 1 << 50
    hidden = r.integers(0, 1000, n).astype(np.int64); hidden[64:] += np.where(np.arange(64, n) % 1024 >= 64, 1 << 30, 0)   # sampled words hold 0..999 only
    t = pa.table({"rid": rid, "price": price, "date": date, "low": low, "high": high, "hidden": hidden})
    t = t.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in t.schema]))
    assert np.array_equal(sorted_row_ids(tc, t, [("price", True)]), np.argsort(price, kind="stable"))
    assert np.array_equal(sorted_row_ids(tc, t, [("price", False), ("date", True)]), np.lexsort((date, -price)))
    assert np.array_equal(sorted_row_ids(tc, t, [("low", True)]), np.argsort(low, kind="stable"))
    assert np.array_equal(sorted_row_ids(tc, t, [("high", False)]), np.argsort(-high, kind="stable"))
    assert np.array_equal(sorted_row_ids(tc, t, [("date", True), ("hidden", True)]), np.lexsort((hidden, date)))
    # a NULL the sample does not see: the guessed layout has no null bit
    mask = np.zeros(n, bool); mask[200] = True; mask[n - 1] = True
    tn = pa.table({"rid": rid, "v": pa.array(price, mask=mask)})
    src = g.MemoryExec([tn]); s = src.schema()
    for nulls_first in (True, False):
        plan = g.SortExec([{"expr": col("v", s), "asc": True, "nulls_first": nulls_first}], src)
        got = np.asarray(g.plan.materialize(tc, plan.execute(0, tc)).to_arrow(tc.ctx).column("rid"))
        vals = np.argsort(np.where(mask, np.iinfo(np.int64).min if nulls_first else np.iinfo(np.int64).max, price), kind="stable")
        assert np.array_equal(got, vals)
        got2 = np.asarray(g.plan.materialize(tc, plan.execute(0, tc)).to_arrow(tc.ctx).column("rid"))      # the operator has stopped guessing by now
        assert np.array_equal(got2, vals)


# ------------------------------------------------------------------ ordered fan-in (gpuq_merge_run)
def run_table(r, n, wide):
    """n rows sorted by the test's key(s): few distinct values (ties across runs) or full-range two-field keys."""
    a = np.sort(r.integers(0, 50, n).astype(np.int64)) if not wide else r.integers(-2**62, 2**62, n).astype(np.int64)
    b = r.integers(-2**62, 2**62, n).astype(np.int64)
    if wide:
        order = np.lexsort((b, a)); a, b = a[order], b[order]
    return a, b


@pytest.mark.parametrize("wide", [False, True], ids=["one-word-ties", "two-word"])
@pytest.mark.parametrize("sizes", [[5, 0, 7], [1], [2048, 2049], [1000, 0, 0, 3000, 17, 4096, 1], [30_000] * 16, [100_003, 5, 70_001]])
def test_merge_of_sorted_runs_is_the_stable_sort_of_their_concatenation(tc, sizes, wide):
    """CoalesceTasksExec(order_by) over MemoryExec partitions that are each sorted: merge-path rounds on the device.  Ties must
    come out in (partition, row) order, empty partitions and a lone partition are fine, tiles of 2048 records are crossed."""
    r = np.random.default_rng(sum(sizes) + len(sizes))
    parts, allk, allb, rid0 = [], [], [], 0
    for n in sizes:
        a, b = run_table(r, n, wide)
        t = pa.table({"rid": np.arange(rid0, rid0 + n, dtype=np.int64), "a": a, "b": b})
        parts.append(t.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in t.schema])))
        allk.append(a); allb.append(b); rid0 += n
    src = g.MemoryExec(parts)
    s = src.schema()
    order = [{"expr": col("a", s), "asc": True, "nulls_first": False}] + ([{"expr": col("b", s), "asc": True, "nulls_first": False}] if wide else [])
    A, Bv = np.concatenate(allk), np.concatenate(allb)
    want = np.lexsort((Bv, A)) if wide else np.argsort(A, kind="stable")
    for plan in (g.CoalesceTasksExec(src, list(range(len(sizes))), order_by=order), g.SortPreservingMergeExec(order, src)):
        got = np.asarray(g.plan.materialize(tc, plan.execute(0, tc)).to_arrow(tc.ctx).column("rid"))
        assert np.array_equal(got, want)
        native = g.NativePlan(plan, tc).execute(0).to_arrow()
        assert np.array_equal(np.asarray(native.column("rid")), want)
