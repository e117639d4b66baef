"""Utf8 keys longer than 15 bytes (q10's c_name, q16's p_type) in the native executor: AggregateExec group
keys and HashJoinExec keys run over exact dictionary codes computed on the device (gpuq_utf8_intern) after the PACKED15 path has
refused them; strings come back with a take.  Checker: plain Python dictionaries over the same rows."""
import collections

import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd.expr import Operator as Op
from arrow_ballista_amd.expr import binary, col, lit
from test_gpu_native_plan import native_rows

pytestmark = pytest.mark.gpu


def names_table(n, seed, distinct, nulls=0.0, prefix="Customer#"):
    r = np.random.default_rng(seed)
    ids = r.integers(0, distinct, n)
    # 18-27 bytes; two families that share their first 15 bytes and their length, so that only a full comparison tells them apart
    s = [("%s%09d" % (prefix, i)) if i % 3 else ("%s%09d-%s" % (prefix, i, "x" * (i % 9))) for i in ids]
    mask = r.random(n) < nulls
    v = r.integers(-1000, 1000, n)
    t = pa.table({"name": pa.array(s, mask=mask if nulls else None), "short": pa.array(["k%d" % (i % 5) for i in ids]), "v": pa.array(v, type=pa.int64())})
    return t.cast(pa.schema([pa.field("name", pa.string(), nulls > 0), pa.field("short", pa.string(), False), pa.field("v", pa.int64(), False)]))


def rows_of(table):
    return [tuple(r.values()) for r in table.to_pylist()]


@pytest.mark.parametrize("n,distinct,nulls", [(1, 1, 0.0), (5000, 37, 0.0), (200_000, 50_000, 0.1), (300_000, 300_000, 0.0)])
@pytest.mark.parametrize("two_phase", [False, True])
def test_group_by_long_strings(tc, n, distinct, nulls, two_phase):
    t = names_table(n, 7 + n, distinct, nulls)
    src = g.MemoryExec([t])
    s = src.schema()
    pred = binary(col("v", s), Op.Gt, lit(-900, "Int64"))      # the aggregate reads the key through a fused filter
    inp = g.FilterExec(pred, src)
    groups = [(col("name", s), "name"), (col("short", s), "short")]
    aggs = [{"fn": "SUM", "expr": col("v", s), "name": "sv"}, {"fn": "COUNT", "expr": lit(1), "name": "c"}]
    if two_phase:
        part = g.AggregateExec("Partial", groups, aggs, inp)
        fs = part.schema()
        plan = g.AggregateExec("Final", [(col("name", fs), "name"), (col("short", fs), "short")], [dict(a, expr=None) for a in aggs], part)
    else:
        plan = g.AggregateExec("Single", groups, aggs, inp)
    got = rows_of(g.NativePlan(plan, tc).execute(0).to_arrow())
    exp = collections.defaultdict(lambda: [0, 0])
    for name, short, v in rows_of(t):
        if v > -900:
            e = exp[(name, short)]; e[0] += v; e[1] += 1
    assert sorted(got, key=repr) == sorted(((k[0], k[1], a, b) for k, (a, b) in exp.items()), key=repr)
    assert len(got) == len(exp)


@pytest.mark.parametrize("jt", ["Inner", "Left", "Right", "LeftAnti", "RightSemi"])
def test_join_on_long_strings(tc, jt):
    lt = names_table(4000, 1, 900, 0.1).append_column("lid", pa.array(np.arange(4000, dtype=np.int64)))
    rt = names_table(9000, 2, 1500, 0.1).append_column("rid", pa.array(np.arange(9000, dtype=np.int64)))       # 600 names the build side never saw
    rt = rt.rename_columns(["r_name", "r_short", "r_v", "rid"])
    L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
    ls, rs = L.schema(), R.schema()
    on = [(col("name", ls), col("r_name", rs)), (col("short", ls), col("r_short", rs))]       # a long and a short key together
    plan = g.HashJoinExec(L, R, on, None, jt, "CollectLeft", False)
    js = plan.schema()
    outs = [n for n in ("lid", "rid", "name", "r_name") if n in [f["name"] for f in js]]
    got = sorted(rows_of(g.NativePlan(g.ProjectionExec([(col(n, js), n) for n in outs], plan), tc).execute(0).to_arrow()), key=repr)
    lrows, rrows = rows_of(lt), rows_of(rt)
    idx = collections.defaultdict(list)
    for name, short, v, lid in lrows:
        if name is not None:
            idx[(name, short)].append(lid)
    pairs, lhit, rhit = [], set(), set()
    for name, short, v, rid in rrows:
        for lid in (idx.get((name, short), []) if name is not None else []):
            pairs.append((lid, rid)); lhit.add(lid); rhit.add(rid)
    ln = {r[3]: r[0] for r in lrows}; rn = {r[3]: r[0] for r in rrows}
    if jt == "Inner":
        exp = [(a, b, ln[a], rn[b]) for a, b in pairs]
    elif jt == "Left":
        exp = [(a, b, ln[a], rn[b]) for a, b in pairs] + [(r[3], None, r[0], None) for r in lrows if r[3] not in lhit]
    elif jt == "Right":
        exp = [(a, b, ln[a], rn[b]) for a, b in pairs] + [(None, r[3], None, r[0]) for r in rrows if r[3] not in rhit]
    elif jt == "LeftAnti":
        exp = [(r[3], r[0]) for r in lrows if r[3] not in lhit]
    else:
        exp = [(r[3], r[0]) for r in rrows if r[3] in rhit]
    assert got == sorted(exp, key=repr) and len(exp) > 0


def test_the_mirror_still_refuses_and_says_why(tc, mirror_layer):
    """The Python mirror (test-side) has no dictionary path: it fails loudly, it does not truncate keys."""
    t = names_table(100, 3, 10)
    src = g.MemoryExec([t])
    s = src.schema()
    plan = g.AggregateExec("Single", [(col("name", s), "name")], [{"fn": "COUNT", "expr": lit(1), "name": "c"}], src)
    with pytest.raises(g.GpuqError, match="15 bytes"):
        g.plan.materialize(tc, plan.execute(0, tc))


# ------------------------------------------------------------------ ORDER BY a long string (q2 / q21's s_name, q16's p_type, q18's c_name)
def sort_key(row, spec):
    """Python key for [(column index, asc, nulls_first)]: bytewise string order (Arrow's), NULLs by their flag."""
    out = []
    for ci, asc, nulls_first in spec:
        v = row[ci]
        if v is None:
            out.append((0 if nulls_first else 2, ()))
            continue
        b = v.encode() if isinstance(v, str) else v
        if not asc:
            b = tuple(255 - x for x in b) + (256,) if isinstance(b, bytes) else -b      # reversed bytewise order: a prefix sorts AFTER its extensions
        elif isinstance(b, bytes):
            b = tuple(b)
        out.append((1, b))
    return tuple(out)


@pytest.mark.parametrize("n,distinct,nulls", [(1, 1, 0.0), (3000, 40, 0.2), (150_000, 20_000, 0.05)])
def test_order_by_long_strings(tc, n, distinct, nulls):
    t = names_table(n, 11 + n, distinct, nulls).append_column("rid", pa.array(np.arange(n, dtype=np.int64)))
    src = g.MemoryExec([t])
    s = src.schema()
    rows = rows_of(t)
    cases = [
        ([("name", True, False)], [(0, True, False)]),
        ([("name", False, True)], [(0, False, True)]),
        ([("short", True, False), ("name", False, False), ("v", True, False)], [(1, True, False), (0, False, False), (2, True, False)]),
        ([("v", False, False), ("name", True, True)], [(2, False, False), (0, True, True)]),
    ]
    for spec, pyspec in cases:
        order = [{"expr": col(c, s), "asc": asc, "nulls_first": nf} for c, asc, nf in spec]
        got = rows_of(g.NativePlan(g.SortExec(order, src), tc).execute(0).to_arrow())
        want = sorted(rows, key=lambda r: sort_key(r, pyspec))      # Python's sort is stable, as SortExec is: ties keep input order
        assert got == want, spec


def test_order_by_strings_that_differ_only_beyond_a_piece_boundary_or_by_trailing_nul(tc):
    """Pieces are 14 bytes: values equal up to byte 14 / 28, a value that is a prefix of another, and "x" vs "x\\0" (zero padding
    alone would call them equal) must come out in bytewise order, under a fused filter and through a view (index vectors)."""
    base = "0123456789abcd"                       # 14 bytes = exactly one piece
    vals = [base + base + "b", base + base + "a", base + base, base + "z", base, base + "\x00", base + base + "a\x00", "", "\x00", "é" * 9, "é" * 8 + "e",
            base + base + base + "1", base + base + base + "0", None]
    r = np.random.default_rng(5)
    pick = r.integers(0, len(vals), 5000)
    t = pa.table({"s": pa.array([vals[i] for i in pick]), "v": pa.array(r.integers(0, 100, 5000), pa.int64()), "rid": pa.array(np.arange(5000, dtype=np.int64))})
    src = g.MemoryExec([t])
    s = src.schema()
    inp = g.FilterExec(binary(col("v", s), Op.Lt, lit(80, "Int64")), src)
    rows = [x for x in rows_of(t) if x[1] < 80]
    for asc in (True, False):
        got = rows_of(g.NativePlan(g.SortExec([{"expr": col("s", s), "asc": asc, "nulls_first": False}], inp), tc).execute(0).to_arrow())
        assert got == sorted(rows, key=lambda x: sort_key(x, [(0, asc, False)]))
    # top-k: fetch
    got = rows_of(g.NativePlan(g.SortExec([{"expr": col("s", s), "asc": True, "nulls_first": True}], inp, fetch=7), tc).execute(0).to_arrow())
    assert got == sorted(rows, key=lambda x: sort_key(x, [(0, True, True)]))[:7]


def test_the_mirror_refuses_a_long_sort_key(tc, mirror_layer):
    """SortExec used to order such rows by their first 15 bytes without a word; the mirror has no piece passes and must say so."""
    t = names_table(500, 3, 50)
    src = g.MemoryExec([t])
    s = src.schema()
    with pytest.raises(g.GpuqError, match="15 bytes"):
        g.plan.materialize(tc, g.SortExec([{"expr": col("name", s), "asc": True, "nulls_first": False}], src).execute(0, tc))
    big = names_table(70_000, 4, 5000)      # beyond the one-block sort: the min/max pass reports it
    bsrc = g.MemoryExec([big])
    with pytest.raises(g.GpuqError, match="15 bytes"):
        g.plan.materialize(tc, g.SortExec([{"expr": col("name", bsrc.schema()), "asc": True, "nulls_first": False}], bsrc).execute(0, tc))


# ------------------------------------------------------------------ = / != against a literal beyond 15 bytes (q19: l_shipinstruct = 'DELIVER IN PERSON')
@pytest.mark.parametrize("native", [False, True], ids=["mirror", "native"])
def test_equality_with_a_long_literal(tc, native, monkeypatch):
    if not native:
        monkeypatch.setenv("GPUQ_PLAN_LAYER", "mirror")
    vals = ["DELIVER IN PERSON", "TAKE BACK RETURN", "COLLECT COD", "NONE", "DELIVER IN PERSON!", "DELIVER IN PERSO", "100%_sure it is long", "100%xsure it is long", "100%_sure it is lon", None]
    r = np.random.default_rng(19)
    pick = r.integers(0, len(vals), 20_000)
    t = pa.table({"s": pa.array([vals[i] for i in pick]), "rid": pa.array(np.arange(20_000, dtype=np.int64))})
    src = g.MemoryExec([t])
    s = src.schema()
    rows = rows_of(t)

    def run(plan):
        return rows_of(g.NativePlan(plan, tc).execute(0).to_arrow()) if native else rows_of(g.plan.materialize(tc, plan.execute(0, tc)).to_arrow(tc.ctx))
    for literal in ("DELIVER IN PERSON", "100%_sure it is long", "a literal no row holds at all"):
        assert run(g.FilterExec(binary(col("s", s), Op.Eq, lit(literal)), src)) == [x for x in rows if x[0] == literal]
        assert run(g.FilterExec(binary(lit(literal), Op.NotEq, col("s", s)), src)) == [x for x in rows if x[0] is not None and x[0] != literal]      # NULL != x is NULL: dropped
        both = binary(binary(col("s", s), Op.Eq, lit(literal)), Op.Or, binary(col("rid", s), Op.Lt, lit(5, "Int64")))
        assert run(g.FilterExec(both, src)) == [x for x in rows if x[0] == literal or x[1] < 5]
    got = run(g.ProjectionExec([(binary(col("s", s), Op.Eq, lit("DELIVER IN PERSON")), "is_dip"), (col("rid", s), "rid")], src))
    assert got == [((None if x[0] is None else x[0] == "DELIVER IN PERSON"), x[1]) for x in rows]


# ------------------------------------------------------------------ comparisons of long strings (ordering, column against column)
def _cmp_table(n=6000, seed=5):
    r = np.random.default_rng(seed)
    stems = ["Customer#000000", "Customer#000001", "a", "", "exactly15bytes!", "exactly15bytes!!", "zebra crossing on a long and winding road", "δοκιμή utf-8 ✓ multibyte"]
    a = [stems[i] + ("%03d" % j if i < 2 else "") for i, j in zip(r.integers(0, len(stems), n), r.integers(0, 40, n))]
    b = [stems[i] + ("%03d" % j if i < 2 else "") for i, j in zip(r.integers(0, len(stems), n), r.integers(0, 40, n))]
    return pa.table({"a": pa.array(a, pa.string(), mask=r.random(n) < 0.1), "b": pa.array(b, pa.string(), mask=r.random(n) < 0.1),
                     "k": pa.array(r.integers(0, 7, n), pa.int64()), "id": pa.array(np.arange(n), pa.int64())})


def _py_cmp(op, x, y):
    if x is None or y is None:
        return None
    x, y = x.encode(), y.encode()
    return {"=": x == y, "!=": x != y, "<": x < y, "<=": x <= y, ">": x > y, ">=": x >= y}[op]


@pytest.mark.parametrize("op", ["=", "!=", "<", "<=", ">", ">="])
def test_long_string_comparisons_in_filter_and_projection(tc, op):
    """`a OP b` between two Utf8 columns and `a OP literal` (15, 16 and 41 bytes; also literal OP a) with values beyond 15 bytes: the
    register program refuses, the native executor lowers the comparison to gpuq_utf8_compare over the bytes (bytewise order, NULL in
    -> NULL out), in a FilterExec, in a computed projection, and twice more deferred."""
    from arrow_ballista_amd.expr import Operator as Op, binary
    t = _cmp_table()
    src = g.MemoryExec([t]); s = src.schema()
    lits = ["exactly15bytes!", "exactly15bytes!!", "Customer#000000020", "zebra crossing on a long and winding road"]
    exprs = [(binary(col("a", s), op, col("b", s)), "ab")] + [(binary(col("a", s), op, lit(v)), "l%d" % i) for i, v in enumerate(lits)] + [(binary(lit(lits[2]), op, col("b", s)), "rev")]
    got, _ = native_rows(tc, g.ProjectionExec(exprs + [(col("id", s), "id")], src))
    A, B = t["a"].to_pylist(), t["b"].to_pylist()
    exp = [tuple([_py_cmp(op, x, y)] + [_py_cmp(op, x, v) for v in lits] + [_py_cmp(op, lits[2], y), i]) for i, (x, y) in enumerate(zip(A, B))]
    assert got == exp
    got, _ = native_rows(tc, g.FilterExec(binary(col("a", s), op, col("b", s)), src))
    assert [r_[3] for r_ in got] == [i for i, (x, y) in enumerate(zip(A, B)) if _py_cmp(op, x, y)]


def test_long_string_comparison_in_a_filter_fused_into_an_aggregate_and_a_join(tc):
    from arrow_ballista_amd.expr import Operator as Op, binary
    t = _cmp_table()
    src = g.MemoryExec([t]); s = src.schema()
    f = g.FilterExec(binary(col("a", s), Op.Lt, lit("Customer#000001017")), src)
    got, _ = native_rows(tc, g.AggregateExec("Single", [(col("k", s), "k")], [{"fn": "COUNT", "expr": lit(1), "name": "c"}], f))
    A = t["a"].to_pylist(); K = t["k"].to_pylist()
    exp = collections.Counter(k for x, k in zip(A, K) if _py_cmp("<", x, "Customer#000001017"))
    assert sorted(got) == sorted(exp.items())
    small = pa.table({"jk": pa.array(np.arange(7), pa.int64())})
    j = g.HashJoinExec(g.MemoryExec([small]), f, [(col("jk", g.MemoryExec([small]).schema()), col("k", s))], None, "Inner", "CollectLeft", False)
    got, _ = native_rows(tc, j)
    assert sorted(r_[4] for r_ in got) == [i for i, x in enumerate(A) if _py_cmp("<", x, "Customer#000001017")]
