"""Parity of every operator on the device path against the oracle (oracle/oracle_np.py), on seeded
random Arrow tables: FilterExec, ProjectionExec, AggregateExec (all modes / strategies), HashJoinExec
(all join types), SortExec, hash repartition and the ShuffleWriterExec stage driver.  Bit-exact for
integer / decimal / string / index results; float64 SUM/AVG within 1e-9 relative (order of adds differs)."""
import decimal
import math
import os

import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd.expr import Operator as Op
from arrow_ballista_amd.expr import (and_, binary, case, cast, col, in_list, is_not_null, is_null, like, lit, negative, not_, or_)
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
D152 = ("Decimal128", 15, 2)


def rand_table(seed, n, nulls=0.0):
    r = np.random.default_rng(seed)

    def mask():
        return (r.random(n) < nulls) if nulls > 0 else None
    dec = [decimal.Decimal(int(v)).scaleb(-2) for v in r.integers(-10**9, 10**9, n)]
    segs = np.array(["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY", "", "x"])
    cols = {
        "k64": pa.array(r.integers(0, max(2, n // 3), n), type=pa.int64(), mask=mask()),
        "k32": pa.array(r.integers(-50, 50, n).astype(np.int32), type=pa.int32(), mask=mask()),
        "d": pa.array(r.integers(8000, 10500, n).astype(np.int32), type=pa.int32(), mask=mask()).cast(pa.date32()),
        "dec": pa.array(dec, type=pa.decimal128(15, 2), mask=mask()),
        "f": pa.array(r.normal(0, 1e3, n), type=pa.float64(), mask=mask()),
        "s": pa.array(segs[r.integers(0, len(segs), n)], type=pa.string(), mask=mask()),
        "flag": pa.array(np.array(["A", "N", "R"])[r.integers(0, 3, n)], type=pa.string(), mask=mask()),
        "b": pa.array(r.integers(0, 2, n).astype(bool), type=pa.bool_(), mask=mask()),
    }
    fields = [pa.field(k, v.type, nullable=nulls > 0) for k, v in cols.items()]
    return pa.Table.from_arrays(list(cols.values()), schema=pa.schema(fields))


def dev_rows(tc, table):
    import tpch_util as T
    return T.table_to_rows(tc, g.plan.materialize(tc, table))


def ora_rows(t):
    return [tuple(r) for r in t.rows()]


def norm(rows):
    return sorted(rows, key=lambda r: tuple((x is None, 0 if x is None else x) if not isinstance(x, float) else (x is None, O.total_order_key(x)) for x in r))


def close_rows(a, b, rel=1e-9):
    assert len(a) == len(b), (len(a), len(b))
    for ra, rb in zip(a, b):
        assert len(ra) == len(rb)
        for x, y in zip(ra, rb):
            if isinstance(x, float) or isinstance(y, float):
                if x is None or y is None:
                    assert x is None and y is None, (ra, rb)
                elif math.isnan(x) or math.isnan(y):
                    assert math.isnan(x) and math.isnan(y), (ra, rb)
                else:
                    assert abs(x - y) <= rel * max(1.0, abs(x), abs(y)), (ra, rb)
            else:
                assert x == y, (ra, rb)


# ------------------------------------------------------------------------------------ filter
PREDICATES = [
    lambda s: binary(col("k32", s), Op.Gt, lit(0, "Int32")),
    lambda s: and_(binary(col("d", s), Op.LtEq, lit(9500, "Date32")), binary(col("dec", s), Op.Gt, lit(0, D152))),
    lambda s: or_(binary(col("s", s), Op.Eq, lit("BUILDING")), binary(col("flag", s), Op.NotEq, lit("A"))),
    lambda s: binary(binary(col("dec", s), Op.Multiply, lit(3)), Op.Lt, binary(col("k64", s), Op.Plus, lit(100))),
    lambda s: in_list(col("s", s), [lit("MACHINERY"), lit("x"), lit("")]),
    lambda s: not_(in_list(col("k32", s), [lit(1, "Int32"), lit(2, "Int32")])),
    lambda s: or_(is_null(col("f", s)), binary(col("f", s), Op.Lt, lit(0.0))),
    lambda s: and_(col("b", s), is_not_null(col("k64", s))),
    lambda s: binary(col("f", s), Op.GtEq, cast(col("k32", s), "Float64")),
    lambda s: binary(case([(binary(col("k32", s), Op.Lt, lit(0, "Int32")), negative(col("k32", s)))], col("k32", s)), Op.Gt, lit(25, "Int32")),
]


@pytest.mark.parametrize("nulls", [0.0, 0.2])
@pytest.mark.parametrize("n", [0, 1, 64, 1000, 70_001])
def test_filter_exec(tc, n, nulls):
    t = rand_table(n + 1, n, nulls)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    for mk in PREDICATES:
        pred = mk(s)
        got = dev_rows(tc, g.FilterExec(pred, src).execute(0, tc))
        exp = ora_rows(ot.take(O.filter_rows(ot, pred)))
        close_rows(got, exp)        # order preserved: FilterExec keeps input order


def test_filter_of_filter_and_projection(tc):
    t = rand_table(5, 5000, 0.1)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    p1 = binary(col("k32", s), Op.Gt, lit(-10, "Int32"))
    p2 = binary(col("dec", s), Op.Lt, lit(500000, D152))
    exprs = [(binary(col("dec", s), Op.Multiply, binary(lit(1, ("Decimal128", 20, 0)), Op.Minus, col("dec", s))), "x"),
             (binary(col("k64", s), Op.Plus, cast(col("k32", s), "Int64")), "y"), (col("s", s), "s"), (col("f", s), "f"),
             (binary(col("f", s), Op.Multiply, lit(2.0)), "f2"), (binary(col("k32", s), Op.Gt, lit(3, "Int32")), "flagb")]
    plan = g.ProjectionExec(exprs, g.FilterExec(p2, g.CoalesceBatchesExec(g.FilterExec(p1, src))))
    got = dev_rows(tc, plan.execute(0, tc))
    sub = ot.take(O.filter_rows(ot, and_(p1, p2)))
    exp = ora_rows(O.project(sub, [e for e, _ in exprs], [n for _, n in exprs]))
    close_rows(got, exp)


def test_projection_decimal_types_and_cast(tc):
    t = rand_table(9, 3000, 0.1)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    exprs = [(binary(col("dec", s), Op.Plus, lit(1)), "a"), (binary(col("dec", s), Op.Minus, col("k64", s)), "b"),
             (binary(binary(col("dec", s), Op.Multiply, col("dec", s)), Op.Multiply, lit(7, D152)), "c"),
             (cast(col("dec", s), ("Decimal128", 20, 4)), "up"), (cast(col("dec", s), ("Decimal128", 12, 0)), "down"),
             (cast(col("dec", s), "Float64"), "tf"), (cast(col("k32", s), ("Decimal128", 12, 2)), "i2d"),
             (binary(col("k64", s), Op.Divide, cast(col("k32", s), "Int64")), "idiv"), (binary(col("k64", s), Op.Modulo, lit(7)), "imod")]
    plan = g.ProjectionExec(exprs, src)
    got = dev_rows(tc, plan.execute(0, tc))
    exp = ora_rows(O.project(ot, [e for e, _ in exprs], [n for _, n in exprs]))
    close_rows(got, exp)
    # declared result types follow the DataFusion rules
    ty = {f["name"]: f["type"] for f in plan.schema()}
    assert ty["a"] == {"Decimal128": [23, 2]} and ty["c"] == {"Decimal128": [38, 6]} and ty["up"] == {"Decimal128": [20, 4]}


def test_decimal_division_and_modulo(tc):
    """Decimal `/` and `%` (arrow-arith 49 decimal_op: Div -> scale s1+4, precision p1+4+s2; Rem -> scale max(s1,s2)); x/0 -> NULL."""
    t = rand_table(19, 4000, 0.1)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    exprs = [(binary(col("dec", s), Op.Divide, lit(7, D152)), "dq"), (binary(col("dec", s), Op.Divide, col("k32", s)), "di"),
             (binary(col("k64", s), Op.Divide, col("dec", s)), "id"), (binary(col("dec", s), Op.Modulo, lit(333, D152)), "dm"),
             (binary(col("dec", s), Op.Modulo, col("k32", s)), "dmi"),
             (binary(binary(lit(10000, ("Decimal128", 5, 2)), Op.Multiply, col("dec", s)), Op.Divide, cast(col("k64", s), ("Decimal128", 20, 2))), "q14")]
    plan = g.ProjectionExec(exprs, src)
    got = dev_rows(tc, plan.execute(0, tc))
    exp = ora_rows(O.project(ot, [e for e, _ in exprs], [n for _, n in exprs]))
    assert got == exp
    assert any(r[1] is None for r in got) and any(r[1] is not None and r[1] < 0 for r in got)
    ty = {f["name"]: f["type"] for f in plan.schema()}
    assert ty["dq"] == {"Decimal128": [21, 6]} and ty["di"] == {"Decimal128": [19, 6]} and ty["id"] == {"Decimal128": [26, 4]}
    assert ty["dm"] == {"Decimal128": [15, 2]} and ty["dmi"] == {"Decimal128": [12, 2]}


LIKE_PATTERNS = ["BUILDING", "%", "", "MACH%", "%ING", "%U%", "%special%requests%", "_UILDING", "B%G", "%_", "__", "a\\%b", "%\\_%", "50\\%", "é_%", "%o_ %",
                 "%fox%dog%", "x", "%x", "%%", "_%_", "%D__G", "line%two"]


def like_table(seed, n, nulls):
    r = np.random.default_rng(seed)
    words = np.array(["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY", "", "x", "a%b", "a_b", "50%", "50x", "é", "éa", "日本語", "ab",
                      "the quick brown fox jumps over the lazy dog", "special deposits requests", "no special pending requests here", "line one\nline two",
                      "GUILDING", "B\nG", "_", "%", "fox dog"])
    mask = (r.random(n) < nulls) if nulls > 0 else None
    return pa.table({"s": pa.array(words[r.integers(0, len(words), n)], type=pa.string(), mask=mask), "k": pa.array(r.integers(0, 50, n), type=pa.int64())})


@pytest.mark.parametrize("nulls", [0.0, 0.2])
def test_like_expr(tc, nulls):
    """LikeExpr (PhysicalLikeExprNode): every pattern shape of arrow's scalar `like` (equality, prefix, suffix, infix, general
    with '_' / inner '%' / escapes; multi-byte characters; the regex path's newline rule), LIKE and NOT LIKE, NULL operands,
    in FilterExec (alone, under another filter = through an index vector, fused under an aggregate) and in ProjectionExec."""
    t = like_table(77, 5000, nulls)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    for pat in LIKE_PATTERNS:
        for neg in (False, True):
            e = like(col("s", s), pat, negated=neg)
            got = dev_rows(tc, g.FilterExec(e, src).execute(0, tc))
            exp = ora_rows(ot.take(O.filter_rows(ot, e)))
            assert got == exp, (pat, neg)
    # as a projected value (NULL where the operand is NULL), next to other expressions
    exprs = [(like(col("s", s), "%requests%"), "l1"), (like(col("s", s), "_", negated=True), "l2"), (binary(col("k", s), Op.Plus, lit(1)), "k1"),
             (and_(like(col("s", s), "%o%"), binary(col("k", s), Op.Lt, lit(25))), "both")]
    plan = g.ProjectionExec(exprs, src)
    assert dev_rows(tc, plan.execute(0, tc)) == ora_rows(O.project(ot, [e for e, _ in exprs], [n for _, n in exprs]))
    assert [f["type"] for f in plan.schema()] == ["Boolean", "Boolean", "Int64", "Boolean"]
    # through a view (filter over a filter), then fused under an aggregate
    inner = g.FilterExec(binary(col("k", s), Op.Gt, lit(10)), src)
    outer = g.FilterExec(or_(like(col("s", s), "%ING"), like(col("s", s), "50_")), inner)
    keep = O.filter_rows(ot, and_(binary(col("k", s), Op.Gt, lit(10)), or_(like(col("s", s), "%ING"), like(col("s", s), "50_"))))
    assert dev_rows(tc, outer.execute(0, tc)) == ora_rows(ot.take(keep))
    aggs = [{"fn": "COUNT", "expr": lit(1), "name": "c"}]
    agg = g.AggregateExec("Single", [(col("s", s), "s")], aggs, outer)
    assert norm(dev_rows(tc, agg.execute(0, tc))) == norm(ora_rows(O.aggregate(ot.take(keep), [(col("s", s), "s")], aggs, "Single")))
    # a string produced by an operator (PACKED15) as the operand
    grouped = g.AggregateExec("Single", [(col("s", s), "s")], aggs, g.FilterExec(binary(col("k", s), Op.Lt, lit(40)), src))
    gs = grouped.schema()
    sub = O.aggregate(ot.take(O.filter_rows(ot, binary(col("k", s), Op.Lt, lit(40)))), [(col("s", s), "s")], aggs, "Single")
    short = [r for r in ora_rows(sub) if r[0] is None or len(r[0].encode()) <= 15]
    if len(short) == len(ora_rows(sub)):
        got = dev_rows(tc, g.FilterExec(like(col("s", gs), "%I%"), grouped).execute(0, tc))
        assert norm(got) == norm([r for r in short if r[0] is not None and "I" in r[0]])
    with pytest.raises(g.GpuqError) as ei:
        g.FilterExec(like(col("s", s), "abc", case_insensitive=True), src).execute(0, tc)
    assert ei.value.status == 3


# ------------------------------------------------------------------------------------ aggregate
def agg_cases(s):
    return [
        ([(col("flag", s), "flag")], [{"fn": "SUM", "expr": col("dec", s), "name": "sd"}, {"fn": "AVG", "expr": col("dec", s), "name": "ad"},
                                       {"fn": "COUNT", "expr": lit(1), "name": "c"}, {"fn": "COUNT", "expr": col("k64", s), "name": "ck"},
                                       {"fn": "MIN", "expr": col("k32", s), "name": "mn"}, {"fn": "MAX", "expr": col("d", s), "name": "mx"}]),
        ([(col("flag", s), "flag"), (col("s", s), "s")], [{"fn": "SUM", "expr": col("k64", s), "name": "sk"}, {"fn": "AVG", "expr": col("k32", s), "name": "ak"},
                                                           {"fn": "SUM", "expr": col("f", s), "name": "sf"}, {"fn": "MIN", "expr": col("f", s), "name": "mf"},
                                                           {"fn": "MAX", "expr": col("dec", s), "name": "xd"}]),
        ([], [{"fn": "SUM", "expr": binary(col("dec", s), Op.Multiply, col("dec", s)), "name": "s2"}, {"fn": "COUNT", "expr": lit(1), "name": "c"},
              {"fn": "AVG", "expr": col("f", s), "name": "af"}, {"fn": "MIN", "expr": col("dec", s), "name": "md"}]),
        ([(col("k64", s), "k64")], [{"fn": "SUM", "expr": col("dec", s), "name": "sd"}, {"fn": "COUNT", "expr": lit(1), "name": "c"},
                                    {"fn": "MAX", "expr": col("k32", s), "name": "mx"}, {"fn": "AVG", "expr": col("dec", s), "name": "ad"}]),
        ([(col("k64", s), "k64"), (col("d", s), "d"), (col("k32", s), "k32")], [{"fn": "SUM", "expr": col("dec", s), "name": "sd"}]),
    ]


@pytest.mark.parametrize("nulls", [0.0, 0.15])
@pytest.mark.parametrize("n", [0, 1, 500, 40_000])
def test_aggregate_single(tc, n, nulls):
    t = rand_table(100 + n, n, nulls)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    for groups, aggs in agg_cases(s):
        got = norm(dev_rows(tc, g.AggregateExec("Single", groups, aggs, src).execute(0, tc)))
        exp = norm(ora_rows(O.aggregate(ot, groups, aggs, "Single")))
        close_rows(got, exp)


@pytest.mark.parametrize("nulls", [0.0, 0.15])
@pytest.mark.parametrize("n", [0, 1, 700, 60_000])
def test_aggregate_radix_strategy(tc, n, nulls):
    """The radix-partitioned, LDS-resident aggregate (high-cardinality path) on every aggregate shape, forced at small sizes:
    several buckets, string and multi-column keys, NULL keys, all accumulator kinds, a fused predicate."""
    t = rand_table(4100 + n, n, nulls)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    pred = binary(col("k32", s), Op.Gt, lit(-30, "Int32"))
    for groups, aggs in agg_cases(s):
        if not groups:
            continue                       # ungrouped aggregates always take the LDS kernel
        plan = g.AggregateExec("Single", groups, aggs, g.FilterExec(pred, src), strategy="radix", expected_groups=max(1, n // 3))
        got = norm(dev_rows(tc, plan.execute(0, tc)))
        exp = norm(ora_rows(O.aggregate(ot, groups, aggs, "Single", predicate=pred)))
        close_rows(got, exp)
        close_rows(norm(native_rows_of(tc, plan)), exp)          # float sums: atomic order differs run to run


@pytest.mark.parametrize("nulls", [0.0, 0.15])
@pytest.mark.parametrize("n", [1, 700, 60_000, 300_000])
def test_aggregate_lds_strategy(tc, n, nulls):
    """Block-local pre-aggregation (k_agg_lds: every block folds its rows into an LDS table, the table's entries go to the
    global one), forced with strategy "lds" on every aggregate shape: few groups (everything stays in LDS), more groups than
    an LDS table holds (k64 at 300,000 rows: 100,000 groups -- blocks run full and send rows to the global table directly),
    string / multi-column / NULL keys, all accumulator kinds, a fused predicate.  Then the way "auto" reaches it: the second
    run of an operator whose first run produced few groups."""
    t = rand_table(5100 + n, n, nulls)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    pred = binary(col("k32", s), Op.Gt, lit(-30, "Int32"))
    for groups, aggs in agg_cases(s):
        if not groups:
            continue
        plan = g.AggregateExec("Single", groups, aggs, g.FilterExec(pred, src), strategy="lds")
        exp = norm(ora_rows(O.aggregate(ot, groups, aggs, "Single", predicate=pred)))
        close_rows(norm(dev_rows(tc, plan.execute(0, tc))), exp)
        close_rows(norm(native_rows_of(tc, plan)), exp)
    if n >= 300_000:
        groups, aggs = agg_cases(s)[1]                  # (flag, s): ~28 groups, more accumulators than the register-cached kernel takes
        plan = g.AggregateExec("Single", groups, aggs + [{"fn": "MAX", "expr": col("k64", s), "name": "xk"}, {"fn": "MIN", "expr": col("d", s), "name": "nd"}], src)
        exp = norm(ora_rows(O.aggregate(ot, groups, plan.aggr_expr, "Single")))
        for _ in range(3):                               # run 1: global table; from run 2 on the operator knows its group count
            close_rows(norm(dev_rows(tc, plan.execute(0, tc))), exp)


def native_rows_of(tc, plan):
    t = g.NativePlan(plan, tc).execute(0).to_arrow()
    cols = []
    for f, c in zip(t.schema, t.columns):
        if pa.types.is_decimal128(f.type):
            cols.append([None if v is None else int(v.scaleb(f.type.scale)) for v in c.to_pylist()])
        elif pa.types.is_date32(f.type):
            cols.append(c.cast(pa.int32()).to_pylist())
        else:
            cols.append(c.to_pylist())
    return list(zip(*cols)) if cols else []


@pytest.mark.parametrize("strategy", ["tiny", "hash"])
def test_aggregate_partial_final_and_strategies(tc, strategy):
    parts = [rand_table(7 + i, 20_000, 0.1) for i in range(3)]
    src = g.MemoryExec(parts)
    s = src.schema()
    ot_all = O.Table.from_arrow(pa.concat_tables(parts))
    cases = agg_cases(s)[:3] if strategy == "hash" else [agg_cases(s)[0], agg_cases(s)[2]]   # (flag, s) x 7 accumulators exceeds the LDS path
    for groups, aggs in cases:
        partial = g.AggregateExec("Partial", groups, aggs, g.FilterExec(is_not_null(col("k32", s)), src), strategy=strategy)
        states = [g.plan.materialize(tc, partial.execute(p, tc)).to_arrow(tc.ctx) for p in range(3)]
        merged = g.MemoryExec([pa.concat_tables(states)])
        fs = merged.schema()
        final = g.AggregateExec("FinalPartitioned", [(col(n, fs), n) for _, n in groups], [dict(a, expr=None) for a in aggs], merged)
        got = norm(dev_rows(tc, final.execute(0, tc)))
        exp = norm(ora_rows(O.aggregate(ot_all, groups, aggs, "Single", predicate=is_not_null(col("k32", s)))))
        close_rows(got, exp)
        # the oracle's own two-phase path gives the same
        ost = [O.aggregate(O.Table.from_arrow(p), groups, aggs, "Partial", predicate=is_not_null(col("k32", s))) for p in parts]
        cat = O.Table(ost[0].names, ost[0].types, [sum((o.cols[i] for o in ost), []) for i in range(len(ost[0].names))])
        exp2 = norm(ora_rows(O.aggregate(cat, [({"column": {"name": n}}, n) for _, n in groups], aggs, "Final")))
        close_rows(got, exp2)


def test_aggregate_kat_alltypes_plain(tc):
    """Known answers of the reference's own test-suite (ballista/client/src/context.rs:762-967) over
    ballista/client/testdata/alltypes_plain.parquet, committed as tests/golden/alltypes_plain.arrow."""
    with pa.ipc.open_file(os.path.join(os.path.dirname(__file__), "golden", "alltypes_plain.arrow")) as f:
        t = f.read_all().select(["id", "bigint_col", "double_col"])
    src = g.MemoryExec([t])
    s = src.schema()
    aggs = [{"fn": fn, "expr": col("id", s), "name": fn} for fn in ("MIN", "MAX", "SUM", "AVG", "COUNT")]
    got = dev_rows(tc, g.AggregateExec("Single", [], aggs, src).execute(0, tc))
    assert got == [(0, 7, 28, 3.5, 8)]


VAR_KATS = [("VARIANCE", None, "6.000000000000001"), ("VARIANCE_POP", None, "5.250000000000001"), ("STDDEV", None, "2.4494897427831783"),
            ("COVARIANCE", "tinyint_col", "0.28571428571428586"), ("CORRELATION", "tinyint_col", "0.21821789023599245")]


def test_variance_family_kat_alltypes_plain(tc):
    """context.rs:845-937.  The reference folds rows one at a time (Welford), the device adds order-free power
    sums: stated tolerance 4 ulp (|rel| <= 1e-15), not bit equality.  The oracle reproduces the KAT strings exactly."""
    with pa.ipc.open_file(os.path.join(os.path.dirname(__file__), "golden", "alltypes_plain.arrow")) as f:
        t = f.read_all().select(["id", "tinyint_col", "double_col"])
    src = g.MemoryExec([t])
    s = src.schema()
    aggs = [dict({"fn": fn, "expr": col("id", s), "name": fn}, **({"expr2": col(y, s)} if y else {})) for fn, y, _ in VAR_KATS]
    for lo in (0, 3):   # at most 12 accumulators per operator
        got = dev_rows(tc, g.AggregateExec("Single", [], aggs[lo:lo + 3], src).execute(0, tc))[0]
        for v, (_, _, kat) in zip(got, VAR_KATS[lo:lo + 3]):
            assert abs(v - float(kat)) <= 1e-15 * float(kat), (v, kat)


@pytest.mark.parametrize("nulls", [0.0, 0.2])
def test_variance_family_grouped_two_phase(tc, nulls):
    """Grouped VAR/STDDEV/COVAR/CORR over Float64, Int and Decimal arguments, Single and Partial->Final, vs the oracle's
    sequential accumulators.  Tolerance 1e-9 relative to the spread of the data (power sums cancel where Welford does not)."""
    parts = [rand_table(300 + i, 5_000, nulls) for i in range(2)]
    src = g.MemoryExec([pa.concat_tables(parts)])
    s = src.schema()
    ot = O.Table.from_arrow(pa.concat_tables(parts))
    groups = [(col("flag", s), "flag")]
    sets = [[{"fn": "VARIANCE", "expr": col("f", s), "name": "v"}, {"fn": "STDDEV_POP", "expr": col("k32", s), "name": "sp"},
             {"fn": "VARIANCE_POP", "expr": col("dec", s), "name": "vd"}],
            [{"fn": "COVARIANCE", "expr": col("f", s), "expr2": col("k32", s), "name": "cv"}, {"fn": "COVARIANCE_POP", "expr": col("k32", s), "expr2": col("f", s), "name": "cp"}],
            [{"fn": "CORRELATION", "expr": col("f", s), "expr2": col("dec", s), "name": "cr"}, {"fn": "STDDEV", "expr": col("f", s), "name": "sd"}]]
    for aggs in sets:
        exp = norm(ora_rows(O.aggregate(ot, groups, aggs, "Single")))
        got = norm(dev_rows(tc, g.AggregateExec("Single", groups, aggs, src).execute(0, tc)))
        close_rows(got, exp, rel=1e-9)
        psrc = g.MemoryExec(parts)
        partial = g.AggregateExec("Partial", groups, aggs, psrc)
        states = [g.plan.materialize(tc, partial.execute(p, tc)).to_arrow(tc.ctx) for p in range(2)]
        merged = g.MemoryExec([pa.concat_tables(states)])
        fs = merged.schema()
        final = g.AggregateExec("FinalPartitioned", [(col("flag", fs), "flag")], [dict(a, expr=None, expr2=None) for a in aggs], merged)
        close_rows(norm(dev_rows(tc, final.execute(0, tc))), exp, rel=1e-9)
        # state layout is the reference's: count, mean[, m2 | mean2, algoConst | m2_1, mean2, m2_2, algoConst]
        ost = O.aggregate(O.Table.from_arrow(parts[0]), groups, aggs, "Partial")
        assert states[0].schema.names == ost.names
        close_rows(norm([tuple(r.values()) for r in states[0].to_pylist()]), norm(ora_rows(ost)), rel=1e-9)


def test_variance_of_single_row_and_empty_groups(tc):
    t = pa.table({"g": pa.array([1, 2, 2], pa.int64()), "x": pa.array([5.0, None, None], pa.float64())})
    src = g.MemoryExec([t])
    s = src.schema()
    aggs = [{"fn": "VARIANCE", "expr": col("x", s), "name": "v"}, {"fn": "VARIANCE_POP", "expr": col("x", s), "name": "vp"},
            {"fn": "STDDEV", "expr": col("x", s), "name": "sd"}, {"fn": "CORRELATION", "expr": col("x", s), "expr2": col("x", s), "name": "c"}]
    got = norm(dev_rows(tc, g.AggregateExec("Single", [(col("g", s), "g")], aggs, src).execute(0, tc)))
    assert got == [(1, None, 0.0, None, 0.0), (2, None, None, None, None)]
    assert got == norm(ora_rows(O.aggregate(O.Table.from_arrow(t), [(col("g", s), "g")], aggs, "Single")))


# ------------------------------------------------------------------------------------ join
def pairs_of(tc, view, nl):
    """(left_row | None, right_row | None) pairs of a join view built over tables that carry a row-id column."""
    rows = dev_rows(tc, view)
    return rows


JOIN_TYPES = ["Inner", "Left", "Right", "Full", "LeftSemi", "LeftAnti", "RightSemi", "RightAnti"]


@pytest.mark.parametrize("nulls", [0.0, 0.2])
@pytest.mark.parametrize("jt", JOIN_TYPES)
def test_hash_join_types(tc, jt, nulls):
    nl, nr = 3000, 7000
    lt = rand_table(21, nl, nulls).append_column("lid", pa.array(np.arange(nl, dtype=np.int64)))
    rt = rand_table(22, nr, nulls).append_column("rid", pa.array(np.arange(nr, dtype=np.int64)))
    rt = rt.rename_columns([c if c == "rid" else "r_" + c for c in rt.schema.names])
    L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
    ls, rs = L.schema(), R.schema()
    ol, orr = O.Table.from_arrow(lt), O.Table.from_arrow(rt)
    for on in ([(col("k64", ls), col("r_k64", rs))],                       # duplicate keys on both sides
               [(col("k32", ls), col("r_k32", rs)), (col("flag", ls), col("r_flag", rs))],   # composite int + string key
               [(col("dec", ls), col("r_dec", rs))]):
        plan = g.HashJoinExec(L, R, on, None, jt, "CollectLeft", False)
        if jt in ("LeftSemi", "LeftAnti"):
            proj = g.ProjectionExec([(col("lid", ls), "lid")], plan)
            got = sorted(r[0] for r in dev_rows(tc, proj.execute(0, tc)))
            exp = sorted(i for i, _ in O.hash_join(ol, orr, on, jt))
        elif jt in ("RightSemi", "RightAnti"):
            proj = g.ProjectionExec([(col("rid", rs), "rid")], plan)
            got = sorted(r[0] for r in dev_rows(tc, proj.execute(0, tc)))
            exp = sorted(j for _, j in O.hash_join(ol, orr, on, jt))
        else:
            js = plan.schema()
            proj = g.ProjectionExec([(col("lid", js), "lid"), (col("rid", js), "rid"), (col("s", js), "s"), (col("r_dec", js), "r_dec")], plan)
            got = norm(dev_rows(tc, proj.execute(0, tc)))
            exp = norm([(i, j, None if i is None else ol.col("s")[i], None if j is None else orr.col("r_dec")[j]) for i, j in O.hash_join(ol, orr, on, jt)])
        assert got == exp, (jt, on)


def test_hash_join_null_equals_null_and_fused_filters(tc):
    nl, nr = 2000, 5000
    lt = rand_table(31, nl, 0.3).append_column("lid", pa.array(np.arange(nl, dtype=np.int64)))
    rt = rand_table(32, nr, 0.3).append_column("rid", pa.array(np.arange(nr, dtype=np.int64)))
    rt = rt.rename_columns([c if c == "rid" else "r_" + c for c in rt.schema.names])
    L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
    ls, rs = L.schema(), R.schema()
    ol, orr = O.Table.from_arrow(lt), O.Table.from_arrow(rt)
    lp = binary(col("k32", ls), Op.Gt, lit(-20, "Int32"))
    rp = binary(col("r_d", rs), Op.Lt, lit(10000, "Date32"))
    on = [(col("k32", ls), col("r_k32", rs))]
    plan = g.HashJoinExec(g.FilterExec(lp, L), g.CoalesceBatchesExec(g.FilterExec(rp, R)), on, None, "Inner", "CollectLeft", True)
    js = plan.schema()
    got = norm(dev_rows(tc, g.ProjectionExec([(col("lid", js), "lid"), (col("rid", js), "rid")], plan).execute(0, tc)))
    exp = norm(O.hash_join(ol, orr, on, "Inner", null_equals_null=True, left_pred=lp, right_pred=rp))
    assert got == exp
    # residual JoinFilter on an inner join
    jf = binary(col("dec", js), Op.Lt, col("r_dec", js))
    plan2 = g.HashJoinExec(L, R, on, jf, "Inner", "CollectLeft", False)
    got2 = norm(dev_rows(tc, g.ProjectionExec([(col("lid", js), "lid"), (col("rid", js), "rid")], plan2).execute(0, tc)))
    exp2 = norm([(i, j) for i, j in O.hash_join(ol, orr, on, "Inner")
                 if ol.col("dec")[i] is not None and orr.col("r_dec")[j] is not None and ol.col("dec")[i] < orr.col("r_dec")[j]])
    assert got2 == exp2


@pytest.mark.parametrize("jt", JOIN_TYPES[1:])
def test_hash_join_filter_on_outer_semi_anti(tc, jt, monkeypatch):
    """JoinFilter on the non-inner join types (q21 / q22 shape: semi / anti join with a residual predicate): a key match the
    filter rejects -- False or NULL -- is no match; sides' own predicates apply before the join."""
    nl, nr = 1500, 4000
    lt = rand_table(41, nl, 0.15).append_column("lid", pa.array(np.arange(nl, dtype=np.int64)))
    rt = rand_table(42, nr, 0.15).append_column("rid", pa.array(np.arange(nr, dtype=np.int64)))
    rt = rt.rename_columns([c if c == "rid" else "r_" + c for c in rt.schema.names])
    L, R = g.MemoryExec([lt]), g.MemoryExec([rt])
    ls, rs = L.schema(), R.schema()
    ol, orr = O.Table.from_arrow(lt), O.Table.from_arrow(rt)
    on = [(col("k32", ls), col("r_k32", rs))]
    both = ls + rs
    jf = and_(binary(col("dec", both), Op.Lt, col("r_dec", both)), binary(col("flag", both), Op.NotEq, col("r_flag", both)))

    def pf(i, j):
        a, b, c, d = ol.col("dec")[i], orr.col("r_dec")[j], ol.col("flag")[i], orr.col("r_flag")[j]
        x = None if a is None or b is None else a < b
        y = None if c is None or d is None else c != d
        return False if (x is False or y is False) else (None if (x is None or y is None) else True)
    lp = binary(col("k64", ls), Op.Gt, lit(20))
    rp = binary(col("r_d", rs), Op.Lt, lit(10200, "Date32"))
    for left, right, lpred, rpred in ((L, R, None, None), (g.FilterExec(lp, L), g.CoalesceBatchesExec(g.FilterExec(rp, R)), lp, rp)):
        plan = g.HashJoinExec(left, right, on, jf, jt, "CollectLeft", False)
        exp_pairs = O.hash_join(ol, orr, on, jt, left_pred=lpred, right_pred=rpred, pair_filter=pf)
        for runner in ("mirror", "native"):
            monkeypatch.setenv("GPUQ_PLAN_LAYER", runner)      # (node.execute() goes through the native executor by default)

            def rows(p):
                return dev_rows(tc, p.execute(0, tc)) if runner == "mirror" else __import__("test_gpu_native_plan").native_rows(tc, p)[0]
            if jt in ("LeftSemi", "LeftAnti"):
                got = sorted(r[0] for r in rows(g.ProjectionExec([(col("lid", ls), "lid")], plan)))
                exp = sorted(i for i, _ in exp_pairs)
            elif jt in ("RightSemi", "RightAnti"):
                got = sorted(r[0] for r in rows(g.ProjectionExec([(col("rid", rs), "rid")], plan)))
                exp = sorted(j for _, j in exp_pairs)
            else:
                js = plan.schema()
                got = norm(rows(g.ProjectionExec([(col("lid", js), "lid"), (col("rid", js), "rid"), (col("s", js), "s"), (col("r_f", js), "r_f")], plan)))
                exp = norm([(i, j, None if i is None else ol.col("s")[i], None if j is None else orr.col("r_f")[j]) for i, j in exp_pairs])
            assert got == exp, (jt, runner, lpred is not None)
            assert len(exp) > 0


def test_join_then_aggregate_then_sort_pipeline(tc):
    """q3-shaped: filter -> join -> join -> group-by -> sort, all through late-materialised views."""
    r = np.random.default_rng(3)
    nc, no, nl = 500, 3000, 12000
    cust = pa.table({"c_custkey": pa.array(np.arange(1, nc + 1, dtype=np.int64)), "c_seg": pa.array(np.array(["BUILDING", "MACHINERY", "HOUSEHOLD"])[r.integers(0, 3, nc)])})
    orders = pa.table({"o_orderkey": pa.array(np.arange(1, no + 1, dtype=np.int64) * 4), "o_custkey": pa.array(r.integers(1, nc + 1, no)),
                       "o_orderdate": pa.array(r.integers(9000, 9400, no).astype(np.int32)).cast(pa.date32()), "o_shippriority": pa.array(np.zeros(no, dtype=np.int32))})
    li = pa.table({"l_orderkey": pa.array(r.integers(1, no + 1, nl) * 4), "l_extendedprice": pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in r.integers(90000, 10**7, nl)], type=pa.decimal128(15, 2)),
                   "l_discount": pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in r.integers(0, 11, nl)], type=pa.decimal128(15, 2)),
                   "l_shipdate": pa.array(r.integers(9000, 9500, nl).astype(np.int32)).cast(pa.date32())})
    C, Od, Li = g.MemoryExec([cust]), g.MemoryExec([orders]), g.MemoryExec([li])
    cs, os_, lsch = C.schema(), Od.schema(), Li.schema()
    cut = 9200
    j1 = g.HashJoinExec(g.FilterExec(binary(col("c_seg", cs), Op.Eq, lit("BUILDING")), C),
                        g.FilterExec(binary(col("o_orderdate", os_), Op.Lt, lit(cut, "Date32")), Od),
                        [(col("c_custkey", cs), col("o_custkey", os_))], None, "Inner", "CollectLeft", False)
    j1s = j1.schema()
    j2 = g.HashJoinExec(j1, g.FilterExec(binary(col("l_shipdate", lsch), Op.Gt, lit(cut, "Date32")), Li),
                        [(col("o_orderkey", j1s), col("l_orderkey", lsch))], None, "Inner", "CollectLeft", False)
    j2s = j2.schema()
    rev = binary(col("l_extendedprice", j2s), Op.Multiply, binary(lit(1, ("Decimal128", 20, 0)), Op.Minus, col("l_discount", j2s)))
    agg = g.AggregateExec("Single", [(col("l_orderkey", j2s), "l_orderkey"), (col("o_orderdate", j2s), "o_orderdate"), (col("o_shippriority", j2s), "o_shippriority")],
                          [{"fn": "SUM", "expr": rev, "name": "revenue"}], j2)
    as_ = agg.schema()
    plan = g.SortExec([{"expr": col("revenue", as_), "asc": False, "nulls_first": True}, {"expr": col("o_orderdate", as_), "asc": True, "nulls_first": False}], agg)
    got = dev_rows(tc, plan.execute(0, tc))
    # oracle
    oc, oo, ol = O.Table.from_arrow(cust), O.Table.from_arrow(orders), O.Table.from_arrow(li)
    p1 = O.hash_join(oc, oo, [(col("c_custkey", cs), col("o_custkey", os_))], "Inner", left_pred=binary(col("c_seg", cs), Op.Eq, lit("BUILDING")),
                     right_pred=binary(col("o_orderdate", os_), Op.Lt, lit(cut, "Date32")))
    okeys = {}
    for _, j in p1:
        okeys.setdefault(oo.col("o_orderkey")[j], []).append(j)
    groups = {}
    for i in range(ol.n):
        if ol.col("l_shipdate")[i] > cut:
            for j in okeys.get(ol.col("l_orderkey")[i], []):
                k = (ol.col("l_orderkey")[i], oo.col("o_orderdate")[j], oo.col("o_shippriority")[j])
                groups[k] = groups.get(k, 0) + ol.col("l_extendedprice")[i] * (100 - ol.col("l_discount")[i])
    exp = sorted(((k[0], k[1], k[2], v) for k, v in groups.items()), key=lambda x: (-x[3], x[1]))
    assert len(got) == len(exp)
    assert [(r_[3], r_[1]) for r_ in got] == [(e[3], e[1]) for e in exp]       # ORDER BY columns in order
    assert norm(got) == norm(exp)


# ------------------------------------------------------------------------------------ sort
@pytest.mark.parametrize("nulls", [0.0, 0.2])
@pytest.mark.parametrize("n", [0, 1, 2, 257, 10_000, 300_000])
def test_sort_exec(tc, n, nulls):
    t = rand_table(40 + n, n, nulls)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    for spec in ([{"expr": col("k32", s), "asc": True, "nulls_first": False}],
                 [{"expr": col("dec", s), "asc": False, "nulls_first": True}, {"expr": col("d", s), "asc": True, "nulls_first": False}],
                 [{"expr": col("flag", s), "asc": True, "nulls_first": True}, {"expr": col("s", s), "asc": False, "nulls_first": False}, {"expr": col("k64", s), "asc": True, "nulls_first": False}],
                 [{"expr": col("f", s), "asc": False, "nulls_first": False}],
                 [{"expr": binary(col("k64", s), Op.Minus, cast(col("k32", s), "Int64")), "asc": True, "nulls_first": False}]):
        plan = g.SortExec(spec, g.ProjectionExec([(col(c, s), c) for c in ("k64", "k32", "d", "dec", "f", "s", "flag")], src))
        got = dev_rows(tc, plan.execute(0, tc))
        keys = O.sort_keys(ot, spec)
        # sortedness on the oracle's key function + same multiset of rows
        got_t = O.Table(ot.names[:7], ot.types[:7], [list(c) for c in zip(*got)] if got else [[] for _ in range(7)])
        gk = O.sort_keys(got_t, spec)
        assert all(gk[i] <= gk[i + 1] for i in range(len(gk) - 1))
        exp = [tuple(c[i] for c in ot.cols[:7]) for i in O.sort_perm(ot, spec)]
        close_rows(norm(got), norm(exp))
        assert [k for k in gk] == sorted(keys)


def test_sort_fetch(tc):
    t = rand_table(77, 5000, 0.0)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    spec = [{"expr": col("dec", s), "asc": False, "nulls_first": True}, {"expr": col("k64", s), "asc": True, "nulls_first": False}]
    got = dev_rows(tc, g.SortExec(spec, g.ProjectionExec([(col("dec", s), "dec"), (col("k64", s), "k64")], src), fetch=10).execute(0, tc))
    exp = [(ot.col("dec")[i], ot.col("k64")[i]) for i in O.sort_perm(ot, spec)[:10]]
    assert got == exp


# ------------------------------------------------------------------------------------ partition / shuffle
@pytest.mark.parametrize("nparts", [1, 2, 16, 200, 1000])
def test_hash_partition(tc, nparts):
    t = rand_table(55, 20_000, 0.1)
    ot = O.Table.from_arrow(t)
    src = g.MemoryExec([t])
    s = src.schema()
    for keys in ([col("k64", s)], [col("k32", s), col("flag", s)], [col("dec", s)]):
        views = g.RepartitionExec(src, keys, nparts).execute_all(0, tc)
        pid = O.hash_partition(ot, keys, nparts)
        assert sum(v.num_rows for v in views) == t.num_rows
        for p, v in enumerate(views):
            got = dev_rows(tc, v)
            exp = [tuple(c[i] for c in ot.cols) for i in range(ot.n) if pid[i] == p]    # input order kept inside a partition
            close_rows(got, exp)


def test_shuffle_writer_round_trip(tc, tmp_path):
    """Stage driver: files in the reference's layout, LZ4-framed Arrow IPC streams, exact row counts
    (readers drop locations with num_rows == 0, shuffle_reader.rs:251)."""
    parts = [rand_table(60 + i, 5000, 0.1) for i in range(2)]
    src = g.MemoryExec(parts)
    s = src.schema()
    plan = g.ShuffleWriterExec("job1", 3, g.FilterExec(binary(col("k32", s), Op.Gt, lit(0, "Int32")), src), "", ([col("k64", s)], 4))
    stage = g.DefaultExecutionEngine().create_query_stage_exec("job1", 3, plan, str(tmp_path))
    out = stage.execute_query_stage([0, 1], tc)
    ot = O.Table.from_arrow(pa.concat_tables(parts))
    keep = O.filter_rows(ot, binary(col("k32", s), Op.Gt, lit(0, "Int32")))
    assert sum(o["num_rows"] for o in out) == len(keep)
    seen = []
    for o in out:
        assert o["path"].startswith(os.path.join(str(tmp_path), "job1", "3", str(o["partition_id"])))
        with pa.OSFile(o["path"], "rb") as f:
            rd = pa.ipc.open_stream(f).read_all()
        assert rd.num_rows == o["num_rows"] and o["num_rows"] > 0
        pid = O.hash_partition(O.Table.from_arrow(rd), [col("k64", s)], 4)
        assert set(pid) == {o["partition_id"]}
        seen += ora_rows(O.Table.from_arrow(rd))
    close_rows(norm(seen), norm(ora_rows(ot.take(keep))))
    with pytest.raises(g.GpuqError):
        g.DefaultExecutionEngine().create_query_stage_exec("j", 1, src, str(tmp_path))
    assert stage.collect_plan_metrics()[0]["output_rows"] > 0          # the child plan's metrics (execution_engine.rs:130-132)
    m0 = stage.shuffle_writer.metrics.as_dict()                          # ShuffleWriteMetrics, shuffle_writer.rs:139-160
    assert m0["output_rows"] > 0 and m0["input_rows"] >= m0["output_rows"] and m0["write_time"] > 0 and m0["repart_time"] > 0


# ------------------------------------------------------------------------------------ fan-in / merge / limit / union
@pytest.mark.parametrize("nulls", [0.0, 0.2])
def test_coalesce_tasks_unordered_and_ordered(tc, nulls):
    """CoalesceTasksExec (coalesce_tasks.rs:130-229): unordered = concatenation of the listed partitions; with order_by =
    k-way merge of ALL input partitions.  Sizes are chosen so that piece boundaries fall inside bitmap words."""
    parts = [rand_table(500 + i, n, nulls) for i, n in enumerate([1000, 0, 77, 4099, 1])]
    src = g.MemoryExec(parts)
    s = src.schema()
    ots = [O.Table.from_arrow(p) for p in parts]
    got = dev_rows(tc, g.CoalesceTasksExec(src, [0, 1, 2, 3, 4]).execute(0, tc))
    exp = sum((ora_rows(o) for o in ots), [])
    close_rows(got, exp, rel=0.0)
    got = dev_rows(tc, g.CoalesceTasksExec(src, [3, 2]).execute(0, tc))
    close_rows(got, ora_rows(ots[3]) + ora_rows(ots[2]), rel=0.0)
    assert dev_rows(tc, g.CoalesceTasksExec(src, [2]).execute(0, tc)) == dev_rows(tc, src.execute(2, tc))
    assert dev_rows(tc, g.CoalescePartitionsExec(src).execute(0, tc)) == dev_rows(tc, g.CoalesceTasksExec(src, [0, 1, 2, 3, 4]).execute(0, tc))
    # ordered: every partition sorted by (flag asc, dec desc), then merged; ties resolved by (partition, row) order
    order = [{"expr": col("flag", s), "asc": True, "nulls_first": False}, {"expr": col("dec", s), "asc": False, "nulls_first": True}]
    sorted_src = g.SortExec(order, src, preserve_partitioning=True)
    merged = g.CoalesceTasksExec(sorted_src, [0, 3], order_by=order)      # the reference merges all partitions here
    got = dev_rows(tc, merged.execute(0, tc))
    cat = O.Table(ots[0].names, ots[0].types, [sum((o.cols[i] for o in ots), []) for i in range(len(ots[0].names))])
    perm = O.sort_perm(cat, order)
    # concatenation of per-partition sorted runs followed by a stable sort == stable sort of the sorted runs
    runs = []
    for o in ots:
        pi = O.sort_perm(o, order)
        runs += [tuple(o.cols[c][i] for c in range(len(o.names))) for i in pi]
    rt = O.Table(ots[0].names, ots[0].types, [[r[c] for r in runs] for c in range(len(ots[0].names))])
    exp = [runs[i] for i in O.sort_perm(rt, order)]
    close_rows(got, exp, rel=0.0)
    assert len(perm) == len(got)
    got2 = dev_rows(tc, g.SortPreservingMergeExec(order, sorted_src, fetch=10).execute(0, tc))
    close_rows(got2, exp[:10], rel=0.0)


def test_limits_and_union_kat(tc):
    t = rand_table(9, 1000, 0.1)
    src = g.MemoryExec([t, rand_table(10, 5, 0.0)])
    rows0 = dev_rows(tc, src.execute(0, tc))
    assert dev_rows(tc, g.LocalLimitExec(src, 7).execute(0, tc)) == rows0[:7]
    assert len(dev_rows(tc, g.LocalLimitExec(src, 7).execute(1, tc))) == 5
    one = g.CoalescePartitionsExec(src)
    assert dev_rows(tc, g.GlobalLimitExec(one, skip=998, fetch=4).execute(0, tc)) == (rows0 + dev_rows(tc, src.execute(1, tc)))[998:1002]
    assert dev_rows(tc, g.GlobalLimitExec(one, skip=2000, fetch=4).execute(0, tc)) == []
    # over a view (filter output) as well
    s = src.schema()
    f = g.FilterExec(binary(col("k32", s), Op.Gt, lit(0, "Int32")), src)
    fr = dev_rows(tc, f.execute(0, tc))
    assert dev_rows(tc, g.GlobalLimitExec(g.CoalesceTasksExec(f, [0]), skip=3, fetch=5).execute(0, tc)) == fr[3:8]
    # context.rs:691-733: "SELECT 1 as NUMBER union SELECT 1 as NUMBER" -> one row; "union all" -> two rows
    a = pa.table({"NUMBER": pa.array([1], pa.int64())})
    u = g.UnionExec([g.MemoryExec([a]), g.MemoryExec([a])])
    assert u.output_partition_count() == 2
    assert dev_rows(tc, g.CoalescePartitionsExec(u).execute(0, tc)) == [(1,), (1,)]
    us = u.schema()
    dedup = g.AggregateExec("Single", [(col("NUMBER", us), "NUMBER")], [], g.CoalescePartitionsExec(u))
    assert dev_rows(tc, dedup.execute(0, tc)) == [(1,)]


# ------------------------------------------------------------------------------------ run-time failures are loud
def test_runtime_limits_fail_loudly(tc, mirror_layer):
    """What the device path cannot do is reported as an error (GPUQ_ERR_UNSUPPORTED / CAPACITY), never answered wrongly or
    by a fallback: a Utf8 value longer than 15 bytes reaching a comparison or a group key, more groups than the LDS
    strategy was asked to hold."""
    t = pa.table({"s": pa.array(["short", "exactly15bytes!", "a string value longer than fifteen bytes", None]), "v": pa.array([1, 2, 3, 4], pa.int64())})
    src = g.MemoryExec([t])
    s = src.schema()
    with pytest.raises(g.GpuqError) as e:
        dev_rows(tc, g.FilterExec(binary(col("s", s), Op.Lt, lit("short")), src).execute(0, tc))
    assert e.value.status == 3 and "15 bytes" in str(e.value)
    with pytest.raises(g.GpuqError) as e:      # the same comparison as a computed projection (asynchronous call: the executors read the status word)
        dev_rows(tc, g.ProjectionExec([(binary(col("s", s), Op.Lt, lit("short")), "b"), (col("v", s), "v")], src).execute(0, tc))
    assert e.value.status == 3 and "15 bytes" in str(e.value)
    # (the native executor answers this one since round 3: after the refusal it lowers the comparison to a kernel over the bytes, see
    # tests/test_gpu_long_string_keys.py; what it still refuses is a string-VALUED result beyond 15 bytes, e.g. a substring reaching past byte 15)
    r = g.NativePlan(g.ProjectionExec([(binary(col("s", s), Op.Lt, lit("short")), "b"), (col("v", s), "v")], src), tc).execute(0).to_arrow()
    assert r["b"].to_pylist() == [False, True, True, None]
    from arrow_ballista_amd.expr import substr
    with pytest.raises(g.GpuqError) as e:
        g.NativePlan(g.ProjectionExec([(substr(col("s", s), 14, 5), "b"), (col("v", s), "v")], src), tc).execute(0)
    assert e.value.status == 3 and "15 bytes" in str(e.value)
    with pytest.raises(g.GpuqError) as e:      # two columns: equal prefixes and lengths would compare equal
        dev_rows(tc, g.FilterExec(binary(col("s", s), Op.Eq, col("s", s)), src).execute(0, tc))
    assert e.value.status == 3
    # (in)equality with a literal and IS NULL are exact at any length (the packed form carries the true length): no refusal
    assert dev_rows(tc, g.FilterExec(binary(col("s", s), Op.Eq, lit("short")), src).execute(0, tc)) == [("short", 1)]
    assert dev_rows(tc, g.FilterExec(binary(col("s", s), Op.NotEq, lit("exactly15bytes!")), src).execute(0, tc)) == [("short", 1), ("a string value longer than fifteen bytes", 3)]
    assert dev_rows(tc, g.FilterExec(binary(binary(col("s", s), Op.Eq, lit("a string value")), Op.Or, {"is_null_expr": {"expr": col("s", s)}}), src).execute(0, tc)) == [(None, 4)]
    with pytest.raises(g.GpuqError) as e:
        dev_rows(tc, g.AggregateExec("Single", [(col("s", s), "s")], [{"fn": "SUM", "expr": col("v", s), "name": "x"}], src).execute(0, tc))
    assert e.value.status == 3
    # as a carried (never compared) payload the long value is fine: Arrow-layout strings are taken as they are
    assert dev_rows(tc, g.FilterExec(binary(col("v", s), Op.Gt, lit(1)), src).execute(0, tc)) == \
        [("exactly15bytes!", 2), ("a string value longer than fifteen bytes", 3), (None, 4)]
    # ... also through a fan-in of partitions (offsets re-based, bytes joined)
    both = dev_rows(tc, g.CoalescePartitionsExec(g.MemoryExec([t, t.slice(1, 2)])).execute(0, tc))
    assert both == [("short", 1), ("exactly15bytes!", 2), ("a string value longer than fifteen bytes", 3), (None, 4), ("exactly15bytes!", 2), ("a string value longer than fifteen bytes", 3)]
    # and after the failure the operators keep working (flags are reset)
    ok = pa.table({"s": pa.array(["a", "b", "a"]), "v": pa.array([1, 2, 3], pa.int64())})
    osrc = g.MemoryExec([ok])
    os_ = osrc.schema()
    assert sorted(dev_rows(tc, g.AggregateExec("Single", [(col("s", os_), "s")], [{"fn": "SUM", "expr": col("v", os_), "name": "x"}], osrc).execute(0, tc))) == [("a", 4), ("b", 2)]
    big = rand_table(3, 5000, 0.0)
    bs = g.MemoryExec([big]).schema()
    with pytest.raises(g.GpuqError) as e:
        g.AggregateExec("Single", [(col("k64", bs), "k")], [{"fn": "COUNT", "expr": lit(1), "name": "c"}], g.MemoryExec([big]), strategy="tiny").execute(0, tc)
    assert e.value.status == 4
