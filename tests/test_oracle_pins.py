"""CPU: pin the oracle to every known answer the reference's own tests hold for this path
(SURVEY.md §8c): the ungrouped aggregates over alltypes_plain (ballista/client/src/context.rs:762-967)
and the q1 plan shape over the 10-row TPC-H tables (scheduler/src/planner.rs:376-392 -- shape only,
the reference asserts no values there).  Everything else is 'parity unpinned' and is cross-checked
against pyarrow Acero in test_oracle_vs_acero.py."""
import os

import pyarrow as pa

from oracle import oracle_np as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _alltypes():
    with pa.ipc.open_file(os.path.join(GOLD, "alltypes_plain.arrow")) as f:
        return O.Table.from_arrow(f.read_all().select(["id", "bigint_col", "double_col", "int_col", "tinyint_col"]))


def col(n):
    return {"column": {"name": n}}


def test_alltypes_plain_known_answers():
    t = _alltypes()
    assert t.col("id") == [4, 5, 6, 7, 2, 3, 0, 1]        # row order documented in the reference test file
    aggs = [{"fn": fn, "expr": col("id"), "name": fn} for fn in ("MIN", "MAX", "SUM", "AVG", "COUNT")]
    out = O.aggregate(t, [], aggs, "Single")
    # context.rs: MIN(test.id)=0  MAX=7  SUM=28  AVG=3.5  COUNT=8
    assert out.rows() == [(0, 7, 28, 3.5, 8)]
    assert out.types == ["Int32", "Int32", "Int64", "Float64", "Int64"]


def test_alltypes_plain_variance_family_known_answers():
    """context.rs:845-937: VAR / VAR_POP / VAR_SAMP / STDDEV / STDDEV_SAMP / COVAR / CORR, printed with Rust's
    shortest round-trip formatting, i.e. the exact f64."""
    t = _alltypes()
    one = lambda fn, **kw: O.aggregate(t, [], [dict({"fn": fn, "expr": col("id"), "name": "v"}, **kw)], "Single").rows()[0][0]
    assert repr(one("VARIANCE")) == "6.000000000000001"
    assert repr(one("VAR_SAMP")) == "6.000000000000001"
    assert repr(one("VARIANCE_POP")) == "5.250000000000001"
    assert repr(one("STDDEV")) == "2.4494897427831783"
    assert repr(one("STDDEV_SAMP")) == "2.4494897427831783"
    assert repr(one("COVARIANCE", expr2=col("tinyint_col"))) == "0.28571428571428586"
    assert repr(one("CORRELATION", expr2=col("tinyint_col"))) == "0.21821789023599245"


def test_variance_two_phase_merges_to_single_within_rounding():
    t = _alltypes()
    for fn, kw in (("VARIANCE", {}), ("STDDEV_POP", {}), ("COVARIANCE_POP", {"expr2": col("tinyint_col")}), ("CORRELATION", {"expr2": col("tinyint_col")})):
        aggs = [dict({"fn": fn, "expr": col("id"), "name": "v"}, **kw)]
        single = O.aggregate(t, [], aggs, "Single").rows()[0][0]
        part = O.aggregate(t, [(col("int_col"), "g")], aggs, "Partial")       # 2 groups -> 2 partial states
        final = O.aggregate(part.select(part.names[1:]) if hasattr(part, "select") else O.Table(part.names[1:], part.types[1:], part.cols[1:]), [], aggs, "Final").rows()[0][0]
        assert abs(final - single) <= 1e-12 * max(1.0, abs(single)), (fn, final, single)


def test_two_phase_equals_single():
    t = _alltypes()
    aggs = [{"fn": fn, "expr": col("bigint_col"), "name": fn} for fn in ("MIN", "MAX", "SUM", "AVG", "COUNT")]
    single = O.aggregate(t, [(col("int_col"), "g")], aggs, "Single")
    part = O.aggregate(t, [(col("int_col"), "g")], aggs, "Partial")
    assert part.names == ["g", "MIN[min]", "MAX[max]", "SUM[sum]", "AVG[count]", "AVG[sum]", "COUNT[count]"]
    final = O.aggregate(part, [(col("g"), "g")], aggs, "Final")
    assert sorted(final.rows()) == sorted(single.rows())


def _tbl(name, cols):
    rows = [l.rstrip("\n").split("|") for l in open(os.path.join(GOLD, "tpch10", name))]
    return [[r[i] for r in rows] for i in cols]


def test_q1_shape_on_reference_testdata():
    """q1-like stage from planner.rs:376-392 over the reference's lineitem testdata: the oracle's grouped
    decimal SUM equals plain Python arithmetic on the .tbl text."""
    import decimal
    ext, rf = [], []
    for part in ("lineitem.partition0.tbl", "lineitem.partition1.tbl"):
        e, r = _tbl(part, [5, 8])
        ext += e; rf += r
    t = O.Table(["l_extendedprice", "l_returnflag"], [O.dec(15, 2), "Utf8"], [[int(decimal.Decimal(x).scaleb(2)) for x in ext], rf])
    one = {"literal": {"type": "Int64", "value": 1}}
    out = O.aggregate(t, [(col("l_returnflag"), "l_returnflag")],
                      [{"fn": "SUM", "expr": {"binary_expr": {"l": col("l_extendedprice"), "r": one, "op": "*"}}, "name": "s"}], "Single")
    exp = {}
    for e, r in zip(ext, rf):
        exp[r] = exp.get(r, 0) + int(decimal.Decimal(e).scaleb(2))
    assert dict(out.rows()) == exp
    # l_extendedprice(15,2) * Int64 -> Decimal(15,2)*Decimal(20,0) = (36,2); SUM -> (38,2)
    assert out.types[1] == O.dec(38, 2)


def test_decimal_division_rules_by_hand():
    """Decimal `/` and `%` as arrow-arith 49 computes them (hand-worked): 1.00/3.00 -> 0.333333 (15,2)/(15,2) -> (21,6);
    -7.50 / 2 (Int32 -> Decimal(10,0)) -> -3.750000 (19,6); truncation toward zero; % keeps the dividend's sign; x/0 -> NULL."""
    from oracle import oracle_np as O
    import decimal
    D = decimal.Decimal
    t = O.Table(["a", "b", "i"], [{"Decimal128": [15, 2]}, {"Decimal128": [15, 2]}, "Int32"],
                [[100, -750, 200, -1, None], [300, 200, 0, 3, 5], [3, 2, 7, -2, 1]])
    from arrow_ballista_amd.expr import Operator as Op, binary, col
    sch = [{"name": n, "type": ty} for n, ty in zip(t.names, t.types)]
    ca, cb, ci = col("a", sch), col("b", sch), col("i", sch)
    ty, v = O.eval_expr(binary(ca, Op.Divide, cb), t)
    assert ty == {"Decimal128": [21, 6]} and v == [333333, -3750000, None, -333333, None]
    ty, v = O.eval_expr(binary(ca, Op.Divide, ci), t)
    assert ty == {"Decimal128": [19, 6]} and v == [333333, -3750000, 285714, 5000, None]
    ty, v = O.eval_expr(binary(ca, Op.Modulo, cb), t)
    assert ty == {"Decimal128": [15, 2]} and v == [100, -150, None, -1, None]
    ty, v = O.eval_expr(binary(ca, Op.Modulo, ci), t)
    assert ty == {"Decimal128": [12, 2]} and v == [100, -150, 200, -1, None]


def test_like_rules_by_hand():
    """LIKE as arrow-string 49 evaluates a scalar pattern (hand-worked): the four regex-free shapes, '_' = one character (not one
    byte), escapes, and the regex path's '.'-does-not-match-newline rule that only general patterns are subject to."""
    from oracle.oracle_np import like_match as L
    cases = [("abc", "abc", True), ("abc", "ab", False), ("abcdef", "abc%", True), ("abcdef", "%def", True), ("abcdef", "%cd%", True), ("abcdef", "a_c%f", True),
             ("abcdef", "a%c_e%", True), ("50%", "50\\%", True), ("50x", "50\\%", False), ("a_b", "a\\_b", True), ("axb", "a\\_b", False),
             ("a\nb", "a%b", False), ("a\nb", "a%", True), ("a\nb", "%b", True), ("a\nb", "%\n%", True), ("", "%", True), ("", "_", False), ("é", "_", True),
             ("日本語", "___", True), ("日本語", "__", False), ("special requests", "%special%requests%", True), ("abc", "___", True), ("abc", "____", False),
             ("a\\b", "a\\b", True), ("a.c", "a.c", True), ("abc", "a.c", False), ("a+c", "a+_", True)]
    for s_, p_, e_ in cases:
        assert L(s_, p_) == e_, (s_, p_)
