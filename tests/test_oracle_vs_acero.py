"""CPU: the oracle against pyarrow Acero as a SECOND OPINION on integer-domain results (Acero is not
the reference; its decimal typing differs from DataFusion's, so comparisons use raw unscaled integers)."""
import decimal

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

from oracle import oracle_np as O


def col(n):
    return {"column": {"name": n}}


def _table(seed, n):
    r = np.random.default_rng(seed)
    return pa.table({
        "k": pa.array(r.integers(0, 50, n), type=pa.int64()),
        "g": pa.array(np.array(["A", "N", "R"])[r.integers(0, 3, n)]),
        "v": pa.array(r.integers(-10**6, 10**6, n), type=pa.int64()),
        "d": pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in r.integers(-10**7, 10**7, n)], type=pa.decimal128(15, 2)),
    })


def test_group_by_sum_count_min_max():
    t = _table(1, 5000)
    o = O.aggregate(O.Table.from_arrow(t), [(col("g"), "g"), (col("k"), "k")],
                    [{"fn": "SUM", "expr": col("v"), "name": "s"}, {"fn": "COUNT", "expr": col("v"), "name": "c"},
                     {"fn": "MIN", "expr": col("v"), "name": "mn"}, {"fn": "MAX", "expr": col("v"), "name": "mx"},
                     {"fn": "SUM", "expr": col("d"), "name": "sd"}], "Single")
    a = t.group_by(["g", "k"]).aggregate([("v", "sum"), ("v", "count"), ("v", "min"), ("v", "max"), ("d", "sum")])
    exp = sorted(zip(a["g"].to_pylist(), a["k"].to_pylist(), a["v_sum"].to_pylist(), a["v_count"].to_pylist(), a["v_min"].to_pylist(),
                     a["v_max"].to_pylist(), [int(x.scaleb(2)) for x in a["d_sum"].to_pylist()]))
    assert sorted(o.rows()) == exp


def test_inner_and_outer_join_pairs():
    l, r = _table(2, 800), _table(3, 1500)
    l = l.append_column("lid", pa.array(np.arange(l.num_rows)))
    r = r.append_column("rid", pa.array(np.arange(r.num_rows))).rename_columns(["rk", "rg", "rv", "rd", "rid"])
    ol, orr = O.Table.from_arrow(l), O.Table.from_arrow(r)
    for jt, how in (("Inner", "inner"), ("Left", "left outer"), ("Right", "right outer"), ("Full", "full outer"),
                    ("LeftSemi", "left semi"), ("LeftAnti", "left anti"), ("RightSemi", "right semi"), ("RightAnti", "right anti")):
        pairs = O.hash_join(ol, orr, [(col("k"), col("rk")), (col("g"), col("rg"))], jt)
        a = l.join(r, keys=["k", "g"], right_keys=["rk", "rg"], join_type=how)
        if jt.startswith("LeftS") or jt.startswith("LeftA"):
            assert sorted(i for i, _ in pairs) == sorted(a["lid"].to_pylist())
        elif jt.startswith("RightS") or jt.startswith("RightA"):
            assert sorted(j for _, j in pairs) == sorted(a["rid"].to_pylist())
        else:
            key = lambda p: (p[0] is None, p[0] or 0, p[1] is None, p[1] or 0)
            assert sorted(pairs, key=key) == sorted(zip(a["lid"].to_pylist(), a["rid"].to_pylist()), key=key)


def test_filter_and_sort():
    t = _table(4, 3000)
    ot = O.Table.from_arrow(t)
    pred = {"binary_expr": {"l": {"binary_expr": {"l": col("v"), "r": {"literal": {"type": "Int64", "value": 0}}, "op": ">"}},
                            "r": {"binary_expr": {"l": col("g"), "r": {"literal": {"type": "Utf8", "value": "N"}}, "op": "!="}}, "op": "AND"}}
    rows = O.filter_rows(ot, pred)
    m = pc.and_(pc.greater(t["v"], 0), pc.not_equal(t["g"], "N"))
    assert rows == [i for i, x in enumerate(m.to_pylist()) if x]
    spec = [{"expr": col("g"), "asc": True, "nulls_first": False}, {"expr": col("v"), "asc": False, "nulls_first": True}]
    perm = O.sort_perm(ot, spec)
    idx = pc.sort_indices(t, sort_keys=[("g", "ascending"), ("v", "descending")]).to_pylist()
    assert [(ot.col("g")[i], ot.col("v")[i]) for i in perm] == [(ot.col("g")[i], ot.col("v")[i]) for i in idx]


def test_decimal_expression_exactness():
    t = _table(5, 1000)
    ot = O.Table.from_arrow(t)
    one = {"literal": {"type": {"Decimal128": [20, 0]}, "value": "1"}}
    e = {"binary_expr": {"l": col("d"), "r": {"binary_expr": {"l": one, "r": col("d"), "op": "-"}}, "op": "*"}}
    ty, v = O.eval_expr(e, ot)
    assert ty == O.dec(38, 4)          # (15,2) * (23,2)
    d = [int(x.scaleb(2)) for x in t["d"].to_pylist()]
    assert v == [x * (100 - x) for x in d]
