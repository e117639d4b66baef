"""Builders for the PhysicalExpr JSON mirror (datafusion.proto:1142-1180 PhysicalExprNode).

Names follow datafusion::physical_plan::expressions (Column, Literal, BinaryExpr, CastExpr, ...),
so plans in tests read like the reference's (`binary(col("a", schema), Operator.Plus, lit(1))`).
"""
import copy

# Operator display strings as they travel in PhysicalBinaryExprNode.op (datafusion.proto:1228-1232)
class Operator:
    Plus, Minus, Multiply, Divide, Modulo = "+", "-", "*", "/", "%"
    Eq, NotEq, Lt, LtEq, Gt, GtEq = "=", "!=", "<", "<=", ">", ">="
    And, Or = "AND", "OR"


def _type_json(t):
    if isinstance(t, (list, tuple)) and t[0] == "Decimal128":
        return {"Decimal128": [int(t[1]), int(t[2])]}
    if isinstance(t, (list, tuple)) and t[0] == "Timestamp":      # ("Timestamp", unit[, tz])
        return {"Timestamp": [t[1], t[2] if len(t) > 2 else None]}
    return t


def col(name, schema=None, index=None):
    """Column reference by name (resolved against `schema`: a list of field dicts) or explicit index."""
    if index is None and schema is not None:
        names = [f["name"] for f in schema]
        if name not in names:
            raise KeyError("column '%s' not in schema %s" % (name, names))
        index = names.index(name)
    d = {"name": name}
    if index is not None:
        d["index"] = int(index)
    return {"column": d}


def lit(value, type=None):
    """Literal.  Python int -> Int64, float -> Float64, str -> Utf8, bool -> Boolean unless `type` says otherwise.
    Decimal128: lit(unscaled_int, ("Decimal128", p, s)).  Date32: lit(days, "Date32").  Timestamp: lit(count, ("Timestamp", "Microsecond"))."""
    if type is None:
        if isinstance(value, bool):
            type = "Boolean"
        elif isinstance(value, int):
            type = "Int64"
        elif isinstance(value, float):
            type = "Float64"
        elif isinstance(value, str):
            type = "Utf8"
        elif value is None:
            type = "Null"
    t = _type_json(type)
    if value is None:
        return {"literal": {"type": t, "value": None}}
    if t in ("Utf8", "Boolean", "Float64", "Float32"):
        return {"literal": {"type": t, "value": value}}
    return {"literal": {"type": t, "value": str(int(value))}}


def binary(l, op, r):
    return {"binary_expr": {"l": l, "r": r, "op": op}}


def and_(*es):
    out = es[0]
    for e in es[1:]:
        out = binary(out, Operator.And, e)
    return out


def or_(*es):
    out = es[0]
    for e in es[1:]:
        out = binary(out, Operator.Or, e)
    return out


def cast(e, type):
    return {"cast": {"expr": e, "arrow_type": _type_json(type)}}


def try_cast(e, type):
    return {"try_cast": {"expr": e, "arrow_type": _type_json(type)}}


def not_(e):
    return {"not_expr": {"expr": e}}


def is_null(e):
    return {"is_null_expr": {"expr": e}}


def is_not_null(e):
    return {"is_not_null_expr": {"expr": e}}


def negative(e):
    return {"negative": {"expr": e}}


def in_list(e, values, negated=False):
    return {"in_list": {"expr": e, "list": list(values), "negated": bool(negated)}}


def like(e, pattern, negated=False, case_insensitive=False):
    """PhysicalLikeExprNode (datafusion.proto:1240-1245): expr [NOT] LIKE pattern; pattern is a Utf8 literal."""
    return {"like_expr": {"negated": bool(negated), "case_insensitive": bool(case_insensitive), "expr": e,
                          "pattern": pattern if isinstance(pattern, dict) else lit(pattern)}}


def scalar_function(name, args):
    """PhysicalScalarFunctionNode (datafusion.proto): built on the device are date_part('YEAR' | 'MONTH' | 'DAY', Date32) -> Float64 and
    substr(Utf8, start [, length]) with literal bounds (ASCII)."""
    return {"scalar_function": {"name": name, "args": list(args)}}


def date_part(part, e):
    return scalar_function("date_part", [lit(part), e])


def substr(e, start, length=None):
    return scalar_function("substr", [e, lit(int(start), "Int64")] + ([lit(int(length), "Int64")] if length is not None else []))


def case(when_then, else_expr=None, expr=None):
    return {"case_": {"expr": expr, "when_then_expr": [{"when_expr": w, "then_expr": t} for w, t in when_then],
                      "else_expr": else_expr}}


# ---------------------------------------------------------------- tree utilities (host-side planning)
def columns_of(e, acc=None):
    """Set of column names referenced by an expression."""
    acc = set() if acc is None else acc
    if isinstance(e, dict):
        if "column" in e and isinstance(e["column"], dict) and "name" in e["column"]:
            acc.add(e["column"]["name"])
        else:
            for v in e.values():
                columns_of(v, acc)
    elif isinstance(e, list):
        for v in e:
            columns_of(v, acc)
    return acc


def rewrite_columns(e, fn):
    """Return a copy of `e` with every column node replaced by fn(column_dict)."""
    if isinstance(e, dict):
        if "column" in e and isinstance(e["column"], dict) and "name" in e["column"] and len(e) == 1:
            return fn(e["column"])
        return {k: rewrite_columns(v, fn) for k, v in e.items()}
    if isinstance(e, list):
        return [rewrite_columns(v, fn) for v in e]
    return copy.deepcopy(e)


def inline_projection(e, proj):
    """Substitute column references by the expressions of a ProjectionExec below (proj: {name: expr})."""
    return rewrite_columns(e, lambda c: copy.deepcopy(proj[c["name"]]) if c["name"] in proj else {"column": dict(c)})


def rebind(e, schema):
    """Re-resolve column indices by name against `schema` (list of field dicts)."""
    names = [f["name"] for f in schema]

    def fix(c):
        if c["name"] not in names:
            raise KeyError("column '%s' not in schema %s" % (c["name"], names))
        return {"column": {"name": c["name"], "index": names.index(c["name"])}}
    return rewrite_columns(e, fix)


def has_like(e):
    if isinstance(e, dict):
        return "like_expr" in e or any(has_like(v) for v in e.values())
    if isinstance(e, list):
        return any(has_like(v) for v in e)
    return False


def rewrite_like(e, fn):
    """Copy of `e` with every like_expr node replaced by fn(node_body)."""
    if isinstance(e, dict):
        if "like_expr" in e and len(e) == 1:
            return fn(e["like_expr"])
        return {k: rewrite_like(v, fn) for k, v in e.items()}
    if isinstance(e, list):
        return [rewrite_like(v, fn) for v in e]
    return copy.deepcopy(e)


def lower_long_string_eq(e):
    """`column = 'literal'` / `!=` with a Utf8 literal beyond the 15 bytes a register holds becomes LIKE without wildcards (the
    literal's % and _ escaped): LIKE runs over the Arrow-layout bytes at any length.  Same rewrite as plan_exec.cpp's."""
    if isinstance(e, dict):
        b = e.get("binary_expr") if len(e) == 1 else None
        if isinstance(b, dict) and b.get("op") in ("=", "!=", "Eq", "NotEq"):
            def is_col(v):
                return isinstance(v, dict) and len(v) == 1 and "column" in v

            def long_lit(v):
                return (isinstance(v, dict) and len(v) == 1 and isinstance(v.get("literal"), dict) and v["literal"].get("type") == "Utf8"
                        and isinstance(v["literal"].get("value"), str) and len(v["literal"]["value"].encode()) > 15)
            pair = (b["l"], b["r"]) if is_col(b.get("l")) and long_lit(b.get("r")) else (b["r"], b["l"]) if is_col(b.get("r")) and long_lit(b.get("l")) else None
            if pair:
                pat = pair[1]["literal"]["value"].replace("%", "\\%").replace("_", "\\_")
                return {"like_expr": {"negated": b["op"] in ("!=", "NotEq"), "case_insensitive": False, "expr": copy.deepcopy(pair[0]),
                                      "pattern": {"literal": {"type": "Utf8", "value": pat}}}}
        return {k: lower_long_string_eq(v) for k, v in e.items()}
    if isinstance(e, (list, tuple)):
        return type(e)(lower_long_string_eq(v) for v in e)
    return e


def like_placeholder(e):
    """For type inference only: LIKE is Boolean and NULL exactly where its operand is -- as `operand = operand`."""
    return rewrite_like(e, lambda v: {"binary_expr": {"l": v["expr"], "r": copy.deepcopy(v["expr"]), "op": "Eq"}})
