"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (arrow-ballista_amd/);
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, as the checker.

CPU restatement, in plain Python integers (exact, arbitrary precision), of the relational semantics
of the operators on the reference's hot path.  The arithmetic of that path is NOT in the reference
tree: it lives in the third-party crates `datafusion` (git coralogix/arrow-datafusion tag
v34.0.0-cx.1) and `arrow` 49.0.0 (reference Cargo.toml:33-43), which are absent from the container
(SURVEY.md §0.2, §8c).  Each function therefore restates the published algorithm and cites the
reference call site / parameter surface it follows:

  eval_expr      PhysicalExprNode semantics        ballista/core/proto/datafusion.proto:1142-1180, ops :1228-1232
  filter_rows    FilterExec                        datafusion.proto:1291-1294 ; named task_group.rs:26,155
  aggregate      AggregateExec (all modes)         datafusion.proto:1405-1450, fns :631-669
  hash_join      HashJoinExec                      datafusion.proto:1346-1360, join types :280-289 ; ctor task_group.rs:306-315
  sort_perm      SortExec                          datafusion.proto:1465-1471, options :1247-1251
  hash_partition BatchPartitioner::partition       ballista/core/src/execution_plans/shuffle_writer.rs:336-391

PARITY PINNING.  Pinned by the reference's own known answers only for ungrouped aggregates over the
8-row alltypes_plain table (ballista/client/src/context.rs:762-967: SUM(id)=28, AVG(id)=3.5, MIN=0,
MAX=7, COUNT=8) -- tests/test_oracle_pins.py.  For hash-join output, sort order, filter output,
grouped aggregation, Decimal128 result precision/scale and hash-partition assignment the reference
holds NO asserting test: for those rows this oracle is "parity unpinned"; it is
cross-checked against pyarrow Acero on raw integers as a second opinion (tests/test_oracle_vs_acero.py).

Decimal type rules restated from DataFusion v34 / arrow-rs 49 [UPSTREAM-KNOWLEDGE]:
  Int64 -> Decimal128(20,0), Int32 -> Decimal128(10,0) in decimal context
  add/sub: s=max(s1,s2), p=min(38,max(p1-s1,p2-s2)+s+1);  mul: s=s1+s2, p=min(38,p1+p2+1)
  SUM(decimal(p,s)) -> (min(38,p+10), s);  AVG(decimal(p,s)) -> (min(38,p+4), min(38,s+4)),
  value = sum*10^(s_avg-s) / count truncated toward zero;  AVG(int|float) -> Float64
"""
import math
import struct

MASK64 = (1 << 64) - 1


# ---------------------------------------------------------------- types
def is_dec(t):
    return isinstance(t, dict)


def dec(p, s):
    return {"Decimal128": [min(38, p), min(38, s)]}


def as_dec(t):
    if is_dec(t):
        return t
    if t in ("Int32", "UInt32"):
        return dec(10, 0)
    if t in ("Int64", "UInt64"):
        return dec(20, 0)
    raise TypeError("not decimal-able: %s" % (t,))


def ps(t):
    return t["Decimal128"][0], t["Decimal128"][1]


def is_int(t):
    return t in ("Int32", "Int64", "UInt32", "UInt64")


def trunc_div(a, b):
    q = abs(a) // abs(b)
    return -q if (a < 0) != (b < 0) else q


def total_order_key(x):
    """IEEE-754 totalOrder as an integer (arrow-ord cmp kernels compare floats by total order)."""
    b = struct.unpack("<q", struct.pack("<d", x))[0]
    return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFF)


class Table:
    """name -> (type, values) with python values (None = NULL); decimals as unscaled ints; Date32 as days."""

    def __init__(self, names, types, cols):
        self.names, self.types, self.cols = list(names), list(types), [list(c) for c in cols]
        self.n = len(self.cols[0]) if self.cols else 0

    @staticmethod
    def from_arrow(t):
        import pyarrow as pa
        names, types, cols = [], [], []
        for f, c in zip(t.schema, t.columns):
            ty = f.type
            if pa.types.is_decimal128(ty):
                tj = dec(ty.precision, ty.scale)
                vals = [None if v is None else int(v.scaleb(ty.scale)) for v in c.to_pylist()]
            elif pa.types.is_date32(ty):
                tj = "Date32"
                vals = c.cast(pa.int32()).to_pylist()
            elif pa.types.is_string(ty) or pa.types.is_large_string(ty):
                tj, vals = "Utf8", c.to_pylist()
            elif pa.types.is_binary(ty):
                tj, vals = "Utf8", [None if v is None else v.decode() for v in c.to_pylist()]
            elif pa.types.is_boolean(ty):
                tj, vals = "Boolean", c.to_pylist()
            elif pa.types.is_floating(ty):
                tj, vals = "Float64", [None if v is None else float(v) for v in c.to_pylist()]
            else:
                tj = {pa.int32(): "Int32", pa.int64(): "Int64", pa.uint32(): "UInt32", pa.uint64(): "UInt64",
                      pa.int8(): "Int32", pa.int16(): "Int32"}[ty]
                vals = c.to_pylist()
            names.append(f.name); types.append(tj); cols.append(vals)
        return Table(names, types, cols)

    def col(self, name):
        return self.cols[self.names.index(name)]

    def take(self, rows):
        return Table(self.names, self.types, [[None if r is None else c[r] for r in rows] for c in self.cols])

    def rows(self):
        return list(zip(*self.cols)) if self.cols else []


# ---------------------------------------------------------------- expressions
_OPS = {"Plus": "+", "Minus": "-", "Multiply": "*", "Divide": "/", "Modulo": "%", "Eq": "=", "NotEq": "!=", "Lt": "<", "LtEq": "<=",
        "Gt": ">", "GtEq": ">=", "And": "AND", "Or": "OR"}


def _rescale(vals, t, s_new):
    p, s = ps(as_dec(t))
    if s_new == s:
        return vals, dec(p, s)
    if s_new > s:
        m = 10 ** (s_new - s)
        return [None if v is None else v * m for v in vals], dec(p + s_new - s, s_new)
    d = 10 ** (s - s_new)
    out = []
    for v in vals:   # round half away from zero (arrow cast decimal->decimal)
        if v is None:
            out.append(None)
        else:
            q, r = trunc_div(v, d), v - trunc_div(v, d) * d
            if v >= 0 and r >= d // 2 and d > 1:
                q += 1
            elif v < 0 and r <= -(d // 2) and d > 1:
                q -= 1
            out.append(q)
    return out, dec(max(1, p - (s - s_new)), s_new)


def _to_float(vals, t):
    if t == "Float64":
        return vals
    if is_dec(t):
        s = ps(t)[1]
        return [None if v is None else float(v) / (10.0 ** s) if s else float(v) for v in vals]
    return [None if v is None else float(v) for v in vals]


def like_match(s, pat):
    """arrow-string 49 like.rs with a scalar pattern [UPSTREAM-KNOWLEDGE]: equality / starts_with / ends_with / contains for
    the four wildcard-free shapes, else the regex ^...$ built by replace_like_wildcards ('%' -> '.*', '_' -> '.', "\\%" and
    "\\_" literal, every other character literal; '.' does not match a newline)."""
    import re
    wild = lambda t: any(c in "%_" for c in t)
    if not wild(pat):
        return s == pat
    if pat.endswith("%") and not pat.endswith("\\%") and not wild(pat[:-1]):
        return s.startswith(pat[:-1])
    if pat.startswith("%") and not wild(pat[1:]):
        return s.endswith(pat[1:])
    if pat.startswith("%") and pat.endswith("%") and not pat.endswith("\\%") and not wild(pat[1:-1]):
        return pat[1:-1] in s
    out, i = [], 0
    while i < len(pat):
        c = pat[i]
        if c == "\\" and i + 1 < len(pat) and pat[i + 1] in "%_":
            out.append(re.escape(pat[i + 1])); i += 2; continue
        out.append(".*" if c == "%" else ("." if c == "_" else re.escape(c)))
        i += 1
    return re.fullmatch("".join(out), s) is not None


def eval_expr(e, tab):
    """-> (type, values)."""
    (kind, v), = e.items()
    n = tab.n
    if kind == "column":
        i = tab.names.index(v["name"]) if v.get("name") in tab.names else v["index"]
        return tab.types[i], list(tab.cols[i])
    if kind == "literal":
        t, val = v["type"], v.get("value")
        if val is None:
            return t, [None] * n
        if t == "Utf8":
            return t, [val] * n
        if t == "Float64":
            return t, [float(val)] * n
        if t == "Boolean":
            return t, [bool(val)] * n
        return t, [int(val)] * n
    if kind in ("cast", "try_cast"):
        t, vals = eval_expr(v["expr"], tab)
        to = v["arrow_type"]
        return to, _cast(vals, t, to)
    if kind == "not_expr":
        t, vals = eval_expr(v["expr"], tab)
        return "Boolean", [None if x is None else (not x) for x in vals]
    if kind == "is_null_expr":
        t, vals = eval_expr(v["expr"], tab)
        return "Boolean", [x is None for x in vals]
    if kind == "is_not_null_expr":
        t, vals = eval_expr(v["expr"], tab)
        return "Boolean", [x is not None for x in vals]
    if kind == "negative":
        t, vals = eval_expr(v["expr"], tab)
        return t, [None if x is None else -x for x in vals]
    if kind == "like_expr":
        _, vals = eval_expr(v["expr"], tab)
        pat = v["pattern"]["literal"]["value"]
        if v.get("case_insensitive"):
            raise NotImplementedError("ILIKE")
        neg = bool(v.get("negated"))
        return "Boolean", [None if x is None else (like_match(x, pat) != neg) for x in vals]
    if kind == "in_list":
        acc = None
        for it in v["list"]:
            eq = {"binary_expr": {"l": v["expr"], "r": it, "op": "="}}
            acc = eq if acc is None else {"binary_expr": {"l": acc, "r": eq, "op": "OR"}}
        if acc is None:
            return "Boolean", [False] * n
        t, vals = eval_expr(acc, tab)
        if v.get("negated"):
            vals = [None if x is None else (not x) for x in vals]
        return "Boolean", vals
    if kind == "case_":
        base = v.get("expr")
        et, ev = eval_expr(v["else_expr"], tab) if v.get("else_expr") is not None else ("Null", [None] * n)
        branches = []
        for wt in v["when_then_expr"]:
            w = wt["when_expr"] if base is None else {"binary_expr": {"l": base, "r": wt["when_expr"], "op": "="}}
            branches.append((eval_expr(w, tab)[1], eval_expr(wt["then_expr"], tab)))
        # common result type
        types = [bt for _, (bt, _) in branches] + ([et] if et != "Null" else [])
        rt = _common_type(types)
        out = _cast(ev, et, rt) if et != "Null" else [None] * n
        for cond, (bt, bv) in reversed(branches):
            bv = _cast(bv, bt, rt) if bt != "Null" else [None] * n
            out = [bv[i] if cond[i] else out[i] for i in range(n)]
        return rt, out
    if kind == "scalar_function":
        # PhysicalScalarFunctionNode.  date_part [UPSTREAM-KNOWLEDGE: datafusion 34 returns Float64; arrow-arith 49 temporal kernels]
        # and substr (1-based start, characters) -- the two functions TPC-H q7-q9 / q22 need (benchmarks/queries/q7.sql:11, q22.sql:8)
        nm = v["name"].lower()
        if nm in ("date_part", "datepart"):
            import datetime
            part = v["args"][0]["literal"]["value"].upper()
            _, vals = eval_expr(v["args"][1], tab)
            def f(d):
                x = datetime.date(1970, 1, 1) + datetime.timedelta(days=int(d))
                return float({"YEAR": x.year, "MONTH": x.month, "DAY": x.day}[part])
            return "Float64", [None if d is None else f(d) for d in vals]
        if nm in ("substr", "substring"):
            _, vals = eval_expr(v["args"][0], tab)
            start = int(v["args"][1]["literal"]["value"])
            ln = int(v["args"][2]["literal"]["value"]) if len(v["args"]) > 2 else None
            return "Utf8", [None if x is None else (x[start - 1:] if ln is None else x[start - 1: start - 1 + ln]) for x in vals]
        raise NotImplementedError(nm)
    if kind == "binary_expr":
        op = _OPS.get(v["op"], v["op"])
        lt, lv = eval_expr(v["l"], tab)
        rt, rv = eval_expr(v["r"], tab)
        return _binary(op, lt, lv, rt, rv)
    raise NotImplementedError(kind)


def _common_type(types):
    types = [t for t in types if t != "Null"]
    if not types:
        return "Null"
    t0 = types[0]
    for t in types[1:]:
        if t == t0:
            continue
        if "Float64" in (t, t0):
            t0 = "Float64"
        elif is_dec(t) or is_dec(t0):
            (p1, s1), (p2, s2) = ps(as_dec(t0)), ps(as_dec(t))
            s = max(s1, s2)
            t0 = dec(min(38, max(p1 - s1, p2 - s2) + s), s)
        else:
            t0 = "Int64"
    return t0


def _cast(vals, t, to):
    if t == to or t == "Null":
        return list(vals)
    if to == "Float64":
        return _to_float(vals, t)
    if is_dec(to):
        return _rescale(vals, t, ps(to)[1])[0]
    if is_int(to) or to == "Date32":
        if is_dec(t):
            return _rescale(vals, t, 0)[0]
        if t == "Float64":
            return [None if x is None else int(x) for x in vals]
        return [None if x is None else int(x) for x in vals]
    if to == "Boolean":
        return [None if x is None else x != 0 for x in vals]
    raise NotImplementedError("cast %s -> %s" % (t, to))


def _binary(op, lt, lv, rt, rv):
    n = len(lv)
    if op in ("AND", "OR"):
        out = []
        for a, b in zip(lv, rv):   # Kleene
            if op == "AND":
                out.append(False if (a is False or b is False) else (None if (a is None or b is None) else True))
            else:
                out.append(True if (a is True or b is True) else (None if (a is None or b is None) else False))
        return "Boolean", out
    if lt == "Null":
        lt = rt
    if rt == "Null":
        rt = lt
    cmp = op in ("=", "!=", "<", "<=", ">", ">=")
    if cmp:
        if lt == "Float64" or rt == "Float64":
            a, b = [None if x is None else total_order_key(x) for x in _to_float(lv, lt)], [None if x is None else total_order_key(x) for x in _to_float(rv, rt)]
        elif is_dec(lt) or is_dec(rt):
            s = max(ps(as_dec(lt))[1], ps(as_dec(rt))[1])
            a, b = _rescale(lv, lt, s)[0], _rescale(rv, rt, s)[0]
        elif lt == "Utf8":
            a, b = [None if x is None else x.encode() for x in lv], [None if x is None else x.encode() for x in rv]
        else:
            a, b = lv, rv
        f = {"=": lambda x, y: x == y, "!=": lambda x, y: x != y, "<": lambda x, y: x < y, "<=": lambda x, y: x <= y,
             ">": lambda x, y: x > y, ">=": lambda x, y: x >= y}[op]
        return "Boolean", [None if (x is None or y is None) else f(x, y) for x, y in zip(a, b)]
    if lt == "Float64" or rt == "Float64":
        a, b = _to_float(lv, lt), _to_float(rv, rt)
        f = {"+": lambda x, y: x + y, "-": lambda x, y: x - y, "*": lambda x, y: x * y,
             "/": lambda x, y: (x / y if y != 0 else (math.nan if x == 0 or x != x else math.copysign(math.inf, x) * math.copysign(1, y)))}[op]
        return "Float64", [None if (x is None or y is None) else f(x, y) for x, y in zip(a, b)]
    if is_dec(lt) or is_dec(rt):
        (p1, s1), (p2, s2) = ps(as_dec(lt)), ps(as_dec(rt))
        if op in ("+", "-"):
            s = max(s1, s2)
            a, b = _rescale(lv, lt, s)[0], _rescale(rv, rt, s)[0]
            t = dec(min(38, max(p1 - s1, p2 - s2) + s + 1), s)
            return t, [None if (x is None or y is None) else (x + y if op == "+" else x - y) for x, y in zip(a, b)]
        if op == "*":
            return dec(min(38, p1 + p2 + 1), min(38, s1 + s2)), [None if (x is None or y is None) else x * y for x, y in zip(lv, rv)]
        # arrow-arith 49 numeric.rs decimal_op [UPSTREAM-KNOWLEDGE]: Div -> scale s1+4 ("follow postgres and MySQL adding a
        # fixed scale increment of 4"), precision p1 + (4 + s2), value = l*10^(4+s2) / r truncated toward zero;
        # Rem -> scale max(s1,s2), precision min(p1-s1, p2-s2) + scale, sign of the dividend.  x/0 raises DivideByZero in
        # arrow; here (as for integers) the slot becomes NULL -- the device cannot raise per row.
        if op == "/":
            rs = min(38, s1 + 4)
            k = rs - s1 + s2
            a = _rescale(lv, lt, s1)[0]; b = _rescale(rv, rt, s2)[0]
            return dec(min(38, p1 + k), rs), [None if (x is None or y is None or y == 0) else trunc_div(x * 10 ** k, y) for x, y in zip(a, b)]
        if op == "%":
            s = max(s1, s2)
            a, b = _rescale(lv, lt, s)[0], _rescale(rv, rt, s)[0]
            return dec(min(38, min(p1 - s1, p2 - s2) + s), s), [None if (x is None or y is None or y == 0) else x - trunc_div(x, y) * y for x, y in zip(a, b)]
        raise NotImplementedError("decimal %s" % op)
    # integers / dates
    if lt == "Date32" and rt == "Date32" and op == "-":
        t = "Int32"
    elif "Date32" in (lt, rt):
        t = "Date32"
    else:
        t = "Int64" if ("Int64" in (lt, rt) or "UInt64" in (lt, rt)) else "Int32"
    out = []
    for x, y in zip(lv, rv):
        if x is None or y is None:
            out.append(None)
        elif op == "+":
            out.append(x + y)
        elif op == "-":
            out.append(x - y)
        elif op == "*":
            out.append(x * y)
        elif op == "/":
            out.append(None if y == 0 else trunc_div(x, y))
        elif op == "%":
            out.append(None if y == 0 else x - trunc_div(x, y) * y)
    return t, out


# ---------------------------------------------------------------- operators
def filter_rows(tab, predicate):
    """FilterExec: indices of rows whose predicate is TRUE (NULL drops the row), in input order."""
    _, vals = eval_expr(predicate, tab)
    return [i for i, v in enumerate(vals) if v is True]


def project(tab, exprs, names):
    types, cols = [], []
    for e in exprs:
        t, v = eval_expr(e, tab)
        types.append(t); cols.append(v)
    return Table(names, types, cols)


def aggregate(tab, group_exprs, aggs, mode="Single", predicate=None):
    """AggregateExec.  group_exprs: [(expr, name)]; aggs: [{"fn","expr","name"}].
    Returns Table (group columns, then per aggregate: Partial -> state columns, else final value); group order = first appearance."""
    if predicate is not None:
        tab = tab.take(filter_rows(tab, predicate))
    gvals = [eval_expr(e, tab) for e, _ in group_exprs]
    final = mode in ("Final", "FinalPartitioned")
    ng = len(group_exprs)
    groups, order = {}, []
    for i in range(tab.n):
        k = tuple(g[1][i] for g in gvals)
        if k not in groups:
            groups[k] = []
            order.append(k)
        groups[k].append(i)
    if not group_exprs and not order:
        order, groups = [()], {(): []}      # ungrouped aggregate over zero rows still yields one row
    names = [n for _, n in group_exprs]
    types = [g[0] for g in gvals]
    cols = [[k[j] for k in order] for j in range(ng)]
    state_col = ng
    for a in aggs:
        fn = a["fn"].upper()
        if not final:
            if a.get("expr") is None:
                at, av = "Int64", [1] * tab.n
            else:
                at, av = eval_expr(a["expr"], tab)
            if a.get("filter") is not None:
                # agg(x) FILTER (WHERE p): only the rows where p is TRUE take part (AggregateExecNode.filter_expr, datafusion.proto:1437-1450)
                _, fv_ = eval_expr(a["filter"], tab)
                av = [x if f is True else None for x, f in zip(av, fv_)]
            if a.get("distinct"):
                # agg(DISTINCT x): every value counts once per group (NULLs never count)
                seen_ = {}
                for k_ in order:
                    s_ = set()
                    for i in groups[k_]:
                        if av[i] is not None and av[i] in s_:
                            seen_[i] = True
                        elif av[i] is not None:
                            s_.add(av[i])
                av = [None if seen_.get(i) else x for i, x in enumerate(av)]
            if fn == "COUNT":
                names.append(a["name"] + ("[count]" if mode == "Partial" else "")); types.append("Int64")
                cols.append([sum(1 for i in groups[k] if av[i] is not None) for k in order])
            elif fn in ("SUM", "AVG"):
                if is_dec(at):
                    p, s = ps(at)
                    st = dec(p + 10, s)
                    sums = [_sum_or_none([av[i] for i in groups[k]]) for k in order]
                elif is_int(at) and fn == "SUM":
                    st = "Int64"
                    sums = [_wrap64(_sum_or_none([av[i] for i in groups[k]])) for k in order]
                else:
                    st = "Float64"
                    fv = _to_float(av, at)
                    sums = [_fsum_or_none([fv[i] for i in groups[k]]) for k in order]
                cnts = [sum(1 for i in groups[k] if av[i] is not None) for k in order]
                if fn == "SUM":
                    names.append(a["name"] + ("[sum]" if mode == "Partial" else "")); types.append(st); cols.append(sums)
                elif mode == "Partial":
                    names += [a["name"] + "[count]", a["name"] + "[sum]"]; types += ["UInt64", st]; cols += [cnts, sums]
                else:
                    rt, vals = _avg_final(st, sums, cnts, at)
                    names.append(a["name"]); types.append(rt); cols.append(vals)
            elif fn in ("MIN", "MAX"):
                pick = min if fn == "MIN" else max
                key = (lambda x: total_order_key(x)) if at == "Float64" else (lambda x: x)
                vals = []
                for k in order:
                    xs = [av[i] for i in groups[k] if av[i] is not None]
                    vals.append(pick(xs, key=key) if xs else None)
                names.append(a["name"] + (("[min]" if fn == "MIN" else "[max]") if mode == "Partial" else "")); types.append(at); cols.append(vals)
            elif fn in _VAR1 or fn in _VAR2:
                xv = _to_float(av, at)
                yv = None
                if fn in _VAR2:
                    yt, yraw = eval_expr(a["expr2"], tab)
                    yv = _to_float(yraw, yt)
                sts = [_var_state([(xv[i], yv[i] if yv is not None else 0.0) for i in groups[k]
                                   if xv[i] is not None and (yv is None or yv[i] is not None)]) for k in order]
                if mode == "Partial":
                    lay = _var_layout(fn)
                    names += [a["name"] + "[%s]" % nm for nm, _ in lay]; types += ["UInt64"] + ["Float64"] * (len(lay) - 1)
                    cols += [[st[key] for st in sts] for _, key in lay]
                else:
                    names.append(a["name"]); types.append("Float64"); cols.append([_var_final(fn, st) for st in sts])
            else:
                raise NotImplementedError(fn)
        else:
            if fn in _VAR1 or fn in _VAR2:
                lay = _var_layout(fn)
                sv = [tab.cols[state_col + j] for j in range(len(lay))]; state_col += len(lay)
                vals = []
                for k in order:
                    st = _var_state([])
                    for i in groups[k]:
                        st = _var_merge(st, {key: sv[j][i] for j, (_, key) in enumerate(lay)})
                    vals.append(_var_final(fn, st))
                names.append(a["name"]); types.append("Float64"); cols.append(vals)
            elif fn == "COUNT":
                sv = tab.cols[state_col]; state_col += 1
                names.append(a["name"]); types.append("Int64"); cols.append([sum(sv[i] for i in groups[k]) for k in order])
            elif fn == "SUM":
                st, sv = tab.types[state_col], tab.cols[state_col]; state_col += 1
                f = _fsum_or_none if st == "Float64" else _sum_or_none
                names.append(a["name"]); types.append(st); cols.append([f([sv[i] for i in groups[k]]) for k in order])
            elif fn == "AVG":
                cv = tab.cols[state_col]; st, sv = tab.types[state_col + 1], tab.cols[state_col + 1]; state_col += 2
                cnts = [sum(cv[i] for i in groups[k]) for k in order]
                f = _fsum_or_none if st == "Float64" else _sum_or_none
                sums = [f([sv[i] for i in groups[k]]) for k in order]
                at = dec(max(1, ps(st)[0] - 10), ps(st)[1]) if is_dec(st) else "Float64"
                rt, vals = _avg_final(st, sums, cnts, at)
                names.append(a["name"]); types.append(rt); cols.append(vals)
            elif fn in ("MIN", "MAX"):
                st, sv = tab.types[state_col], tab.cols[state_col]; state_col += 1
                pick = min if fn == "MIN" else max
                key = (lambda x: total_order_key(x)) if st == "Float64" else (lambda x: x)
                vals = []
                for k in order:
                    xs = [sv[i] for i in groups[k] if sv[i] is not None]
                    vals.append(pick(xs, key=key) if xs else None)
                names.append(a["name"]); types.append(st); cols.append(vals)
            else:
                raise NotImplementedError(fn)
    return Table(names, types, cols)


# ---- VARIANCE / STDDEV / COVARIANCE / CORRELATION (datafusion.proto:639-645).
# Restates DataFusion v34's running accumulators [UPSTREAM-KNOWLEDGE: VarianceAccumulator / CovarianceAccumulator /
# CorrelationAccumulator]: rows are folded one at a time IN ROW ORDER with Welford updates, partial states merge with
# Chan's formula.  Pinned by ballista/client/src/context.rs:845-937 (tests/test_oracle_pins.py): the KAT values
# 6.000000000000001 / 5.250000000000001 / 2.4494897427831783 / 0.28571428571428586 / 0.21821789023599245 are
# reproduced bit for bit only by this update order.
_VAR1 = ("VARIANCE", "VAR", "VAR_SAMP", "VARIANCE_POP", "VAR_POP", "STDDEV", "STDDEV_SAMP", "STDDEV_POP")
_VAR2 = ("COVARIANCE", "COVAR", "COVAR_SAMP", "COVARIANCE_POP", "COVAR_POP", "CORRELATION", "CORR")


def _var_layout(fn):
    """Partial-state columns in the reference's order: (suffix, state key)."""
    if fn in _VAR1:
        return [("count", "n"), ("mean", "m1"), ("m2", "v1")]
    if fn in ("CORRELATION", "CORR"):
        return [("count", "n"), ("mean1", "m1"), ("m2_1", "v1"), ("mean2", "m2"), ("m2_2", "v2"), ("algoConst", "al")]
    return [("count", "n"), ("mean1", "m1"), ("mean2", "m2"), ("algoConst", "al")]


def _var_state(pairs):
    n, m1, m2, v1, v2, al = 0, 0.0, 0.0, 0.0, 0.0, 0.0
    for x, y in pairs:
        n += 1
        d1 = x - m1; nm1 = d1 / n + m1
        d2 = y - m2; nm2 = d2 / n + m2
        v1 += d1 * (x - nm1)
        v2 += d2 * (y - nm2)
        al += d1 * (y - nm2)
        m1, m2 = nm1, nm2
    return {"n": n, "m1": m1, "m2": m2, "v1": v1, "v2": v2, "al": al}


def _var_merge(a, b):
    b = dict({"m2": 0.0, "v1": 0.0, "v2": 0.0, "al": 0.0}, **b)
    if b["n"] == 0:
        return a
    if a["n"] == 0:
        return dict(b)
    n = a["n"] + b["n"]
    d1 = a["m1"] - b["m1"]; d2 = a["m2"] - b["m2"]
    return {"n": n,
            "m1": a["m1"] * a["n"] / n + b["m1"] * b["n"] / n,
            "m2": a["m2"] * a["n"] / n + b["m2"] * b["n"] / n,
            "v1": a["v1"] + b["v1"] + d1 * d1 * a["n"] * b["n"] / n,
            "v2": a["v2"] + b["v2"] + d2 * d2 * a["n"] * b["n"] / n,
            "al": a["al"] + b["al"] + d1 * d2 * a["n"] * b["n"] / n}


def _var_final(fn, st):
    import math
    n = st["n"]
    pop = fn.endswith("_POP")
    if fn in ("CORRELATION", "CORR"):
        if n == 0:
            return None
        s1, s2 = math.sqrt(st["v1"] / n), math.sqrt(st["v2"] / n)
        return 0.0 if s1 == 0 or s2 == 0 else st["al"] / n / s1 / s2
    if n == 0 or (n == 1 and not pop):
        return None
    d = n if pop else n - 1
    if fn in _VAR2:
        return st["al"] / d
    v = st["v1"] / d
    return math.sqrt(v) if fn.startswith("STDDEV") else v


def _sum_or_none(xs):
    xs = [x for x in xs if x is not None]
    return sum(xs) if xs else None


def _fsum_or_none(xs):
    xs = [x for x in xs if x is not None]
    if not xs:
        return None
    s = 0.0
    for x in xs:     # sequential left-to-right, like a CPU accumulator
        s += x
    return s


def _wrap64(v):
    if v is None:
        return None
    v &= MASK64
    return v - (1 << 64) if v >> 63 else v


def _avg_final(st, sums, cnts, at):
    if is_dec(st):
        p_arg, s = ps(at)[0], ps(st)[1]
        rt = dec(p_arg + 4, s + 4)
        mul = 10 ** (ps(rt)[1] - s)
        return rt, [None if (c == 0 or v is None) else trunc_div(v * mul, c) for v, c in zip(sums, cnts)]
    return "Float64", [None if (c == 0 or v is None) else v / c for v, c in zip(sums, cnts)]


def hash_join(left, right, on, join_type="Inner", null_equals_null=False, left_pred=None, right_pred=None, pair_filter=None):
    """HashJoinExec: list of (left_row | None, right_row | None).  Build = LEFT input.  Pair order unspecified.
    pair_filter(i, j) -> True/False/None: the JoinFilter; a key match it does not accept (False or NULL) is no match, for the
    outer / semi / anti bookkeeping as well (datafusion HashJoinExec applies the filter before it marks rows visited)."""
    lrows = filter_rows(left, left_pred) if left_pred is not None else list(range(left.n))
    rrows = filter_rows(right, right_pred) if right_pred is not None else list(range(right.n))
    lk = [eval_expr(l, left)[1] for l, _ in on]
    rk = [eval_expr(r, right)[1] for _, r in on]
    table = {}
    for i in lrows:
        k = tuple(c[i] for c in lk)
        if any(x is None for x in k) and not null_equals_null:
            continue
        table.setdefault(k, []).append(i)
    pairs, visited = [], set()
    matched_right = []
    for j in rrows:
        k = tuple(c[j] for c in rk)
        ms = [] if (any(x is None for x in k) and not null_equals_null) else table.get(k, [])
        if pair_filter is not None:
            ms = [i for i in ms if pair_filter(i, j) is True]
        for i in ms:
            pairs.append((i, j)); visited.add(i)
        matched_right.append((j, bool(ms)))
    jt = join_type
    if jt == "Inner":
        return pairs
    if jt == "Right":
        return pairs + [(None, j) for j, m in matched_right if not m]
    if jt == "Left":
        return pairs + [(i, None) for i in lrows if i not in visited]
    if jt == "Full":
        return pairs + [(None, j) for j, m in matched_right if not m] + [(i, None) for i in lrows if i not in visited]
    if jt == "LeftSemi":
        return [(i, None) for i in lrows if i in visited]
    if jt == "LeftAnti":
        return [(i, None) for i in lrows if i not in visited]
    if jt == "RightSemi":
        return [(None, j) for j, m in matched_right if m]
    if jt == "RightAnti":
        return [(None, j) for j, m in matched_right if not m]
    raise NotImplementedError(jt)


def sort_keys(tab, sort_exprs):
    """Per-row comparable key tuples for SortExec (asc/desc, nulls_first)."""
    cols = []
    for s in sort_exprs:
        t, v = eval_expr(s["expr"], tab)
        asc = s.get("asc", True)
        nf = s.get("nulls_first", not asc)
        col = []
        for x in v:
            if x is None:
                col.append((0 if nf else 2, 0))
            else:
                if t == "Float64":
                    x = total_order_key(x)
                elif t == "Utf8":
                    x = int.from_bytes(x.encode()[:15].ljust(15, b"\0"), "big") * 256 + min(len(x.encode()), 255)
                elif t == "Boolean":
                    x = int(x)
                col.append((1, x if asc else -x))
        cols.append(col)
    return list(zip(*cols)) if cols else [()] * tab.n


def sort_perm(tab, sort_exprs):
    """SortExec: a stable permutation."""
    keys = sort_keys(tab, sort_exprs)
    return sorted(range(tab.n), key=lambda i: keys[i])


# ---------------------------------------------------------------- hashing (gpuq's own partition function)
def mix64(x):
    x &= MASK64
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & MASK64
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & MASK64
    x ^= x >> 31
    return x


def _reg128(t, v):
    """(lo, hi) of the 128-bit register value the device holds for a non-null value."""
    if t == "Utf8":
        b = v.encode()
        hi = int.from_bytes(b[:8].ljust(8, b"\0"), "big")
        lo = int.from_bytes(b[8:15].ljust(7, b"\0"), "big") << 8 | min(len(b), 255)
        return lo, hi
    if t == "Float64":
        return struct.unpack("<Q", struct.pack("<d", v))[0], 0
    if t == "Boolean":
        return int(v), 0
    x = int(v) & ((1 << 128) - 1)
    return x & MASK64, x >> 64


def hash_row(types, values):
    h = 0x243F6A8885A308D3
    for t, v in zip(types, values):
        wide = is_dec(t) or t == "Utf8"
        if v is None:
            c = 0x9E3779B97F4A7C15
        else:
            lo, hi = _reg128(t, v)
            c = mix64(lo ^ mix64(((hi if wide else 0) + 0x632BE59BD9B4E019) & MASK64))
        h = mix64((h * 31 + c + 0x9E3779B97F4A7C15) & MASK64)
    return h


def hash_partition(tab, hash_exprs, n):
    """Partition id per row: hash(keys) % n with gpuq's mix64 hash (not ahash)."""
    ev = [eval_expr(e, tab) for e in hash_exprs]
    return [hash_row([t for t, _ in ev], [v[i] for _, v in ev]) % n for i in range(tab.n)]
