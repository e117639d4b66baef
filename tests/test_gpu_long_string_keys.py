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


def test_the_mirror_still_refuses_and_says_why(tc):
    """The Python mirror (test-side) has no dictionary path: it fails loudly, it does not truncate keys."""
    t = names_table(100, 3, 10)
    src = g.MemoryExec([t])
    s = src.schema()
    plan = g.AggregateExec("Single", [(col("name", s), "name")], [{"fn": "COUNT", "expr": lit(1), "name": "c"}], src)
    with pytest.raises(g.GpuqError, match="15 bytes"):
        g.plan.materialize(tc, plan.execute(0, tc))
