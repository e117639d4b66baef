"""The C-ABI boundary exactly as a non-Python host would drive it: Arrow C Data Interface in,
operator calls on raw device pointers, Arrow C Data Interface out -- no torch tensors involved."""
import ctypes as C
import json

import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd import binding as B
from arrow_ballista_amd.expr import Operator as Op
from arrow_ballista_amd.expr import binary, col, lit

pytestmark = pytest.mark.gpu


class ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


def test_import_filter_export_round_trip(tc):
    import decimal
    ctx, L = tc.ctx, tc.ctx.L
    r = np.random.default_rng(0)
    n = 10_000
    t = pa.table({
        "k": pa.array(r.integers(-100, 100, n), type=pa.int64()),
        "d": pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in r.integers(-10**6, 10**6, n)], type=pa.decimal128(15, 2)),
        "s": pa.array(np.array(["AIR", "MAIL", "SHIP", ""])[r.integers(0, 4, n)], mask=r.random(n) < 0.1),
        "dt": pa.array(r.integers(8000, 9000, n).astype(np.int32)).cast(pa.date32()),
        "f": pa.array(r.normal(size=n)),
        "b": pa.array(r.integers(0, 2, n).astype(bool)),
    })
    batch = t.slice(37, n - 100).to_batches()[0]          # non-zero offset: buffers must be re-based
    ca, cs = ArrowArray(), ArrowSchema()
    batch._export_to_c(C.addressof(ca), C.addressof(cs))
    h = C.c_void_p()
    ctx.check(L.gpuq_table_import_arrow(ctx.h, None, C.addressof(ca), C.addressof(cs), C.byref(h)))
    nr, nc = L.gpuq_table_num_rows(h), L.gpuq_table_num_columns(h)
    assert nr == batch.num_rows and nc == 6
    cols = (B.gpuq_column * nc)()
    fields = (B.gpuq_field_info * nc)()
    for i in range(nc):
        ctx.check(L.gpuq_table_column(h, i, C.byref(cols[i]), C.byref(fields[i])))
    assert [f.name.decode() for f in fields] == ["k", "d", "s", "dt", "f", "b"]
    # FilterExec on the imported columns, selection in a gpuq buffer
    schema = [{"name": f.name.decode(), "type": g.table.type_json(f.type, f.precision, f.scale), "nullable": bool(f.nullable)} for f in fields]
    pred = binary(binary(col("k", schema), Op.Gt, lit(0)), Op.And, binary(col("s", schema), Op.NotEq, lit("MAIL")))
    op = g.Op(ctx, {"op": "filter", "input": {"fields": schema}, "predicate": pred})
    sel, cnt = C.c_void_p(), C.c_void_p()
    ctx.check(L.gpuq_buffer_alloc(ctx.h, 4 * nr, C.byref(sel)))
    ctx.check(L.gpuq_buffer_alloc(ctx.h, 8, C.byref(cnt)))
    inp = B.gpuq_input(); inp.cols = C.cast(cols, C.POINTER(B.gpuq_column)); inp.n_cols = nc; inp.n_via = 0; inp.n_rows = nr
    ctx.check(L.gpuq_filter_run(op.h, None, C.byref(inp), 0, sel, cnt))
    k = C.c_uint64(0)
    ctx.check(L.gpuq_copy_d2h(ctx.h, None, C.byref(k), cnt, 8))
    op.check()
    # ProjectionExec through the selection = materialise the surviving rows
    vschema = [dict(f, side=1) for f in schema]
    pop = g.Op(ctx, {"op": "project", "input": {"fields": vschema}, "exprs": [{"expr": col(f["name"], vschema), "name": f["name"]} for f in vschema]})
    outs = (B.gpuq_column * nc)()
    ofields = (B.gpuq_field_info * nc)()
    bufs = []
    for i, f in enumerate(pop.fields):
        d, v = C.c_void_p(), C.c_void_p()
        ctx.check(L.gpuq_buffer_alloc(ctx.h, max(16, k.value * max(f["width"], 1)) + 64, C.byref(d)))
        ctx.check(L.gpuq_buffer_alloc(ctx.h, (k.value + 63) // 64 * 8 + 8, C.byref(v)))
        outs[i].data, outs[i].validity = d, (v if f["nullable"] else None)
        bufs += [d, v]
        L.gpuq_op_output_field(pop.h, i, C.byref(ofields[i]))
    vin = B.gpuq_input(); vin.cols = C.cast(cols, C.POINTER(B.gpuq_column)); vin.n_cols = nc; vin.n_via = 1; vin.n_rows = k.value; vin.via[0] = sel.value
    ctx.check(L.gpuq_project_run(pop.h, None, C.byref(vin), outs, nc))
    oa, osch = ArrowArray(), ArrowSchema()
    ctx.check(L.gpuq_export_arrow(ctx.h, None, outs, ofields, nc, k.value, C.addressof(oa), C.addressof(osch)))
    got = pa.RecordBatch._import_from_c(C.addressof(oa), C.addressof(osch))
    import pyarrow.compute as pc
    bt = pa.Table.from_batches([batch])
    m = pc.fill_null(pc.and_(pc.greater(bt["k"], 0), pc.not_equal(bt["s"], "MAIL")), False)
    exp = bt.filter(m)
    assert got.num_rows == exp.num_rows
    for name in bt.schema.names:
        assert got.column(name).to_pylist() == exp[name].to_pylist(), name
    for b in bufs + [sel, cnt]:
        L.gpuq_buffer_free(ctx.h, b)
    L.gpuq_table_free(h)


def test_import_refuses_offsets_that_decrease(tc):
    """The Utf8 offsets of an imported batch are the producer's: a decreasing pair must be an error, not a copy length."""
    ctx, L = tc.ctx, tc.ctx.L
    good = pa.array(["ab", "cde", "", "f"])
    bufs = good.buffers()
    bad_offsets = np.array([0, 2, 5, 4, 6], dtype=np.int32)          # 5 -> 4
    bad = pa.Array.from_buffers(pa.string(), 4, [bufs[0], pa.py_buffer(bad_offsets.tobytes()), bufs[2]])
    batch = pa.RecordBatch.from_arrays([bad], ["s"])
    ca, cs = ArrowArray(), ArrowSchema()
    batch._export_to_c(C.addressof(ca), C.addressof(cs))
    h = C.c_void_p()
    rc = L.gpuq_table_import_arrow(ctx.h, None, C.addressof(ca), C.addressof(cs), C.byref(h))
    assert rc == 1 and b"non-decreasing" in L.gpuq_last_error(ctx.h)
