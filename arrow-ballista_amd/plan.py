"""Host-side mirror of the reference's operator interface for the hot path.

Class and argument names follow DataFusion's ExecutionPlan implementations as Ballista names them
(ballista/core/src/physical_optimizer/task_group.rs:23-31, ballista/core/src/utils.rs:40-48) and
the stage driver of ballista/core/src/execution_plans/shuffle_writer.rs:234-456 /
ballista/executor/src/execution_engine.rs:34-133, so that a test written against the reference
reads the same here.  `execute(partition, context)` returns the partition's whole result as a
DeviceTable (possibly a late-materialised view): on a 288 GB HBM device a task's partition is
processed as whole columns by chip-filling launches, not as a stream of 8192-row batches.

All operator work is done by libgpuq.so; this file only plans launches (operator fusion by
expression inlining, index-vector composition, output allocation).
"""
import ctypes as C
import json
import os
import time
import uuid

from . import binding as B
from . import expr as E
from .table import DeviceColumn, DeviceTable, arrow_type_json, record_layout, type_id, type_json, type_width

NULL_ROW = 0xFFFFFFFF


def _torch():
    import torch
    return torch


class TaskContext:
    """Mirror of datafusion::execution::TaskContext for this engine: device context, op cache, config."""

    def __init__(self, ctx=None, device=0, batch_size=8192, task_id="task", session_id="session"):
        self.ctx = ctx if ctx is not None else B.Context(device)
        self.device = "cuda:%d" % self.ctx.device
        self.batch_size = batch_size
        self.task_id, self.session_id = task_id, session_id
        self._ops = {}
        self._memo = {}      # (kind, plan-node / expression identity, table signature) -> compiled Op

    def op(self, descriptor):
        key = json.dumps(descriptor, sort_keys=True)
        o = self._ops.get(key)
        if o is None:
            o = B.Op(self.ctx, descriptor)
            self._ops[key] = o
        return o

    def stream_ptr(self):
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream(self.ctx.device).cuda_stream)

    def sync(self):
        _torch().cuda.current_stream(self.ctx.device).synchronize()


def table_sig(table):
    """Cheap hashable signature of a table's layout (what a compiled operator depends on)."""
    return (len(table.via), table.dense) + tuple((c.name, c.type if isinstance(c.type, str) else (c.type["Decimal128"][0], c.type["Decimal128"][1]),
                                      c.nullable, sd, c.repr) for c, sd in zip(table.columns, table.sides))


class Metrics:
    def __init__(self):
        self.output_rows = 0
        self.elapsed_compute_ns = 0
        self.extra = {}

    def as_dict(self):
        d = {"output_rows": self.output_rows, "elapsed_compute": self.elapsed_compute_ns}
        d.update(self.extra)
        return d


# ---------------------------------------------------------------------------------- helpers
def _alloc_outputs(op, n, device):
    """Allocate caller-side output columns for op.fields with capacity n rows: one device allocation laid out by
    table.record_layout.  Returns (columns, C array, record buffer)."""
    torch = _torch()
    nf = len(op.fields)
    lay = op.__dict__.get("_out_layout")
    if lay is None:
        # per field: (type json, width (0 = Boolean), nullable, repr, name)
        lay = op._out_layout = [(type_json(f["type"], f["precision"], f["scale"]), 0 if f["type"] == B.T_BOOL else f["width"], f["nullable"], f["repr"], f["name"]) for f in op.fields]
    total, pieces = record_layout([(w, nl) for _, w, nl, _, _ in lay], n)
    buf = torch.empty(total, dtype=torch.uint8, device=device)
    cols, arr = [], (B.gpuq_column * max(1, nf))()
    for i, ((tj, width, nullable, rp, name), (doff, dbytes, voff, vbytes)) in enumerate(zip(lay, pieces)):
        data = buf[doff: doff + dbytes]
        validity = buf[voff: voff + vbytes] if nullable else None       # every validity word is written by the kernel
        col = DeviceColumn(name, tj, data, n, validity=validity, nullable=nullable, repr=rp)
        cols.append(col)
        arr[i] = col.to_c()
    return cols, arr, buf


def lower_like(tc, table, exprs):
    """LIKE is evaluated by its own kernel over the Arrow-layout bytes (gpuq_like_utf8), not inside the register-based
    expression programs: every like_expr node of `exprs` becomes a reference to a Boolean column appended to (a copy of)
    the table.  The operand must be a column of `table`; the pattern a Utf8 literal.  Returns (table, exprs)."""
    if not any(E.has_like(e) for e in exprs):
        return table, exprs
    torch = _torch()
    found = []

    def repl(v):
        operand, pat = v["expr"], v["pattern"]
        if not (isinstance(operand, dict) and "column" in operand):
            raise B.GpuqError(3, "LIKE over a computed expression is not supported on device (operand must be a column)")
        if not (isinstance(pat, dict) and "literal" in pat and pat["literal"].get("type") == "Utf8" and pat["literal"].get("value") is not None):
            raise B.GpuqError(3, "LIKE needs a non-NULL Utf8 literal pattern")
        found.append((operand["column"]["name"], pat["literal"]["value"], bool(v.get("negated")), bool(v.get("case_insensitive"))))
        return {"column": {"name": "__like_%d" % (len(found) - 1)}}
    new_exprs = [E.rewrite_like(e, repl) for e in exprs]
    cols, sides = list(table.columns), list(table.sides)
    n = table.num_rows
    names = [c.name for c in table.columns]
    for k, (name, pattern, negated, ci) in enumerate(found):
        if name not in names:
            raise KeyError("column '%s' not in schema %s" % (name, names))
        i = names.index(name)
        c, sd = table.columns[i], table.sides[i]
        if c.repr == B.REPR_PACKED15:      # a string produced by an operator: back to offsets + bytes first
            m = c.length
            off = torch.zeros(m + 4, dtype=torch.int32, device=tc.device)
            dat = torch.zeros(max(16, m * 15), dtype=torch.uint8, device=tc.device)
            dl = C.c_int64(0)
            tc.ctx.check(tc.ctx.L.gpuq_unpack_utf8(tc.ctx.h, tc.stream_ptr(), c.data.data_ptr() if m else None, m, off.data_ptr(), dat.data_ptr(),
                                                   dat.numel(), C.byref(dl)))
            c = DeviceColumn(c.name, c.type, dat, m, offsets=off, validity=c.validity, nullable=c.nullable)
        nb = ((n + 63) // 64) * 8 + 8
        bits = torch.zeros(nb, dtype=torch.uint8, device=tc.device)
        nullable = bool(c.nullable or (sd > 0 and not table.dense))
        valid = torch.zeros(nb, dtype=torch.uint8, device=tc.device) if nullable else None
        idx = table.via[sd - 1] if sd > 0 else None
        tc.ctx.check(tc.ctx.L.gpuq_like_utf8(tc.ctx.h, tc.stream_ptr(), C.byref(c.to_c()), idx.data_ptr() if idx is not None else None, n,
                                             pattern.encode(), int(negated), int(ci), bits.data_ptr(), valid.data_ptr() if valid is not None else None))
        cols.append(DeviceColumn("__like_%d" % k, "Boolean", bits, n, validity=valid, nullable=nullable))
        sides.append(0)
    return DeviceTable(cols, n, via=table.via, sides=sides, dense=table.dense), new_exprs


def _project(tc, table, exprs, names, memo_key=None, memo=None):
    """ProjectionExec kernel call: evaluate `exprs` over (a view of) `table` into a materialised table.
    memo_key: hashable identity of (exprs, names) so the compiled operator is found without rebuilding the descriptor."""
    op = None
    if not callable(exprs):
        table, exprs = lower_like(tc, table, exprs)
    if memo is None:
        memo = tc._memo
    if memo_key is not None:
        mk = ("project", id(tc), memo_key, table_sig(table))
        op = memo.get(mk)
    if op is None:
        schema = table.schema()
        if callable(exprs):
            exprs, names = exprs()
        desc = {"op": "project", "input": {"fields": schema},
                "exprs": [{"expr": E.rebind(e, schema), "name": n} for e, n in zip(exprs, names)]}
        op = tc.op(desc)
        if memo_key is not None:
            memo[mk] = op
    n = table.num_rows
    cols, arr, _ = _alloc_outputs(op, n, tc.device)
    inp, keep = table.input_struct()
    tc.ctx.check(tc.ctx.L.gpuq_project_run(op.h, tc.stream_ptr(), C.byref(inp), arr, len(cols)))
    # a computed expression over a table that holds strings: read the status word (as plan_exec.cpp's project does)
    if not callable(exprs) and any(c.type == "Utf8" for c in table.columns) and any(not (isinstance(e, dict) and len(e) == 1 and ("column" in e or "literal" in e)) for e in exprs):
        tc.ctx.check(tc.ctx.L.gpuq_op_check(op.h, tc.stream_ptr()))
    return DeviceTable(cols, n)


def _take_u32(tc, vec, idx, n):
    """new[j] = vec[idx[j]] with NULL_ROW propagated; vec/idx are int32-typed uint32 tensors."""
    src = DeviceTable([DeviceColumn("v", "UInt32", vec, vec.numel())], n, via=[idx], sides=[1])
    out = _project(tc, src, lambda: ([E.case([(E.is_null(E.col("v", index=0)), E.lit(NULL_ROW, "UInt32"))], E.col("v", index=0))], ["v"]), None, memo_key="take_u32")
    return out.columns[0].data[: max(1, n) * 4].view(_torch().int32)[:n]


def _reindex(tc, table, idx, n):
    """Address `table`'s rows through idx[0..n): returns (columns, index vectors, sides).  A view may mix
    columns read at the driving position (side 0) with columns read through index vectors; the former
    get `idx` itself as their vector, the latter get their vector composed with idx."""
    if not table.is_view():
        return table.columns, [idx], [1] * len(table.columns)
    has0 = any(s == 0 for s in table.sides)
    if len(table.via) + (1 if has0 else 0) > 3:
        table = materialize(tc, table)
        return table.columns, [idx], [1] * len(table.columns)
    vias = ([idx] if has0 else []) + [_take_u32(tc, v, idx, n) for v in table.via]
    shift = 1 if has0 else 0
    return table.columns, vias, [1 if s == 0 else s + shift for s in table.sides]


def _select_view(tc, table, idx, n):
    """View of `table` at driving positions idx[0..n)."""
    cols, vias, sides = _reindex(tc, table, idx, n)
    return DeviceTable(cols, n, via=vias, sides=sides)


def _take_utf8(tc, col, idx, n, nullable):
    """Arrow-layout Utf8 column read through idx (None: identity) into a fresh Arrow-layout column: strings of any length."""
    torch = _torch()
    L = tc.ctx.L
    offs = torch.empty(n + 4, dtype=torch.int32, device=tc.device)
    valid = torch.empty(((n + 63) // 64) * 8 + 8, dtype=torch.uint8, device=tc.device)
    cc = col.to_c()
    dl = C.c_int64(0)
    ip = idx.data_ptr() if idx is not None and idx.numel() > 0 else None
    rc = L.gpuq_take_utf8(tc.ctx.h, tc.stream_ptr(), C.byref(cc), ip, n, offs.data_ptr(), valid.data_ptr(), None, 0, C.byref(dl))
    if rc not in (0, 4):
        tc.ctx.check(rc)
    data = torch.empty(dl.value + 16, dtype=torch.uint8, device=tc.device)
    if dl.value > 0:
        tc.ctx.check(L.gpuq_take_utf8(tc.ctx.h, tc.stream_ptr(), C.byref(cc), ip, n, offs.data_ptr(), valid.data_ptr(), data.data_ptr(), dl.value + 16, C.byref(dl)))
    return DeviceColumn(col.name, "Utf8", data, n, offsets=offs, validity=valid if nullable else None, nullable=nullable, repr=B.REPR_ARROW)


def materialize(tc, table, force=False, pack_strings=None):
    """Gather a late-materialised view into plain columns (<= 12 columns per kernel call).
    Utf8 columns in Arrow layout (offsets + bytes) are taken as they are, whatever their length (gpuq_take_utf8); with
    pack_strings (the default under force=True: concat / exchange need fixed-width columns) every string is re-encoded as
    PACKED15, which holds 15 bytes and refuses longer values."""
    if pack_strings is None:
        pack_strings = force
    if not table.is_view() and not force:
        return table
    out = [None] * len(table.columns)
    fixed = []
    for i, (c, sd) in enumerate(zip(table.columns, table.sides)):
        if c.offsets is not None and not pack_strings:
            if not table.is_view():
                out[i] = c
            else:
                out[i] = _take_utf8(tc, c, table.via[sd - 1] if sd > 0 else None, table.num_rows, bool(c.nullable or (sd > 0 and not table.dense)))
        else:
            fixed.append(i)
    for a in range(0, len(fixed), 12):
        idxs = fixed[a: a + 12]
        sub = DeviceTable([table.columns[i] for i in idxs], table.num_rows, via=table.via, sides=[table.sides[i] for i in idxs], dense=table.dense)
        def mk(sub=sub):
            ss = sub.plain_schema()
            return [E.col(f["name"], index=i) for i, f in enumerate(ss)], [f["name"] for f in ss]
        for i, c in zip(idxs, _project(tc, sub, mk, None, memo_key="materialize").columns):
            out[i] = c
    return DeviceTable(out, table.num_rows)


def concat_tables(tc, tables):
    """Fan-in of partitions into one table (row order = partition order, then row order inside a partition).
    Views are materialised first; fixed-width data buffers are joined with device copies (torch.cat), validity / Boolean
    bitmaps with gpuq_concat_bitmap, Arrow-layout strings by re-basing their offsets (gpuq_offsets_rebase) and joining
    their bytes."""
    torch = _torch()
    tables = list(tables)
    live = [t for t in tables if t.num_rows > 0]
    if len(live) == 1 and not live[0].is_view():
        return live[0]
    if not live:
        live = tables[:1]
    parts = [materialize(tc, t) for t in live]
    n = sum(p.num_rows for p in parts)
    cols = []
    L = tc.ctx.L

    def bitmap(pieces):      # pieces: [(uint8 tensor | None, n_bits)]
        out = torch.zeros(((n + 63) // 64) * 8 + 8, dtype=torch.uint8, device=tc.device)
        off = 0
        for src, nb in pieces:
            if nb > 0:
                tc.ctx.check(L.gpuq_concat_bitmap(tc.ctx.h, tc.stream_ptr(), out.data_ptr(), off, src.data_ptr() if src is not None else None, nb))
            off += nb
        return out
    for i, c0 in enumerate(parts[0].columns):
        pcs = [p.columns[i] for p in parts]
        if any(c.repr != c0.repr or c.type != c0.type for c in pcs):
            raise B.GpuqError(2, "concat: column %r has different layouts across partitions" % c0.name)
        w = type_width(c0.type)
        offsets = None
        if c0.offsets is not None:      # Utf8 in Arrow layout: any string length
            if any(c.offsets is None for c in pcs):
                raise B.GpuqError(2, "concat: column %r has different layouts across partitions" % c0.name)
            ends = torch.stack([torch.stack([c.offsets[0], c.offsets[p.num_rows]]) for c, p in zip(pcs, parts)]).tolist()
            offsets = torch.empty(n + 4, dtype=torch.int32, device=tc.device)
            row, base, chunks = 0, 0, []
            for (first, last), c, p in zip(ends, pcs, parts):
                tc.ctx.check(L.gpuq_offsets_rebase(tc.ctx.h, tc.stream_ptr(), c.offsets.data_ptr(), p.num_rows + 1, base - first, offsets.data_ptr() + 4 * row))
                chunks.append(c.data[first:last])
                row += p.num_rows
                base += last - first
            data = torch.cat(chunks + [torch.zeros(16, dtype=torch.uint8, device=tc.device)])
        elif w == 0:           # Boolean: bit-packed
            data = bitmap([(c.data, p.num_rows) for c, p in zip(pcs, parts)])
        else:
            data = torch.cat([c.data[: p.num_rows * w] for c, p in zip(pcs, parts)] + [torch.zeros(16, dtype=torch.uint8, device=tc.device)])
        nullable = any(c.nullable for c in pcs)
        validity = bitmap([(c.validity, p.num_rows) for c, p in zip(pcs, parts)]) if any(c.validity is not None for c in pcs) else None
        cols.append(DeviceColumn(c0.name, c0.type, data, n, offsets=offsets, validity=validity, nullable=nullable, repr=c0.repr))
    return DeviceTable(cols, n)


def sort_table(tc, table, sort_expr, fetch=None, memo=None):
    """Stable sort of one table by [{"expr","asc","nulls_first"}]: returns a view (permutation applied lazily)."""
    torch = _torch()
    memo = tc._memo if memo is None else memo
    mk = ("sort", id(tc), repr(sort_expr) if memo is tc._memo else None, table_sig(table))
    op = memo.get(mk)
    if op is None:
        schema = table.schema()
        desc = {"op": "sort", "input": {"fields": schema},
                "expr": [{"expr": E.rebind(s["expr"], schema), "asc": bool(s.get("asc", True)),
                          "nulls_first": bool(s.get("nulls_first", not s.get("asc", True)))} for s in sort_expr]}
        op = memo[mk] = tc.op(desc)
    n = table.num_rows
    perm = torch.empty(max(1, n), dtype=torch.int32, device=tc.device)
    inp, keep = table.input_struct()
    tc.ctx.check(tc.ctx.L.gpuq_sort_run(op.h, tc.stream_ptr(), C.byref(inp), perm.data_ptr()))
    k = n if fetch is None or fetch < 0 else min(n, int(fetch))
    return _select_view(tc, table, perm[:k], k)


def merge_tables(tc, parts, sort_expr, fetch=None, memo=None, pre_concatenated=None):
    """Ordered fan-in of tables that are each sorted by `sort_expr` (gpuq_merge_run): the merged order as a view over their
    concatenation; ties keep (part, row) order.  pre_concatenated: the table the parts are consecutive slices of, when the
    caller already holds it (the runs received by an exchange); `parts` may then be plain row counts."""
    torch = _torch()
    parts = [p for p in parts]
    table = pre_concatenated if pre_concatenated is not None else concat_tables(tc, parts)
    if len(parts) <= 1:      # one partition (it may hold several runs, e.g. gathered by BroadcastExec): sort
        return sort_table(tc, table, sort_expr, fetch, memo=memo)
    memo = tc._memo if memo is None else memo
    mk = ("sort", id(tc), repr(sort_expr) if memo is tc._memo else None, table_sig(table))
    op = memo.get(mk)
    if op is None:
        schema = table.schema()
        desc = {"op": "sort", "input": {"fields": schema},
                "expr": [{"expr": E.rebind(s["expr"], schema), "asc": bool(s.get("asc", True)),
                          "nulls_first": bool(s.get("nulls_first", not s.get("asc", True)))} for s in sort_expr]}
        op = memo[mk] = tc.op(desc)
    n = table.num_rows
    offs = [0]
    for p in parts:
        offs.append(offs[-1] + (p if isinstance(p, int) else p.num_rows))
    perm = torch.empty(max(1, n), dtype=torch.int32, device=tc.device)
    inp, keep = table.input_struct()
    tc.ctx.check(tc.ctx.L.gpuq_merge_run(op.h, tc.stream_ptr(), C.byref(inp), (C.c_int64 * len(offs))(*offs), len(parts), perm.data_ptr()))
    k = n if fetch is None or fetch < 0 else min(n, int(fetch))
    return _select_view(tc, table, perm[:k], k)


def slice_table(tc, table, skip, fetch):
    """Rows [skip, skip+fetch) of a table as a view (fetch None: to the end)."""
    torch = _torch()
    n = table.num_rows
    lo = min(max(0, int(skip)), n)
    hi = n if fetch is None or fetch < 0 else min(n, lo + int(fetch))
    if lo == 0 and hi == n:
        return table
    idx = torch.arange(lo, max(hi, lo + 1), dtype=torch.int32, device=tc.device)
    return _select_view(tc, table, idx[: hi - lo], hi - lo)


def _fuse(plan):
    """Walk down through Filter / Projection / CoalesceBatches: returns (source_plan, predicate, colmap).
    colmap maps an output column name of `plan` to an expression over source columns (None = identity)."""
    if isinstance(plan, CoalesceBatchesExec):
        return _fuse(plan.input)
    if isinstance(plan, FilterExec):
        if E.has_like(plan.predicate):
            return plan, None, None          # LIKE runs as its own kernel over the filter's input: not inlined into consumers
        src, p, m = _fuse(plan.input)
        mine = E.inline_projection(plan.predicate, m) if m else plan.predicate
        return src, (E.and_(p, mine) if p is not None else mine), m
    if isinstance(plan, ProjectionExec):
        if any(E.has_like(e) for e, _ in plan.expr):
            return plan, None, None
        src, p, m = _fuse(plan.input)
        new = {name: (E.inline_projection(e, m) if m else e) for e, name in plan.expr}
        return src, p, new
    return plan, None, None


def _inl(e, m):
    return E.inline_projection(e, m) if m else e


# ---------------------------------------------------------------------------------- plan nodes
def _plan_layer():
    return os.environ.get("GPUQ_PLAN_LAYER", "native")


def _native_execute(node, partition, context, mirror):
    """`node.execute` through the native executor: the plan is serialised once per (node, context) and kept."""
    from . import native as N
    cache = node.__dict__.setdefault("_native_plans", {})
    np_ = cache.get(id(context))
    if np_ is None:
        try:
            np_ = N.NativePlan(node, context)
        except B.GpuqError as e:
            if "not executed natively" in str(e):      # a node only the mirror implements
                cache[id(context)] = False
                return mirror(node, partition, context)
            raise
        cache[id(context)] = np_
    if np_ is False:
        return mirror(node, partition, context)
    t0 = time.perf_counter()
    res = np_.execute(partition)
    return node._timed(t0, res.to_device_table(context.device))


class ExecutionPlan:
    def __init__(self):
        self.metrics = Metrics()
        self._memo = {}       # compiled operators of this node, keyed by (context, input table signature)

    def children(self):
        return []

    def schema(self):
        raise NotImplementedError

    def output_partition_count(self):
        ch = self.children()
        return ch[0].output_partition_count() if ch else 1

    def execute(self, partition, context):
        raise NotImplementedError

    # ---- one plan layer.  The classes below are (a) the plan description a client builds -- what the native executor serialises and
    # runs (native.py: NativePlan) -- and (b) a Python restatement of the executor over the same C entry points ("the mirror"), which
    # most operator tests were written against.  `node.execute(partition, context)` itself goes through the NATIVE executor (the whole
    # sub-tree as one gpuq_plan_execute; round 3: the default), so that those tests exercise the product's plan layer;
    # GPUQ_PLAN_LAYER=mirror runs the restatement instead (the tests that are about the mirror itself ask for it: tests/conftest.py
    # `mirror_layer`), and a node only the mirror implements falls back to it.  Leaves (MemoryExec) return their table either way.
    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        own = cls.__dict__.get("execute")
        if own is None or getattr(own, "_dispatch", False):
            return
        cls._mirror_execute = own

        def execute(self, partition, context, _own=own):
            if _plan_layer() != "native" or isinstance(self, MemoryExec) or getattr(context, "_in_native", False):
                return _own(self, partition, context)
            return _native_execute(self, partition, context, _own)
        execute._dispatch = True
        execute.__doc__ = own.__doc__
        cls.execute = execute

    def _timed(self, t0, table):
        self.metrics.elapsed_compute_ns += int((time.perf_counter() - t0) * 1e9)
        self.metrics.output_rows += table.num_rows
        return table

    def __str__(self):
        return type(self).__name__


class MemoryExec(ExecutionPlan):
    """datafusion MemoryExec: partitions of pyarrow RecordBatches/Tables (uploaded on first use) or DeviceTables."""

    def __init__(self, partitions, schema=None):
        super().__init__()
        self.partitions = list(partitions)
        self._dev = {}
        self._schema = schema

    def output_partition_count(self):
        return len(self.partitions)

    def schema(self):
        if self._schema is None:
            p = self.partitions[0]
            if isinstance(p, DeviceTable):
                self._schema = p.plain_schema()
            else:
                import pyarrow as pa
                t = pa.Table.from_batches(p) if isinstance(p, (list, tuple)) else p
                self._schema = _arrow_schema(t.schema)
        return self._schema

    def execute(self, partition, context):
        t0 = time.perf_counter()
        p = self.partitions[partition]
        if isinstance(p, DeviceTable):
            return self._timed(t0, p)
        if partition not in self._dev:
            import pyarrow as pa
            t = pa.Table.from_batches(p) if isinstance(p, (list, tuple)) else p
            self._dev[partition] = DeviceTable.from_arrow(t, context.device)
        return self._timed(t0, self._dev[partition])


def _arrow_schema(s):
    import pyarrow as pa
    out = []
    for f in s:
        tj = arrow_type_json(f.type)
        if tj is None:
            raise B.GpuqError(3, "Arrow type %s is not supported on device" % f.type)
        out.append({"name": f.name, "type": tj, "nullable": f.nullable})
    return out


class CoalesceBatchesExec(ExecutionPlan):
    """Pass-through: the device engine already works on whole partitions (datafusion.proto:1487-1490)."""

    def __init__(self, input, target_batch_size=8192):
        super().__init__()
        self.input, self.target_batch_size = input, target_batch_size

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def execute(self, partition, context):
        return self.input.execute(partition, context)


class FilterExec(ExecutionPlan):
    """FilterExec(predicate, input) -- datafusion.proto:1291-1294.  Returns a selection view in input order."""

    def __init__(self, predicate, input):
        super().__init__()
        self.predicate, self.input = E.lower_long_string_eq(predicate), input

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def execute(self, partition, context):
        table = self.input.execute(partition, context)
        t0 = time.perf_counter()
        return self._timed(t0, filter_table(context, table, self.predicate, memo_key="self", memo=self._memo))


def filter_table(tc, table, predicate, memo_key=None, memo=None, return_sel=False):
    torch = _torch()
    op = None
    source = table
    table, (predicate,) = lower_like(tc, table, [predicate])
    if memo is None:
        memo = tc._memo
    if memo_key is not None:
        mk = ("filter", id(tc), memo_key, table_sig(table))
        op = memo.get(mk)
    if op is None:
        schema = table.schema()
        op = tc.op({"op": "filter", "input": {"fields": schema}, "predicate": E.rebind(predicate, schema)})
        if memo_key is not None:
            memo[mk] = op
    n = table.num_rows
    sel = torch.empty(max(1, n), dtype=torch.int32, device=tc.device)
    cnt = torch.zeros(2, dtype=torch.int64, device=tc.device)
    inp, keep = table.input_struct()
    tc.ctx.check(tc.ctx.L.gpuq_filter_run(op.h, tc.stream_ptr(), C.byref(inp), 0, sel.data_ptr(), cnt.data_ptr()))
    k = int(cnt[0].item())
    op.check(tc.stream_ptr())
    if return_sel:
        return sel[:k], k
    return _select_view(tc, source, sel[:k], k)


class ProjectionExec(ExecutionPlan):
    """ProjectionExec(expr: [(PhysicalExpr, name)], input) -- datafusion.proto:1399-1403."""

    def __init__(self, expr, input):
        super().__init__()
        self.expr, self.input = [(E.lower_long_string_eq(e), n) for e, n in expr], input

    def children(self):
        return [self.input]

    def schema(self):
        d = B.compile_check({"op": "project", "input": {"fields": self.input.schema()},
                             "exprs": [{"expr": E.rebind(E.like_placeholder(e), self.input.schema()), "name": n} for e, n in self.expr]})
        return [_field_from_desc(o) for o in d["outputs"]]

    def execute(self, partition, context):
        if not hasattr(self, "_fused"):
            self._fused = _fuse(self)          # plan nodes are immutable once built: fuse once
            self._like = self._fused[0] is self
            if self._like:                      # own expressions hold a LIKE: fuse what lies below, evaluate this node by itself
                s2, p2, m2 = _fuse(self.input)
                self._fused = (s2, p2, {name: (E.inline_projection(e, m2) if m2 else e) for e, name in self.expr})
        src, pred, m = self._fused
        table = src.execute(partition, context)
        t0 = time.perf_counter()
        if pred is not None:
            table = filter_table(context, table, pred, memo_key="pf", memo=self._memo)
        if self._like:
            return self._timed(t0, _project(context, table, [m[name] for _, name in self.expr], [name for _, name in self.expr], memo_key="pl", memo=self._memo))
        out = _project(context, table, lambda: ([m[name] for _, name in self.expr], [name for _, name in self.expr]), None, memo_key="pe", memo=self._memo)
        return self._timed(t0, out)


def _field_from_desc(o):
    t = o["type"]
    if t.startswith("Decimal128("):
        p, s = t[len("Decimal128("):-1].split(",")
        t = {"Decimal128": [int(p), int(s)]}
    return {"name": o["name"], "type": t, "nullable": bool(o["nullable"])}


class AggregateExec(ExecutionPlan):
    """AggregateExec(mode, group_expr, aggr_expr, input) -- datafusion.proto:1405-1450.
    group_expr: [(expr, name)];  aggr_expr: [{"fn": "SUM"|"AVG"|"COUNT"|"MIN"|"MAX"|"VARIANCE[_POP]"|"STDDEV[_POP]"|
    "COVARIANCE[_POP]"|"CORRELATION", "expr": e[, "expr2": e2], "name": n}]  (function names: datafusion.proto:631-669).
    Modes: Partial (emits state columns), Final / FinalPartitioned (merge states), Single."""

    def __init__(self, mode, group_expr, aggr_expr, input, strategy="auto", expected_groups=0):
        super().__init__()
        self.mode, self.group_expr, self.aggr_expr, self.input = mode, list(group_expr), list(aggr_expr), input
        self.strategy, self.expected_groups = strategy, expected_groups

    def children(self):
        return [self.input]

    def _descriptor(self, schema, pred, m):
        d = {"op": "aggregate", "mode": self.mode, "input": {"fields": schema}, "strategy": self.strategy,
             "group_expr": [{"expr": E.rebind(_inl(e, m), schema), "name": n} for e, n in self.group_expr],
             "aggr_expr": [dict(fn=a["fn"], name=a["name"], **{k: E.rebind(_inl(a[k], m), schema) for k in ("expr", "expr2", "filter") if a.get(k) is not None})
                           for a in self.aggr_expr]}
        if pred is not None:
            d["predicate"] = E.rebind(pred, schema)
        if self.expected_groups:
            d["expected_groups"] = int(self.expected_groups)
        return d

    def schema(self):
        d = B.compile_check(self._descriptor(self.input.schema(), None, None))
        return [_field_from_desc(o) for o in d["outputs"]]

    def execute(self, partition, context):
        final = self.mode in ("Final", "FinalPartitioned")
        fused_for = getattr(self, "_fused_for", None)
        if fused_for is not self.input:
            self._fused = (self.input, None, None) if final else _fuse(self.input)
            self._fused_for = self.input
        src, pred, m = self._fused
        table = src.execute(partition, context)
        t0 = time.perf_counter()
        mk = ("agg", id(context), id(self.input), table_sig(table))
        op = self._memo.get(mk)
        if op is None:
            op = self._memo[mk] = context.op(self._descriptor(table.schema(), pred, m))
        out = aggregate_table(context, table, op.descriptor, cap=getattr(self, "output_capacity", None), op=op)
        return self._timed(t0, out)


def aggregate_table(tc, table, descriptor, cap=None, op=None):
    op = op if op is not None else tc.op(descriptor)
    n = table.num_rows
    if cap is None:
        cap = 4096 if not descriptor["group_expr"] else max(4096, min(n, 1 << 22))
        eg = int(descriptor.get("expected_groups", 0) or 0)
        if eg > 0:          # a cardinality hint also sizes the output: a too-small capacity costs a second run of the aggregate
            cap = max(cap, min(n, eg + eg // 4))
    inp, keep = table.input_struct()
    while True:
        cols, arr, buf = _alloc_outputs(op, cap, tc.device)
        ng = C.c_int64(0)
        rc = tc.ctx.L.gpuq_aggregate_run(op.h, tc.stream_ptr(), C.byref(inp), arr, len(cols), cap, C.byref(ng))
        if rc == 4 and ng.value > cap:
            cap = int(ng.value)
            continue
        tc.ctx.check(rc)
        break
    for c in cols:
        c.set_length(ng.value)
    out = DeviceTable(cols, ng.value)
    out._record = (buf, cap)          # all columns live in one allocation of known layout (parallel.allgather_table)
    return out


class HashJoinExec(ExecutionPlan):

    """HashJoinExec(left, right, on: [(left_col, right_col)], filter, join_type, partition_mode, null_equals_null)
    -- ctor surface of ballista/core/src/physical_optimizer/task_group.rs:306-315, datafusion.proto:1346-1360.
    The LEFT input is the build side.  Output columns = left columns then right columns (a view)."""

    def __init__(self, left, right, on, filter=None, join_type="Inner", partition_mode="CollectLeft", null_equals_null=False):
        super().__init__()
        self.left, self.right, self.on, self.filter = left, right, list(on), filter
        self.join_type, self.partition_mode, self.null_equals_null = join_type, partition_mode, null_equals_null

    def children(self):
        return [self.left, self.right]

    def output_partition_count(self):
        return self.right.output_partition_count()

    def _residual_join(self, tc, ltab, rtab, pairs, ob, opb, k):
        return _residual_join_impl(tc, self.join_type, self.filter, ltab, rtab, pairs, ob, opb, k)

    def schema(self):
        ls, rs = self.left.schema(), self.right.schema()
        jt = self.join_type
        if jt in ("LeftSemi", "LeftAnti"):
            return ls
        if jt in ("RightSemi", "RightAnti"):
            return rs
        ln = jt in ("Right", "Full")
        rn = jt in ("Left", "Full")
        return [dict(f, nullable=f["nullable"] or ln) for f in ls] + [dict(f, nullable=f["nullable"] or rn) for f in rs]

    def _side(self, plan, partition, context, key_exprs):
        src, pred, m = _fuse(plan)
        if m is not None:
            # computed projection below the join: execute it (materialised) and do not fuse
            table = plan.execute(partition, context)
            return table, None, key_exprs
        table = src.execute(partition, context)
        return table, pred, key_exprs

    def execute(self, partition, context):
        torch = _torch()
        tc = context
        jt = self.join_type
        lkeys = [l for l, _ in self.on]
        rkeys = [r for _, r in self.on]
        # CollectLeft: every probe partition sees the whole build side; Partitioned: co-partitioned inputs
        lpart = partition if self.partition_mode == "Partitioned" else None
        if lpart is None:
            cache = getattr(self, "_build_cache", None)
            if cache is None:
                parts = [self._side(self.left, p, tc, lkeys) for p in range(self.left.output_partition_count())]
                if len(parts) != 1:
                    raise B.GpuqError(3, "CollectLeft with a multi-partition build side: wrap the left input in a single partition")
                cache = self._build_cache = parts[0]
            ltab, lpred, _ = cache
        else:
            ltab, lpred, _ = self._side(self.left, lpart, tc, lkeys)
        rtab, rpred, _ = self._side(self.right, partition, tc, rkeys)
        t0 = time.perf_counter()
        residual = self.filter is not None and jt != "Inner"
        if residual:
            # a pair that fails the JoinFilter is no match: probe as Inner over the rows that pass the sides' own predicates,
            # filter the pairs, derive the outer / semi / anti parts from the surviving pairs (_residual_join)
            if lpred is not None:
                ltab, lpred = filter_table(tc, ltab, lpred), None
            if rpred is not None:
                rtab, rpred = filter_table(tc, rtab, rpred), None
            jt = "Inner"
        mk = ("join", id(tc), table_sig(ltab), table_sig(rtab), jt)
        ops = self._memo.get(mk)
        lschema, rschema = (None, None) if ops else (ltab.schema(), rtab.schema())
        bdesc = None if ops else {"op": "join_build", "input": {"fields": lschema}, "on": [E.rebind(k, lschema) for k in lkeys],
                 "null_equals_null": bool(self.null_equals_null)}
        if bdesc is not None and lpred is not None:
            bdesc["predicate"] = E.rebind(lpred, lschema)
        pdesc = None if ops else {"op": "join_probe", "input": {"fields": rschema}, "on": [E.rebind(k, rschema) for k in rkeys],
                 "join_type": jt, "null_equals_null": bool(self.null_equals_null)}
        if pdesc is not None and rpred is not None:
            pdesc["predicate"] = E.rebind(rpred, rschema)
        if ops is None:
            ops = self._memo[mk] = (tc.op(bdesc), tc.op(pdesc))
        bop, pop = ops
        linp, lk = ltab.input_struct()
        h = C.c_void_p()
        tc.ctx.check(tc.ctx.L.gpuq_join_build_run(bop.h, tc.stream_ptr(), C.byref(linp), 0, ltab.num_rows, C.byref(h)))
        jtab = B.JoinTable(tc.ctx, h)
        try:
            rinp, rk = rtab.input_struct()
            nprobe = rtab.num_rows
            cap = max(1, nprobe) + (ltab.num_rows if jt in ("Left", "Full") else 0)
            cnt = torch.zeros(2, dtype=torch.int64, device=tc.device)
            while True:
                ob = torch.empty(cap, dtype=torch.int32, device=tc.device)
                opb = torch.empty(cap, dtype=torch.int32, device=tc.device)
                tc.ctx.check(tc.ctx.L.gpuq_join_probe_run(pop.h, tc.stream_ptr(), jtab.h, C.byref(rinp), 0, ob.data_ptr(), opb.data_ptr(),
                                                          cap - (ltab.num_rows if jt in ("Left", "Full") else 0), cnt.data_ptr()))
                k = int(cnt[0].item())
                try:
                    pop.check(tc.stream_ptr())
                    break
                except B.GpuqError as e:
                    if e.status != 4:
                        raise
                    cap = k + (ltab.num_rows if jt in ("Left", "Full") else 0) + 1
            if jt in ("LeftSemi", "LeftAnti", "Left", "Full"):
                extra = torch.zeros(2, dtype=torch.int64, device=tc.device)
                if jt in ("LeftSemi", "LeftAnti"):
                    rows = torch.empty(max(1, ltab.num_rows), dtype=torch.int32, device=tc.device)
                    tc.ctx.check(tc.ctx.L.gpuq_join_build_side_rows(jtab.h, tc.stream_ptr(), 1 if jt == "LeftSemi" else 0, rows.data_ptr(), extra.data_ptr()))
                    m = int(extra[0].item())
                    out = _select_view(tc, ltab, rows[:m], m)
                    return self._timed(t0, out)
                tc.ctx.check(tc.ctx.L.gpuq_join_build_side_rows(jtab.h, tc.stream_ptr(), 0, ob.data_ptr() + 4 * k, extra.data_ptr()))
                m = int(extra[0].item())
                opb[k:k + m] = -1       # NULL_ROW on the probe side
                k += m
            if jt in ("RightSemi", "RightAnti"):
                return self._timed(t0, _select_view(tc, rtab, opb[:k], k))
            out = _join_view(tc, ltab, rtab, ob[:k], opb[:k], k)
            if residual:
                return self._timed(t0, self._residual_join(tc, ltab, rtab, out, ob[:k], opb[:k], k))
            if self.filter is not None:
                out = filter_table(tc, out, self.filter)
            return self._timed(t0, out)
        finally:
            jtab.close()


def _marked_rows(tc, rows, k, n, want_marked):
    """Positions of [0, n) that occur (want_marked) / do not occur in rows[0..k): (int32 tensor, count)."""
    torch = _torch()
    bits = torch.zeros(((n + 63) // 64) * 8 + 8, dtype=torch.uint8, device=tc.device)
    tc.ctx.check(tc.ctx.L.gpuq_mark_rows(tc.ctx.h, tc.stream_ptr(), rows.data_ptr() if k else None, k, bits.data_ptr()))
    t = DeviceTable([DeviceColumn("m", "Boolean", bits, n, nullable=False)], n)
    m = E.col("m", index=0)
    return filter_table(tc, t, m if want_marked else E.not_(m), return_sel=True)


def _residual_join_impl(tc, jt, flt, ltab, rtab, pairs, ob, opb, k):
    torch = _torch()
    sel, k2 = filter_table(tc, pairs, flt, return_sel=True)
    ob2 = _take_u32(tc, ob, sel, k2) if k else ob[:0]
    opb2 = _take_u32(tc, opb, sel, k2) if k else opb[:0]
    nl, nr = ltab.num_rows, rtab.num_rows
    if jt in ("LeftSemi", "LeftAnti"):
        rows, m = _marked_rows(tc, ob2, k2, nl, jt == "LeftSemi")
        return _select_view(tc, ltab, rows, m)
    if jt in ("RightSemi", "RightAnti"):
        rows, m = _marked_rows(tc, opb2, k2, nr, jt == "RightSemi")
        return _select_view(tc, rtab, rows, m)
    lparts, rparts = [ob2], [opb2]
    if jt in ("Left", "Full"):
        rows, m = _marked_rows(tc, ob2, k2, nl, False)
        lparts.append(rows); rparts.append(torch.full((m,), -1, dtype=torch.int32, device=tc.device))
    if jt in ("Right", "Full"):
        rows, m = _marked_rows(tc, opb2, k2, nr, False)
        lparts.append(torch.full((m,), -1, dtype=torch.int32, device=tc.device)); rparts.append(rows)
    ob3, opb3 = torch.cat(lparts), torch.cat(rparts)
    return _join_view(tc, ltab, rtab, ob3, opb3, int(ob3.numel()))


def _join_view(tc, ltab, rtab, ob, opb, k):
    lcols, lv, lsides = _reindex(tc, ltab, ob, k)
    rcols, rv, rsides = _reindex(tc, rtab, opb, k)
    if len(lv) + len(rv) > 3:
        # too many index vectors for one kernel call: materialise the side that carries more of them
        if len(lv) >= len(rv):
            m = materialize(tc, DeviceTable(lcols, k, via=lv, sides=lsides))
            return DeviceTable(m.columns + list(rcols), k, via=rv, sides=[0] * len(m.columns) + list(rsides))
        m = materialize(tc, DeviceTable(rcols, k, via=rv, sides=rsides))
        return DeviceTable(list(lcols) + m.columns, k, via=lv, sides=list(lsides) + [0] * len(m.columns))
    return DeviceTable(list(lcols) + list(rcols), k, via=lv + rv, sides=list(lsides) + [s + len(lv) for s in rsides])


class SortExec(ExecutionPlan):
    """SortExec(expr: [{"expr", "asc", "nulls_first"}], input, fetch) -- datafusion.proto:1465-1471, :1247-1251."""

    def __init__(self, expr, input, fetch=None, preserve_partitioning=False):
        super().__init__()
        self.expr, self.input, self.fetch, self.preserve_partitioning = list(expr), input, fetch, preserve_partitioning

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def execute(self, partition, context):
        table = self.input.execute(partition, context)
        t0 = time.perf_counter()
        return self._timed(t0, sort_table(context, table, self.expr, self.fetch, memo=self._memo))


class SortPreservingMergeExec(ExecutionPlan):
    """SortPreservingMergeExec(expr, input, fetch) -- datafusion.proto:1473-1478: merges the input's sorted partitions
    into one sorted partition.  On the device: pairwise merge-path rounds over the packed composite keys (gpuq_merge_run);
    equal keys keep (partition, row) order, which is what a merge that prefers the lower-numbered stream on ties
    produces [UPSTREAM-KNOWLEDGE: streaming_merge loser tree]."""

    def __init__(self, expr, input, fetch=None):
        super().__init__()
        self.expr, self.input, self.fetch = list(expr), input, fetch

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def output_partition_count(self):
        return 1

    def execute(self, partition, context):
        parts = [self.input.execute(p, context) for p in range(self.input.output_partition_count())]
        t0 = time.perf_counter()
        return self._timed(t0, merge_tables(context, parts, self.expr, self.fetch, memo=self._memo))


class CoalesceTasksExec(ExecutionPlan):
    """CoalesceTasksExec(input, partitions, order_by) -- ballista/core/src/execution_plans/coalesce_tasks.rs:46-70.
    One listed partition: passed through (:143-145).  order_by given: ALL input partitions are merged preserving
    the order (:148-171, the reference iterates 0..input_partitions there, not `partitions`); otherwise the listed
    partitions are concatenated (:172-221; the reference interleaves batches in arrival order, row order across
    partitions is unspecified -- here it is partition order)."""

    def __init__(self, input, partitions, order_by=None):
        super().__init__()
        self.input, self.partitions, self.order_by = input, list(partitions), (list(order_by) if order_by else None)

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def output_partition_count(self):
        return 1

    def execute(self, partition, context):
        if len(self.partitions) == 1:
            return self.input.execute(self.partitions[0], context)
        which = range(self.input.output_partition_count()) if self.order_by else self.partitions
        parts = [self.input.execute(p, context) for p in which]
        t0 = time.perf_counter()
        # ordered: the k-way merge of coalesce_tasks.rs:162-170 (pairwise merge-path rounds on the device)
        out = merge_tables(context, parts, self.order_by, None, memo=self._memo) if self.order_by else concat_tables(context, parts)
        return self._timed(t0, out)

    def __str__(self):
        return "CoalesceTasksExec" + (": sort_expr=%s" % (self.order_by,) if self.order_by else "")


class CoalescePartitionsExec(CoalesceTasksExec):
    """datafusion CoalescePartitionsExec (datafusion.proto:1492-1494): all input partitions -> one, unordered."""

    def __init__(self, input):
        ExecutionPlan.__init__(self)
        self.input, self.order_by = input, None

    @property
    def partitions(self):
        return list(range(self.input.output_partition_count()))

    def __str__(self):
        return "CoalescePartitionsExec"


class CrossJoinExec(ExecutionPlan):
    """CrossJoinExec(left, right) -- datafusion.proto:1382-1385.  Every row of the (collected) left input with every row of the right
    partition; DataFusion plans an uncorrelated scalar subquery this way (a one-row side, then a FilterExec).  Executed by the native
    executor (NativePlan); this mirror only types it."""

    def __init__(self, left, right):
        super().__init__()
        self.left, self.right = left, right

    def children(self):
        return [self.left, self.right]

    def output_partition_count(self):
        return self.right.output_partition_count()

    def schema(self):
        return list(self.left.schema()) + list(self.right.schema())

    def execute(self, partition, context):
        raise B.GpuqError(3, "CrossJoinExec runs in the native executor (NativePlan)")


class UnionExec(ExecutionPlan):
    """UnionExec(inputs) -- datafusion.proto:1319-1321: output partitions are the inputs' partitions, one after another
    (UNION ALL; UNION adds an AggregateExec over all columns, client/src/context.rs:691-733)."""

    def __init__(self, inputs):
        super().__init__()
        self.inputs = list(inputs)

    def children(self):
        return list(self.inputs)

    def schema(self):
        return self.inputs[0].schema()

    def output_partition_count(self):
        return sum(i.output_partition_count() for i in self.inputs)

    def execute(self, partition, context):
        for i in self.inputs:
            k = i.output_partition_count()
            if partition < k:
                t = i.execute(partition, context)
                self.metrics.output_rows += t.num_rows
                return t
            partition -= k
        raise IndexError("UnionExec partition out of range")


class LocalLimitExec(ExecutionPlan):
    """LocalLimitExec(input, fetch) -- datafusion.proto:1460-1463: the first `fetch` rows of EACH partition."""

    def __init__(self, input, fetch):
        super().__init__()
        self.input, self.fetch = input, int(fetch)

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def execute(self, partition, context):
        table = self.input.execute(partition, context)
        t0 = time.perf_counter()
        return self._timed(t0, slice_table(context, table, 0, self.fetch))


class GlobalLimitExec(ExecutionPlan):
    """GlobalLimitExec(input, skip, fetch) -- datafusion.proto:1453-1458: rows [skip, skip+fetch) of a single-partition input."""

    def __init__(self, input, skip=0, fetch=None):
        super().__init__()
        self.input, self.skip, self.fetch = input, int(skip), fetch

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def execute(self, partition, context):
        if self.input.output_partition_count() != 1:
            raise B.GpuqError(1, "GlobalLimitExec requires a single input partition")
        table = self.input.execute(0, context)
        t0 = time.perf_counter()
        return self._timed(t0, slice_table(context, table, self.skip, self.fetch))


class RepartitionExec(ExecutionPlan):
    """Hash repartition of ONE input partition into n outputs (BatchPartitioner call site,
    shuffle_writer.rs:336-391).  execute_all(partition) returns the n per-partition views."""

    def __init__(self, input, hash_expr, partition_count):
        super().__init__()
        self.input, self.hash_expr, self.partition_count = input, list(hash_expr), int(partition_count)

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def execute_all(self, partition, context):
        table = self.input.execute(partition, context)
        t0 = time.perf_counter()
        outs = partition_table(context, table, self.hash_expr, self.partition_count)
        self.metrics.elapsed_compute_ns += int((time.perf_counter() - t0) * 1e9)
        self.metrics.output_rows += table.num_rows
        return outs


class ExchangeExec(ExecutionPlan):
    """Base of the two exchange nodes the NATIVE executor runs across the ranks of a node (csrc/plan_exec.cpp, csrc/exchange.cpp;
    attach the ranks with NativePlan.set_comm): they are plan-description classes only -- the mirror executes single-rank plans."""

    def __init__(self, input):
        super().__init__()
        self.input = input

    def children(self):
        return [self.input]

    def schema(self):
        return self.input.schema()

    def output_partition_count(self):
        return self.input.output_partition_count()

    def execute(self, partition, context):
        raise B.GpuqError(3, "%s runs in the native plan executor (NativePlan + set_comm)" % type(self).__name__)


class RepartitionExchangeExec(ExchangeExec):
    """RepartitionExec(Hash(hash_expr, ranks)) + the exchange: this rank receives the rows of ALL ranks whose key hashes to it
    (the stage boundary of planner.rs:137-151, as one RCCL all-to-all instead of shuffle files)."""

    def __init__(self, input, hash_expr, partition_count):
        super().__init__(input)
        self.hash_expr, self.partition_count = list(hash_expr), int(partition_count)


class BroadcastExec(ExchangeExec):
    """Every rank receives all ranks' rows (a CollectLeft build side read by every reduce task)."""


class RangeRepartitionExec(ExchangeExec):
    """Distributed SortExec (SURVEY.md section 8e "Sort"; the single-partition SortPreservingMergeExec stage of planner.rs:120-136 as
    sample -> splitters -> ONE range exchange -> ordered fan-in of the received runs): this rank ends up with the `expr`-ordered rows
    of one range; rank order is the global order."""

    def __init__(self, input, expr, partition_count, samples=1024):
        super().__init__(input)
        self.expr, self.partition_count, self.samples = list(expr), int(partition_count), int(samples)


def partition_perm(tc, table, hash_expr, partition_count):
    """(perm, offsets): driving positions grouped by partition (input order inside a partition) and the partition_count + 1
    boundaries as a host list -- gpuq_partition_run."""
    torch = _torch()
    schema = table.schema()
    op = tc.op({"op": "partition", "input": {"fields": schema}, "hash_expr": [E.rebind(e, schema) for e in hash_expr],
                "partition_count": int(partition_count)})
    n = table.num_rows
    perm = torch.empty(max(1, n), dtype=torch.int32, device=tc.device)
    offs = torch.zeros(partition_count + 2, dtype=torch.int64, device=tc.device)
    inp, keep = table.input_struct()
    tc.ctx.check(tc.ctx.L.gpuq_partition_run(op.h, tc.stream_ptr(), C.byref(inp), perm.data_ptr(), offs.data_ptr()))
    return perm, offs.cpu().tolist()[: partition_count + 1]


def partition_table(tc, table, hash_expr, partition_count):
    torch = _torch()
    schema = table.schema()
    op = tc.op({"op": "partition", "input": {"fields": schema}, "hash_expr": [E.rebind(e, schema) for e in hash_expr],
                "partition_count": int(partition_count)})
    n = table.num_rows
    perm = torch.empty(max(1, n), dtype=torch.int32, device=tc.device)
    offs = torch.zeros(partition_count + 2, dtype=torch.int64, device=tc.device)
    inp, keep = table.input_struct()
    tc.ctx.check(tc.ctx.L.gpuq_partition_run(op.h, tc.stream_ptr(), C.byref(inp), perm.data_ptr(), offs.data_ptr()))
    o = offs.cpu().tolist()
    return [_select_view(tc, table, perm[o[p]:o[p + 1]], o[p + 1] - o[p]) for p in range(partition_count)]


class ShuffleWritePartition(dict):
    """Stats record of shuffle_writer.rs:410-420 (partition_id, path, num_batches, num_rows, num_bytes)."""


class ShuffleWriterExec(ExecutionPlan):
    """Stage driver: ShuffleWriterExec(job_id, stage_id, plan, work_dir, shuffle_output_partitioning)
    -- ballista/core/src/execution_plans/shuffle_writer.rs:234-456.  Output files are Arrow IPC
    streams with LZ4_FRAME bodies in the reference's path layout (Appendix B.6 of SURVEY.md), so
    unchanged ShuffleReaderExec / Flight peers can consume them."""

    def __init__(self, job_id, stage_id, plan, work_dir, shuffle_output_partitioning=None, partitions=None):
        super().__init__()
        self.job_id, self.stage_id, self.plan, self.work_dir = job_id, stage_id, plan, work_dir
        self.shuffle_output_partitioning = shuffle_output_partitioning   # None or (hash_exprs, n)
        self.partitions = partitions

    def children(self):
        return [self.plan]

    def schema(self):
        return self.plan.schema()

    def execute_shuffle_write(self, input_partitions, context):
        from . import shuffle as S
        t0 = time.perf_counter()
        parts = list(input_partitions) if input_partitions is not None else (self.partitions or range(self.plan.output_partition_count()))
        out = []
        # the reference's ShuffleWriteMetrics (shuffle_writer.rs:139-160): write_time, repart_time (ns), input_rows, output_rows
        mx = self.metrics.extra
        for k in ("write_time", "repart_time", "input_rows"):
            mx.setdefault(k, 0)

        def sink(path, table):
            # IPC stream, LZ4_FRAME buffers compressed on the device (csrc/kernels_lz4.hip); batches of the session's batch size
            os.makedirs(os.path.dirname(path), exist_ok=True)
            return S.write_ipc_stream(context, path, table, batch_size=max(context.batch_size, S.SHUFFLE_BATCH_ROWS), codec=0)
        for p in parts:
            table = self.plan.execute(p, context)
            mx["input_rows"] += table.num_rows
            base = os.path.join(self.work_dir, self.job_id, str(self.stage_id))
            if self.shuffle_output_partitioning is None:
                path = os.path.join(base, str(uuid.uuid4()), "data.arrow")
                tw = time.perf_counter()
                nb, rows, nbytes = sink(path, table)
                mx["write_time"] += int((time.perf_counter() - tw) * 1e9)
                out.append(ShuffleWritePartition(partition_id=p, path=path, num_batches=nb, num_rows=rows, num_bytes=nbytes))
            else:
                exprs, n = self.shuffle_output_partitioning
                tr = time.perf_counter()
                views = partition_table(context, table, exprs, n)
                mx["repart_time"] += int((time.perf_counter() - tr) * 1e9)
                tw = time.perf_counter()
                for q, v in enumerate(views):
                    if v.num_rows == 0:
                        continue    # lazily created writers: empty partitions produce no file (shuffle_writer.rs:329-334)
                    path = os.path.join(base, str(q), "%s.arrow" % uuid.uuid4())
                    nb, rows, nbytes = sink(path, v)
                    out.append(ShuffleWritePartition(partition_id=q, path=path, num_batches=nb, num_rows=rows, num_bytes=nbytes))
                mx["write_time"] += int((time.perf_counter() - tw) * 1e9)
        self.metrics.elapsed_compute_ns += int((time.perf_counter() - t0) * 1e9)
        self.metrics.output_rows += sum(o["num_rows"] for o in out)
        return out


class ShuffleReaderExec(ExecutionPlan):
    """ShuffleReaderExec(partition: Vec<Vec<PartitionLocation>>, schema) -- ballista/core/src/execution_plans/shuffle_reader.rs:149-177.
    Output partition p is the concatenation of its locations' IPC streams, decoded on the device (gpuq_ipc_decode_batch).
    Locations are dicts with a "path" (the PartitionLocation field, serde/scheduler/mod.rs:47-55); only local files are read
    here -- fetching from a remote executor over Flight is the control plane's job (out of scope)."""

    def __init__(self, partition, schema):
        super().__init__()
        self.partition, self._schema = [list(p) for p in partition], schema

    def schema(self):
        return self._schema

    def output_partition_count(self):
        return len(self.partition)

    def execute(self, partition, context):
        from . import shuffle as S
        t0 = time.perf_counter()
        tables = []
        for loc in self.partition[partition]:
            path = loc["path"] if isinstance(loc, dict) else loc
            if not os.path.exists(path):
                # shuffle_reader.rs:654: a missing map output is a FetchFailed, which makes the scheduler re-run the map stage
                raise B.GpuqError(1, "FetchFailed: shuffle partition file %s does not exist" % path)
            t, _ = S.read_ipc_stream(context, path)
            tables.append(t)
        if not tables:
            out = S.empty_table(context, self._schema)
        else:
            out = tables[0] if len(tables) == 1 else concat_tables(context, tables)
        self.metrics.elapsed_compute_ns += int((time.perf_counter() - t0) * 1e9)
        self.metrics.output_rows += out.num_rows
        return out


class DefaultQueryStageExec:
    """QueryStageExecutor (execution_engine.rs:49-60, :97-133)."""

    def __init__(self, shuffle_writer):
        self.shuffle_writer = shuffle_writer

    def execute_query_stage(self, input_partitions, context):
        return self.shuffle_writer.execute_shuffle_write(input_partitions, context)

    def collect_plan_metrics(self):
        out = []

        def walk(p):
            out.append(p.metrics.as_dict())
            for c in p.children():
                walk(c)
        walk(self.shuffle_writer.plan)
        return out

    def schema(self):
        return self.shuffle_writer.schema()

    def __str__(self):
        lines = []

        def walk(p, d):
            lines.append("  " * d + "%s, metrics=%s" % (p, p.metrics.as_dict()))
            for c in p.children():
                walk(c, d + 1)
        walk(self.shuffle_writer, 0)
        return "\n".join(lines)


class DefaultExecutionEngine:
    """ExecutionEngine (execution_engine.rs:34-43, :62-95): the plan root must be a ShuffleWriterExec; it is
    re-created with the executor's work_dir."""

    def create_query_stage_exec(self, job_id, stage_id, plan, work_dir, sender=None):
        if not isinstance(plan, ShuffleWriterExec):
            raise B.GpuqError(1, "Plan passed to new_query_stage_exec is not a ShuffleWriterExec")
        w = ShuffleWriterExec(job_id, stage_id, plan.plan, work_dir, plan.shuffle_output_partitioning, plan.partitions)
        return DefaultQueryStageExec(w)
