"""Driver for the native (C++) plan executor, csrc/plan_exec.cpp.

`NativePlan(plan)` takes a tree of the plan classes in plan.py, serialises it to the JSON mirror of the reference's
PhysicalPlanNode messages (ballista/core/proto/datafusion.proto) and hands it to gpuq_plan_create; MemoryExec leaves become
input slots.  execute() runs one output partition entirely inside libgpuq: no per-operator Python, ctypes or torch
allocation sits between operators.  The plan classes in plan.py remain the test-side mirror of the same logic."""
import ctypes as C
import json

from . import binding as B
from . import plan as P
from .table import DeviceTable


class ArrowSchema(C.Structure):
    pass


class ArrowArray(C.Structure):
    pass


ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                        ("children", C.POINTER(C.POINTER(ArrowSchema))), ("dictionary", C.POINTER(ArrowSchema)), ("release", C.c_void_p), ("private_data", C.c_void_p)]
ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                       ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArray))), ("dictionary", C.POINTER(ArrowArray)),
                       ("release", C.c_void_p), ("private_data", C.c_void_p)]


def _check(L, rc):
    if rc != 0:
        raise B.GpuqError(rc, (L.gpuq_plan_last_error() or b"").decode())


def plan_to_json(node, tc, inputs):
    """plan.py tree -> JSON for gpuq_plan_create.  `inputs` collects the DeviceTables the MemoryExec leaves refer to."""
    t = type(node).__name__
    sub = lambda n: plan_to_json(n, tc, inputs)
    if isinstance(node, P.MemoryExec):
        slots, schema, dense, sides = [], None, False, None
        if all(p is None for p in node.partitions):      # a schema-only leaf (plan validation / typing without data)
            return {"MemoryExec": {"schema": [{"name": f["name"], "type": f["type"], "nullable": bool(f.get("nullable", True))} for f in node.schema()],
                                   "partitions": list(range(len(inputs), len(inputs) + len(node.partitions))), "dense": False}}
        for p in range(node.output_partition_count()):
            tab = node.execute(p, tc)
            slots.append(len(inputs))
            inputs.append(tab)
            schema = [{"name": c.name, "type": c.type, "nullable": bool(c.nullable)} for c in tab.columns]
            dense, sides = tab.dense, (list(tab.sides) if tab.is_view() else None)
        d = {"schema": schema, "partitions": slots, "dense": bool(dense)}
        if sides is not None:
            d["sides"] = sides
        return {"MemoryExec": d}
    if isinstance(node, P.CoalesceBatchesExec):
        return {"CoalesceBatchesExec": {"input": sub(node.input)}}
    if isinstance(node, P.FilterExec):
        return {"FilterExec": {"input": sub(node.input), "expr": node.predicate}}
    if isinstance(node, P.ProjectionExec):
        return {"ProjectionExec": {"input": sub(node.input), "expr": [e for e, _ in node.expr], "expr_name": [n for _, n in node.expr]}}
    if isinstance(node, P.AggregateExec):
        d = {"input": sub(node.input), "mode": node.mode, "strategy": node.strategy,
             "group_expr": [{"expr": e, "name": n} for e, n in node.group_expr],
             "aggr_expr": [{k: v for k, v in a.items() if v is not None} for a in node.aggr_expr]}
        if node.expected_groups:
            d["expected_groups"] = int(node.expected_groups)
        if getattr(node, "output_capacity", None):
            d["output_capacity"] = int(node.output_capacity)
        return {"AggregateExec": d}
    if isinstance(node, P.HashJoinExec):
        d = {"left": sub(node.left), "right": sub(node.right), "on": [{"left": l, "right": r} for l, r in node.on], "join_type": node.join_type,
             "partition_mode": node.partition_mode, "null_equals_null": bool(node.null_equals_null)}
        if node.filter is not None:
            d["filter"] = node.filter
        return {"HashJoinExec": d}
    if isinstance(node, P.CrossJoinExec):
        return {"CrossJoinExec": {"left": sub(node.left), "right": sub(node.right)}}
    if isinstance(node, P.SortPreservingMergeExec):
        return {"SortPreservingMergeExec": {"input": sub(node.input), "expr": node.expr, "fetch": -1 if node.fetch is None else int(node.fetch)}}
    if isinstance(node, P.SortExec):
        return {"SortExec": {"input": sub(node.input), "expr": node.expr, "fetch": -1 if node.fetch is None else int(node.fetch)}}
    if isinstance(node, P.UnionExec):
        return {"UnionExec": {"inputs": [sub(i) for i in node.inputs]}}
    if isinstance(node, P.CoalescePartitionsExec):
        return {"CoalescePartitionsExec": {"input": sub(node.input)}}
    if isinstance(node, P.CoalesceTasksExec):
        d = {"input": sub(node.input), "partitions": [int(p) for p in node.partitions]}
        if node.order_by:
            d["order_by"] = node.order_by
        return {"CoalesceTasksExec": d}
    if isinstance(node, P.GlobalLimitExec):
        return {"GlobalLimitExec": {"input": sub(node.input), "skip": int(node.skip), "fetch": -1 if node.fetch is None else int(node.fetch)}}
    if isinstance(node, P.LocalLimitExec):
        return {"LocalLimitExec": {"input": sub(node.input), "fetch": int(node.fetch)}}
    if isinstance(node, P.RepartitionExchangeExec):
        return {"RepartitionExec": {"input": sub(node.input), "hash_expr": list(node.hash_expr), "partition_count": int(node.partition_count)}}
    if isinstance(node, P.RangeRepartitionExec):
        return {"RangeRepartitionExec": {"input": sub(node.input), "expr": list(node.expr), "partition_count": int(node.partition_count), "samples": int(node.samples)}}
    if isinstance(node, P.BroadcastExec):
        return {"BroadcastExec": {"input": sub(node.input)}}
    if isinstance(node, P.ShuffleWriterExec):
        d = {"input": sub(node.plan), "job_id": node.job_id, "stage_id": int(node.stage_id), "work_dir": node.work_dir}
        if node.shuffle_output_partitioning is not None:
            exprs, n = node.shuffle_output_partitioning
            d["output_partitioning"] = {"hash_expr": list(exprs), "partition_count": int(n)}
        if node.partitions is not None:
            d["partitions"] = [int(x) for x in node.partitions]
        return {"ShuffleWriterExec": d}
    if isinstance(node, P.ShuffleReaderExec):
        return {"ShuffleReaderExec": {"schema": [{"name": f["name"], "type": f["type"], "nullable": bool(f.get("nullable", True))} for f in node.schema()],
                                      "partition": [[{"path": (l["path"] if isinstance(l, dict) else l)} for l in p] for p in node.partition]}}
    raise B.GpuqError(3, "plan node %s is not executed natively" % t)


class NativeResult:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        L = ctx.L
        self.num_rows = int(L.gpuq_result_num_rows(self.h))
        self.num_columns = int(L.gpuq_result_num_columns(self.h))

    def to_arrow(self):
        import pyarrow as pa
        L = self.ctx.L
        nc = self.num_columns
        cols, fields = (B.gpuq_column * max(1, nc))(), (B.gpuq_field_info * max(1, nc))()
        for i in range(nc):
            _check(L, L.gpuq_result_column(self.h, i, C.byref(cols[i]), C.byref(fields[i])))
        oa, osch = ArrowArray(), ArrowSchema()
        self.ctx.check(L.gpuq_export_arrow(self.ctx.h, None, cols, fields, nc, self.num_rows, C.addressof(oa), C.addressof(osch)))
        return pa.Table.from_batches([pa.RecordBatch._import_from_c(C.addressof(oa), C.addressof(osch))])

    def to_device_table(self, device):
        """The result as a DeviceTable whose tensors alias the library's buffers (no copy); they keep this NativeResult alive."""
        import torch
        from .table import DeviceColumn, DeviceTable, type_json
        cols, fields = self.columns_c()
        n = self.num_rows
        owner = self

        def alias(ptr, nb):
            if not ptr or nb <= 0:
                return torch.zeros(max(int(nb), 16), dtype=torch.uint8, device=device)

            class _A:
                pass
            a = _A()
            a.__cuda_array_interface__ = {"shape": (int(nb),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
            a.owner = owner
            return torch.as_tensor(a, device=device)
        out = []
        words = ((n + 63) // 64) * 8
        for i in range(self.num_columns):
            c, f = cols[i], fields[i]
            ty = type_json(f.type, f.precision, f.scale)
            validity = alias(c.validity, max(words, 8)) if c.validity else None
            packed = f.type == B.T_UTF8 and (c.repr == B.REPR_PACKED15 or f.repr == B.REPR_PACKED15 or (not c.offsets and n > 0))
            if f.type == B.T_UTF8 and not packed:
                offs = alias(c.offsets, (n + 1) * 4).view(torch.int32) if c.offsets else torch.zeros(4, dtype=torch.int32, device=device)
                last = int(offs[n].item()) if n and c.offsets else 0
                out.append(DeviceColumn(f.name.decode(), ty, alias(c.data, max(last, 1)), n, offsets=offs, validity=validity, nullable=bool(f.nullable)))
            elif f.type == B.T_UTF8:
                out.append(DeviceColumn(f.name.decode(), ty, alias(c.data, max(n, 1) * 16), n, validity=validity, nullable=bool(f.nullable), repr=B.REPR_PACKED15))
            elif f.type == B.T_BOOL:
                out.append(DeviceColumn(f.name.decode(), ty, alias(c.data, max(words, 8)), n, validity=validity, nullable=bool(f.nullable)))
            else:
                out.append(DeviceColumn(f.name.decode(), ty, alias(c.data, max(n, 1) * max(int(f.width), 1)), n, validity=validity, nullable=bool(f.nullable)))
        t = DeviceTable(out, n)
        t._keep = self
        rec = self.record(device)
        if rec is not None:
            t._record = rec
        return t

    def columns_c(self):
        """(gpuq_column array, gpuq_field_info array) describing the result's device buffers."""
        L = self.ctx.L
        nc = self.num_columns
        cols, fields = (B.gpuq_column * max(1, nc))(), (B.gpuq_field_info * max(1, nc))()
        for i in range(nc):
            _check(L, L.gpuq_result_column(self.h, i, C.byref(cols[i]), C.byref(fields[i])))
        return cols, fields

    def record(self, device):
        """The result as one fixed-layout device record: (uint8 tensor aliasing the library's allocation, row capacity), or
        None.  The tensor is only valid while this NativeResult is alive."""
        import torch
        base, nbytes, cap = C.c_void_p(), C.c_size_t(0), C.c_int64(0)
        _check(self.ctx.L, self.ctx.L.gpuq_result_record(self.h, C.byref(base), C.byref(nbytes), C.byref(cap)))
        if not base.value:
            return None

        class _Alias:
            pass
        a = _Alias()
        a.__cuda_array_interface__ = {"shape": (int(nbytes.value),), "typestr": "|u1", "data": (int(base.value), False), "version": 2}
        a.owner = self
        return torch.as_tensor(a, device=device), int(cap.value)

    def close(self):
        if self.h:
            self.ctx.L.gpuq_result_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeTask:
    def __init__(self, ctx, handle, plan):
        self.ctx, self.h, self._plan = ctx, handle, plan

    def done(self):
        d = C.c_int(0)
        _check(self.ctx.L, self.ctx.L.gpuq_task_poll(self.h, C.byref(d)))
        return bool(d.value)

    def cancel(self):
        _check(self.ctx.L, self.ctx.L.gpuq_task_cancel(self.h))

    def wait(self):
        """NativeResult, or raises GpuqError (status 6 = cancelled)."""
        out = C.c_void_p()
        rc = self.ctx.L.gpuq_task_wait(self.h, C.byref(out))
        _check(self.ctx.L, rc)
        return NativeResult(self.ctx, out)

    def close(self):
        if self.h:
            self.ctx.L.gpuq_task_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativePlan:
    def __init__(self, plan, tc):
        self.tc = tc
        self.inputs = []
        self.json = json.dumps(plan_to_json(plan, tc, self.inputs))
        L = tc.ctx.L
        h = C.c_void_p()
        _check(L, L.gpuq_plan_create(tc.ctx.h, self.json.encode(), C.byref(h)))
        self.h = h
        self._bind()

    def _bind(self):
        n = len(self.inputs)
        self._arr = (B.gpuq_input * max(1, n))()
        self._keep = []
        for i, t in enumerate(self.inputs):
            inp, keep = t.input_struct()
            self._arr[i] = inp
            self._keep.append((inp, keep, t))

    def set_input(self, slot, table):
        """Replace the table behind input slot `slot` (same schema / layout)."""
        self.inputs[slot] = table
        inp, keep = table.input_struct()
        self._arr[slot] = inp
        self._keep[slot] = (inp, keep, table)

    def set_input_result(self, slot, result):
        """Feed another plan's NativeResult in as input slot `slot` (no torch tensors involved); `result` must outlive the call."""
        cols, fields = result.columns_c()
        inp = B.gpuq_input()
        inp.cols = C.cast(cols, C.POINTER(B.gpuq_column))
        inp.n_cols = result.num_columns
        inp.n_via = 0
        inp.n_rows = result.num_rows
        self._arr[slot] = inp
        self._keep[slot] = (inp, cols, result)

    def schema(self):
        """[(name, type json, nullable)] of the plan's output, from the plan alone (gpuq_plan_schema: nothing runs)."""
        from .table import type_json
        L = self.tc.ctx.L
        n = C.c_int(0)
        _check(L, L.gpuq_plan_schema(self.h, None, 0, C.byref(n)))
        f = (B.gpuq_field_info * max(1, n.value))()
        _check(L, L.gpuq_plan_schema(self.h, f, n.value, C.byref(n)))
        return [(f[i].name.decode(), type_json(f[i].type, f[i].precision, f[i].scale), bool(f[i].nullable)) for i in range(n.value)]

    def set_comm(self, comm):
        """Attach the ranks of the node (parallel.Comm) for RepartitionExec / BroadcastExec nodes; the comm must outlive the plan."""
        self._comm = comm
        _check(self.tc.ctx.L, self.tc.ctx.L.gpuq_plan_set_comm(self.h, comm.h if comm is not None else None))

    def execute(self, partition=0):
        L = self.tc.ctx.L
        out = C.c_void_p()
        _check(L, L.gpuq_plan_execute(self.h, self.tc.stream_ptr(), int(partition), self._arr, len(self.inputs), C.byref(out)))
        return NativeResult(self.tc.ctx, out)

    def execute_async(self, partition=0):
        """Start the plan on a library worker thread; returns a NativeTask (wait / poll / cancel)."""
        L = self.tc.ctx.L
        h = C.c_void_p()
        _check(L, L.gpuq_plan_execute_async(self.h, self.tc.stream_ptr(), int(partition), self._arr, len(self.inputs), C.byref(h)))
        return NativeTask(self.tc.ctx, h, self)

    def profile(self, enable=True):
        """(kernel ms, launches, operator descriptor) of the plan's most expensive operator since profiling was enabled."""
        ms, n = C.c_float(0), C.c_int(0)
        buf = C.create_string_buffer(1 << 15)
        _check(self.tc.ctx.L, self.tc.ctx.L.gpuq_plan_profile(self.h, 1 if enable else 0, C.byref(ms), C.byref(n), buf, len(buf)))
        return ms.value, n.value, buf.value.decode()

    def profile_all(self):
        """[{op, kernel_ms, launches, desc}] for every operator the plan has compiled (accumulators are reset)."""
        buf = C.create_string_buffer(1 << 20)
        _check(self.tc.ctx.L, self.tc.ctx.L.gpuq_plan_profile_all(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def exec_stats(self):
        """How the last execution went: {deferred, settles, host_syncs, retries} (include/gpuq.h gpuq_plan_exec_stats)."""
        v = [C.c_int(0) for _ in range(4)]
        _check(self.tc.ctx.L, self.tc.ctx.L.gpuq_plan_exec_stats(self.h, *[C.byref(a) for a in v]))
        return {"deferred": bool(v[0].value), "settles": v[1].value, "host_syncs": v[2].value, "retries": v[3].value}

    def metrics(self):
        buf = C.create_string_buffer(1 << 16)
        _check(self.tc.ctx.L, self.tc.ctx.L.gpuq_plan_metrics(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def close(self):
        if self.h:
            self.tc.ctx.L.gpuq_plan_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
