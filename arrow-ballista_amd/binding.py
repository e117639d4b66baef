"""ctypes binding of include/gpuq.h.  One-to-one with the C ABI; no compute happens here."""
import ctypes as C
import json
import os

T_NULL, T_BOOL, T_INT32, T_INT64, T_DATE32, T_FLOAT64, T_DECIMAL128, T_UTF8, T_UINT32, T_UINT64 = range(10)
T_INT8, T_INT16, T_UINT8, T_UINT16, T_FLOAT32, T_TIMESTAMP, T_DATE64 = range(10, 17)
REPR_ARROW, REPR_PACKED15 = 0, 1
_STATUS = {6: "CANCELLED", 1: "INVALID", 2: "HIP", 3: "UNSUPPORTED", 4: "CAPACITY", 5: "INTERNAL"}


class GpuqError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("gpuq %s: %s" % (_STATUS.get(status, status), message))
        self.status = status
        self.message = message


class gpuq_column(C.Structure):
    _fields_ = [("type", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32), ("repr", C.c_int32),
                ("data", C.c_void_p), ("offsets", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_int64)]


class gpuq_field_info(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("type", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32),
                ("nullable", C.c_int32), ("repr", C.c_int32), ("width", C.c_int32)]


class gpuq_csv_options(C.Structure):
    _fields_ = [("delimiter", C.c_char), ("quote", C.c_char), ("has_header", C.c_int32)]


class gpuq_input(C.Structure):
    _fields_ = [("cols", C.POINTER(gpuq_column)), ("n_cols", C.c_int32), ("n_via", C.c_int32), ("n_rows", C.c_int64),
                ("via", C.c_void_p * 3), ("n_rows_dev", C.c_void_p)]


_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path():
    # GPUQ_LIB: another build of the same library (A/B measurements of a kernel change on one box)
    return os.environ.get("GPUQ_LIB") or os.path.join(_HERE, "libgpuq.so")


def lib():
    """Load libgpuq.so (built in-tree by build.py).  Raises when it is missing: there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise GpuqError(5, "libgpuq.so not built (%s); run `python arrow-ballista_amd/build.py`" % p)
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7.  Load torch first so
    # libgpuq.so's NEEDED libamdhip64.so.7 resolves to that same instance; otherwise two runtimes would
    # be live and torch tensors' device pointers would belong to a different runtime than our streams.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(p)
    vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
    sig = {
        "gpuq_abi_version": (i32, []),
        "gpuq_ctx_create": (vp, [i32, C.c_char_p]),
        "gpuq_ctx_free": (None, [vp]),
        "gpuq_last_error": (C.c_char_p, [vp]),
        "gpuq_ctx_device_info": (i32, [vp, C.c_char_p, C.c_size_t]),
        "gpuq_buffer_alloc": (i32, [vp, C.c_size_t, C.POINTER(vp)]),
        "gpuq_buffer_free": (i32, [vp, vp]),
        "gpuq_copy_h2d": (i32, [vp, vp, vp, vp, C.c_size_t]),
        "gpuq_copy_d2h": (i32, [vp, vp, vp, vp, C.c_size_t]),
        "gpuq_table_import_arrow": (i32, [vp, vp, vp, vp, C.POINTER(vp)]),
        "gpuq_cross_pairs": (i32, [vp, vp, i64, i64, vp, vp]),
        "gpuq_sort_run_keys": (i32, [vp, vp, C.POINTER(gpuq_input), vp, vp, vp, C.POINTER(i32)]),
        "gpuq_memory_limit": (i32, [i64]),
        "gpuq_memory_stats": (i32, [C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), i32]),
        "gpuq_utf8_compare": (i32, [vp, vp, C.POINTER(gpuq_column), vp, C.POINTER(gpuq_column), vp, C.c_char_p, i64, i64, i32, vp, vp]),
        "gpuq_ingest_create": (i32, [vp, vp, i64, i64, i32, C.POINTER(vp)]),
        "gpuq_ingest_push": (i32, [vp, vp]),
        "gpuq_ingest_rows_landed": (i32, [vp, C.POINTER(i64)]),
        "gpuq_ingest_wait_rows": (i32, [vp, i64, C.POINTER(i64)]),
        "gpuq_ingest_columns": (i32, [vp, C.POINTER(gpuq_column), C.POINTER(gpuq_field_info), i32, C.POINTER(i32)]),
        "gpuq_ingest_stats": (i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
        "gpuq_ingest_free": (None, [vp]),
        "gpuq_ingest_last_error": (C.c_char_p, []),
        "gpuq_csv_decode": (i32, [vp, vp, vp, i64, C.POINTER(gpuq_field_info), i32, C.POINTER(C.c_int32), i32, C.POINTER(gpuq_csv_options), C.POINTER(vp)]),
        "gpuq_parquet_decode": (i32, [vp, vp, vp, i64, C.POINTER(C.c_char_p), i32, C.POINTER(vp)]),
        "gpuq_parquet_decode_groups": (i32, [vp, vp, vp, i64, C.POINTER(C.c_char_p), i32, C.POINTER(C.c_int32), i32, C.POINTER(vp)]),
        "gpuq_parquet_row_groups": (i32, [vp, i64, C.POINTER(i64), i32, C.POINTER(i32)]),
        "gpuq_parquet_schema": (i32, [vp, i64, C.POINTER(gpuq_field_info), i32, C.POINTER(i32), C.POINTER(i64)]),
        "gpuq_scan_last_error": (C.c_char_p, []),
        "gpuq_table_num_rows": (i64, [vp]),
        "gpuq_table_num_columns": (i32, [vp]),
        "gpuq_table_column": (i32, [vp, i32, C.POINTER(gpuq_column), C.POINTER(gpuq_field_info)]),
        "gpuq_table_free": (None, [vp]),
        "gpuq_export_arrow": (i32, [vp, vp, C.POINTER(gpuq_column), C.POINTER(gpuq_field_info), i32, i64, vp, vp]),
        "gpuq_ctx_set_option": (i32, [vp, C.c_char_p, C.c_char_p]),
        "gpuq_ctx_set_jit": (i32, [vp, C.c_char_p, i64]),
        "gpuq_ctx_jit_wait": (i32, [vp]),
        "gpuq_jit_quiesce": (None, []),
        "gpuq_jit_cache_stats": (i32, [C.POINTER(i32), C.POINTER(i32)]),
        "gpuq_ctx_jit_stats": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_size_t]),
        "gpuq_op_jit_source": (i32, [vp, i32, C.c_char_p, C.c_size_t]),
        "gpuq_op_create": (i32, [vp, C.c_char_p, C.POINTER(vp)]),
        "gpuq_compile_check": (i32, [C.c_char_p, C.c_char_p, C.c_size_t]),
        "gpuq_compile_jit_source": (i32, [C.c_char_p, i32, C.c_char_p, C.c_size_t]),
        "gpuq_op_free": (None, [vp]),
        "gpuq_op_num_outputs": (i32, [vp]),
        "gpuq_op_output_field": (i32, [vp, i32, C.POINTER(gpuq_field_info)]),
        "gpuq_filter_run": (i32, [vp, vp, C.POINTER(gpuq_input), i32, vp, vp]),
        "gpuq_project_run": (i32, [vp, vp, C.POINTER(gpuq_input), C.POINTER(gpuq_column), i32]),
        "gpuq_aggregate_run": (i32, [vp, vp, C.POINTER(gpuq_input), C.POINTER(gpuq_column), i32, i64, C.POINTER(i64)]),
        "gpuq_aggregate_run_deferred": (i32, [vp, vp, C.POINTER(gpuq_input), C.POINTER(gpuq_column), i32, i64, C.POINTER(i64), C.POINTER(vp)]),
        "gpuq_op_set_deferred": (i32, [vp, i32]),
        "gpuq_op_can_defer": (i32, [vp]),
        "gpuq_ops_settle": (i32, [vp, vp, C.POINTER(vp), i32, C.POINTER(vp), i32, C.POINTER(u64)]),
        "gpuq_join_build_run": (i32, [vp, vp, C.POINTER(gpuq_input), i32, i64, C.POINTER(vp)]),
        "gpuq_join_table_free": (None, [vp]),
        "gpuq_join_build_run_semi": (i32, [vp, vp, C.POINTER(gpuq_input), i32, i64, vp, vp, vp, C.POINTER(vp)]),
        "gpuq_join_table_has_duplicates": (i32, [vp]),
        "gpuq_join_probe_run": (i32, [vp, vp, vp, C.POINTER(gpuq_input), i32, vp, vp, u64, vp]),
        "gpuq_join_build_side_rows": (i32, [vp, vp, i32, vp, vp]),
        "gpuq_mark_rows": (i32, [vp, vp, vp, i64, vp]),
        "gpuq_sort_run": (i32, [vp, vp, C.POINTER(gpuq_input), vp]),
        "gpuq_merge_run": (i32, [vp, vp, C.POINTER(gpuq_input), C.POINTER(i64), i32, vp]),
        "gpuq_partition_run": (i32, [vp, vp, C.POINTER(gpuq_input), vp, vp]),
        "gpuq_op_check": (i32, [vp, vp]),
        "gpuq_unpack_utf8": (i32, [vp, vp, vp, i64, vp, vp, i64, C.POINTER(i64)]),
        "gpuq_offsets_rebase": (i32, [vp, vp, vp, i64, i32, vp]),
        "gpuq_take_utf8": (i32, [vp, vp, C.POINTER(gpuq_column), vp, i64, vp, vp, vp, i64, C.POINTER(i64)]),
        "gpuq_utf8_max_len": (i32, [vp, vp, C.POINTER(gpuq_column), vp, i64, C.POINTER(C.c_int32)]),
        "gpuq_utf8_sort_piece": (i32, [vp, vp, C.POINTER(gpuq_column), vp, i64, i32, vp, vp]),
        "gpuq_utf8_dict_create": (i32, [vp, vp, i64, C.POINTER(vp)]),
        "gpuq_utf8_intern": (i32, [vp, vp, C.POINTER(gpuq_column), vp, i64, i32, vp, vp]),
        "gpuq_utf8_dict_free": (None, [vp]),
        "gpuq_utf8_code_rows": (i32, [vp, vp, C.POINTER(gpuq_column), i64, vp]),
        "gpuq_like_utf8": (i32, [vp, vp, C.POINTER(gpuq_column), vp, i64, C.c_char_p, i32, i32, vp, vp]),
        "gpuq_concat_bitmap": (i32, [vp, vp, vp, i64, vp, i64]),
        "gpuq_copy_bits": (i32, [vp, vp, vp, i64, vp, i64, i64]),
        "gpuq_plan_create": (i32, [vp, C.c_char_p, C.POINTER(vp)]),
        "gpuq_plan_free": (None, [vp]),
        "gpuq_plan_num_partitions": (i32, [vp]),
        "gpuq_plan_schema": (i32, [vp, C.POINTER(gpuq_field_info), i32, C.POINTER(i32)]),
        "gpuq_plan_execute": (i32, [vp, vp, i32, C.POINTER(gpuq_input), i32, C.POINTER(vp)]),
        "gpuq_plan_execute_async": (i32, [vp, vp, i32, C.POINTER(gpuq_input), i32, C.POINTER(vp)]),
        "gpuq_task_poll": (i32, [vp, C.POINTER(i32)]),
        "gpuq_task_cancel": (i32, [vp]),
        "gpuq_task_wait": (i32, [vp, C.POINTER(vp)]),
        "gpuq_task_free": (None, [vp]),
        "gpuq_plan_metrics": (i32, [vp, C.c_char_p, C.c_size_t]),
        "gpuq_plan_set_comm": (i32, [vp, vp]),
        "gpuq_plan_exec_stats": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "gpuq_comm_unique_id": (i32, [vp]),
        "gpuq_table_piece_rows": (i32, [vp, C.POINTER(i64), i32, C.POINTER(i32)]),
        "gpuq_comm_set_status": (i32, [vp, i32]),
        "gpuq_comm_announce": (i32, [vp, vp]),
        "gpuq_comm_create": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "gpuq_comm_create_host": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "gpuq_comm_free": (None, [vp]),
        "gpuq_comm_rank": (i32, [vp]),
        "gpuq_comm_world": (i32, [vp]),
        "gpuq_exchange_partitions": (i32, [vp, vp, C.POINTER(gpuq_column), C.POINTER(gpuq_field_info), i32, C.POINTER(i64), C.POINTER(vp)]),
        "gpuq_allgather_table": (i32, [vp, vp, C.POINTER(gpuq_column), C.POINTER(gpuq_field_info), i32, i64, C.POINTER(vp)]),
        "gpuq_exchange_last_error": (C.c_char_p, []),
        "gpuq_plan_last_error": (C.c_char_p, []),
        "gpuq_plan_profile": (i32, [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
        "gpuq_plan_profile_all": (i32, [vp, C.c_char_p, C.c_size_t]),
        "gpuq_result_num_rows": (i64, [vp]),
        "gpuq_result_num_columns": (i32, [vp]),
        "gpuq_result_column": (i32, [vp, i32, C.POINTER(gpuq_column), C.POINTER(gpuq_field_info)]),
        "gpuq_result_free": (None, [vp]),
        "gpuq_result_record": (i32, [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(i64)]),
        "gpuq_ipc_peek": (i32, [vp, i64, vp]),
        "gpuq_ipc_schema_message": (i32, [C.POINTER(gpuq_field_info), i32, vp, i64, C.POINTER(i64)]),
        "gpuq_ipc_schema_message_kv": (i32, [C.POINTER(gpuq_field_info), i32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), i32, vp, i64, C.POINTER(i64)]),
        "gpuq_ipc_schema_metadata": (i32, [vp, i64, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(i32)]),
        "gpuq_ipc_encode_batch": (i32, [vp, vp, C.POINTER(gpuq_column), i32, i64, i32, vp, i64, C.POINTER(i64)]),
        "gpuq_ipc_decode_batch": (i32, [vp, vp, vp, i64, C.POINTER(gpuq_field_info), i32, C.POINTER(vp)]),
        "gpuq_ipc_decode_stream": (i32, [vp, vp, vp, i64, C.POINTER(gpuq_field_info), i32, C.POINTER(vp)]),
        "gpuq_ipc_batch_num_rows": (i64, [vp]),
        "gpuq_ipc_batch_num_columns": (i32, [vp]),
        "gpuq_ipc_batch_column": (i32, [vp, i32, C.POINTER(gpuq_column)]),
        "gpuq_ipc_batch_free": (None, [vp]),
        "gpuq_ipc_last_error": (C.c_char_p, []),
        "gpuq_timer_create": (i32, [vp, C.POINTER(vp)]),
        "gpuq_timer_start": (i32, [vp, vp]),
        "gpuq_timer_stop": (i32, [vp, vp]),
        "gpuq_timer_elapsed_ms": (i32, [vp, C.POINTER(C.c_float)]),
        "gpuq_timer_free": (None, [vp]),
        "gpuq_op_profile": (i32, [vp, i32, C.POINTER(C.c_float), C.POINTER(i32)]),
        "gpuq_op_profile_total": (i32, [vp, C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError when the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    L._gpuq_symbols = sorted(sig)
    import atexit
    atexit.register(L.gpuq_jit_quiesce)      # background compiles must not outlive the interpreter (include/gpuq.h)
    _LIB = L
    return L


def memory_limit(nbytes):
    """Process-wide budget for the device memory the library holds (0 = none): include/gpuq.h gpuq_memory_limit."""
    rc = lib().gpuq_memory_limit(int(nbytes))
    if rc != 0:
        raise GpuqError(rc, "bad memory limit")


def memory_stats(reset_peak=False):
    a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
    lib().gpuq_memory_stats(C.byref(a), C.byref(b), C.byref(c), C.byref(d), 1 if reset_peak else 0)
    return {"in_use": a.value, "peak": b.value, "cached": c.value, "limit": d.value}


def compile_check(descriptor):
    """Compile an operator descriptor on the host only (no device) and return its description."""
    L = lib()
    buf = C.create_string_buffer(1 << 18)
    rc = L.gpuq_compile_check(json.dumps(descriptor).encode(), buf, len(buf))
    if rc != 0:
        raise GpuqError(rc, L.gpuq_last_error(None).decode())
    return json.loads(buf.value.decode())


def compile_jit_source(descriptor, kernel_id):
    """Host only: the source the JIT path hands to hiprtc for this descriptor and sink kernel."""
    L = lib()
    buf = C.create_string_buffer(1 << 20)
    rc = L.gpuq_compile_jit_source(json.dumps(descriptor).encode(), int(kernel_id), buf, len(buf))
    if rc != 0:
        raise GpuqError(rc, L.gpuq_last_error(None).decode())
    return buf.value.decode()


class Context:
    """gpuq_ctx: one device.  Raises GpuqError when no HIP device is usable (no CPU fallback)."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = self.L.gpuq_ctx_create(int(device), None)
        if not self.h:
            raise GpuqError(2, self.L.gpuq_last_error(None).decode())
        self.device = int(device)

    def check(self, rc):
        if rc != 0:
            raise GpuqError(rc, self.L.gpuq_last_error(self.h).decode())

    def set_jit(self, mode, min_rows=-1):
        self.check(self.L.gpuq_ctx_set_jit(self.h, mode.encode(), int(min_rows)))

    def set_option(self, key, value):
        self.check(self.L.gpuq_ctx_set_option(self.h, key.encode(), str(value).encode()))

    def jit_wait(self):
        """Wait for the background specialisation of hot small-input programs requested so far."""
        self.check(self.L.gpuq_ctx_jit_wait(self.h))

    def jit_stats(self):
        a, n = C.c_int(0), C.c_int(0)
        buf = C.create_string_buffer(8192)
        self.L.gpuq_ctx_jit_stats(self.h, C.byref(a), C.byref(n), buf, len(buf))
        return {"available": bool(a.value), "launches": n.value, "last_error": buf.value.decode(errors="replace")}

    def info(self):
        buf = C.create_string_buffer(1024)
        self.check(self.L.gpuq_ctx_device_info(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def close(self):
        if self.h:
            self.L.gpuq_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Op:
    """gpuq_op: one compiled operator."""

    def __init__(self, ctx, descriptor):
        self.ctx = ctx
        self.L = ctx.L
        self.descriptor = descriptor
        h = C.c_void_p()
        ctx.check(self.L.gpuq_op_create(ctx.h, json.dumps(descriptor).encode(), C.byref(h)))
        self.h = h
        self.fields = []
        for i in range(self.L.gpuq_op_num_outputs(h)):
            f = gpuq_field_info()
            ctx.check(self.L.gpuq_op_output_field(h, i, C.byref(f)))
            self.fields.append(dict(name=f.name.decode(), type=f.type, precision=f.precision, scale=f.scale,
                                    nullable=bool(f.nullable), repr=f.repr, width=f.width))

    def check(self, stream=None):
        self.ctx.check(self.L.gpuq_op_check(self.h, stream))

    def jit_source(self, kernel_id):
        buf = C.create_string_buffer(1 << 20)
        self.ctx.check(self.L.gpuq_op_jit_source(self.h, kernel_id, buf, len(buf)))
        return buf.value.decode()

    def profile(self, enable=True):
        ms, n = C.c_float(0), C.c_int(0)
        self.ctx.check(self.L.gpuq_op_profile(self.h, 1 if enable else 0, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "h", None):
            self.L.gpuq_op_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class JoinTable:
    def __init__(self, ctx, handle):
        self.ctx, self.L, self.h = ctx, ctx.L, handle

    def close(self):
        if getattr(self, "h", None):
            self.L.gpuq_join_table_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
