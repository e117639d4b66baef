"""Shuffle files on the device codec: Arrow IPC streams whose buffers are LZ4 frames, written from / read into HBM columns.

What it mirrors in the reference: the sink of ShuffleWriterExec (`StreamWriter::try_new_with_options(.., LZ4_FRAME)` +
`writer.write(&batch)` + `finish()`, ballista/core/src/execution_plans/shuffle_writer.rs:365-378 and
ballista/core/src/utils.rs:179-219 `write_stream_to_disk`) and the IPC stream reader behind ShuffleReaderExec
(ballista/core/src/execution_plans/shuffle_reader.rs; local files are opened with `StreamReader::try_new`).  All buffer
(de)compression runs in libgpuq.so (`gpuq_ipc_encode_batch` / `gpuq_ipc_decode_batch`, csrc/kernels_lz4.hip); this module
only frames the stream: the Schema message comes from the host's Arrow library (pyarrow here, arrow-rs in the reference),
then one encapsulated message per batch, then the end-of-stream marker."""
import ctypes as C

from . import binding as B
from .table import DeviceColumn, DeviceTable, json_arrow_type, type_id, type_json

EOS = b"\xff\xff\xff\xff\x00\x00\x00\x00"
# Rows per RecordBatch the device sink writes: the reference cuts its stream at the session batch size (8192) because that is
# what flows through its operators; a GPU partition is whole columns, and every IPC reader accepts any batch length.
SHUFFLE_BATCH_ROWS = 1 << 20


class gpuq_ipc_info(C.Structure):
    _fields_ = [("header_type", C.c_int32), ("codec", C.c_int32), ("metadata_bytes", C.c_int64), ("body_bytes", C.c_int64),
                ("n_rows", C.c_int64), ("n_nodes", C.c_int32), ("n_buffers", C.c_int32)]


def _check(L, rc):
    if rc != 0:
        raise B.GpuqError(rc, L.gpuq_ipc_last_error().decode())


def _arrow_type(t):
    import pyarrow as pa
    return json_arrow_type(t)


def _arrow_schema(table):
    import pyarrow as pa
    return pa.schema([pa.field(c.name, _arrow_type(c.type), nullable=bool(c.nullable)) for c in table.columns])


def empty_table(tc, schema):
    """Zero-row device table for a list of field dicts ({"name","type","nullable"})."""
    import pyarrow as pa
    fields = [pa.field(f["name"], _arrow_type(f["type"]), nullable=bool(f.get("nullable", True))) for f in schema]
    return DeviceTable.from_arrow(pa.Table.from_batches([], schema=pa.schema(fields)), tc.device)


def _arrow_layout(tc, table):
    """Plain (non-view) table with every Utf8 column in Arrow layout."""
    from . import plan as P
    torch = P._torch()
    t = P.materialize(tc, table, pack_strings=False)
    cols = []
    for c in t.columns:
        if c.repr == B.REPR_PACKED15:
            n = c.length
            off = torch.zeros(n + 4, dtype=torch.int32, device=tc.device)
            dat = torch.zeros(max(16, n * 15), dtype=torch.uint8, device=tc.device)
            dl = C.c_int64(0)
            tc.ctx.check(tc.ctx.L.gpuq_unpack_utf8(tc.ctx.h, tc.stream_ptr(), c.data.data_ptr() if n else None, n, off.data_ptr(), dat.data_ptr(),
                                                   dat.numel(), C.byref(dl)))
            c = DeviceColumn(c.name, c.type, dat, n, offsets=off, validity=c.validity, nullable=c.nullable)
        cols.append(c)
    return DeviceTable(cols, t.num_rows)


def _slice_plain(tc, table, lo, n):
    """Rows [lo, lo+n) of a plain Arrow-layout table as column views: pointer arithmetic for the data (fixed width: byte
    offset; Utf8: the offsets array is entered at `lo`, the encoder re-bases it), a bit copy for bitmaps unless lo % 8 == 0."""
    from . import plan as P
    torch = P._torch()
    L = tc.ctx.L
    out = []

    def bits(src):
        if src is None:
            return None
        if lo % 64 == 0:
            return src[lo // 8:]
        dst = torch.zeros(((n + 63) // 64) * 8 + 8, dtype=torch.uint8, device=tc.device)
        tc.ctx.check(L.gpuq_copy_bits(tc.ctx.h, tc.stream_ptr(), dst.data_ptr(), 0, src.data_ptr(), lo, n))
        return dst
    for c in table.columns:
        tid = type_id(c.type)[0]
        if tid == B.T_UTF8:
            out.append(DeviceColumn(c.name, c.type, c.data, n, offsets=c.offsets[lo:], validity=bits(c.validity), nullable=c.nullable))
        elif tid == B.T_BOOL:
            out.append(DeviceColumn(c.name, c.type, bits(c.data), n, validity=bits(c.validity), nullable=c.nullable))
        else:
            w = {B.T_INT32: 4, B.T_DATE32: 4, B.T_UINT32: 4, B.T_INT64: 8, B.T_UINT64: 8, B.T_FLOAT64: 8, B.T_DECIMAL128: 16}[tid]
            out.append(DeviceColumn(c.name, c.type, c.data.view(torch.uint8)[lo * w:], n, validity=bits(c.validity), nullable=c.nullable))
    return DeviceTable(out, n)


def encode_batch(tc, table, codec=0, out=None):
    """One encapsulated RecordBatch message (bytes-like) for a plain Arrow-layout table.  `out`: optional reusable pinned
    uint8 torch tensor; returns a memoryview over the message."""
    import torch
    L = tc.ctx.L
    n = table.num_rows
    arr = (B.gpuq_column * max(1, len(table.columns)))(*[c.to_c() for c in table.columns])
    raw = sum(c.nbytes() for c in table.columns) + 4096 + 64 * len(table.columns)
    if out is None or out.numel() < raw:
        out = torch.empty(raw, dtype=torch.uint8, pin_memory=True)
    ln = C.c_int64(0)
    rc = L.gpuq_ipc_encode_batch(tc.ctx.h, tc.stream_ptr(), arr, len(table.columns), n, int(codec), out.data_ptr(), out.numel(), C.byref(ln))
    if rc == 4:      # a bound that was too tight: the call reports the size it needs
        out = torch.empty(int(ln.value), dtype=torch.uint8, pin_memory=True)
        rc = L.gpuq_ipc_encode_batch(tc.ctx.h, tc.stream_ptr(), arr, len(table.columns), n, int(codec), out.data_ptr(), out.numel(), C.byref(ln))
    _check(L, rc)
    return memoryview(out.numpy())[: int(ln.value)], out


def encoded_size(tc, table, codec=0):
    """Bytes the message would take (the compression runs on the device, nothing is copied back)."""
    L = tc.ctx.L
    arr = (B.gpuq_column * max(1, len(table.columns)))(*[c.to_c() for c in table.columns])
    ln = C.c_int64(0)
    _check(L, L.gpuq_ipc_encode_batch(tc.ctx.h, tc.stream_ptr(), arr, len(table.columns), table.num_rows, int(codec), None, 0, C.byref(ln)))
    return int(ln.value)


def write_ipc_stream(tc, sink, table, batch_size=None, codec=0):
    """Write `table` to `sink` (a path or a binary file object) as an Arrow IPC stream.  Returns (num_batches, num_rows,
    num_bytes) -- the figures of ShuffleWritePartition (shuffle_writer.rs:410-420)."""
    t = _arrow_layout(tc, table)
    schema = _arrow_schema(t)
    own = isinstance(sink, (str, bytes))
    f = open(sink, "wb") if own else sink
    try:
        nbytes = f.write(schema.serialize().to_pybytes())
        nb, n = 0, t.num_rows
        bs = n if not batch_size else int(batch_size)
        lo, buf = 0, None
        while lo < n:
            k = min(bs, n - lo)
            piece = t if (lo == 0 and k == n) else _slice_plain(tc, t, lo, k)
            msg, buf = encode_batch(tc, piece, codec, buf)
            nbytes += f.write(msg)
            nb += 1
            lo += k
        nbytes += f.write(EOS)
    finally:
        if own:
            f.close()
    return nb, n, nbytes


def _fields_of(schema):
    import pyarrow as pa
    out = (B.gpuq_field_info * max(1, len(schema)))()
    types = []
    for i, fl in enumerate(schema):
        ty = fl.type
        if pa.types.is_decimal128(ty):
            tid, p, s = B.T_DECIMAL128, ty.precision, ty.scale
        else:
            m = [(pa.types.is_int32, B.T_INT32), (pa.types.is_int64, B.T_INT64), (pa.types.is_date32, B.T_DATE32), (pa.types.is_float64, B.T_FLOAT64),
                 (pa.types.is_uint32, B.T_UINT32), (pa.types.is_uint64, B.T_UINT64), (pa.types.is_string, B.T_UTF8), (pa.types.is_boolean, B.T_BOOL)]
            tid = next((t for pred, t in m if pred(ty)), None)
            if tid is None:
                raise B.GpuqError(3, "column '%s' of type %s is not supported on device" % (fl.name, ty))
            p = s = 0
        out[i].name = fl.name.encode()[:255]
        out[i].type, out[i].precision, out[i].scale, out[i].nullable, out[i].repr = tid, p, s, int(fl.nullable), B.REPR_ARROW
        types.append(type_json(tid, p, s))
    return out, types


class _Batch:
    """Owner of one gpuq_ipc_batch."""

    def __init__(self, L, h):
        self.L, self.h = L, h

    def __del__(self):
        try:
            if self.h:
                self.L.gpuq_ipc_batch_free(self.h)
                self.h = None
        except Exception:
            pass


def _wrap_batch(tc, L, h, schema, fields, types):
    import torch
    owner = _Batch(L, h)
    n = int(L.gpuq_ipc_batch_num_rows(h))
    cols = []

    def alias(ptr, nb):
        class _A:
            pass
        a = _A()
        a.__cuda_array_interface__ = {"shape": (int(nb),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        a.owner = owner
        return torch.as_tensor(a, device=tc.device)
    for i, fl in enumerate(schema):
        c = B.gpuq_column()
        L.gpuq_ipc_batch_column(h, i, C.byref(c))
        tid = fields[i].type
        vb = ((n + 63) // 64) * 8 + 8
        validity = alias(c.validity, vb) if c.validity else None
        if tid == B.T_UTF8:
            offs = alias(c.offsets, (n + 1) * 4).view(torch.int32)
            dlen = int(offs[n].item()) if n else 0
            cols.append(DeviceColumn(fl.name, "Utf8", alias(c.data, max(16, dlen)), n, offsets=offs, validity=validity, nullable=fl.nullable))
        elif tid == B.T_BOOL:
            cols.append(DeviceColumn(fl.name, "Boolean", alias(c.data, vb), n, validity=validity, nullable=fl.nullable))
        else:
            w = {B.T_INT32: 4, B.T_DATE32: 4, B.T_UINT32: 4, B.T_INT64: 8, B.T_UINT64: 8, B.T_FLOAT64: 8, B.T_DECIMAL128: 16}[tid]
            cols.append(DeviceColumn(fl.name, types[i], alias(c.data, max(1, n) * w + 16), n, validity=validity, nullable=fl.nullable))
    t = DeviceTable(cols, n)
    t._keep = owner
    return t


def read_ipc_stream(tc, source):
    """Read an Arrow IPC stream (path, bytes or memoryview) into ONE device table: every batch of the stream is decoded in a
    single library call (gpuq_ipc_decode_stream), in stream order, as ShuffleReaderExec's consumer sees them.
    Returns (DeviceTable, pyarrow schema)."""
    import numpy as np
    import pyarrow as pa
    L = tc.ctx.L
    data = np.fromfile(source, dtype=np.uint8) if isinstance(source, str) else np.frombuffer(source, dtype=np.uint8)
    addr, total = data.ctypes.data, int(data.size)
    info = gpuq_ipc_info()
    _check(L, L.gpuq_ipc_peek(C.c_void_p(addr), total, C.byref(info)))
    if info.header_type != 1:
        raise B.GpuqError(1, "IPC stream does not start with a Schema message")
    first = info.metadata_bytes + info.body_bytes
    schema = pa.ipc.read_schema(pa.py_buffer(data[:first].tobytes()))
    fields, types = _fields_of(schema)
    h = C.c_void_p()
    _check(L, L.gpuq_ipc_decode_stream(tc.ctx.h, tc.stream_ptr(), C.c_void_p(addr + first), total - first, fields, len(schema), C.byref(h)))
    return _wrap_batch(tc, L, h, schema, fields, types), schema
