"""Device-resident Arrow-layout tables.  torch tensors own the HBM; Arrow <-> device is a byte copy."""
import ctypes as C

from . import binding as B

_TYPE_IDS = {"Boolean": B.T_BOOL, "Int32": B.T_INT32, "Int64": B.T_INT64, "Date32": B.T_DATE32, "Float64": B.T_FLOAT64,
             "Utf8": B.T_UTF8, "UInt32": B.T_UINT32, "UInt64": B.T_UINT64, "Int8": B.T_INT8, "Int16": B.T_INT16, "UInt8": B.T_UINT8,
             "UInt16": B.T_UINT16, "Float32": B.T_FLOAT32, "Date64": B.T_DATE64}
_ID_TYPES = {v: k for k, v in _TYPE_IDS.items()}
_TIME_UNITS = ["Second", "Millisecond", "Microsecond", "Nanosecond"]
_PA_UNITS = ["s", "ms", "us", "ns"]


def type_id(t):
    """(gpuq_type, precision, scale) of a type descriptor (the JSON form of datafusion.proto's ArrowType)."""
    if isinstance(t, dict):
        if "Timestamp" in t:
            u = t["Timestamp"][0] if isinstance(t["Timestamp"], (list, tuple)) else t["Timestamp"]
            return B.T_TIMESTAMP, (_TIME_UNITS.index(u) if u in _TIME_UNITS else _PA_UNITS.index(u)), 0
        if "Dictionary" in t:
            return type_id(t["Dictionary"][1])
        return B.T_DECIMAL128, int(t["Decimal128"][0]), int(t["Decimal128"][1])
    if t == "LargeUtf8":
        return B.T_UTF8, 0, 0
    return _TYPE_IDS[t], 0, 0


def type_json(tid, p=0, s=0):
    if tid == B.T_DECIMAL128:
        return {"Decimal128": [int(p), int(s)]}
    if tid == B.T_TIMESTAMP:
        return {"Timestamp": [_TIME_UNITS[int(p) & 3], None]}
    return _ID_TYPES[tid]


def type_width(t):
    tid = type_id(t)[0]
    return {B.T_INT32: 4, B.T_DATE32: 4, B.T_UINT32: 4, B.T_INT64: 8, B.T_UINT64: 8, B.T_FLOAT64: 8, B.T_DECIMAL128: 16, B.T_UTF8: 16,
            B.T_INT8: 1, B.T_UINT8: 1, B.T_INT16: 2, B.T_UINT16: 2, B.T_FLOAT32: 4, B.T_TIMESTAMP: 8, B.T_DATE64: 8}.get(tid, 0)


def arrow_type_json(t):
    """Type descriptor of a pyarrow type, or None.  LargeUtf8 / Binary / Dictionary columns are described by what they are inside
    the engine (Utf8 / the value type): from_arrow converts them where they enter."""
    import pyarrow as pa
    if pa.types.is_decimal128(t):
        return {"Decimal128": [t.precision, t.scale]}
    if pa.types.is_timestamp(t):
        return {"Timestamp": [_TIME_UNITS[_PA_UNITS.index(t.unit)], t.tz]}
    if pa.types.is_dictionary(t):
        return arrow_type_json(t.value_type)
    return {pa.int32(): "Int32", pa.int64(): "Int64", pa.date32(): "Date32", pa.float64(): "Float64", pa.string(): "Utf8",
            pa.large_string(): "Utf8", pa.bool_(): "Boolean", pa.uint32(): "UInt32", pa.uint64(): "UInt64", pa.binary(): "Utf8",
            pa.int8(): "Int8", pa.int16(): "Int16", pa.uint8(): "UInt8", pa.uint16(): "UInt16", pa.float32(): "Float32",
            pa.date64(): "Date64"}.get(t)


def json_arrow_type(t):
    """pyarrow type of a type descriptor."""
    import pyarrow as pa
    tid, p, s = type_id(t)
    if tid == B.T_DECIMAL128:
        return pa.decimal128(p, s)
    if tid == B.T_TIMESTAMP:
        tz = t["Timestamp"][1] if isinstance(t, dict) and isinstance(t.get("Timestamp"), (list, tuple)) and len(t["Timestamp"]) > 1 else None
        return pa.timestamp(_PA_UNITS[p], tz=tz)
    return {B.T_INT32: pa.int32(), B.T_INT64: pa.int64(), B.T_DATE32: pa.date32(), B.T_FLOAT64: pa.float64(), B.T_UINT32: pa.uint32(),
            B.T_UINT64: pa.uint64(), B.T_UTF8: pa.string(), B.T_BOOL: pa.bool_(), B.T_INT8: pa.int8(), B.T_INT16: pa.int16(),
            B.T_UINT8: pa.uint8(), B.T_UINT16: pa.uint16(), B.T_FLOAT32: pa.float32(), B.T_DATE64: pa.date64()}[tid]


RECORD_HEADER = 256


def record_layout(specs, n):
    """Byte layout of `n`-row output columns inside ONE device allocation ("record"): a 256-byte header (word 0 = row
    count, filled in by whoever ships the record), then per column its data and, when nullable, its validity bitmap, each
    256-byte aligned.  specs: [(width_bytes or 0 for bit-packed Boolean, nullable)].  Pure function of (specs, n): every
    rank computes the same layout, which is what lets parallel.allgather_table ship a whole result in one collective.
    Returns (total_bytes, [(data_off, data_bytes, validity_off or -1, validity_bytes)])."""
    bm = ((n + 63) // 64) * 8 + 8
    out, off = [], RECORD_HEADER
    for width, nullable in specs:
        dbytes = bm if width == 0 else max(1, n) * width + 16
        doff = off
        off += (dbytes + 255) & ~255
        voff = -1
        if nullable:
            voff = off
            off += (bm + 255) & ~255
        out.append((doff, dbytes, voff, bm if nullable else 0))
    return off, out


def _torch():
    import torch
    return torch


class DeviceColumn:
    """One column in HBM.  repr 0 = Arrow layout (data[/offsets][/validity]); repr 1 = Utf8 PACKED15 (16 B per row)."""

    def __init__(self, name, type, data, length, offsets=None, validity=None, nullable=None, repr=B.REPR_ARROW):
        self.name, self.type, self.data, self.length = name, type, data, int(length)
        self.offsets, self.validity, self.repr = offsets, validity, repr
        self.nullable = (validity is not None) if nullable is None else bool(nullable)

    def field(self, side=0, dense=False):
        f = {"name": self.name, "type": self.type, "nullable": bool(self.nullable), "side": int(side)}
        if side > 0 and dense:
            f["dense"] = 1
        if self.repr == B.REPR_PACKED15:
            f["raw128"] = 1
        return f

    def to_c(self):
        """The C view of this column.  Built once: a column's buffers never change after construction (set_length only
        narrows the row count), and building ctypes structs is what a small query's host time consists of."""
        c = self.__dict__.get("_c")
        if c is None:
            tid, p, s = type_id(self.type)
            c = B.gpuq_column()
            c.type, c.precision, c.scale, c.repr = tid, p, s, self.repr
            c.data = self.data.data_ptr() if self.data is not None and self.data.numel() > 0 else None
            c.offsets = self.offsets.data_ptr() if self.offsets is not None else None
            c.validity = self.validity.data_ptr() if self.validity is not None else None
            self._c = c
        c.length = self.length
        return c

    def set_length(self, n):
        self.length = int(n)

    def nbytes(self):
        n = 0
        for t in (self.data, self.offsets, self.validity):
            if t is not None:
                n += t.numel() * t.element_size()
        return n


class DeviceTable:
    """Columns + optional index vectors (a late-materialised view: column i is read at via[side_i-1][pos])."""

    def __init__(self, columns, num_rows, via=None, sides=None, dense=False):
        self.columns = list(columns)
        self.num_rows = int(num_rows)
        self.via = list(via or [])            # uint32-as-int32 tensors of length num_rows
        self.sides = list(sides) if sides is not None else [0] * len(self.columns)
        self.dense = bool(dense)              # no index vector holds NULL_ROW: reading through them adds no nulls

    # ---- schema
    def schema(self):
        return [c.field(s, self.dense) for c, s in zip(self.columns, self.sides)]

    def plain_schema(self):
        return [{"name": c.name, "type": c.type, "nullable": bool(c.nullable or (s > 0 and not self.dense))} for c, s in zip(self.columns, self.sides)]

    def column(self, name):
        for c in self.columns:
            if c.name == name:
                return c
        raise KeyError(name)

    def is_view(self):
        return len(self.via) > 0

    def input_struct(self):
        """(gpuq_input, keepalive) for a C call.  Cached: a table is immutable once built."""
        cached = self.__dict__.get("_inp")
        if cached is not None and cached[2] == self.num_rows:
            return cached[0], cached[1]
        arr = (B.gpuq_column * max(1, len(self.columns)))()
        for i, c in enumerate(self.columns):
            arr[i] = c.to_c()
        inp = B.gpuq_input()
        inp.cols = C.cast(arr, C.POINTER(B.gpuq_column))
        inp.n_cols = len(self.columns)
        inp.n_via = len(self.via)
        inp.n_rows = self.num_rows
        for k, v in enumerate(self.via):
            inp.via[k] = v.data_ptr() if v.numel() > 0 else None
        self._inp = (inp, arr, self.num_rows)
        return inp, arr

    def nbytes(self):
        return sum(c.nbytes() for c in self.columns)

    # ---- Arrow interchange
    @staticmethod
    def from_arrow(tbl, device="cuda:0"):
        """Copy a pyarrow Table / RecordBatch to the device, buffer for buffer (Arrow physical layout)."""
        import pyarrow as pa
        torch = _torch()
        if isinstance(tbl, pa.RecordBatch):
            tbl = pa.Table.from_batches([tbl])
        cols = []
        for name, chunked in zip(tbl.schema.names, tbl.columns):
            arr = chunked.combine_chunks() if chunked.num_chunks != 1 else chunked.chunk(0)
            # (this is the tests' and benches' way in; the C ABI's own is gpuq_import_arrow / gpuq_ingest_*, which convert LargeUtf8
            # offsets and decode dictionaries themselves)
            if pa.types.is_dictionary(arr.type):
                arr = arr.cast(arr.type.value_type)
            if pa.types.is_large_string(arr.type) or pa.types.is_binary(arr.type):
                arr = arr.cast(pa.string())
            if arr.offset != 0:
                arr = pa.concat_arrays([arr])
            t = arr.type
            tj = arrow_type_json(t)
            if tj is None:
                raise B.GpuqError(3, "Arrow type %s is not supported on device" % t)
            bufs = arr.buffers()

            def up(b, min_bytes=0):
                if b is None:
                    return None
                host = torch.frombuffer(memoryview(b), dtype=torch.uint8) if b.size > 0 else torch.zeros(0, dtype=torch.uint8)
                dev = torch.zeros(max(host.numel(), min_bytes) + 16, dtype=torch.uint8, device=device)
                dev[: host.numel()].copy_(host)
                return dev
            validity = up(bufs[0]) if arr.null_count > 0 else None
            if tj == "Utf8":
                offsets = up(bufs[1], 4).view(torch.int32) if bufs[1] is not None else torch.zeros(4, dtype=torch.int32, device=device)
                data = up(bufs[2]) if bufs[2] is not None else torch.zeros(16, dtype=torch.uint8, device=device)
                cols.append(DeviceColumn(name, tj, data, len(arr), offsets=offsets, validity=validity,
                                         nullable=tbl.schema.field(name).nullable))
            else:
                data = up(bufs[1])
                cols.append(DeviceColumn(name, tj, data, len(arr), validity=validity, nullable=tbl.schema.field(name).nullable))
        return DeviceTable(cols, tbl.num_rows)

    def to_arrow(self, ctx=None):
        """Copy back to a pyarrow Table (materialised tables only)."""
        import numpy as np
        import pyarrow as pa
        torch = _torch()
        if self.is_view():
            raise B.GpuqError(1, "materialise the view before to_arrow()")
        arrays, names = [], []
        n = self.num_rows
        for c in self.columns:
            tid, p, s = type_id(c.type)
            validity = None
            null_count = 0
            if c.validity is not None:
                vb = c.validity.cpu().numpy()[: (n + 7) // 8].copy()
                validity = pa.py_buffer(vb.tobytes())
                bits = np.unpackbits(vb, bitorder="little")[:n]
                null_count = int(n - bits.sum())
            if tid == B.T_UTF8:
                if c.repr == B.REPR_PACKED15:
                    if ctx is None:
                        raise B.GpuqError(1, "a Context is needed to unpack PACKED15 strings")
                    off = torch.zeros(n + 4, dtype=torch.int32, device=c.data.device)
                    dat = torch.zeros(max(16, n * 15), dtype=torch.uint8, device=c.data.device)
                    dl = C.c_int64(0)
                    ctx.check(ctx.L.gpuq_unpack_utf8(ctx.h, None, c.data.data_ptr() if n else None, n, off.data_ptr(), dat.data_ptr(),
                                                     dat.numel(), C.byref(dl)))
                    ob = off.cpu().numpy()[: n + 1].tobytes()
                    db = dat.cpu().numpy()[: dl.value].tobytes()
                else:
                    ob = c.offsets.cpu().numpy()[: n + 1].tobytes()
                    last = int(c.offsets[n].item()) if n else 0
                    db = c.data.cpu().numpy()[:last].tobytes()
                arrays.append(pa.Array.from_buffers(pa.string(), n, [validity, pa.py_buffer(ob), pa.py_buffer(db)], null_count=null_count))
            elif tid == B.T_BOOL:
                db = c.data.cpu().numpy().view(np.uint8)[: (n + 7) // 8].tobytes()
                arrays.append(pa.Array.from_buffers(pa.bool_(), n, [validity, pa.py_buffer(db)], null_count=null_count))
            else:
                pt = json_arrow_type(c.type)
                w = type_width(c.type)
                db = c.data.cpu().numpy().view(np.uint8)[: n * w].tobytes()
                arrays.append(pa.Array.from_buffers(pt, n, [validity, pa.py_buffer(db)], null_count=null_count))
            names.append(c.name)
        return pa.Table.from_arrays(arrays, names=names)
