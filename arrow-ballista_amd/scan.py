"""Scan-side decode (include/gpuq.h "scan-side decode", csrc/scanfmt.cpp + kernels_scanfmt.hip): the CsvExec / ParquetExec
leaves of the reference's TPC-H plans (benchmarks/src/bin/tpch.rs:801-862).  File bytes go to the device once; the columns are
parsed there.  No host fallback: without the HIP library these raise."""
import ctypes as C
import mmap
import os

from . import binding as B
from .table import DeviceColumn, DeviceTable, type_id, type_json


def _check(L, rc):
    if rc != 0:
        raise B.GpuqError(rc, (L.gpuq_scan_last_error() or b"").decode())


def _host_bytes(src):
    """bytes-like or a path -> (object keeping the memory alive, address, length).  Paths are mmap'ed, not read."""
    if isinstance(src, (str, os.PathLike)):
        f = open(src, "rb")
        n = os.fstat(f.fileno()).st_size
        if n == 0:
            return (f, b""), 0, 0
        m = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_COPY)      # private mapping: writable view for from_buffer, file untouched
        buf = (C.c_ubyte * n).from_buffer(m)
        return (f, m, buf), C.addressof(buf), n
    import numpy as np
    mv = memoryview(src).cast("B")
    n = mv.nbytes
    if n == 0:
        return mv, 0, 0
    arr = np.frombuffer(mv, dtype=np.uint8)      # no copy, read-only buffers included: the library only reads the bytes
    return (mv, arr), arr.ctypes.data, n


def _wrap_table(tc, h):
    """gpuq_table -> DeviceTable aliasing the library's buffers (names / types from the table's own field info)."""
    import torch
    from .parallel import _OwnedTable
    L = tc.ctx.L
    n = int(L.gpuq_table_num_rows(h))
    owner = _OwnedTable(L, h)

    def alias(ptr, nb):
        class _A:
            pass
        a = _A()
        a.__cuda_array_interface__ = {"shape": (int(max(nb, 1)),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        a.owner = owner
        return torch.as_tensor(a, device=tc.device)
    bm = ((n + 63) // 64) * 8 + 8
    cols = []
    for i in range(int(L.gpuq_table_num_columns(h))):
        cc, ff = B.gpuq_column(), B.gpuq_field_info()
        L.gpuq_table_column(h, i, C.byref(cc), C.byref(ff))
        ty = type_json(ff.type, ff.precision, ff.scale)
        if cc.offsets:
            offs = alias(cc.offsets, (n + 1) * 4).view(torch.int32)
            nbytes = int(offs[n].item()) if n else 0
            data = alias(cc.data, nbytes + 16)
        else:
            offs = None
            data = alias(cc.data, bm if ff.type == B.T_BOOL else n * ff.width + 16)
        validity = alias(cc.validity, bm) if cc.validity else None
        cols.append(DeviceColumn(ff.name.decode(), ty, data, n, offsets=offs, validity=validity, nullable=bool(ff.nullable)))
    t = DeviceTable(cols, n)
    t._keep = owner
    return t


def read_csv(tc, src, schema, projection=None, delimiter=",", has_header=False, quote='"'):
    """schema: [(name, type, nullable)] for EVERY field of a line, type as in DeviceColumn.type ("Int64", "Date32",
    {"Decimal128": [15, 2]}, "Utf8" ...).  projection: file column indices or names.  Returns a DeviceTable."""
    L = tc.ctx.L
    keep, addr, n = _host_bytes(src)
    fields = (B.gpuq_field_info * len(schema))()
    for i, (name, ty, nullable) in enumerate(schema):
        tid, p, s = type_id(ty)
        fields[i].name = name.encode()[:255]
        fields[i].type, fields[i].precision, fields[i].scale, fields[i].nullable = tid, p, s, 1 if nullable else 0
    proj = None
    if projection is not None:
        names = [s[0] for s in schema]
        idx = [names.index(p) if isinstance(p, str) else int(p) for p in projection]
        proj = (C.c_int32 * max(1, len(idx)))(*idx)
    opt = B.gpuq_csv_options(delimiter.encode(), quote.encode(), 1 if has_header else 0)
    h = C.c_void_p()
    _check(L, L.gpuq_csv_decode(tc.ctx.h, tc.stream_ptr(), C.c_void_p(addr), n, fields, len(schema), proj, 0 if proj is None else len(idx), C.byref(opt), C.byref(h)))
    del keep
    return _wrap_table(tc, h)


def read_csv_chunked(tc, src, schema, projection=None, delimiter=",", has_header=False, chunk_bytes=1 << 30):
    """Files beyond one call's 4 GiB (SF100's lineitem.tbl is 75 GB): decoded in pieces cut at line boundaries on the host (a scan
    backwards for the last newline of every chunk -- no value is looked at).  Returns the pieces as a list of DeviceTables: the
    partitions of the scan leaf (MemoryExec(parts))."""
    keep, addr, n = _host_bytes(src)
    if n == 0:
        return [read_csv(tc, b"", schema, projection, delimiter, has_header)]
    mv = memoryview((C.c_ubyte * n).from_address(addr)).cast("B")
    parts, at, first = [], 0, True
    while at < n:
        end = min(n, at + int(chunk_bytes))
        if end < n:
            cut = bytes(mv[at:end]).rfind(b"\n")
            if cut < 0:
                raise B.GpuqError(1, "csv: a line longer than chunk_bytes")
            end = at + cut + 1
        parts.append(read_csv(tc, mv[at:end], schema, projection, delimiter, has_header and first))
        at, first = end, False
    del keep
    return parts


def parquet_schema(L, src):
    """[(name, type or None when the device has no decoder, nullable)], num_rows -- from the footer, host only."""
    keep, addr, n = _host_bytes(src)
    k, rows = C.c_int(0), C.c_int64(0)
    _check(L, L.gpuq_parquet_schema(C.c_void_p(addr), n, None, 0, C.byref(k), C.byref(rows)))
    fields = (B.gpuq_field_info * max(1, k.value))()
    _check(L, L.gpuq_parquet_schema(C.c_void_p(addr), n, fields, k.value, C.byref(k), C.byref(rows)))
    del keep
    return [(f.name.decode(), None if f.type < 0 else type_json(f.type, f.precision, f.scale), bool(f.nullable)) for f in fields[:k.value]], int(rows.value)


def parquet_row_groups(L, src):
    """rows of every row group of the file (host only)."""
    keep, addr, n = _host_bytes(src)
    k = C.c_int(0)
    _check(L, L.gpuq_parquet_row_groups(C.c_void_p(addr), n, None, 0, C.byref(k)))
    rows = (C.c_int64 * max(1, k.value))()
    _check(L, L.gpuq_parquet_row_groups(C.c_void_p(addr), n, rows, k.value, C.byref(k)))
    del keep
    return [int(rows[i]) for i in range(k.value)]


def read_parquet(tc, src, columns=None, row_groups=None):
    """row_groups: indices of the row groups to decode (the caller's pruning), in that order; None = all."""
    L = tc.ctx.L
    keep, addr, n = _host_bytes(src)
    names = None
    if columns is not None:
        names = (C.c_char_p * max(1, len(columns)))(*[c.encode() for c in columns])
    h = C.c_void_p()
    rg = None if row_groups is None else (C.c_int32 * max(1, len(row_groups)))(*[int(x) for x in row_groups])
    _check(L, L.gpuq_parquet_decode_groups(tc.ctx.h, tc.stream_ptr(), C.c_void_p(addr), n, names, 0 if columns is None else len(columns), rg,
                                           0 if row_groups is None else len(row_groups), C.byref(h)))
    del keep
    return _wrap_table(tc, h)
