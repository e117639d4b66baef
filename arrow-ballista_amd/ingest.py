"""Streaming ingest of a partition (include/gpuq.h "streaming ingest", csrc/ingest.cpp): host Arrow RecordBatches are appended to
one set of device columns by a pool of staging threads inside libgpuq, while consumers already read the landed prefix."""
import ctypes as C

from . import binding as B
from .native import ArrowArray, ArrowSchema
from .table import DeviceColumn, DeviceTable, type_json, type_width


class Ingest:
    def __init__(self, tc, schema, max_rows, max_utf8_bytes=0, n_threads=0):
        """schema: pyarrow.Schema of the batches that will be pushed."""
        import pyarrow as pa
        self.tc, self.schema, self.max_rows = tc, schema, int(max_rows)
        L = tc.ctx.L
        cs = ArrowSchema()
        pa.struct(list(schema))._export_to_c(C.addressof(cs))
        h = C.c_void_p()
        rc = L.gpuq_ingest_create(tc.ctx.h, C.addressof(cs), self.max_rows, int(max_utf8_bytes), int(n_threads), C.byref(h))
        if cs.release:
            C.CFUNCTYPE(None, C.c_void_p)(cs.release)(C.addressof(cs))
        self._check(rc)
        self.h = h
        n = C.c_int(0)
        self._check(L.gpuq_ingest_columns(self.h, None, None, 0, C.byref(n)))
        self._cols, self._fields = (B.gpuq_column * n.value)(), (B.gpuq_field_info * n.value)()
        self._check(L.gpuq_ingest_columns(self.h, self._cols, self._fields, n.value, C.byref(n)))
        self._utf8_cap = int(max_utf8_bytes)

    def _check(self, rc):
        if rc != 0:
            raise B.GpuqError(rc, (self.tc.ctx.L.gpuq_ingest_last_error() or b"").decode())

    def push(self, batch):
        """Queue one pyarrow.RecordBatch; it is moved into the library (released there once its copies have landed)."""
        arr = ArrowArray()
        batch._export_to_c(C.addressof(arr))
        self._check(self.tc.ctx.L.gpuq_ingest_push(self.h, C.addressof(arr)))

    def rows_landed(self):
        n = C.c_int64(0)
        self._check(self.tc.ctx.L.gpuq_ingest_rows_landed(self.h, C.byref(n)))
        return int(n.value)

    def wait_rows(self, rows):
        n = C.c_int64(0)
        self._check(self.tc.ctx.L.gpuq_ingest_wait_rows(self.h, int(rows), C.byref(n)))
        return int(n.value)

    def stats(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self.tc.ctx.L.gpuq_ingest_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"rows_pushed": int(a.value), "rows_landed": int(b.value), "bytes_copied": int(c.value)}

    def table(self, row0=0, rows=None):
        """DeviceTable over rows [row0, row0 + rows) of the ingested columns (no copy; row0 a multiple of 8 when a column is
        nullable / Boolean).  Only rows that have landed may be read."""
        import torch
        rows = self.rows_landed() - row0 if rows is None else int(rows)
        owner = self
        out = []

        def alias(ptr, nb):
            class _A:
                pass
            a = _A()
            a.__cuda_array_interface__ = {"shape": (int(max(nb, 1)),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
            a.owner = owner
            return torch.as_tensor(a, device=self.tc.device)
        for i in range(len(self._cols)):
            c, f = self._cols[i], self._fields[i]
            ty = type_json(f.type, f.precision, f.scale)
            validity = None
            if c.validity:
                assert row0 % 8 == 0
                validity = alias(c.validity + row0 // 8, (rows + 7) // 8 + 8)
            if f.type == B.T_UTF8:
                offs = alias(c.offsets + 4 * row0, (rows + 1) * 4).view(torch.int32)
                data = alias(c.data, self._utf8_cap + 16)          # offsets are absolute positions in the partition's byte buffer
                out.append(DeviceColumn(f.name.decode(), ty, data, rows, offsets=offs, validity=validity, nullable=bool(f.nullable)))
            elif f.type == B.T_BOOL:
                assert row0 % 8 == 0
                out.append(DeviceColumn(f.name.decode(), ty, alias(c.data + row0 // 8, (rows + 7) // 8 + 8), rows, validity=validity, nullable=bool(f.nullable)))
            else:
                w = type_width(ty)
                out.append(DeviceColumn(f.name.decode(), ty, alias(c.data + row0 * w, rows * w + 16), rows, validity=validity, nullable=bool(f.nullable)))
        t = DeviceTable(out, rows)
        t._keep = self
        return t

    def close(self):
        if getattr(self, "h", None):
            self.tc.ctx.L.gpuq_ingest_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass
