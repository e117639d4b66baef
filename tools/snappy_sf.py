"""Snappy Parquet decode at a given scale factor: best of 3, device vs pyarrow (host).  usage: snappy_sf.py SF"""
import io, json, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pyarrow as pa
import pyarrow.parquet as pq
import arrow_ballista_amd as g
import tpch_util as T
from arrow_ballista_amd import scan
sf = int(sys.argv[1])
tc = g.TaskContext(device=0)
n = T.LINEITEM_ROWS.get(sf, 6_000_000 * sf)
li = T.lineitem_host_to_arrow(T.gen_lineitem_host(n), n)
li = li.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in li.schema]))
buf = io.BytesIO()
pq.write_table(li, buf, compression="SNAPPY", use_dictionary=True, data_page_size=1 << 20, row_group_size=1 << 20)
sfile = buf.getvalue()
del buf
best = None
for _ in range(3):
    tc.sync(); t0 = time.perf_counter(); r = scan.read_parquet(tc, sfile); tc.sync(); dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
rows = r.num_rows
ok = r.to_arrow(tc.ctx).column("l_orderkey").equals(li.column("l_orderkey")) if sf <= 1 else None
del r
t0 = time.perf_counter(); pq.read_table(pa.BufferReader(sfile)); host = time.perf_counter() - t0
print(json.dumps({"sf": sf, "rows": rows, "file_bytes": len(sfile), "device_ms": best * 1e3, "rows_per_s": rows / best, "pyarrow_host_ms": host * 1e3, "pj": os.environ.get("GPUQ_SNAPPY_PJ", "1"), "orderkey_equal": ok}))
