"""Parquet decode of lineitem at a given scale factor under a codec: best of 3, device vs pyarrow (host).  usage: codec_sf.py SF CODEC [level]"""
import io, json, os, sys, time
_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests"))
import pyarrow as pa
import pyarrow.parquet as pq
import arrow_ballista_amd as g
import tpch_util as T
from arrow_ballista_amd import scan
sf = int(sys.argv[1]); codec = sys.argv[2]; lvl = {"compression_level": int(sys.argv[3])} if len(sys.argv) > 3 else {}
tc = g.TaskContext(device=0)
n = T.LINEITEM_ROWS.get(sf, 6_000_000 * sf)
li = T.lineitem_host_to_arrow(T.gen_lineitem_host(n), n)
li = li.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in li.schema]))
buf = io.BytesIO()
pq.write_table(li, buf, compression=codec, **lvl, use_dictionary=True, data_page_size=1 << 20, row_group_size=1 << 20)
sfile = buf.getvalue()
del buf
best = None
for _ in range(3):
    tc.sync(); t0 = time.perf_counter(); r = scan.read_parquet(tc, sfile); tc.sync(); dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
rows = r.num_rows
ok = r.to_arrow(tc.ctx).column("l_orderkey").equals(li.column("l_orderkey")) if sf <= 1 else None
del r
per_col = None
if os.environ.get("CODEC_PER_COLUMN"):      # one column at a time: which column's pages are the slow ones
    per_col = {}
    for name in li.schema.names:
        bc = None
        for _ in range(2):
            tc.sync(); t0 = time.perf_counter(); rr = scan.read_parquet(tc, sfile, [name]); tc.sync(); dt = time.perf_counter() - t0; del rr
            bc = dt if bc is None or dt < bc else bc
        per_col[name] = round(bc * 1e3, 2)
t0 = time.perf_counter(); pq.read_table(pa.BufferReader(sfile)); host = time.perf_counter() - t0
print(json.dumps({"sf": sf, "codec": codec, "level": lvl.get("compression_level"), "rows": rows, "file_bytes": len(sfile), "device_ms": best * 1e3, "rows_per_s": rows / best, "pyarrow_host_ms": host * 1e3, "orderkey_equal": ok, "per_column_ms": per_col}))
