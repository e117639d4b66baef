"""SF1 lineitem written as Parquet under the codec in argv[1] (default ZSTD), decoded three times on the device (run under rocprofv3 --kernel-trace --stats)."""
import io
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pyarrow as pa
import pyarrow.parquet as pq
import arrow_ballista_amd as g
import tpch_util as T
from arrow_ballista_amd import scan
tc = g.TaskContext(device=0)
n = T.LINEITEM_ROWS[1]
li = T.lineitem_host_to_arrow(T.gen_lineitem_host(n), n)
li = li.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in li.schema]))
buf = io.BytesIO()
pq.write_table(li, buf, compression=(sys.argv[1] if len(sys.argv) > 1 else "ZSTD"), use_dictionary=True, data_page_size=1 << 20, row_group_size=1 << 20)
sfile = buf.getvalue()
for _ in range(3):
    r = scan.read_parquet(tc, sfile)
    tc.sync()
print("rows", r.num_rows, "file bytes", len(sfile))
