"""One 16 M-row batch written by Arrow C++ (LZ4 frames with LINKED 64 KB blocks) decoded on the device: time and equality."""
import io, json, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import pyarrow as pa
import arrow_ballista_amd as g
from arrow_ballista_amd import shuffle as S
tc = g.TaskContext(device=0)
n = 1 << 24
r = np.random.default_rng(1)
t = pa.table({"k": pa.array((np.arange(n) // 7).astype(np.int64)), "v": pa.array(r.integers(0, 50, n).astype(np.int32)), "s": pa.array(np.array(["AIR", "MAIL", "SHIP", "TRUCK"])[r.integers(0, 4, n)])})
sink = io.BytesIO()
with pa.ipc.new_stream(sink, t.schema, options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
    w.write_batch(t.combine_chunks().to_batches()[0])
raw = sink.getvalue()
best = None
for _ in range(3):
    tc.sync(); t0 = time.perf_counter(); got, _ = S.read_ipc_stream(tc, raw); tc.sync(); dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
ok = got.to_arrow(tc.ctx).equals(t)
t0 = time.perf_counter(); pa.ipc.open_stream(raw).read_all(); host = time.perf_counter() - t0
print(json.dumps({"rows": n, "stream_bytes": len(raw), "device_ms": best * 1e3, "equal": bool(ok), "arrow_cpp_host_ms": host * 1e3, "pj": os.environ.get("GPUQ_LZ4_PJ", "1")}))
