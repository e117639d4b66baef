"""Regenerates the fixtures under tests/golden/ from DATA files the reference's own tests hold
(run in the build container, where /root/reference exists; the GPU box only sees the outputs).
  alltypes_plain.arrow  <- ballista/client/testdata/alltypes_plain.parquet  (8 rows; KATs context.rs:762-967)
  tpch10/*.tbl          <- ballista/scheduler/testdata/**.tbl               (8 TPC-H tables x 10 rows)
  shuffle_data.arrow    <- ballista/core/tests/data.arrow                   (561-row Utf8 shuffle file, Arrow IPC stream with LZ4_FRAME
                                                                             buffers; read 1000 x by async_reader/mod.rs:331-357)
  alltypes_plain.parquet <- ballista/client/testdata/alltypes_plain.parquet (the file itself: input of the device Parquet decoder)
  single_nan.parquet     <- ballista/client/testdata/single_nan.parquet     (one optional DOUBLE, SNAPPY pages: the Snappy unpack path)
  aggregate_test_100.csv <- examples/testdata/aggregate_test_100.csv        (100 rows x 13 columns; input of the device CSV parser)
Only data is copied -- no reference source text."""
import os
import shutil

import pyarrow as pa
import pyarrow.parquet as pq

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

t = pq.read_table(os.path.join(REF, "ballista/client/testdata/alltypes_plain.parquet"))
with pa.OSFile(os.path.join(HERE, "alltypes_plain.arrow"), "wb") as f, pa.ipc.new_file(f, t.schema) as w:
    w.write_table(t)
os.makedirs(os.path.join(HERE, "tpch10"), exist_ok=True)
base = os.path.join(REF, "ballista/scheduler/testdata")
for table in sorted(os.listdir(base)):
    for fn in sorted(os.listdir(os.path.join(base, table))):
        dst = "%s.tbl" % table if fn == table + ".tbl" else "%s.%s" % (table, fn)
        shutil.copyfile(os.path.join(base, table, fn), os.path.join(HERE, "tpch10", dst))
shutil.copyfile(os.path.join(REF, "ballista/core/tests/data.arrow"), os.path.join(HERE, "shuffle_data.arrow"))
shutil.copyfile(os.path.join(REF, "ballista/client/testdata/alltypes_plain.parquet"), os.path.join(HERE, "alltypes_plain.parquet"))
shutil.copyfile(os.path.join(REF, "ballista/client/testdata/single_nan.parquet"), os.path.join(HERE, "single_nan.parquet"))
shutil.copyfile(os.path.join(REF, "examples/testdata/aggregate_test_100.csv"), os.path.join(HERE, "aggregate_test_100.csv"))
print(sorted(os.listdir(os.path.join(HERE, "tpch10"))))
