"""Streaming ingest through the C ABI (include/gpuq.h "streaming ingest", csrc/ingest.cpp): >= 1000 host RecordBatches pushed
into one partition by the library's staging threads, consumed while they land."""
import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd.ingest import Ingest
import tpch_util as T

pytestmark = pytest.mark.gpu


def test_thousand_batches_of_every_column_kind_land_bit_for_bit(tc):
    """1100 batches x 4096 rows + one ragged batch of 777: Int64, Int32, Date32, Float64, Decimal128, Boolean, Utf8 (empty, short
    and > 15-byte strings), a nullable Int64 and a nullable Utf8; batches sliced out of a bigger table (non-zero Arrow offsets).
    The device columns, exported back through gpuq_export_arrow, equal the concatenation of what was pushed."""
    import decimal
    r = np.random.default_rng(3)
    nb, rows = 1100, 4096
    n = nb * rows + 777
    words = np.array(["", "a", "BUILDING", "a-string-longer-than-fifteen-bytes", "x" * 40])
    tbl = pa.table({
        "k": pa.array(r.integers(-2**62, 2**62, n), type=pa.int64()),
        "i": pa.array(r.integers(-2**31, 2**31 - 1, n).astype(np.int32)),
        "d": pa.array(r.integers(8000, 11000, n).astype(np.int32)).cast(pa.date32()),
        "f": pa.array(r.normal(0, 1e6, n)),
        "dec": pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in r.integers(-10**12, 10**12, 4096)] * (n // 4096 + 1), type=pa.decimal128(15, 2)).slice(0, n),
        "b": pa.array(r.integers(0, 2, n).astype(bool)),
        "s": pa.array(words[r.integers(0, len(words), n)]),
        "nk": pa.array(r.integers(0, 1000, n), type=pa.int64(), mask=r.random(n) < 0.1),
        "ns": pa.array(words[r.integers(0, len(words), n)], mask=r.random(n) < 0.2),
    })
    fields = [pa.field(f.name, f.type, nullable=f.name in ("nk", "ns")) for f in tbl.schema]
    tbl = tbl.cast(pa.schema(fields)).combine_chunks()
    nbytes = max(sum(len(x) for x in tbl.column(c).to_pylist() if x) for c in ("s", "ns")) + 1024
    ing = Ingest(tc, tbl.schema, n, nbytes, n_threads=6)
    pushed = 0
    for bi in range(nb + 1):
        k = rows if bi < nb else 777
        b = tbl.slice(pushed, k).to_batches()[0]
        assert b.num_rows == k
        ing.push(b)
        pushed += k
    assert ing.wait_rows(n) == n
    st = ing.stats()
    assert st["rows_pushed"] == n and st["rows_landed"] == n and st["bytes_copied"] > 40 * n
    got = ing.table(0, n).to_arrow(tc.ctx)
    assert got.num_rows == n
    for name in tbl.schema.names:
        assert got.column(name).combine_chunks().equals(tbl.column(name).combine_chunks()), name
    # over capacity -> GPUQ_ERR_CAPACITY, nothing is corrupted
    with pytest.raises(g.GpuqError) as e:
        ing.push(tbl.slice(0, 8).to_batches()[0])
    assert e.value.status == 4
    ing.close()


def test_q1_consumes_the_landed_prefix_while_later_batches_are_in_flight(tc):
    """BASELINE configs[1] shape at test size: q1's 7 lineitem columns arrive as 1024 batches of 8192 rows.  Every 128 batches
    the consumer waits for that prefix (gpuq_ingest_wait_rows) and runs the fused filter + projection + PARTIAL aggregate over
    just those rows on the compute stream while the staging threads keep copying; the partial states of the 8 chunks are merged
    by the final aggregate.  Rows = the oracle's q1 over all rows."""
    nb, rows = 1024, 8192
    n = nb * rows
    host = T.lineitem_host_to_arrow(T.gen_lineitem_host(n), n).select(["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"])
    host = host.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in host.schema])).combine_chunks()
    ing = Ingest(tc, host.schema, n, 2 * n + 64, n_threads=8)
    for bi in range(nb):
        ing.push(host.slice(bi * rows, rows).to_batches()[0])
    chunk = 128 * rows
    states, done = [], 0
    partial_plan = None
    while done < n:
        landed = ing.wait_rows(done + chunk)
        assert landed >= min(n, done + chunk)
        view = ing.table(done, chunk)
        partial_py, full_py, final_src = T.q1_split_plan(view, 64)
        res = g.NativePlan(partial_py, tc).execute(0)
        states.append(g.plan.materialize(tc, res.to_device_table(tc.device), force=True))
        done += chunk
    merged = g.plan.concat_tables(tc, states)
    final_src.partitions[0] = merged
    out = g.NativePlan(full_py, tc).execute(0)
    from test_gpu_native_plan import arrow_rows
    assert [tuple(r) for r in arrow_rows(out.to_arrow())] == T.q1_oracle_rows(n)
    ing.close()
