"""Shuffle codec on the device: Arrow IPC stream messages with LZ4_FRAME buffers.

The independent implementation on the other side of every check is Arrow C++ (pyarrow): files written by the device
encoder must read back identically through `pyarrow.ipc.open_stream` (liblz4 validates every frame header and block), and
files written by pyarrow -- frames with LINKED blocks -- must decode identically on the device.  Byte-exact both ways."""
import io

import numpy as np
import pyarrow as pa
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd import shuffle as S

pytestmark = pytest.mark.gpu


def table_for(seed, n, nulls):
    r = np.random.default_rng(seed)

    def mask():
        return (r.random(n) < nulls) if nulls > 0 else None
    import decimal
    words = np.array(["", "a", "BUILDING", "MACHINERY", "the quick brown fox jumps over the lazy dog " * 3, "Ünïcödé ✓", "x" * 300])
    cols = {
        "i64": pa.array(r.integers(0, 1000, n), type=pa.int64(), mask=mask()),                        # compressible
        "rnd": pa.array(r.integers(-2**62, 2**62, n), type=pa.int64(), mask=mask()),                  # incompressible: stored blocks / raw buffers
        "i32": pa.array((np.arange(n) // 7).astype(np.int32), type=pa.int32(), mask=mask()),          # long runs
        "d": pa.array(r.integers(8000, 10500, n).astype(np.int32), type=pa.int32(), mask=mask()).cast(pa.date32()),
        "dec": pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in r.integers(-10**9, 10**9, n)], type=pa.decimal128(15, 2), mask=mask()),
        "f": pa.array(np.round(r.normal(0, 1e3, n), 1), type=pa.float64(), mask=mask()),
        "s": pa.array(words[r.integers(0, len(words), n)], type=pa.string(), mask=mask()),
        "b": pa.array(r.integers(0, 2, n).astype(bool), type=pa.bool_(), mask=mask()),
        "z": pa.array(np.zeros(n, dtype=np.int64), type=pa.int64()),                                  # one long match per block
    }
    return pa.table(cols)


def same(a, b):
    assert a.schema.names == b.schema.names
    assert a.num_rows == b.num_rows
    for name in a.schema.names:
        x, y = a.column(name).combine_chunks(), b.column(name).combine_chunks()
        assert x.type == y.type, (name, x.type, y.type)
        assert x.equals(y), name


@pytest.mark.parametrize("codec", [0, -1])
@pytest.mark.parametrize("nulls", [0.0, 0.15])
@pytest.mark.parametrize("n,batch", [(0, None), (1, None), (13, None), (5000, 1024), (200_003, 65536), (200_003, None)])
def test_device_written_stream_reads_back_through_arrow_cpp(tc, n, batch, nulls, codec):
    t = table_for(n + 1, n, nulls)
    dt = g.DeviceTable.from_arrow(t, tc.device)
    buf = io.BytesIO()
    nb, rows, nbytes = S.write_ipc_stream(tc, buf, dt, batch_size=batch, codec=codec)
    raw = buf.getvalue()
    assert rows == n and nbytes == len(raw)
    got = pa.ipc.open_stream(raw).read_all()
    same(got, t)
    if n:
        assert nb == (1 if not batch else -(-n // batch))
    if codec == 0 and n >= 5000:
        assert len(raw) < t.nbytes            # the zero / run / small-range columns shrink
    # and back through the device decoder
    back, schema = S.read_ipc_stream(tc, raw)
    same(back.to_arrow(tc.ctx), t)


@pytest.mark.parametrize("compression", ["lz4", None])
@pytest.mark.parametrize("nulls", [0.0, 0.15])
@pytest.mark.parametrize("n,batch", [(0, None), (1, None), (77, None), (5000, 1000), (300_001, 65536), (300_001, None)])
def test_arrow_cpp_written_stream_decodes_on_device(tc, n, batch, nulls, compression):
    t = table_for(n + 7, n, nulls)
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, t.schema, options=pa.ipc.IpcWriteOptions(compression=compression)) as w:
        for b in t.to_batches(max_chunksize=batch):
            w.write_batch(b)
    got, schema = S.read_ipc_stream(tc, sink.getvalue())
    assert schema.equals(t.schema)
    same(got.to_arrow(tc.ctx), t)


def test_independent_block_frames_and_block_index_walk(tc):
    """Frames as lz4_flex (arrow-rs) writes them: independent blocks.  The device encoder writes that form; here its output is
    fed back with a multi-block buffer so the host-side block index walk (one wave per block) is the path taken."""
    n = 1 << 18
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64) % 1000), "v": pa.array(np.arange(n, dtype=np.int64) * 3)})
    dt = g.DeviceTable.from_arrow(t, tc.device)
    buf = io.BytesIO()
    S.write_ipc_stream(tc, buf, dt, codec=0)
    raw = buf.getvalue()
    # header of the first frame: magic, FLG 0x60 (version 01, independent blocks, no checksums), BD 0x40 (64 KiB), HC 0x82
    at = raw.find(bytes([0x04, 0x22, 0x4D, 0x18]))
    assert at > 0 and raw[at + 4: at + 7] == bytes([0x60, 0x40, 0x82])
    import xxhash
    assert (xxhash.xxh32(raw[at + 4: at + 6], seed=0).intdigest() >> 8) & 0xFF == 0x82
    back, _ = S.read_ipc_stream(tc, raw)
    same(back.to_arrow(tc.ctx), t)
    same(pa.ipc.open_stream(raw).read_all(), t)


def test_frame_by_frame_fallback_keeps_raw_stored_bitmaps(tc):
    """A frame whose independent blocks are smaller than its declared maximum (legal: BD only bounds them) defeats the host-side
    block index, and the decoder falls back to walking frames one by one after clearing the OR-merged bitmaps.  Buffers stored raw
    (the -1 length prefix an incompressible validity bitmap gets) are copied, not decoded: the fallback has to issue those copies
    again, or the column reads back all-NULL."""
    import xxhash
    n = 1 << 18
    r = np.random.default_rng(5)
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64) % 1000),
                  "r": pa.array(r.integers(-2**62, 2**62, n), type=pa.int64(), mask=r.random(n) < 0.5),
                  "b": pa.array(r.integers(0, 2, n).astype(bool))})
    buf = io.BytesIO()
    S.write_ipc_stream(tc, buf, g.DeviceTable.from_arrow(t, tc.device), codec=0)
    raw = bytearray(buf.getvalue())
    at = raw.find(bytes([0x04, 0x22, 0x4D, 0x18]))
    assert at > 0 and raw[at + 4: at + 7] == bytes([0x60, 0x40, 0x82])      # 64 KiB independent blocks, 32 of them in k's data
    raw[at + 5] = 0x50                                                       # claim 256 KiB: every block is now "not full"
    raw[at + 6] = (xxhash.xxh32(bytes(raw[at + 4: at + 6]), seed=0).intdigest() >> 8) & 0xFF
    same(pa.ipc.open_stream(bytes(raw)).read_all(), t)                       # still a valid stream for Arrow C++
    back, _ = S.read_ipc_stream(tc, bytes(raw))
    got = back.to_arrow(tc.ctx)
    assert got.column("r").null_count == t.column("r").null_count
    same(got, t)


def test_corrupt_utf8_offsets_are_refused(tc):
    """Offsets of a peer's shuffle file are untrusted: an uncompressed stream whose Utf8 offsets were made to run backwards / past the
    data buffer fails with 'malformed', it does not hand out a column whose offsets point outside its bytes."""
    t = pa.table({"s": pa.array(["row-%05d" % i for i in range(4096)])})
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, t.schema) as w:
        w.write_table(t)
    raw = bytearray(sink.getvalue())
    back, _ = S.read_ipc_stream(tc, bytes(raw))
    same(back.to_arrow(tc.ctx), t)
    # the offsets buffer is the run 0, 9, 18, ...: find entry 100 (= 900) followed by 909 and corrupt it
    pat = np.array([900, 909, 918], dtype=np.int32).tobytes()
    at = bytes(raw).find(pat)
    assert at > 0
    for bad_value in (5_000_000, -7, 700):       # beyond the data, negative, not monotone
        bad = bytearray(raw)
        bad[at + 4: at + 8] = np.array([bad_value], dtype=np.int32).tobytes() if bad_value != 5_000_000 else bad[at + 4: at + 8]
        if bad_value == 5_000_000:               # the LAST offset (4096 * 9) pushed past the data buffer
            last = bytes(raw).find(np.array([4095 * 9, 4096 * 9], dtype=np.int32).tobytes())
            bad[last + 4: last + 8] = np.array([bad_value], dtype=np.int32).tobytes()
        with pytest.raises(g.GpuqError) as e:
            S.read_ipc_stream(tc, bytes(bad))
        assert "offsets" in str(e.value)


def test_malformed_streams_fail_loudly(tc):
    t = table_for(3, 4000, 0.0)
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, t.schema, options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
        w.write_table(t)
    raw = bytearray(sink.getvalue())
    with pytest.raises(g.GpuqError):
        S.read_ipc_stream(tc, bytes(raw[: len(raw) // 2]))
    # corrupt the token stream of a frame: the device decoder must flag it, not write out of bounds
    at = raw.find(bytes([0x04, 0x22, 0x4D, 0x18]))
    bad = bytearray(raw)
    for k in range(at + 11, at + 400):
        bad[k] = 0xFF
    with pytest.raises(g.GpuqError):
        S.read_ipc_stream(tc, bytes(bad))
    zs = io.BytesIO()
    with pa.ipc.new_stream(zs, t.schema, options=pa.ipc.IpcWriteOptions(compression="zstd")) as w:
        w.write_table(t)
    with pytest.raises(g.GpuqError) as e:
        S.read_ipc_stream(tc, zs.getvalue())
    assert e.value.status == 3


def test_two_stages_through_shuffle_files(tc, tmp_path, mirror_layer):
    """Map stage: ShuffleWriterExec hash-partitions its input into files (device LZ4).  Reduce stage: ShuffleReaderExec reads
    output partition q of every map task (device decode) and a FinalPartitioned-style aggregate runs on it -- the file-based
    shuffle of the reference (shuffle_writer.rs:234-456 -> shuffle_reader.rs:149-177), result identical to the one-stage oracle."""
    import os
    from arrow_ballista_amd.expr import Operator as Op, binary, col, lit
    from oracle import oracle_np as O
    import test_gpu_operators as TO
    parts = [TO.rand_table(900 + i, 6000, 0.1) for i in range(3)]
    src = g.MemoryExec(parts)
    s = src.schema()
    nred = 4
    writer = g.ShuffleWriterExec("jobS", 1, g.FilterExec(binary(col("k32", s), Op.GtEq, lit(-40, "Int32")), src), str(tmp_path), ([col("k64", s)], nred))
    stage = g.DefaultExecutionEngine().create_query_stage_exec("jobS", 1, writer, str(tmp_path))
    locs = []
    for p in range(3):                                    # one map task per input partition
        locs += stage.execute_query_stage([p], tc)
    assert all(os.path.getsize(o["path"]) == o["num_bytes"] for o in locs)
    reader = g.ShuffleReaderExec([[o for o in locs if o["partition_id"] == q] for q in range(nred)], s)
    rs = reader.schema()
    aggs = [{"fn": "SUM", "expr": col("dec", rs), "name": "sd"}, {"fn": "COUNT", "expr": lit(1), "name": "c"}, {"fn": "COUNT", "expr": col("s", rs), "name": "cs"},
            {"fn": "MIN", "expr": col("d", rs), "name": "md"}]
    plan = g.AggregateExec("Single", [(col("k64", rs), "k64")], aggs, reader)
    got = []
    for q in range(nred):
        got += TO.dev_rows(tc, plan.execute(q, tc))
    ot = O.Table.from_arrow(pa.concat_tables(parts))
    keep = O.filter_rows(ot, binary(col("k32", s), Op.GtEq, lit(-40, "Int32")))
    exp = TO.ora_rows(O.aggregate(ot.take(keep), [(col("k64", s), "k64")], aggs, "Single"))
    assert TO.norm(got) == TO.norm(exp)
    assert reader.metrics.output_rows == len(keep)
    with pytest.raises(g.GpuqError, match="FetchFailed"):
        g.ShuffleReaderExec([[{"path": str(tmp_path / "missing.arrow")}]], s).execute(0, tc)
    assert g.ShuffleReaderExec([[]], s).execute(0, tc).num_rows == 0


def _pattern_cases():
    """Byte strings that walk the LZ4 block format's boundaries: literal / match length fields at 14, 15, 16, 269, 270, 271
    (token nibble, one and two extension bytes), periods shorter than the match (overlapping copies), the end-of-block rules
    (no match in the last 12 bytes, 5 literal bytes at the end), blocks of exactly 64 KiB and one byte either side, long
    single matches spanning blocks, incompressible data (stored blocks / raw buffers)."""
    r = np.random.default_rng(5)
    cases = []
    for L in list(range(0, 40)) + [63, 64, 65, 254, 255, 256, 269, 270, 271, 272, 510, 511, 512, 513, 4095, 4096, 4097]:
        cases.append(bytes([65]) * L)                                              # one long run
        cases.append(r.integers(0, 256, L, dtype=np.uint8).tobytes())              # incompressible
        cases.append((b"abcdefghijklmnopqrstuvwxyz0123456789" * (L // 36 + 1))[:L])   # period 36
    for lit in (0, 1, 14, 15, 16, 17, 269, 270, 271, 525):
        for ml in (4, 5, 18, 19, 20, 273, 274, 275, 530):
            head = r.integers(0, 256, lit, dtype=np.uint8).tobytes()
            word = r.integers(0, 256, max(ml, 8), dtype=np.uint8).tobytes()
            cases.append(word + head + word[:ml] + r.integers(0, 256, 13, dtype=np.uint8).tobytes())
    for period in (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 63, 64, 65):
        unit = r.integers(0, 256, period, dtype=np.uint8).tobytes()
        for total in (period + 4, 100, 1000):
            cases.append((unit * (total // period + 1))[:total] + b"tail-bytes-xyz")
    for L in (65535, 65536, 65537, 131072, 131073, 200_000):
        cases.append(bytes(L))                                                     # zeros across block boundaries
        cases.append((r.integers(0, 4, L, dtype=np.uint8) + 48).tobytes())        # low entropy
        cases.append(r.integers(0, 256, L, dtype=np.uint8).tobytes())
    return cases


def test_lz4_format_boundaries_both_directions(tc):
    """Each case's bytes (padded to a multiple of 4) ride as the data buffer of an Int32 column, one batch per case: the codec
    sees exactly those bytes.  Arrow C++ is the writer for one direction and the reader for the other."""
    cases = _pattern_cases()
    batches, expect = [], []
    for c in cases:
        pad = (-len(c)) % 4
        arr = np.frombuffer(c + bytes(pad), dtype=np.int32)
        expect.append(arr)
        batches.append(pa.record_batch([pa.array(arr, type=pa.int32())], names=["x"]))
    schema = batches[0].schema
    # 1. Arrow C++ writes (linked frames), the device reads the whole stream in one launch
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, schema, options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
        for b in batches:
            w.write_batch(b)
    got, _ = S.read_ipc_stream(tc, sink.getvalue())
    want = np.concatenate(expect)
    assert np.array_equal(got.to_arrow(tc.ctx).column("x").to_numpy(), want)
    # 2. the device writes every case as its own batch, Arrow C++ and the device read them back
    out = io.BytesIO()
    out.write(schema.serialize().to_pybytes())
    buf = None
    for b in batches:
        dt = g.DeviceTable.from_arrow(pa.Table.from_batches([b]), tc.device)
        msg, buf = S.encode_batch(tc, dt, 0, buf)
        out.write(bytes(msg))
    out.write(S.EOS)
    raw = out.getvalue()
    back = pa.ipc.open_stream(raw)
    n = 0
    for b, e in zip(back, expect):
        assert np.array_equal(b.column(0).to_numpy(), e), n
        n += 1
    assert n == len(cases)
    got, _ = S.read_ipc_stream(tc, raw)
    assert np.array_equal(got.to_arrow(tc.ctx).column("x").to_numpy(), want)
    # compressible cases must actually shrink: 200,000 zero bytes fit in a few hundred
    dt = g.DeviceTable.from_arrow(pa.table({"x": pa.array(np.zeros(50_000, dtype=np.int32))}), tc.device)
    assert S.encoded_size(tc, dt) < 2000


def test_reference_shuffle_file_decodes_on_the_device(tc):
    """The shuffle file the reference's own reader test loads (ballista/core/tests/data.arrow -> tests/golden/shuffle_data.arrow,
    async_reader/mod.rs:331-357; written by the reference's ShuffleWriterExec with lz4_flex frames): decoded by
    gpuq_ipc_decode_stream on the device, byte for byte what Arrow C++ reads from the same file -- 561 Utf8 rows of 190-330
    bytes each (longer than any PACKED15 key: payload strings travel in Arrow layout)."""
    import os
    import pyarrow as pa
    from arrow_ballista_amd import shuffle as S
    path = os.path.join(os.path.dirname(__file__), "golden", "shuffle_data.arrow")
    raw = open(path, "rb").read()
    ref = pa.ipc.open_stream(raw).read_all()
    for source in (path, raw):
        t, schema = S.read_ipc_stream(tc, source)
        got = t.to_arrow(tc.ctx)
        assert schema.equals(ref.schema) and got.num_rows == 561
        assert got.column(0).to_pylist() == ref.column(0).to_pylist()
    # and back: the device writer's stream of the same rows is read by Arrow C++
    import io
    sink = io.BytesIO()
    S.write_ipc_stream(tc, sink, t)
    assert pa.ipc.open_stream(sink.getvalue()).read_all().column(0).to_pylist() == ref.column(0).to_pylist()
