"""Scan-side decode on the device (include/gpuq.h "scan-side decode"): the CsvExec / ParquetExec leaves
of the reference's TPC-H plans (benchmarks/src/bin/tpch.rs:801-862).  Checker: pyarrow's CSV and Parquet readers on the same
bytes (bit-exact, including Float64: only the exactly-rounded fast path is decoded, anything else is refused)."""
import decimal
import io
import os

import numpy as np
import pyarrow as pa
import pyarrow.csv as pacsv
import pyarrow.parquet as pq
import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd import scan

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

D152 = {"Decimal128": [15, 2]}
# benchmarks/src/bin/tpch.rs get_schema("lineitem") + the trailing `__placeholder` of get_tbl_tpch_table_schema (:959-964)
LINEITEM = [("l_orderkey", "Int64", False), ("l_partkey", "Int64", False), ("l_suppkey", "Int64", False), ("l_linenumber", "Int32", False),
            ("l_quantity", D152, False), ("l_extendedprice", D152, False), ("l_discount", D152, False), ("l_tax", D152, False),
            ("l_returnflag", "Utf8", False), ("l_linestatus", "Utf8", False), ("l_shipdate", "Date32", False), ("l_commitdate", "Date32", False),
            ("l_receiptdate", "Date32", False), ("l_shipinstruct", "Utf8", False), ("l_shipmode", "Utf8", False), ("l_comment", "Utf8", False),
            ("__placeholder", "Utf8", False)]


def pa_type(t):
    if isinstance(t, dict):
        return pa.decimal128(*t["Decimal128"])
    return {"Int32": pa.int32(), "Int64": pa.int64(), "Date32": pa.date32(), "Float64": pa.float64(), "Boolean": pa.bool_(), "Utf8": pa.string()}[t]


def pyarrow_csv(data, schema, delimiter=",", header=False, include=None):
    names = [s[0] for s in schema]
    ro = pacsv.ReadOptions(column_names=None if header else names, autogenerate_column_names=False)
    co = pacsv.ConvertOptions(column_types={s[0]: pa_type(s[1]) for s in schema}, strings_can_be_null=True, null_values=[""], quoted_strings_can_be_null=False,
                              include_columns=include, true_values=["true", "True", "TRUE"], false_values=["false", "False", "FALSE"])
    t = pacsv.read_csv(io.BytesIO(data), read_options=ro, parse_options=pacsv.ParseOptions(delimiter=delimiter, quote_char=False), convert_options=co)
    # the reference's reader (arrow-csv): an empty field is NULL only in a nullable column; in a required Utf8 column it is ""
    cols = []
    for f in t.schema:
        nullable = dict((s[0], s[2]) for s in schema)[f.name]
        c = t.column(f.name).combine_chunks()
        if not nullable and pa.types.is_string(f.type):
            c = c.fill_null("")
        cols.append(c)
    return pa.table(cols, names=t.schema.names)


def same(got, want):
    assert got.num_rows == want.num_rows
    assert got.schema.names == want.schema.names
    for name in want.schema.names:
        a, b = got.column(name).combine_chunks(), want.column(name).combine_chunks()
        assert a.type == b.type, (name, a.type, b.type)
        assert a.null_count == b.null_count, name
        if pa.types.is_floating(a.type):      # bit patterns, not ==
            bits = np.uint32 if a.type == pa.float32() else np.uint64
            assert np.array_equal(np.asarray(a.fill_null(0.0)).view(bits), np.asarray(b.fill_null(0.0)).view(bits)), name
            assert a.is_null().equals(b.is_null()), name
        else:
            assert a.equals(b), (name, a.to_pylist()[:5], b.to_pylist()[:5])


# ------------------------------------------------------------------ delimited text
@pytest.mark.parametrize("part", ["lineitem.partition0.tbl", "lineitem.partition1.tbl"])
def test_tpch_tbl_lineitem_as_the_benchmark_registers_it(tc, part):
    """The reference's own .tbl fixture (ballista/scheduler/testdata/lineitem), '|' delimited, no header, trailing delimiter."""
    data = open(os.path.join(GOLD, "tpch10", part), "rb").read()
    got = scan.read_csv(tc, data, LINEITEM, delimiter="|").to_arrow(tc.ctx)
    same(got, pyarrow_csv(data, LINEITEM, delimiter="|"))
    assert got.num_rows == 10 and got.column("__placeholder").to_pylist() == [""] * 10


def test_tbl_from_a_path_with_projection_by_name(tc):
    path = os.path.join(GOLD, "tpch10", "orders.tbl")
    orders = [("o_orderkey", "Int64", False), ("o_custkey", "Int64", False), ("o_orderstatus", "Utf8", False), ("o_totalprice", D152, False),
              ("o_orderdate", "Date32", False), ("o_orderpriority", "Utf8", False), ("o_clerk", "Utf8", False), ("o_shippriority", "Int32", False),
              ("o_comment", "Utf8", False), ("__placeholder", "Utf8", False)]
    want = pyarrow_csv(open(path, "rb").read(), orders, delimiter="|", include=["o_orderdate", "o_orderkey", "o_totalprice", "o_comment"])
    got = scan.read_csv(tc, path, orders, projection=["o_orderdate", "o_orderkey", "o_totalprice", "o_comment"], delimiter="|").to_arrow(tc.ctx)
    same(got, want)


AGG100 = [("c1", "Utf8", False), ("c2", "Int32", False), ("c3", "Int32", False), ("c4", "Int32", False), ("c5", "Int64", False), ("c6", "Int64", False),
          ("c7", "Int32", False), ("c8", "Int32", False), ("c9", "Int64", False), ("c10", "Utf8", False), ("c11", "Float64", False), ("c12", "Float64", False),
          ("c13", "Utf8", False)]


def test_reference_example_csv_with_header(tc):
    """examples/testdata/aggregate_test_100.csv (header line; c10 exceeds Int64 and is read as text; c12 carries 16-17 significant
    digits and is outside the exact fast path: projecting it is refused, not rounded differently)."""
    data = open(os.path.join(GOLD, "aggregate_test_100.csv"), "rb").read()
    keep = [s[0] for s in AGG100 if s[0] != "c12"]
    got = scan.read_csv(tc, data, AGG100, projection=keep, has_header=True).to_arrow(tc.ctx)
    same(got, pyarrow_csv(data, AGG100, header=True, include=keep))
    assert got.num_rows == 100
    with pytest.raises(g.GpuqError) as e:
        scan.read_csv(tc, data, AGG100, has_header=True)
    assert e.value.status == 3 and "Float64" in str(e.value)


def synthetic_csv(n, seed, crlf=False, terminated=True):
    r = np.random.default_rng(seed)
    words = np.array(["", "a", "BUILDING", "a-string-longer-than-fifteen-bytes", "x" * 40, "café ☃"])
    k = r.integers(-2**62, 2**62, n)
    i = r.integers(-2**31, 2**31 - 1, n)
    d = r.integers(-5000, 30000, n).astype("datetime64[D]")
    dec = r.integers(-10**12, 10**12, n)
    frac = r.integers(0, 3, n)             # 0, 1 or 2 fraction digits written
    f_m = r.integers(-10**14, 10**14, n)   # <= 15 digits
    f_e = r.integers(-22, 23, n)
    b = r.integers(0, 2, n).astype(bool)
    s = words[r.integers(0, len(words), n)]
    nk_null = r.random(n) < 0.1
    ns = words[r.integers(0, len(words), n)]
    lines = []
    for j in range(n):
        v = int(dec[j]); fd = int(frac[j])
        sign = "-" if v < 0 else ""
        a = abs(v)
        if fd == 0:
            a -= a % 100; ds = "%s%d" % (sign, a // 100)
        elif fd == 1:
            a -= a % 10; ds = "%s%d.%d" % (sign, a // 100, (a % 100) // 10)
        else:
            ds = "%s%d.%02d" % (sign, a // 100, a % 100)
        fs = "%de%d" % (f_m[j], f_e[j]) if j % 3 else ("%.6f" % (f_m[j] / 1e6))
        lines.append(",".join([str(k[j]), str(i[j]), str(d[j]), ds, fs, "true" if b[j] else "false", s[j],
                               "" if nk_null[j] else str(k[j] // 3), ns[j]]))
    eol = "\r\n" if crlf else "\n"
    text = eol.join(lines) + (eol if terminated and n else "")
    schema = [("k", "Int64", False), ("i", "Int32", False), ("d", "Date32", False), ("dec", D152, False), ("f", "Float64", False), ("b", "Boolean", False),
              ("s", "Utf8", False), ("nk", "Int64", True), ("ns", "Utf8", True)]
    return text.encode(), schema


@pytest.mark.parametrize("n,crlf,terminated", [(0, False, True), (1, False, False), (63, False, True), (64, True, True), (65, False, False), (50_000, True, False),
                                                 (200_003, False, True)])
def test_every_column_kind_against_pyarrow(tc, n, crlf, terminated):
    """Int64, Int32, Date32 before and after 1970, Decimal128 with 0-2 written fraction digits, Float64 in fixed and
    exponent notation, Boolean, Utf8 (empty, > 15 bytes, multi-byte), nullable Int64 and nullable Utf8; LF and CRLF; the last
    line with and without its newline; row counts around the 64-lane validity words."""
    data, schema = synthetic_csv(n, 11 + n, crlf, terminated)
    got = scan.read_csv(tc, data, schema).to_arrow(tc.ctx)
    if n == 0:
        assert got.num_rows == 0 and got.schema.names == [s[0] for s in schema]
        return
    same(got, pyarrow_csv(data, schema))


@pytest.mark.parametrize("text,code,what", [
    (b'1,"a,b"\n', 3, "quoted"), (b"1\n", 1, "number of fields"), (b"1,2,3\n", 1, "number of fields"), (b"1x,a\n", 1, "parse"), (b",a\n", 1, "non-nullable"),
    (b"9223372036854775808,a\n", 1, "parse"), (b"-9223372036854775809,a\n", 1, "parse"), (b"123456789012345678901234567890,a\n", 1, "parse"),
])
def test_malformed_text_is_refused(tc, text, code, what):
    with pytest.raises(g.GpuqError) as e:
        scan.read_csv(tc, text, [("a", "Int64", False), ("b", "Utf8", False)])
    assert e.value.status == code and what in str(e.value)


def test_numeric_limits_of_the_text_parser(tc):
    """The extremes of every integer type parse exactly; one past them, and decimals beyond the column's precision, are refused."""
    schema = [("a", "Int64", False), ("b", "Int32", False), ("c", {"Decimal128": [5, 2]}, False)]
    ok = b"9223372036854775807,2147483647,999.99\n-9223372036854775808,-2147483648,-999.99\n"
    got = scan.read_csv(tc, ok, schema).to_arrow(tc.ctx)
    same(got, pyarrow_csv(ok, schema))
    for bad in (b"1,2147483648,1.00\n", b"1,-2147483649,1.00\n", b"1,2,1000.00\n", b"1,2,1.001\n"):
        with pytest.raises(g.GpuqError, match="parse"):
            scan.read_csv(tc, bad, schema)


# ------------------------------------------------------------------ Parquet
DEVICE_COLS = ["id", "bool_col", "tinyint_col", "smallint_col", "int_col", "bigint_col", "double_col", "date_string_col", "string_col"]


def test_reference_alltypes_plain_parquet(tc):
    """ballista/client/testdata/alltypes_plain.parquet (Impala-written, v1 pages, PLAIN_DICTIONARY, optional columns): the file the
    reference's client KATs read (context.rs:762-967).  Round 3: every column is decoded on the device -- float_col (FLOAT ->
    Float32) and timestamp_col (INT96 -> Timestamp(Nanosecond), as arrow's reader maps Impala's timestamps) included."""
    path = os.path.join(GOLD, "alltypes_plain.parquet")
    fields, rows = scan.parquet_schema(tc.ctx.L, path)
    assert rows == 8 and [f[0] for f in fields if f[1] is None] == []
    got = scan.read_parquet(tc, path, DEVICE_COLS).to_arrow(tc.ctx)
    want = pq.read_table(path, columns=DEVICE_COLS)
    want = want.cast(pa.schema([pa.field(f.name, pa.string() if pa.types.is_binary(f.type) else f.type) for f in want.schema]))
    same(got, want)
    assert got.column("id").to_pylist() == [4, 5, 6, 7, 2, 3, 0, 1]
    allc = scan.read_parquet(tc, path).to_arrow(tc.ctx)
    want = pq.read_table(path)
    assert allc.schema.field("float_col").type == pa.float32() and allc.schema.field("timestamp_col").type == pa.timestamp("ns")
    assert allc.column("float_col").to_pylist() == want.column("float_col").to_pylist() == [0.0, 1.100000023841858] * 4
    assert allc.column("timestamp_col").cast(pa.int64()).to_pylist() == want.column("timestamp_col").cast(pa.int64()).to_pylist()
    assert allc.column("timestamp_col").cast(pa.int64()).to_pylist()[0] == 1235865600000000000      # 2009-03-01T00:00:00


def parquet_table(n, seed):
    r = np.random.default_rng(seed)
    words = np.array(["", "a", "BUILDING", "a-string-longer-than-fifteen-bytes", "x" * 40, "café ☃"])
    dec = [decimal.Decimal(int(v)).scaleb(-2) for v in r.integers(-10**12, 10**12, min(n, 4096))] * (n // 4096 + 1)
    t = pa.table({
        "k": pa.array(r.integers(-2**62, 2**62, n), type=pa.int64()),
        "lowcard": pa.array(r.integers(0, 7, n), type=pa.int64()),
        "i": pa.array(r.integers(-2**31, 2**31 - 1, n).astype(np.int32)),
        "d": pa.array(r.integers(-5000, 30000, n).astype(np.int32)).cast(pa.date32()),
        "f": pa.array(r.normal(0, 1e6, n)),
        "dec": pa.array(dec[:n], type=pa.decimal128(15, 2)),
        "dec38": pa.array([x * 10**20 for x in dec[:n]], type=pa.decimal128(38, 2)),
        "b": pa.array(r.integers(0, 2, n).astype(bool)),
        "s": pa.array(words[r.integers(0, len(words), n)]),
        "u": pa.array(["row-%d" % j for j in range(n)]),
        "nk": pa.array(r.integers(0, 1000, n), type=pa.int64(), mask=r.random(n) < 0.1),
        "nb": pa.array(r.integers(0, 2, n).astype(bool), mask=r.random(n) < 0.3),
        "ns": pa.array(words[r.integers(0, len(words), n)], mask=r.random(n) < 0.2),
        "allnull": pa.array([None] * n, type=pa.int32()),
    })
    req = ("k", "lowcard", "i", "d", "f", "dec", "dec38", "b", "s", "u")
    return t.cast(pa.schema([pa.field(f.name, f.type, nullable=f.name not in req) for f in t.schema]))


def write(t, **kw):
    buf = io.BytesIO()
    kw.setdefault("compression", "NONE")
    pq.write_table(t, buf, **kw)
    return buf.getvalue()


@pytest.mark.parametrize("n", [0, 1, 63, 1000, 70_001])      # 0 = the schema of a 10-row table, no rows
@pytest.mark.parametrize("opts", [
    dict(use_dictionary=True, data_page_version="1.0"),
    dict(use_dictionary=False, data_page_version="1.0"),
    dict(use_dictionary=True, data_page_version="2.0", data_page_size=4096),
    dict(use_dictionary=False, data_page_version="2.0", data_page_size=4096, row_group_size=9000),
    dict(use_dictionary=["lowcard", "s", "ns"], data_page_version="1.0", data_page_size=2048, row_group_size=20_000, dictionary_pagesize_limit=64),
    dict(use_dictionary=True, data_page_version="1.0", compression="SNAPPY", data_page_size=16384),
    dict(use_dictionary=["lowcard", "s"], data_page_version="2.0", compression="SNAPPY", data_page_size=4096, row_group_size=25_000),
    dict(use_dictionary=True, data_page_version="1.0", compression="LZ4", data_page_size=16384),
    dict(use_dictionary=["lowcard", "s"], data_page_version="2.0", compression="LZ4", data_page_size=4096, row_group_size=25_000),
    dict(use_dictionary=True, data_page_version="1.0", compression="ZSTD", data_page_size=16384),
    dict(use_dictionary=["lowcard", "s"], data_page_version="2.0", compression="ZSTD", data_page_size=4096, row_group_size=25_000),
    dict(use_dictionary=False, data_page_version="1.0", compression="ZSTD", compression_level=19),
], ids=["dict-v1", "plain-v1", "dict-v2-smallpages", "plain-v2-rowgroups", "mixed-dict-fallback", "snappy-v1", "snappy-v2", "lz4raw-v1", "lz4raw-v2",
        "zstd-v1", "zstd-v2", "zstd-19-bigpages"])
def test_pyarrow_written_files(tc, n, opts):
    """Required and optional columns of every decoded type; dictionary and PLAIN pages, v1 and v2 headers, many small pages, several
    row groups, a dictionary that overflows and falls back to PLAIN mid-chunk; decimals as FIXED_LEN_BYTE_ARRAY (15,2) and (38,2)."""
    t = parquet_table(n, 5 + n) if n else parquet_table(10, 5).slice(0, 0)
    data = write(t, **opts)
    got = scan.read_parquet(tc, data).to_arrow(tc.ctx)
    same(got, pq.read_table(io.BytesIO(data)))


def test_reference_single_nan_parquet_snappy(tc):
    """ballista/client/testdata/single_nan.parquet: one optional DOUBLE column, SNAPPY-compressed dictionary + data page."""
    path = os.path.join(GOLD, "single_nan.parquet")
    same(scan.read_parquet(tc, path).to_arrow(tc.ctx), pq.read_table(path))


def test_snappy_streams_with_long_literals_and_overlapping_copies(tc):
    """Snappy's element kinds: highly repetitive columns (copies whose offset is shorter than their length), incompressible ones
    (literals with 2- and 3-byte length fields), and pages far beyond 64 KiB (2- and 4-byte offset copies)."""
    n = 300_000
    r = np.random.default_rng(4)
    t = pa.table({"same": pa.array(np.full(n, 123456789, np.int64)), "ramp": pa.array(np.arange(n, dtype=np.int64) % 7),
                  "noise": pa.array(r.integers(-2**62, 2**62, n)), "txt": pa.array(["abcabcabc-%d" % (j % 13) for j in range(n)]),
                  "opt": pa.array(r.integers(0, 3, n), type=pa.int32(), mask=r.random(n) < 0.5)})
    for opts in (dict(use_dictionary=False, data_page_size=1 << 20), dict(use_dictionary=False, data_page_size=1 << 14, data_page_version="2.0")):
        for codec in ("SNAPPY", "LZ4", "ZSTD"):      # (ZSTD: pages of several 128 KiB blocks, RLE / raw / Huffman literals, repeat offsets; LZ4 = LZ4_RAW: one raw block per page, long matches and 255-runs of length bytes in the repetitive columns)
            data = write(t, compression=codec, **opts)
            same(scan.read_parquet(tc, data).to_arrow(tc.ctx), pq.read_table(io.BytesIO(data)))


def test_decimals_stored_as_integers_and_projection_order(tc):
    t = parquet_table(5000, 2).select(["dec", "k", "ns"])
    small = t.set_column(0, "dec", t.column("dec").cast(pa.decimal128(9, 2), safe=False) if False else pa.array([decimal.Decimal(v).scaleb(-2) for v in range(-2500, 2500)], type=pa.decimal128(9, 2)))
    big = small.append_column("dec18", pa.array([decimal.Decimal(v * 10**9).scaleb(-2) for v in range(-2500, 2500)], type=pa.decimal128(18, 2)))
    try:
        data = write(big, store_decimal_as_integer=True)
    except TypeError:
        pytest.skip("this pyarrow cannot store decimals as integers")
    got = scan.read_parquet(tc, data, ["dec18", "ns", "dec"]).to_arrow(tc.ctx)
    same(got, pq.read_table(io.BytesIO(data), columns=["dec18", "ns", "dec"]))


def test_zstd_pages_as_the_reference_convert_writes_them(tc):
    """`tpch convert` writes ZSTD Parquet unless told otherwise (/root/reference/benchmarks/src/bin/tpch.rs:225-226, 777): lineitem-shaped
    columns at several levels, pages of 1 MiB (eight zstd blocks each), against pyarrow's reader (libzstd)."""
    n = 400_000
    r = np.random.default_rng(11)
    modes = np.array(["DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"])
    t = pa.table({"l_orderkey": pa.array(np.sort(r.integers(1, 6_000_000, n))), "l_quantity": pa.array(r.integers(1, 51, n).astype(np.float64)),
                  "l_extendedprice": pa.array(np.round(r.uniform(900, 105000, n), 2)), "l_shipdate": pa.array(r.integers(8036, 10561, n).astype(np.int32), type=pa.date32()),
                  "l_returnflag": pa.array(np.array(["A", "N", "R"])[r.integers(0, 3, n)]), "l_shipinstruct": pa.array(modes[r.integers(0, 4, n)]),
                  "l_comment": pa.array(["%s the %s foxes %d" % ("carefully" if j % 3 else "quickly", "final" if j % 5 else "bold", j % 977) for j in range(n)])})
    for level in (1, 3, 12):
        for dic in (True, False):
            data = write(t, compression="ZSTD", compression_level=level, use_dictionary=dic)
            same(scan.read_parquet(tc, data).to_arrow(tc.ctx), pq.read_table(io.BytesIO(data)))


def test_compressed_chunks_and_garbage_are_refused(tc):
    t = parquet_table(100, 1)
    buf = io.BytesIO()
    pq.write_table(t, buf, compression="GZIP")
    with pytest.raises(g.GpuqError) as e:
        scan.read_parquet(tc, buf.getvalue())
    assert e.value.status == 3 and "compressed" in str(e.value)
    # a corrupted Snappy stream is flagged by the unpack kernel, not decoded into garbage
    snap = bytearray(write(t.select(["k"]), compression="SNAPPY", use_dictionary=False))
    for k in range(40, 120):
        snap[k] = 0xFF
    with pytest.raises(g.GpuqError):
        scan.read_parquet(tc, bytes(snap))
    with pytest.raises(g.GpuqError):
        scan.read_parquet(tc, b"PAR1" + b"\x00" * 64 + b"PAR1")
    with pytest.raises(g.GpuqError) as e:
        scan.read_parquet(tc, write(t), ["no_such_column"])
    assert "not in the file" in str(e.value)


def test_decoded_leaves_feed_q1(tc):
    """TPC-H q1 over lineitem columns decoded on the device from Parquet bytes and from '|' text == the oracle's q1 over the same
    generated rows (the plan's MemoryExec leaf holds what CsvExec / ParquetExec would have produced)."""
    import tpch_util as T
    n = 30_000
    li = T.lineitem_host_to_arrow(T.gen_lineitem_host(n), n)
    li = li.cast(pa.schema([pa.field(f.name, f.type, nullable=False) for f in li.schema]))
    want = T.q1_oracle_rows(n)
    from_parquet = scan.read_parquet(tc, write(li, use_dictionary=True, data_page_size=8192))
    assert T.q1_result_to_rows(tc, T.run_q1(tc, from_parquet)) == want
    cols = [li.column(i).to_pylist() for i in range(li.num_columns)]
    text = "".join("|".join(str(c[j]) for c in cols) + "|\n" for j in range(n)).encode()
    schema = [(f.name, {"int64": "Int64", "string": "Utf8", "date32[day]": "Date32"}.get(str(f.type), D152), False) for f in li.schema] + [("__placeholder", "Utf8", False)]
    from_text = scan.read_csv(tc, text, schema, projection=list(range(li.num_columns)), delimiter="|")
    assert T.q1_result_to_rows(tc, T.run_q1(tc, from_text)) == want


@pytest.mark.parametrize("compression", ["NONE", "SNAPPY", "LZ4", "ZSTD"])
def test_corrupted_pages_never_read_out_of_bounds(tc, compression):
    """Page bytes are untrusted: random corruptions of the data region of a valid file (levels, run headers, dictionary indices, length
    prefixes, Snappy tags) must end in an error or in (different) in-bounds values -- every length the kernels follow is checked
    against the page it came from.  A clean decode of the untouched file afterwards shows the device is still healthy."""
    t = parquet_table(20_000, 77)
    good = write(t, compression=compression, use_dictionary=["lowcard", "s", "ns"], data_page_size=8192, row_group_size=7000)
    flen = int.from_bytes(good[-8:-4], "little")
    data_end = len(good) - 8 - flen
    r = np.random.default_rng(9)
    outcomes = {"ok": 0, "err": 0}
    for it in range(48):
        bad = bytearray(good)
        for _ in range(1 + it % 6):
            at = int(r.integers(4, data_end))
            bad[at] = int(r.integers(0, 256)) if it % 2 else 0xFF
        try:
            scan.read_parquet(tc, bytes(bad)); outcomes["ok"] += 1
        except g.GpuqError:
            outcomes["err"] += 1
    assert outcomes["err"] > 0
    same(scan.read_parquet(tc, good).to_arrow(tc.ctx), pq.read_table(io.BytesIO(good)))


def test_random_text_is_parsed_or_refused(tc):
    r = np.random.default_rng(3)
    schema = [("a", "Int64", True), ("b", {"Decimal128": [15, 2]}, True), ("c", "Date32", True), ("d", "Float64", True), ("e", "Utf8", True)]
    alphabet = np.frombuffer(b"0123456789-+.,eE\n\r x", dtype=np.uint8)
    for it in range(30):
        text = alphabet[r.integers(0, len(alphabet), 5000)].tobytes()
        try:
            scan.read_csv(tc, text, schema)
        except g.GpuqError:
            pass
    data, sch = synthetic_csv(1000, 5)
    same(scan.read_csv(tc, data, sch).to_arrow(tc.ctx), pyarrow_csv(data, sch))


def test_row_group_selection_and_chunked_text(tc):
    """Predicate push-down stays on the host (the reference prunes row groups from footer statistics): the engine decodes the row
    groups it is given, in the order given.  Text beyond one call's size is decoded in chunks cut at line boundaries; the chunks are
    the partitions of the scan leaf."""
    t = parquet_table(10_000, 21)
    data = write(t, row_group_size=3000, use_dictionary=True)
    assert scan.parquet_row_groups(tc.ctx.L, data) == [3000, 3000, 3000, 1000]
    got = scan.read_parquet(tc, data, ["k", "ns", "b"], row_groups=[3, 1]).to_arrow(tc.ctx)
    want = pa.concat_tables([pq.ParquetFile(io.BytesIO(data)).read_row_group(g_, columns=["k", "ns", "b"]) for g_ in (3, 1)])
    same(got, want)
    assert scan.read_parquet(tc, data, ["k"], row_groups=[]).num_rows == 0
    with pytest.raises(g.GpuqError, match="not in the file"):
        scan.read_parquet(tc, data, row_groups=[4])
    text, schema = synthetic_csv(20_000, 4)
    parts = scan.read_csv_chunked(tc, text, schema, chunk_bytes=200_000)
    assert len(parts) > 5 and sum(p.num_rows for p in parts) == 20_000
    same(pa.concat_tables([p.to_arrow(tc.ctx) for p in parts]), pyarrow_csv(text, schema))
    # the chunks feed a plan as the partitions of its leaf
    src = g.MemoryExec(parts)
    s_ = src.schema()
    from arrow_ballista_amd.expr import col as C_, lit as L_
    plan = g.AggregateExec("Single", [], [{"fn": "COUNT", "expr": L_(1), "name": "c"}, {"fn": "SUM", "expr": C_("i", s_), "name": "si"}], g.CoalescePartitionsExec(src))
    row = g.NativePlan(plan, tc).execute(0).to_arrow().to_pylist()[0]
    ref = pyarrow_csv(text, schema)
    assert row["c"] == 20_000 and row["si"] == sum(ref.column("i").to_pylist())
