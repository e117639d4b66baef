"""CPU: the C-ABI library loads and exports every symbol include/gpuq.h declares; the host logic
(descriptor parsing, expression typing = DataFusion's decimal rules, register allocation, plan schema
propagation) works without a device; the product path refuses to run without a GPU (no fallback)."""
import ctypes as C
import os
import re

import pytest

import arrow_ballista_amd as g
from arrow_ballista_amd.expr import Operator as Op
from arrow_ballista_amd.expr import binary, case, cast, col, in_list, lit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D152 = {"Decimal128": [15, 2]}


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gpuq.h")).read()
    declared = set(re.findall(r"\b(gpuq_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gpuq_status"}
    L = C.CDLL(g.lib_path())
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert len(declared) >= 30
    assert g.lib().gpuq_abi_version() == 2
    assert set(g.lib()._gpuq_symbols) <= declared | {"gpuq_abi_version"}


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(g.GpuqError) as e:
        g.Context(0)
    assert "no CPU fallback" in str(e.value)


FIELDS = [{"name": "q", "type": D152, "nullable": False}, {"name": "e", "type": D152, "nullable": False}, {"name": "d", "type": D152, "nullable": True},
          {"name": "i", "type": "Int64", "nullable": False}, {"name": "j", "type": "Int32", "nullable": False}, {"name": "f", "type": "Float64", "nullable": False},
          {"name": "s", "type": "Utf8", "nullable": False}, {"name": "dt", "type": "Date32", "nullable": False}]


def out_types(exprs):
    d = g.compile_check({"op": "project", "input": {"fields": FIELDS}, "exprs": [{"expr": e, "name": "c%d" % i} for i, e in enumerate(exprs)]})
    return [o["type"] for o in d["outputs"]], d


def test_decimal_type_rules():
    one = lit(1, ("Decimal128", 20, 0))
    t, d = out_types([
        binary(one, Op.Minus, col("d", FIELDS)),                                   # (20,0)-(15,2) -> (23,2)
        binary(col("e", FIELDS), Op.Multiply, binary(one, Op.Minus, col("d", FIELDS))),   # (15,2)*(23,2) -> (38,4)
        binary(binary(col("e", FIELDS), Op.Multiply, binary(one, Op.Minus, col("d", FIELDS))), Op.Multiply, binary(one, Op.Plus, col("q", FIELDS))),
        binary(col("q", FIELDS), Op.Plus, col("i", FIELDS)),                       # Int64 -> (20,0): max(13,20)+2+1
        binary(col("q", FIELDS), Op.Multiply, col("j", FIELDS)),                   # Int32 -> (10,0): 15+10+1
        binary(col("i", FIELDS), Op.Plus, col("j", FIELDS)),
        binary(col("f", FIELDS), Op.Multiply, col("q", FIELDS)),
        cast(col("q", FIELDS), ("Decimal128", 20, 4)),
        binary(col("q", FIELDS), Op.Lt, col("i", FIELDS)),
        case([(binary(col("j", FIELDS), Op.Gt, lit(0, "Int32")), col("q", FIELDS))], lit(0)),
    ])
    assert t == ["Decimal128(23,2)", "Decimal128(38,4)", "Decimal128(38,6)", "Decimal128(23,2)", "Decimal128(26,2)", "Int64", "Float64",
                 "Decimal128(20,4)", "Boolean", "Decimal128(22,2)"]
    # the shared subexpression e*(1-d) is evaluated once (CSE) and 64x64->128 multiplies are used when ranges allow
    insns = " ".join(d["program"]["insns"])
    assert insns.count("MULW") >= 1 and "MUL r" in insns


def test_aggregate_schema_and_states():
    groups = [{"expr": col("s", FIELDS), "name": "s"}]
    aggs = [{"fn": "SUM", "expr": col("q", FIELDS), "name": "SUM(q)"}, {"fn": "AVG", "expr": col("q", FIELDS), "name": "AVG(q)"},
            {"fn": "AVG", "expr": col("i", FIELDS), "name": "AVG(i)"}, {"fn": "COUNT", "expr": lit(1), "name": "COUNT(*)"},
            {"fn": "SUM", "expr": col("j", FIELDS), "name": "SUM(j)"}, {"fn": "MIN", "expr": col("dt", FIELDS), "name": "MIN(dt)"}]
    part = g.compile_check({"op": "aggregate", "mode": "Partial", "input": {"fields": FIELDS}, "group_expr": groups, "aggr_expr": aggs})
    assert [(o["name"], o["type"]) for o in part["outputs"]] == [
        ("s", "Utf8"), ("SUM(q)[sum]", "Decimal128(25,2)"), ("AVG(q)[count]", "UInt64"), ("AVG(q)[sum]", "Decimal128(25,2)"),
        ("AVG(i)[count]", "UInt64"), ("AVG(i)[sum]", "Float64"), ("COUNT(*)[count]", "Int64"), ("SUM(j)[sum]", "Int64"), ("MIN(dt)[min]", "Date32")]
    # SUM(q) and AVG(q) share one accumulator; COUNT(*) and the AVG counts share another
    assert len(part["acc_kinds"]) == 5
    single = g.compile_check({"op": "aggregate", "mode": "Single", "input": {"fields": FIELDS}, "group_expr": groups, "aggr_expr": aggs})
    assert [o["type"] for o in single["outputs"]] == ["Utf8", "Decimal128(25,2)", "Decimal128(19,6)", "Float64", "Int64", "Int64", "Date32"]


def test_descriptor_errors_are_loud():
    for bad, msg in [
        ({"op": "filter", "input": {"fields": FIELDS}, "predicate": col("q", FIELDS)}, "predicate must be boolean"),
        ({"op": "project", "input": {"fields": FIELDS}, "exprs": [{"expr": binary(col("s", FIELDS), Op.Plus, lit(1)), "name": "x"}]}, "unsupported operands"),
        # (a long literal is taken in COMPARISONS since round 3 -- compared through its 15-byte prefix, exact for every value a register holds --
        # and still refused as a value)
        ({"op": "project", "input": {"fields": FIELDS}, "exprs": [{"expr": lit("a string longer than fifteen bytes"), "name": "x"}]}, "15 bytes"),
        ({"op": "frobnicate", "input": {"fields": FIELDS}}, "unknown op"),
        ({"op": "sort", "input": {"fields": FIELDS}, "expr": []}, "sort needs"),
        ({"op": "aggregate", "mode": "Single", "input": {"fields": FIELDS}, "group_expr": [], "aggr_expr": [{"fn": "MEDIAN", "expr": col("q", FIELDS), "name": "m"}]}, "MEDIAN"),
    ]:
        with pytest.raises(g.GpuqError) as e:
            g.compile_check(bad)
        assert msg in str(e.value), (msg, str(e.value))


def test_in_list_and_register_pressure():
    e = in_list(col("j", FIELDS), [lit(k, "Int32") for k in range(8)])
    d = g.compile_check({"op": "filter", "input": {"fields": FIELDS}, "predicate": e})
    assert d["program"]["pred_reg"] >= 0 and len(d["program"]["insns"]) <= 48
    with pytest.raises(g.GpuqError):
        g.compile_check({"op": "filter", "input": {"fields": FIELDS}, "predicate": in_list(col("j", FIELDS), [lit(k, "Int32") for k in range(60)])})      # > 48 immediates
    # 40 values fit since round 3 (q19's JoinFilter holds 25 literals: 48 immediates, 192 instructions)
    d40 = g.compile_check({"op": "filter", "input": {"fields": FIELDS}, "predicate": in_list(col("j", FIELDS), [lit(k, "Int32") for k in range(40)])})
    assert len(d40["program"]["insns"]) <= 192


def test_q1_plan_schema_matches_reference_answer_shape():
    """benchmarks/src/bin/tpch.rs:1101-1112 get_answer_schema(1): 10 columns, Utf8 x2, 7 decimals, Int64 count."""
    import tpch_util as T
    cols = [g.DeviceColumn(n, t, None, 0, nullable=False) for n, t in [("l_quantity", D152), ("l_extendedprice", D152), ("l_discount", D152), ("l_tax", D152),
                                                                       ("l_returnflag", "Utf8"), ("l_linestatus", "Utf8"), ("l_shipdate", "Date32")]]
    plan = T.q1_plan(g.MemoryExec([g.DeviceTable(cols, 0)]))
    sch = plan.schema()
    assert [f["name"] for f in sch] == ["l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order"]
    assert [f["type"] for f in sch] == ["Utf8", "Utf8", {"Decimal128": [25, 2]}, {"Decimal128": [25, 2]}, {"Decimal128": [38, 4]}, {"Decimal128": [38, 6]},
                                        {"Decimal128": [19, 6]}, {"Decimal128": [19, 6]}, {"Decimal128": [19, 6]}, "Int64"]
    # stage tree as in scheduler/src/planner.rs:376-392
    names = []
    p = plan
    while True:
        names.append(type(p).__name__)
        ch = p.children()
        if not ch:
            break
        p = ch[0]
    assert names == ["SortExec", "ProjectionExec", "AggregateExec", "CoalesceBatchesExec", "AggregateExec", "ProjectionExec", "CoalesceBatchesExec", "FilterExec", "MemoryExec"]


def test_native_plan_json_is_parsed_without_a_device():
    """gpuq_plan_create only builds the node tree (operators are compiled at first execution): the plan grammar can be
    validated on a host without a GPU.  Unknown node types are refused loudly (GPUQ_ERR_UNSUPPORTED), malformed plans are
    GPUQ_ERR_INVALID."""
    import ctypes as C
    import json
    from arrow_ballista_amd import binding as B
    L = B.lib()
    schema = [{"name": "k", "type": "Int64", "nullable": False}, {"name": "v", "type": {"Decimal128": [15, 2]}, "nullable": True}]
    col = lambda n: {"column": {"name": n}}
    plan = {"SortExec": {"expr": [{"expr": col("s"), "asc": False, "nulls_first": True}], "fetch": 10, "input":
            {"AggregateExec": {"mode": "Single", "group_expr": [{"expr": col("k"), "name": "k"}], "aggr_expr": [{"fn": "SUM", "expr": col("v"), "name": "s"}], "input":
             {"CoalesceBatchesExec": {"input": {"FilterExec": {"expr": {"is_not_null_expr": {"expr": col("v")}}, "input":
              {"HashJoinExec": {"left": {"MemoryExec": {"schema": schema, "partitions": [0]}}, "right": {"MemoryExec": {"schema": schema, "partitions": [1, 2]}},
                                "on": [{"left": col("k"), "right": col("k")}], "join_type": "LeftSemi", "partition_mode": "CollectLeft", "null_equals_null": False}}}}}}}}}}
    h = C.c_void_p()
    # a context handle is only dereferenced at execution time
    fake_ctx = C.c_void_p(1)
    assert L.gpuq_plan_create(fake_ctx, json.dumps(plan).encode(), C.byref(h)) == 0, L.gpuq_plan_last_error()
    assert L.gpuq_plan_num_partitions(h) == 2            # the join follows its probe (right) side: 2 partitions, kept by Filter / Aggregate / Sort
    L.gpuq_plan_free(h)
    h = C.c_void_p()
    assert L.gpuq_plan_create(fake_ctx, json.dumps({"WindowAggExec": {"input": plan}}).encode(), C.byref(h)) == 3
    assert b"WindowAggExec" in L.gpuq_plan_last_error()
    assert L.gpuq_plan_create(fake_ctx, b'{"FilterExec": {"input": 1}}', C.byref(h)) == 1
    assert L.gpuq_plan_create(fake_ctx, b'not json', C.byref(h)) == 1


def test_ipc_peek_walks_an_arrow_cpp_stream_on_the_host():
    """gpuq_ipc_peek (host-only flatbuffer reader) against a stream written by Arrow C++: message kinds, body sizes, row
    and buffer counts, compression codec; the walk lands exactly on the end-of-stream marker."""
    import ctypes as C
    import io
    import numpy as np
    import pyarrow as pa
    from arrow_ballista_amd import binding as B
    from arrow_ballista_amd.shuffle import gpuq_ipc_info
    L = B.lib()
    t = pa.table({"a": pa.array(np.arange(5000, dtype=np.int64)), "s": pa.array(["x%d" % (i % 7) for i in range(5000)]),
                  "d": pa.array([None if i % 5 == 0 else i for i in range(5000)], type=pa.int32())})
    for comp, codec in (("lz4", 0), ("zstd", 1), (None, -1)):
        sink = io.BytesIO()
        with pa.ipc.new_stream(sink, t.schema, options=pa.ipc.IpcWriteOptions(compression=comp)) as w:
            for b in t.to_batches(max_chunksize=2000):
                w.write_batch(b)
        raw = sink.getvalue()
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        pos, kinds, rows = 0, [], 0
        info = gpuq_ipc_info()
        while True:
            rc = L.gpuq_ipc_peek(C.c_void_p(C.addressof(buf) + pos), len(raw) - pos, C.byref(info))
            assert rc == 0, L.gpuq_ipc_last_error()
            kinds.append(info.header_type)
            if info.header_type == 0:
                break
            if info.header_type == 3:
                assert info.codec == codec and info.n_nodes == 3 and info.n_buffers == 7
                rows += info.n_rows
            pos += info.metadata_bytes + info.body_bytes
        assert kinds == [1, 3, 3, 3, 0] and rows == 5000 and pos + 8 == len(raw)
    # too few bytes: CAPACITY with the size needed; garbage: INVALID
    assert L.gpuq_ipc_peek(C.c_void_p(C.addressof(buf)), 12, C.byref(info)) == 4 and info.metadata_bytes > 12
    junk = (C.c_uint8 * 64)(*([7] * 64))
    assert L.gpuq_ipc_peek(C.c_void_p(C.addressof(junk)), 64, C.byref(info)) == 1


def test_ipc_schema_message_is_read_by_arrow_cpp():
    """gpuq_ipc_schema_message (hand-written Schema.fbs flatbuffer) parsed by Arrow C++: every supported type, nullability,
    empty and non-ASCII names; schema + end-of-stream marker is a valid empty stream."""
    import ctypes as C
    import pyarrow as pa
    from arrow_ballista_amd import binding as B
    from arrow_ballista_amd.shuffle import EOS, _fields_of
    L = B.lib()
    sch = pa.schema([pa.field("a", pa.int64(), False), pa.field("b", pa.int32()), pa.field("u32", pa.uint32()), pa.field("u64", pa.uint64(), False),
                     pa.field("f", pa.float64()), pa.field("s", pa.string()), pa.field("flag", pa.bool_(), False), pa.field("dec", pa.decimal128(15, 2)),
                     pa.field("d", pa.date32()), pa.field("", pa.decimal128(38, 10)), pa.field("ünï", pa.string(), False)])
    for s in (sch, pa.schema([]), pa.schema([pa.field("x" * 200, pa.date32())])):
        fields, _ = _fields_of(s)
        ln = C.c_int64(0)
        assert L.gpuq_ipc_schema_message(fields, len(s), None, 0, C.byref(ln)) == 0 and ln.value % 8 == 0
        buf = (C.c_uint8 * ln.value)()
        assert L.gpuq_ipc_schema_message(fields, len(s), buf, ln.value - 1, C.byref(ln)) == 4          # CAPACITY
        assert L.gpuq_ipc_schema_message(fields, len(s), buf, ln.value, C.byref(ln)) == 0
        raw = bytes(buf)
        assert pa.ipc.read_schema(pa.py_buffer(raw)).equals(s)
        t = pa.ipc.open_stream(raw + EOS).read_all()
        assert t.num_rows == 0 and t.schema.equals(s)


def test_native_plan_parses_the_stage_driver_nodes():
    """ShuffleWriterExec / ShuffleReaderExec in the native plan grammar: parsed and validated on the host (no device)."""
    import ctypes as C
    import json
    from arrow_ballista_amd import binding as B
    L = B.lib()
    leaf = {"ShuffleReaderExec": {"schema": [{"name": "k", "type": "Int64", "nullable": False}, {"name": "v", "type": {"Decimal128": [15, 2]}, "nullable": True}],
                                  "partition": [[{"path": "/tmp/a.arrow"}, {"path": "/tmp/b.arrow"}], [], ["/tmp/c.arrow"]]}}
    col = {"column": {"name": "k", "index": 0}}
    plan = {"ShuffleWriterExec": {"input": leaf, "job_id": "j", "stage_id": 4, "work_dir": "/tmp/w", "output_partitioning": {"hash_expr": [col], "partition_count": 8}}}
    h = C.c_void_p()
    ctx = C.c_void_p(1)       # never dereferenced by gpuq_plan_create
    assert L.gpuq_plan_create(ctx, json.dumps(plan).encode(), C.byref(h)) == 0, L.gpuq_plan_last_error()
    assert L.gpuq_plan_num_partitions(h) == 3
    L.gpuq_plan_free(h)
    plan["ShuffleWriterExec"]["work_dir"] = ""
    assert L.gpuq_plan_create(ctx, json.dumps(plan).encode(), C.byref(h)) == 1 and b"work_dir" in L.gpuq_plan_last_error()


def test_ipc_peek_walks_the_reference_shuffle_file():
    """tests/golden/shuffle_data.arrow is the shuffle file the reference's own reader test loads
    (ballista/core/tests/data.arrow, async_reader/mod.rs:331-357: 561 Utf8 rows written by the reference's
    ShuffleWriterExec -- Arrow IPC stream, LZ4_FRAME buffers).  The host-side walk sees Schema, the RecordBatch(es) with
    561 rows in total and LZ4_FRAME compression, and ends on the end-of-stream marker (or the end of the file: the
    reference's reader tolerates a missing marker, async_reader/mod.rs:179-189)."""
    import ctypes as C
    import os
    import pyarrow as pa
    from arrow_ballista_amd import binding as B
    from arrow_ballista_amd.shuffle import gpuq_ipc_info
    L = B.lib()
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "shuffle_data.arrow"), "rb").read()
    ref = pa.ipc.open_stream(raw).read_all()
    assert ref.num_rows == 561 and ref.schema.names == ["$d"]
    buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
    pos, kinds, rows, codecs = 0, [], 0, set()
    info = gpuq_ipc_info()
    while pos < len(raw):
        rc = L.gpuq_ipc_peek(C.c_void_p(C.addressof(buf) + pos), len(raw) - pos, C.byref(info))
        assert rc == 0, L.gpuq_ipc_last_error()
        kinds.append(info.header_type)
        if info.header_type == 0:
            pos += 8
            break
        if info.header_type == 3:
            rows += info.n_rows
            codecs.add(info.codec)
        pos += info.metadata_bytes + info.body_bytes
    assert kinds[0] == 1 and 3 in kinds and rows == 561 and codecs == {0}
    assert pos == len(raw)


def test_ipc_schema_metadata_round_trip_with_arrow_cpp():
    """Schema.custom_metadata written by gpuq_ipc_schema_message_kv is what Arrow C++ reads as Schema.metadata, and
    gpuq_ipc_schema_metadata finds keys in messages written by either side (the partition-function marker of hash-partitioned
    shuffle files, GPUQ_PARTITION_FN_KEY in include/gpuq.h)."""
    import ctypes as C
    import pyarrow as pa
    from arrow_ballista_amd import binding as B
    L = B.lib()
    fields = (B.gpuq_field_info * 2)()
    fields[0].name, fields[0].type, fields[0].nullable = b"k", 3, 0          # Int64
    fields[1].name, fields[1].type, fields[1].precision, fields[1].scale, fields[1].nullable = b"d", 6, 15, 2, 1   # Decimal128(15,2)
    keys = (C.c_char_p * 2)(b"gpuq.partition_fn", b"other")
    vals = (C.c_char_p * 2)(b"gpuq-mix64-v1", b"")
    ln = C.c_int64(0)
    assert L.gpuq_ipc_schema_message_kv(fields, 2, keys, vals, 2, None, 0, C.byref(ln)) == 0
    buf = (C.c_uint8 * ln.value)()
    assert L.gpuq_ipc_schema_message_kv(fields, 2, keys, vals, 2, buf, ln.value, C.byref(ln)) == 0, L.gpuq_ipc_last_error()
    sch = pa.ipc.read_schema(pa.py_buffer(bytes(buf)))
    assert sch.names == ["k", "d"] and sch.field("d").type == pa.decimal128(15, 2) and not sch.field("k").nullable
    assert sch.metadata == {b"gpuq.partition_fn": b"gpuq-mix64-v1", b"other": b""}

    def lookup(raw, key):
        b = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        out = C.create_string_buffer(64)
        found = C.c_int(0)
        assert L.gpuq_ipc_schema_metadata(b, len(raw), key, out, 64, C.byref(found)) == 0, L.gpuq_ipc_last_error()
        return out.value if found.value else None
    assert lookup(bytes(buf), b"gpuq.partition_fn") == b"gpuq-mix64-v1" and lookup(bytes(buf), b"other") == b"" and lookup(bytes(buf), b"absent") is None
    theirs = pa.schema([pa.field("x", pa.int32())], metadata={"a": "1", "gpuq.partition_fn": "something-else"}).serialize().to_pybytes()
    assert lookup(theirs, b"gpuq.partition_fn") == b"something-else" and lookup(theirs, b"a") == b"1"
    plain = pa.schema([pa.field("x", pa.int32())]).serialize().to_pybytes()
    assert lookup(plain, b"gpuq.partition_fn") is None
    # the message without metadata is unchanged by the new entry point
    l0, l1 = C.c_int64(0), C.c_int64(0)
    assert L.gpuq_ipc_schema_message(fields, 2, None, 0, C.byref(l0)) == 0 and L.gpuq_ipc_schema_message_kv(fields, 2, None, None, 0, None, 0, C.byref(l1)) == 0
    assert l0.value == l1.value


def test_plan_schema_is_known_before_execution_and_without_a_device():
    """QueryStageExecutor::schema() (execution_engine.rs:59) is read BEFORE the stage runs (executor_server.rs:530-534).
    gpuq_plan_schema types a whole plan on the host -- here q1's two-stage shape and q3 (joins, decimal arithmetic with
    DataFusion's precision rules, AVG widening), built without a GPU context -- and must agree with what the oracle's typing says
    (the GPU suite checks the same schemas against executed results)."""
    import ctypes as C
    import json
    import pyarrow as pa
    import arrow_ballista_amd as g
    from arrow_ballista_amd import binding as B
    from arrow_ballista_amd.native import plan_to_json
    from arrow_ballista_amd.table import type_json
    import tpch_util as T
    L = B.lib()

    def host_schema(plan):
        class _TC:      # plan_to_json only asks MemoryExec leaves for their tables
            pass
        inputs = []
        js = json.dumps(plan_to_json(plan, _TC(), inputs))
        h = C.c_void_p()
        assert L.gpuq_plan_create(None, js.encode(), C.byref(h)) == 0, L.gpuq_plan_last_error()
        n = C.c_int(0)
        assert L.gpuq_plan_schema(h, None, 0, C.byref(n)) == 0, L.gpuq_plan_last_error()
        f = (B.gpuq_field_info * n.value)()
        assert L.gpuq_plan_schema(h, f, n.value, C.byref(n)) == 0, L.gpuq_plan_last_error()
        out = C.c_void_p()
        assert L.gpuq_plan_execute(h, None, 0, None, 0, C.byref(out)) != 0 and b"without a context" in L.gpuq_plan_last_error()
        L.gpuq_plan_free(h)
        return [(f[i].name.decode(), type_json(f[i].type, f[i].precision, f[i].scale), bool(f[i].nullable)) for i in range(n.value)]

    def empty(cols):
        return g.MemoryExec([None], schema=[{"name": n, "type": t, "nullable": False} for n, t in cols])
    D = T.D152
    li = empty([("l_orderkey", "Int64"), ("l_suppkey", "Int64"), ("l_quantity", D), ("l_extendedprice", D), ("l_discount", D), ("l_tax", D),
                ("l_returnflag", "Utf8"), ("l_linestatus", "Utf8"), ("l_shipdate", "Date32")])
    q1 = host_schema(T.q1_plan(li))
    assert [n for n, _, _ in q1] == ["l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order"]
    dec = lambda p, s: {"Decimal128": [p, s]}
    assert [t for _, t, _ in q1] == ["Utf8", "Utf8", dec(25, 2), dec(25, 2), dec(38, 4), dec(38, 6), dec(19, 6), dec(19, 6), dec(19, 6), "Int64"]
    od = empty([("o_orderkey", "Int64"), ("o_custkey", "Int64"), ("o_orderdate", "Date32"), ("o_shippriority", "Int32")])
    cu = empty([("c_custkey", "Int64"), ("c_nationkey", "Int64"), ("c_mktsegment", "Utf8")])
    q3 = host_schema(T.q3_plan(cu, od, li))
    assert [(n, t) for n, t, _ in q3] == [("l_orderkey", "Int64"), ("revenue", dec(38, 4)), ("o_orderdate", "Date32"), ("o_shippriority", "Int32")]
    # the distributed plans (exchange nodes inside) type the same way
    su = empty([("s_suppkey", "Int64"), ("s_nationkey", "Int64")])
    na = empty([("n_nationkey", "Int64"), ("n_name", "Utf8"), ("n_regionkey", "Int64")])
    re_ = empty([("r_regionkey", "Int64"), ("r_name", "Utf8")])
    assert [(n, t) for n, t, _ in host_schema(T.q5_dist_plan(cu, od, li, su, na, re_, 2))] == [("n_name", "Utf8"), ("revenue", dec(38, 4))]
    assert [(n, t) for n, t, _ in host_schema(T.q3_dist_plan(cu, od, li, 2, "partitioned"))] == [(n, t) for n, t, _ in q3]
    # outer joins make the non-preserved side nullable; semi joins keep one side
    cs, os_ = cu.schema(), od.schema()
    from arrow_ballista_amd.expr import col
    on = [(col("c_custkey", cs), col("o_custkey", os_))]
    full = host_schema(g.HashJoinExec(cu, od, on, None, "Full", "CollectLeft", False))
    assert all(nullable for _, _, nullable in full) and len(full) == 7
    assert host_schema(g.HashJoinExec(cu, od, on, None, "RightSemi", "CollectLeft", False)) == [(f["name"], f["type"], False) for f in os_]
    # the stage root's result batch (shuffle_writer.rs:470-520)
    w = host_schema(g.ShuffleWriterExec("j", 1, od, "/tmp/x", ([col("o_custkey", os_)], 4)))
    assert [n for n, _, _ in w] == ["partition", "path", "num_rows", "num_batches", "num_bytes"] and [t for _, t, _ in w] == ["UInt32", "Utf8", "UInt64", "UInt64", "UInt64"]


def test_parquet_footer_is_read_on_the_host():
    """gpuq_parquet_schema walks the Thrift-compact footer without a device: the reference's alltypes_plain.parquet
    (ballista/client/testdata) and a pyarrow-written file with logical types."""
    import io
    import pyarrow as pa
    import pyarrow.parquet as pq
    from arrow_ballista_amd import scan
    L = g.lib()
    fields, rows = scan.parquet_schema(L, os.path.join(os.path.dirname(__file__), "golden", "alltypes_plain.parquet"))
    assert rows == 8
    assert fields == [("id", "Int32", True), ("bool_col", "Boolean", True), ("tinyint_col", "Int32", True), ("smallint_col", "Int32", True),
                      ("int_col", "Int32", True), ("bigint_col", "Int64", True), ("float_col", "Float32", True), ("double_col", "Float64", True),
                      ("date_string_col", "Utf8", True), ("string_col", "Utf8", True), ("timestamp_col", {"Timestamp": ["Nanosecond", None]}, True)]
    t = pa.table({"d": pa.array([1, 2], pa.date32()), "x": pa.array([1, None], pa.decimal128(15, 2)), "s": ["a", "b"]})
    t = t.cast(pa.schema([pa.field("d", pa.date32(), False), pa.field("x", pa.decimal128(15, 2), True), pa.field("s", pa.string(), False)]))
    buf = io.BytesIO()
    pq.write_table(t, buf)
    assert scan.parquet_schema(L, buf.getvalue()) == ([("d", "Date32", False), ("x", {"Decimal128": [15, 2]}, True), ("s", "Utf8", False)], 2)
    with pytest.raises(g.GpuqError):
        scan.parquet_schema(L, b"PAR1garbagePAR1")


def test_every_runtime_kernel_source_compiles_for_gfx950():
    """The sources the JIT front-end hands to hiprtc (one per kernel id) are cross-compiled with hipcc here: a source that stops
    compiling would make operators fall back to their AOT kernels on the GPU box -- correct, silently slower, and invisible to
    the parity tests."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "jit_compile_check.py")
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ok = [l for l in r.stdout.splitlines() if " OK: " in l]
    assert len(ok) == 18, r.stdout


def test_parquet_footer_parser_survives_corruption():
    """The Thrift-compact reader takes untrusted bytes: 400 random single- and multi-byte corruptions and every truncation of a valid
    footer either parse or fail with an error -- no crash, no hang (run on the host: gpuq_parquet_schema needs no device)."""
    import io
    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq
    from arrow_ballista_amd import scan
    L = g.lib()
    t = pa.table({"a": pa.array(range(1000), pa.int64()), "s": pa.array(["x%d" % i for i in range(1000)]), "d": pa.array([1] * 1000, pa.int32()).cast(pa.date32())})
    buf = io.BytesIO()
    pq.write_table(t, buf, row_group_size=300)
    good = buf.getvalue()
    flen = int.from_bytes(good[-8:-4], "little")
    r = np.random.default_rng(1)
    outcomes = {"ok": 0, "err": 0}
    for it in range(400):
        bad = bytearray(good)
        for _ in range(1 + it % 4):
            bad[len(good) - 8 - int(r.integers(1, flen + 1))] = int(r.integers(0, 256))
        try:
            scan.parquet_schema(L, bytes(bad)); outcomes["ok"] += 1
        except g.GpuqError:
            outcomes["err"] += 1
    for cut in range(1, flen, max(1, flen // 60)):
        bad = good[: len(good) - 8 - cut] + good[-8:]            # footer shorter than its length field says / shifted
        try:
            scan.parquet_schema(L, bad)
        except g.GpuqError:
            outcomes["err"] += 1
    assert outcomes["err"] > 50


def _thrift_footer(schema_elems, extra_struct_depth=0):
    """A minimal Parquet file image whose footer is hand-written Thrift compact: FileMetaData {1: version, 2: schema list,
    3: num_rows, 4: row_groups (empty)} -- enough for gpuq_parquet_schema.  schema_elems: list of dicts (type, type_length,
    repetition, name, num_children, converted, scale, precision)."""
    def varint(v):
        out = bytearray()
        while True:
            b = v & 0x7F; v >>= 7
            if v:
                out.append(b | 0x80)
            else:
                out.append(b); return bytes(out)

    def zz(v):
        return varint((v << 1) ^ (v >> 63))

    def field(delta, typ):
        return bytes([(delta << 4) | typ])

    def elem(e):
        out = bytearray(); last = 0
        for fid, key, typ in ((1, "type", 5), (2, "type_length", 5), (3, "repetition", 5), (4, "name", 8), (5, "num_children", 5), (6, "converted", 5), (7, "scale", 5), (8, "precision", 5)):
            if key not in e:
                continue
            out += field(fid - last, typ); last = fid
            if typ == 8:
                nb = e[key].encode(); out += varint(len(nb)) + nb
            else:
                out += zz(e[key])
        return bytes(out) + b"\x00"
    meta = bytearray()
    meta += field(1, 5) + zz(1)                                            # version
    meta += field(1, 9) + bytes([(len(schema_elems) << 4) | 12]) + b"".join(elem(e) for e in schema_elems)
    meta += field(1, 6) + zz(0)                                            # num_rows
    meta += field(1, 9) + bytes([(0 << 4) | 12])                           # row_groups: empty list of structs
    if extra_struct_depth:                                                 # field 5 (key_value_metadata slot) abused as a struct nested N deep
        meta += field(1, 12) + bytes([0x1C]) * extra_struct_depth + b"\x00" * (extra_struct_depth + 1)
    meta += b"\x00"
    return b"PAR1" + bytes(meta) + len(meta).to_bytes(4, "little") + b"PAR1"


def test_parquet_footer_with_hostile_type_length_and_nesting():
    """FIXED_LEN_BYTE_ARRAY's type_length is a zigzag varint from an untrusted footer: 0 and negative values must not
    reach the page decoders (they index file + vi * type_length), and a footer of nested struct headers must not recurse the host
    stack away.  Host only: gpuq_parquet_schema announces such a column as undecodable; the deep footer is an error, not a crash."""
    from arrow_ballista_amd import scan
    L = g.lib()
    root = {"name": "schema", "num_children": 1}
    for tl, ok in ((16, True), (5, True), (0, False), (-4, False), (17, False), (-2**31, False)):
        img = _thrift_footer([root, {"type": 7, "type_length": tl, "repetition": 0, "name": "x", "converted": 5, "scale": 2, "precision": 9}])
        fields, rows = scan.parquet_schema(L, img)
        assert rows == 0 and len(fields) == 1 and fields[0][0] == "x"
        assert (fields[0][1] is not None) == ok, (tl, fields)
    # 60 nested structs are walked; 100 000 are refused (they used to overflow the stack)
    scan.parquet_schema(L, _thrift_footer([root, {"type": 2, "repetition": 0, "name": "y"}], extra_struct_depth=60))
    with pytest.raises(g.GpuqError, match="nested deeper"):
        scan.parquet_schema(L, _thrift_footer([root, {"type": 2, "repetition": 0, "name": "y"}], extra_struct_depth=100_000))
