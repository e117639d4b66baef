/*
 * gpuq.h -- C ABI of libgpuq.so, the MI355X (gfx950) columnar physical-operator engine that stands
 * behind Ballista's executor as the task runner for the DataFusion hot path
 * (FilterExec / ProjectionExec / AggregateExec / HashJoinExec / SortExec / hash repartition).
 *
 * Drop-in boundary (citations relative to the reference tree, coralogix/arrow-ballista @ 2025-02-02):
 *   - ballista/executor/src/execution_engine.rs:34-60   ExecutionEngine / QueryStageExecutor traits:
 *     a GPU engine implements them in Rust and forwards the operator work to the functions below
 *     (binding shown in INTEGRATION.md).
 *   - ballista/core/src/execution_plans/shuffle_writer.rs:255,341-392  the per-batch pull loop and
 *     BatchPartitioner::partition that gpuq_partition_* replaces.
 *   - DataFusion v34 operators named by ballista/core/src/physical_optimizer/task_group.rs:23-31,
 *     parameter surface pinned by ballista/core/proto/datafusion.proto:1109-1525.
 *
 * Conventions
 *   - Plain C: pointers, sizes, POD structs.  No C++/torch/Arrow-library types.
 *   - Every function returns a gpuq_status (0 = OK); gpuq_last_error(ctx) gives the UTF-8 message.
 *     No exception crosses the boundary.
 *   - All data pointers in gpuq_column / index vectors are DEVICE pointers (HBM) unless a parameter
 *     says otherwise.  Layouts are Arrow's physical layouts (little-endian values, int32 offsets +
 *     bytes for Utf8, LSB-first validity bitmaps), so a host that owns Arrow buffers copies them to
 *     the device byte-for-byte.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls enqueue work on that
 *     stream; the ones documented as synchronous also wait for it.
 *   - Handles are freed by exactly one *_free; distinct handles may be used from distinct threads.
 *   - There is no CPU fallback: without a usable HIP device gpuq_ctx_create fails.
 */
#ifndef GPUQ_H
#define GPUQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPUQ_ABI_VERSION 2

typedef enum gpuq_status {
  GPUQ_OK = 0,
  GPUQ_ERR_INVALID = 1,      /* bad argument / malformed descriptor */
  GPUQ_ERR_HIP = 2,          /* HIP runtime error (message has the hipError) */
  GPUQ_ERR_UNSUPPORTED = 3,  /* valid request the device path does not implement (fails loudly, never falls back) */
  GPUQ_ERR_CAPACITY = 4,     /* caller-provided output capacity too small; required size reported */
  GPUQ_ERR_INTERNAL = 5,
  GPUQ_ERR_CANCELLED = 6,    /* gpuq_task_cancel reached the task before it finished */
  GPUQ_ERR_RETRY = 7,        /* gpuq_ops_settle: something a deferred run assumed did not hold; run again without deferral */
  GPUQ_ERR_PEER = 8          /* exchange: a rank of the node announced a failure (or cannot hold what it would receive); no payload moved */
} gpuq_status;

/* Logical types (subset of ballista/core/proto/datafusion.proto:1004-1040 ArrowType). */
typedef enum gpuq_type {
  GPUQ_NULL = 0, GPUQ_BOOL = 1, GPUQ_INT32 = 2, GPUQ_INT64 = 3, GPUQ_DATE32 = 4, GPUQ_FLOAT64 = 5,
  GPUQ_DECIMAL128 = 6, GPUQ_UTF8 = 7, GPUQ_UINT32 = 8, GPUQ_UINT64 = 9,
  /* second wave: Arrow physical layout as well (1 / 2 / 4 / 8 bytes per value); kernels widen on load and narrow on store */
  GPUQ_INT8 = 10, GPUQ_INT16 = 11, GPUQ_UINT8 = 12, GPUQ_UINT16 = 13, GPUQ_FLOAT32 = 14,
  GPUQ_TIMESTAMP = 15, /* int64 count of `precision` = TimeUnit (0 s, 1 ms, 2 us, 3 ns) since the epoch; the time zone is schema metadata */
  GPUQ_DATE64 = 16     /* int64 milliseconds since the epoch */
  /* LargeUtf8 and Dictionary(_, T) columns are converted where they enter: gpuq_table_import_arrow and gpuq_ingest_* narrow the offsets / decode the dictionary while staging; inside, Utf8 has 32-bit offsets */
} gpuq_type;

/* Device representation of an output column. */
#define GPUQ_REPR_ARROW 0     /* Arrow physical layout */
#define GPUQ_REPR_PACKED15 1  /* Utf8 of <= 15 bytes as 16-byte value: bytes big-endian in bits 127..8, length in bits 7..0
                                 (convert with gpuq_unpack_utf8) */

typedef struct gpuq_column {
  int32_t type;            /* gpuq_type */
  int32_t precision;       /* Decimal128 */
  int32_t scale;           /* Decimal128 */
  int32_t repr;            /* GPUQ_REPR_* (inputs: ARROW) */
  const void* data;        /* values; Utf8: bytes; Bool: bitmap */
  const int32_t* offsets;  /* Utf8: length+1 offsets, else NULL */
  const uint8_t* validity; /* Arrow validity bitmap or NULL (no nulls) */
  int64_t length;
} gpuq_column;

typedef struct gpuq_field_info {
  char name[256];
  int32_t type, precision, scale, nullable, repr, width; /* width = bytes per row in the fixed-width device layout */
} gpuq_field_info;

typedef struct gpuq_ctx gpuq_ctx;
typedef struct gpuq_op gpuq_op;
typedef struct gpuq_join_table gpuq_join_table;

/* ---- context ----------------------------------------------------------------------------- */
/* Streams and threads: every call takes the HIP stream its work is queued on.  Calls made on ONE stream are ordered by that
   stream.  Several host threads may drive one device -- through one context or several -- each on its own stream and with its
   own operators / plans / join tables (the reference runs `concurrent_tasks` tasks on its task-runner pool,
   cpu_bound_executor.rs:94-131): the library recycles device memory through a process-wide pool and orders a block released
   under one stream behind its next user on another with an event (csrc/devbuf.h).  One operator, plan or join table must not be
   used from two threads at once.  gpuq_last_error() is per calling thread. */
int gpuq_abi_version(void);
/* json_opts: NULL or {"device":N}.  Fails (returns NULL) when no HIP device is usable;
   gpuq_last_error(NULL) then holds the reason. */
gpuq_ctx* gpuq_ctx_create(int device_ordinal, const char* json_opts);
void gpuq_ctx_free(gpuq_ctx* ctx);
const char* gpuq_last_error(gpuq_ctx* ctx);
int gpuq_ctx_device_info(gpuq_ctx* ctx, char* buf, size_t cap); /* JSON: name, arch, cus, hbm_bytes */
/* Tuning switches (strings): "join_dense" = "0" | "1" (default 1; env GPUQ_JOIN_DENSE at ctx creation): join tables over ONE
   Int32 / Int64 / Date32 key whose build values span a bounded range are direct-addressed arrays instead of hash tables;
   "join_dense_ratio" = largest range / key-count ratio that still takes the array (default 4096);
   "join_radix" = "off" | "auto" | "force": partitioned probe (one LDS-staged radix pass over the probe rows on the high bits of
   key - min, then lookups that stay inside one L2-sized slice of the array at a time) for Inner / RightSemi joins over such
   an array; the pairs then come out in partition order instead of probe order.  Default "off": on MI355X the pass costs about
   what the random accesses it removes cost (2^28 probes: 5.5 vs 6.3 ms at 2^24 keys, 7.2 vs 7.3 ms at 2^27).
   "auto" takes it for >= 2^24 probe rows against a table of >= 64 MiB when a sample of the probe keys shows no locality;
   "join_radix_slice_log2" = table entries per partition (default 18 = 1 MiB slices). */
int gpuq_ctx_set_option(gpuq_ctx* ctx, const char* key, const char* value);

/* ---- runtime specialisation (JIT) ---------------------------------------------------------- */
/* The row front-end (column loads + expressions) of every operator exists twice: as interpreter kernels
   compiled ahead of time (always available, best for small inputs) and as a typed straight-line function
   generated from the same expression DAG and compiled with hiprtc on first use (best for large inputs).
   mode: "off" | "auto" | "wait" | "force".  "auto" (the default): nobody waits for a compile -- an input of >= min_rows rows hands its
   source to a worker thread on its first run and runs the interpreter kernels until the code object is there (a one-shot query does not
   pay hiprtc's 0.2-0.3 s per pipeline; smaller inputs are specialised when their program comes back a third time); gpuq_jit_wait
   blocks until the queue is empty (steady-state benches call it after their warm-up).  "wait": an input of >= min_rows rows waits for
   its compile (round 2's "auto").  "force": every input waits, errors are reported.  Falls back to the interpreter kernels when hiprtc
   is missing.  min_rows default 2^21; env GPUQ_JIT overrides at ctx creation. */
int gpuq_ctx_set_jit(gpuq_ctx* ctx, const char* mode, int64_t min_rows /* <0 = keep */);
int gpuq_ctx_jit_stats(gpuq_ctx* ctx, int* available, int* launches, char* last_error, size_t cap);
/* In "auto" mode a program that runs for the third time on inputs below min_rows is handed to a background thread for
   specialisation; its callers keep running the interpreter kernels until the compiled function is there.  This call waits
   until every compile requested so far has finished (benchmarks call it at the end of their warm-up). */
int gpuq_ctx_jit_wait(gpuq_ctx* ctx);
/* Call before the process exits (the Python binding registers it with atexit): waits for background compiles still in flight.  The
   run-time compiler's libraries are loaded on first use, i.e. after this library registered its own exit handler, so they are torn
   down first -- a worker thread still inside one of them at that moment takes the process down with it. */
void gpuq_jit_quiesce(void);
/* Compiled code objects are kept on disk ($GPUQ_JIT_CACHE_DIR, default $XDG_CACHE_HOME/gpuq-jit or ~/.cache/gpuq-jit; "off"
   disables), keyed by a hash of the whole translation unit, so only the first process on a host pays hiprtc for a pipeline
   (0.3-1.9 s); later processes load the code object (~ms).  Counters of this process: */
int gpuq_jit_cache_stats(int* disk_hits, int* compiles);

/* ---- device memory + Arrow C Data Interface ingest/egress ---------------------------------- */
/* A host that owns Arrow RecordBatches (arrow-rs `arrow::ffi`, pyarrow `_export_to_c`) hands them over
   here; buffers are copied to HBM through double-buffered pinned staging on `stream`.
   The structs are the Arrow C Data Interface ABI (https://arrow.apache.org/docs/format/CDataInterface.html). */
struct ArrowSchema {
  const char* format; const char* name; const char* metadata; int64_t flags; int64_t n_children;
  struct ArrowSchema** children; struct ArrowSchema* dictionary; void (*release)(struct ArrowSchema*); void* private_data;
};
struct ArrowArray {
  int64_t length; int64_t null_count; int64_t offset; int64_t n_buffers; int64_t n_children;
  const void** buffers; struct ArrowArray** children; struct ArrowArray* dictionary; void (*release)(struct ArrowArray*); void* private_data;
};
typedef struct gpuq_table gpuq_table;

/* Memory budget.  The reference runs operators under DataFusion's MemoryPool (RuntimeConfig::with_memory_limit; a SortExec or a join
   build that cannot reserve fails with ResourcesExhausted).  Every device byte this library holds -- operator workspaces, join tables,
   plan intermediates, imported / decoded tables -- is counted process-wide; with a limit set (bytes > 0; 0 = none; also the environment
   variable GPUQ_MEMORY_LIMIT at start-up) the allocation that would cross it fails with GPUQ_ERR_CAPACITY ("Resources exhausted: ...") and
   the call that needed it returns that status: the task fails, the process and its other tasks go on.  The caller's own input columns
   are not counted.  Nothing is spilled: the executors hold whole partitions, a partition that does not fit the budget has to be split
   by the planner (more partitions), as on the reference when spilling is disabled.
   gpuq_memory_stats: bytes in use now, the high-water mark (reset_peak != 0 sets it back to the current value after reading), bytes cached
   in the library's free list, and the limit. */
int gpuq_memory_limit(int64_t bytes);
int gpuq_memory_stats(int64_t* in_use_out, int64_t* peak_out, int64_t* cached_out, int64_t* limit_out, int reset_peak);

int gpuq_buffer_alloc(gpuq_ctx* ctx, size_t bytes, void** dev_out);
int gpuq_buffer_free(gpuq_ctx* ctx, void* dev);
int gpuq_copy_h2d(gpuq_ctx* ctx, void* stream, void* dst_dev, const void* src_host, size_t bytes); /* returns after the copy is enqueued and the host buffer is reusable */
int gpuq_copy_d2h(gpuq_ctx* ctx, void* stream, void* dst_host, const void* src_dev, size_t bytes); /* synchronous */

/* Import one RecordBatch (a struct-typed ArrowArray + its ArrowSchema).  The batch is NOT consumed: the
   caller keeps ownership and releases it as usual.  Supported column formats: b c C s S i I l L f g tdD tdm ts{s,m,u,n}:tz d:p,s u,
   and, converted while they are staged: U (LargeUtf8 -> 32-bit offsets; a column beyond 2^31 bytes is refused) and dictionary-encoded
   columns of any of these (decoded: the device column has the value type). */
int gpuq_table_import_arrow(gpuq_ctx* ctx, void* stream, const struct ArrowArray* batch, const struct ArrowSchema* schema, gpuq_table** out);
int64_t gpuq_table_num_rows(const gpuq_table* t);
int gpuq_table_num_columns(const gpuq_table* t);
int gpuq_table_column(const gpuq_table* t, int i, gpuq_column* col_out, gpuq_field_info* field_out);
void gpuq_table_free(gpuq_table* t);
/* Export device columns (GPUQ_REPR_ARROW or PACKED15) as a host RecordBatch; the caller releases `out`
   and `out_schema` through their release callbacks.  Synchronous. */
int gpuq_export_arrow(gpuq_ctx* ctx, void* stream, const gpuq_column* cols, const gpuq_field_info* fields, int n_cols, int64_t n_rows,
                      struct ArrowArray* out, struct ArrowSchema* out_schema);

/* ---- streaming ingest of a partition ------------------------------------------------------------------------------------
   The per-batch pull loop of a task (`stream.next()`, shuffle_writer.rs:341 / utils.rs:198) seen from the device: host Arrow
   RecordBatches (BASELINE configs[1]: 64 Ki rows each) are appended to ONE set of preallocated device columns.
   gpuq_ingest_push only queues a batch (it is MOVED, Arrow C Data Interface: the library calls its release callback once the
   copies have landed); n_threads workers (0 = 8) copy the buffers into their own pinned slots in parallel -- one thread feeding
   pinned memory cannot fill PCIe -- and issue the H2D copies on their own streams.  Batches land out of order; "rows landed" is
   the contiguous prefix that is complete.  A consumer reads the landed prefix through gpuq_ingest_columns (narrow `length`, or
   offset the pointers, to the row range it wants) on any stream while later batches are still in flight.
   max_rows / max_utf8_bytes (per Utf8 column) size the device columns; exceeding them is GPUQ_ERR_CAPACITY.  Nullable and
   Boolean columns need every batch but the last to hold a multiple of 8 rows.  Errors: gpuq_ingest_last_error(). */
typedef struct gpuq_ingest gpuq_ingest;
int gpuq_ingest_create(gpuq_ctx* ctx, const struct ArrowSchema* schema, int64_t max_rows, int64_t max_utf8_bytes, int n_threads, gpuq_ingest** out);
int gpuq_ingest_push(gpuq_ingest* ingest, struct ArrowArray* batch);
int gpuq_ingest_rows_landed(gpuq_ingest* ingest, int64_t* rows_out);
/* blocks until at least `rows` rows have landed, or everything pushed so far has; *rows_out = the landed prefix */
int gpuq_ingest_wait_rows(gpuq_ingest* ingest, int64_t rows, int64_t* rows_out);
int gpuq_ingest_columns(gpuq_ingest* ingest, gpuq_column* cols_out, gpuq_field_info* fields_out, int cap, int* n_out);
int gpuq_ingest_stats(gpuq_ingest* ingest, int64_t* rows_pushed, int64_t* rows_landed, int64_t* bytes_copied);
void gpuq_ingest_free(gpuq_ingest* ingest);      /* also frees the device columns */
const char* gpuq_ingest_last_error(void);

/* ---- scan-side decode ------------------------------------------- */
/* The leaves of the reference's TPC-H plans are CsvExec / ParquetExec over the files benchmarks/src/bin/tpch.rs:801-862
   registers (`.tbl` text with '|' and no header, or Parquet written by its `convert` command).  These entry points take the
   FILE bytes in host memory (read or mmap'ed by the caller -- the object-store read stays the reference's), move them to the
   device once and produce Arrow-layout columns in HBM as a gpuq_table; no value is parsed on the host.
   gpuq_csv_decode: `file_fields` describes every field of a line in order (name, type, precision/scale, nullable: an empty field
   of a nullable non-Utf8 column is NULL, as arrow-csv reads it); `projection` (or NULL = all) picks output columns by file index.
   A trailing delimiter before the newline (TPC-H .tbl) is accepted.  Types: Int32 Int64 Date32 (yyyy-mm-dd) Decimal128 Float64
   Boolean Utf8.  Refused loudly rather than decoded differently: quoted fields, a Float64 field outside the exactly-rounded fast
   path (> 15 significant digits or |exponent| > 22), lines whose field count differs from the schema (blank lines included), > 4 GiB per call.
   gpuq_parquet_decode: flat schemas; physical BOOLEAN INT32 INT64 DOUBLE BYTE_ARRAY FIXED_LEN_BYTE_ARRAY(decimal); DATE and
   DECIMAL annotations; required / optional columns; PLAIN, PLAIN_DICTIONARY / RLE_DICTIONARY and RLE-boolean pages (v1 and v2); UNCOMPRESSED
   SNAPPY, LZ4_RAW and ZSTD column chunks (Snappy / LZ4 pages are decompressed without a serial walk: element positions by doubling, copies by
   pointer jumping; ZSTD pages -- the reference's `convert` default -- one wave per page: FSE / Huffman tables in LDS, the four literal streams on four
   lanes, copies on all 64; GZIP, BROTLI, LZO and the deprecated framed LZ4 are GPUQ_ERR_UNSUPPORTED).  `columns` (or NULL = all) projects by name; only the projected
   chunks' bytes cross PCIe.  The host parses the Thrift footer and page headers, one wave decodes one page.
   Errors: gpuq_scan_last_error() (thread-local). */
typedef struct gpuq_csv_options { char delimiter; char quote; int32_t has_header; } gpuq_csv_options;   /* 0 = default ',' / '"' */
int gpuq_csv_decode(gpuq_ctx* ctx, void* stream, const uint8_t* text, int64_t n_bytes, const gpuq_field_info* file_fields, int n_file_fields,
                    const int32_t* projection, int n_projection, const gpuq_csv_options* options, gpuq_table** out);
int gpuq_parquet_decode(gpuq_ctx* ctx, void* stream, const uint8_t* file, int64_t n_bytes, const char* const* columns, int n_columns, gpuq_table** out);
/* the same over selected row groups (indices into the file's row groups, in the order given): predicate push-down stays where the
   reference has it -- ParquetExec prunes row groups from footer statistics on the host -- and only the surviving groups' chunks
   cross PCIe.  gpuq_parquet_row_groups: rows of every row group (host only), so that a caller can map its pruning to indices. */
int gpuq_parquet_decode_groups(gpuq_ctx* ctx, void* stream, const uint8_t* file, int64_t n_bytes, const char* const* columns, int n_columns,
                               const int32_t* row_groups, int n_row_groups, gpuq_table** out);
int gpuq_parquet_row_groups(const uint8_t* file, int64_t n_bytes, int64_t* rows_out, int cap, int* n_out);
/* host only: leaf columns (type = -1 where the device has no decoder) and row count from the footer */
int gpuq_parquet_schema(const uint8_t* file, int64_t n_bytes, gpuq_field_info* fields_out, int cap, int* n_out, int64_t* rows_out);
const char* gpuq_scan_last_error(void);

/* ---- compiled operators ------------------------------------------------------------------ */
/* Descriptor JSON (see INTEGRATION.md for the grammar).  Expression nodes mirror PhysicalExprNode
   (datafusion.proto:1142-1180): column, literal, binary_expr, cast, try_cast, not_expr, is_null_expr,
   is_not_null_expr, negative, in_list, case_ (like_expr is evaluated by gpuq_like_utf8 and lowered to a Boolean column by the plan
   executor).  "op" is one of
     "filter"      FilterExec        {input, predicate}
     "project"     ProjectionExec    {input, exprs:[{expr,name}]}
     "aggregate"   AggregateExec     {input, mode:Partial|Final|FinalPartitioned|Single, group_expr:[{expr,name}],
                                      aggr_expr:[{fn:SUM|AVG|COUNT|MIN|MAX, expr, name}], predicate?, strategy?:auto|tiny|hash|lds|radix, expected_groups?}
     "join_build"  HashJoinExec build (left) side  {input, on:[expr], predicate?, null_equals_null?,
                                                     build_side_rows?: false = gpuq_join_build_side_rows will not be called (Inner / Right / RightSemi / RightAnti)}
     "join_probe"  HashJoinExec probe (right) side {input, on:[expr], predicate?, join_type, null_equals_null?}
     "sort"        SortExec          {input, expr:[{expr, asc, nulls_first}], fetch?}
     "partition"   BatchPartitioner  {input, hash_expr:[expr], partition_count}
   "input" = {"fields":[{"name","type","nullable","side"}]}; side k>0 means the column is addressed
   through index vector k of the call (late materialisation after a filter or join). */
int gpuq_op_create(gpuq_ctx* ctx, const char* json, gpuq_op** out);
/* Host-only: compiles the descriptor without a device and writes a JSON description (program
   listing, output schema) to buf.  Used for plan validation and by the CPU test-suite. */
int gpuq_compile_check(const char* json, char* buf, size_t cap);
/* Host-only: the source the JIT path would hand to hiprtc for this descriptor and sink kernel id. */
int gpuq_compile_jit_source(const char* json, int kernel_id, char* buf, size_t cap);
void gpuq_op_free(gpuq_op* op);
int gpuq_op_num_outputs(gpuq_op* op);
int gpuq_op_output_field(gpuq_op* op, int i, gpuq_field_info* out);

/* Input of one operator call: columns in the order of the descriptor's input.fields.
   n_rows = number of driving positions: the table length, or the length of the index vectors when
   any are given (every index vector has n_rows entries; 0xFFFFFFFF = no row -> NULLs). */
typedef struct gpuq_input {
  const gpuq_column* cols;
  int32_t n_cols;
  int32_t n_via;
  int64_t n_rows;
  const uint32_t* via[3];
  /* Deferred execution (below): NULL, or a device u64 holding the ACTUAL number of driving positions; n_rows is then only an
     upper bound (it sizes grids and workspaces) and the kernels stop at min(n_rows, *n_rows_dev).  This is how one operator's
     count -- join pairs, filter survivors, groups -- reaches the next without a host round trip.  Honoured by gpuq_filter_run,
     gpuq_project_run, gpuq_aggregate_run_deferred, gpuq_join_build_run, gpuq_join_probe_run and gpuq_sort_run; every other
     entry point refuses an input that carries it (GPUQ_ERR_INVALID). */
  const uint64_t* n_rows_dev;
} gpuq_input;

/* FilterExec.  Writes the passing driving positions (or via[payload_via-1][pos] when payload_via>0),
   in input order, to sel_out (capacity n_rows) and their number to *count_out (device u64).
   Asynchronous.  NULL predicate results drop the row. */
int gpuq_filter_run(gpuq_op* op, void* stream, const gpuq_input* in, int payload_via, uint32_t* sel_out, uint64_t* count_out);

/* ProjectionExec / take: evaluates the expressions for every driving position into caller-allocated
   fixed-width columns (outs[i].data: n_rows * width bytes; outs[i].validity: ceil(n_rows/64)*8 bytes or
   NULL).  Asynchronous. */
int gpuq_project_run(gpuq_op* op, void* stream, const gpuq_input* in, gpuq_column* outs, int n_outs);

/* AggregateExec.  Strategy "auto" picks the kernel by cardinality: <= 64 groups an LDS dictionary with per-lane accumulators;
   a few hundred to a few thousand, block-local LDS tables flushed to a global one ("lds"); >= 4096 known groups on >= 2^20
   rows, radix partitioning + one LDS table per bucket ("radix"); otherwise / unknown a global hash table ("hash").  The
   group count is taken from expected_groups, else from this operator's previous run -- keep operators alive across the
   partitions of a stage.  outs: caller-allocated columns of capacity `cap` rows in output-schema order.
   Synchronous (the group count decides the strategy and the result length); *n_groups_out (host)
   receives the number of groups.  GPUQ_ERR_CAPACITY when cap is too small. */
int gpuq_aggregate_run(gpuq_op* op, void* stream, const gpuq_input* in, gpuq_column* outs, int n_outs, int64_t cap,
                       int64_t* n_groups_out);

/* HashJoinExec build side: inserts every passing row; the table keeps, per key, the chain of build
   row ids (payload = driving position, or via[payload_via-1][pos]).  build_rows_bound = exclusive
   upper bound of payload values.  Synchronous. */
int gpuq_join_build_run(gpuq_op* op, void* stream, const gpuq_input* in, int payload_via, int64_t build_rows_bound,
                        gpuq_join_table** out);
void gpuq_join_table_free(gpuq_join_table* t);
/* Chain fusion for (A |x| B) |x| C when A's keys are unique: the second join's build side is "the rows of B whose key is in A's
   table".  Instead of probing A with B, materialising the pairs and building from them through an index vector, build the second
   table straight from B: the operator's descriptor carries "semi_on" (B's key expressions for A, next to "on", B's key expressions
   for C), every row of `in` (= B) that passes the fused predicate is looked up in semi_table (= A's table, gpuq_join_build_run)
   and inserted when found -- one pass over B, the build row is B's position (payload_via must be 0).  semi_hits_out (device u32 per
   row of B, or NULL) receives A's row for every surviving position (A's columns are then read through it as an index vector);
   rows_out (device u64, or NULL) the number of surviving rows.  GPUQ_ERR_INVALID when semi_table holds duplicate keys
   (gpuq_join_table_has_duplicates): fall back to the two-step form.  Otherwise as gpuq_join_build_run. */
int gpuq_join_build_run_semi(gpuq_op* op, void* stream, const gpuq_input* in, int payload_via, int64_t build_rows_bound,
                             gpuq_join_table* semi_table, uint32_t* semi_hits_out, uint64_t* rows_out, gpuq_join_table** out);
int gpuq_join_table_has_duplicates(const gpuq_join_table* t);
/* Probe side.  Emits matching (build_row, probe_row) pairs (probe_row = position or via[..]) into
   out_build/out_probe (capacity out_cap pairs; out_build may be NULL for RightSemi/RightAnti) and the
   total pair count into *count_out (device u64).  Join types follow datafusion.proto:280-289; the
   build side is DataFusion's LEFT input.  For Left/Full/LeftSemi/LeftAnti the build rows reached are
   marked in the table; fetch them with gpuq_join_build_side_rows.  Asynchronous; call
   gpuq_op_check afterwards to learn about capacity overflow. */
int gpuq_join_probe_run(gpuq_op* op, void* stream, gpuq_join_table* t, const gpuq_input* in, int payload_via,
                        uint32_t* out_build, uint32_t* out_probe, uint64_t out_cap, uint64_t* count_out);
/* matched != 0: build rows with >= 1 match (LeftSemi); matched == 0: build rows never matched
   (Left/Full outer remainder, LeftAnti).  Ordered.  Asynchronous. */
int gpuq_join_build_side_rows(gpuq_join_table* t, void* stream, int matched, uint32_t* rows_out, uint64_t* count_out);

/* HashJoinExec with a JoinFilter on a non-inner join (datafusion.proto:1346-1360 `filter`): a pair that fails the filter does not
   count as a match.  The executors probe as Inner, filter the pairs, and derive the outer / semi / anti parts from the pairs
   that survive: gpuq_mark_rows sets bitmap bit rows[i] for every surviving pair (0xFFFFFFFF skipped; bitmap zeroed by the
   caller, 8-byte aligned, one bit per row of that side); the unmarked / marked rows are then selected with an ordinary filter
   over that Boolean column.  Asynchronous. */
int gpuq_mark_rows(gpuq_ctx* ctx, void* stream, const uint32_t* rows, int64_t n, uint8_t* bitmap);

/* CrossJoinExec (datafusion.proto:1382-1385; what DataFusion plans for an uncorrelated scalar subquery: q11 / q15 / q22's threshold is a
   one-row side): the row pairs of left x right, left-major -- left_rows_out[i] = i / n_right, right_rows_out[i] = i % n_right for
   i < n_left * n_right (< 2^32).  The executor reads both sides through these index vectors like a hash join's pairs.  Asynchronous. */
int gpuq_cross_pairs(gpuq_ctx* ctx, void* stream, int64_t n_left, int64_t n_right, uint32_t* left_rows_out, uint32_t* right_rows_out);

/* gpuq_sort_run that also hands back the FIRST key's column in sorted order when it can be rebuilt from the sorted records (one key of
   an integer-like type -- Int8..Int64, UInt*, Date32 / Date64, Timestamp, Decimal128 -- whose values span at most 2^63, more rows than one block
   sorts): key_data_out = n values of the key's width, key_validity_out = (n + 63) / 64 * 8 bytes (NULL when the key cannot be NULL).
   *decoded_out = 1: written (a sequential pass; the caller need not gather that column through the permutation), 0: not written. */
int gpuq_sort_run_keys(gpuq_op* op, void* stream, const gpuq_input* in, uint32_t* perm_out, void* key_data_out, uint8_t* key_validity_out, int* decoded_out);

/* SortExec: writes the permutation (driving positions in sorted order; stable) to perm_out (n_rows).
   The call reads the key range back (one stream synchronisation in the middle).  From 2^22 rows on the range comes from a row
   sample and the pack kernel verifies it; the call then waits for one word at the end and re-runs with the exact range when the
   guess did not hold (env GPUQ_SORT_SPECULATE=0: always the exact range, the permutation is the same either way). */
int gpuq_sort_run(gpuq_op* op, void* stream, const gpuq_input* in, uint32_t* perm_out);

/* Ordered fan-in -- CoalesceTasksExec with `order_by` / SortPreservingMergeExec (coalesce_tasks.rs:162-170 `streaming_merge`):
   `in` is the concatenation of n_runs runs, run r = rows [run_offsets[r], run_offsets[r+1]) (host array of n_runs + 1 entries,
   empty runs allowed), each already in the order of `op` (a "sort" operator).  Writes the permutation of the merged order; equal
   keys keep (run, row) order, i.e. the result equals the stable sort of the concatenation.  log2(n_runs) merge-path rounds on the
   device, one read + one write of the (key, row) records per round -- or, when the packed key is so narrow that its radix passes
   cost less than the rounds (many runs, <= 32 key bits), those passes: the permutation is the same.  Runs that are NOT sorted
   give an unspecified permutation through the rounds. */
int gpuq_merge_run(gpuq_op* op, void* stream, const gpuq_input* in, const int64_t* run_offsets, int n_runs, uint32_t* perm_out);

/* Hash repartition (BatchPartitioner::partition call site, shuffle_writer.rs:336-391):
   perm_out lists driving positions grouped by partition (input order inside a partition),
   part_offsets_out[p]..[p+1] (device u64, partition_count+1 entries) delimit partition p.
   partition = mix64-hash(keys) % partition_count -- gpuq's own function (ahash
   assignment is not a portable contract).  Asynchronous. */
int gpuq_partition_run(gpuq_op* op, void* stream, const gpuq_input* in, uint32_t* perm_out, uint64_t* part_offsets_out);

/* Deferred execution: one host round trip per plan instead of one or two per operator.
   An operator's synchronous entry points read small things back (a build side's key range, "did the table fill up", the number of
   groups, a sort key's range) because the next decision depends on them.  An operator that has completed a synchronous run
   REMEMBERS what it learned; with gpuq_op_set_deferred(op, 1) its next runs assume the same answers, queue all their kernels
   without waiting, let the kernels verify the assumptions (a key outside the remembered range, a table or an output that turned
   out too small, a row that does not fit the remembered sort-key layout raise bits in the operator's status word) and hand row
   counts on as device words (gpuq_input.n_rows_dev).  The caller then settles everything at once: gpuq_ops_settle waits for the
   stream ONCE, reads the status words of all the operators and any number of count words in one copy, and returns GPUQ_OK, or
   GPUQ_ERR_RETRY when an assumption did not hold -- the results of the deferred runs are then void and the caller runs the
   operators again synchronously (which refreshes what they remember).  gpuq_op_can_defer tells whether an operator has a
   completed synchronous run to go by; without one a deferred call simply runs synchronously.  The native plan executor drives
   its plans this way from their second execution on (csrc/plan_exec.cpp). */
int gpuq_op_set_deferred(gpuq_op* op, int on);
int gpuq_op_can_defer(gpuq_op* op);
/* gpuq_aggregate_run without the wait: *n_bound_out (host) = capacity bound of the result, *n_groups_dev_out = device u64 holding
   the actual group count (valid until the operator's next run); the result columns hold that many rows.  When the operator
   cannot defer (no completed synchronous run, or a strategy without a deferred form) this IS gpuq_aggregate_run: *n_bound_out is
   then exact and *n_groups_dev_out NULL. */
int gpuq_aggregate_run_deferred(gpuq_op* op, void* stream, const gpuq_input* in, gpuq_column* outs, int n_outs, int64_t cap,
                                int64_t* n_bound_out, const uint64_t** n_groups_dev_out);
/* One wait for everything deferred on `stream`: the status words of ops[0..n_ops) and the device u64 words[0..n_words) (copied to
   words_out, host).  GPUQ_ERR_RETRY when an operator's assumptions did not hold (it forgets them); other errors as gpuq_op_check. */
int gpuq_ops_settle(gpuq_ctx* ctx, void* stream, gpuq_op* const* ops, int n_ops, const uint64_t* const* words, int n_words, uint64_t* words_out);

/* Waits for `stream`, then reports device-side conditions raised by the op's kernels since the last
   check (string longer than 15 bytes in a packed comparison, output capacity overflow, ...). */
int gpuq_op_check(gpuq_op* op, void* stream);
/* Debug: the full source handed to hiprtc for sink kernel `kernel_id` (1 filter, 2 project, 3 aggregate-LDS,
   4 aggregate-hash, 5 join build, 6 probe chained, 7 probe unique, 8 sort min/max, 9 sort pack, 10 partition, 14 join key range). */
int gpuq_op_jit_source(gpuq_op* op, int kernel_id, char* buf, size_t cap);

/* arrow `take` for Utf8 payload columns of ANY string length (datafusion's materialisation of a join / filter / sort output):
   out[j] = col[idx[j]] (idx == NULL: identity; 0xFFFFFFFF or a NULL value -> NULL, length 0).  col must be Arrow layout
   (offsets + bytes).  offsets_out: n+1 int32; validity_out (optional, 8-byte aligned, ((n+63)/64)*8 bytes): written in full;
   data_out capacity data_cap bytes.  Two-call protocol: with data_cap too small (e.g. 0) the call computes offsets and
   validity, reports the required size in *data_len_out and returns GPUQ_ERR_CAPACITY; call again with a large enough
   buffer.  Synchronous up to the size read-back; the byte copy is queued on `stream`. */
int gpuq_take_utf8(gpuq_ctx* ctx, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, int32_t* offsets_out, uint8_t* validity_out,
                   uint8_t* data_out, int64_t data_cap, int64_t* data_len_out);

/* Utf8 values of ANY length as group-by / join keys (q10's c_name, q16's p_type): the expression
   programs carry strings as 16-byte integers (<= 15 bytes); longer key columns are replaced by exact dictionary codes first.
   gpuq_utf8_max_len: the longest value of an Arrow-layout Utf8 column (rows through `idx` when given) -- tells an executor
   whether the rewrite is needed.  gpuq_utf8_dict_create(capacity_rows = rows that will be inserted) / gpuq_utf8_intern:
   insert != 0 fills the dictionary from ONE column and returns every row's code = the row id (in that column) of the string's
   representative; insert == 0 looks rows of ANOTHER column up (a join's probe side).  codes_out: int64[n];
   validity_out: (n + 63) / 64 * 8 bytes, bit i = row i has a code (0: the value is NULL, or -- lookup -- not in the dictionary, which
   an equi-join treats the same way).  Equal strings get equal codes, different strings different ones, whatever they hash to: a
   tag match is verified byte for byte.  The string of a code comes back with gpuq_take_utf8(dictionary column, rows = codes). */
/* Utf8 values of any length as SORT keys: gpuq_utf8_sort_piece writes piece `piece` of every string (rows through `idx` when
   given) as a 16-byte integer that compares like the bytes [14 piece, 14 piece + 14) do -- the bytes big-endian and zero padded,
   times 256, plus the number of bytes the piece holds, so that "ab" < "ab\0".  Sorting stably by the last piece, then the one
   before, ... (ceil(max_len / 14) passes; the executors' SortExec does this after the packed 15-byte key has refused) yields
   the bytewise order of the strings, which is Arrow's.  keys_out: n x 16 bytes (values < 2^120: a Decimal128(38, 0) column);
   validity_out: (n + 63) / 64 * 8 bytes, bit i = row i is not NULL.  Asynchronous. */
int gpuq_utf8_sort_piece(gpuq_ctx* ctx, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, int piece, void* keys_out, uint8_t* validity_out);
typedef struct gpuq_utf8_dict gpuq_utf8_dict;
int gpuq_utf8_max_len(gpuq_ctx* ctx, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, int32_t* max_len_out);
int gpuq_utf8_dict_create(gpuq_ctx* ctx, void* stream, int64_t capacity_rows, gpuq_utf8_dict** out);
int gpuq_utf8_intern(gpuq_utf8_dict* dict, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, int insert, int64_t* codes_out, uint8_t* validity_out);
void gpuq_utf8_dict_free(gpuq_utf8_dict* dict);
/* an Int64 (or UInt32: codes are row ids) code column, with or without validity -> the row ids a take wants (NULL -> 0xFFFFFFFF) */
int gpuq_utf8_code_rows(gpuq_ctx* ctx, void* stream, const gpuq_column* codes, int64_t n, uint32_t* rows_out);

/* LikeExpr (PhysicalLikeExprNode, datafusion.proto:1240-1245: negated, case_insensitive, expr, pattern) of a Utf8 column in Arrow
   layout against a literal pattern: '%' any run of characters, '_' one character, "\\%" / "\\_" the literal characters (arrow-string's
   `like` / `nlike` with a scalar pattern).  idx (optional) addresses the column through an index vector; 0xFFFFFFFF or a NULL
   value gives NULL.  bits_out / validity_out: bitmaps of ((n+63)/64)*8 bytes, 8-byte aligned, written in full (validity_out may
   be NULL).  case_insensitive != 0 is GPUQ_ERR_UNSUPPORTED.  Asynchronous.  The result is a Boolean column an operator's
   expression can read like any other. */
int gpuq_like_utf8(gpuq_ctx* ctx, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, const char* pattern, int negated, int case_insensitive,
                   uint8_t* bits_out, uint8_t* validity_out);

/* A comparison of two Utf8 operands of ANY length -- the register programs hold 15 bytes: `<` between longer values, or `=` between two
   columns, is refused there and the executors lower the comparison to a Boolean column computed here, over the Arrow-layout bytes
   (arrow-ord's order: bytewise, a proper prefix first).  op: 0 = , 1 != , 2 < , 3 <= , 4 > , 5 >= .  a (through idx_a) against b (through
   idx_b) or, with b == NULL, against the `literal_len` bytes at `literal` (host memory).  A NULL on either side gives NULL.
   bits_out / validity_out as gpuq_like_utf8.  Asynchronous. */
int gpuq_utf8_compare(gpuq_ctx* ctx, void* stream, const gpuq_column* a, const uint32_t* idx_a, const gpuq_column* b, const uint32_t* idx_b, const char* literal,
                      int64_t literal_len, int64_t n, int op, uint8_t* bits_out, uint8_t* validity_out);

/* dst[i] = src[i] + delta for n Utf8 offsets: joining the offset arrays of partitions that are concatenated (fan-in). */
int gpuq_offsets_rebase(gpuq_ctx* ctx, void* stream, const int32_t* src, int64_t n, int32_t delta, int32_t* dst);

/* Utf8 PACKED15 -> Arrow offsets+bytes.  offsets_out: n+1 int32; data_out capacity data_cap bytes.
   Synchronous; *data_len_out (host) = bytes written. */
int gpuq_unpack_utf8(gpuq_ctx* ctx, void* stream, const void* packed, int64_t n, int32_t* offsets_out, uint8_t* data_out,
                     int64_t data_cap, int64_t* data_len_out);

/* Bitmap concatenation for CoalesceTasksExec / CoalescePartitionsExec / UnionExec fan-in (coalesce_tasks.rs:130-229:
   P partition streams -> 1): dst bits [dst_bit_offset, +n_bits) |= src bits [0, n_bits); src == NULL appends ones
   (a piece without a validity buffer).  dst (8-byte aligned, padded to a multiple of 8 bytes) must be zeroed first.
   Fixed-width data buffers are concatenated with plain device copies.  Asynchronous on `stream`. */
int gpuq_concat_bitmap(gpuq_ctx* ctx, void* stream, uint8_t* dst, int64_t dst_bit_offset, const uint8_t* src, int64_t n_bits);
/* Same with a source bit offset: dst bits [dst_bit_offset, +n_bits) |= src bits [src_bit_offset, +n_bits).  Row ranges of a
   bitmap (GlobalLimitExec skip, datafusion.proto:1453-1458). */
int gpuq_copy_bits(gpuq_ctx* ctx, void* stream, uint8_t* dst, int64_t dst_bit_offset, const uint8_t* src, int64_t src_bit_offset, int64_t n_bits);

/* ---- native plan executor ------------------------------------------------------------------
   The C++ host side above the operator calls: executes a physical plan tree, keeping data on the device between operators
   (fused Filter -> Projection -> consumer chains, index-vector views, pooled allocations).  Stands in for the walk an
   ExecutionEngine makes over the stage plan (ballista/executor/src/execution_engine.rs:34-60; `plan.execute(0, ctx)` at
   ballista/core/src/execution_plans/shuffle_writer.rs:255).  plan_json mirrors the PhysicalPlanNode messages of
   ballista/core/proto/datafusion.proto, one key per node:
     {"MemoryExec": {"schema": [{"name","type","nullable"}], "partitions": [input_slot, ...], "dense"?: bool, "sides"?: [..]}}
     {"FilterExec": {"input": node, "expr": expr}}                                   (:1291-1294)
     {"ProjectionExec": {"input": node, "expr": [expr], "expr_name": [str]}}         (:1399-1403)
     {"AggregateExec": {"input", "mode", "group_expr": [{"expr","name"}], "aggr_expr": [{"fn","expr","expr2"?,"name"}],
                        "strategy"?, "expected_groups"?, "output_capacity"?}}        (:1405-1450)
     {"HashJoinExec": {"left", "right", "on": [{"left": expr, "right": expr}], "join_type", "partition_mode",
                       "null_equals_null", "filter"?}}                               (:1346-1360)
     {"SortExec" | "SortPreservingMergeExec": {"input", "expr": [{"expr","asc","nulls_first"}], "fetch"?}}   (:1465-1478)
     {"CoalesceBatchesExec": {"input"}}  {"LocalLimitExec": {"input","fetch"}}       (:1487-1490, :1460-1463)
     {"GlobalLimitExec": {"input","skip","fetch"}}  {"UnionExec": {"inputs": [node]}}  {"CoalescePartitionsExec": {"input"}}   (:1453, :1319, :1492)
     {"CoalesceTasksExec": {"input","partitions": [p],"order_by"?: [sort expr]}}     (ballista coalesce_tasks.rs:46-70)
     {"ShuffleWriterExec": {"input","job_id","stage_id","work_dir","output_partitioning"?: {"hash_expr": [expr], "partition_count": n},
                            "batch_rows"?: rows per RecordBatch, default 2^20,
                            "partitions"?: [stage partitions of this task, shuffle_writer.rs:118-119: echoed by gpuq_plan_metrics
                                            for ShuffleWritePartition.partitions, the CoalesceTasksExec child does the work]}}
                                                                                   (ballista shuffle_writer.rs:234-456: the stage root)
         writes <work_dir>/<job_id>/<stage_id>/<q>/<uuid>.arrow per non-empty output partition q (unpartitioned:
         .../<stage_id>/<uuid>/data.arrow), Arrow IPC stream + LZ4_FRAME; result = one row per file:
         partition UInt32, path Utf8, num_rows / num_batches / num_bytes UInt64 (shuffle_writer.rs:470-520)
     {"RepartitionExec": {"input","hash_expr": [expr],"partition_count": ranks}}  {"BroadcastExec": {"input"}}
         the exchange between the GPUs of one node (gpuq_plan_set_comm; "exchange" section below): the stand-ins for a stage
         boundary of the reference -- RepartitionExec(Hash) is where planner.rs:137-151 splits stages; BroadcastExec is a
         CollectLeft build side read by every reduce task
     {"ShuffleReaderExec": {"schema": [{"name","type","nullable"}], "partition": [[{"path"} | path, ...], ...]}}   (shuffle_reader.rs:149-177;
         local files; a file that cannot be opened is reported as "FetchFailed: ...")
   Expressions are the PhysicalExprNode mirror of gpuq_op_create; columns are resolved by NAME against each operator's input.
   gpuq_plan_execute runs one output partition: inputs[k] is the table MemoryExec leaves refer to as slot k (caller-owned device
   memory, must stay valid until the call returns); *out is a materialised result owned by the library (gpuq_result_free).
   Synchronous.  Errors: status code + gpuq_plan_last_error(). */
typedef struct gpuq_plan gpuq_plan;
typedef struct gpuq_result gpuq_result;
int gpuq_plan_create(gpuq_ctx* ctx, const char* plan_json, gpuq_plan** out);
void gpuq_plan_free(gpuq_plan* plan);
int gpuq_plan_num_partitions(gpuq_plan* plan);
/* The plan's output schema BEFORE anything runs -- QueryStageExecutor::schema() (execution_engine.rs:59), which the executor
   reads to parse the stage's output partitioning ahead of execution (executor_server.rs:530-534).  Host only (no device work):
   expressions are typed by the same compiler that builds the operators.  Size query with fields_out == NULL && cap == 0;
   GPUQ_ERR_CAPACITY when cap is too small (*n_out set). */
int gpuq_plan_schema(gpuq_plan* plan, gpuq_field_info* fields_out, int cap, int* n_out);
int gpuq_plan_execute(gpuq_plan* plan, void* stream, int partition, const gpuq_input* inputs, int n_inputs, gpuq_result** out);
/* Asynchronous execution and cancellation.  The reference runs a task as a future on its task-runner pool and cancels it by
   dropping that future (ballista/executor/src/executor.rs:201-240; whatever holds resources cleans up in Drop).  Here
   gpuq_plan_execute_async starts the plan on a worker thread (same semantics as gpuq_plan_execute; the gpuq_input / gpuq_column
   arrays are copied, the device buffers they point to must stay valid until the task is done) and returns at once;
   gpuq_task_wait blocks and hands over the result (or the error: GPUQ_ERR_CANCELLED after a cancel that came in time);
   gpuq_task_cancel raises a flag the executor checks between operator calls -- the worker then drains `stream` and returns every
   pooled buffer it held, so the plan, the stream and the context are immediately usable again; gpuq_task_free on a running task
   cancels and joins it (the Drop of the Rust wrapper).  One task per plan at a time. */
typedef struct gpuq_task gpuq_task;
int gpuq_plan_execute_async(gpuq_plan* plan, void* stream, int partition, const gpuq_input* inputs, int n_inputs, gpuq_task** out);
int gpuq_task_poll(gpuq_task* task, int* done_out);
int gpuq_task_cancel(gpuq_task* task);
int gpuq_task_wait(gpuq_task* task, gpuq_result** out);
void gpuq_task_free(gpuq_task* task);
/* The ranks of the node for RepartitionExec / BroadcastExec nodes (the comm outlives the plan; NULL detaches). */
struct gpuq_comm;
int gpuq_plan_set_comm(gpuq_plan* plan, struct gpuq_comm* comm);
/* How the plan's last execution went: *deferred_out = 1 when it ran deferred (see "Deferred execution" above: from a plan's second
   execution on, env GPUQ_DEFER=0 switches it off) and held; *settles_out = host round trips that settled deferred operators (1 for
   a single-GPU plan without files); *host_syncs_out = count / status read-backs of the operators that ran synchronously;
   *retries_out = deferred executions of this plan that had to be redone synchronously so far.  Any pointer may be NULL. */
int gpuq_plan_exec_stats(gpuq_plan* plan, int* deferred_out, int* settles_out, int* host_syncs_out, int* retries_out);
int gpuq_plan_metrics(gpuq_plan* plan, char* json_out, size_t cap);     /* per node: output_rows, elapsed_compute (ns) -- utils.rs:470-481; ShuffleWriterExec adds write_time, repart_time, input_rows (shuffle_writer.rs:139-160) */
const char* gpuq_plan_last_error(void);
/* gpuq_op_profile over every operator the plan has compiled: enable/disable the HIP-event bracket around each operator's
   dominant kernel and report the operator with the most accumulated kernel time (its descriptor text in op_desc_out). */
int gpuq_plan_profile(gpuq_plan* plan, int enable, float* kernel_ms_out, int* launches_out, char* op_desc_out, size_t cap);
/* Every operator the plan has compiled, as a JSON array [{"op": kind, "kernel_ms": accumulated ms, "launches": n, "desc": descriptor}]
   (reads and resets the accumulators; profiling stays as it was set by gpuq_plan_profile).  GPUQ_ERR_CAPACITY when cap is too small. */
int gpuq_plan_profile_all(gpuq_plan* plan, char* json_out, size_t cap);
int64_t gpuq_result_num_rows(const gpuq_result* r);
int gpuq_result_num_columns(const gpuq_result* r);
int gpuq_result_column(const gpuq_result* r, int i, gpuq_column* col_out, gpuq_field_info* field_out);
void gpuq_result_free(gpuq_result* r);
/* When all columns of the result live in ONE device allocation of the fixed "record" layout (an aggregate's output: 256-byte
   header, then per column data and validity, 256-byte aligned, for *cap_out rows), returns it: such a result can be shipped
   between GPUs as it lies.  *base_out = NULL otherwise. */
int gpuq_result_record(const gpuq_result* r, void** base_out, size_t* bytes_out, int64_t* cap_out);

/* ---- shuffle codec: Arrow IPC stream messages with LZ4_FRAME buffers, device side -------------------
   Stands in for the reference's shuffle sink and source: `StreamWriter::try_new_with_options(.., LZ4_FRAME)` + `write(&batch)`
   (ballista/core/src/execution_plans/shuffle_writer.rs:365-378, ballista/core/src/utils.rs:179-219) and the IPC stream reader
   behind ShuffleReaderExec / the Flight client (ballista/core/src/execution_plans/shuffle_reader.rs, ballista/core/src/client.rs).
   A shuffle file is: the Schema message (the caller's Arrow library serialises it -- arrow-rs `IpcDataGenerator::schema_to_bytes`,
   pyarrow `Schema.serialize()`), RecordBatch messages, then the end-of-stream marker FF FF FF FF 00 00 00 00.  These entry points
   encode / decode ONE encapsulated RecordBatch message; the buffers are (de)compressed on the device (one wave per 64 KiB block),
   the host only frames.  Frames are written with independent 64 KiB blocks, no checksums (what lz4_flex writes for arrow-ipc);
   frames with linked blocks (Arrow C++), stored blocks, block checksums and content size are read.  ZSTD is GPUQ_ERR_UNSUPPORTED.
   Errors: status code + gpuq_ipc_last_error(). */
typedef struct gpuq_ipc_info {
  int32_t header_type;     /* Message.fbs MessageHeader: 1 Schema, 2 DictionaryBatch, 3 RecordBatch; 0 = end-of-stream marker */
  int32_t codec;           /* -1 none, 0 LZ4_FRAME, 1 ZSTD */
  int64_t metadata_bytes;  /* from the start of the message to the start of its body (8 + flatbuffer incl. padding) */
  int64_t body_bytes;
  int64_t n_rows;
  int32_t n_nodes, n_buffers;
} gpuq_ipc_info;
typedef struct gpuq_ipc_batch gpuq_ipc_batch;
/* Host only: parses the message that starts at `bytes`.  GPUQ_ERR_CAPACITY when `avail` does not cover the metadata yet
   (out->metadata_bytes is set once the first 8 bytes are there). */
int gpuq_ipc_peek(const uint8_t* bytes, int64_t avail, gpuq_ipc_info* out);
/* Host only: the encapsulated Schema message that opens a stream of these fields (name, type, precision, scale, nullable are
   read) -- what `IpcDataGenerator::schema_to_bytes` / `Schema.serialize()` produce.  Size query with out == NULL && cap == 0. */
int gpuq_ipc_schema_message(const gpuq_field_info* fields, int n_cols, uint8_t* out, int64_t cap, int64_t* len_out);
/* Host only: the same with Schema.custom_metadata key / value pairs (what arrow-rs writes for `Schema::metadata`). */
int gpuq_ipc_schema_message_kv(const gpuq_field_info* fields, int n_cols, const char* const* keys, const char* const* values, int n_kv, uint8_t* out, int64_t cap,
                               int64_t* len_out);
/* Host only: looks `key` up in the custom_metadata of the Schema message that starts at `bytes`; *found_out = 0 / 1. */
int gpuq_ipc_schema_metadata(const uint8_t* bytes, int64_t avail, const char* key, char* value_out, size_t cap, int* found_out);
/* The partition function of a hash-partitioned shuffle file written by this engine, recorded as schema metadata
   "gpuq.partition_fn" (files written by DataFusion's BatchPartitioner -- fixed-seed ahash -- carry no such key).  Equal keys only
   meet in one output partition when EVERY map task of a stage used the same function: ShuffleReaderExec refuses a partition
   whose files disagree (all map tasks of a hash-partitioned stage must run on the same engine). */
#define GPUQ_PARTITION_FN_KEY "gpuq.partition_fn"
#define GPUQ_PARTITION_FN "gpuq-mix64-v1"
/* cols: device columns in GPUQ_REPR_ARROW layout, all of n_rows rows.  codec: 0 = LZ4_FRAME, -1 = uncompressed.  Writes the
   message to out_host (host memory, cap bytes) and its length to *len_out.  out_host == NULL && cap == 0: size query (the
   compression runs, nothing is copied).  GPUQ_ERR_CAPACITY (with *len_out set) when cap is too small.  Synchronous. */
int gpuq_ipc_encode_batch(gpuq_ctx* ctx, void* stream, const gpuq_column* cols, int n_cols, int64_t n_rows, int codec, uint8_t* out_host, int64_t cap,
                          int64_t* len_out);
/* msg: one whole encapsulated RecordBatch message in host memory; fields: the stream's schema (type / precision / scale are
   read).  The decoded columns live in device memory owned by *out (gpuq_ipc_batch_free).  Synchronous. */
int gpuq_ipc_decode_batch(gpuq_ctx* ctx, void* stream, const uint8_t* msg, int64_t msg_bytes, const gpuq_field_info* fields, int n_cols, gpuq_ipc_batch** out);
/* The same for a whole byte range of encapsulated messages (a shuffle file after -- or including -- its Schema message, up to
   the end-of-stream marker or n_bytes): every RecordBatch is decoded in ONE launch (a unit of work per LZ4 block of every
   buffer of every batch) straight into the concatenated columns.  This is the call for files written with the reference's
   8192-row batches, whose buffers hold one or two blocks each. */
int gpuq_ipc_decode_stream(gpuq_ctx* ctx, void* stream, const uint8_t* bytes, int64_t n_bytes, const gpuq_field_info* fields, int n_cols, gpuq_ipc_batch** out);
int64_t gpuq_ipc_batch_num_rows(const gpuq_ipc_batch* b);
int gpuq_ipc_batch_num_columns(const gpuq_ipc_batch* b);
int gpuq_ipc_batch_column(const gpuq_ipc_batch* b, int i, gpuq_column* col_out);
void gpuq_ipc_batch_free(gpuq_ipc_batch* b);
const char* gpuq_ipc_last_error(void);

/* ---- exchange between the GPUs of one node ---------------------------------------------------------
   Stands in for a stage boundary of the reference between executors that each own one GPU of a node: ShuffleWriterExec
   hash-partitioning into Arrow-IPC files (ballista/core/src/execution_plans/shuffle_writer.rs:328-392) that the next stage's
   ShuffleReaderExec fetches (shuffle_reader.rs:226-298).  Here it is one exchange step: an all-to-all of per-destination counts,
   then one variable-size all-to-all per column buffer, posted as grouped RCCL sends / receives so that all xGMI links of a GPU
   carry traffic at once; nothing is compressed or written to disk.  One process per GPU; every rank makes the same calls in
   the same order (they are collectives).  Errors: status code + gpuq_exchange_last_error(). */
typedef struct gpuq_comm gpuq_comm;
#define GPUQ_COMM_ID_BYTES 128
/* Rank 0 creates the id; the host hands it to the other ranks over its own channel (the executors' gRPC, a torch.distributed
   store, a file), then every rank calls gpuq_comm_create (collective: ncclCommInitRank on the context's device). */
int gpuq_comm_unique_id(uint8_t* id_out /* GPUQ_COMM_ID_BYTES */);
int gpuq_comm_create(gpuq_ctx* ctx, const uint8_t* id /* GPUQ_COMM_ID_BYTES */, int rank, int world, gpuq_comm** out);
/* Host-staged transport instead of RCCL: the library stages every buffer through pinned host memory and the caller moves the
   bytes (tests with several ranks on one GPU; a host that ships bytes over its own network service).  all_to_all_v: this rank
   sends send_counts[d] bytes to rank d from `send` (host memory, packed in rank order) and receives recv_counts[s] bytes from
   rank s into `recv` (packed in rank order); returns 0 on success.  Collective. */
typedef struct gpuq_transport {
  void* user;
  int (*all_to_all_v)(void* user, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts, int world);
} gpuq_transport;
int gpuq_comm_create_host(gpuq_ctx* ctx, const gpuq_transport* transport, int rank, int world, gpuq_comm** out);
void gpuq_comm_free(gpuq_comm* comm);
/* Failing together.  Every exchange starts with a fixed-size meta round that carries, next to the row counts, a status word per
   rank: 0 fine, 1 "I have failed", 2 "my deferred execution did not hold" (GPUQ_ERR_RETRY on every rank: all of them run again,
   synchronously).  gpuq_comm_set_status arms the word for this rank's NEXT collective call; gpuq_comm_announce is that meta round
   alone (no table): a rank whose plan failed below an exchange calls set_status(1) + announce so that the peers waiting in the
   exchange return GPUQ_ERR_PEER instead of hanging, and with status 0 everywhere it is the barrier that closes a deferred execution
   ("nobody has to redo anything").  A second, one-word round agrees on limits the counts imply (2^32 rows / 2 GiB of strings on a
   rank) before any payload moves; a failure inside the payload round aborts the communicator (ncclCommAbort).  Collective. */
/* The rows an exchanged table received from rank 0, 1, ... (it holds them in this order): the run boundaries of an ordered fan-in
   (gpuq_merge_run) after a range exchange.  n_out = number of ranks; rows_out may be NULL to ask for it. */
int gpuq_table_piece_rows(const gpuq_table* t, int64_t* rows_out, int cap, int* n_out);
int gpuq_comm_set_status(gpuq_comm* comm, int status);
int gpuq_comm_announce(gpuq_comm* comm, void* stream);
int gpuq_comm_rank(const gpuq_comm* comm);
int gpuq_comm_world(const gpuq_comm* comm);
/* Hash-repartition exchange.  cols / fields: device columns (Arrow layout, Utf8 of any length included, or PACKED15) whose rows
   are GROUPED BY DESTINATION: rows [dest_offsets[d], dest_offsets[d+1]) go to rank d (host array of world+1 entries; produce
   the grouping with gpuq_partition_run(partition_count = world) + a take, as the native executor's RepartitionExec does).
   *out = the rows every rank sent here, concatenated in rank order, in buffers the library owns (gpuq_table_free).
   fields[i].nullable decides whether a validity bitmap travels (the same on every rank).  Synchronous. */
int gpuq_exchange_partitions(gpuq_comm* comm, void* stream, const gpuq_column* cols, const gpuq_field_info* fields, int n_cols, const int64_t* dest_offsets,
                             gpuq_table** out);
/* Every rank receives all ranks' rows, in rank order: CollectLeft build sides (the reference's reduce task reads every partition
   of the build stage), partial-aggregate states. */
int gpuq_allgather_table(gpuq_comm* comm, void* stream, const gpuq_column* cols, const gpuq_field_info* fields, int n_cols, int64_t n_rows, gpuq_table** out);
const char* gpuq_exchange_last_error(void);

/* ---- timing support for bench.py: HIP events on the caller's stream ---------------------- */
typedef struct gpuq_timer gpuq_timer;
int gpuq_timer_create(gpuq_ctx* ctx, gpuq_timer** out);
int gpuq_timer_start(gpuq_timer* t, void* stream);
int gpuq_timer_stop(gpuq_timer* t, void* stream);
int gpuq_timer_elapsed_ms(gpuq_timer* t, float* ms_out); /* synchronises on the stop event */
void gpuq_timer_free(gpuq_timer* t);
/* Per-op device time of the last call(s): the op brackets its dominant kernel with HIP events when
   enabled; returns the accumulated ms and launch count since the last reset.  enable < 0 keeps the current setting. */
int gpuq_op_profile(gpuq_op* op, int enable, float* kernel_ms_out, int* launches_out);
/* ... and of EVERYTHING the op's run calls queued in the interval the last gpuq_op_profile call closed (the dominant kernel plus table
   initialisation, scans, compactions, the result projection): the operator's share of a step. */
int gpuq_op_profile_total(gpuq_op* op, float* total_ms_out);

#ifdef __cplusplus
}
#endif
#endif /* GPUQ_H */
