// Scan-side decode, host part: the leaves of the reference's TPC-H plans -- CsvExec / ParquetExec
// over the files benchmarks/src/bin/tpch.rs:801-862 registers -- produce their columns in HBM.  The host frames (line count
// read-back, Thrift metadata of a Parquet file: footer + page headers) and never touches values; kernels_scanfmt.hip parses.
#include "gpuq_internal.h"
#include "gpuq_kernels.h"
#include "expr_compile.h"
#include <cstring>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

using namespace gpuq;

namespace {
thread_local std::string g_ferr;
template <class F> int guarded_f(F&& f) {
  try { f(); return GPUQ_OK; }
  catch (const HipError& e) { g_ferr = e.what(); return GPUQ_ERR_HIP; }
  catch (const Unsupported& e) { g_ferr = e.what(); return GPUQ_ERR_UNSUPPORTED; }
  catch (const Capacity& e) { g_ferr = e.what(); return GPUQ_ERR_CAPACITY; }
  catch (const std::exception& e) { g_ferr = e.what(); return GPUQ_ERR_INVALID; }
}
// File bytes are pageable host memory: one thread staging them through a pinned buffer moves ~6-10 GB/s, a sixth of the link.
// Bulk uploads therefore go through a few staging lanes (thread + stream + two pinned slots each, kept for the process lifetime):
// lanes pull 4 MiB chunks off a shared counter, memcpy into a free slot and queue the DMA, so memcpys and DMAs of all lanes overlap.
struct Uploader {
  static constexpr size_t SLOT = (size_t)4 << 20;
  static constexpr int LANES = 8, SLOTS = 2;
  struct Lane { void* pin[SLOTS] = {}; hipEvent_t ev[SLOTS] = {}; hipStream_t st = nullptr; int dev = -1; };
  Lane lanes[LANES]; std::mutex mu;
  void prepare(Lane& L, int device) {
    if (L.dev == device) return;
    if (L.dev >= 0) throw Unsupported("scan decode: bulk uploads of one process go to one device");
    HIPCHECK(hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking));
    for (int k = 0; k < SLOTS; ++k) { HIPCHECK(hipHostMalloc(&L.pin[k], SLOT, hipHostMallocDefault)); HIPCHECK(hipEventCreateWithFlags(&L.ev[k], hipEventDisableTiming)); }
    L.dev = device;
  }
  struct Seg { size_t dst_off; const uint8_t* src; size_t n; };      // one byte range of the source -> dst + dst_off
  void copy(int device, hipStream_t s, void* dst, const void* src, size_t n) { copy_segments(device, s, dst, {Seg{0, (const uint8_t*)src, n}}); }
  void copy_segments(int device, hipStream_t s, void* dst_base, const std::vector<Seg>& segs) {
    struct Chunk { size_t dst_off; const uint8_t* src; size_t n; };
    std::vector<Chunk> chunks; size_t total = 0;
    for (auto& g : segs) { total += g.n; for (size_t o = 0; o < g.n; o += SLOT) chunks.push_back({g.dst_off + o, g.src + o, std::min(SLOT, g.n - o)}); }
    if (total < 2 * SLOT) {
      for (auto& c : chunks) HIPCHECK(hipMemcpyAsync((char*)dst_base + c.dst_off, c.src, c.n, hipMemcpyHostToDevice, s));
      HIPCHECK(hipStreamSynchronize(s)); return;
    }
    HIPCHECK(hipStreamSynchronize(s));      // earlier users of the (pooled) destination on `s` are done before other streams write it
    std::lock_guard<std::mutex> g(mu);
    const size_t nchunks = chunks.size();
    const int nt = (int)std::min<size_t>(LANES, nchunks);
    std::atomic<size_t> next{0}; std::mutex emu; std::string err;
    auto work = [&](int t) {
      try {
        HIPCHECK(hipSetDevice(device));
        Lane& L = lanes[t]; prepare(L, device);
        bool used[SLOTS] = {}; int k = 0;
        for (;;) {
          const size_t c = next.fetch_add(1); if (c >= nchunks) break;
          const Chunk& ch = chunks[c];
          const int sl = k++ % SLOTS;
          if (used[sl]) HIPCHECK(hipEventSynchronize(L.ev[sl]));
          std::memcpy(L.pin[sl], ch.src, ch.n);
          HIPCHECK(hipMemcpyAsync((char*)dst_base + ch.dst_off, L.pin[sl], ch.n, hipMemcpyHostToDevice, L.st));
          HIPCHECK(hipEventRecord(L.ev[sl], L.st)); used[sl] = true;
        }
        HIPCHECK(hipStreamSynchronize(L.st));
      } catch (const std::exception& e) { std::lock_guard<std::mutex> g2(emu); err = e.what(); next.store(nchunks); }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (!err.empty()) throw HipError("scan decode: upload failed: " + err);
  }
};
Uploader g_uploader;
void h2d(gpuq_ctx* ctx, hipStream_t s, void* dst, const void* src, size_t n) {
  if (n) g_uploader.copy(ctx->device, s, dst, src, n);
}
template <class T> T d2h_value(hipStream_t s, const T* dev) { T v; HIPCHECK(hipMemcpyAsync(&v, dev, sizeof(T), hipMemcpyDeviceToHost, s)); HIPCHECK(hipStreamSynchronize(s)); return v; }
gpuq_field_info field_of(const std::string& name, int type, int p, int sc, bool nullable) {
  gpuq_field_info f{}; std::snprintf(f.name, sizeof(f.name), "%s", name.c_str());
  f.type = type; f.precision = p; f.scale = sc; f.nullable = nullable; f.repr = GPUQ_REPR_ARROW;
  DType dt; dt.id = type; dt.p = p; dt.s = sc; f.width = type == T_BOOL ? 0 : type_width(dt);
  return f;
}

// lengths (int32[n], entry n is scratch) -> offsets in place; bytes copied by `copy`; fills the ImportedCol
void finish_strings(hipStream_t s, ImportedCol& ic, int32_t* lens_then_offsets, int64_t n, const std::function<void(const int32_t*, uint8_t*, int64_t)>& copy) {
  DevBuf sws; const size_t swb = exclusive_scan_ws_bytes(n + 1); sws.ensure(swb);
  launch_exclusive_scan_i32(s, lens_then_offsets, n, sws.p, swb);
  const int32_t total = d2h_value(s, lens_then_offsets + n);
  if (total < 0) throw Unsupported("scan decode: a Utf8 column exceeds 2 GiB (int32 offsets)");
  ic.data.ensure((size_t)total + 16);
  copy(lens_then_offsets, (uint8_t*)ic.data.p, (int64_t)total);
  ic.col.data = ic.data.p; ic.col.offsets = lens_then_offsets;
  HIPCHECK(hipStreamSynchronize(s));      // the scan workspace dies here
}
}  // namespace

extern "C" {

const char* gpuq_scan_last_error(void) { return g_ferr.c_str(); }

// ------------------------------------------------------------------ delimited text
int gpuq_csv_decode(gpuq_ctx* ctx, void* stream, const uint8_t* text, int64_t n_bytes, const gpuq_field_info* file_fields, int n_file_fields, const int32_t* projection,
                    int n_proj, const gpuq_csv_options* opt, gpuq_table** out) {
  if (out) *out = nullptr;
  return guarded_f([&]() {
    if (!ctx || !out || !file_fields || (n_bytes > 0 && !text)) throw std::runtime_error("ctx / text / file_fields / out is NULL");
    if (n_file_fields < 1 || n_file_fields > CSV_MAX_FIELDS) throw Unsupported("csv: 1.." + std::to_string(CSV_MAX_FIELDS) + " fields per line");
    if (n_bytes >= (1ll << 32)) throw Unsupported("csv: decode at most 4 GiB of text per call (split the file at line boundaries)");
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t s = use_stream(stream);
    const uint8_t delim = opt && opt->delimiter ? (uint8_t)opt->delimiter : (uint8_t)',';
    const uint8_t quote = opt && opt->quote ? (uint8_t)opt->quote : (uint8_t)'"';
    const bool header = opt && opt->has_header;
    std::vector<int32_t> proj;
    if (projection) proj.assign(projection, projection + n_proj); else for (int i = 0; i < n_file_fields; ++i) proj.push_back(i);
    CsvSpec S{}; S.n_fields = n_file_fields; S.delim = delim; S.quote = quote;
    for (int f = 0; f < n_file_fields; ++f) S.kind[f] = CSV_SKIP;
    for (size_t o = 0; o < proj.size(); ++o) {
      const int f = proj[o];
      if (f < 0 || f >= n_file_fields) throw std::runtime_error("csv: projection index out of range");
      if (S.kind[f] != CSV_SKIP) throw Unsupported("csv: a file column may be projected once");
      const gpuq_field_info& fi = file_fields[f];
      int k;
      switch (fi.type) {
        case T_INT32: k = CSV_I32; break; case T_INT64: k = CSV_I64; break; case T_DATE32: k = CSV_DATE32; break; case T_DECIMAL128: k = CSV_DEC128; break;
        case T_FLOAT64: k = CSV_F64; break; case T_BOOL: k = CSV_BOOL; break; case T_UTF8: k = CSV_UTF8; break;
        default: throw Unsupported("csv: column type " + std::to_string(fi.type));
      }
      S.kind[f] = k; S.out[f] = (int32_t)o; S.scale[f] = fi.scale; S.nullable[f] = fi.nullable; S.prec[f] = fi.type == T_DECIMAL128 ? std::min(std::max(fi.precision, 1), 38) : 0;
      if (fi.type == T_DECIMAL128 && (fi.scale < 0 || fi.scale > 38)) throw std::runtime_error("csv: Decimal128 scale out of range");
    }
    // text on the device (+ one '\n' so that an unterminated last line ends like the others)
    DevBuf dtext; dtext.ensure((size_t)n_bytes + 64);
    h2d(ctx, s, dtext.p, text, (size_t)n_bytes);
    const bool unterminated = n_bytes > 0 && text[n_bytes - 1] != '\n';
    if (unterminated) { const char nl = '\n'; HIPCHECK(hipMemcpyAsync((char*)dtext.p + n_bytes, &nl, 1, hipMemcpyHostToDevice, s)); }
    const int64_t nb = n_bytes + (unterminated ? 1 : 0);
    // line starts
    const i64 chunk = 1 << 16;
    const int nblocks = (int)std::max<i64>(1, (nb + chunk - 1) / chunk);
    DevBuf counts; counts.ensure((size_t)nblocks * 4 + 32);
    u64* total_dev = (u64*)((char*)counts.p + (((size_t)nblocks * 4 + 15) & ~(size_t)15));
    launch_csv_count_lines(s, (const uint8_t*)dtext.p, nb, chunk, nblocks, (uint32_t*)counts.p);
    launch_scan_block_counts(s, (uint32_t*)counts.p, nblocks, total_dev);
    const i64 n_lines = (i64)d2h_value(s, total_dev);
    DevBuf starts; starts.ensure((size_t)(n_lines + 2) * 8);
    HIPCHECK(hipMemsetAsync(starts.p, 0, 8, s));
    launch_csv_line_starts(s, (const uint8_t*)dtext.p, nb, chunk, nblocks, (const uint32_t*)counts.p, (i64*)starts.p);
    const i64 row0 = header ? 1 : 0;
    const i64 n_rows = std::max<i64>(0, n_lines - row0);
    if (n_rows > 0xFFFFFFFEll) throw Unsupported("csv: more than 2^32-2 rows per call");
    // outputs
    std::unique_ptr<gpuq_table> t(new gpuq_table()); t->ctx = ctx; t->n_rows = n_rows;
    CsvOut O{};
    std::vector<std::unique_ptr<DevBuf>> tmp;
    const size_t bm = (size_t)((n_rows + 63) / 64) * 8 + 16;
    for (size_t o = 0; o < proj.size(); ++o) {
      const gpuq_field_info& fi = file_fields[proj[o]];
      std::unique_ptr<ImportedCol> ic(new ImportedCol());
      ic->field = field_of(fi.name, fi.type, fi.precision, fi.scale, fi.nullable != 0);
      ic->col.type = fi.type; ic->col.precision = fi.precision; ic->col.scale = fi.scale; ic->col.repr = GPUQ_REPR_ARROW; ic->col.length = n_rows;
      if (fi.type == T_UTF8) {
        ic->offsets.ensure((size_t)(n_rows + 2) * 4 + 16);
        tmp.push_back(std::make_unique<DevBuf>()); tmp.back()->ensure((size_t)std::max<i64>(n_rows, 1) * 4);
        O.str_len[o] = (int32_t*)ic->offsets.p; O.str_start[o] = (uint32_t*)tmp.back()->p;
      } else if (fi.type == T_BOOL) { ic->data.ensure(bm); HIPCHECK(hipMemsetAsync(ic->data.p, 0, bm, s)); O.data[o] = ic->data.p; ic->col.data = ic->data.p; }
      else { DType dt; dt.id = fi.type; dt.p = fi.precision; dt.s = fi.scale; ic->data.ensure((size_t)std::max<i64>(n_rows, 1) * (size_t)type_width(dt) + 16); O.data[o] = ic->data.p; ic->col.data = ic->data.p; }
      if (fi.nullable) { ic->validity.ensure(bm); HIPCHECK(hipMemsetAsync(ic->validity.p, 0, bm, s)); O.valid[o] = (u64*)ic->validity.p; ic->col.validity = (const uint8_t*)ic->validity.p; }
      t->cols.push_back(std::move(ic));
    }
    DevBuf flags; flags.ensure(16); HIPCHECK(hipMemsetAsync(flags.p, 0, 16, s));
    launch_csv_parse(s, (const uint8_t*)dtext.p, nb, (const i64*)starts.p, row0, n_rows, S, O, (uint32_t*)flags.p);
    HIPCHECK(hipGetLastError());
    const uint32_t fl = d2h_value(s, (const uint32_t*)flags.p);
    if (fl & CSVF_QUOTE) throw Unsupported("csv: quoted fields are not parsed on the device");
    if (fl & CSVF_FIELD_COUNT) throw std::runtime_error("csv: a line does not have the schema's number of fields");
    if (fl & CSVF_BAD_NUMBER) throw std::runtime_error("csv: a field does not parse as its column's type");
    if (fl & CSVF_NULL_IN_REQUIRED) throw std::runtime_error("csv: an empty field in a non-nullable column");
    if (fl & CSVF_FLOAT_PRECISION) throw Unsupported("csv: a Float64 field needs more than the exact fast path (> 15 significant digits or |exponent| > 22)");
    size_t ti = 0;
    for (size_t o = 0; o < proj.size(); ++o) {
      if (file_fields[proj[o]].type != T_UTF8) continue;
      ImportedCol& ic = *t->cols[o];
      const uint32_t* st = (const uint32_t*)tmp[ti++]->p;
      const uint8_t* tx = (const uint8_t*)dtext.p;
      finish_strings(s, ic, (int32_t*)ic.offsets.p, n_rows, [&](const int32_t* offs, uint8_t* dst, int64_t) { launch_csv_copy_strings(s, tx, st, offs, n_rows, dst); });
    }
    HIPCHECK(hipStreamSynchronize(s));
    *out = t.release();
  });
}

}  // extern "C"

// ------------------------------------------------------------------ Parquet: Thrift compact protocol, the parts the format uses
namespace {
struct TReader {
  const uint8_t* p; const uint8_t* e;
  void need(size_t n) const { if ((size_t)(e - p) < n) throw std::runtime_error("parquet: metadata ends inside a Thrift value"); }
  uint64_t varint() { uint64_t v = 0; int sh = 0; for (;;) { need(1); const uint8_t c = *p++; v |= (uint64_t)(c & 0x7F) << sh; if (!(c & 0x80)) return v; sh += 7; if (sh > 63) throw std::runtime_error("parquet: varint too long"); } }
  int64_t zigzag() { const uint64_t v = varint(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
  std::string binary() { const uint64_t n = varint(); need(n); std::string s((const char*)p, (size_t)n); p += n; return s; }
  // field header: returns false at STOP; type in the low nibble, id by delta or explicit
  bool field(int& type, int& id, int& last) {
    need(1); const uint8_t b = *p++;
    if (b == 0) return false;
    type = b & 0x0F; const int delta = b >> 4;
    id = delta ? last + delta : (int)zigzag();
    last = id; return true;
  }
  int depth = 0;      // nesting of the value being skipped: a crafted footer of struct headers must not walk the host stack down
  struct Nest { int& d; Nest(int& d_) : d(d_) { if (++d > 64) throw std::runtime_error("parquet: Thrift metadata nested deeper than 64 levels"); } ~Nest() { --d; } };
  void skip(int type) {
    Nest nest(depth);
    switch (type) {
      case 1: case 2: break;                                   // bool true / false (value in the type)
      case 3: need(1); ++p; break;                             // i8
      case 4: case 5: case 6: (void)zigzag(); break;           // i16 / i32 / i64
      case 7: need(8); p += 8; break;                          // double
      case 8: { const uint64_t n = varint(); need(n); p += n; break; }     // binary
      case 9: case 10: { need(1); const uint8_t h = *p++; uint64_t n = h >> 4; const int et = h & 0x0F; if (n == 15) n = varint(); for (uint64_t i = 0; i < n; ++i) skip_elem(et); break; }
      case 11: { const uint64_t n = varint(); if (n) { need(1); const uint8_t kv = *p++; for (uint64_t i = 0; i < n; ++i) { skip_elem(kv >> 4); skip_elem(kv & 0x0F); } } break; }
      case 12: { int t, id, last = 0; while (field(t, id, last)) skip(t); break; }
      default: throw std::runtime_error("parquet: unknown Thrift type " + std::to_string(type));
    }
  }
  void skip_elem(int et) { if (et == 1 || et == 2) { need(1); ++p; } else skip(et); }      // booleans inside collections take a byte
  uint64_t list_header(int& elem_type) { need(1); const uint8_t h = *p++; uint64_t n = h >> 4; elem_type = h & 0x0F; if (n == 15) n = varint(); return n; }
};

struct PqSchemaElem { int type = -1, type_length = 0, repetition = 0, num_children = 0, converted = -1, scale = 0, precision = 0; std::string name; bool logical_decimal = false, logical_date = false, logical_string = false;
                      int int_bits = 0; bool int_signed = true; int ts_unit = -1; };      // LogicalType INTEGER {bitWidth, isSigned}, TIMESTAMP {unit}: gpuq TimeUnit 1 ms / 2 us / 3 ns
struct PqChunk { int type = -1, codec = 0; int64_t num_values = 0, total_compressed = 0, data_page_offset = 0, dict_page_offset = -1; std::vector<std::string> path; };
struct PqRowGroup { std::vector<PqChunk> cols; int64_t num_rows = 0; };
struct PqFileMeta { std::vector<PqSchemaElem> schema; std::vector<PqRowGroup> groups; int64_t num_rows = 0; };

PqSchemaElem read_schema_elem(TReader& r) {
  PqSchemaElem s; int t, id, last = 0;
  while (r.field(t, id, last)) {
    switch (id) {
      case 1: s.type = (int)r.zigzag(); break; case 2: s.type_length = (int)r.zigzag(); break; case 3: s.repetition = (int)r.zigzag(); break;
      case 4: s.name = r.binary(); break; case 5: s.num_children = (int)r.zigzag(); break; case 6: s.converted = (int)r.zigzag(); break;
      case 7: s.scale = (int)r.zigzag(); break; case 8: s.precision = (int)r.zigzag(); break;
      case 10: {      // LogicalType union: field id = which
        int t2, id2, last2 = 0;
        while (r.field(t2, id2, last2)) {
          if (id2 == 1) s.logical_string = true; else if (id2 == 6) s.logical_date = true;
          if (id2 == 5) {      // DecimalType {1 scale, 2 precision}
            s.logical_decimal = true; int t3, id3, last3 = 0;
            while (r.field(t3, id3, last3)) { if (id3 == 1) s.scale = (int)r.zigzag(); else if (id3 == 2) s.precision = (int)r.zigzag(); else r.skip(t3); }
          } else if (id2 == 10) {      // IntType {1 bitWidth: i8, 2 isSigned: bool (in the field header)}
            int t3, id3, last3 = 0;
            while (r.field(t3, id3, last3)) { if (id3 == 1 && t3 == 3) { r.need(1); s.int_bits = (int)(int8_t)*r.p++; } else if (id3 == 2 && (t3 == 1 || t3 == 2)) s.int_signed = t3 == 1; else r.skip(t3); }
          } else if (id2 == 8) {      // TimestampType {1 isAdjustedToUTC, 2 unit: union {1 MILLIS, 2 MICROS, 3 NANOS}}
            int t3, id3, last3 = 0;
            while (r.field(t3, id3, last3)) {
              if (id3 == 2 && t3 == 12) { int t4, id4, last4 = 0; while (r.field(t4, id4, last4)) { if (id4 >= 1 && id4 <= 3) s.ts_unit = id4; r.skip(t4); } }
              else r.skip(t3);
            }
          } else r.skip(t2);
        }
        break;
      }
      default: r.skip(t);
    }
  }
  return s;
}
// parquet.thrift Type / ConvertedType / LogicalType of a leaf -> gpuq_type (+ precision / scale / bytes per value on the device); false = not read
// on the device.  INT96 is Impala's timestamp (nanoseconds of the day + Julian day) -> Timestamp(Nanosecond), as arrow's reader maps it.
bool pq_leaf_type(const PqSchemaElem& L, int& gt, int& gp, int& gs, int& width) {
  const bool dec = L.logical_decimal || L.converted == 5;
  gp = 0; gs = 0; width = 0;
  switch (L.type) {
    case 0: gt = T_BOOL; return true;
    case 1:
      if (dec) { gt = T_DECIMAL128; gp = L.precision; gs = L.scale; width = 16; return true; }
      if (L.logical_date || L.converted == 6) { gt = T_DATE32; width = 4; return true; }
      if (L.int_bits == 8 || L.converted == 15 || L.converted == 11) { const bool sg = L.int_bits ? L.int_signed : L.converted == 15; gt = sg ? T_INT8 : T_UINT8; width = 1; return true; }
      if (L.int_bits == 16 || L.converted == 16 || L.converted == 12) { const bool sg = L.int_bits ? L.int_signed : L.converted == 16; gt = sg ? T_INT16 : T_UINT16; width = 2; return true; }
      gt = ((L.int_bits == 32 && !L.int_signed) || L.converted == 13) ? T_UINT32 : T_INT32; width = 4; return true;
    case 2:
      if (dec) { gt = T_DECIMAL128; gp = L.precision; gs = L.scale; width = 16; return true; }
      if (L.ts_unit > 0 || L.converted == 9 || L.converted == 10) { gt = T_TIMESTAMP; gp = L.ts_unit > 0 ? L.ts_unit : (L.converted == 9 ? 1 : 2); width = 8; return true; }
      gt = ((L.int_bits == 64 && !L.int_signed) || L.converted == 14) ? T_UINT64 : T_INT64; width = 8; return true;
    case 3: gt = T_TIMESTAMP; gp = 3; width = 8; return true;
    case 4: gt = T_FLOAT32; width = 4; return true;
    case 5: gt = T_FLOAT64; width = 8; return true;
    case 6: if (dec) return false; gt = T_UTF8; return true;
    case 7: if (!dec || L.type_length < 1 || L.type_length > 16) return false; gt = T_DECIMAL128; gp = L.precision; gs = L.scale; width = 16; return true;
    default: return false;
  }
}
PqChunk read_chunk(TReader& r) {
  PqChunk c; int t, id, last = 0;
  while (r.field(t, id, last)) {
    if (id == 3 && t == 12) {
      int t2, id2, last2 = 0;
      while (r.field(t2, id2, last2)) {
        switch (id2) {
          case 1: c.type = (int)r.zigzag(); break;
          case 3: { int et; const uint64_t n = r.list_header(et); for (uint64_t i = 0; i < n; ++i) c.path.push_back(r.binary()); break; }
          case 4: c.codec = (int)r.zigzag(); break; case 5: c.num_values = r.zigzag(); break; case 7: c.total_compressed = r.zigzag(); break;
          case 9: c.data_page_offset = r.zigzag(); break; case 11: c.dict_page_offset = r.zigzag(); break;
          default: r.skip(t2);
        }
      }
    } else r.skip(t);
  }
  return c;
}
PqFileMeta read_footer(const uint8_t* file, int64_t n) {
  if (n < 12 || std::memcmp(file, "PAR1", 4) != 0 || std::memcmp(file + n - 4, "PAR1", 4) != 0) throw std::runtime_error("parquet: missing PAR1 magic");
  uint32_t flen; std::memcpy(&flen, file + n - 8, 4);
  if ((int64_t)flen + 12 > n) throw std::runtime_error("parquet: footer length exceeds the file");
  TReader r{file + n - 8 - flen, file + n - 8};
  PqFileMeta m; int t, id, last = 0;
  while (r.field(t, id, last)) {
    if (id == 2) { int et; const uint64_t k = r.list_header(et); for (uint64_t i = 0; i < k; ++i) m.schema.push_back(read_schema_elem(r)); }
    else if (id == 3) m.num_rows = r.zigzag();
    else if (id == 4) {
      int et; const uint64_t k = r.list_header(et);
      for (uint64_t i = 0; i < k; ++i) {
        PqRowGroup g; int t2, id2, last2 = 0;
        while (r.field(t2, id2, last2)) {
          if (id2 == 1) { int et2; const uint64_t kc = r.list_header(et2); for (uint64_t j = 0; j < kc; ++j) g.cols.push_back(read_chunk(r)); }
          else if (id2 == 3) g.num_rows = r.zigzag();
          else r.skip(t2);
        }
        m.groups.push_back(std::move(g));
      }
    } else r.skip(t);
  }
  return m;
}
struct PqPageHeader { int type = -1, uncompressed = 0, compressed = 0, num_values = 0, encoding = 0, def_v2 = 0, rep_v2 = 0; bool v2_compressed = true; int64_t header_bytes = 0; };
PqPageHeader read_page_header(const uint8_t* p, const uint8_t* e) {
  TReader r{p, e}; PqPageHeader h; int t, id, last = 0;
  while (r.field(t, id, last)) {
    switch (id) {
      case 1: h.type = (int)r.zigzag(); break; case 2: h.uncompressed = (int)r.zigzag(); break; case 3: h.compressed = (int)r.zigzag(); break;
      case 5: case 7: { int t2, id2, last2 = 0; while (r.field(t2, id2, last2)) { if (id2 == 1) h.num_values = (int)r.zigzag(); else if (id2 == 2) h.encoding = (int)r.zigzag(); else r.skip(t2); } break; }
      case 8: { int t2, id2, last2 = 0;
        while (r.field(t2, id2, last2)) {
          if (id2 == 1) h.num_values = (int)r.zigzag(); else if (id2 == 4) h.encoding = (int)r.zigzag(); else if (id2 == 5) h.def_v2 = (int)r.zigzag();
          else if (id2 == 6) h.rep_v2 = (int)r.zigzag(); else if (id2 == 7) h.v2_compressed = (t2 == 1); else r.skip(t2);
        }
        break; }
      default: r.skip(t);
    }
  }
  h.header_bytes = r.p - p;
  return h;
}
}  // namespace

extern "C" {

int gpuq_parquet_decode(gpuq_ctx* ctx, void* stream, const uint8_t* file, int64_t n_bytes, const char* const* columns, int n_columns, gpuq_table** out) {
  return gpuq_parquet_decode_groups(ctx, stream, file, n_bytes, columns, n_columns, nullptr, 0, out);
}

int gpuq_parquet_decode_groups(gpuq_ctx* ctx, void* stream, const uint8_t* file, int64_t n_bytes, const char* const* columns, int n_columns,
                               const int32_t* row_groups, int n_row_groups, gpuq_table** out) {
  if (out) *out = nullptr;
  return guarded_f([&]() {
    if (!ctx || !out || !file) throw std::runtime_error("ctx / file / out is NULL");
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t s = use_stream(stream);
    PqFileMeta M = read_footer(file, n_bytes);
    if (row_groups) {      // the caller's pruning (statistics, partition filters): only these row groups, in the order given
      std::vector<PqRowGroup> pick; int64_t rows = 0;
      for (int i = 0; i < n_row_groups; ++i) {
        if (row_groups[i] < 0 || (size_t)row_groups[i] >= M.groups.size()) throw std::runtime_error("parquet: row group " + std::to_string(row_groups[i]) + " is not in the file");
        pick.push_back(M.groups[(size_t)row_groups[i]]); rows += pick.back().num_rows;
      }
      M.groups.swap(pick); M.num_rows = rows;
    }
    if (M.schema.empty()) throw std::runtime_error("parquet: empty schema");
    // flat schemas: the root and its leaf children
    std::vector<PqSchemaElem> leaves(M.schema.begin() + 1, M.schema.end());
    for (auto& l : leaves) if (l.num_children > 0 || l.repetition == 2) throw Unsupported("parquet: nested / repeated column '" + l.name + "'");
    std::vector<int> proj;
    if (columns) {
      for (int i = 0; i < n_columns; ++i) {
        int found = -1; for (size_t k = 0; k < leaves.size(); ++k) if (leaves[k].name == columns[i]) found = (int)k;
        if (found < 0) throw std::runtime_error(std::string("parquet: column '") + columns[i] + "' is not in the file");
        proj.push_back(found);
      }
    } else for (size_t k = 0; k < leaves.size(); ++k) proj.push_back((int)k);
    const int64_t n_rows = M.num_rows;
    if (n_rows > 0xFFFFFFFEll) throw Unsupported("parquet: more than 2^32-2 rows per call");
    // ---- pass 1 (host): per projected column the type mapping, the byte ranges of its chunks and one descriptor per page
    struct DictSrc { int64_t src; int bytes, n; };
    struct ColPlan {
      int gt = -1, gp = 0, gs = 0, width = 0, max_values = 0; bool optional = false;
      std::vector<PqPage> pages; std::vector<DictSrc> dict_src; std::vector<PqDict> dicts;
      DevBuf str_src, dvalues, dstroffs, ddicts, dpages, scratch;
    };
    std::vector<std::unique_ptr<ColPlan>> plans;
    std::vector<Uploader::Seg> segs; int64_t up_bytes = 0;      // projected chunks back to back (64-byte aligned) in one device buffer
    // Snappy chunks: every page (of every projected column) goes through one unpack launch into a second buffer the decoders read
    bool any_compressed = false;
    for (int li : proj) for (const PqRowGroup& G : M.groups) if ((size_t)li < G.cols.size() && G.cols[(size_t)li].num_values > 0 && G.cols[(size_t)li].codec != 0) any_compressed = true;
    std::vector<UnpackJob> jobs; int64_t page_bytes = 0; bool any_lz4 = false;
    auto place = [&](int64_t src, int comp, int uncomp, int raw_prefix, int mode) -> int64_t {      // -> the page's position for the decoders
      if (!any_compressed) return src;
      const int64_t dst = page_bytes; page_bytes += ((int64_t)uncomp + 63) & ~(int64_t)63;
      jobs.push_back({src, dst, comp, uncomp, raw_prefix, mode, 0, 0, 0, 0, 0});
      return dst;
    };
    for (int li : proj) {
      const PqSchemaElem& L = leaves[(size_t)li];
      auto P = std::make_unique<ColPlan>();
      // type mapping (parquet.thrift Type / ConvertedType / LogicalType -> gpuq_type)
      if (!pq_leaf_type(L, P->gt, P->gp, P->gs, P->width))
        throw Unsupported("parquet: column '" + L.name + "' (physical type " + std::to_string(L.type) + (L.type == 7 ? ", type_length " + std::to_string(L.type_length) : std::string()) + ": BYTE_ARRAY decimals and FIXED_LEN_BYTE_ARRAY other than decimals of 1..16 bytes are not read on the device)");
      P->optional = L.repetition == 1;
      int64_t row = 0;
      for (const PqRowGroup& G : M.groups) {
        if ((size_t)li >= G.cols.size()) throw std::runtime_error("parquet: row group without column " + L.name);
        const PqChunk& K = G.cols[(size_t)li];
        if (K.num_values == 0) continue;
        if (K.num_values < 0 || K.total_compressed <= 0 || K.data_page_offset < 0) throw std::runtime_error("parquet: negative size / offset in a column chunk's metadata");
        if (K.codec != 0 && K.codec != 1 && K.codec != 6 && K.codec != 7) throw Unsupported("parquet: compressed column chunk (codec " + std::to_string(K.codec) + ") in '" + L.name + "': UNCOMPRESSED, SNAPPY, ZSTD and LZ4_RAW pages are decoded on the device");
        const int cmode = K.codec == 1 ? 1 : (K.codec == 7 ? 2 : (K.codec == 6 ? 4 : 0));      // UnpackJob::mode: 1 Snappy, 2 one raw LZ4 block per page (LZ4_RAW), 4 ZSTD frames
        if (cmode == 2) any_lz4 = true;
        int64_t pos = K.dict_page_offset >= 0 && K.dict_page_offset < K.data_page_offset ? K.dict_page_offset : K.data_page_offset;
        const int64_t chunk_end = pos + K.total_compressed;
        if (pos < 4 || chunk_end > n_bytes - 8) throw std::runtime_error("parquet: column chunk outside the file");
        const int64_t base = up_bytes - pos;      // device position = file position + base
        segs.push_back({(size_t)up_bytes, file + pos, (size_t)K.total_compressed}); up_bytes += (K.total_compressed + 63) & ~(int64_t)63;
        int64_t seen = 0; int dict_id = -1;
        while (pos < chunk_end && seen < K.num_values) {
          const PqPageHeader H = read_page_header(file + pos, file + chunk_end);
          const int64_t payload = pos + H.header_bytes;
          if (H.compressed < 0 || H.uncompressed < 0 || H.num_values < 0 || H.def_v2 < 0 || H.rep_v2 < 0 || H.header_bytes <= 0) throw std::runtime_error("parquet: negative size in a page header");
          if (payload + H.compressed > chunk_end) throw std::runtime_error("parquet: page exceeds its column chunk");
          if (H.type == 3 && (int64_t)H.def_v2 + H.rep_v2 > H.compressed) throw std::runtime_error("parquet: v2 page levels exceed the page");
          if (H.type == 2) {       // dictionary page
            if (H.encoding != 0 && H.encoding != 2) throw Unsupported("parquet: dictionary page encoding " + std::to_string(H.encoding));
            dict_id = (int)P->dict_src.size(); P->dict_src.push_back({place(payload + base, H.compressed, H.uncompressed, 0, cmode), any_compressed ? H.uncompressed : H.compressed, H.num_values});
          } else if (H.type == 0 || H.type == 3) {
            // a v2 page stores its levels uncompressed in front of the (optionally) compressed values
            const int prefix = H.type == 3 ? H.def_v2 + H.rep_v2 : 0;
            const int pmode = (H.type == 3 && !H.v2_compressed) ? 0 : cmode;
            PqPage G2{}; G2.src = place(payload + base, H.compressed, H.uncompressed, prefix, pmode); G2.bytes = any_compressed ? H.uncompressed : H.compressed;
            G2.n_values = H.num_values; G2.row0 = row + seen; G2.dict = dict_id;
            if (H.encoding == 0) G2.enc = PQE_PLAIN; else if (H.encoding == 2 || H.encoding == 8) G2.enc = PQE_DICT; else if (H.encoding == 3 && L.type == 0) G2.enc = PQE_RLE;
            else throw Unsupported("parquet: data page encoding " + std::to_string(H.encoding) + " in '" + L.name + "' (PLAIN, RLE_DICTIONARY and RLE booleans are decoded)");
            if (H.type == 3) { if (H.rep_v2 != 0) throw Unsupported("parquet: repetition levels"); G2.def_v2 = H.def_v2; }
            if (H.type == 3 && P->optional && H.def_v2 == 0 && H.num_values > 0) throw std::runtime_error("parquet: v2 page of an optional column without definition levels");
            if (G2.enc == PQE_DICT && dict_id < 0) throw std::runtime_error("parquet: dictionary-encoded page before any dictionary page");
            P->pages.push_back(G2); seen += H.num_values; P->max_values = std::max(P->max_values, H.num_values);
          }
          pos = payload + H.compressed;
        }
        if (seen != K.num_values) throw std::runtime_error("parquet: pages of '" + L.name + "' hold " + std::to_string(seen) + " values, the chunk declares " + std::to_string(K.num_values));
        row += K.num_values;
      }
      if (row != n_rows) throw std::runtime_error("parquet: column '" + L.name + "' has " + std::to_string(row) + " values for " + std::to_string(n_rows) + " rows");
      plans.push_back(std::move(P));
    }
    // ---- the projected chunks cross PCIe once, all staging lanes busy
    DevBuf dfile; dfile.ensure((size_t)up_bytes + 64);
    g_uploader.copy_segments(ctx->device, s, dfile.p, segs);
    DevBuf flags; flags.ensure(16); HIPCHECK(hipMemsetAsync(flags.p, 0, 16, s));
    DevBuf dpagebuf, djobs, dresolve, dblk, dcblk, dcnt, dja, djb, dolen, dmark, dscan;
    const uint8_t* pages_base = (const uint8_t*)dfile.p; int64_t pages_bytes = up_bytes;
    if (any_compressed) {
      dpagebuf.ensure((size_t)page_bytes + 64); djobs.ensure(jobs.size() * sizeof(UnpackJob) + 64);
      HIPCHECK(hipMemcpyAsync(djobs.p, jobs.data(), jobs.size() * sizeof(UnpackJob), hipMemcpyHostToDevice, s));
      // Snappy pages are decoded without a serial element walk (kernels_lz4.hip: positions -> path by doubling -> scan -> resolve words ->
      // pointer jumping), in batches of consecutive pages whose scratch -- a 32-bit word per uncompressed byte, 13 bytes per compressed
      // byte -- stays within ~5 GB (2^30 words); GPUQ_SNAPPY_PJ=0 (or a single page beyond that) keeps the serial decoder
      static const bool pj_on = []() { const char* e = std::getenv("GPUQ_SNAPPY_PJ"); return !(e && e[0] == '0'); }();
      const int64_t max_words = (int64_t)1 << 30;
      bool any_snappy = false, fits = true;
      for (auto& j : jobs) if (j.mode == 1 || j.mode == 2) { any_snappy = true; if (j.dst_len > max_words || j.src_len > max_words) fits = false; }
      if (any_lz4 && !(pj_on && fits)) throw Unsupported("parquet: LZ4_RAW pages are decoded by the parallel path only (GPUQ_SNAPPY_PJ=0, or a page beyond 2^30 bytes)");
      if (pj_on && any_snappy && fits) {
        HIPCHECK(hipMemsetAsync(dmark.ensure(64), 0, 64, s));
        size_t first = 0;
        while (first < jobs.size()) {
          int64_t words = 0, max_len = 0, slots = 0, max_in = 0;
          size_t last = first;
          std::vector<uint2> blk, cblk;
          for (; last < jobs.size(); ++last) {
            UnpackJob& j = jobs[last];
            const bool sn = (j.mode == 1 || j.mode == 2) && j.raw_prefix <= j.src_len && j.raw_prefix <= j.dst_len;
            const int64_t len = sn ? j.dst_len - j.raw_prefix : 0, cl = sn ? j.src_len - j.raw_prefix : 0;
            if (last > first && sn && (words + len > max_words || slots + cl + 1 > max_words)) break;
            j.s_off = words; j.c_off = slots; j.f_off = words; j.p_base = 0;
            if (!sn) continue;
            const uint32_t k = (uint32_t)(last - first);
            for (int64_t bb = 0; bb < len; bb += 4096) blk.push_back(make_uint2(k, (uint32_t)bb));
            for (int64_t bb = 0; bb <= cl; bb += 4096) cblk.push_back(make_uint2(k, (uint32_t)bb));
            words += (len + 3) & ~(int64_t)3; slots += cl + 1;
            if (len > max_len) max_len = len; if (cl > max_in) max_in = cl;
          }
          const size_t nj = last - first;
          int rounds = 1; while (((int64_t)1 << rounds) < max_len) ++rounds;
          rounds += 1;
          int mrounds = 1; while (((int64_t)1 << mrounds) < max_in + 1) ++mrounds;
          mrounds += 1;
          dresolve.ensure((size_t)words * 4 + 64); dblk.ensure(blk.size() * sizeof(uint2) + 64); dcblk.ensure(cblk.size() * sizeof(uint2) + 64);
          dcnt.ensure((size_t)(rounds + mrounds + 3) * nj * 4 + 64);
          dja.ensure((size_t)(slots + 1) * 4 + 64); djb.ensure((size_t)(slots + 1) * 4 + 64); dolen.ensure((size_t)(slots + 1) * 4 + 64); dmark.ensure((size_t)slots + 64);
          const size_t swb = exclusive_scan_ws_bytes(slots + 1); dscan.ensure(swb);
          HIPCHECK(hipMemcpyAsync((UnpackJob*)djobs.p + first, jobs.data() + first, nj * sizeof(UnpackJob), hipMemcpyHostToDevice, s));      // (again: with the s_off / c_off fields)
          if (!blk.empty()) HIPCHECK(hipMemcpyAsync(dblk.p, blk.data(), blk.size() * sizeof(uint2), hipMemcpyHostToDevice, s));
          if (!cblk.empty()) HIPCHECK(hipMemcpyAsync(dcblk.p, cblk.data(), cblk.size() * sizeof(uint2), hipMemcpyHostToDevice, s));
          HIPCHECK(hipMemsetAsync(dcnt.p, 0, (size_t)(rounds + mrounds + 3) * nj * 4, s));
          HIPCHECK(hipMemsetAsync(dmark.p, 0, (size_t)slots + 64, s));
          SnappyPjBuffers B{};
          B.resolve = (uint32_t*)dresolve.p; B.blkmap = (const uint2*)dblk.p; B.n_blocks = (int)blk.size(); B.rounds = rounds;
          B.jump_a = (uint32_t*)dja.p; B.jump_b = (uint32_t*)djb.p; B.olen = (uint32_t*)dolen.p; B.mark = (uint8_t*)dmark.p; B.cmap = (const uint2*)dcblk.p; B.n_cblocks = (int)cblk.size();
          B.mark_rounds = mrounds; B.c_slots = slots; B.scan_ws = dscan.p; B.scan_ws_bytes = swb; B.counts = (uint32_t*)dcnt.p;
          launch_unpack_pages_pj(s, (const uint8_t*)dfile.p, (uint8_t*)dpagebuf.p, (const UnpackJob*)djobs.p + first, (int)nj, B, (uint32_t*)flags.p + 1);
          HIPCHECK(hipStreamSynchronize(s));      // the block maps (pageable host memory) and the scratch are reused by the next batch
          first = last;
        }
      } else
      launch_unpack_pages(s, (const uint8_t*)dfile.p, (uint8_t*)dpagebuf.p, (const UnpackJob*)djobs.p, (int)jobs.size(), (uint32_t*)flags.p + 1);
      // ZSTD pages (the reference's `tpch convert` default): one wave per page, in launches of at most 4096 pages
      std::vector<int32_t> zjobs;
      for (size_t j = 0; j < jobs.size(); ++j) if (jobs[j].mode == 4) zjobs.push_back((int32_t)j);
      DevBuf dzwhich, dzscratch;
      if (!zjobs.empty()) {
        const int zbatch = 4096;
        dzwhich.ensure(zjobs.size() * 4 + 64); dzscratch.ensure(zstd_scratch_bytes((int)std::min<size_t>(zjobs.size(), (size_t)zbatch)) + 64);
        HIPCHECK(hipMemcpyAsync(dzwhich.p, zjobs.data(), zjobs.size() * 4, hipMemcpyHostToDevice, s));
        for (size_t z = 0; z < zjobs.size(); z += (size_t)zbatch)
          launch_zstd_pages(s, (const uint8_t*)dfile.p, (uint8_t*)dpagebuf.p, (const UnpackJob*)djobs.p, (const int32_t*)dzwhich.p + z, (int)std::min<size_t>((size_t)zbatch, zjobs.size() - z),
                            (uint8_t*)dzscratch.p, (uint32_t*)flags.p + 1);
        HIPCHECK(hipStreamSynchronize(s));      // (zjobs is pageable host memory; the scratch goes away with this scope)
      }
      pages_base = (const uint8_t*)dpagebuf.p; pages_bytes = page_bytes;
    }
    // ---- pass 2 (device): dictionaries, pages, strings; one synchronisation at the end
    std::unique_ptr<gpuq_table> t(new gpuq_table()); t->ctx = ctx; t->n_rows = n_rows;
    const size_t bm = (size_t)((n_rows + 63) / 64) * 8 + 16;
    const size_t rows1 = (size_t)std::max<int64_t>(n_rows, 1);
    for (size_t pi = 0; pi < plans.size(); ++pi) {
      ColPlan& P = *plans[pi]; const PqSchemaElem& L = leaves[(size_t)proj[pi]];
      const int gt = P.gt, width = P.width;
      std::unique_ptr<ImportedCol> ic(new ImportedCol());
      ic->field = field_of(L.name, gt, P.gp, P.gs, P.optional);
      ic->col.type = gt; ic->col.precision = P.gp; ic->col.scale = P.gs; ic->col.repr = GPUQ_REPR_ARROW; ic->col.length = n_rows;
      PqCol C{}; C.phys = L.type; C.width = width; C.flba_len = L.type_length; C.optional = P.optional ? 1 : 0;
      if (gt == T_UTF8) { ic->offsets.ensure((size_t)(n_rows + 2) * 4 + 16); P.str_src.ensure(rows1 * 8); C.str_len = (int32_t*)ic->offsets.p; C.str_src = (i64*)P.str_src.p; }
      else if (gt == T_BOOL) { ic->data.ensure(bm); HIPCHECK(hipMemsetAsync(ic->data.p, 0, bm, s)); C.data = ic->data.p; ic->col.data = ic->data.p; }
      else { ic->data.ensure(rows1 * (size_t)width + 16); if (P.optional) HIPCHECK(hipMemsetAsync(ic->data.p, 0, rows1 * (size_t)width, s)); C.data = ic->data.p; ic->col.data = ic->data.p; }
      if (P.optional) { ic->validity.ensure(bm); HIPCHECK(hipMemsetAsync(ic->validity.p, 0, bm, s)); C.valid = (u64*)ic->validity.p; ic->col.validity = (const uint8_t*)ic->validity.p; }
      // dictionaries -> output-width values / contiguous strings
      size_t vbytes = 0, obytes = 0;
      for (auto& d : P.dict_src) { vbytes += (gt == T_UTF8 ? (size_t)d.bytes : (size_t)d.n * (size_t)width) + 64; obytes += ((size_t)d.n + 2) * 4; }
      P.dvalues.ensure(vbytes + 64); P.dstroffs.ensure(obytes + 64);
      size_t va = 0, oa = 0;
      for (auto& d : P.dict_src) {
        PqDict D{}; D.values = (i64)va; D.str_offsets = (i64)(oa / 4); D.n = d.n;
        if (gt == T_UTF8) { launch_pq_dict_strings(s, pages_base, d.src, d.bytes, d.n, (int32_t*)P.dstroffs.p + oa / 4, (uint8_t*)P.dvalues.p + va, (uint32_t*)flags.p); va += ((size_t)d.bytes + 63) & ~(size_t)63; oa += ((size_t)d.n + 2) * 4; }
        else {
          const int64_t elem = (L.type == 1 || L.type == 4) ? 4 : (L.type == 7 ? L.type_length : (L.type == 3 ? 12 : 8));
          if ((int64_t)d.n * elem > d.bytes) throw std::runtime_error("parquet: dictionary page of '" + L.name + "' is shorter than its value count");
          launch_pq_dict_fixed(s, pages_base, d.src, d.n, L.type, L.type_length, width, (uint8_t*)P.dvalues.p + va); va += ((size_t)d.n * (size_t)width + 63) & ~(size_t)63; }
        P.dicts.push_back(D);
      }
      P.ddicts.ensure(P.dicts.size() * sizeof(PqDict) + 64);
      if (!P.dicts.empty()) HIPCHECK(hipMemcpyAsync(P.ddicts.p, P.dicts.data(), P.dicts.size() * sizeof(PqDict), hipMemcpyHostToDevice, s));
      // pages: row index + (dictionary indices | 8-byte string positions) per page, 8-byte aligned rows
      P.dpages.ensure(P.pages.size() * sizeof(PqPage) + 64);
      if (!P.pages.empty()) HIPCHECK(hipMemcpyAsync(P.dpages.p, P.pages.data(), P.pages.size() * sizeof(PqPage), hipMemcpyHostToDevice, s));
      const i64 stride = (3 * (i64)P.max_values + 9) & ~(i64)1;
      P.scratch.ensure((size_t)std::max<size_t>(P.pages.size(), 1) * (size_t)stride * 4 + 64);
      launch_pq_decode(s, pages_base, pages_bytes, (const PqPage*)P.dpages.p, (int)P.pages.size(), C, (const PqDict*)P.ddicts.p, (const uint8_t*)P.dvalues.p, (const int32_t*)P.dstroffs.p,
                       (uint32_t*)P.scratch.p, stride, (uint32_t*)flags.p);
      HIPCHECK(hipGetLastError());
      if (gt == T_UTF8) {
        // lengths and positions of a malformed page are garbage: look at the flags BEFORE anything copies through them
        uint32_t early[2]; HIPCHECK(hipMemcpyAsync(early, flags.p, 8, hipMemcpyDeviceToHost, s)); HIPCHECK(hipStreamSynchronize(s));
        if (early[1]) throw std::runtime_error("parquet: malformed Snappy data in a page");
        if (early[0] & PQF_MALFORMED) throw std::runtime_error("parquet: malformed page (levels / indices / lengths run past the page, or an index beyond its dictionary)");
        const uint8_t* df = pages_base; const uint8_t* dv = (const uint8_t*)P.dvalues.p; const i64* src = (const i64*)P.str_src.p;
        finish_strings(s, *ic, (int32_t*)ic->offsets.p, n_rows, [&](const int32_t* offs, uint8_t* dst, int64_t) { launch_pq_copy_strings(s, df, dv, src, offs, n_rows, dst); });
      }
      t->cols.push_back(std::move(ic));
    }
    uint32_t fl2[2]; HIPCHECK(hipMemcpyAsync(fl2, flags.p, 8, hipMemcpyDeviceToHost, s)); HIPCHECK(hipStreamSynchronize(s));      // synchronises: the plans' host vectors and scratch may go
    if (fl2[1]) throw std::runtime_error("parquet: malformed Snappy data in a page");
    const uint32_t fl = fl2[0];
    if (fl & PQF_MALFORMED) throw std::runtime_error("parquet: malformed page (levels / indices / lengths run past the page, or an index beyond its dictionary)");
    if (fl & PQF_UNSUPPORTED) throw Unsupported("parquet: a page holds a value type the device does not decode");
    *out = t.release();
  });
}

int gpuq_parquet_row_groups(const uint8_t* file, int64_t n_bytes, int64_t* rows_out, int cap, int* n_out) {
  return guarded_f([&]() {
    if (!file || !n_out) throw std::runtime_error("file / n_out is NULL");
    const PqFileMeta M = read_footer(file, n_bytes);
    *n_out = (int)M.groups.size();
    if (!rows_out) return;
    if (cap < *n_out) throw Capacity("the file has " + std::to_string(*n_out) + " row groups");
    for (size_t i = 0; i < M.groups.size(); ++i) rows_out[i] = M.groups[i].num_rows;
  });
}

// Host only: the leaf columns of a Parquet file as gpuq fields (type = -1 for a column the device does not decode), and its row count.
int gpuq_parquet_schema(const uint8_t* file, int64_t n_bytes, gpuq_field_info* fields_out, int cap, int* n_out, int64_t* rows_out) {
  return guarded_f([&]() {
    if (!file || !n_out) throw std::runtime_error("file / n_out is NULL");
    const PqFileMeta M = read_footer(file, n_bytes);
    const int n = M.schema.empty() ? 0 : (int)M.schema.size() - 1;
    *n_out = n; if (rows_out) *rows_out = M.num_rows;
    if (!fields_out || cap < n) { if (!fields_out && cap == 0) return; throw Capacity("parquet schema has " + std::to_string(n) + " columns"); }
    for (int i = 0; i < n; ++i) {
      const PqSchemaElem& L = M.schema[(size_t)i + 1];
      int gt = -1, gp = 0, gs = 0, gw = 0;
      if (!pq_leaf_type(L, gt, gp, gs, gw)) gt = -1;
      if (L.num_children > 0 || L.repetition == 2) gt = -1;
      gpuq_field_info f{}; std::snprintf(f.name, sizeof(f.name), "%s", L.name.c_str());
      f.type = gt; f.precision = gp; f.scale = gs; f.nullable = L.repetition == 1; f.repr = GPUQ_REPR_ARROW;
      fields_out[i] = f;
    }
  });
}

}  // extern "C"
