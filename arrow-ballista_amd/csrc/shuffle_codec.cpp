// Shuffle sink / source codec: one Arrow IPC RecordBatch message <-> device columns, buffers compressed as
// LZ4 frames on the device.  What it stands in for in the reference:
//   * write: `StreamWriter::try_new_with_options(file, schema, IpcWriteOptions.try_with_compression(Some(LZ4_FRAME)))` + `write(&batch)`
//     -- ballista/core/src/execution_plans/shuffle_writer.rs:365-378 (hash-partitioned), ballista/core/src/utils.rs:179-219 (unpartitioned);
//   * read: `StreamReader` / `read_record_batch` -- ballista/core/src/execution_plans/shuffle_reader.rs, ballista/core/src/client.rs,
//     ballista/executor/.../async_reader/mod.rs:168-258.
// The codec itself lives in third-party crates (arrow-ipc 49.0.0 `compression.rs` over lz4_flex; not in the tree): the formats
// restated here are the public ones -- Arrow columnar IPC (Message.fbs / Schema.fbs: encapsulated message = 0xFFFFFFFF, int32
// metadata length, flatbuffer, body; BodyCompression: every buffer = int64 uncompressed length (-1 = stored raw) + codec frame)
// and the LZ4 frame format 1.6.x.  This file is a client of devbuf.h and the launchers in kernels_lz4.hip; no CPU (de)compressor exists
// here: the host only frames (headers, flatbuffer metadata, block index walk).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gpuq.h"
#include "devbuf.h"
#include "gpuq_kernels.h"

using namespace gpuq;

namespace {
thread_local std::string g_ipc_error;

template <class F> int guarded_ipc(F&& f) {
  try { f(); return GPUQ_OK; }
  catch (const HipError& e) { g_ipc_error = e.what(); return GPUQ_ERR_HIP; }
  catch (const Unsupported& e) { g_ipc_error = e.what(); return GPUQ_ERR_UNSUPPORTED; }
  catch (const Capacity& e) { g_ipc_error = e.what(); return GPUQ_ERR_CAPACITY; }
  catch (const std::bad_alloc&) { g_ipc_error = "out of host memory"; return GPUQ_ERR_INTERNAL; }
  catch (const std::exception& e) { g_ipc_error = e.what(); return GPUQ_ERR_INVALID; }
}

// ---------------------------------------------------------------- flatbuffer access (read side: generic; write side: one fixed layout)
struct FbView {
  const uint8_t* b; size_t n;
  void need(size_t at, size_t k) const { if (at > n || k > n - at) throw std::runtime_error("IPC metadata: offset outside the flatbuffer"); }
  uint16_t u16(size_t at) const { need(at, 2); uint16_t v; std::memcpy(&v, b + at, 2); return v; }
  uint32_t u32(size_t at) const { need(at, 4); uint32_t v; std::memcpy(&v, b + at, 4); return v; }
  int32_t i32(size_t at) const { return (int32_t)u32(at); }
  int64_t i64(size_t at) const { need(at, 8); int64_t v; std::memcpy(&v, b + at, 8); return v; }
  uint8_t u8(size_t at) const { need(at, 1); return b[at]; }
  size_t field(size_t table, int k) const {      // absolute position of field k of a table, 0 when absent
    const int64_t vt = (int64_t)table - (int64_t)i32(table);
    if (vt < 0) throw std::runtime_error("IPC metadata: bad vtable offset");
    const uint16_t vsize = u16((size_t)vt);
    const size_t slot = 4 + 2 * (size_t)k;
    if (slot + 2 > vsize) return 0;
    const uint16_t off = u16((size_t)vt + slot);
    return off ? table + off : 0;
  }
  size_t indirect(size_t at) const { return at + u32(at); }
};

struct BatchMeta {
  int header_type = 0, codec = -1; int64_t body_len = 0, n_rows = 0;
  std::vector<std::pair<int64_t, int64_t>> nodes, buffers;      // (length, null_count), (offset, length)
};

BatchMeta parse_message(const uint8_t* fb, size_t n) {
  FbView v{fb, n};
  BatchMeta m;
  const size_t msg = v.indirect(0);
  if (size_t p = v.field(msg, 1)) m.header_type = v.u8(p);
  if (size_t p = v.field(msg, 3)) m.body_len = v.i64(p);
  if (m.header_type != 3) return m;
  size_t hp = v.field(msg, 2);
  if (!hp) throw std::runtime_error("IPC metadata: RecordBatch message without a header");
  const size_t rb = v.indirect(hp);
  if (size_t p = v.field(rb, 0)) m.n_rows = v.i64(p);
  if (size_t p = v.field(rb, 1)) {
    const size_t vec = v.indirect(p); const uint32_t cnt = v.u32(vec);
    for (uint32_t i = 0; i < cnt; ++i) m.nodes.push_back({v.i64(vec + 4 + 16 * (size_t)i), v.i64(vec + 12 + 16 * (size_t)i)});
  }
  if (size_t p = v.field(rb, 2)) {
    const size_t vec = v.indirect(p); const uint32_t cnt = v.u32(vec);
    for (uint32_t i = 0; i < cnt; ++i) m.buffers.push_back({v.i64(vec + 4 + 16 * (size_t)i), v.i64(vec + 12 + 16 * (size_t)i)});
  }
  if (size_t p = v.field(rb, 3)) {
    const size_t bc = v.indirect(p);
    m.codec = 0;
    if (size_t c = v.field(bc, 0)) m.codec = (int8_t)v.u8(c);
    if (size_t c = v.field(bc, 1)) if (v.u8(c) != 0) throw Unsupported("IPC BodyCompression method other than BUFFER");
  }
  return m;
}

// One RecordBatch message as a flatbuffer (Message{version V5, header RecordBatch{length, nodes, buffers, compression?}, bodyLength}).
std::vector<uint8_t> build_batch_metadata(const BatchMeta& m) {
  const size_t nn = m.nodes.size(), nb = m.buffers.size();
  const size_t nodes_cnt = 92, nodes_dat = 96, bufs_cnt = nodes_dat + 16 * nn + 4, bufs_dat = bufs_cnt + 4, total = bufs_dat + 16 * nb;
  std::vector<uint8_t> b(total, 0);
  auto p16 = [&](size_t at, uint16_t x) { std::memcpy(&b[at], &x, 2); };
  auto p32 = [&](size_t at, uint32_t x) { std::memcpy(&b[at], &x, 4); };
  auto p64 = [&](size_t at, int64_t x) { std::memcpy(&b[at], &x, 8); };
  p32(0, 16);
  // Message vtable @4, table @16
  p16(4, 12); p16(6, 20); p16(8, 16); p16(10, 18); p16(12, 4); p16(14, 8);
  p32(16, 12); p32(20, 48 - 20); p64(24, m.body_len); p16(32, 4 /* MetadataVersion V5 */); b[34] = 3 /* MessageHeader RecordBatch */;
  // RecordBatch vtable @36, table @48
  p16(36, 12); p16(38, 24); p16(40, 8); p16(42, 4); p16(44, 16); p16(46, m.codec >= 0 ? 20 : 0);
  p32(48, 12); p32(52, (uint32_t)(nodes_cnt - 52)); p64(56, m.n_rows); p32(64, (uint32_t)(bufs_cnt - 64)); p32(68, m.codec >= 0 ? 80 - 68 : 0);
  // BodyCompression vtable @72, table @80
  p16(72, 8); p16(74, 8); p16(76, 4); p16(78, 5);
  p32(80, 8); b[84] = (uint8_t)(m.codec >= 0 ? m.codec : 0); b[85] = 0;
  p32(nodes_cnt, (uint32_t)nn);
  for (size_t i = 0; i < nn; ++i) { p64(nodes_dat + 16 * i, m.nodes[i].first); p64(nodes_dat + 16 * i + 8, m.nodes[i].second); }
  p32(bufs_cnt, (uint32_t)nb);
  for (size_t i = 0; i < nb; ++i) { p64(bufs_dat + 16 * i, m.buffers[i].first); p64(bufs_dat + 16 * i + 8, m.buffers[i].second); }
  return b;
}

// ---- Schema message (Schema.fbs): Message{V5, Schema{endianness Little, fields:[Field{name, nullable, type, children:[]}]}, bodyLength 0}.
// A small front-to-back flatbuffer writer: a table is written with its vtable in front of it and zeroed offset slots; whatever
// a slot refers to is appended later and the slot patched (uoffsets point forward, as the format requires).
struct FbWriter {
  std::vector<uint8_t> b;
  struct Slot { int id; int size; uint64_t value; };
  void align(size_t a) { while (b.size() % a) b.push_back(0); }
  void put(size_t at, const void* p, size_t n) { std::memcpy(&b[at], p, n); }
  size_t grow(size_t n) { const size_t at = b.size(); b.resize(at + n, 0); return at; }
  // returns the table position; where[id] = absolute position of the field
  size_t table(int n_ids, const std::vector<Slot>& slots, std::vector<size_t>& where) {
    align(4);
    const size_t vsize = 4 + 2 * (size_t)n_ids;
    size_t vt = grow(vsize); align(4);
    // 8-byte fields need absolute 8-byte alignment: start the table so that (table + 4) is 8-aligned when one is present
    bool wide = false; for (auto& sl : slots) wide |= sl.size == 8;
    if (wide) while ((b.size() + 4) % 8) b.push_back(0);
    const size_t t = grow(4);
    where.assign((size_t)n_ids, 0);
    for (auto& sl : slots) {
      while ((b.size()) % (size_t)sl.size) b.push_back(0);
      const size_t at = grow((size_t)sl.size);
      put(at, &sl.value, (size_t)sl.size);
      where[(size_t)sl.id] = at;
    }
    align(4);
    const uint16_t vs = (uint16_t)vsize, ts = (uint16_t)(b.size() - t);
    put(vt, &vs, 2); put(vt + 2, &ts, 2);
    for (auto& sl : slots) { const uint16_t off = (uint16_t)(where[(size_t)sl.id] - t); put(vt + 4 + 2 * (size_t)sl.id, &off, 2); }
    const int32_t so = (int32_t)(t - vt); put(t, &so, 4);
    return t;
  }
  void point(size_t slot_at, size_t target) { const uint32_t u = (uint32_t)(target - slot_at); put(slot_at, &u, 4); }
  size_t string(const std::string& x) {
    align(4);
    const size_t at = grow(4 + x.size() + 1);
    const uint32_t n = (uint32_t)x.size(); put(at, &n, 4); if (n) put(at + 4, x.data(), n);
    align(4);
    return at;
  }
};

std::vector<uint8_t> build_schema_metadata(const gpuq_field_info* fields, int n_cols, const std::vector<std::pair<std::string, std::string>>& kv = {}) {
  FbWriter w;
  std::vector<size_t> at;
  w.grow(4);                                                                        // root uoffset
  const size_t msg = w.table(4, {{3, 8, 0}, {2, 4, 0}, {0, 2, 4 /* V5 */}, {1, 1, 1 /* MessageHeader Schema */}}, at);
  w.point(0, msg);
  const size_t msg_header = at[2];
  const size_t sch = kv.empty() ? w.table(4, {{1, 4, 0}, {0, 2, 0 /* Endianness Little */}}, at)
                                : w.table(4, {{1, 4, 0}, {2, 4, 0}, {0, 2, 0 /* Endianness Little */}}, at);
  w.point(msg_header, sch);
  const size_t sch_fields = at[1], sch_meta = kv.empty() ? 0 : at[2];
  w.align(4);
  const size_t vec = w.grow(4 + 4 * (size_t)n_cols);
  { const uint32_t n = (uint32_t)n_cols; w.put(vec, &n, 4); }
  w.point(sch_fields, vec);
  for (int c = 0; c < n_cols; ++c) {
    const gpuq_field_info& f = fields[c];
    int type_type = 0;
    switch (f.type) {
      case GPUQ_INT32: case GPUQ_INT64: case GPUQ_UINT32: case GPUQ_UINT64: case GPUQ_INT8: case GPUQ_INT16: case GPUQ_UINT8: case GPUQ_UINT16: type_type = 2; break;
      case GPUQ_FLOAT64: case GPUQ_FLOAT32: type_type = 3; break;
      case GPUQ_DATE64: type_type = 8; break;
      case GPUQ_TIMESTAMP: type_type = 10; break;
      case GPUQ_UTF8: type_type = 5; break;
      case GPUQ_BOOL: type_type = 6; break;
      case GPUQ_DECIMAL128: type_type = 7; break;
      case GPUQ_DATE32: type_type = 8; break;
      default: throw Unsupported("IPC schema for column type " + std::to_string(f.type));
    }
    const size_t ft = w.table(6, {{0, 4, 0}, {3, 4, 0}, {5, 4, 0}, {1, 1, (uint64_t)(f.nullable ? 1 : 0)}, {2, 1, (uint64_t)type_type}}, at);
    w.point(vec + 4 + 4 * (size_t)c, ft);
    const size_t a_name = at[0], a_type = at[3], a_children = at[5];
    w.point(a_name, w.string(std::string(f.name, strnlen(f.name, sizeof(f.name)))));
    std::vector<size_t> ta;
    size_t tt = 0;
    switch (f.type) {
      case GPUQ_INT32: tt = w.table(2, {{0, 4, 32}, {1, 1, 1}}, ta); break;
      case GPUQ_INT64: tt = w.table(2, {{0, 4, 64}, {1, 1, 1}}, ta); break;
      case GPUQ_UINT32: tt = w.table(2, {{0, 4, 32}, {1, 1, 0}}, ta); break;
      case GPUQ_UINT64: tt = w.table(2, {{0, 4, 64}, {1, 1, 0}}, ta); break;
      case GPUQ_INT8: tt = w.table(2, {{0, 4, 8}, {1, 1, 1}}, ta); break;
      case GPUQ_INT16: tt = w.table(2, {{0, 4, 16}, {1, 1, 1}}, ta); break;
      case GPUQ_UINT8: tt = w.table(2, {{0, 4, 8}, {1, 1, 0}}, ta); break;
      case GPUQ_UINT16: tt = w.table(2, {{0, 4, 16}, {1, 1, 0}}, ta); break;
      case GPUQ_FLOAT32: tt = w.table(1, {{0, 2, 1 /* SINGLE */}}, ta); break;
      case GPUQ_DATE64: tt = w.table(1, {{0, 2, 1 /* DateUnit MILLISECOND */}}, ta); break;
      case GPUQ_TIMESTAMP: tt = w.table(1, {{0, 2, (uint64_t)(f.precision & 3) /* TimeUnit; no timezone: the schema layer owns it */}}, ta); break;
      case GPUQ_FLOAT64: tt = w.table(1, {{0, 2, 2 /* DOUBLE */}}, ta); break;
      case GPUQ_UTF8: case GPUQ_BOOL: tt = w.table(0, {}, ta); break;
      case GPUQ_DECIMAL128: tt = w.table(3, {{0, 4, (uint64_t)f.precision}, {1, 4, (uint64_t)f.scale}, {2, 4, 128}}, ta); break;
      case GPUQ_DATE32: tt = w.table(1, {{0, 2, 0 /* DateUnit DAY */}}, ta); break;
    }
    w.point(a_type, tt);
    w.align(4);
    const size_t kids = w.grow(4);                                                 // empty children vector (Arrow C++ insists on its presence)
    w.point(a_children, kids);
  }
  if (!kv.empty()) {      // Schema.custom_metadata: [KeyValue{key, value}]
    w.align(4);
    const size_t mv = w.grow(4 + 4 * kv.size());
    { const uint32_t n = (uint32_t)kv.size(); w.put(mv, &n, 4); }
    w.point(sch_meta, mv);
    for (size_t i = 0; i < kv.size(); ++i) {
      std::vector<size_t> ka;
      const size_t t = w.table(2, {{0, 4, 0}, {1, 4, 0}}, ka);
      w.point(mv + 4 + 4 * i, t);
      const size_t a_k = ka[0], a_v = ka[1];
      w.point(a_k, w.string(kv[i].first));
      w.point(a_v, w.string(kv[i].second));
    }
  }
  w.align(8);
  return w.b;
}

int type_width_of(int type) {
  switch (type) {
    case GPUQ_INT32: case GPUQ_DATE32: case GPUQ_UINT32: case GPUQ_FLOAT32: return 4;
    case GPUQ_INT64: case GPUQ_FLOAT64: case GPUQ_UINT64: case GPUQ_TIMESTAMP: case GPUQ_DATE64: return 8;
    case GPUQ_INT8: case GPUQ_UINT8: return 1;
    case GPUQ_INT16: case GPUQ_UINT16: return 2;
    case GPUQ_DECIMAL128: return 16;
    default: return 0;
  }
}

struct SrcBuf { const uint8_t* p; int64_t len; };

struct PinnedScratch {      // small pinned area for the descriptor upload / size read-back of one call
  void* p = nullptr; size_t cap = 0;
  ~PinnedScratch() { if (p) (void)hipHostFree(p); }
  void* ensure(size_t n) { if (n > cap) { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; HIPCHECK(hipHostMalloc(&p, n, hipHostMallocDefault)); cap = n; } return p; }
};
thread_local PinnedScratch g_pin_up, g_pin_down;

}  // namespace

struct gpuq_ipc_batch {
  int64_t n_rows = 0;
  struct Col { gpuq_column col{}; DevBuf data, offsets, validity; };
  std::vector<std::unique_ptr<Col>> cols;
};

extern "C" {

const char* gpuq_ipc_last_error(void) { return g_ipc_error.c_str(); }

int gpuq_ipc_peek(const uint8_t* bytes, int64_t avail, gpuq_ipc_info* out) {
  return guarded_ipc([&]() {
    if (!bytes || !out) throw std::runtime_error("bytes/out is NULL");
    std::memset(out, 0, sizeof(*out)); out->codec = -1;
    if (avail < 8) throw Capacity("need at least 8 bytes of an encapsulated IPC message");
    uint32_t cont; int32_t mlen; std::memcpy(&cont, bytes, 4); std::memcpy(&mlen, bytes + 4, 4);
    if (cont != 0xFFFFFFFFu) throw std::runtime_error("IPC message does not start with the continuation marker (pre-0.15 framing is not supported)");
    if (mlen == 0) { out->header_type = 0; out->metadata_bytes = 8; return; }      // end-of-stream
    if (mlen < 0) throw std::runtime_error("negative IPC metadata length");
    out->metadata_bytes = 8 + (int64_t)mlen;
    if (avail < out->metadata_bytes) throw Capacity("IPC metadata is " + std::to_string(out->metadata_bytes) + " bytes");
    const BatchMeta m = parse_message(bytes + 8, (size_t)mlen);
    out->header_type = m.header_type; out->codec = m.codec; out->body_bytes = m.body_len; out->n_rows = m.n_rows;
    out->n_nodes = (int32_t)m.nodes.size(); out->n_buffers = (int32_t)m.buffers.size();
  });
}

int gpuq_ipc_schema_message(const gpuq_field_info* fields, int n_cols, uint8_t* out, int64_t cap, int64_t* len_out) {
  return guarded_ipc([&]() {
    if (!fields && n_cols > 0) throw std::runtime_error("fields is NULL");
    const std::vector<uint8_t> fb = build_schema_metadata(fields, n_cols);
    const int64_t total = 8 + (int64_t)fb.size();
    if (len_out) *len_out = total;
    if (!out || cap < total) { if (!out && cap == 0) return; throw Capacity("IPC schema message needs " + std::to_string(total) + " bytes"); }
    const uint32_t cont = 0xFFFFFFFFu; const int32_t mlen = (int32_t)fb.size();
    std::memcpy(out, &cont, 4); std::memcpy(out + 4, &mlen, 4); std::memcpy(out + 8, fb.data(), fb.size());
  });
}

int gpuq_ipc_schema_message_kv(const gpuq_field_info* fields, int n_cols, const char* const* keys, const char* const* values, int n_kv, uint8_t* out, int64_t cap,
                               int64_t* len_out) {
  return guarded_ipc([&]() {
    if (!fields && n_cols > 0) throw std::runtime_error("fields is NULL");
    if (n_kv < 0 || (n_kv > 0 && (!keys || !values))) throw std::runtime_error("keys / values are NULL");
    std::vector<std::pair<std::string, std::string>> kv;
    for (int i = 0; i < n_kv; ++i) kv.emplace_back(keys[i] ? keys[i] : "", values[i] ? values[i] : "");
    const std::vector<uint8_t> fb = build_schema_metadata(fields, n_cols, kv);
    const int64_t total = 8 + (int64_t)fb.size();
    if (len_out) *len_out = total;
    if (!out || cap < total) { if (!out && cap == 0) return; throw Capacity("IPC schema message needs " + std::to_string(total) + " bytes"); }
    const uint32_t cont = 0xFFFFFFFFu; const int32_t mlen = (int32_t)fb.size();
    std::memcpy(out, &cont, 4); std::memcpy(out + 4, &mlen, 4); std::memcpy(out + 8, fb.data(), fb.size());
  });
}

int gpuq_ipc_schema_metadata(const uint8_t* bytes, int64_t avail, const char* key, char* value_out, size_t cap, int* found_out) {
  return guarded_ipc([&]() {
    if (!bytes || !key || !found_out) throw std::runtime_error("bytes / key / found_out is NULL");
    *found_out = 0;
    if (avail < 8) throw Capacity("IPC message header needs 8 bytes");
    size_t pos = 0; uint32_t first; std::memcpy(&first, bytes, 4);
    int32_t mlen;
    if (first == 0xFFFFFFFFu) { std::memcpy(&mlen, bytes + 4, 4); pos = 8; } else { mlen = (int32_t)first; pos = 4; }      // pre-0.15 streams have no continuation marker
    if (mlen <= 0 || (int64_t)pos + mlen > avail) throw Capacity("IPC schema message is longer than the bytes given");
    FbView v{bytes + pos, (size_t)mlen};
    const size_t msg = v.indirect(0);
    size_t p = v.field(msg, 1);
    if (!p || v.u8(p) != 1) throw std::runtime_error("not a Schema message");
    p = v.field(msg, 2); if (!p) return;
    const size_t sch = v.indirect(p);
    p = v.field(sch, 2); if (!p) return;
    const size_t vec = v.indirect(p); const uint32_t cnt = v.u32(vec);
    auto str = [&](size_t at) { const size_t s0 = v.indirect(at); const uint32_t n = v.u32(s0); v.need(s0 + 4, n); return std::string((const char*)v.b + s0 + 4, n); };
    for (uint32_t i = 0; i < cnt; ++i) {
      const size_t kvt = v.indirect(vec + 4 + 4 * (size_t)i);
      const size_t kp = v.field(kvt, 0), vp = v.field(kvt, 1);
      if (!kp || str(kp) != key) continue;
      const std::string val = vp ? str(vp) : std::string();
      if (value_out && cap) std::snprintf(value_out, cap, "%s", val.c_str());
      *found_out = 1;
      return;
    }
  });
}

int gpuq_ipc_encode_batch(gpuq_ctx* ctx, void* stream, const gpuq_column* cols, int n_cols, int64_t n_rows, int codec, uint8_t* out_host, int64_t cap,
                          int64_t* len_out) {
  return guarded_ipc([&]() {
    if (!ctx) throw std::runtime_error("ctx is NULL");
    if (!cols && n_cols > 0) throw std::runtime_error("cols is NULL");
    if (codec != -1 && codec != 0) throw Unsupported("IPC body compression codec " + std::to_string(codec) + " (only LZ4_FRAME = 0 or none = -1)");
    hipStream_t s = use_stream(stream);
    BatchMeta meta; meta.header_type = 3; meta.codec = codec; meta.n_rows = n_rows;
    // ---- the Arrow buffers of every column, in IPC order (validity, [offsets], data)
    std::vector<SrcBuf> bufs;
    DevBuf counts; counts.ensure(8 * (size_t)(n_cols + 1) + 16 * (size_t)n_cols);
    HIPCHECK(hipMemsetAsync(counts.p, 0, 8 * (size_t)(n_cols + 1), s));
    std::vector<std::unique_ptr<DevBuf>> temps;
    int32_t* ends_dev = (int32_t*)((char*)counts.p + 8 * (size_t)(n_cols + 1));      // per column: first and last Utf8 offset
    bool any_utf8 = false;
    for (int c = 0; c < n_cols; ++c) {
      const gpuq_column& k = cols[c];
      if (k.length != n_rows) throw std::runtime_error("column " + std::to_string(c) + " has " + std::to_string(k.length) + " rows, batch has " + std::to_string(n_rows));
      if (k.repr != GPUQ_REPR_ARROW) throw Unsupported("IPC encode needs Arrow-layout columns (unpack PACKED15 strings with gpuq_unpack_utf8 first)");
      if (k.validity && n_rows > 0) launch_popcount_bits(s, k.validity, n_rows, (unsigned long long*)counts.p + c);
      if (k.type == GPUQ_UTF8 && n_rows > 0) {
        any_utf8 = true;
        HIPCHECK(hipMemcpyAsync(ends_dev + 2 * c, k.offsets, 4, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(ends_dev + 2 * c + 1, k.offsets + n_rows, 4, hipMemcpyDeviceToDevice, s));
      }
    }
    std::vector<unsigned long long> valid_counts((size_t)n_cols + 1, 0);
    std::vector<int32_t> ends(2 * (size_t)n_cols + 2, 0);
    {
      char* pd = (char*)g_pin_down.ensure(counts.cap);
      HIPCHECK(hipMemcpyAsync(pd, counts.p, 8 * (size_t)(n_cols + 1) + 8 * (size_t)n_cols, hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      std::memcpy(valid_counts.data(), pd, 8 * (size_t)n_cols);
      if (any_utf8) std::memcpy(ends.data(), pd + 8 * (size_t)(n_cols + 1), 8 * (size_t)n_cols);
    }
    for (int c = 0; c < n_cols; ++c) {
      const gpuq_column& k = cols[c];
      const int64_t nulls = (k.validity && n_rows > 0) ? n_rows - (int64_t)valid_counts[(size_t)c] : 0;
      meta.nodes.push_back({n_rows, nulls});
      bufs.push_back({nulls > 0 ? k.validity : nullptr, nulls > 0 ? (n_rows + 7) / 8 : 0});
      if (k.type == GPUQ_UTF8) {
        const int32_t first = ends[2 * (size_t)c], last = ends[2 * (size_t)c + 1];
        const int32_t* offs = k.offsets;
        if (first != 0 && n_rows > 0) {       // a sliced column: offsets are rebased so that the data buffer starts at the first string
          temps.push_back(std::make_unique<DevBuf>()); temps.back()->ensure((size_t)(n_rows + 1) * 4);
          launch_offsets_rebase(s, k.offsets, n_rows + 1, -first, temps.back()->as<int32_t>());
          offs = temps.back()->as<int32_t>();
        }
        if (n_rows == 0) {                     // one zero offset
          temps.push_back(std::make_unique<DevBuf>()); temps.back()->ensure(16); HIPCHECK(hipMemsetAsync(temps.back()->p, 0, 16, s));
          offs = temps.back()->as<int32_t>();
        }
        bufs.push_back({(const uint8_t*)offs, (n_rows + 1) * 4});
        bufs.push_back({(const uint8_t*)k.data + first, (int64_t)last - first});
      } else if (k.type == GPUQ_BOOL) {
        bufs.push_back({(const uint8_t*)k.data, (n_rows + 7) / 8});
      } else {
        const int w = type_width_of(k.type);
        if (!w) throw Unsupported("IPC encode of column type " + std::to_string(k.type));
        bufs.push_back({(const uint8_t*)k.data, n_rows * w});
      }
    }
    const int nb = (int)bufs.size();
    std::vector<int64_t> boff((size_t)nb + 1, 0), blen((size_t)nb, 0);
    DevBuf body, slots, desc, sizes;
    if (codec < 0) {
      int64_t off = 0;
      for (int j = 0; j < nb; ++j) { boff[(size_t)j] = off; blen[(size_t)j] = bufs[(size_t)j].len; off += (bufs[(size_t)j].len + 7) & ~(int64_t)7; }
      boff[(size_t)nb] = off;
    } else {
      // ---- blocks of <= 64 KiB, one wave each
      std::vector<Lz4Block> blocks; std::vector<int32_t> blk_buffer, first((size_t)nb + 1, 0);
      int64_t slot = 0, worst = 0;
      for (int j = 0; j < nb; ++j) {
        first[(size_t)j] = (int32_t)blocks.size();
        for (int64_t at = 0; at < bufs[(size_t)j].len; at += LZ4_BLOCK_BYTES) {
          const int64_t l = std::min<int64_t>(LZ4_BLOCK_BYTES, bufs[(size_t)j].len - at);
          blocks.push_back({(int64_t)(uintptr_t)(bufs[(size_t)j].p + at), slot, (int32_t)l, 0});
          blk_buffer.push_back(j); slot += lz4_slot_bytes(l);
        }
        if (bufs[(size_t)j].len > 0) worst += (bufs[(size_t)j].len + 8 + 7) & ~(int64_t)7;
      }
      first[(size_t)nb] = (int32_t)blocks.size();
      const int nblk = (int)blocks.size();
      if (nblk > 0) {
        const size_t b_blocks = sizeof(Lz4Block) * (size_t)nblk, b_bb = 4 * (size_t)nblk, b_first = 4 * ((size_t)nb + 1);
        const size_t o_bb = (b_blocks + 15) & ~(size_t)15, o_first = (o_bb + b_bb + 15) & ~(size_t)15, up_total = o_first + b_first;
        char* up = (char*)g_pin_up.ensure(up_total);
        std::memcpy(up, blocks.data(), b_blocks); std::memcpy(up + o_bb, blk_buffer.data(), b_bb); std::memcpy(up + o_first, first.data(), b_first);
        desc.ensure(up_total);
        HIPCHECK(hipMemcpyAsync(desc.p, up, up_total, hipMemcpyHostToDevice, s));
        // sizes: csize[nblk] i32 | blk_dst[nblk] i64 | buf_off[nb+1] i64 | buf_len[nb] i64
        const size_t o_dst = (4 * (size_t)nblk + 15) & ~(size_t)15, o_off = o_dst + 8 * (size_t)nblk, o_len = o_off + 8 * ((size_t)nb + 1), sz_total = o_len + 8 * (size_t)nb;
        sizes.ensure(sz_total);
        slots.ensure((size_t)slot + 16);
        body.ensure((size_t)worst + 16);
        const Lz4Block* d_blocks = (const Lz4Block*)desc.p;
        const int32_t* d_bb = (const int32_t*)((char*)desc.p + o_bb); const int32_t* d_first = (const int32_t*)((char*)desc.p + o_first);
        int32_t* d_csize = (int32_t*)sizes.p; int64_t* d_dst = (int64_t*)((char*)sizes.p + o_dst);
        int64_t* d_off = (int64_t*)((char*)sizes.p + o_off); int64_t* d_len = (int64_t*)((char*)sizes.p + o_len);
        launch_lz4_compress(s, nullptr, slots.as<uint8_t>(), d_blocks, nblk, d_csize);
        launch_lz4_layout(s, d_blocks, d_csize, d_first, nb, d_off, d_len, d_dst);
        launch_lz4_pack(s, nullptr, slots.as<uint8_t>(), d_blocks, nblk, d_csize, d_bb, d_first, d_off, d_len, d_dst, body.as<uint8_t>());
        char* pd = (char*)g_pin_down.ensure(8 * (2 * (size_t)nb + 1));
        HIPCHECK(hipMemcpyAsync(pd, d_off, 8 * (2 * (size_t)nb + 1), hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        std::memcpy(boff.data(), pd, 8 * ((size_t)nb + 1));
        std::memcpy(blen.data(), pd + 8 * ((size_t)nb + 1), 8 * (size_t)nb);
        for (auto& l : blen) if (l < 0) l = -l;
        if (boff[(size_t)nb] > worst) throw std::runtime_error("internal: compressed body larger than its bound");
      }
    }
    meta.body_len = boff[(size_t)nb];
    for (int j = 0; j < nb; ++j) meta.buffers.push_back({boff[(size_t)j], blen[(size_t)j]});
    const std::vector<uint8_t> fb = build_batch_metadata(meta);
    const int64_t total = 8 + (int64_t)fb.size() + meta.body_len;
    if (len_out) *len_out = total;
    if (!out_host || cap < total) {
      if (!out_host && cap == 0) return;      // size query
      throw Capacity("IPC message needs " + std::to_string(total) + " bytes");
    }
    const uint32_t cont = 0xFFFFFFFFu; const int32_t mlen = (int32_t)fb.size();
    std::memcpy(out_host, &cont, 4); std::memcpy(out_host + 4, &mlen, 4); std::memcpy(out_host + 8, fb.data(), fb.size());
    uint8_t* hb = out_host + 8 + fb.size();
    if (codec < 0) {
      for (int j = 0; j < nb; ++j) {
        const int64_t l = bufs[(size_t)j].len;
        if (l > 0) HIPCHECK(hipMemcpyAsync(hb + boff[(size_t)j], bufs[(size_t)j].p, (size_t)l, hipMemcpyDeviceToHost, s));
        std::memset(hb + boff[(size_t)j] + l, 0, (size_t)(((l + 7) & ~(int64_t)7) - l));
      }
    } else if (meta.body_len > 0) {
      HIPCHECK(hipMemcpyAsync(hb, body.p, (size_t)meta.body_len, hipMemcpyDeviceToHost, s));
    }
    HIPCHECK(hipStreamSynchronize(s));
  });
}

// All RecordBatch messages of a byte range decoded in ONE pass: every LZ4 block (independent frames) or frame (linked blocks)
// of every buffer of every batch is a unit of the same launch, and each lands at its final place in the concatenated column.
// That is what makes the reference's 8192-row batches (one or two blocks per buffer) fill the chip.
int gpuq_ipc_decode_stream(gpuq_ctx* ctx, void* stream, const uint8_t* bytes, int64_t n_bytes, const gpuq_field_info* fields, int n_cols, gpuq_ipc_batch** out) {
  gpuq_ipc_batch* B = nullptr;
  int rc = guarded_ipc([&]() {
    if (!ctx) throw std::runtime_error("ctx is NULL");
    if (!bytes || !out || (!fields && n_cols > 0)) throw std::runtime_error("bytes/fields/out is NULL");
    hipStream_t s = use_stream(stream);
    // ---- host pass over the message headers
    struct Msg { BatchMeta m; int64_t body; int64_t row0; };
    std::vector<Msg> msgs;
    size_t want = 0;
    for (int c = 0; c < n_cols; ++c) want += fields[c].type == GPUQ_UTF8 ? 3 : 2;
    int64_t pos = 0, N = 0;
    while (pos + 8 <= n_bytes) {
      uint32_t cont; int32_t mlen; std::memcpy(&cont, bytes + pos, 4); std::memcpy(&mlen, bytes + pos + 4, 4);
      if (cont != 0xFFFFFFFFu) throw std::runtime_error("IPC stream: message at byte " + std::to_string(pos) + " does not start with the continuation marker");
      if (mlen == 0) break;                                          // end-of-stream marker
      if (mlen < 0 || pos + 8 + (int64_t)mlen > n_bytes) throw std::runtime_error("truncated IPC message metadata");
      Msg g; g.m = parse_message(bytes + pos + 8, (size_t)mlen); g.body = pos + 8 + mlen; g.row0 = N;
      if (g.m.body_len < 0 || g.body + g.m.body_len > n_bytes) throw std::runtime_error("truncated IPC body");
      pos = g.body + g.m.body_len;
      if (g.m.header_type == 1) continue;                            // Schema: the caller passed the fields
      if (g.m.header_type != 3) throw Unsupported("IPC message type " + std::to_string(g.m.header_type) + " (dictionary batches are not supported on device)");
      if (g.m.codec > 0) throw Unsupported("IPC body compression codec " + std::to_string(g.m.codec) + " (ZSTD) is not supported on device");
      if ((int)g.m.nodes.size() != n_cols) throw std::runtime_error("IPC batch has " + std::to_string(g.m.nodes.size()) + " field nodes, schema has " + std::to_string(n_cols));
      if (g.m.buffers.size() != want) throw std::runtime_error("IPC batch has " + std::to_string(g.m.buffers.size()) + " buffers, schema needs " + std::to_string(want));
      for (auto& bf : g.m.buffers) if (bf.first < 0 || bf.second < 0 || bf.first + bf.second > g.m.body_len) throw std::runtime_error("IPC buffer outside the body");
      for (auto& nd : g.m.nodes) if (nd.first != g.m.n_rows) throw std::runtime_error("IPC field node length differs from the batch length");
      N += g.m.n_rows;
      msgs.push_back(std::move(g));
    }
    B = new gpuq_ipc_batch(); B->n_rows = N;
    DevBuf dsrc;
    dsrc.ensure((size_t)pos + 16);
    if (pos > 0) HIPCHECK(hipMemcpyAsync(dsrc.p, bytes, (size_t)pos, hipMemcpyHostToDevice, s));
    const uint8_t* db = dsrc.as<uint8_t>();
    std::vector<Lz4Unit> units, units_seq;      // fast path (independent blocks assumed full) and the always-valid per-frame walk
    bool split_any = false;
    // linked frames of two blocks or more (what Arrow C++ writes: every block may copy from the 64 KB before it) are decoded WITHOUT the
    // serial walk (kernels_lz4.hip, launch_unpack_pages_pj: the machinery of the Snappy pages with LZ4's sequences as elements and the
    // frame as the space copies point into) -- one job per block; the per-frame walk stays as the fallback (units_seq)
    std::vector<UnpackJob> ljobs; int64_t lwords = 0, lslots = 0, lmax_frame = 0, lmax_in = 0;
    static const bool lz4_pj_on = []() { const char* e = std::getenv("GPUQ_LZ4_PJ"); return !(e && e[0] == '0'); }();
    // uncompressed length of buffer j of message g (host only)
    struct CopyFix { uint8_t* dst; const uint8_t* src; int64_t n; };       // device-to-device copies, in issue order
    std::vector<CopyFix> copyfix;
    auto ulen_of = [&](const Msg& g, size_t j) -> int64_t {
      const int64_t len = g.m.buffers[j].second;
      if (len == 0 || g.m.codec < 0) return len;
      if (len < 8) throw std::runtime_error("compressed IPC buffer shorter than its length prefix");
      int64_t u; std::memcpy(&u, bytes + g.body + g.m.buffers[j].first, 8);
      if (u == -1) return len - 8;
      if (u < 0) throw std::runtime_error("bad uncompressed length in an IPC buffer");
      return u;
    };
    // queue the decode of buffer j of message g into dst (capacity dst_cap bytes)
    auto emit = [&](const Msg& g, size_t j, uint8_t* dst, int64_t dst_cap) {
      const int64_t off = g.body + g.m.buffers[j].first, len = g.m.buffers[j].second;
      if (len == 0) return;
      const int64_t ulen = ulen_of(g, j);
      if (ulen > dst_cap) throw std::runtime_error("IPC buffer " + std::to_string(j) + " holds " + std::to_string(ulen) + " bytes, its column allows " + std::to_string(dst_cap));
      if (ulen == 0) return;
      // raw bytes (an uncompressed batch, or a buffer stored with the -1 length prefix): recorded, not issued -- run() replays the
      // list on every attempt, so the frame-by-frame fallback (which clears the OR-merged bitmaps first) sees them again
      if (g.m.codec < 0) { copyfix.push_back({dst, db + off, len}); return; }
      int64_t pre; std::memcpy(&pre, bytes + off, 8);
      if (pre == -1) { copyfix.push_back({dst, db + off + 8, len - 8}); return; }
      // LZ4 frame header
      const uint8_t* f = bytes + off + 8; const int64_t flen = len - 8;
      if (flen < 7 + 4) throw std::runtime_error("truncated LZ4 frame");
      uint32_t magic; std::memcpy(&magic, f, 4);
      if (magic != 0x184D2204u) throw std::runtime_error("IPC buffer is not an LZ4 frame (magic " + std::to_string(magic) + ")");
      const uint8_t flg = f[4], bd = f[5];
      if ((flg >> 6) != 1) throw std::runtime_error("unsupported LZ4 frame version");
      const bool indep = (flg >> 5) & 1, bchk = (flg >> 4) & 1, has_size = (flg >> 3) & 1, has_dict = flg & 1;
      if (has_dict) throw Unsupported("LZ4 frame with a dictionary id");
      const int bmax_id = (bd >> 4) & 7;
      if (bmax_id < 4) throw std::runtime_error("bad LZ4 frame block size id");
      const int64_t bmax = (int64_t)1 << (8 + 2 * bmax_id);
      const int64_t hp = 6 + (has_size ? 8 : 0) + 1;
      if (hp + 4 > flen) throw std::runtime_error("truncated LZ4 frame");
      Lz4Unit whole{off + 8 + hp, (int64_t)(uintptr_t)dst, flen - hp, ulen, LZ4_UNIT_FRAME_BLOCKS, bchk ? LZ4_UNIT_BLOCK_CHECKSUM : 0};
      units_seq.push_back(whole);
      if (!indep) {
        // the block index, on the host (as for independent blocks below): every block but the last decodes to the frame's block size
        std::vector<UnpackJob> fj; int64_t ip = hp, op = 0; bool okw = lz4_pj_on && ulen > bmax && ulen < ((int64_t)1 << 30);
        int64_t words0 = (lwords + 3) & ~(int64_t)3, slots = lslots, max_in = 0;
        while (okw) {
          if (ip + 4 > flen) { okw = false; break; }
          uint32_t h; std::memcpy(&h, f + ip, 4); ip += 4;
          if (h == 0) break;
          const int64_t bs = h & 0x7FFFFFFFu;
          if (bs > flen - ip || op >= ulen || bs == 0) { okw = false; break; }
          const int64_t dl = std::min(bmax, ulen - op);
          const bool stored = (h & 0x80000000u) != 0;
          UnpackJob j{off + 8 + ip, (int64_t)(uintptr_t)dst + op, bs, dl, 0, stored ? 3 : 2, 0, words0 + op, slots, words0, op};
          fj.push_back(j);
          if (!stored) { slots += bs + 1; if (bs > max_in) max_in = bs; }
          ip += bs + (bchk ? 4 : 0); op += dl;
        }
        if (okw && op == ulen && fj.size() >= 2 && words0 + ulen < ((int64_t)1 << 30) && slots < ((int64_t)1 << 30)) {
          for (auto& j : fj) ljobs.push_back(j);
          lwords = words0 + ulen; lslots = slots; lmax_frame = std::max(lmax_frame, ulen); lmax_in = std::max(lmax_in, max_in);
          return;
        }
        units.push_back(whole); return;
      }
      // independent blocks: walk the block index on the host, one unit per block (all but the last assumed full; verified on the device)
      int64_t ip = hp, op = 0; const size_t first_unit = units.size(); bool okw = true;
      while (true) {
        if (ip + 4 > flen) { okw = false; break; }
        uint32_t h; std::memcpy(&h, f + ip, 4); ip += 4;
        if (h == 0) break;
        const int64_t bs = h & 0x7FFFFFFFu;
        if (bs > flen - ip || op >= ulen) { okw = false; break; }
        const int64_t dl = std::min(bmax, ulen - op);
        units.push_back({off + 8 + ip, (int64_t)(uintptr_t)dst + op, bs, dl, (h & 0x80000000u) ? LZ4_UNIT_STORED : LZ4_UNIT_BLOCK, 0});
        ip += bs + (bchk ? 4 : 0); op += dl;
      }
      if (!okw || op != ulen) { units.resize(first_unit); units.push_back(whole); }
      else if (units.size() - first_unit > 1) split_any = true;
    };
    struct BitFix { uint8_t* dst; int64_t bit0; const uint8_t* src; int64_t n; };          // after the decode: dst bits [bit0, +n) |= src (NULL: ones)
    struct Utf8Col { int col; std::vector<Utf8Piece> pieces; int64_t max_rows, total; };
    std::vector<Utf8Col> utf8_cols;
    std::vector<Utf8Piece> checks; int64_t max_check_rows = 0;      // offsets decoded in place (one message): validated like the pieces
    std::vector<BitFix> bitfix;
    std::vector<std::unique_ptr<DevBuf>> temps;
    auto temp = [&](size_t nbytes) -> uint8_t* { temps.push_back(std::make_unique<DevBuf>()); temps.back()->ensure(nbytes + 80); return temps.back()->as<uint8_t>(); };
    // one temp area per kind, carved per batch (a DevBuf per batch would be thousands of pool round trips)
    const size_t vbytes = (size_t)((N + 63) / 64) * 8;
    // a bitmap piece lands in place when it starts on a byte and either ends on one or is the last piece; everything else goes
    // through a temp and an OR-merge (which also masks the producer's padding bits)
    auto bitmap_piece = [&](const Msg& g, size_t j, uint8_t* dst, bool last, uint8_t*& scratch) {
      const int64_t n = g.m.n_rows;
      if (n == 0) return;
      if (g.m.buffers[j].second == 0) { bitfix.push_back({dst, g.row0, nullptr, n}); return; }       // no buffer: all ones
      const int64_t cap = (n + 7) / 8 + 64;
      (void)last;
      if ((g.row0 & 7) == 0 && (n & 7) == 0 && ulen_of(g, j) == n / 8) { emit(g, j, dst + g.row0 / 8, n / 8); return; }      // exact fit: concurrent pieces never overlap
      uint8_t* t = scratch; scratch += (cap + 15) & ~(int64_t)15;
      emit(g, j, t, cap);
      bitfix.push_back({dst, g.row0, t, n});
    };
    size_t bit_scratch = 0, off_scratch = 0;
    for (auto& g : msgs) { bit_scratch += (size_t)(((g.m.n_rows + 7) / 8 + 64 + 15) & ~(int64_t)15); off_scratch += (size_t)((g.m.n_rows + 1) * 4 + 64 + 15) & ~(size_t)15; }
    size_t j0 = 0;
    for (int c = 0; c < n_cols; ++c) {
      auto col = std::make_unique<gpuq_ipc_batch::Col>();
      const gpuq_field_info& f = fields[c];
      col->col.type = f.type; col->col.precision = f.precision; col->col.scale = f.scale; col->col.repr = GPUQ_REPR_ARROW; col->col.length = N;
      bool any_nulls = false;
      for (auto& g : msgs) any_nulls |= g.m.nodes[(size_t)c].second > 0 && g.m.buffers[j0].second > 0;
      if (any_nulls) {
        col->validity.ensure(vbytes + 80); HIPCHECK(hipMemsetAsync(col->validity.p, 0, vbytes + 80, s));
        uint8_t* scratch = temp(bit_scratch); HIPCHECK(hipMemsetAsync(scratch, 0, bit_scratch, s));
        for (size_t k = 0; k < msgs.size(); ++k) {
          const Msg& g = msgs[k];
          if (g.m.nodes[(size_t)c].second > 0 && g.m.buffers[j0].second > 0) bitmap_piece(g, j0, col->validity.as<uint8_t>(), k + 1 == msgs.size(), scratch);
          else if (g.m.n_rows > 0) bitfix.push_back({col->validity.as<uint8_t>(), g.row0, nullptr, g.m.n_rows});
        }
        col->col.validity = col->validity.as<uint8_t>();
      }
      if (f.type == GPUQ_UTF8) {
        int64_t total = 0;
        for (auto& g : msgs) total += ulen_of(g, j0 + 2);
        if (total > 0x7FFFFFFF) throw Unsupported("Utf8 column of more than 2 GiB in one shuffle partition (int32 offsets)");
        col->offsets.ensure((size_t)(N + 1) * 4 + 80); HIPCHECK(hipMemsetAsync(col->offsets.p, 0, (size_t)(N + 1) * 4 + 80, s));
        col->data.ensure((size_t)total + 80);
        uint8_t* scratch = msgs.size() > 1 ? temp(off_scratch) : nullptr;
        if (scratch) HIPCHECK(hipMemsetAsync(scratch, 0, off_scratch, s));
        int64_t at = 0, max_piece_rows = 0;
        std::vector<Utf8Piece> pieces;
        for (auto& g : msgs) {
          const int64_t n = g.m.n_rows, cap = (n + 1) * 4 + 64;
          if (msgs.size() == 1) { emit(g, j0 + 1, col->offsets.as<uint8_t>(), cap); checks.push_back({col->offsets.as<int32_t>(), n, nullptr, 0, ulen_of(g, j0 + 2), 0, 0, 0, 0}); max_check_rows = std::max(max_check_rows, n); }
          else if (n > 0) {
            uint8_t* t = scratch; scratch += (cap + 15) & ~(int64_t)15;
            emit(g, j0 + 1, t, cap);
            pieces.push_back({(const int32_t*)t, n, col->offsets.as<int32_t>() + g.row0, at, ulen_of(g, j0 + 2), 0, 0, 0, 0});
            max_piece_rows = std::max(max_piece_rows, n);
          }
          const int64_t dl = ulen_of(g, j0 + 2);
          emit(g, j0 + 2, col->data.as<uint8_t>() + at, dl);
          at += dl;
        }
        col->col.offsets = col->offsets.as<int32_t>();
        if (!pieces.empty()) { utf8_cols.push_back({(int)B->cols.size(), std::move(pieces), max_piece_rows, total}); }
        j0 += 3;
      } else if (f.type == GPUQ_BOOL) {
        col->data.ensure(vbytes + 80); HIPCHECK(hipMemsetAsync(col->data.p, 0, vbytes + 80, s));
        uint8_t* scratch = temp(bit_scratch); HIPCHECK(hipMemsetAsync(scratch, 0, bit_scratch, s));
        for (size_t k = 0; k < msgs.size(); ++k) bitmap_piece(msgs[k], j0 + 1, col->data.as<uint8_t>(), k + 1 == msgs.size(), scratch);
        j0 += 2;
      } else {
        const int w = type_width_of(f.type);
        if (!w) throw Unsupported("IPC decode of column type " + std::to_string(f.type));
        col->data.ensure((size_t)N * w + 80);
        for (size_t k = 0; k < msgs.size(); ++k) {
          const Msg& g = msgs[k];
          const int64_t need = g.m.n_rows * w;
          if (ulen_of(g, j0 + 1) < need) throw std::runtime_error("IPC data buffer of column " + std::to_string(c) + " has " + std::to_string(ulen_of(g, j0 + 1)) + " bytes, expected " + std::to_string(need));
          const int64_t ul = ulen_of(g, j0 + 1);
          // a producer's padding beyond the rows may only spill into the slack at the end of the column; elsewhere the pieces of
          // one launch must not overlap: a padded piece is decoded aside and its rows copied in afterwards
          if (ul == need || (k + 1 == msgs.size() && ul <= need + 64)) emit(g, j0 + 1, col->data.as<uint8_t>() + g.row0 * w, ul);
          else if (need > 0) { uint8_t* t = temp((size_t)ul); emit(g, j0 + 1, t, ul); copyfix.push_back({col->data.as<uint8_t>() + g.row0 * w, t, need}); }
        }
        j0 += 2;
      }
      col->col.data = col->data.p;
      B->cols.push_back(std::move(col));
    }
    DevBuf dunits, dstatus, dpieces; dstatus.ensure(16);
    DevBuf pj_resolve, pj_blk, pj_cblk, pj_jobs, pj_cnt, pj_ja, pj_jb, pj_olen, pj_mark, pj_scan;
    // Utf8 piece descriptors of all columns in one upload
    size_t n_pieces = checks.size(); for (auto& u : utf8_cols) { n_pieces += u.pieces.size(); max_check_rows = std::max(max_check_rows, u.max_rows); }
    std::vector<size_t> piece0;
    if (n_pieces) {
      dpieces.ensure(sizeof(Utf8Piece) * n_pieces);
      std::vector<Utf8Piece> all; all.reserve(n_pieces);
      for (auto& u : utf8_cols) { piece0.push_back(all.size()); all.insert(all.end(), u.pieces.begin(), u.pieces.end()); }
      all.insert(all.end(), checks.begin(), checks.end());
      // (pageable source: the copy is staged before the call returns)
      HIPCHECK(hipMemcpyAsync(dpieces.p, all.data(), sizeof(Utf8Piece) * n_pieces, hipMemcpyHostToDevice, s));
      HIPCHECK(hipStreamSynchronize(s));
    }
    bool gap = false;
    auto run = [&](const std::vector<Lz4Unit>& us) -> uint32_t {
      HIPCHECK(hipMemsetAsync(dstatus.p, 0, 8, s));
      if (!us.empty()) {
        char* up = (char*)g_pin_up.ensure(sizeof(Lz4Unit) * us.size());
        std::memcpy(up, us.data(), sizeof(Lz4Unit) * us.size());
        dunits.ensure(sizeof(Lz4Unit) * us.size());
        HIPCHECK(hipMemcpyAsync(dunits.p, up, sizeof(Lz4Unit) * us.size(), hipMemcpyHostToDevice, s));
        launch_lz4_decode(s, db, pos, nullptr, (const Lz4Unit*)dunits.p, (int)us.size(), dstatus.as<uint32_t>());
      }
      if (&us == &units && !ljobs.empty()) {      // (the fallback list holds those frames as whole units)
        std::vector<uint2> blk, cblk;
        for (size_t k = 0; k < ljobs.size(); ++k) {
          for (int64_t bb = 0; bb < ljobs[k].dst_len; bb += 4096) blk.push_back(make_uint2((uint32_t)k, (uint32_t)bb));
          if (ljobs[k].mode == 2) for (int64_t bb = 0; bb <= ljobs[k].src_len; bb += 4096) cblk.push_back(make_uint2((uint32_t)k, (uint32_t)bb));
        }
        int rounds = 1; while (((int64_t)1 << rounds) < lmax_frame) ++rounds;
        rounds += 1;
        int mrounds = 1; while (((int64_t)1 << mrounds) < lmax_in + 1) ++mrounds;
        mrounds += 1;
        const size_t nj = ljobs.size();
        pj_resolve.ensure((size_t)lwords * 4 + 64); pj_blk.ensure(blk.size() * sizeof(uint2) + 64); pj_cblk.ensure(cblk.size() * sizeof(uint2) + 64); pj_jobs.ensure(nj * sizeof(UnpackJob) + 64);
        pj_cnt.ensure((size_t)(rounds + mrounds + 3) * nj * 4 + 64);
        pj_ja.ensure((size_t)(lslots + 1) * 4 + 64); pj_jb.ensure((size_t)(lslots + 1) * 4 + 64); pj_olen.ensure((size_t)(lslots + 1) * 4 + 64); pj_mark.ensure((size_t)lslots + 64);
        const size_t swb = exclusive_scan_ws_bytes(lslots + 1); pj_scan.ensure(swb);
        HIPCHECK(hipMemcpyAsync(pj_jobs.p, ljobs.data(), nj * sizeof(UnpackJob), hipMemcpyHostToDevice, s));
        HIPCHECK(hipMemcpyAsync(pj_blk.p, blk.data(), blk.size() * sizeof(uint2), hipMemcpyHostToDevice, s));
        if (!cblk.empty()) HIPCHECK(hipMemcpyAsync(pj_cblk.p, cblk.data(), cblk.size() * sizeof(uint2), hipMemcpyHostToDevice, s));
        HIPCHECK(hipMemsetAsync(pj_cnt.p, 0, (size_t)(rounds + mrounds + 3) * nj * 4, s));
        HIPCHECK(hipMemsetAsync(pj_mark.p, 0, (size_t)lslots + 64, s));
        SnappyPjBuffers PB{};
        PB.resolve = (uint32_t*)pj_resolve.p; PB.blkmap = (const uint2*)pj_blk.p; PB.n_blocks = (int)blk.size(); PB.rounds = rounds;
        PB.jump_a = (uint32_t*)pj_ja.p; PB.jump_b = (uint32_t*)pj_jb.p; PB.olen = (uint32_t*)pj_olen.p; PB.mark = (uint8_t*)pj_mark.p; PB.cmap = (const uint2*)pj_cblk.p; PB.n_cblocks = (int)cblk.size();
        PB.mark_rounds = mrounds; PB.c_slots = lslots; PB.scan_ws = pj_scan.p; PB.scan_ws_bytes = swb; PB.counts = (uint32_t*)pj_cnt.p;
        launch_unpack_pages_pj(s, db, nullptr, (const UnpackJob*)pj_jobs.p, (int)nj, PB, dstatus.as<uint32_t>());
        HIPCHECK(hipStreamSynchronize(s));      // the block maps are pageable host memory
      }
      for (auto& x : copyfix) HIPCHECK(hipMemcpyAsync(x.dst, x.src, (size_t)x.n, hipMemcpyDeviceToDevice, s));
      for (auto& x : bitfix) launch_concat_bitmap(s, (u64*)x.dst, x.bit0, x.src, 0, x.n);
      // untrusted offsets: monotone, from >= 0, ending inside their data buffer -- before anything is derived from them
      launch_utf8_piece_validate(s, (const Utf8Piece*)dpieces.p, (int)n_pieces, max_check_rows, dstatus.as<uint32_t>());
      for (size_t k = 0; k < utf8_cols.size(); ++k) {
        Utf8Piece* dp = (Utf8Piece*)dpieces.p + piece0[k];
        launch_utf8_piece_starts(s, dp, (int)utf8_cols[k].pieces.size(), dstatus.as<uint32_t>() + 1);
        launch_utf8_piece_offsets(s, dp, (int)utf8_cols[k].pieces.size(), utf8_cols[k].max_rows);
      }
      uint32_t* pd = (uint32_t*)g_pin_down.ensure(16);
      HIPCHECK(hipMemcpyAsync(pd, dstatus.p, 8, hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      gap = pd[1] != 0;
      return pd[0];
    };
    HIPCHECK(hipMemsetAsync(dstatus.p, 0, 16, s));
    uint32_t st = run(units);
    if (st && (split_any || !ljobs.empty())) {       // a frame whose inner blocks are not full-size: decode frame by frame (bitmaps are OR-merged: clear them first)
      for (auto& c : B->cols) {
        if (c->col.validity) HIPCHECK(hipMemsetAsync(c->validity.p, 0, vbytes + 80, s));
        if (c->col.type == GPUQ_BOOL) HIPCHECK(hipMemsetAsync(c->data.p, 0, vbytes + 80, s));
      }
      HIPCHECK(hipMemsetAsync(dstatus.p, 0, 16, s));
      st = run(units_seq);
    }
    if (st & 4u) throw std::runtime_error("malformed Utf8 offsets in an IPC buffer (not monotone, negative, or beyond the data buffer)");
    if (st) throw std::runtime_error("malformed LZ4 data in an IPC buffer");
    if (gap) {                   // some producer padded its string data: move every column's bytes to where the merged offsets point
      for (size_t k = 0; k < utf8_cols.size(); ++k) {
        auto& col = *B->cols[(size_t)utf8_cols[k].col];
        DevBuf packed; packed.ensure((size_t)utf8_cols[k].total + 80);
        int64_t max_bytes = 0; for (auto& pc : utf8_cols[k].pieces) max_bytes = std::max(max_bytes, pc.dl);
        launch_utf8_piece_compact(s, (const Utf8Piece*)dpieces.p + piece0[k], (int)utf8_cols[k].pieces.size(), max_bytes, col.data.as<uint8_t>(), packed.as<uint8_t>());
        std::swap(col.data.p, packed.p); std::swap(col.data.cap, packed.cap);
        col.col.data = col.data.p;
      }
      HIPCHECK(hipStreamSynchronize(s));
    }
    *out = B;
  });
  if (rc != GPUQ_OK) { delete B; if (out) *out = nullptr; }
  return rc;
}

int gpuq_ipc_decode_batch(gpuq_ctx* ctx, void* stream, const uint8_t* msg, int64_t msg_bytes, const gpuq_field_info* fields, int n_cols, gpuq_ipc_batch** out) {
  gpuq_ipc_info info;
  int rc = gpuq_ipc_peek(msg, msg_bytes, &info);
  if (rc != GPUQ_OK) return rc;
  if (info.header_type != 3) { g_ipc_error = "IPC message is not a RecordBatch (header type " + std::to_string(info.header_type) + ")"; return GPUQ_ERR_INVALID; }
  if (info.metadata_bytes + info.body_bytes > msg_bytes) { g_ipc_error = "truncated IPC body"; return GPUQ_ERR_INVALID; }
  return gpuq_ipc_decode_stream(ctx, stream, msg, info.metadata_bytes + info.body_bytes, fields, n_cols, out);
}

int64_t gpuq_ipc_batch_num_rows(const gpuq_ipc_batch* b) { return b ? b->n_rows : 0; }
int gpuq_ipc_batch_num_columns(const gpuq_ipc_batch* b) { return b ? (int)b->cols.size() : 0; }
int gpuq_ipc_batch_column(const gpuq_ipc_batch* b, int i, gpuq_column* col_out) {
  if (!b || !col_out || i < 0 || i >= (int)b->cols.size()) return GPUQ_ERR_INVALID;
  *col_out = b->cols[(size_t)i]->col;
  return GPUQ_OK;
}
void gpuq_ipc_batch_free(gpuq_ipc_batch* b) { delete b; }

}  // extern "C"
