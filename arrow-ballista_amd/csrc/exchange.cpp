// Exchange between the GPUs of one node, under the C ABI (include/gpuq.h, "exchange").
//
// What it replaces in the reference: a stage boundary = ShuffleWriterExec hash-partitioning its input into Arrow-IPC files
// (ballista/core/src/execution_plans/shuffle_writer.rs:328-392) that the next stage's ShuffleReaderExec fetches, locally or
// over Flight (shuffle_reader.rs:226-298).  Between executors that each own one GPU of a node the same data movement is one
// exchange step: (1) an all-to-all of per-destination counts, (2) one variable-size all-to-all per column buffer, posted as
// grouped point-to-point sends / receives so that all of a GPU's xGMI links carry traffic at once (xGMI is point-to-point:
// a ring would be bound by one link).  Nothing is compressed or written to disk.
//
// Transports: RCCL (dlopen'ed: the library still loads where it is absent; ncclSend / ncclRecv inside a group) -- the measured
// configuration, one process per GPU -- and a host-staged transport behind a caller-supplied all-to-all callback (tests with
// several ranks on one GPU; any host that moves bytes itself, e.g. over the executors' Flight service).
//
// Host logic + data movement only: packing rows by destination is the caller's job (gpuq_partition_run + a take, as the
// native plan executor's RepartitionExec does); bitmaps are cut and re-joined at bit granularity with the library's own kernels.
#include "gpuq_internal.h"
#include "gpuq_kernels.h"
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <numeric>

using namespace gpuq;

namespace {

thread_local std::string g_xerr;

// ---- RCCL, loaded on first use (rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer, ncclUniqueId = 128 bytes)
struct NcclId { char internal[128]; };
struct Rccl {
  void* h = nullptr; bool ok = false;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    // an instance the process already has (PyTorch bundles its own librccl.so) is reused: two RCCL runtimes in one process
    // would each build their own topology and IPC state
    for (const char* n : {"librccl.so", "librccl.so.1"}) { r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.h, "ncclGroupEnd");
    r.Send = (decltype(r.Send))dlsym(r.h, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(r.h, "ncclRecv");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
  });
  return r;
}
constexpr int NCCL_UINT8 = 1;
void nccl_check(int rc, const char* what) {
  if (rc == 0) return;
  Rccl& r = rccl();
  throw std::runtime_error(std::string("RCCL ") + what + ": " + (r.GetErrorString ? r.GetErrorString(rc) : std::to_string(rc).c_str()));
}

template <class F> int guarded_x(F&& f) {
  try { f(); return GPUQ_OK; }
  catch (const HipError& e) { g_xerr = e.what(); return GPUQ_ERR_HIP; }
  catch (const Unsupported& e) { g_xerr = e.what(); return GPUQ_ERR_UNSUPPORTED; }
  catch (const Capacity& e) { g_xerr = e.what(); return GPUQ_ERR_CAPACITY; }
  catch (const std::exception& e) { g_xerr = e.what(); return GPUQ_ERR_INVALID; }
}

struct PinnedHost {
  void* p = nullptr; size_t cap = 0;
  ~PinnedHost() { if (p) (void)hipHostFree(p); }
  void* ensure(size_t n) { if (n > cap) { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; HIPCHECK(hipHostMalloc(&p, n ? n : 64, hipHostMallocDefault)); cap = n ? n : 64; } return p; }
};

int width_of(const gpuq_column& c) {
  if (c.type == GPUQ_UTF8 && c.repr == GPUQ_REPR_PACKED15) return 16;
  switch (c.type) {
    case GPUQ_INT32: case GPUQ_DATE32: case GPUQ_UINT32: return 4;
    case GPUQ_INT64: case GPUQ_FLOAT64: case GPUQ_UINT64: return 8;
    case GPUQ_DECIMAL128: return 16;
    default: return 0;
  }
}

}  // namespace

struct gpuq_comm {
  gpuq_ctx* ctx = nullptr; int rank = 0, world = 1;
  void* nccl = nullptr;                    // RCCL communicator, or
  gpuq_transport host{}; bool use_host = false;
  PinnedHost hs, hr;                       // staging of the host transport
  DevBuf meta_s, meta_r;                   // small device words for the counts exchange

  // one variable-size all-to-all of BYTES: rank d receives send[soff[d], +scnt[d]); what rank s sent lands at recv[roff[s], +rcnt[s])
  void xfer(hipStream_t st, const void* send, const int64_t* soff, const int64_t* scnt, void* recv, const int64_t* roff, const int64_t* rcnt) {
    if (!use_host) {
      Rccl& r = rccl();
      nccl_check(r.GroupStart(), "ncclGroupStart");
      for (int p = 0; p < world; ++p) {
        if (scnt[p] > 0) nccl_check(r.Send((const char*)send + soff[p], (size_t)scnt[p], NCCL_UINT8, p, nccl, st), "ncclSend");
        if (rcnt[p] > 0) nccl_check(r.Recv((char*)recv + roff[p], (size_t)rcnt[p], NCCL_UINT8, p, nccl, st), "ncclRecv");
      }
      nccl_check(r.GroupEnd(), "ncclGroupEnd");
      return;
    }
    // host-staged: pack in rank order, hand to the caller's all-to-all, unpack
    int64_t stot = 0, rtot = 0;
    for (int p = 0; p < world; ++p) { stot += scnt[p]; rtot += rcnt[p]; }
    char* hsp = (char*)hs.ensure((size_t)stot + 64); char* hrp = (char*)hr.ensure((size_t)rtot + 64);
    int64_t at = 0;
    for (int p = 0; p < world; ++p) { if (scnt[p] > 0) HIPCHECK(hipMemcpyAsync(hsp + at, (const char*)send + soff[p], (size_t)scnt[p], hipMemcpyDeviceToHost, st)); at += scnt[p]; }
    HIPCHECK(hipStreamSynchronize(st));
    const int rc = host.all_to_all_v(host.user, hsp, scnt, hrp, rcnt, world);
    if (rc != 0) throw std::runtime_error("exchange: the host transport's all_to_all_v failed with code " + std::to_string(rc));
    at = 0;
    for (int p = 0; p < world; ++p) { if (rcnt[p] > 0) HIPCHECK(hipMemcpyAsync((char*)recv + roff[p], hrp + at, (size_t)rcnt[p], hipMemcpyHostToDevice, st)); at += rcnt[p]; }
    HIPCHECK(hipStreamSynchronize(st));      // the staging buffer is reused by the next call
  }

  // all-to-all of k int64 words per peer (host arrays of world * k entries)
  void xfer_meta(hipStream_t st, const int64_t* send, int64_t* recv, int k) {
    const size_t bytes = (size_t)world * k * 8;
    std::vector<int64_t> off((size_t)world), cnt((size_t)world);
    for (int p = 0; p < world; ++p) { off[(size_t)p] = (int64_t)p * k * 8; cnt[(size_t)p] = (int64_t)k * 8; }
    if (use_host) {      // already host memory: no device round trip
      const int rc = host.all_to_all_v(host.user, send, cnt.data(), recv, cnt.data(), world);
      if (rc != 0) throw std::runtime_error("exchange: the host transport's all_to_all_v failed with code " + std::to_string(rc));
      return;
    }
    meta_s.ensure(bytes); meta_r.ensure(bytes);
    HIPCHECK(hipMemcpyAsync(meta_s.p, send, bytes, hipMemcpyHostToDevice, st));
    xfer(st, meta_s.p, off.data(), cnt.data(), meta_r.p, off.data(), cnt.data());
    HIPCHECK(hipMemcpyAsync(recv, meta_r.p, bytes, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
  }
};

namespace {

// Rows [doff[d], doff[d+1]) of every column go to rank d (doff[world] = n rows; a broadcast sends [0, n) to everybody).
// Returns the concatenation, in rank order, of what every rank sent here.
gpuq_table* exchange_impl(gpuq_comm* c, hipStream_t st, const gpuq_column* cols, const gpuq_field_info* fields, int n_cols, const std::vector<int64_t>& dlo,
                          const std::vector<int64_t>& dhi) {
  const int W = c->world;
  // ---- meta: per destination the row count and, per Arrow-layout Utf8 column, the byte count
  std::vector<int> utf8;
  for (int i = 0; i < n_cols; ++i) if (cols[i].type == GPUQ_UTF8 && cols[i].repr == GPUQ_REPR_ARROW) utf8.push_back(i);
  const int K = 1 + (int)utf8.size();
  std::vector<int64_t> smeta((size_t)W * K, 0), rmeta((size_t)W * K, 0);
  // first / last string offsets of every piece: one small read-back per Utf8 column
  std::vector<std::vector<int32_t>> ubeg(utf8.size(), std::vector<int32_t>((size_t)W, 0)), uend(utf8.size(), std::vector<int32_t>((size_t)W, 0));
  for (size_t u = 0; u < utf8.size(); ++u) {
    const gpuq_column& k = cols[utf8[u]];
    if (!k.offsets) throw std::runtime_error("exchange: Utf8 column without offsets");
    for (int d = 0; d < W; ++d) {
      if (dhi[(size_t)d] == dlo[(size_t)d]) continue;
      HIPCHECK(hipMemcpyAsync(&ubeg[u][(size_t)d], k.offsets + dlo[(size_t)d], 4, hipMemcpyDeviceToHost, st));
      HIPCHECK(hipMemcpyAsync(&uend[u][(size_t)d], k.offsets + dhi[(size_t)d], 4, hipMemcpyDeviceToHost, st));
    }
  }
  if (!utf8.empty()) HIPCHECK(hipStreamSynchronize(st));
  for (int d = 0; d < W; ++d) {
    smeta[(size_t)d * K] = dhi[(size_t)d] - dlo[(size_t)d];
    for (size_t u = 0; u < utf8.size(); ++u) smeta[(size_t)d * K + 1 + u] = (int64_t)uend[u][(size_t)d] - (int64_t)ubeg[u][(size_t)d];
  }
  c->xfer_meta(st, smeta.data(), rmeta.data(), K);
  std::vector<int64_t> rrows((size_t)W), rstart((size_t)W + 1, 0);
  for (int s = 0; s < W; ++s) { rrows[(size_t)s] = rmeta[(size_t)s * K]; rstart[(size_t)s + 1] = rstart[(size_t)s] + rrows[(size_t)s]; }
  const int64_t total = rstart[(size_t)W];
  if (total > 0xFFFFFFFEll) throw Unsupported("exchange: more than 2^32-2 rows would land on one rank");

  std::unique_ptr<gpuq_table> out(new gpuq_table());
  out->ctx = c->ctx; out->n_rows = total;
  std::vector<int64_t> soff((size_t)W), scnt((size_t)W), roff((size_t)W), rcnt((size_t)W);
  DevBuf sbits, rbits;      // bitmap pieces travel as whole 8-byte words per piece

  // a validity / Boolean bitmap: cut into word-aligned pieces, exchange, re-join at bit granularity (src == nullptr: all ones)
  auto exchange_bits = [&](const uint8_t* src, DevBuf& dst) {
    auto wbytes = [](int64_t rows) { return ((rows + 63) / 64) * 8; };
    int64_t stot = 0, rtot = 0;
    for (int d = 0; d < W; ++d) { soff[(size_t)d] = stot; scnt[(size_t)d] = wbytes(dhi[(size_t)d] - dlo[(size_t)d]); stot += scnt[(size_t)d]; }
    for (int s = 0; s < W; ++s) { roff[(size_t)s] = rtot; rcnt[(size_t)s] = wbytes(rrows[(size_t)s]); rtot += rcnt[(size_t)s]; }
    sbits.ensure((size_t)stot + 16); rbits.ensure((size_t)rtot + 16);
    HIPCHECK(hipMemsetAsync(sbits.p, 0, (size_t)stot + 16, st));
    for (int d = 0; d < W; ++d) {
      const int64_t rows = dhi[(size_t)d] - dlo[(size_t)d];
      if (rows > 0) launch_concat_bitmap(st, (u64*)((char*)sbits.p + soff[(size_t)d]), 0, src, src ? dlo[(size_t)d] : 0, rows);
    }
    c->xfer(st, sbits.p, soff.data(), scnt.data(), rbits.p, roff.data(), rcnt.data());
    const size_t ob = (size_t)((total + 63) / 64) * 8 + 16;
    dst.ensure(ob);
    HIPCHECK(hipMemsetAsync(dst.p, 0, ob, st));
    for (int s = 0; s < W; ++s)
      if (rrows[(size_t)s] > 0) launch_concat_bitmap(st, (u64*)dst.p, rstart[(size_t)s], (const uint8_t*)rbits.p + roff[(size_t)s], 0, rrows[(size_t)s]);
  };

  for (int i = 0; i < n_cols; ++i) {
    const gpuq_column& k = cols[i];
    std::unique_ptr<ImportedCol> ic(new ImportedCol());
    ic->field = fields[i];
    ic->col = k; ic->col.length = total; ic->col.data = nullptr; ic->col.offsets = nullptr; ic->col.validity = nullptr;
    const bool arrow_utf8 = k.type == GPUQ_UTF8 && k.repr == GPUQ_REPR_ARROW;
    const int w = width_of(k);
    if (k.type == GPUQ_BOOL) {
      exchange_bits((const uint8_t*)k.data, ic->data);
    } else if (arrow_utf8) {
      const size_t u = (size_t)(std::find(utf8.begin(), utf8.end(), i) - utf8.begin());
      // offsets: piece d = rows+1 entries (absolute on the sender); rebased on arrival to the running byte total
      DevBuf roffs;
      int64_t rt = 0;
      for (int d = 0; d < W; ++d) { const int64_t rows = dhi[(size_t)d] - dlo[(size_t)d]; soff[(size_t)d] = dlo[(size_t)d] * 4; scnt[(size_t)d] = rows > 0 ? (rows + 1) * 4 : 0; }
      for (int s = 0; s < W; ++s) { roff[(size_t)s] = rt; rcnt[(size_t)s] = rrows[(size_t)s] > 0 ? (rrows[(size_t)s] + 1) * 4 : 0; rt += rcnt[(size_t)s]; }
      roffs.ensure((size_t)rt + 16);
      c->xfer(st, k.offsets, soff.data(), scnt.data(), roffs.p, roff.data(), rcnt.data());
      // first offset of every received piece (the sender's absolute position): needed for the rebase
      std::vector<int32_t> first((size_t)W, 0);
      for (int s = 0; s < W; ++s) if (rrows[(size_t)s] > 0) HIPCHECK(hipMemcpyAsync(&first[(size_t)s], (const char*)roffs.p + roff[(size_t)s], 4, hipMemcpyDeviceToHost, st));
      HIPCHECK(hipStreamSynchronize(st));
      std::vector<int64_t> bstart((size_t)W + 1, 0);
      for (int s = 0; s < W; ++s) bstart[(size_t)s + 1] = bstart[(size_t)s] + rmeta[(size_t)s * K + 1 + u];
      if (bstart[(size_t)W] > 0x7FFFFFFFll) throw Unsupported("exchange: a Utf8 column would exceed 2 GiB on one rank (int32 offsets)");
      ic->offsets.ensure((size_t)(total + 1) * 4 + 16);
      HIPCHECK(hipMemsetAsync(ic->offsets.p, 0, 4, st));
      for (int s = 0; s < W; ++s) {
        if (rrows[(size_t)s] == 0) continue;
        launch_offsets_rebase(st, (const int32_t*)((const char*)roffs.p + roff[(size_t)s]), rrows[(size_t)s] + 1, (int32_t)(bstart[(size_t)s] - first[(size_t)s]),
                              (int32_t*)ic->offsets.p + rstart[(size_t)s]);
      }
      if (total == 0) HIPCHECK(hipMemsetAsync(ic->offsets.p, 0, 8, st));
      // bytes
      for (int d = 0; d < W; ++d) { soff[(size_t)d] = ubeg[u][(size_t)d]; scnt[(size_t)d] = smeta[(size_t)d * K + 1 + u]; }
      for (int s = 0; s < W; ++s) { roff[(size_t)s] = bstart[(size_t)s]; rcnt[(size_t)s] = rmeta[(size_t)s * K + 1 + u]; }
      ic->data.ensure((size_t)bstart[(size_t)W] + 16);
      c->xfer(st, k.data, soff.data(), scnt.data(), ic->data.p, roff.data(), rcnt.data());
      ic->col.offsets = (const int32_t*)ic->offsets.p;
      HIPCHECK(hipStreamSynchronize(st));      // roffs dies with this scope
    } else {
      if (!w) throw Unsupported("exchange: column type " + std::to_string(k.type));
      for (int d = 0; d < W; ++d) { soff[(size_t)d] = dlo[(size_t)d] * w; scnt[(size_t)d] = (dhi[(size_t)d] - dlo[(size_t)d]) * w; }
      for (int s = 0; s < W; ++s) { roff[(size_t)s] = rstart[(size_t)s] * w; rcnt[(size_t)s] = rrows[(size_t)s] * w; }
      ic->data.ensure((size_t)total * w + 16);
      c->xfer(st, k.data, soff.data(), scnt.data(), ic->data.p, roff.data(), rcnt.data());
    }
    ic->col.data = ic->data.p;
    // nullability is a property of the schema, so that every rank takes the same branch (a collective per column buffer)
    if (fields[i].nullable) { exchange_bits(k.validity, ic->validity); ic->col.validity = (const uint8_t*)ic->validity.p; }
    out->cols.push_back(std::move(ic));
  }
  HIPCHECK(hipStreamSynchronize(st));      // the send-side scratch (sbits / rbits) dies with this frame
  return out.release();
}

}  // namespace

extern "C" {

const char* gpuq_exchange_last_error(void) { return g_xerr.c_str(); }

int gpuq_comm_unique_id(uint8_t* id_out) {
  return guarded_x([&]() {
    if (!id_out) throw std::runtime_error("id_out is NULL");
    Rccl& r = rccl();
    if (!r.ok) throw Unsupported("RCCL (librccl.so) is not available on this host");
    NcclId id; std::memset(&id, 0, sizeof(id));
    nccl_check(r.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(id_out, &id, GPUQ_COMM_ID_BYTES);
  });
}

int gpuq_comm_create(gpuq_ctx* ctx, const uint8_t* id, int rank, int world, gpuq_comm** out) {
  if (out) *out = nullptr;
  return guarded_x([&]() {
    if (!ctx || !id || !out) throw std::runtime_error("ctx / id / out is NULL");
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("rank / world out of range");
    Rccl& r = rccl();
    if (!r.ok) throw Unsupported("RCCL (librccl.so) is not available on this host");
    HIPCHECK(hipSetDevice(ctx->device));
    std::unique_ptr<gpuq_comm> c(new gpuq_comm());
    c->ctx = ctx; c->rank = rank; c->world = world;
    NcclId nid; std::memcpy(&nid, id, GPUQ_COMM_ID_BYTES);
    nccl_check(r.CommInitRank(&c->nccl, world, nid, rank), "ncclCommInitRank");
    *out = c.release();
  });
}

int gpuq_comm_create_host(gpuq_ctx* ctx, const gpuq_transport* transport, int rank, int world, gpuq_comm** out) {
  if (out) *out = nullptr;
  return guarded_x([&]() {
    if (!ctx || !transport || !transport->all_to_all_v || !out) throw std::runtime_error("ctx / transport / out is NULL");
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("rank / world out of range");
    std::unique_ptr<gpuq_comm> c(new gpuq_comm());
    c->ctx = ctx; c->rank = rank; c->world = world; c->host = *transport; c->use_host = true;
    *out = c.release();
  });
}

void gpuq_comm_free(gpuq_comm* c) {
  if (!c) return;
  if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
  delete c;
}
int gpuq_comm_rank(const gpuq_comm* c) { return c ? c->rank : -1; }
int gpuq_comm_world(const gpuq_comm* c) { return c ? c->world : 0; }

int gpuq_exchange_partitions(gpuq_comm* c, void* stream, const gpuq_column* cols, const gpuq_field_info* fields, int n_cols, const int64_t* dest_offsets, gpuq_table** out) {
  if (out) *out = nullptr;
  return guarded_x([&]() {
    if (!c || !out || !dest_offsets || (n_cols > 0 && (!cols || !fields))) throw std::runtime_error("comm / cols / fields / dest_offsets / out is NULL");
    HIPCHECK(hipSetDevice(c->ctx->device));
    hipStream_t st = use_stream(stream);
    std::vector<int64_t> lo((size_t)c->world), hi((size_t)c->world);
    for (int d = 0; d < c->world; ++d) {
      lo[(size_t)d] = dest_offsets[d]; hi[(size_t)d] = dest_offsets[d + 1];
      if (lo[(size_t)d] < 0 || hi[(size_t)d] < lo[(size_t)d]) throw std::runtime_error("dest_offsets must be non-decreasing");
    }
    for (int i = 0; i < n_cols; ++i) if (cols[i].length < hi[(size_t)c->world - 1]) throw std::runtime_error("column " + std::to_string(i) + " is shorter than dest_offsets[world]");
    *out = exchange_impl(c, st, cols, fields, n_cols, lo, hi);
  });
}

int gpuq_allgather_table(gpuq_comm* c, void* stream, const gpuq_column* cols, const gpuq_field_info* fields, int n_cols, int64_t n_rows, gpuq_table** out) {
  if (out) *out = nullptr;
  return guarded_x([&]() {
    if (!c || !out || (n_cols > 0 && (!cols || !fields))) throw std::runtime_error("comm / cols / fields / out is NULL");
    if (n_rows < 0) throw std::runtime_error("n_rows is negative");
    HIPCHECK(hipSetDevice(c->ctx->device));
    hipStream_t st = use_stream(stream);
    std::vector<int64_t> lo((size_t)c->world, 0), hi((size_t)c->world, n_rows);      // everybody gets everything
    *out = exchange_impl(c, st, cols, fields, n_cols, lo, hi);
  });
}

}  // extern "C"
