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
  int (*CommAbort)(void*) = nullptr;
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
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.h, "ncclCommAbort");      // optional
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

struct PeerFailed : std::runtime_error { using std::runtime_error::runtime_error; };      // a rank of the node failed: GPUQ_ERR_PEER on all of them
template <class F> int guarded_x(F&& f) {
  try { f(); return GPUQ_OK; }
  catch (const HipError& e) { g_xerr = e.what(); return GPUQ_ERR_HIP; }
  catch (const Unsupported& e) { g_xerr = e.what(); return GPUQ_ERR_UNSUPPORTED; }
  catch (const Capacity& e) { g_xerr = e.what(); return GPUQ_ERR_CAPACITY; }
  catch (const Retry& e) { g_xerr = e.what(); return GPUQ_ERR_RETRY; }
  catch (const PeerFailed& e) { g_xerr = e.what(); return GPUQ_ERR_PEER; }
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
    case GPUQ_INT32: case GPUQ_DATE32: case GPUQ_UINT32: case GPUQ_FLOAT32: return 4;
    case GPUQ_INT64: case GPUQ_FLOAT64: case GPUQ_UINT64: case GPUQ_TIMESTAMP: case GPUQ_DATE64: return 8;
    case GPUQ_INT8: case GPUQ_UINT8: return 1;
    case GPUQ_INT16: case GPUQ_UINT16: return 2;
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
  int status = 0;                          // travels with the next exchange's meta round (gpuq_comm_set_status)
  bool broken = false;                     // a failure in the middle of a payload round: the communicator was aborted

  // one buffer's variable-size all-to-all: rank d receives send[soff[d], +scnt[d]); what rank s sent lands at recv[roff[s], +rcnt[s])
  struct Piece { const void* send = nullptr; void* recv = nullptr; std::vector<int64_t> soff, scnt, roff, rcnt; };
  // ALL pieces in one grouped round (RCCL: one ncclGroupStart / End around every send and receive of every buffer, matched per peer
  // in issue order; host transport: one staged call per piece)
  void xfer_many(hipStream_t st, const std::vector<Piece>& ps) {
    if (use_host) { for (auto& p : ps) xfer(st, p.send, p.soff.data(), p.scnt.data(), p.recv, p.roff.data(), p.rcnt.data()); return; }
    Rccl& r = rccl();
    nccl_check(r.GroupStart(), "ncclGroupStart");
    for (auto& p : ps)
      for (int q = 0; q < world; ++q) {
        if (p.scnt[(size_t)q] > 0) nccl_check(r.Send((const char*)p.send + p.soff[(size_t)q], (size_t)p.scnt[(size_t)q], NCCL_UINT8, q, nccl, st), "ncclSend");
        if (p.rcnt[(size_t)q] > 0) nccl_check(r.Recv((char*)p.recv + p.roff[(size_t)q], (size_t)p.rcnt[(size_t)q], NCCL_UINT8, q, nccl, st), "ncclRecv");
      }
    nccl_check(r.GroupEnd(), "ncclGroupEnd");
  }
  void abort() {
    broken = true;
    if (nccl && rccl().CommAbort) { (void)rccl().CommAbort(nccl); nccl = nullptr; }
  }

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

constexpr int MAX_XCHG_UTF8 = 15;                  // Arrow-layout Utf8 columns per exchanged table
constexpr int META_K = 2 + 2 * MAX_XCHG_UTF8;      // words per destination: status, rows, then (bytes, first offset) per Utf8 column

// Rows [dlo[d], dhi[d]) of every column go to rank d (a broadcast sends [0, n) to everybody).  Returns the concatenation, in rank
// order, of what every rank sent here.
//
// Every rank makes the same sequence of collective calls whatever happens locally:
//   1. ONE fixed-size meta round: per destination the caller's status word (gpuq_comm_set_status: 0 fine, 1 "I have failed", 2 "my
//      deferred execution did not hold"), the row count and per Utf8 column the byte count and first offset.  A rank whose plan
//      failed BELOW the exchange takes part in this round with its status and no rows (gpuq_comm_announce) -- its peers then stop
//      here, together, instead of waiting for data that never comes.
//   2. ONE agreement round over what the counts imply locally (2^32 rows / 2 GiB of strings on one rank): a limit hit on one
//      rank stops all of them before any payload moves.
//   3. ONE grouped send / receive round for ALL buffers of the table (fixed-width data, bitmap pieces, Utf8 offsets and bytes):
//      round 2 had one group per column buffer, i.e. n_buffers x (launch + proxy hand-shake) per exchange.
gpuq_table* exchange_impl(gpuq_comm* c, hipStream_t st, const gpuq_column* cols, const gpuq_field_info* fields, int n_cols, const std::vector<int64_t>& dlo,
                          const std::vector<int64_t>& dhi) {
  const int W = c->world;
  if (c->broken) throw std::runtime_error("exchange: this communicator was aborted by an earlier failure");
  std::vector<int> utf8;
  for (int i = 0; i < n_cols; ++i) if (cols[i].type == GPUQ_UTF8 && cols[i].repr == GPUQ_REPR_ARROW) utf8.push_back(i);
  int my_status = c->status; c->status = 0;
  std::string my_error;
  if (!my_status && (int)utf8.size() > MAX_XCHG_UTF8) { my_status = 1; my_error = "exchange: more than " + std::to_string(MAX_XCHG_UTF8) + " Arrow-layout Utf8 columns in one table"; }
  for (int i = 0; i < n_cols && !my_status; ++i) {
    const gpuq_column& k = cols[i];
    const bool arrow_utf8 = k.type == GPUQ_UTF8 && k.repr == GPUQ_REPR_ARROW;
    if (arrow_utf8 && !k.offsets) { my_status = 1; my_error = "exchange: Utf8 column without offsets"; }
    else if (!arrow_utf8 && k.type != GPUQ_BOOL && !width_of(k)) { my_status = 1; my_error = "exchange: column type " + std::to_string(k.type); }
  }
  // ---- 1. meta
  const int K = META_K;
  std::vector<int64_t> smeta((size_t)W * K, 0), rmeta((size_t)W * K, 0);
  std::vector<std::vector<int32_t>> ubeg(utf8.size(), std::vector<int32_t>((size_t)W, 0)), uend(utf8.size(), std::vector<int32_t>((size_t)W, 0));
  if (!my_status) {
    // first / last string offsets of every piece: one small read-back per Utf8 column
    for (size_t u = 0; u < utf8.size(); ++u) {
      const gpuq_column& k = cols[utf8[u]];
      for (int d = 0; d < W; ++d) {
        if (dhi[(size_t)d] == dlo[(size_t)d]) continue;
        HIPCHECK(hipMemcpyAsync(&ubeg[u][(size_t)d], k.offsets + dlo[(size_t)d], 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(&uend[u][(size_t)d], k.offsets + dhi[(size_t)d], 4, hipMemcpyDeviceToHost, st));
      }
    }
    if (!utf8.empty()) HIPCHECK(hipStreamSynchronize(st));
  }
  for (int d = 0; d < W; ++d) {
    int64_t* m = &smeta[(size_t)d * K];
    m[0] = my_status;
    if (my_status) continue;
    m[1] = dhi[(size_t)d] - dlo[(size_t)d];
    for (size_t u = 0; u < utf8.size(); ++u) { m[2 + 2 * u] = (int64_t)uend[u][(size_t)d] - (int64_t)ubeg[u][(size_t)d]; m[3 + 2 * u] = ubeg[u][(size_t)d]; }
  }
  c->xfer_meta(st, smeta.data(), rmeta.data(), K);
  {
    int worst = my_status, who = my_status ? c->rank : -1;
    for (int s = 0; s < W; ++s) if ((int)rmeta[(size_t)s * K] > worst) { worst = (int)rmeta[(size_t)s * K]; who = s; }
    if (worst == 2) throw Retry("exchange: rank " + std::to_string(who) + " has to redo its deferred execution: every rank does");
    if (worst) throw PeerFailed(my_status ? (my_error.empty() ? std::string("exchange: this rank announced a failure") : my_error)
                                           : "exchange: rank " + std::to_string(who) + " failed; nothing was exchanged");
  }
  std::vector<int64_t> rrows((size_t)W), rstart((size_t)W + 1, 0);
  for (int s = 0; s < W; ++s) { rrows[(size_t)s] = rmeta[(size_t)s * K + 1]; rstart[(size_t)s + 1] = rstart[(size_t)s] + rrows[(size_t)s]; }
  const int64_t total = rstart[(size_t)W];
  std::vector<std::vector<int64_t>> bstart(utf8.size(), std::vector<int64_t>((size_t)W + 1, 0));
  for (size_t u = 0; u < utf8.size(); ++u) for (int s = 0; s < W; ++s) bstart[u][(size_t)s + 1] = bstart[u][(size_t)s] + rmeta[(size_t)s * K + 2 + 2 * u];
  // ---- 2. agreement on what the counts imply here
  {
    std::string why;
    if (total > 0xFFFFFFFEll) why = "exchange: more than 2^32-2 rows would land on one rank";
    for (size_t u = 0; u < utf8.size() && why.empty(); ++u) if (bstart[u][(size_t)W] > 0x7FFFFFFFll) why = "exchange: a Utf8 column would exceed 2 GiB on one rank (int32 offsets)";
    std::vector<int64_t> sa((size_t)W, why.empty() ? 0 : 1), ra((size_t)W, 0);
    c->xfer_meta(st, sa.data(), ra.data(), 1);
    if (!why.empty()) throw Unsupported(why);
    for (int s = 0; s < W; ++s) if (ra[(size_t)s]) throw PeerFailed("exchange: rank " + std::to_string(s) + " cannot hold what it would receive; nothing was exchanged");
  }

  // ---- 3. every buffer of the table in one grouped round
  std::unique_ptr<gpuq_table> out(new gpuq_table());
  out->ctx = c->ctx; out->n_rows = total; out->piece_rows = rrows;
  std::vector<gpuq_comm::Piece> pieces;
  std::vector<std::unique_ptr<DevBuf>> scratch;      // send-side bitmap pieces, received bitmap pieces, received offsets: alive until the round has run
  struct BitJob { const DevBuf* rbits; std::vector<int64_t> roff; DevBuf* dst; };
  struct OffJob { const DevBuf* roffs; std::vector<int64_t> roff; size_t u; ImportedCol* ic; };
  std::vector<BitJob> bit_jobs; std::vector<OffJob> off_jobs;
  auto wbytes = [](int64_t rows) { return ((rows + 63) / 64) * 8; };
  // a validity / Boolean bitmap: cut into word-aligned pieces, exchange, re-join at bit granularity (src == nullptr: all ones)
  auto add_bits = [&](const uint8_t* src, DevBuf& dst) {
    gpuq_comm::Piece p; p.soff.resize((size_t)W); p.scnt.resize((size_t)W); p.roff.resize((size_t)W); p.rcnt.resize((size_t)W);
    int64_t stot = 0, rtot = 0;
    for (int d = 0; d < W; ++d) { p.soff[(size_t)d] = stot; p.scnt[(size_t)d] = wbytes(dhi[(size_t)d] - dlo[(size_t)d]); stot += p.scnt[(size_t)d]; }
    for (int s = 0; s < W; ++s) { p.roff[(size_t)s] = rtot; p.rcnt[(size_t)s] = wbytes(rrows[(size_t)s]); rtot += p.rcnt[(size_t)s]; }
    scratch.emplace_back(new DevBuf()); DevBuf& sb = *scratch.back();
    scratch.emplace_back(new DevBuf()); DevBuf& rb = *scratch.back();
    sb.ensure((size_t)stot + 16); rb.ensure((size_t)rtot + 16);
    HIPCHECK(hipMemsetAsync(sb.p, 0, (size_t)stot + 16, st));
    for (int d = 0; d < W; ++d) {
      const int64_t rows = dhi[(size_t)d] - dlo[(size_t)d];
      if (rows > 0) launch_concat_bitmap(st, (u64*)((char*)sb.p + p.soff[(size_t)d]), 0, src, src ? dlo[(size_t)d] : 0, rows);
    }
    p.send = sb.p; p.recv = rb.p;
    const size_t ob = (size_t)((total + 63) / 64) * 8 + 16;
    dst.ensure(ob);
    HIPCHECK(hipMemsetAsync(dst.p, 0, ob, st));
    bit_jobs.push_back({&rb, p.roff, &dst});
    pieces.push_back(std::move(p));
  };
  for (int i = 0; i < n_cols; ++i) {
    const gpuq_column& k = cols[i];
    std::unique_ptr<ImportedCol> ic(new ImportedCol());
    ic->field = fields[i];
    ic->col = k; ic->col.length = total; ic->col.data = nullptr; ic->col.offsets = nullptr; ic->col.validity = nullptr;
    const bool arrow_utf8 = k.type == GPUQ_UTF8 && k.repr == GPUQ_REPR_ARROW;
    const int w = width_of(k);
    if (k.type == GPUQ_BOOL) {
      add_bits((const uint8_t*)k.data, ic->data);
    } else if (arrow_utf8) {
      const size_t u = (size_t)(std::find(utf8.begin(), utf8.end(), i) - utf8.begin());
      // offsets: piece d = rows+1 entries (absolute on the sender); rebased on arrival to the running byte total
      gpuq_comm::Piece po; po.soff.resize((size_t)W); po.scnt.resize((size_t)W); po.roff.resize((size_t)W); po.rcnt.resize((size_t)W);
      int64_t rt = 0;
      for (int d = 0; d < W; ++d) { const int64_t rows = dhi[(size_t)d] - dlo[(size_t)d]; po.soff[(size_t)d] = dlo[(size_t)d] * 4; po.scnt[(size_t)d] = rows > 0 ? (rows + 1) * 4 : 0; }
      for (int s = 0; s < W; ++s) { po.roff[(size_t)s] = rt; po.rcnt[(size_t)s] = rrows[(size_t)s] > 0 ? (rrows[(size_t)s] + 1) * 4 : 0; rt += po.rcnt[(size_t)s]; }
      scratch.emplace_back(new DevBuf()); DevBuf& roffs = *scratch.back();
      roffs.ensure((size_t)rt + 16);
      po.send = k.offsets; po.recv = roffs.p;
      ic->offsets.ensure((size_t)(total + 1) * 4 + 16);
      HIPCHECK(hipMemsetAsync(ic->offsets.p, 0, 8, st));
      off_jobs.push_back({&roffs, po.roff, u, ic.get()});
      pieces.push_back(std::move(po));
      // bytes
      gpuq_comm::Piece pb; pb.soff.resize((size_t)W); pb.scnt.resize((size_t)W); pb.roff.resize((size_t)W); pb.rcnt.resize((size_t)W);
      for (int d = 0; d < W; ++d) { pb.soff[(size_t)d] = ubeg[u][(size_t)d]; pb.scnt[(size_t)d] = smeta[(size_t)d * K + 2 + 2 * u]; }
      for (int s = 0; s < W; ++s) { pb.roff[(size_t)s] = bstart[u][(size_t)s]; pb.rcnt[(size_t)s] = rmeta[(size_t)s * K + 2 + 2 * u]; }
      ic->data.ensure((size_t)bstart[u][(size_t)W] + 16);
      pb.send = k.data; pb.recv = ic->data.p;
      pieces.push_back(std::move(pb));
      ic->col.offsets = (const int32_t*)ic->offsets.p;
    } else {
      gpuq_comm::Piece p; p.soff.resize((size_t)W); p.scnt.resize((size_t)W); p.roff.resize((size_t)W); p.rcnt.resize((size_t)W);
      for (int d = 0; d < W; ++d) { p.soff[(size_t)d] = dlo[(size_t)d] * w; p.scnt[(size_t)d] = (dhi[(size_t)d] - dlo[(size_t)d]) * w; }
      for (int s = 0; s < W; ++s) { p.roff[(size_t)s] = rstart[(size_t)s] * w; p.rcnt[(size_t)s] = rrows[(size_t)s] * w; }
      ic->data.ensure((size_t)total * w + 16);
      p.send = k.data; p.recv = ic->data.p;
      pieces.push_back(std::move(p));
    }
    ic->col.data = ic->data.p;
    // nullability is a property of the schema, so that every rank sends the same set of buffers
    if (fields[i].nullable) { add_bits(k.validity, ic->validity); ic->col.validity = (const uint8_t*)ic->validity.p; }
    out->cols.push_back(std::move(ic));
  }
  try { c->xfer_many(st, pieces); }
  catch (...) { c->abort(); throw; }      // the peers may already be inside the round: tear the communicator down rather than leave them waiting
  // ---- 4. arrival: bitmaps re-joined, offsets rebased (the sender's first offset of every piece came with the meta)
  for (auto& j : bit_jobs)
    for (int s = 0; s < W; ++s)
      if (rrows[(size_t)s] > 0) launch_concat_bitmap(st, (u64*)j.dst->p, rstart[(size_t)s], (const uint8_t*)j.rbits->p + j.roff[(size_t)s], 0, rrows[(size_t)s]);
  for (auto& j : off_jobs)
    for (int s = 0; s < W; ++s) {
      if (rrows[(size_t)s] == 0) continue;
      const int64_t first = rmeta[(size_t)s * K + 3 + 2 * j.u];
      launch_offsets_rebase(st, (const int32_t*)((const char*)j.roffs->p + j.roff[(size_t)s]), rrows[(size_t)s] + 1, (int32_t)(bstart[j.u][(size_t)s] - first),
                            (int32_t*)j.ic->offsets.p + rstart[(size_t)s]);
    }
  HIPCHECK(hipStreamSynchronize(st));      // the scratch buffers die with this frame
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
int gpuq_table_piece_rows(const gpuq_table* t, int64_t* rows_out, int cap, int* n_out) {
  if (!t || !n_out) return GPUQ_ERR_INVALID;
  *n_out = (int)t->piece_rows.size();
  if (rows_out) { if (cap < *n_out) return GPUQ_ERR_CAPACITY; for (int i = 0; i < *n_out; ++i) rows_out[i] = t->piece_rows[(size_t)i]; }
  return GPUQ_OK;
}
int gpuq_comm_set_status(gpuq_comm* c, int status) { if (!c || status < 0 || status > 2) return GPUQ_ERR_INVALID; c->status = status; return GPUQ_OK; }
int gpuq_comm_announce(gpuq_comm* c, void* stream) {
  // the meta round of an exchange with no table behind it: delivers this rank's status (set it first) to peers that are entering an
  // exchange, or -- with status 0 on every rank -- is a barrier that agrees "everybody is fine" (the end of a deferred execution)
  return guarded_x([&]() {
    if (!c) throw std::runtime_error("comm is NULL");
    HIPCHECK(hipSetDevice(c->ctx->device));
    hipStream_t st = use_stream(stream);
    if (c->broken) throw std::runtime_error("exchange: this communicator was aborted by an earlier failure");
    const int my = c->status; c->status = 0;
    std::vector<int64_t> sm((size_t)c->world * META_K, 0), rm((size_t)c->world * META_K, 0);
    for (int d = 0; d < c->world; ++d) sm[(size_t)d * META_K] = my;
    c->xfer_meta(st, sm.data(), rm.data(), META_K);
    int worst = my, who = my ? c->rank : -1;
    for (int s = 0; s < c->world; ++s) if ((int)rm[(size_t)s * META_K] > worst) { worst = (int)rm[(size_t)s * META_K]; who = s; }
    if (worst == 2) throw Retry("rank " + std::to_string(who) + " has to redo its deferred execution: every rank does");
    if (worst) throw PeerFailed(my ? std::string("this rank announced a failure") : "rank " + std::to_string(who) + " failed");
  });
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
