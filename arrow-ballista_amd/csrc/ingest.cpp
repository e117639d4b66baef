// Streaming ingest of a partition: host Arrow RecordBatches -> one set of device columns, pipelined.
//
// What it stands for in the reference: the per-batch pull loop of a task (`stream.next()` at
// ballista/core/src/execution_plans/shuffle_writer.rs:341, utils.rs:198) -- the place where host-resident Arrow batches
// (a scan's output, a Flight fetch) enter the operator path.  BASELINE configs[1] feeds 64 Ki-row batches.  On the device a
// partition is ONE table, so ingest = append every batch to preallocated device columns.  North star:
// "Arrow buffers pinned and streamed to HBM".
//
// A pageable host buffer cannot be DMA'd directly; one thread copying into pinned memory tops out near 10 GB/s, a sixth of
// PCIe Gen5 x16.  So gpuq_ingest_push only queues the batch; K worker threads each take whole batches, copy the column
// buffers into their own pinned slot (parallel host memcpy: this is what fills the link), issue the H2D copies on their own
// copy stream and release the batch when its copies have landed.  Batches land out of order; `rows_landed` is the length of
// the contiguous prefix that is complete, and gpuq_ingest_wait makes a consumer stream wait for exactly the copies of that
// prefix -- the consumer (a partial aggregate over rows [a, b)) runs while later batches are still in flight.
#include "gpuq_internal.h"
#include "expr_compile.h"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

using namespace gpuq;

namespace {
thread_local std::string g_ierr;
template <class F> int guarded_i(F&& f) {
  try { f(); return GPUQ_OK; }
  catch (const HipError& e) { g_ierr = e.what(); return GPUQ_ERR_HIP; }
  catch (const Unsupported& e) { g_ierr = e.what(); return GPUQ_ERR_UNSUPPORTED; }
  catch (const Capacity& e) { g_ierr = e.what(); return GPUQ_ERR_CAPACITY; }
  catch (const std::exception& e) { g_ierr = e.what(); return GPUQ_ERR_INVALID; }
}
// bits [src_off, src_off + n) of src -> bits [dst_off, ...) of dst (dst bytes beyond the range are preserved by OR-ing into zeroed memory)
void copy_bits_host(uint8_t* dst, int64_t dst_off, const uint8_t* src, int64_t src_off, int64_t n) {
  if (((dst_off | src_off) & 7) == 0) { std::memcpy(dst + (dst_off >> 3), src + (src_off >> 3), (size_t)((n + 7) >> 3)); return; }
  for (int64_t i = 0; i < n; ++i) if ((src[(src_off + i) >> 3] >> ((src_off + i) & 7)) & 1) dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
}
}  // namespace

struct gpuq_ingest {
  gpuq_ctx* ctx = nullptr;
  struct Col { DType type; bool nullable = false, large = false; std::string name; DevBuf data, offsets, validity; int64_t bytes_cap = 0; };
  // Utf8 / LargeUtf8 offset i of a host array
  static int64_t utf8_off(const Col& k, const ArrowArray* a, int64_t i) { return k.large ? ((const int64_t*)a->buffers[1])[i] : (int64_t)((const int32_t*)a->buffers[1])[i]; }
  std::deque<Col> cols;      // DevBuf is not movable: a deque never relocates its elements
  int64_t cap_rows = 0;
  // producer side (gpuq_ingest_push): row / byte cursors
  int64_t next_row = 0; std::vector<int64_t> next_byte;
  struct Job { ArrowArray batch{}; int64_t row0 = 0, rows = 0; std::vector<int64_t> byte0; int64_t seq = 0; };
  std::mutex mu; std::condition_variable cv_work, cv_done;
  std::deque<Job*> queue; bool stop = false; std::string error;
  // completion tracking: batches finish out of order
  int64_t pushed = 0, completed_prefix = 0, rows_prefix = 0; std::vector<char> done_flags; std::vector<int64_t> rows_of;
  std::vector<hipEvent_t> last_event;       // per worker: event after its latest completed batch's copies
  struct Worker { std::thread th; hipStream_t stream = nullptr; void* pin = nullptr; size_t pin_cap = 0; hipEvent_t ev = nullptr; };
  std::vector<Worker> workers;
  std::atomic<int64_t> bytes_copied{0};

  void fail(const std::string& m) { std::lock_guard<std::mutex> lk(mu); if (error.empty()) error = m; }

  void run_worker(int wi) {
    (void)hipSetDevice(ctx->device);
    Worker& w = workers[(size_t)wi];
    for (;;) {
      Job* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return stop || !queue.empty(); });
        if (queue.empty()) return;
        job = queue.front(); queue.pop_front();
      }
      try { copy_batch(w, *job); }
      catch (const std::exception& e) { fail(e.what()); }
      if (job->batch.release) job->batch.release(&job->batch);
      {
        std::lock_guard<std::mutex> lk(mu);
        done_flags[(size_t)job->seq] = 1;
        while (completed_prefix < pushed && done_flags[(size_t)completed_prefix]) { rows_prefix += rows_of[(size_t)completed_prefix]; ++completed_prefix; }
      }
      cv_done.notify_all();
      delete job;
    }
  }

  // stage + H2D of one batch on the worker's stream; returns when the copies have landed (the pinned slot is reused next)
  void copy_batch(Worker& w, const Job& job) {
    const ArrowArray& b = job.batch;
    size_t need = 0;
    for (size_t c = 0; c < cols.size(); ++c) {
      const ArrowArray* a = b.children[c];
      const Col& k = cols[c];
      if (k.type.id == T_UTF8) { const int64_t o0 = utf8_off(k, a, a->offset + b.offset), o1 = utf8_off(k, a, a->offset + b.offset + job.rows); need += (size_t)(job.rows + 1) * 4 + (size_t)(o1 - o0) + 128; }
      else if (k.type.id == T_BOOL) need += (size_t)(job.rows + 7) / 8 + 72;
      else need += (size_t)job.rows * (size_t)type_width(k.type) + 64;
      if (k.nullable) need += (size_t)(job.rows + 7) / 8 + 72;
    }
    if (need > w.pin_cap) { if (w.pin) (void)hipHostFree(w.pin); w.pin = nullptr; w.pin_cap = 0; HIPCHECK(hipHostMalloc(&w.pin, need + (need >> 2), hipHostMallocDefault)); w.pin_cap = need + (need >> 2); }
    char* p = (char*)w.pin; size_t at = 0;
    auto stage = [&](size_t bytes) { char* q = p + at; at += (bytes + 63) & ~(size_t)63; return q; };
    for (size_t c = 0; c < cols.size(); ++c) {
      const ArrowArray* a = b.children[c];
      Col& k = cols[c];
      const int64_t off = a->offset + b.offset, n = job.rows;
      if (n == 0) continue;
      // validity: destination bit offset = row0 (a batch boundary is a byte boundary when the batches before it hold multiples of 8 rows)
      if (k.nullable) {
        const bool has = a->null_count != 0 && a->n_buffers > 0 && a->buffers[0];
        if ((job.row0 & 7) == 0) {
          const size_t vb = (size_t)(n + 7) / 8;
          char* q = stage(vb);
          if (has) { std::memset(q, 0, vb); copy_bits_host((uint8_t*)q, 0, (const uint8_t*)a->buffers[0], off, n); } else std::memset(q, 0xFF, vb);
          if ((n & 7) && !has) q[vb - 1] = (char)((1u << (n & 7)) - 1);
          HIPCHECK(hipMemcpyAsync((char*)k.validity.p + (job.row0 >> 3), q, vb, hipMemcpyHostToDevice, w.stream));
        } else throw Unsupported("ingest: a nullable column needs every batch but the last to hold a multiple of 8 rows");
      }
      if (k.type.id == T_UTF8) {
        // (LargeUtf8: 64-bit offsets on the host side; the partition's own offsets are 32-bit, its byte total is bounded by max_utf8_bytes)
        const int64_t o0 = utf8_off(k, a, off);
        const int64_t nbytes = utf8_off(k, a, off + n) - o0;
        if (nbytes < 0) throw std::runtime_error("ingest: Utf8 offsets of column '" + k.name + "' decrease");
        int32_t* q = (int32_t*)stage((size_t)(n + 1) * 4);
        const int64_t delta = job.byte0[c] - o0;
        for (int64_t i = 0; i <= n; ++i) {      // rebased to the partition's running byte total
          const int64_t o = utf8_off(k, a, off + i);
          if (o < o0 || o > o0 + nbytes || (i > 0 && o < utf8_off(k, a, off + i - 1))) throw std::runtime_error("ingest: Utf8 offsets of column '" + k.name + "' are not non-decreasing");
          q[i] = (int32_t)(o + delta);
        }
        HIPCHECK(hipMemcpyAsync((int32_t*)k.offsets.p + job.row0, q, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, w.stream));
        if (nbytes > 0) {
          char* d = stage((size_t)nbytes);
          std::memcpy(d, (const char*)a->buffers[2] + o0, (size_t)nbytes);
          HIPCHECK(hipMemcpyAsync((char*)k.data.p + job.byte0[c], d, (size_t)nbytes, hipMemcpyHostToDevice, w.stream));
        }
        bytes_copied += (n + 1) * 4 + nbytes;
      } else if (k.type.id == T_BOOL) {
        if (job.row0 & 7) throw Unsupported("ingest: a Boolean column needs every batch but the last to hold a multiple of 8 rows");
        const size_t vb = (size_t)(n + 7) / 8;
        char* q = stage(vb); std::memset(q, 0, vb);
        copy_bits_host((uint8_t*)q, 0, (const uint8_t*)a->buffers[1], off, n);
        HIPCHECK(hipMemcpyAsync((char*)k.data.p + (job.row0 >> 3), q, vb, hipMemcpyHostToDevice, w.stream));
        bytes_copied += (int64_t)vb;
      } else {
        const size_t wd = (size_t)type_width(k.type), bytes = (size_t)n * wd;
        char* q = stage(bytes);
        std::memcpy(q, (const char*)a->buffers[1] + (size_t)off * wd, bytes);
        HIPCHECK(hipMemcpyAsync((char*)k.data.p + (size_t)job.row0 * wd, q, bytes, hipMemcpyHostToDevice, w.stream));
        bytes_copied += (int64_t)bytes;
      }
    }
    HIPCHECK(hipEventRecord(w.ev, w.stream));
    HIPCHECK(hipEventSynchronize(w.ev));
  }
};

extern "C" {

const char* gpuq_ingest_last_error(void) { return g_ierr.c_str(); }

int gpuq_ingest_create(gpuq_ctx* ctx, const struct ArrowSchema* schema, int64_t max_rows, int64_t max_utf8_bytes, int n_threads, gpuq_ingest** out) {
  if (out) *out = nullptr;
  return guarded_i([&]() {
    if (!ctx || !schema || !out) throw std::runtime_error("ctx / schema / out is NULL");
    if (std::string(schema->format ? schema->format : "") != "+s") throw std::runtime_error("expected a struct-typed ArrowSchema (RecordBatch)");
    if (max_rows < 0 || max_rows > 0xFFFFFFFEll) throw std::runtime_error("max_rows out of range");
    HIPCHECK(hipSetDevice(ctx->device));
    std::unique_ptr<gpuq_ingest> g(new gpuq_ingest());
    g->ctx = ctx; g->cap_rows = max_rows;
    for (int64_t c = 0; c < schema->n_children; ++c) {
      const ArrowSchema* f = schema->children[c];
      if (f->dictionary) throw Unsupported("dictionary-encoded column");
      g->cols.emplace_back();
      gpuq_ingest::Col& k = g->cols.back(); k.type = dtype_from_arrow_format(f->format, &k.large); k.nullable = (f->flags & 2) != 0; k.name = f->name ? f->name : "";
    }
    (void)use_stream(nullptr);
    for (auto& k : g->cols) {
      const size_t bm = (size_t)((max_rows + 63) / 64) * 8 + 16;
      if (k.type.id == T_UTF8) { k.offsets.ensure((size_t)(max_rows + 1) * 4 + 16); k.bytes_cap = max_utf8_bytes; k.data.ensure((size_t)max_utf8_bytes + 16); HIPCHECK(hipMemset(k.offsets.p, 0, 4)); }
      else if (k.type.id == T_BOOL) k.data.ensure(bm);
      else k.data.ensure((size_t)max_rows * (size_t)type_width(k.type) + 16);
      if (k.nullable) k.validity.ensure(bm);
    }
    g->next_byte.assign(g->cols.size(), 0);
    int nt = n_threads > 0 ? n_threads : 8; if (nt > 32) nt = 32;
    g->workers.resize((size_t)nt);
    gpuq_ingest* raw = g.get();
    for (int i = 0; i < nt; ++i) {
      HIPCHECK(hipStreamCreateWithFlags(&raw->workers[(size_t)i].stream, hipStreamNonBlocking));
      HIPCHECK(hipEventCreateWithFlags(&raw->workers[(size_t)i].ev, hipEventDisableTiming));
    }
    for (int i = 0; i < nt; ++i) raw->workers[(size_t)i].th = std::thread([raw, i]() { raw->run_worker(i); });
    *out = g.release();
  });
}

int gpuq_ingest_push(gpuq_ingest* g, struct ArrowArray* batch) {
  return guarded_i([&]() {
    if (!g || !batch) throw std::runtime_error("ingest / batch is NULL");
    if (batch->n_children != (int64_t)g->cols.size()) throw std::runtime_error("batch has " + std::to_string(batch->n_children) + " columns, the ingest schema has " + std::to_string(g->cols.size()));
    const int64_t n = batch->length;
    if (g->next_row + n > g->cap_rows) throw Capacity("ingest: more rows than max_rows (" + std::to_string(g->cap_rows) + ")");
    std::unique_ptr<gpuq_ingest::Job> job(new gpuq_ingest::Job());
    job->row0 = g->next_row; job->rows = n; job->byte0 = g->next_byte;
    for (size_t c = 0; c < g->cols.size(); ++c) {
      if (g->cols[c].type.id != T_UTF8 || n == 0) continue;
      const ArrowArray* a = batch->children[c];
      const int64_t nb = gpuq_ingest::utf8_off(g->cols[c], a, a->offset + batch->offset + n) - gpuq_ingest::utf8_off(g->cols[c], a, a->offset + batch->offset);
      if (nb < 0) throw std::runtime_error("ingest: Utf8 offsets of column '" + g->cols[c].name + "' decrease");
      if (g->next_byte[c] + nb > g->cols[c].bytes_cap || g->next_byte[c] + nb > 0x7FFFFFFFll) throw Capacity("ingest: column '" + g->cols[c].name + "' exceeds max_utf8_bytes");
      g->next_byte[c] += nb;
    }
    g->next_row += n;
    // the batch is MOVED (Arrow C Data Interface): the library releases it once its copies have landed
    job->batch = *batch; batch->release = nullptr;
    {
      std::lock_guard<std::mutex> lk(g->mu);
      if (!g->error.empty()) { job->batch.release(&job->batch); throw std::runtime_error(g->error); }
      job->seq = g->pushed++;
      g->done_flags.push_back(0); g->rows_of.push_back(n);
      g->queue.push_back(job.release());
    }
    g->cv_work.notify_one();
  });
}

int gpuq_ingest_rows_landed(gpuq_ingest* g, int64_t* rows_out) {
  if (!g || !rows_out) return GPUQ_ERR_INVALID;
  std::lock_guard<std::mutex> lk(g->mu);
  *rows_out = g->rows_prefix;
  return GPUQ_OK;
}

int gpuq_ingest_wait_rows(gpuq_ingest* g, int64_t rows, int64_t* rows_out) {
  return guarded_i([&]() {
    if (!g) throw std::runtime_error("ingest is NULL");
    std::unique_lock<std::mutex> lk(g->mu);
    // the copies are synchronised by their worker before a batch counts as landed: device memory is valid for any stream afterwards
    g->cv_done.wait(lk, [&] { return !g->error.empty() || g->rows_prefix >= rows || g->completed_prefix == g->pushed; });
    if (!g->error.empty()) throw std::runtime_error(g->error);
    if (rows_out) *rows_out = g->rows_prefix;
  });
}

int gpuq_ingest_columns(gpuq_ingest* g, gpuq_column* cols_out, gpuq_field_info* fields_out, int cap, int* n_out) {
  return guarded_i([&]() {
    if (!g || !n_out) throw std::runtime_error("ingest / n_out is NULL");
    *n_out = (int)g->cols.size();
    if (!cols_out || cap < (int)g->cols.size()) { if (!cols_out && cap == 0) return; throw Capacity("ingest has " + std::to_string(g->cols.size()) + " columns"); }
    for (size_t c = 0; c < g->cols.size(); ++c) {
      const gpuq_ingest::Col& k = g->cols[c];
      gpuq_column o{}; o.type = k.type.id; o.precision = k.type.p; o.scale = k.type.s; o.repr = GPUQ_REPR_ARROW;
      o.data = k.data.p; o.offsets = k.type.id == T_UTF8 ? (const int32_t*)k.offsets.p : nullptr; o.validity = k.nullable ? (const uint8_t*)k.validity.p : nullptr;
      o.length = g->cap_rows;      // a consumer narrows `length` to the landed prefix it reads
      cols_out[c] = o;
      if (fields_out) {
        gpuq_field_info f{}; std::snprintf(f.name, sizeof(f.name), "%s", k.name.c_str());
        f.type = k.type.id; f.precision = k.type.p; f.scale = k.type.s; f.nullable = k.nullable; f.repr = GPUQ_REPR_ARROW; f.width = k.type.id == T_BOOL ? 0 : type_width(k.type);
        fields_out[c] = f;
      }
    }
  });
}

int gpuq_ingest_stats(gpuq_ingest* g, int64_t* rows_pushed, int64_t* rows_landed, int64_t* bytes_copied) {
  if (!g) return GPUQ_ERR_INVALID;
  std::lock_guard<std::mutex> lk(g->mu);
  if (rows_pushed) *rows_pushed = g->next_row;
  if (rows_landed) *rows_landed = g->rows_prefix;
  if (bytes_copied) *bytes_copied = g->bytes_copied.load();
  return GPUQ_OK;
}

void gpuq_ingest_free(gpuq_ingest* g) {
  if (!g) return;
  { std::lock_guard<std::mutex> lk(g->mu); g->stop = true; }
  g->cv_work.notify_all();
  for (auto& w : g->workers) if (w.th.joinable()) w.th.join();
  { std::lock_guard<std::mutex> lk(g->mu); for (auto* j : g->queue) { if (j->batch.release) j->batch.release(&j->batch); delete j; } g->queue.clear(); }
  (void)hipSetDevice(g->ctx->device);
  for (auto& w : g->workers) { if (w.stream) { (void)hipStreamSynchronize(w.stream); (void)hipStreamDestroy(w.stream); } if (w.ev) (void)hipEventDestroy(w.ev); if (w.pin) (void)hipHostFree(w.pin); }
  delete g;
}

}  // extern "C"
