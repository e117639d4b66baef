// Device buffers recycled through a process-wide pool, and the exception types that map to the C ABI's status codes.
// Shared by capi.cpp (operators) and plan_exec.cpp (native plan executor).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace gpuq {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };
struct Capacity : std::runtime_error { using std::runtime_error::runtime_error; };
struct Retry : std::runtime_error { using std::runtime_error::runtime_error; };      // deferred execution: an assumption did not hold (GPUQ_ERR_RETRY)

#define HIPCHECK(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e));             \
  } while (0)

// The stream the calling thread is currently queueing work on.  Every C entry point that takes a stream sets it first
// (use_stream); DevBuf remembers the stream it was last (re)allocated under, and the pool orders a block's next user behind
// its previous one when the two differ.
inline hipStream_t& tls_stream() { thread_local hipStream_t s = nullptr; return s; }
inline hipStream_t use_stream(void* s) { tls_stream() = (hipStream_t)s; return (hipStream_t)s; }

// Device allocations are recycled through a process-wide pool: hipMalloc + hipFree of a join table cost ~0.55 ms per join
// (a q5 run builds five), more than most of the kernels around them.  A released block is handed to the next request it
// fits (cap within 2x).  A block released under stream A carries an event recorded on A at release time; a taker on the same
// stream relies on stream order, a taker on ANY OTHER stream (another host thread running its own task on the device, as the
// reference's task-runner pool does: cpu_bound_executor.rs:94-131) first makes its stream wait for that event, so kernels of
// the previous owner that are still in flight finish before the new owner's first kernel touches the block.
// Memory budget (the reference runs its operators under DataFusion's MemoryPool: RuntimeConfig::with_memory_limit, SortExec /
// HashJoinExec reserve what they hold and fail with ResourcesExhausted -- or, for the sort, spill -- when the pool is used up): every
// byte an operator, plan or table of this library holds is a DevBuf, so the pool counts them.  `limit` (0 = none; gpuq_memory_limit or
// GPUQ_MEMORY_LIMIT) bounds that count: the allocation that would cross it fails with GPUQ_ERR_CAPACITY "Resources exhausted" and the
// task fails loudly instead of taking the device down for its neighbours.  Cached free blocks do not count (they are given back to the
// driver before an allocation is refused); the caller's own input columns are the caller's.
struct DevPool {
  struct Blk { void* p; size_t cap; int dev; hipStream_t st; hipEvent_t ev; };
  std::mutex mu; std::vector<Blk> free_; std::vector<hipEvent_t> events_; size_t held = 0;
  size_t in_use = 0, peak = 0, limit = 0;
  DevPool() { if (const char* e = std::getenv("GPUQ_MEMORY_LIMIT")) limit = (size_t)std::strtoull(e, nullptr, 10); }
  void charge(size_t bytes) {
    std::lock_guard<std::mutex> lk(mu);
    if (limit && in_use + bytes > limit)
      throw Capacity("Resources exhausted: an allocation of " + std::to_string(bytes) + " bytes with " + std::to_string(in_use) + " in use would exceed the memory limit of " + std::to_string(limit) + " bytes");
    in_use += bytes; if (in_use > peak) peak = in_use;
  }
  void uncharge(size_t bytes) { std::lock_guard<std::mutex> lk(mu); in_use = in_use >= bytes ? in_use - bytes : 0; }
  static DevPool& get() { static DevPool* P = new DevPool(); return *P; }      // leaked on purpose: no hipFree at process exit
  void* take(size_t bytes, size_t* cap_out) {
    int dev = 0; (void)hipGetDevice(&dev);
    Blk b{};
    {
      std::lock_guard<std::mutex> lk(mu);
      int best = -1;
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].dev == dev && free_[i].cap >= bytes && free_[i].cap <= 2 * bytes + (1u << 20) && (best < 0 || free_[i].cap < free_[(size_t)best].cap)) best = (int)i;
      if (best < 0) return nullptr;
      b = free_[(size_t)best]; free_.erase(free_.begin() + best); held -= b.cap;
    }
    if (b.ev) {
      const hipStream_t cur = tls_stream();
      if (b.st != cur && hipEventQuery(b.ev) != hipSuccess) {
        (void)hipGetLastError();
        if (hipStreamWaitEvent(cur, b.ev, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipEventSynchronize(b.ev); }
      }
      std::lock_guard<std::mutex> lk(mu);
      events_.push_back(b.ev);      // an event that is waited on keeps the recorded state until it is recorded again
    }
    *cap_out = b.cap; return b.p;
  }
  void give(void* p, size_t cap, hipStream_t st) {
    int dev = 0; (void)hipGetDevice(&dev);
    hipEvent_t ev = nullptr;
    { std::lock_guard<std::mutex> lk(mu); if (!events_.empty()) { ev = events_.back(); events_.pop_back(); } }
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
    if (ev && hipEventRecord(ev, st) != hipSuccess) {      // e.g. the caller destroyed the stream: nothing can be in flight on it any more
      (void)hipGetLastError();
      std::lock_guard<std::mutex> lk(mu); events_.push_back(ev); ev = nullptr;
    }
    { std::lock_guard<std::mutex> lk(mu);
      if (held + cap <= (size_t)24 << 30 && free_.size() < 256) { free_.push_back({p, cap, dev, st, ev}); held += cap; return; }
      if (ev) events_.push_back(ev); }
    (void)hipFree(p);      // synchronises with the device
  }
  void trim() {      // out of memory somewhere: give everything back and let the caller retry
    std::vector<Blk> v; { std::lock_guard<std::mutex> lk(mu); v.swap(free_); held = 0; for (auto& b : v) if (b.ev) events_.push_back(b.ev); }
    for (auto& b : v) (void)hipFree(b.p);
  }
};

struct DevBuf {
  void* p = nullptr; size_t cap = 0; hipStream_t st = nullptr;      // st: the stream of the thread that last (re)allocated or touched the buffer
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) { DevPool::get().uncharge(cap); DevPool::get().give(p, cap, st); } }
  void* ensure(size_t bytes) {
    if (bytes > cap) {
      // the old block goes back tagged with the stream that last used it (its kernels may still be in flight there), not with the
      // stream of the caller that happens to grow the buffer
      if (p) { DevPool::get().uncharge(cap); DevPool::get().give(p, cap, st); p = nullptr; cap = 0; }
      size_t want = bytes < 256 ? 256 : bytes;
      DevPool::get().charge(want);      // throws Capacity when the memory limit would be crossed (nothing is held then)
      p = DevPool::get().take(want, &cap);
      if (!p) {
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); DevPool::get().trim(); p = nullptr; e = hipMalloc(&p, want); }
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; DevPool::get().uncharge(want); throw HipError(std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); }
        cap = want;
      }
      if (cap > want) { try { DevPool::get().charge(cap - want); } catch (...) { DevPool::get().uncharge(want); DevPool::get().give(p, cap, tls_stream()); p = nullptr; cap = 0; throw; } }      // a recycled block is as large as it is
    }
    st = tls_stream();
    return p;
  }
  template <class T> T* as() const { return (T*)p; }
};


}  // namespace gpuq
