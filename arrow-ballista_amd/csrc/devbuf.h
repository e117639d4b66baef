// Device buffers recycled through a process-wide pool, and the exception types that map to the C ABI's status codes.
// Shared by capi.cpp (operators) and plan_exec.cpp (native plan executor).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace gpuq {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };
struct Capacity : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIPCHECK(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e));             \
  } while (0)

// Device allocations are recycled through a process-wide pool: hipMalloc + hipFree of a join table cost ~0.55 ms per join
// (a q5 run builds five), more than most of the kernels around them.  A released block is handed to the next request it
// fits (cap within 2x).  Reuse relies on stream order: all calls of a context are issued on one stream (or are ordered
// by the caller), so whoever reuses a block is queued behind the kernels that last touched it.
struct DevPool {
  struct Blk { void* p; size_t cap; int dev; };
  std::mutex mu; std::vector<Blk> free_; size_t held = 0;
  static DevPool& get() { static DevPool* P = new DevPool(); return *P; }      // leaked on purpose: no hipFree at process exit
  void* take(size_t bytes, size_t* cap_out) {
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    int best = -1;
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i].dev == dev && free_[i].cap >= bytes && free_[i].cap <= 2 * bytes + (1u << 20) && (best < 0 || free_[i].cap < free_[(size_t)best].cap)) best = (int)i;
    if (best < 0) return nullptr;
    Blk b = free_[(size_t)best]; free_.erase(free_.begin() + best); held -= b.cap; *cap_out = b.cap; return b.p;
  }
  void give(void* p, size_t cap) {
    int dev = 0; (void)hipGetDevice(&dev);
    { std::lock_guard<std::mutex> lk(mu);
      if (held + cap <= (size_t)24 << 30 && free_.size() < 256) { free_.push_back({p, cap, dev}); held += cap; return; } }
    (void)hipFree(p);
  }
  void trim() {      // out of memory somewhere: give everything back and let the caller retry
    std::vector<Blk> v; { std::lock_guard<std::mutex> lk(mu); v.swap(free_); held = 0; }
    for (auto& b : v) (void)hipFree(b.p);
  }
};

struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) DevPool::get().give(p, cap); }
  void* ensure(size_t bytes) {
    if (bytes > cap) {
      if (p) { DevPool::get().give(p, cap); p = nullptr; cap = 0; }
      size_t want = bytes < 256 ? 256 : bytes;
      p = DevPool::get().take(want, &cap);
      if (!p) {
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); DevPool::get().trim(); p = nullptr; HIPCHECK(hipMalloc(&p, want)); }
        cap = want;
      }
    }
    return p;
  }
  template <class T> T* as() const { return (T*)p; }
};


}  // namespace gpuq
