#pragma once
#include <string>
namespace gpuq {
struct JitFn { void* module = nullptr; void* fn = nullptr; };
bool jit_available();
// compiled + loaded function for (front-end source, sink kernel id); cached per process; throws on failure
const JitFn* jit_get(const std::string& eval_src, int kernel_id);
// non-blocking: the function if it is ready; otherwise nullptr, and (once per key) a background thread starts compiling it
const JitFn* jit_try_get(const std::string& eval_src, int kernel_id);
void jit_drain();      // waits until every background compile requested so far has finished
void jit_cache_stats(int* disk_hits, int* compiles);      // code objects taken from the on-disk cache / compiled by hiprtc, this process
std::string jit_full_source(const std::string& eval_src, int kernel_id);
}  // namespace gpuq
