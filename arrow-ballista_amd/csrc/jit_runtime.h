#pragma once
#include <string>
namespace gpuq {
struct JitFn { void* module = nullptr; void* fn = nullptr; };
bool jit_available();
// compiled + loaded function for (front-end source, sink kernel id); cached per process; throws on failure
const JitFn* jit_get(const std::string& eval_src, int kernel_id);
std::string jit_full_source(const std::string& eval_src, int kernel_id);
}  // namespace gpuq
