// gpuq -- host side of the ZSTD page kernel: the code object (kernels_zstd_dev.hip, embedded by build.py) is loaded as a module of its
// own on the first launch, per device, and stays loaded.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include "devbuf.h"
#include "gpuq_kernels.h"

extern "C" const unsigned char gpuq_zstd_hsaco[];      // build/embedded_zstd.cpp (.incbin of the code object)

namespace gpuq {

static hipFunction_t zstd_kernel() {
  static std::mutex mu; static std::map<int, hipFunction_t> fns;
  int dev = 0; HIPCHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu);
  auto it = fns.find(dev);
  if (it != fns.end()) return it->second;
  hipModule_t mod = nullptr; hipFunction_t fn = nullptr;
  HIPCHECK(hipModuleLoadData(&mod, gpuq_zstd_hsaco));
  HIPCHECK(hipModuleGetFunction(&fn, mod, "gpuq_k_zstd_pages"));
  fns[dev] = fn;
  return fn;
}

size_t zstd_scratch_bytes(int n_pages) { return (size_t)n_pages * (size_t)(131072 + 64); }      // zs::BLOCK_MAX + 64 per page in flight

void launch_zstd_pages(hipStream_t s, const uint8_t* src, uint8_t* dst, const UnpackJob* jobs, const int32_t* which, int n, uint8_t* scratch, uint32_t* status) {
  if (n <= 0) return;
  void* args[] = {(void*)&src, (void*)&dst, (void*)&jobs, (void*)&which, (void*)&n, (void*)&scratch, (void*)&status};
  HIPCHECK(hipModuleLaunchKernel(zstd_kernel(), (unsigned)n, 1, 1, 64, 1, 1, 0, s, args, nullptr));
}

}  // namespace gpuq
