// feasibility: compile a kernel string with hiprtc for gfx950, load and run it; report compile time
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>
static const char* SRC = R"(
typedef unsigned long long u64; typedef long long i64; typedef __int128 i128;
extern "C" __global__ void k(const u64* a, const u64* b, u64* out, long n) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) { i128 x = (i128)(i64)a[i] * (i128)(i64)b[i]; out[2*i] = (u64)x; out[2*i+1] = (u64)((unsigned __int128)x >> 64); }
}
)";
int main() {
  auto t0 = std::chrono::steady_clock::now();
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, SRC, "k.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { printf("create failed\n"); return 1; }
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls); std::string log(ls, 0); if (ls) hiprtcGetProgramLog(prog, &log[0]);
  if (r != HIPRTC_SUCCESS) { printf("compile failed: %s\n", log.c_str()); return 1; }
  size_t cs = 0; hiprtcGetCodeSize(prog, &cs); std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
  auto t1 = std::chrono::steady_clock::now();
  hipModule_t mod; hipFunction_t fn;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "k") != hipSuccess) { printf("load failed\n"); return 1; }
  long n = 1024; u_int64_t *a, *b, *o; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&o, n * 16);
  std::vector<u_int64_t> ha(n), hb(n), ho(2 * n); for (long i = 0; i < n; ++i) { ha[i] = i + 3; hb[i] = 1000003 * i + 7; }
  hipMemcpy(a, ha.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), n * 8, hipMemcpyHostToDevice);
  void* args[] = {&a, &b, &o, &n};
  hipModuleLaunchKernel(fn, (n + 255) / 256, 1, 1, 256, 1, 1, 0, 0, args, nullptr);
  hipMemcpy(ho.data(), o, n * 16, hipMemcpyDeviceToHost);
  int bad = 0; for (long i = 0; i < n; ++i) if (ho[2 * i] != ha[i] * hb[i]) ++bad;
  printf("hiprtc ok: code %zu bytes, compile %.3f s, bad=%d\n", cs, std::chrono::duration<double>(t1 - t0).count(), bad);
  return bad;
}
