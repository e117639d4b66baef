// micro-test: does `v_mov_b64 v[a:b], <32-bit literal>` zero the high dword on this gfx950 + runtime?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__global__ void k(u64* out) {
  u64 v;
  asm volatile("v_mov_b64_e32 %0, 0x64" : "=v"(v));
  out[threadIdx.x] = v;
}
__global__ void k2(u64* out, const unsigned* in) {
  unsigned r = in[threadIdx.x];
  out[threadIdx.x] = (u64)r * 100ull + 100ull;
}
int main() {
  u64* d; unsigned* in; hipMalloc(&d, 64 * 8); hipMalloc(&in, 64 * 4);
  unsigned hin[64]; for (int i = 0; i < 64; ++i) hin[i] = i; hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
  u64 h[64];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) if (h[i] != 0x64) { if (bad < 5) printf("k lane %d: %llx\n", i, h[i]); ++bad; }
  printf("v_mov_b64 literal: %d bad lanes\n", bad);
  hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d, in); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  bad = 0; for (int i = 0; i < 64; ++i) if (h[i] != (u64)i * 100 + 100) { if (bad < 5) printf("k2 lane %d: %llx\n", i, h[i]); ++bad; }
  printf("mad u64: %d bad lanes\n", bad);
  return 0;
}
