// Random-gather ceiling of MI355X for the uniform-key join probe: how many independent 4-byte lookups per second
// the part sustains into a table of 2^20 / 2^24 / 2^27 entries (4 MiB / 64 MiB / 512 MiB: inside L2 + Infinity Cache, beyond L2,
// beyond both) when NOTHING else is done -- keys come from a streamed 8-byte column (as the probe's do), every lane keeps U lookups
// in flight, results are summed.  The probe kernel (direct-addressed table: one 4-byte load per probe + the pair write) cannot beat
// this; the 24 B/probe + 12 B/match of SURVEY.md section 8d is priced against it in profiles/r03_gather_ceiling.txt.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_ceiling.hip -o tools/gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void __launch_bounds__(256) k_fill_keys(u64* keys, long long n, u64 mask, int sorted) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) keys[i] = sorted ? (u64)(((__int128)i * (mask + 1)) / n) : (mix((u64)i + 12345) & mask);
}
template <int U, bool PAIRS>
__global__ void __launch_bounds__(256) k_gather(const u64* __restrict__ keys, const uint32_t* __restrict__ table, long long n, uint32_t* __restrict__ out_b, uint32_t* __restrict__ out_p, u64* sink) {
  u64 acc = 0;
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long base = (long long)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
    u64 k[U]; uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { long long i = base + (long long)u * 256; if (i >= n) i = n - 1; k[u] = keys[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = table[k[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc += v[u];
      if (PAIRS) { long long i = base + (long long)u * 256; if (i < n) { out_b[i] = v[u]; out_p[i] = (uint32_t)i; } }      // 100 % hits: the pair write is a plain stream
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

int main() {
  const long long n = 1ll << 28;
  u64* keys; uint32_t* table; uint32_t *ob, *op; u64* sink;
  CK(hipMalloc(&keys, (size_t)n * 8)); CK(hipMalloc(&table, ((size_t)1 << 27) * 4)); CK(hipMalloc(&ob, (size_t)n * 4)); CK(hipMalloc(&op, (size_t)n * 4)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(table, 1, ((size_t)1 << 27) * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int sorted = 0; sorted < 2; ++sorted)
    for (int lg : {20, 24, 27}) {
      hipLaunchKernelGGL(k_fill_keys, dim3(4096), dim3(256), 0, 0, keys, n, ((u64)1 << lg) - 1, sorted);
      CK(hipDeviceSynchronize());
#define RUN(U, P, grid) { float best = 1e9; for (int it = 0; it < 5; ++it) { CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_gather<U, P>), dim3(grid), dim3(256), 0, 0, keys, table, n, ob, op, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
        printf("%s keys, table 2^%d x 4 B, U=%d rows in flight, %s grid=%5d: %.3f ms  %.1f G lookups/s  (section 8d bytes %.0f GB/s = %.3f of 8 TB/s)\n", sorted ? "sorted " : "uniform", lg, U, P ? "lookup + pair write," : "lookup only,        ", grid, best, n / best / 1e6, (P ? 36.0 : 24.0) * n / best / 1e6, (P ? 36.0 : 24.0) * n / best / 1e6 / 8000.0); }
      RUN(1, false, 4096); RUN(4, false, 4096); RUN(8, false, 4096); RUN(8, false, 16384); RUN(4, true, 4096); RUN(8, true, 4096);
    }
  return 0;
}
