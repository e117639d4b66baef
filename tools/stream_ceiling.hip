// Read-only streaming ceiling for the q1 column set (4 x 16 B + 4 B + 2 x (4 B + 1 B) per row) on MI355X.
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_ceiling.hip -o tools/stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;

// (a) one flat buffer, 16 B per lane per iteration, grid-stride
__global__ void __launch_bounds__(256) k_flat(const ulonglong2* __restrict__ p, long long n16, u64* out) {
  u64 acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) { ulonglong2 v = p[i]; acc += v.x ^ v.y; }
  if (acc == 0x1234567) out[0] = acc;
}
// (b) q1 pattern: per row, 4 x 16 B columns + i32 + 2 x (i32 offsets, u8 data); row per lane, UNROLL rows in flight
template <int U>
__global__ void __launch_bounds__(256) k_q1(const ulonglong2* __restrict__ c0, const ulonglong2* __restrict__ c1, const ulonglong2* __restrict__ c2,
                                            const ulonglong2* __restrict__ c3, const int* __restrict__ d, const int* __restrict__ o1, const uint8_t* __restrict__ b1,
                                            const int* __restrict__ o2, const uint8_t* __restrict__ b2, long long n, u64* out) {
  u64 acc = 0;
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long base = ((long long)blockIdx.x * 256) * U + threadIdx.x; base < n; base += stride) {
    ulonglong2 v0[U], v1[U], v2[U], v3[U]; int dd[U], oa[U], ob[U]; uint8_t ba[U], bb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long i = base + (long long)u * 256; if (i >= n) i = n - 1;
      v0[u] = c0[i]; v1[u] = c1[i]; v2[u] = c2[i]; v3[u] = c3[i]; dd[u] = d[i]; oa[u] = o1[i]; ob[u] = o2[i]; ba[u] = b1[i]; bb[u] = b2[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += (v0[u].x ^ v1[u].x) + (v2[u].x ^ v3[u].x) + v0[u].y + v1[u].y + v2[u].y + v3[u].y + dd[u] + oa[u] + ob[u] + ba[u] + bb[u];
  }
  if (acc == 0x1234567) out[0] = acc;
}

int main() {
  const long long n = 59986052;
  const size_t bytes = (size_t)n * 78; const double fbytes = (double)bytes;
  char* buf; u64* out;
  CK(hipMalloc(&buf, bytes + 4096)); CK(hipMalloc(&out, 8)); CK(hipMemset(buf, 1, bytes));
  char* buf2; CK(hipMalloc(&buf2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto col = [&](size_t off) { return buf + off; };
  size_t off = 0; const ulonglong2* c[4]; for (int k = 0; k < 4; ++k) { c[k] = (const ulonglong2*)col(off); off += (size_t)n * 16; }
  const int* d = (const int*)col(off); off += (size_t)n * 4; const int* o1 = (const int*)col(off); off += (size_t)n * 4; const int* o2 = (const int*)col(off); off += (size_t)n * 4;
  const uint8_t* b1 = (const uint8_t*)col(off); off += n; const uint8_t* b2 = (const uint8_t*)col(off); off += n;
  for (int grid : {2048, 4096, 8192, 16384, 65536}) {
    float best = 1e9;
    for (int it = 0; it < 6; ++it) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, (const ulonglong2*)buf, (long long)(bytes / 16), out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("flat 16B/lane grid=%6d: %.3f ms  %.0f GB/s\n", grid, best, fbytes / best / 1e6);
  }
#define RUNQ(U, grid) { float best = 1e9; for (int it = 0; it < 6; ++it) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_q1<U>, dim3(grid), dim3(256), 0, 0, c[0], c[1], c[2], c[3], d, o1, b1, o2, b2, n, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } printf("q1 pattern U=%d grid=%6d: %.3f ms  %.0f GB/s\n", U, grid, best, fbytes / best / 1e6); }
  for (int grid : {2048, 4096, 8192, 32768}) { RUNQ(1, grid); RUNQ(2, grid); RUNQ(4, grid); }
  { float best = 1e9; for (int it = 0; it < 5; ++it) { CK(hipEventRecord(e0)); CK(hipMemcpyAsync(buf2, buf, bytes, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("hipMemcpyDtoD %zu B: %.3f ms  copy rate %.0f GB/s (read+write traffic %.0f GB/s)\n", bytes, best, fbytes / best / 1e6, 2.0 * fbytes / best / 1e6); }
  return 0;
}
