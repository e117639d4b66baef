// How the ORDER in which resident waves walk two streamed columns (8-byte keys + 4-byte dates, the q3 lineitem probe's inputs) changes the
// read rate: (a) grid-stride -- consecutive waves read consecutive 64-row words, the resident waves cover one contiguous window that moves
// through the columns; (b) per-wave segments -- every wave owns a contiguous run of `wpw` words (what the ordered probe does: 32768 segments
// of ~18 K rows), so 8192 resident waves read 8192 separate streams; (c) per-BLOCK segments, the block's four waves interleaving words.
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_pattern.hip -o tools/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;
template <int U>
__global__ void __launch_bounds__(256) k_stride(const u64* __restrict__ k, const int* __restrict__ d, long long nwords, u64* sink) {
  u64 acc = 0; const int lane = threadIdx.x & 63; const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
  for (long long w0 = wave * U; w0 < nwords; w0 += nw * U) {
    u64 kv[U]; int dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { long long w = w0 + u; if (w >= nwords) w = nwords - 1; kv[u] = k[(w << 6) + lane]; dv[u] = d[(w << 6) + lane]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += kv[u] + dv[u];
  }
  if (acc == 0x1234567) sink[0] = acc;
}
template <int U>
__global__ void __launch_bounds__(256) k_wave_seg(const u64* __restrict__ k, const int* __restrict__ d, long long nwords, long long wpw, int nsegs, u64* sink) {
  u64 acc = 0; const int lane = threadIdx.x & 63; const long long seg = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= nsegs) return;
  const long long a = seg * wpw; long long b = a + wpw; if (b > nwords) b = nwords;
  for (long long w0 = a; w0 < b; w0 += U) {
    u64 kv[U]; int dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { long long w = w0 + u; if (w >= b) w = b - 1; kv[u] = k[(w << 6) + lane]; dv[u] = d[(w << 6) + lane]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += kv[u] + dv[u];
  }
  if (acc == 0x1234567) sink[0] = acc;
}
template <int U>
__global__ void __launch_bounds__(256) k_block_seg(const u64* __restrict__ k, const int* __restrict__ d, long long nwords, long long wpb, int nblk, u64* sink) {
  u64 acc = 0; const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long a = (long long)blockIdx.x * wpb; long long b = a + wpb; if (b > nwords) b = nwords;
  for (long long w0 = a + wv * U; w0 < b; w0 += 4 * U) {
    u64 kv[U]; int dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { long long w = w0 + u; if (w >= b) w = b - 1; kv[u] = k[(w << 6) + lane]; dv[u] = d[(w << 6) + lane]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += kv[u] + dv[u];
  }
  if (acc == 0x1234567) sink[0] = acc;
}
int main() {
  const long long n = 600037902ll, nwords = (n + 63) >> 6;
  u64* k; int* d; u64* sink;
  CK(hipMalloc(&k, (size_t)(nwords << 6) * 8)); CK(hipMalloc(&d, (size_t)(nwords << 6) * 4)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(k, 1, (size_t)(nwords << 6) * 8)); CK(hipMemset(d, 1, (size_t)(nwords << 6) * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)n * 12;
#define TIME(name, launch) { float best = 1e9, worst = 0; for (int it = 0; it < 8; ++it) { CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; if (it > 0 && ms > worst) worst = ms; } \
    printf("%-58s best %.3f ms (%.0f GB/s)  worst %.3f ms\n", name, best, bytes / best / 1e6, worst); }
  TIME("grid-stride, 2048 blocks, 4 words in flight", hipLaunchKernelGGL(k_stride<4>, dim3(2048), dim3(256), 0, 0, k, d, nwords, sink));
  TIME("grid-stride, 2048 blocks, 2 words in flight", hipLaunchKernelGGL(k_stride<2>, dim3(2048), dim3(256), 0, 0, k, d, nwords, sink));
  for (int nsegs : {32768, 131072, 524288}) {
    const long long wpw = (nwords + nsegs - 1) / nsegs; char nm[128];
    snprintf(nm, sizeof nm, "per-wave segments: %d segments of %lld words, U=4", nsegs, wpw);
    TIME(nm, hipLaunchKernelGGL(k_wave_seg<4>, dim3((nsegs + 3) / 4), dim3(256), 0, 0, k, d, nwords, wpw, nsegs, sink));
  }
  for (int nblk : {8192, 32768, 131072}) {
    const long long wpb = (nwords + nblk - 1) / nblk; char nm[128];
    snprintf(nm, sizeof nm, "per-block segments: %d blocks of %lld words, 4 waves interleaved, U=4", nblk, wpb);
    TIME(nm, hipLaunchKernelGGL(k_block_seg<4>, dim3(nblk), dim3(256), 0, 0, k, d, nwords, wpb, nblk, sink));
    snprintf(nm, sizeof nm, "per-block segments: %d blocks of %lld words, 4 waves interleaved, U=1", nblk, wpb);
    TIME(nm, hipLaunchKernelGGL(k_block_seg<1>, dim3(nblk), dim3(256), 0, 0, k, d, nwords, wpb, nblk, sink));
  }
  return 0;
}
