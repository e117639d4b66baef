// gpuq -- ZSTD Parquet pages: one wave per page (the decoder itself: zstd_dec.h).
//
// This file is NOT linked into libgpuq.so's fat binary: build.py compiles it to a code object of its own (`--genco`), embeds the bytes,
// and zstd_launch.cpp loads them as a module when the first ZSTD page arrives.  The runtime loads a library's device code when its first
// kernel is launched, and this one kernel is as large as the rest of the scan path together: inside the fat binary it cost every process
// 0.25-0.5 s of its first launch (SF100 q3's first execution 33 -> 275-520 ms), whether or not it ever saw a ZSTD file.
#include <hip/hip_runtime.h>
#include "gpuq_kernels.h"
#include "zstd_dec.h"

// jobs[which[b]] with mode 4: frames at src + raw_prefix -> dst + raw_prefix (the prefix itself is copied by the unpack kernels);
// scratch: (BLOCK_MAX + 64) bytes per workgroup of the launch (the literals of the block being decoded)
extern "C" __global__ __launch_bounds__(64) void gpuq_k_zstd_pages(const uint8_t* __restrict__ src_base, uint8_t* dst_base, const gpuq::UnpackJob* __restrict__ jobs,
                                                                   const int32_t* __restrict__ which, int n, uint8_t* scratch, uint32_t* __restrict__ status) {
  using namespace gpuq;
  __shared__ zs::Shared S;
  // (no output ring: measured, it does not shorten a sequence -- the wave is issue-bound -- and its 32 KB would halve the pages in flight per CU)
  __shared__ __attribute__((aligned(16))) uint8_t litw[zs::LITW];
  __shared__ __attribute__((aligned(16))) uint64_t bitw[zs::BITW / 8 + 2];
  __shared__ __attribute__((aligned(16))) uint64_t hufw[4 * (zs::HUFW / 8 + 2)];
  const int b = (int)blockIdx.x;
  if (b >= n) return;
  const UnpackJob J = jobs[which[b]];
  if (J.mode != 4 || J.raw_prefix < 0 || J.raw_prefix > J.src_len || J.raw_prefix > J.dst_len) { if (threadIdx.x == 0) atomicOr(status, 1u); return; }
  const bool ok = zs::decode_frames(src_base + J.src + J.raw_prefix, J.src_len - J.raw_prefix, dst_base + J.dst + J.raw_prefix, J.dst_len - J.raw_prefix,
                                    scratch + (size_t)b * (size_t)(zs::BLOCK_MAX + 64), S, zs::Lds{nullptr, litw, bitw, hufw});
  if (!ok && threadIdx.x == 0) atomicOr(status, 1u);
}
