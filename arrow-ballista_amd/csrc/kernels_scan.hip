// Scan-side operators for gfx950: FilterExec (ordered ballot compaction),
// ProjectionExec (expression materialisation / gather) and the low-cardinality
// AggregateExec path (LDS-privatised per-lane accumulators).
//
// Replaces, on the reference's path: a5 FilterExec
// (datafusion.proto:1291-1294), a9 ProjectionExec/CoalesceBatchesExec
// (:1399-1403, :1487-1490) and a6 AggregateExec for small group counts
// (:1405-1450), as driven by ballista/core/src/execution_plans/shuffle_writer.rs:255,341.
// All of them are HBM-bound integer paths: every input byte is read once with
// coalesced 4/8/16-B-per-lane loads, expression intermediates stay in VGPRs.
#include "gpuq_kernels.h"

namespace gpuq {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ u64 uniform_u64(u64 v) {   // value known to be equal in all lanes -> SGPR pair
  return (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}

// ------------------------------------------------------------------ filter
// Pass 1: evaluate the predicate once per row, keep it as a bitmap (N/8 bytes) plus one
// count per block.  Block b owns the contiguous word range [b*wpb, (b+1)*wpb).
template <int MAXC>
__device__ __forceinline__ void k_filter_bitmap_body(const DevProgram P, const i64 n_arg, u64* __restrict__ bitmap,
                                                         uint32_t* __restrict__ block_counts, const i64 wpb) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  __shared__ uint32_t wave_cnt[WAVES];
  const i64 nwords = (n_arg + 63) >> 6;      // every word up to the BOUND is written (zeros beyond the actual rows): the compaction reads them all
  const i64 w0 = (i64)blockIdx.x * wpb;
  i64 w1 = w0 + wpb; if (w1 > nwords) w1 = nwords;
  uint32_t cnt = 0;
  for (i64 w = w0 + wave_id(); w < w1; w += WAVES) {
    const i64 pos = (w << 6) + lane_id();
    bool pass = false;
    if (pos < n) {
      GPUQ_REGS_DECL;
      pass = GPUQ_EVAL(MAXC, P, pos);
    }
    const u64 m = __ballot(pass);
    if (lane_id() == 0) bitmap[w] = m;
    cnt += (uint32_t)__popcll(m);
  }
  if (lane_id() == 0) wave_cnt[wave_id()] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int k = 0; k < WAVES; ++k) t += wave_cnt[k];
    block_counts[blockIdx.x] = t;
  }
}
#ifndef GPUQ_JIT
template <int MAXC>
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_filter_bitmap(const DevProgram P, const i64 n, u64* __restrict__ bitmap,
                                                         uint32_t* __restrict__ block_counts, const i64 wpb) { k_filter_bitmap_body<MAXC>(P, n, bitmap, block_counts, wpb); }
#endif
#elif GPUQ_JIT_KERNEL == 1
extern "C" __global__ void __launch_bounds__(BLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, u64* __restrict__ bitmap,
                                                         uint32_t* __restrict__ block_counts, const i64 wpb) { k_filter_bitmap_body<0>(P, n, bitmap, block_counts, wpb); }
#endif

// Exclusive scan of <= 1024*ITEMS block counts in place; single block of 1024 threads.
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(1024) k_scan_counts(uint32_t* __restrict__ counts, const int n, u64* __restrict__ total_out) {
  // every wave owns a contiguous 1/16 of the counts and walks it 64 at a time (coalesced loads and stores; a thread-per-range
  // layout made every load of a wave touch 64 different cache lines: 57 us for 32 Ki counts, most of it latency)
  __shared__ u64 wsum[16];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int per = ((n + 15) / 16 + 63) & ~63;            // counts per wave, a multiple of 64
  const int a = w * per, b = a + per < n ? a + per : n;
  // pass 1: the wave's total
  u64 tot = 0;
  for (int i = a + l; i < b; i += 64) tot += counts[i];
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
  if (l == 0) wsum[w] = tot;
  __syncthreads();
  if (t == 0) {
    u64 run = 0;
    for (int k = 0; k < 16; ++k) { u64 v = wsum[k]; wsum[k] = run; run += v; }
    if (total_out) *total_out = run;
  }
  __syncthreads();
  // pass 2: exclusive prefix inside the wave's range
  u64 carry = wsum[w];
  for (int i0 = a; i0 < b; i0 += 64) {
    const int i = i0 + l;
    const uint32_t c = i < b ? counts[i] : 0u;
    u64 x = c;
    for (int off = 1; off < 64; off <<= 1) { const u64 y = __shfl_up(x, off); if (l >= off) x += y; }
    if (i < b) counts[i] = (uint32_t)(carry + x - c);
    carry += __shfl(x, 63);
  }
}
#endif

// Pass 2: ordered compaction.  Each wave owns a contiguous slice of the block's words, so row
// order is preserved (FilterExec keeps input order); one coalesced store per bitmap word.
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_compact(const u64* __restrict__ bitmap, const uint32_t* __restrict__ block_offsets,
                                                   const i64 wpb, const i64 n, const uint32_t* __restrict__ sel_in,
                                                   uint32_t* __restrict__ sel_out) {
  __shared__ uint32_t wave_cnt[WAVES];
  const i64 nwords = (n + 63) >> 6;
  const i64 w0 = (i64)blockIdx.x * wpb;
  i64 w1 = w0 + wpb; if (w1 > nwords) w1 = nwords;
  const i64 span = w1 > w0 ? (w1 - w0) : 0;
  const i64 per = (span + WAVES - 1) / WAVES;
  const i64 a = w0 + per * wave_id();
  i64 b = a + per; if (b > w1) b = w1;
  uint32_t cnt = 0;
  for (i64 w = a + lane_id(); w < b; w += 64) cnt += (uint32_t)__popcll(bitmap[w]);
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (lane_id() == 0) wave_cnt[wave_id()] = cnt;
  __syncthreads();
  u64 out = block_offsets[blockIdx.x];
  for (int k = 0; k < wave_id(); ++k) out += wave_cnt[k];
  for (i64 w = a; w < b; ++w) {
    const u64 m = bitmap[w];
    const int l = lane_id();
    if ((m >> l) & 1) {
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << l) - 1));
      const i64 pos = (w << 6) + l;
      sel_out[out + rank] = sel_in ? sel_in[pos] : (uint32_t)pos;
    }
    out += (uint32_t)__popcll(m);
  }
}
#endif

// ------------------------------------------------------------------ project
template <int MAXC>
__device__ __forceinline__ void k_project_body(const DevProgram P, const i64 n_arg, const OutSpec O) {
  const i64 n = rows_of(P, n_arg);
  const i64 nwords = (n + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * WAVES + wave_id(); w < nwords; w += (i64)gridDim.x * WAVES) {
    const i64 pos = (w << 6) + lane_id();
    const bool active = pos < n;
    GPUQ_REGS_DECL;
    if (active) (void)GPUQ_EVAL(MAXC, P, pos);
#pragma unroll
    for (int k = 0; k < MAX_OUTS; ++k) {
      if (k < O.n_out) {
        const OutCol oc = O.cols[k];
        const int r = __builtin_amdgcn_readfirstlane(oc.reg);
        u64 lo = 0, hi = 0; bool isnull = true;
        if (active) { lo = rlo[r]; hi = rhi[r]; isnull = (rnulls >> r) & 1; }
        if (oc.validity) {
          const u64 vm = __ballot(active && !isnull);
          if (lane_id() == 0) oc.validity[w] = vm;
        }
        if (isnull) { lo = 0; hi = 0; }
        switch (oc.cls) {
          case CC_I32: case CC_U32: if (active) ((uint32_t*)oc.data)[pos] = (uint32_t)lo; break;
          case CC_I64: if (active) ((u64*)oc.data)[pos] = lo; break;
          case CC_I8: case CC_U8: if (active) ((uint8_t*)oc.data)[pos] = (uint8_t)lo; break;
          case CC_I16: case CC_U16: if (active) ((uint16_t*)oc.data)[pos] = (uint16_t)lo; break;
          case CC_F32: if (active) ((float*)oc.data)[pos] = (float)__longlong_as_double((i64)lo); break;
          case CC_I128: case CC_STR: if (active) ((ulonglong2*)oc.data)[pos] = make_ulonglong2(lo, hi); break;
          case CC_BIT: { const u64 bm = __ballot(active && lo != 0); if (lane_id() == 0) ((u64*)oc.data)[w] = bm; break; }
          default: break;
        }
      }
    }
  }
}
#ifndef GPUQ_JIT
template <int MAXC>
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_project(const DevProgram P, const i64 n, const OutSpec O) { k_project_body<MAXC>(P, n, O); }
#endif
#elif GPUQ_JIT_KERNEL == 2
extern "C" __global__ void __launch_bounds__(BLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const OutSpec O) { k_project_body<0>(P, n, O); }
#endif

// ------------------------------------------------------------------ tiny-group aggregate
// LDS layout (dynamic, 16-B aligned):
//   lane_acc [cells][BLOCK] u64     per-lane 64-bit partials, conflict-free (lane-major)
//   wide     [cells][2]     u64     block-shared 128-bit spill accumulators (atomic)
//   dkeys    [gmax][n_keys][2] u64  block dictionary of group keys
//   dnulls   [gmax] u32, dict_n u32, lock u32
// cells = gmax * n_accs.  A lane keeps each SUM in a private 64-bit slot and only spills to the
// shared 128-bit cell when the slot could overflow or the value does not fit 64 bits, so the hot
// loop has no atomics and no bank conflicts.
struct TinyLds {
  u64* lane_acc; u64* wide; u64* dkeys; uint32_t* dnulls; uint32_t* dict_n; uint32_t* lock;
};
__device__ __forceinline__ TinyLds tiny_carve(char* smem, int gmax, int n_keys, int n_accs) {
  TinyLds L;
  const int cells = gmax * n_accs;
  L.lane_acc = (u64*)smem; smem += (size_t)cells * BLOCK * 8;
  L.wide = (u64*)smem; smem += (size_t)cells * 16;
  L.dkeys = (u64*)smem; smem += (size_t)gmax * (n_keys > 0 ? n_keys : 1) * 16;
  L.dnulls = (uint32_t*)smem; smem += (size_t)gmax * 4;
  L.dict_n = (uint32_t*)smem; smem += 4;
  L.lock = (uint32_t*)smem;
  return L;
}
static size_t tiny_lds_bytes(int gmax, int n_keys, int n_accs) {
  const size_t cells = (size_t)gmax * n_accs;
  return cells * BLOCK * 8 + cells * 16 + (size_t)gmax * (n_keys > 0 ? n_keys : 1) * 16 + (size_t)gmax * 4 + 16;
}

__device__ __forceinline__ u64 acc_identity(int kind) {
  switch (kind) {
    case ACC_MIN: return 0x7FFFFFFFFFFFFFFFull;
    case ACC_MAX: return 0x8000000000000000ull;
    case ACC_FMIN: return 0x7FF0000000000000ull;  // +inf
    case ACC_FMAX: return 0xFFF0000000000000ull;  // -inf
    default: return 0;
  }
}

__device__ __forceinline__ void wide_add(u64* wide, int cell, u64 lo, u64 hi) {
  // 128-bit add into an LDS cell with two 64-bit atomics (adds commute, so the carry can trail)
  const u64 old = atomicAdd(&wide[2 * cell], lo);
  const u64 carry = (old + lo < old) ? 1 : 0;
  if (hi + carry) atomicAdd(&wide[2 * cell + 1], hi + carry);
}

// per-block partial record in the workspace
struct TinyPartialHdr { uint32_t n; uint32_t pad; };
static size_t tiny_partial_bytes(int gmax, int n_keys, int n_accs) {
  return 8 + (size_t)gmax * (n_keys > 0 ? n_keys : 1) * 16 + (size_t)gmax * 4 + ((gmax & 1) ? 4 : 0) + (size_t)gmax * n_accs * 16;
}

#ifdef GPUQ_MARKERS
#define GPUQ_MARK(name) asm volatile("; gpuq-mark " name)
#else
#define GPUQ_MARK(name)
#endif

// LDS words shared between the waves of a block.  Relaxed workgroup-scope atomics compile to plain ds_read / ds_write;
// a volatile generic pointer compiles to flat_load sc0 sc1 followed by s_waitcnt vmcnt(0), which drains every global load
// the wave has in flight (and with it the row pipeline below).  The fences are restricted to the local address space for
// the same reason: ordering between LDS accesses only needs lgkmcnt.
__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ u64 lds_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define LDS_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
#define LDS_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")

template <int MAXC>
__device__ __forceinline__ void k_agg_tiny_body(const DevProgram P, const i64 n_arg, const AggSpec A, const int gmax_arg,
                                                    char* __restrict__ workspace, const size_t partial_stride) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Under the JIT the aggregate's shape is a compile-time constant (capi.cpp emits JIT_* into the source):
  // loops unroll, the switch over accumulator kinds folds, every register index is static.
#ifdef GPUQ_JIT_SPEC
  constexpr int n_keys = JIT_NKEYS, n_accs = JIT_NACCS;
  constexpr int gmax = JIT_GMAX;
  (void)gmax_arg;
#define SPEC_KEY_REG(k) JIT_KEY_REG[k]
#define SPEC_ACC_KIND(a) JIT_ACC_KIND[a]
#define SPEC_ACC_REG(a) JIT_ACC_REG[a]
#define SPEC_ACC_BITS(a) JIT_ACC_BITS[a]
#define SPEC_UNROLL _Pragma("unroll")
#else
  const int n_keys = A.n_keys, n_accs = A.n_accs;
  const int gmax = gmax_arg;
#define SPEC_KEY_REG(k) __builtin_amdgcn_readfirstlane(A.key_reg[k])
#define SPEC_ACC_KIND(a) A.acc_kind[a]
#define SPEC_ACC_REG(a) __builtin_amdgcn_readfirstlane(A.acc_reg[a])
#define SPEC_ACC_BITS(a) 127
#define SPEC_UNROLL
#endif
  const int kstride = n_keys > 0 ? n_keys : 1;
  const TinyLds L = tiny_carve(smem, gmax, n_keys, n_accs);
  const int cells = gmax * n_accs;
  const int tid = threadIdx.x;
#ifndef GPUQ_EXP_LOADS_ONLY
  for (int c = 0; c < cells; ++c) L.lane_acc[c * BLOCK + tid] = acc_identity(SPEC_ACC_KIND(c % n_accs));
  for (int c = tid; c < cells * 2; c += BLOCK) L.wide[c] = 0;
#endif
  // an aggregate without GROUP BY always has exactly one group, even over zero rows
  if (tid == 0) { *L.dict_n = (n_keys == 0) ? 1u : 0u; *L.lock = 0; L.dnulls[0] = 0; }
  __syncthreads();


#ifdef GPUQ_JIT_SPEC
  u64 dc_lo[JIT_GMAX][JIT_NKC], dc_hi[JIT_GMAX][JIT_NKC]; uint32_t dc_nl[JIT_GMAX]; uint32_t cached_n = 0;
#pragma unroll
  for (int g = 0; g < JIT_GMAX; ++g) { dc_nl[g] = 0;
#pragma unroll
    for (int k = 0; k < JIT_NKC; ++k) { dc_lo[g][k] = 0; dc_hi[g][k] = 0; } }
#endif
  const i64 nwords = (n + 63) >> 6;
  // one evaluated row -> its group -> its accumulators
  auto accumulate_row = [&](bool active, GPUQ_REGS_PARAM) __attribute__((always_inline)) -> bool {      // true: the block gave up
    GPUQ_MARK("key");
    // group key (explicit scalars: a small array here ends up in scratch once the loader's slots are live)
    u64 k0lo = 0, k0hi = 0, k1lo = 0, k1hi = 0, k2lo = 0, k2hi = 0, k3lo = 0, k3hi = 0; uint32_t knull = 0;
#define GPUQ_KLO(k) ((k) == 0 ? k0lo : (k) == 1 ? k1lo : (k) == 2 ? k2lo : k3lo)
#define GPUQ_KHI(k) ((k) == 0 ? k0hi : (k) == 1 ? k1hi : (k) == 2 ? k2hi : k3hi)
#pragma unroll
    for (int k = 0; k < MAX_KEYS; ++k) {
      if (k < n_keys && active) {
        const int r = SPEC_KEY_REG(k);
        const bool isn = (rnulls >> r) & 1;
        const u64 vlo = isn ? 0 : rlo[r], vhi = isn ? 0 : rhi[r];
        if (k == 0) { k0lo = vlo; k0hi = vhi; } else if (k == 1) { k1lo = vlo; k1hi = vhi; } else if (k == 2) { k2lo = vlo; k2hi = vhi; } else { k3lo = vlo; k3hi = vhi; }
        knull |= (uint32_t)isn << k;
      }
    }
    GPUQ_MARK("lookup");
    int gid = -1;
    uint32_t seen = 0;
#ifndef GPUQ_EXP_NO_LOOKUP
#ifdef GPUQ_JIT_SPEC
    // fast path: every active row finds its group among the entries already cached in registers -- no LDS access at all
    // (the dictionary of a low-cardinality aggregate is complete after the first few steps)
    if (active) {
#pragma unroll
      for (int g = 0; g < JIT_GMAX; ++g) {
        if ((uint32_t)g < cached_n) {
          bool eq = dc_nl[g] == knull;
#pragma unroll
          for (int k = 0; k < JIT_NKC; ++k) if (k < n_keys) eq = eq && dc_lo[g][k] == GPUQ_KLO(k) && dc_hi[g][k] == GPUQ_KHI(k);
          if (eq) gid = g;
        }
      }
    }
    if (__ballot(active && gid < 0) != 0)
#endif
    for (;;) {
      // lock-free lookup over the dictionary entries published so far.  Bit 31 of the counter = "this block ran out of
      // dictionary entries": the launch's result is discarded by the host, so the block stops working (without this every
      // row of the group that did not fit fights for the block lock: q5's 5 groups against a 4-entry first try took
      // 1.67 ms where the successful second try takes 0.06 ms).
      const uint32_t nd_raw = lds_ld(L.dict_n);
      if (nd_raw & 0x80000000u) return true;
      const uint32_t nd = nd_raw;
      LDS_ACQUIRE();
#ifdef GPUQ_JIT_SPEC
      // dictionary entries live in (wave-uniform) registers and are refreshed only when the block's
      // dictionary grew: the common row pays one LDS word (dict_n), not 2*n_keys*groups LDS reads
      if (nd != cached_n) {
#pragma unroll
        for (int g = 0; g < JIT_GMAX; ++g) {
          if ((uint32_t)g >= cached_n && (uint32_t)g < nd) {
            dc_nl[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld(&L.dnulls[g]));
#pragma unroll
            for (int k = 0; k < JIT_NKC; ++k) {
              if (k < n_keys) { dc_lo[g][k] = uniform_u64(lds_ld(&L.dkeys[(g * kstride + k) * 2])); dc_hi[g][k] = uniform_u64(lds_ld(&L.dkeys[(g * kstride + k) * 2 + 1])); }
            }
          }
        }
        cached_n = nd;
      }
      if (active && gid < 0) {
#pragma unroll
        for (int g = 0; g < JIT_GMAX; ++g) {
          if ((uint32_t)g < cached_n) {
            bool eq = dc_nl[g] == knull;
#pragma unroll
            for (int k = 0; k < JIT_NKC; ++k) if (k < n_keys) eq = eq && dc_lo[g][k] == GPUQ_KLO(k) && dc_hi[g][k] == GPUQ_KHI(k);
            if (eq) gid = g;
          }
        }
      }
#else
      if (active && gid < 0) {
        for (uint32_t g = seen; g < nd; ++g) {
          bool eq = lds_ld(&L.dnulls[g]) == knull;
#pragma unroll
          for (int k = 0; k < MAX_KEYS; ++k)
            if (k < n_keys) eq = eq && lds_ld(&L.dkeys[(g * kstride + k) * 2]) == GPUQ_KLO(k) && lds_ld(&L.dkeys[(g * kstride + k) * 2 + 1]) == GPUQ_KHI(k);
          if (eq) gid = (int)g;
        }
      }
#endif
      seen = nd;
      const u64 need = __ballot(active && gid < 0);
      if (need == 0) break;
      const int leader = __ffsll((long long)need) - 1;
      if (lane_id() == leader) {
        while (atomicCAS(L.lock, 0u, 1u) != 0u) {}
        LDS_ACQUIRE();
        const uint32_t n2 = lds_ld(L.dict_n) & 0x7FFFFFFFu;
        int found = -1;
        for (uint32_t g = nd; g < n2; ++g) {
          bool eq = lds_ld(&L.dnulls[g]) == knull;
#pragma unroll
          for (int k = 0; k < MAX_KEYS; ++k)
            if (k < n_keys) eq = eq && lds_ld(&L.dkeys[(g * kstride + k) * 2]) == GPUQ_KLO(k) && lds_ld(&L.dkeys[(g * kstride + k) * 2 + 1]) == GPUQ_KHI(k);
          if (eq) found = (int)g;
        }
        if (found < 0) {
          if (n2 < (uint32_t)gmax) {
#pragma unroll
            for (int k = 0; k < MAX_KEYS; ++k)
              if (k < n_keys) { lds_st(&L.dkeys[(n2 * kstride + k) * 2], GPUQ_KLO(k)); lds_st(&L.dkeys[(n2 * kstride + k) * 2 + 1], GPUQ_KHI(k)); }
            lds_st(&L.dnulls[n2], knull);
            LDS_RELEASE();
            lds_st(L.dict_n, n2 + 1);
            found = (int)n2;
          } else {
            atomicOr(P.flags, FLAG_GROUP_OVERFLOW);
            lds_st(L.dict_n, n2 | 0x80000000u);     // tells every wave of the block to stop
            found = 0;  // result is discarded by the host when the flag is set
          }
        }
        gid = found;
        LDS_RELEASE();
        atomicExch(L.lock, 0u);
      }
      // other lanes with the same key find it in the next lookup round
    }
#endif  // GPUQ_EXP_NO_LOOKUP
    GPUQ_MARK("accumulate");
#ifdef GPUQ_EXP_NO_LOOKUP
    gid = (int)(k0hi >> 56) & 3;     // experiment: no dictionary
#endif
    // accumulate
#ifndef GPUQ_EXP_NO_ACC
    if (active) {
      SPEC_UNROLL
      for (int a = 0; a < n_accs; ++a) {
        const int kind = SPEC_ACC_KIND(a);
        const int cell = gid * n_accs + a;
        u64* slot = &L.lane_acc[cell * BLOCK + tid];
        u64 vlo = 1, vhi = 0; bool vnull = false;
        if (kind != ACC_COUNT_STAR) {
          const int r = SPEC_ACC_REG(a);
          vlo = rlo[r]; vhi = rhi[r]; vnull = (rnulls >> r) & 1;
        }
        if (vnull) continue;
        switch (kind) {
          case ACC_COUNT: case ACC_COUNT_STAR: *slot += 1; break;
          case ACC_SUM: {
            const i64 cur = (i64)*slot;
            if (SPEC_ACC_BITS(a) <= 60) {
              // the argument's declared type bounds |v| < 2^60: it always fits the 64-bit slot, only the running sum can leave
              i64 nv = cur + (i64)vlo;
              if (nv > (1ll << 62) || nv < -(1ll << 62)) { wide_add(L.wide, cell, (u64)nv, (u64)(nv >> 63)); nv = 0; }
              *slot = (u64)nv;
              break;
            }
            const bool fits = (i64)vhi == ((i64)vlo >> 63);
            const i64 v = (i64)vlo;
            // keep |slot| < 2^62 so one more 64-bit add cannot overflow
            const bool small = fits && v > -(1ll << 61) && v < (1ll << 61);
            if (small) {
              i64 nv = cur + v;
              if (nv > (1ll << 62) || nv < -(1ll << 62)) { wide_add(L.wide, cell, (u64)nv, (u64)(nv >> 63)); nv = 0; }
              *slot = (u64)nv;
            } else {
              wide_add(L.wide, cell, vlo, vhi);
            }
            break;
          }
          case ACC_MIN: case ACC_MAX: {
            const bool fits = (i64)vhi == ((i64)vlo >> 63);
            if (!fits) { atomicOr(P.flags, FLAG_WIDE_MINMAX); break; }  // reported as unsupported by the host
            const i64 cur = (i64)*slot, v = (i64)vlo;
            if (kind == ACC_MIN ? (v < cur) : (v > cur)) *slot = (u64)v;
            break;
          }
          case ACC_FSUM: *slot = (u64)__double_as_longlong(__longlong_as_double((i64)*slot) + __longlong_as_double((i64)vlo)); break;
          case ACC_FMIN: if (f64_total_key(vlo) < f64_total_key(*slot)) *slot = vlo; break;
          case ACC_FMAX: if (f64_total_key(vlo) > f64_total_key(*slot)) *slot = vlo; break;
          default: break;
        }
      }
    }
#else
    if (active && gid == 77) atomicOr(P.flags, 1u << 29);
#endif
    return false;
  };
#ifdef GPUQ_JIT
  // Software pipeline over rows.  A wave step covers PU consecutive 64-row words (PU rows per lane).  While the rows of
  // step t are evaluated and accumulated, the column loads of step t+1 ("records") and the dependent-load roots (index
  // vectors, Utf8 offsets) of step t+2 are in flight.  The LDS footprint limits a CU to 3 waves per SIMD, so the bytes
  // in flight have to come from the rows a wave keeps outstanding, not from occupancy.
  // Two record sets alternate and the loop is unrolled twice, so every record keeps its registers: rotating records by
  // assignment would read registers whose loads are still in flight.
#ifndef GPUQ_PIPE_ROWS
#define GPUQ_PIPE_ROWS 1
#endif
  constexpr int PU = GPUQ_PIPE_ROWS;
  const i64 nsteps = (nwords + PU - 1) / PU;
  const i64 tstride = (i64)gridDim.x * WAVES;
  i64 t = (i64)blockIdx.x * WAVES + wave_id();
  // Rows past the end are clamped to the last row (loaded, evaluated, never accumulated): unconditional loads keep the
  // stages free of exec-masked regions, which lets the compiler leave them in flight.
  const i64 last = n - 1;
  auto row_of = [&](i64 step, int u) { return ((step * PU + u) << 6) + lane_id(); };
  JitRaw raws[2][PU]; JitPre pres[2][PU];
#pragma unroll
  for (int u = 0; u < PU; ++u) {
    pres[0][u] = JitPre{}; pres[1][u] = JitPre{}; raws[0][u] = JitRaw{}; raws[1][u] = JitRaw{};
    if (n > 0) {
      const i64 p0 = row_of(t, u), p1 = row_of(t + tstride, u);
      gpuq_jit_pre(P, p0 < n ? p0 : last, pres[0][u]);
      gpuq_jit_pre(P, p1 < n ? p1 : last, pres[1][u]);
    }
  }
#pragma unroll
  for (int u = 0; u < PU; ++u) if (n > 0) { const i64 p0 = row_of(t, u); gpuq_jit_load(P, p0 < n ? p0 : last, pres[0][u], raws[0][u]); }
  while (t < nsteps) {
#pragma unroll
    for (int pj = 0; pj < 2; ++pj) {
      // leaving through a jump (not by skipping the second half) keeps "which loads are pending" exact on the back edge
      if (t >= nsteps) goto pipeline_done;
      // issue order matters: vmcnt retires in order, so the roots of step t+2 go out first (their slot, the roots of step
      // t, is dead), then the records of step t+1, which only wait for roots issued one step earlier
      GPUQ_MARK("issue");
#pragma unroll
      for (int u = 0; u < PU; ++u) { const i64 pp = row_of(t + 2 * tstride, u); gpuq_jit_pre(P, pp < n ? pp : last, pres[pj][u]); }
#pragma unroll
      for (int u = 0; u < PU; ++u) { const i64 pl = row_of(t + tstride, u); gpuq_jit_load(P, pl < n ? pl : last, pres[pj ^ 1][u], raws[pj ^ 1][u]); }
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const i64 pos = row_of(t, u);
        GPUQ_REGS_DECL;
        // evaluated on every lane (clamped rows are real rows) so that the record is consumed on all paths
        GPUQ_MARK("compute");
        const bool pass = gpuq_jit_compute(P, pos < n ? pos : last, raws[pj][u], GPUQ_REGS);
#if defined(GPUQ_EXP_LOADS_ONLY)
        if (pos < n && pass && rlo[3] == 0x123456789ull) atomicOr(P.flags, 1u << 30);     // experiment: loads + evaluation only
#else
        if (accumulate_row(pos < n && pass, GPUQ_REGS)) goto pipeline_done;
#endif
      }
      GPUQ_MARK("end");
      t += tstride;
    }
  }
pipeline_done:
#else
  for (i64 w = (i64)blockIdx.x * WAVES + wave_id(); w < nwords; w += (i64)gridDim.x * WAVES) {
    const i64 pos = (w << 6) + lane_id();
    bool active = pos < n;
    GPUQ_REGS_DECL;
    if (active) active = GPUQ_EVAL(MAXC, P, pos);
    if (accumulate_row(active, GPUQ_REGS)) break;
  }
#endif
  __syncthreads();
#ifdef GPUQ_EXP_LOADS_ONLY
  return;      // experiment: no accumulators were kept
#endif
  // block reduction: fold the per-lane partials of every live cell, add the wide spill cell
  __shared__ u64 red[WAVES * 2];
  const int ng = (int)(*L.dict_n & 0x7FFFFFFFu);
  char* rec = workspace + (size_t)blockIdx.x * partial_stride;
  u64* out_keys = (u64*)(rec + 8);
  uint32_t* out_nulls = (uint32_t*)(rec + 8 + (size_t)gmax * kstride * 16);
  u64* out_cells = (u64*)(rec + 8 + (size_t)gmax * kstride * 16 + (size_t)gmax * 4 + ((gmax & 1) ? 4 : 0));
  if (tid == 0) { ((uint32_t*)rec)[0] = (uint32_t)ng; ((uint32_t*)rec)[1] = 0; }
  for (int i = tid; i < ng * kstride * 2; i += BLOCK) out_keys[i] = L.dkeys[i];
  for (int i = tid; i < ng; i += BLOCK) out_nulls[i] = L.dnulls[i];
  for (int c = 0; c < ng * n_accs; ++c) {
    const int kind = SPEC_ACC_KIND(c % n_accs);
    const u64 v = L.lane_acc[c * BLOCK + tid];
    u64 lo, hi;
    if (kind == ACC_SUM) { lo = v; hi = (u64)((i64)v >> 63); } else { lo = v; hi = 0; }
    for (int off = 32; off > 0; off >>= 1) {
      const u64 olo = __shfl_xor(lo, off), ohi = __shfl_xor(hi, off);
      switch (kind) {
        case ACC_SUM: case ACC_COUNT: case ACC_COUNT_STAR: { const u64 s = lo + olo; hi = hi + ohi + (s < lo ? 1 : 0); lo = s; break; }
        case ACC_MIN: if ((i64)olo < (i64)lo) lo = olo; break;
        case ACC_MAX: if ((i64)olo > (i64)lo) lo = olo; break;
        // lanes combine in a fixed butterfly order -> bitwise reproducible run to run
        case ACC_FSUM: lo = (u64)__double_as_longlong(__longlong_as_double((i64)lo) + __longlong_as_double((i64)olo)); break;
        case ACC_FMIN: if (f64_total_key(olo) < f64_total_key(lo)) lo = olo; break;
        case ACC_FMAX: if (f64_total_key(olo) > f64_total_key(lo)) lo = olo; break;
        default: break;
      }
    }
    if (lane_id() == 0) { red[wave_id() * 2] = lo; red[wave_id() * 2 + 1] = hi; }
    __syncthreads();
    if (tid == 0) {
      u64 rlo = red[0], rhi = red[1];
      for (int k = 1; k < WAVES; ++k) {
        const u64 olo = red[2 * k], ohi = red[2 * k + 1];
        switch (kind) {
          case ACC_SUM: case ACC_COUNT: case ACC_COUNT_STAR: { const u64 s = rlo + olo; rhi = rhi + ohi + (s < rlo ? 1 : 0); rlo = s; break; }
          case ACC_MIN: if ((i64)olo < (i64)rlo) rlo = olo; break;
          case ACC_MAX: if ((i64)olo > (i64)rlo) rlo = olo; break;
          case ACC_FSUM: rlo = (u64)__double_as_longlong(__longlong_as_double((i64)rlo) + __longlong_as_double((i64)olo)); break;
          case ACC_FMIN: if (f64_total_key(olo) < f64_total_key(rlo)) rlo = olo; break;
          case ACC_FMAX: if (f64_total_key(olo) > f64_total_key(rlo)) rlo = olo; break;
          default: break;
        }
      }
      if (kind == ACC_SUM) {
        const u64 wlo = L.wide[2 * c], whi = L.wide[2 * c + 1];
        const u64 s = rlo + wlo; rhi = rhi + whi + (s < rlo ? 1 : 0); rlo = s;
      } else if (kind == ACC_MIN || kind == ACC_MAX) {
        rhi = (u64)((i64)rlo >> 63);
      }
      out_cells[2 * c] = rlo; out_cells[2 * c + 1] = rhi;
    }
    __syncthreads();
  }
}
#undef SPEC_KEY_REG
#undef SPEC_ACC_KIND
#undef SPEC_ACC_REG
#undef SPEC_ACC_BITS
#undef SPEC_UNROLL
#ifndef GPUQ_JIT
template <int MAXC>
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_agg_tiny(const DevProgram P, const i64 n, const AggSpec A, const int gmax,
                                                    char* __restrict__ workspace, const size_t partial_stride) { k_agg_tiny_body<MAXC>(P, n, A, gmax, workspace, partial_stride); }
#endif
#elif GPUQ_JIT_KERNEL == 3
extern "C" __global__ void __launch_bounds__(BLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const AggSpec A, const int gmax,
                                                    char* __restrict__ workspace, const size_t partial_stride) { k_agg_tiny_body<0>(P, n, A, gmax, workspace, partial_stride); }
#endif

// Merge the per-block partial records into the final groups.  One block; the work is
// nblocks*gmax records, a few thousand at most.
#ifndef GPUQ_JIT
constexpr int MERGE_BLOCK = 1024;   // one block; its serial chains of dependent loads are what the merge costs, so many threads
__global__ void __launch_bounds__(MERGE_BLOCK) k_agg_tiny_merge(const AggSpec A, const int gmax, const char* __restrict__ workspace,
                                                          const size_t partial_stride, const int nblocks, const AggOut out,
                                                          uint32_t* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // some block ran out of dictionary slots: the attempt is void (the host retries with a bigger capacity or another strategy),
  // and merging thousands of full partial dictionaries under one lock would cost milliseconds for nothing
  if (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FLAG_GROUP_OVERFLOW) return;
  const int n_keys = A.n_keys, n_accs = A.n_accs;
  const int kstride = n_keys > 0 ? n_keys : 1;
  const int cap = out.cap;
  // LDS: fkeys[cap][kstride][2] u64, fnulls[cap] u32, map[nblocks*gmax] u16, fn, lock
  u64* fkeys = (u64*)smem;
  uint32_t* fnulls = (uint32_t*)(smem + (size_t)cap * kstride * 16);
  uint16_t* map = (uint16_t*)((char*)fnulls + (size_t)cap * 4);
  uint32_t* fn = (uint32_t*)((char*)map + (((size_t)nblocks * gmax * 2 + 15) & ~(size_t)15));
  uint32_t* lock = fn + 1;
  const int tid = threadIdx.x;
  if (tid == 0) { *fn = 0; *lock = 0; }
  __syncthreads();
  const size_t keys_off = 8, nulls_off = 8 + (size_t)gmax * kstride * 16;
  const size_t cells_off = nulls_off + (size_t)gmax * 4 + ((gmax & 1) ? 4 : 0);
  // phase 1: map every (block, local group) to a final group id
  for (int e = tid; e < nblocks * gmax; e += MERGE_BLOCK) {
    const int b = e / gmax, g = e % gmax;
    const char* rec = workspace + (size_t)b * partial_stride;
    const uint32_t ng = ((const uint32_t*)rec)[0];
    if ((uint32_t)g >= ng) { map[e] = 0xFFFF; continue; }
    const u64* k = (const u64*)(rec + keys_off) + (size_t)g * kstride * 2;
    const uint32_t kn = ((const uint32_t*)(rec + nulls_off))[g];
    int found = -1; uint32_t seen = 0;
    while (found < 0) {
      const uint32_t nd = lds_ld(fn);
      LDS_ACQUIRE();
      for (uint32_t f = seen; f < nd && found < 0; ++f) {
        bool eq = lds_ld(&fnulls[f]) == kn;
        for (int q = 0; q < n_keys; ++q) eq = eq && lds_ld(&fkeys[(f * kstride + q) * 2]) == k[2 * q] && lds_ld(&fkeys[(f * kstride + q) * 2 + 1]) == k[2 * q + 1];
        if (eq) found = (int)f;
      }
      seen = nd;
      if (found >= 0) break;
      if (atomicCAS(lock, 0u, 1u) == 0u) {
        LDS_ACQUIRE();
        const uint32_t n2 = lds_ld(fn);
        for (uint32_t f = seen; f < n2 && found < 0; ++f) {
          bool eq = lds_ld(&fnulls[f]) == kn;
          for (int q = 0; q < n_keys; ++q) eq = eq && lds_ld(&fkeys[(f * kstride + q) * 2]) == k[2 * q] && lds_ld(&fkeys[(f * kstride + q) * 2 + 1]) == k[2 * q + 1];
          if (eq) found = (int)f;
        }
        if (found < 0) {
          if (n2 < (uint32_t)cap) {
            for (int q = 0; q < n_keys; ++q) { lds_st(&fkeys[(n2 * kstride + q) * 2], k[2 * q]); lds_st(&fkeys[(n2 * kstride + q) * 2 + 1], k[2 * q + 1]); }
            lds_st(&fnulls[n2], kn);
            LDS_RELEASE();
            lds_st(fn, n2 + 1);
            found = (int)n2;
          } else { atomicOr(flags, FLAG_GROUP_OVERFLOW); found = 0; }
        }
        LDS_RELEASE();
        atomicExch(lock, 0u);
      }
    }
    map[e] = (uint16_t)found;
  }
  __syncthreads();
  const int nf = (int)*fn;
  // phase 2: integer cells are folded with LDS atomics (any order gives the same bits); float sums
  // are folded by one thread per cell in block order so that results are reproducible run to run.
  u64* fcells = (u64*)(lock + 3);   // [cap][n_accs][2], 8-byte aligned by construction
  for (int c = tid; c < nf * n_accs; c += MERGE_BLOCK) {
    const int kind = A.acc_kind[c % n_accs];
    fcells[2 * c] = acc_identity(kind);
    fcells[2 * c + 1] = (kind == ACC_MAX) ? ~0ull : 0;
  }
  __syncthreads();
  bool has_float = false;
  for (int a = 0; a < n_accs; ++a) has_float = has_float || A.acc_kind[a] == ACC_FSUM;
  for (int e = tid; e < nblocks * gmax; e += MERGE_BLOCK) {
    const uint16_t f = map[e];
    if (f == 0xFFFF) continue;
    const int b = e / gmax, g = e % gmax;
    const u64* cellsb = (const u64*)(workspace + (size_t)b * partial_stride + cells_off);
    for (int a = 0; a < n_accs; ++a) {
      const int kind = A.acc_kind[a];
      const u64 olo = cellsb[(size_t)(g * n_accs + a) * 2], ohi = cellsb[(size_t)(g * n_accs + a) * 2 + 1];
      u64* dst = &fcells[((size_t)f * n_accs + a) * 2];
      switch (kind) {
        case ACC_SUM: case ACC_COUNT: case ACC_COUNT_STAR: {
          const u64 old = atomicAdd(dst, olo);
          const u64 carry = (old + olo < old) ? 1 : 0;
          if (ohi + carry) atomicAdd(dst + 1, ohi + carry);
          break;
        }
        case ACC_MIN: atomicMin((long long*)dst, (long long)olo); break;
        case ACC_MAX: atomicMax((long long*)dst, (long long)olo); break;
        case ACC_FMIN: case ACC_FMAX: {
          u64 cur = lds_ld(dst);
          for (;;) {
            const bool better = (kind == ACC_FMIN) ? (f64_total_key(olo) < f64_total_key(cur)) : (f64_total_key(olo) > f64_total_key(cur));
            if (!better) break;
            const u64 seen = atomicCAS(dst, cur, olo);
            if (seen == cur) break;
            cur = seen;
          }
          break;
        }
        default: break;
      }
    }
  }
  __syncthreads();
  if (has_float) {
    for (int c = tid; c < nf * n_accs; c += MERGE_BLOCK) {
      const int f = c / n_accs, a = c % n_accs;
      if (A.acc_kind[a] != ACC_FSUM) continue;
      double acc = 0.0;
      for (int b = 0; b < nblocks; ++b) {
        const u64* cellsb = (const u64*)(workspace + (size_t)b * partial_stride + cells_off);
        for (int g = 0; g < gmax; ++g)
          if (map[b * gmax + g] == (uint16_t)f) acc += __longlong_as_double((i64)cellsb[(size_t)(g * n_accs + a) * 2]);
      }
      fcells[2 * c] = (u64)__double_as_longlong(acc); fcells[2 * c + 1] = 0;
    }
    __syncthreads();
  }
  for (int c = tid; c < nf * n_accs; c += MERGE_BLOCK) {
    const int kind = A.acc_kind[c % n_accs];
    u64 lo = fcells[2 * c], hi = fcells[2 * c + 1];
    if (kind == ACC_MIN || kind == ACC_MAX) hi = (u64)((i64)lo >> 63);
    out.cells[(size_t)c * 2] = lo; out.cells[(size_t)c * 2 + 1] = hi;
  }
  for (int i = tid; i < nf * kstride * 2; i += MERGE_BLOCK) out.keys[i] = fkeys[i];
  for (int i = tid; i < nf; i += MERGE_BLOCK) out.key_nulls[i] = fnulls[i];
  if (tid == 0) *out.n_groups = (uint32_t)nf;
}
#endif

// ------------------------------------------------------------------ deferred execution: status and count words of a whole plan in one copy
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(64) k_gather_words(const GatherWords g, u64* __restrict__ out) {
  const int i = threadIdx.x;
  if (i < g.n) out[i] = *g.src[i];
}
void launch_gather_words(hipStream_t s, const GatherWords& g, u64* out) {
  if (g.n > 0) hipLaunchKernelGGL(k_gather_words, dim3(1), dim3(64), 0, s, g, out);
}
#endif

// ------------------------------------------------------------------ aggregate result AoS -> SoA
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_agg_emit(const AggOut raw, const int n_keys, const int n_accs, const uint32_t ng_arg, const AggSoA soa,
                                                   const uint32_t* __restrict__ ng_dev) {
  uint32_t ng = ng_arg;
  if (ng_dev) { const uint32_t nd = *ng_dev; if (nd < ng) ng = nd; }
  const int kstride = n_keys > 0 ? n_keys : 1;
  const uint32_t nwords = (ng + 63) >> 6;
  for (uint32_t w = blockIdx.x * WAVES + wave_id(); w < nwords; w += gridDim.x * WAVES) {
    const uint32_t g = (w << 6) + lane_id();
    const bool active = g < ng;
    const uint32_t kn = active ? raw.key_nulls[g] : 0u;
    for (int k = 0; k < n_keys; ++k) {
      const bool valid = active && !((kn >> k) & 1);
      if (soa.key_valid[k]) { const u64 m = __ballot(valid); if (lane_id() == 0) soa.key_valid[k][w] = m; }
      if (active) soa.key_col[k][g] = make_ulonglong2(raw.keys[((size_t)g * kstride + k) * 2], raw.keys[((size_t)g * kstride + k) * 2 + 1]);
    }
    for (int a = 0; a < n_accs; ++a)
      if (active) soa.acc_col[a][g] = make_ulonglong2(raw.cells[((size_t)g * n_accs + a) * 2], raw.cells[((size_t)g * n_accs + a) * 2 + 1]);
  }
}
#endif

// ------------------------------------------------------------------ packed Utf8 -> Arrow Utf8
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_unpack_lengths(const ulonglong2* __restrict__ packed, const i64 n, int32_t* __restrict__ lens, uint32_t* __restrict__ too_long) {
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const int32_t len = (int32_t)(packed[i].x & 0xFF);
    // a value that did not fit the 15 packed bytes keeps its original length: it cannot be turned back into text
    if (len > 15 && too_long) atomicOr(too_long, 1u);
    lens[i] = len;
  }
}
#endif
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_unpack_bytes(const ulonglong2* __restrict__ packed, const i64 n, const int32_t* __restrict__ offsets,
                                                        uint8_t* __restrict__ out) {
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const ulonglong2 v = packed[i];
    const int len = (int)(v.x & 0xFF);
    uint8_t* p = out + offsets[i];
    for (int k = 0; k < len && k < 15; ++k) p[k] = (uint8_t)(k < 8 ? (v.y >> (56 - 8 * k)) : (v.x >> (56 - 8 * (k - 8))));
  }
}
#endif

// ------------------------------------------------------------------ take of Arrow-layout Utf8 (strings of any length as payload)
// lens[j] = byte length of source row idx[j] (0 for NULL_ROW / NULL values); validity word per 64 outputs
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_take_utf8_lengths(const int32_t* __restrict__ offsets, const uint8_t* __restrict__ validity, const uint32_t* __restrict__ idx,
                                                             const i64 n, int32_t* __restrict__ lens, u64* __restrict__ valid_out) {
  const i64 nwords = (n + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * WAVES + wave_id(); w < nwords; w += (i64)gridDim.x * WAVES) {
    const i64 j = (w << 6) + lane_id();
    bool ok = false; int32_t len = 0;
    if (j < n) {
      const uint32_t r = idx ? idx[j] : (uint32_t)j;
      ok = r != 0xFFFFFFFFu && (!validity || ((validity[r >> 3] >> (r & 7)) & 1));
      if (ok) len = offsets[r + 1] - offsets[r];
      lens[j] = len;
    }
    const u64 m = __ballot(ok);
    if (valid_out && lane_id() == 0) valid_out[w] = m;
  }
}
// one wave per output string: lanes copy 64 bytes per step (long strings stay coalesced, short ones cost one step)
__global__ void __launch_bounds__(BLOCK) k_take_utf8_bytes(const uint8_t* __restrict__ data, const int32_t* __restrict__ offsets, const uint32_t* __restrict__ idx,
                                                           const i64 n, const int32_t* __restrict__ out_offsets, uint8_t* __restrict__ out) {
  for (i64 j = (i64)blockIdx.x * WAVES + wave_id(); j < n; j += (i64)gridDim.x * WAVES) {
    const int32_t o0 = out_offsets[j], len = out_offsets[j + 1] - o0;
    if (len <= 0) continue;
    const uint32_t r = idx ? idx[j] : (uint32_t)j;
    const uint8_t* src = data + offsets[r];
    for (int32_t k = lane_id(); k < len; k += 64) out[o0 + k] = src[k];
  }
}
// short strings (names, flags: a few bytes each): one lane per string -- a whole wave per 8-byte string leaves 56 lanes idle
__global__ void __launch_bounds__(BLOCK) k_take_utf8_bytes_short(const uint8_t* __restrict__ data, const int32_t* __restrict__ offsets, const uint32_t* __restrict__ idx,
                                                                 const i64 n, const int32_t* __restrict__ out_offsets, uint8_t* __restrict__ out) {
  for (i64 j = (i64)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (i64)gridDim.x * BLOCK) {
    const int32_t o0 = out_offsets[j], len = out_offsets[j + 1] - o0;
    if (len <= 0) continue;
    const uint32_t r = idx ? idx[j] : (uint32_t)j;
    const uint8_t* src = data + offsets[r];
    for (int32_t k = 0; k < len; ++k) out[o0 + k] = src[k];
  }
}
#endif

// Exclusive scan of int32 lengths into offsets (n+1 entries, in place): three-kernel
// reduce / scan-of-tiles / downsweep over tiles of 2048 elements.
constexpr int SCAN_TILE = 2048;
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_scan_tile_sums(const int32_t* __restrict__ d, const i64 n, u64* __restrict__ tile_sums) {
  __shared__ u64 ws[WAVES];
  const i64 base = (i64)blockIdx.x * SCAN_TILE;
  u64 s = 0;
  for (int k = threadIdx.x; k < SCAN_TILE; k += BLOCK) if (base + k < n) s += (u64)(uint32_t)d[base + k];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane_id() == 0) ws[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) { u64 t = 0; for (int k = 0; k < WAVES; ++k) t += ws[k]; tile_sums[blockIdx.x] = t; }
}
#endif
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(1024) k_scan_tiles_serial(u64* __restrict__ tile_sums, const i64 ntiles) {
  // one block; thread-strided sequential chunks then a serial fix-up over <=1024 partials
  __shared__ u64 part[1024];
  const int t = threadIdx.x;
  const i64 per = (ntiles + 1023) / 1024;
  const i64 a = (i64)t * per; i64 b = a + per; if (b > ntiles) b = ntiles;
  u64 s = 0; for (i64 i = a; i < b; ++i) s += tile_sums[i];
  part[t] = s; __syncthreads();
  if (t == 0) { const int lim = (int)((ntiles + per - 1) / per); u64 run = 0; for (int k = 0; k < lim; ++k) { const u64 v = part[k]; part[k] = run; run += v; } }   // chunks past lim are empty
  __syncthreads();
  u64 run = part[t];
  for (i64 i = a; i < b; ++i) { const u64 v = tile_sums[i]; tile_sums[i] = run; run += v; }
}
#endif
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(BLOCK) k_scan_downsweep(int32_t* __restrict__ d, const i64 n, const u64* __restrict__ tile_offsets) {
  __shared__ u64 ws[WAVES];
  const i64 base = (i64)blockIdx.x * SCAN_TILE;
  constexpr int PER = SCAN_TILE / BLOCK;   // consecutive items per thread
  int32_t v[PER]; u64 local = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { const i64 i = base + (i64)threadIdx.x * PER + k; v[k] = (i < n) ? d[i] : 0; local += (u64)(uint32_t)v[k]; }
  u64 x = local;
  for (int off = 1; off < 64; off <<= 1) { const u64 y = __shfl_up(x, off); if (lane_id() >= off) x += y; }
  if (lane_id() == 63) ws[wave_id()] = x;
  __syncthreads();
  u64 wbase = 0; for (int k = 0; k < wave_id(); ++k) wbase += ws[k];
  u64 run = tile_offsets[blockIdx.x] + wbase + x - local;
#pragma unroll
  for (int k = 0; k < PER; ++k) { const i64 i = base + (i64)threadIdx.x * PER + k; if (i <= n) { if (i < n) d[i] = (int32_t)run; else d[i] = (int32_t)run; } run += (u64)(uint32_t)v[k]; }
}
#endif

#ifndef GPUQ_JIT
// ------------------------------------------------------------------ launchers
static int g_num_cus = 256;
JitOverride& jit_override() { static thread_local JitOverride o; return o; }
void set_num_cus(int n) { if (n > 0) g_num_cus = n; }
int num_cus() { return g_num_cus; }

static int grid_for(i64 n, int blocks_per_cu) {
  const i64 nwords = (n + 63) >> 6;
  i64 need = (nwords + WAVES - 1) / WAVES;
  i64 cap = (i64)g_num_cus * blocks_per_cu;
  if (need < 1) need = 1;
  return (int)(need < cap ? need : cap);
}

void launch_filter_bitmap(hipStream_t s, const DevProgram& P, i64 n, u64* bitmap, uint32_t* block_counts, int nblocks, i64 wpb) {
  if (jit_override().fn && jit_override().kernel_id == 1) {
    (void)jit_launch(jit_override().fn, dim3(nblocks), dim3(BLOCK), 0, s, P, n, bitmap, block_counts, wpb);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_filter_bitmap<M>, dim3(nblocks), dim3(BLOCK), 0, s, P, n, bitmap, block_counts, wpb)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
void launch_scan_block_counts(hipStream_t s, uint32_t* block_counts, int nblocks, u64* total_out) {
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, block_counts, nblocks, total_out);
}
void launch_compact(hipStream_t s, const u64* bitmap, const uint32_t* block_offsets, int nblocks, i64 wpb, i64 n,
                    const uint32_t* sel_in, uint32_t* sel_out) {
  hipLaunchKernelGGL(k_compact, dim3(nblocks), dim3(BLOCK), 0, s, bitmap, block_offsets, wpb, n, sel_in, sel_out);
}
void launch_project(hipStream_t s, const DevProgram& P, i64 n, const OutSpec& O) {
  if (n <= 0) return;
  if (jit_override().fn && jit_override().kernel_id == 2) {
    (void)jit_launch(jit_override().fn, dim3(grid_for(n, 8)), dim3(BLOCK), 0, s, P, n, O);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_project<M>, dim3(grid_for(n, 8)), dim3(BLOCK), 0, s, P, n, O)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}

// LDS budget: keep one block within 64 KiB so at least two blocks (8 waves) share a CU.
int agg_tiny_max_groups(int n_accs) {
  int best = 0;
  for (int g = 1; g <= 64; ++g) if (tiny_lds_bytes(g, MAX_KEYS, n_accs) <= 150 * 1024) best = g;
  return best;
}
static int tiny_blocks_per_cu(int gmax, int n_keys, int n_accs) {
  size_t b = tiny_lds_bytes(gmax, n_keys, n_accs);
  int k = (int)((160 * 1024) / b);
  if (k < 1) k = 1;
  if (k > 4) k = 4;
  return k;
}
size_t agg_tiny_workspace_bytes(int gmax, int n_keys, int n_accs, int* nblocks_out) {
  const int nb = g_num_cus * tiny_blocks_per_cu(gmax, n_keys, n_accs);
  if (nblocks_out) *nblocks_out = nb;
  return (size_t)nb * ((tiny_partial_bytes(gmax, n_keys, n_accs) + 15) & ~(size_t)15);
}
void launch_agg_tiny(hipStream_t s, const DevProgram& P, i64 n, const AggSpec& A, int gmax, void* workspace) {
  const size_t stride = (tiny_partial_bytes(gmax, A.n_keys, A.n_accs) + 15) & ~(size_t)15;
  int nb_cap = g_num_cus * tiny_blocks_per_cu(gmax, A.n_keys, A.n_accs);
  int nb = grid_for(n, 64);
  if (nb > nb_cap) nb = nb_cap;
  size_t lds = tiny_lds_bytes(gmax, A.n_keys, A.n_accs);
  if (const char* e = std::getenv("GPUQ_EXP_OCC")) {      // tuning experiment (with GPUQ_JIT_DEFINES=GPUQ_EXP_LOADS_ONLY=1 only): k blocks per CU
    const int k = std::atoi(e); if (k > 0) { nb = g_num_cus * k; const size_t cap_ = (size_t)(160 * 1024) / (size_t)k - 1024; if (lds > cap_) lds = cap_; }
  }
  // > 64 KiB of dynamic LDS needs an explicit opt-in (per template instantiation); the JIT'd function is
  // only used while the request stays within the default 64 KiB
  if (jit_override().fn && jit_override().kernel_id == 3 && lds <= 60 * 1024) {
    (void)jit_launch(jit_override().fn, dim3(nb), dim3(BLOCK), lds, s, P, n, A, gmax, (char*)workspace, stride);
  } else {
#define CALL(M)                                                                                                                          \
  do {                                                                                                                                   \
    static size_t attr_main = 0;                                                                                                         \
    if (lds > 60 * 1024 && lds > attr_main) {                                                                                            \
      if (hipFuncSetAttribute((const void*)k_agg_tiny<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) attr_main = lds; \
      else (void)hipGetLastError();                                                                                                      \
    }                                                                                                                                    \
    hipLaunchKernelGGL(k_agg_tiny<M>, dim3(nb), dim3(BLOCK), lds, s, P, n, A, gmax, (char*)workspace, stride);                           \
  } while (0)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}

void launch_agg_tiny_merge(hipStream_t s, const DevProgram& P, i64 n, const AggSpec& A, int gmax, void* workspace, const AggOut& out) {
  const size_t stride = (tiny_partial_bytes(gmax, A.n_keys, A.n_accs) + 15) & ~(size_t)15;
  int nb_cap = g_num_cus * tiny_blocks_per_cu(gmax, A.n_keys, A.n_accs);
  int nb = grid_for(n, 64);
  if (nb > nb_cap) nb = nb_cap;
  static size_t attr_merge = 0;
  const int kstride = A.n_keys > 0 ? A.n_keys : 1;
  const size_t mlds = (size_t)out.cap * kstride * 16 + (size_t)out.cap * 4 + (((size_t)nb * gmax * 2 + 15) & ~(size_t)15) + 32 + (size_t)out.cap * A.n_accs * 16;
  if (mlds > 60 * 1024 && mlds > attr_merge) {
    if (hipFuncSetAttribute((const void*)k_agg_tiny_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds) == hipSuccess) attr_merge = mlds;
    else (void)hipGetLastError();
  }
  hipLaunchKernelGGL(k_agg_tiny_merge, dim3(1), dim3(MERGE_BLOCK), mlds, s, A, gmax, (const char*)workspace, stride, nb, out, P.flags);
}

void launch_agg_emit(hipStream_t s, const AggOut& raw, int n_keys, int n_accs, uint32_t n_groups, const AggSoA& soa, const uint32_t* n_groups_dev) {
  if (n_groups == 0) return;
  const uint32_t nwords = (n_groups + 63) >> 6;
  uint32_t grid = (nwords + WAVES - 1) / WAVES; if (grid > (uint32_t)g_num_cus * 8) grid = g_num_cus * 8;
  hipLaunchKernelGGL(k_agg_emit, dim3(grid), dim3(BLOCK), 0, s, raw, n_keys, n_accs, n_groups, soa, n_groups_dev);
}
static int lin_grid(i64 n) { i64 need = (n + BLOCK - 1) / BLOCK; if (need < 1) need = 1; const i64 cap = (i64)g_num_cus * 16; return (int)(need < cap ? need : cap); }
void launch_unpack_utf8_lengths(hipStream_t s, const ulonglong2* packed, i64 n, int32_t* lens_out, uint32_t* too_long) {
  if (n > 0) hipLaunchKernelGGL(k_unpack_lengths, dim3(lin_grid(n)), dim3(BLOCK), 0, s, packed, n, lens_out, too_long);
}
__global__ void __launch_bounds__(BLOCK) k_offsets_rebase(const int32_t* __restrict__ src, const i64 n, const int32_t delta, int32_t* __restrict__ dst) {
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) dst[i] = src[i] + delta;
}
void launch_offsets_rebase(hipStream_t s, const int32_t* src, i64 n, int32_t delta, int32_t* dst) {
  if (n > 0) hipLaunchKernelGGL(k_offsets_rebase, dim3(lin_grid(n)), dim3(BLOCK), 0, s, src, n, delta, dst);
}
void launch_take_utf8_lengths(hipStream_t s, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n, int32_t* lens, u64* valid_out) {
  if (n > 0) hipLaunchKernelGGL(k_take_utf8_lengths, dim3(grid_for(n, 8)), dim3(BLOCK), 0, s, offsets, validity, idx, n, lens, valid_out);
}
void launch_take_utf8_bytes(hipStream_t s, const uint8_t* data, const int32_t* offsets, const uint32_t* idx, i64 n, const int32_t* out_offsets, uint8_t* out, i64 total_bytes) {
  if (n <= 0) return;
  if (total_bytes <= n * 24) {       // average string of <= 24 bytes
    hipLaunchKernelGGL(k_take_utf8_bytes_short, dim3(lin_grid(n)), dim3(BLOCK), 0, s, data, offsets, idx, n, out_offsets, out);
    return;
  }
  i64 need = (n + WAVES - 1) / WAVES; const i64 cap = (i64)g_num_cus * 16; if (need < 1) need = 1;
  hipLaunchKernelGGL(k_take_utf8_bytes, dim3((int)(need < cap ? need : cap)), dim3(BLOCK), 0, s, data, offsets, idx, n, out_offsets, out);
}
void launch_unpack_utf8_bytes(hipStream_t s, const ulonglong2* packed, i64 n, const int32_t* offsets, uint8_t* data_out) {
  if (n > 0) hipLaunchKernelGGL(k_unpack_bytes, dim3(lin_grid(n)), dim3(BLOCK), 0, s, packed, n, offsets, data_out);
}
// dst bits [off, off+n) <- src bits [soff, soff+n) (src == nullptr: all ones).  dst is zero-initialised by the caller; words
// shared between neighbouring pieces are merged with atomicOr.  One thread per destination word.
__global__ void __launch_bounds__(BLOCK) k_concat_bitmap(u64* dst, const i64 off, const uint8_t* __restrict__ src, const i64 soff, const i64 n) {
  const i64 w0 = off >> 6, w1 = (off + n - 1) >> 6;
  for (i64 w = w0 + (i64)blockIdx.x * BLOCK + threadIdx.x; w <= w1; w += (i64)gridDim.x * BLOCK) {
    const i64 s0 = w * 64 - off;           // piece-relative bit that lands on bit 0 of this word (may be negative / beyond n)
    u64 v = 0;
    if (!src) {
      const i64 lo = s0 < 0 ? -s0 : 0, hi = (n - s0) < 64 ? (n - s0) : 64;
      v = (hi - lo >= 64) ? ~0ull : (((1ull << (hi - lo)) - 1) << lo);
    } else {
      for (int b = 0; b < 64; ++b) {
        const i64 sb = s0 + b;
        if (sb >= 0 && sb < n) { const i64 q = sb + soff; v |= (u64)((src[q >> 3] >> (q & 7)) & 1) << b; }
      }
    }
    if (v) atomicOr((unsigned long long*)(dst + w), (unsigned long long)v);
  }
}
void launch_concat_bitmap(hipStream_t s, u64* dst, i64 off, const uint8_t* src, i64 src_off, i64 n) {
  if (n > 0) hipLaunchKernelGGL(k_concat_bitmap, dim3(lin_grid((n + 127) / 64)), dim3(BLOCK), 0, s, dst, off, src, src_off, n);
}
// SQL LIKE of an Arrow-layout Utf8 column against a literal pattern (DataFusion LikeExpr -> arrow-string 49 `like` / `nlike`
// with a scalar pattern).  One lane per string; a wave's ballot is one word of the result bitmap.  Pattern tokens: 0..255 a
// literal byte, 256 = '_' (one UTF-8 character), 257 = '%' (any run).  Greedy matching with one backtrack point (the last '%').
// regex_mode: the pattern is none of arrow's fast shapes (equality, prefix%, %suffix, %infix%) and goes through its regex
// translation, where '.' does not match a newline: '_' and '%' then refuse '\n' [UPSTREAM-KNOWLEDGE, arrow-string 49 like.rs].
__device__ __forceinline__ int utf8_len(uint8_t b) { return b < 0x80 ? 1 : (b < 0xE0 ? 2 : (b < 0xF0 ? 3 : 4)); }
// The match itself: `at(i)` = byte i of the string.
template <class At>
__device__ __forceinline__ bool like_match(const int len, const int plen, const uint16_t* tok, const bool regex_mode, At at) {
  int s = 0, p = 0, star_p = -1, star_s = 0;
  bool fail = false;
  while (s < len) {
    const int t = p < plen ? (int)tok[p] : -1;
    const uint8_t ch = at(s);
    if (t >= 0 && t < 256 && ch == (uint8_t)t) { ++s; ++p; }
    else if (t == 256 && !(regex_mode && ch == '\n')) { s += utf8_len(ch); ++p; }
    else if (t == 257) { star_p = p; star_s = s; ++p; }
    else if (star_p >= 0) {
      const uint8_t sc = at(star_s);
      if (regex_mode && sc == '\n') { fail = true; break; }
      star_s += utf8_len(sc); s = star_s; p = star_p + 1;
    } else { fail = true; break; }
  }
  if (!fail && s > len) fail = true;                 // a truncated multi-byte character
  while (!fail && p < plen && tok[p] == 257) ++p;
  return !fail && p == plen;
}
// Patterns made of literal segments and '%' only (no '_'): prefix% / %suffix / %infix% / a%b%c ... -- the segments are searched for in
// order, leftmost first (which is what greedy '%' matching comes to), through a 16-byte funnel over the string: one masked 8-byte
// compare per position instead of the general matcher's state machine per byte (lanes of a wave hold strings in different states: the
// general loop runs all its branches every step; measured 4-7 % of the HBM peak on 50-byte comments, this path: see profiles/).
// seg k = tok[seg_off[k] .. +seg_len[k]) (literal bytes); first8 / mask8 = its first <= 8 bytes as a little-endian word.
__device__ __forceinline__ u64 like_load8(const uint8_t* __restrict__ str, const int q, const int len) {      // bytes [q, q+8) of the string, zero beyond its end
  if (q + 8 <= len) { u64 w; __builtin_memcpy(&w, str + q, 8); return w; }
  u64 w = 0; for (int k = q; k < len; ++k) w |= (u64)str[k] << (8 * (k - q));
  return w;
}
__device__ __forceinline__ bool like_seg_at(const uint8_t* __restrict__ str, const int len, const int q, const uint16_t* seg, const int L, const u64 first8, const u64 mask8) {
  if (q < 0 || q + L > len) return false;
  if ((like_load8(str, q, len) & mask8) != first8) return false;
  for (int k = 8; k < L; ++k) if (str[q + k] != (uint8_t)seg[k]) return false;
  return true;
}
__device__ __forceinline__ bool like_segments(const uint8_t* __restrict__ str, const int len, const LikePattern& pat, const uint16_t* tok) {
  if (pat.regex_mode) {      // '.' refuses a newline and the pattern holds none: eight bytes at a time (zero-byte test on x ^ 0x0A..)
    for (int k = 0; k < len; k += 8) {
      const u64 x = like_load8(str, k, len) ^ 0x0A0A0A0A0A0A0A0Aull;      // (the zero padding beyond the end becomes 0x0A ^ 0 = 0x0A: never zero)
      if ((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) return false;
    }
  }
  int pos = 0;
  const int ns = pat.n_seg;
  for (int k = 0; k < ns; ++k) {
    const int L = pat.seg_len[k]; const uint16_t* seg = tok + pat.seg_off[k];
    const u64 f8 = pat.seg_first8[k], m8 = pat.seg_mask8[k];
    const bool first = k == 0, last = k + 1 == ns;
    if (first && pat.anchored_start) { if (!like_seg_at(str, len, 0, seg, L, f8, m8)) return false; pos = L; if (last && pat.anchored_end && len != L) return false; continue; }
    if (last && pat.anchored_end) return len - L >= pos && like_seg_at(str, len, len - L, seg, L, f8, m8);
    // leftmost occurrence at or after pos.  Per aligned 8-byte block: the positions that hold the segment's FIRST byte (zero-byte test on
    // block ^ splat(byte): exact for the lowest hit, later hits may be false positives, every candidate is verified), then one masked
    // 8-byte compare per candidate through a 16-byte funnel (cur = bytes [b, b+8), nxt = [b+8, b+16))
    int q = pos; bool found = false;
    const int lastq = len - L;
    const u64 splat = (f8 & 0xFFull) * 0x0101010101010101ull;
    while (q <= lastq && !found) {
      const int b = q & ~7;
      const u64 cur = like_load8(str, b, len);
      const u64 x = cur ^ splat;
      u64 cand = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
      if (q > b) cand &= ~0ull << (8 * (q - b));                     // positions before q are behind us
      if (cand) {
        const u64 nxt = like_load8(str, b + 8, len);
        while (cand) {
          const int k = __builtin_ctzll(cand) >> 3; cand &= cand - 1;
          const int at = b + k;
          if (at > lastq) break;
          const int sh = 8 * k;
          const u64 w = sh ? (cur >> sh) | (nxt << (64 - sh)) : cur;
          if ((w & m8) == f8) {
            bool ok = true;
            for (int j = 8; j < L; ++j) if (str[at + j] != (uint8_t)seg[j]) { ok = false; break; }
            if (ok) { found = true; q = at; break; }
          }
        }
      }
      if (!found) q = b + 8;
    }
    if (!found) return false;
    pos = q + L;
  }
  return ns > 0 ? true : (pat.anchored_start && pat.anchored_end ? len == 0 : true);      // no segment at all: '' matches only '', '%' everything
}
__global__ void __launch_bounds__(BLOCK) k_like_utf8(const uint8_t* __restrict__ data, const int32_t* __restrict__ offsets, const uint8_t* __restrict__ validity,
                                                     const uint32_t* __restrict__ idx, const i64 n, const LikePattern pat, const int negated,
                                                     u64* __restrict__ bits_out, u64* __restrict__ valid_out) {
  __shared__ uint16_t tok[LIKE_MAX_TOKENS];
  for (int i = (int)threadIdx.x; i < pat.n; i += BLOCK) tok[i] = pat.tok[i];
  __syncthreads();
  const i64 words = (n + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); w < words; w += (i64)gridDim.x * (BLOCK / 64)) {
    const i64 i = w * 64 + (threadIdx.x & 63);
    bool valid = false, m = false;
    if (i < n) {
      const uint32_t r = idx ? idx[i] : (uint32_t)i;
      valid = r != 0xFFFFFFFFu && (!validity || ((validity[r >> 3] >> (r & 7)) & 1));
      if (valid) {
        const uint8_t* str = data + offsets[r];
        const int len = offsets[r + 1] - offsets[r], plen = pat.n;
        if (pat.n_seg >= 0) m = like_segments(str, len, pat, tok);
        else {
          // the string is walked through an 8-byte register window (one unaligned 8-byte load per 8 bytes instead of a byte load
          // per step); the last < 8 bytes of a string are read byte-wise so that nothing beyond its end is touched
          int wbase = -8; u64 win = 0;
          m = like_match(len, plen, tok, pat.regex_mode != 0, [&](int q) -> uint8_t {
            if ((unsigned)(q - wbase) >= 8u) {
              wbase = q & ~7;
              if (wbase + 8 <= len) __builtin_memcpy(&win, str + wbase, 8);
              else { win = 0; for (int k = wbase; k < len; ++k) win |= (u64)str[k] << (8 * (k - wbase)); }
            }
            return (uint8_t)(win >> (8 * (q - wbase)));
          });
        }
        if (negated) m = !m;
      }
    }
    const u64 mb = __ballot(valid && m), vb = __ballot(valid);
    if ((threadIdx.x & 63) == 0) { bits_out[w] = mb; if (valid_out) valid_out[w] = vb; }
  }
}
void launch_like_utf8(hipStream_t s, const uint8_t* data, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n, const LikePattern& pat, int negated,
                      u64* bits_out, u64* valid_out) {
  if (n > 0) hipLaunchKernelGGL(k_like_utf8, dim3(lin_grid(n)), dim3(BLOCK), 0, s, data, offsets, validity, idx, n, pat, negated, bits_out, valid_out);
}
// ------------------------------------------------------------------ Utf8 comparisons at any length
// The register programs compare strings of up to 15 bytes; an ordering comparison (or column = column) over longer values is run here,
// over the Arrow-layout bytes: op 0 = , 1 != , 2 < , 3 <= , 4 > , 5 >=  by bytes (memcmp, then the shorter string first: arrow's order).
// b is another column (boffs != nullptr, read through bidx) or ONE literal (bdata = its bytes, blen its length).  Either side NULL -> NULL.
struct Utf8Side { const uint8_t* data; const int32_t* offsets; const uint8_t* validity; const uint32_t* idx; };
__global__ void __launch_bounds__(BLOCK) k_utf8_compare(const Utf8Side A, const Utf8Side B, const int32_t blen, const i64 n, const int op, u64* __restrict__ bits_out, u64* __restrict__ valid_out) {
  const i64 words = (n + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); w < words; w += (i64)gridDim.x * (BLOCK / 64)) {
    const i64 i = w * 64 + (threadIdx.x & 63);
    bool valid = false, m = false;
    if (i < n) {
      const uint32_t ra = A.idx ? A.idx[i] : (uint32_t)i;
      valid = ra != 0xFFFFFFFFu && (!A.validity || ((A.validity[ra >> 3] >> (ra & 7)) & 1));
      const uint8_t* pb = B.data; int32_t lb = blen;
      if (B.offsets) {
        const uint32_t rb = B.idx ? B.idx[i] : (uint32_t)i;
        valid = valid && rb != 0xFFFFFFFFu && (!B.validity || ((B.validity[rb >> 3] >> (rb & 7)) & 1));
        if (valid) { pb = B.data + B.offsets[rb]; lb = B.offsets[rb + 1] - B.offsets[rb]; }
      }
      if (valid) {
        const uint8_t* pa = A.data + A.offsets[ra]; const int32_t la = A.offsets[ra + 1] - A.offsets[ra];
        const int32_t k = la < lb ? la : lb;
        int c = 0; int32_t j = 0;
        for (; j + 8 <= k; j += 8) {      // eight bytes at a time, big-endian so that the integer order is the byte order
          u64 x, y; __builtin_memcpy(&x, pa + j, 8); __builtin_memcpy(&y, pb + j, 8);
          if (x != y) { x = __builtin_bswap64(x); y = __builtin_bswap64(y); c = x < y ? -1 : 1; break; }
        }
        if (c == 0) for (; j < k; ++j) if (pa[j] != pb[j]) { c = pa[j] < pb[j] ? -1 : 1; break; }
        if (c == 0) c = la < lb ? -1 : (la > lb ? 1 : 0);
        m = op == 0 ? c == 0 : op == 1 ? c != 0 : op == 2 ? c < 0 : op == 3 ? c <= 0 : op == 4 ? c > 0 : c >= 0;
      }
    }
    const u64 mb = __ballot(valid && m), vb = __ballot(valid);
    if ((threadIdx.x & 63) == 0) { bits_out[w] = mb; if (valid_out) valid_out[w] = vb; }
  }
}
void launch_utf8_compare(hipStream_t s, const uint8_t* adata, const int32_t* aoffs, const uint8_t* avalid, const uint32_t* aidx, const uint8_t* bdata, const int32_t* boffs,
                         const uint8_t* bvalid, const uint32_t* bidx, int32_t blen, i64 n, int op, u64* bits_out, u64* valid_out) {
  if (n > 0) hipLaunchKernelGGL(k_utf8_compare, dim3(lin_grid(n)), dim3(BLOCK), 0, s, Utf8Side{adata, aoffs, avalid, aidx}, Utf8Side{bdata, boffs, bvalid, bidx}, blen, n, op, bits_out, valid_out);
}
// ------------------------------------------------------------------ Utf8 values as exact dictionary codes (f-4: keys longer than 15 bytes)
// Strings of any length become group / join keys through a device dictionary: an open-addressing table of 8-byte entries
// (hash tag << 32 | representative row of the dictionary column).  One 64-bit CAS both claims an entry and publishes its row, the
// bytes it stands for live in the (immutable) dictionary column, so there is nothing to lock or fence.  An entry whose tag matches is
// verified byte for byte; a different string with the same tag just moves on along the probe sequence: codes are EXACT.
// code = the representative's row in the dictionary column (equal strings -> equal codes; the string comes back with a take).
__device__ __forceinline__ u64 utf8_hash(const uint8_t* __restrict__ p, int32_t len) {
  u64 h = 0xCBF29CE484222325ull ^ (u64)(uint32_t)len;
  int32_t i = 0;
  for (; i + 8 <= len; i += 8) { u64 w; __builtin_memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001B3ull; h ^= h >> 29; }
  u64 tail = 0; for (int k = 0; i + k < len; ++k) tail |= (u64)p[i + k] << (8 * k);
  h = (h ^ tail) * 0x100000001B3ull;
  h ^= h >> 32; h *= 0x9FB21C651E98DF25ull; h ^= h >> 29;
  return h;
}
__global__ void __launch_bounds__(BLOCK) k_utf8_max_len(const int32_t* __restrict__ offsets, const uint8_t* __restrict__ validity, const uint32_t* __restrict__ idx, const i64 n,
                                                        int32_t* __restrict__ out) {
  int32_t m = 0;
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const uint32_t r = idx ? idx[i] : (uint32_t)i;
    if (r == 0xFFFFFFFFu) continue;
    if (validity && !((validity[r >> 3] >> (r & 7)) & 1)) continue;
    const int32_t len = offsets[r + 1] - offsets[r];
    m = len > m ? len : m;
  }
  for (int o = 32; o > 0; o >>= 1) { const int32_t y = __shfl_xor(m, o); m = y > m ? y : m; }
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}
// codes[i] = dictionary code of row i (through idx), valid bit i = the row has a code (not NULL; for lookups: the string is in the dictionary)
__global__ void __launch_bounds__(BLOCK) k_utf8_intern(const uint8_t* __restrict__ ddata, const int32_t* __restrict__ doffs,
                                                       const uint8_t* __restrict__ data, const int32_t* __restrict__ offsets, const uint8_t* __restrict__ validity,
                                                       const uint32_t* __restrict__ idx, const i64 n, u64* __restrict__ table, const u64 mask, const int insert,
                                                       i64* __restrict__ codes, u64* __restrict__ valid_out, uint32_t* __restrict__ flags) {
  for (i64 i0 = (i64)blockIdx.x * BLOCK + (threadIdx.x & ~63); i0 < n; i0 += (i64)gridDim.x * BLOCK) {
    const i64 i = i0 + (threadIdx.x & 63);
    bool has = false; i64 code = 0;
    if (i < n) {
      const uint32_t r = idx ? idx[i] : (uint32_t)i;
      const bool isnull = r == 0xFFFFFFFFu || (validity && !((validity[r >> 3] >> (r & 7)) & 1));
      if (!isnull) {
        const int32_t o = offsets[r], len = offsets[r + 1] - o;
        const uint8_t* p = data + o;
        const u64 h = utf8_hash(p, len);
        const u64 tag = (h >> 32) | 1ull;
        u64 slot = h & mask;
        for (u64 probes = 0; probes <= mask; ++probes, slot = (slot + 1) & mask) {
          u64 e = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (e == 0) {
            if (!insert) break;                                     // not in the dictionary
            const u64 mine = (tag << 32) | (u64)r;
            u64 expected = 0;
            if (__hip_atomic_compare_exchange_strong(&table[slot], &expected, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { has = true; code = (i64)r; break; }
            e = expected;                                           // somebody else took the entry: it may be this very string
          }
          if ((e >> 32) != tag) continue;
          const uint32_t rep = (uint32_t)e;
          const int32_t ro = doffs[rep], rlen = doffs[rep + 1] - ro;
          bool same = rlen == len;
          if (same) { const uint8_t* q = ddata + ro; for (int32_t k = 0; k < len; ++k) if (p[k] != q[k]) { same = false; break; } }
          if (same) { has = true; code = (i64)rep; break; }
        }
        if (insert && !has) atomicOr(flags, 1u);                    // table full (cannot happen: it is sized for every row)
      }
      codes[i] = code;
    }
    const u64 m = __ballot(has);
    if ((threadIdx.x & 63) == 0 && valid_out) valid_out[i0 >> 6] = m;
  }
}
// codes (Int64 + validity) -> row ids for a take: NULL -> 0xFFFFFFFF
__global__ void __launch_bounds__(BLOCK) k_utf8_code_rows(const void* __restrict__ codes, const int width, const uint8_t* __restrict__ validity, const i64 n, uint32_t* __restrict__ rows) {
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK)
    rows[i] = (validity && !((validity[i >> 3] >> (i & 7)) & 1)) ? 0xFFFFFFFFu : (width == 8 ? (uint32_t)((const i64*)codes)[i] : ((const uint32_t*)codes)[i]);
}
void launch_utf8_code_rows(hipStream_t s, const void* codes, int width, const uint8_t* validity, i64 n, uint32_t* rows) {
  if (n > 0) hipLaunchKernelGGL(k_utf8_code_rows, dim3(lin_grid(n)), dim3(BLOCK), 0, s, codes, width, validity, n, rows);
}
// Order-preserving fixed-width pieces of a Utf8 column (sort keys of any length): piece j of a string = bytes [14 j, 14 j + 14) as a
// big-endian integer, zero padded, times 256 plus the number of bytes the piece holds (0..14).  Comparing the pieces of two
// strings in order is comparing the strings bytewise (the count separates "ab" from "ab\0").  out: 16-byte integers (< 2^120);
// valid_out bit i = row i is not NULL (same for every piece).
__global__ void __launch_bounds__(BLOCK) k_utf8_sort_piece(const uint8_t* __restrict__ data, const int32_t* __restrict__ offsets, const uint8_t* __restrict__ validity,
                                                           const uint32_t* __restrict__ idx, const i64 n, const int piece, ulonglong2* __restrict__ out, u64* __restrict__ valid_out) {
  for (i64 i0 = (i64)blockIdx.x * BLOCK + (threadIdx.x & ~63); i0 < n; i0 += (i64)gridDim.x * BLOCK) {
    const i64 i = i0 + (threadIdx.x & 63);
    bool has = false; u64 lo = 0, hi = 0;
    if (i < n) {
      const uint32_t r = idx ? idx[i] : (uint32_t)i;
      has = !(r == 0xFFFFFFFFu || (validity && !((validity[r >> 3] >> (r & 7)) & 1)));
      if (has) {
        const int32_t o = offsets[r], len = offsets[r + 1] - o;
        int32_t m = len - 14 * piece; m = m < 0 ? 0 : (m > 14 ? 14 : m);
        const uint8_t* p = data + o + 14 * piece;
        // value bits: byte 0 at bits 119..112, ..., byte 13 at bits 15..8, count at bits 7..0
        for (int b = 0; b < m; ++b) {
          const u64 v = p[b]; const int sh = 112 - 8 * b;
          if (sh >= 64) hi |= v << (sh - 64); else lo |= v << sh;
        }
        lo |= (u64)m;
      }
    }
    const u64 vm = __ballot(has);
    if (valid_out && (threadIdx.x & 63) == 0) valid_out[i0 >> 6] = vm;
    if (i < n) out[i] = make_ulonglong2(lo, hi);
  }
}
void launch_utf8_sort_piece(hipStream_t s, const uint8_t* data, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n, int piece, void* out, u64* valid_out) {
  if (n > 0) hipLaunchKernelGGL(k_utf8_sort_piece, dim3(lin_grid(n)), dim3(BLOCK), 0, s, data, offsets, validity, idx, n, piece, (ulonglong2*)out, valid_out);
}
void launch_utf8_max_len(hipStream_t s, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n, int32_t* out) {
  if (n > 0) hipLaunchKernelGGL(k_utf8_max_len, dim3(lin_grid(n)), dim3(BLOCK), 0, s, offsets, validity, idx, n, out);
}
void launch_utf8_intern(hipStream_t s, const uint8_t* ddata, const int32_t* doffs, const uint8_t* data, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n,
                        u64* table, u64 mask, int insert, i64* codes, u64* valid_out, uint32_t* flags) {
  if (n > 0) hipLaunchKernelGGL(k_utf8_intern, dim3(lin_grid(n)), dim3(BLOCK), 0, s, ddata, doffs, data, offsets, validity, idx, n, table, mask, insert, codes, valid_out, flags);
}
// bitmap[rows[i]] = 1 for every i (0xFFFFFFFF skipped): which rows of a join side survive in the filtered pair list
__global__ void __launch_bounds__(BLOCK) k_mark_rows(const uint32_t* __restrict__ rows, const i64 n, unsigned int* __restrict__ bitmap) {
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * BLOCK) {
    const uint32_t r = rows[i];
    if (r != 0xFFFFFFFFu) atomicOr(bitmap + (r >> 5), 1u << (r & 31));
  }
}
// CrossJoinExec: pair i of left x right, left-major
__global__ void __launch_bounds__(BLOCK) k_cross_pairs(const i64 total, const i64 n_right, uint32_t* __restrict__ lrows, uint32_t* __restrict__ rrows) {
  for (i64 i = (i64)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (i64)gridDim.x * BLOCK) { const i64 l = i / n_right; lrows[i] = (uint32_t)l; rrows[i] = (uint32_t)(i - l * n_right); }
}
void launch_cross_pairs(hipStream_t s, i64 n_left, i64 n_right, uint32_t* left_rows, uint32_t* right_rows) {
  const i64 total = n_left * n_right;
  if (total > 0) hipLaunchKernelGGL(k_cross_pairs, dim3(lin_grid(total)), dim3(BLOCK), 0, s, total, n_right, left_rows, right_rows);
}
void launch_mark_rows(hipStream_t s, const uint32_t* rows, i64 n, uint8_t* bitmap) {
  if (n > 0) hipLaunchKernelGGL(k_mark_rows, dim3(lin_grid(n)), dim3(BLOCK), 0, s, rows, n, (unsigned int*)bitmap);
}
size_t exclusive_scan_ws_bytes(i64 n) { return (size_t)((n + 1 + SCAN_TILE - 1) / SCAN_TILE + 1) * 8; }
void launch_exclusive_scan_i32(hipStream_t s, int32_t* data, i64 n, void* workspace, size_t) {
  // data holds n lengths (entry n is scratch); afterwards data[0..n] are the offsets
  const i64 ntiles = (n + 1 + SCAN_TILE - 1) / SCAN_TILE;
  u64* tiles = (u64*)workspace;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)ntiles), dim3(BLOCK), 0, s, (const int32_t*)data, n, tiles);
  hipLaunchKernelGGL(k_scan_tiles_serial, dim3(1), dim3(1024), 0, s, tiles, ntiles);
  hipLaunchKernelGGL(k_scan_downsweep, dim3((unsigned)ntiles), dim3(BLOCK), 0, s, data, n, (const u64*)tiles);
}

#endif  // GPUQ_JIT

}  // namespace gpuq
