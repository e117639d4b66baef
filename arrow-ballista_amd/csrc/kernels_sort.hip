// SortExec and hash repartition for gfx950.
//
// Replaces, on the reference's path: a8 SortExec (datafusion.proto:1465-1478,
// sort options :1247-1251) and a2 the BatchPartitioner::partition call site
// (ballista/core/src/execution_plans/shuffle_writer.rs:336-391).
//
// MI355X-first design: instead of comparing wide row-format keys, every sort key column is
// range-compressed on the device (value - min, or max - value for DESC, plus one null bit when
// needed) and the columns are bit-concatenated into one 64- or 128-bit composite key.  The sort is
// then a stable LSD radix sort (8-bit digits) over ceil(total_bits/8) passes of (u64 key, u32 row)
// pairs: TPC-H q3's (revenue DESC, o_orderdate) is 7 passes instead of the 20 of a byte-wise sort
// over the declared Decimal128+Date32 widths.  Hash repartition is one such pass with
// digit = partition id.  All passes stream HBM with coalesced reads; LDS holds the digit counters.
#include "gpuq_kernels.h"

namespace gpuq {

constexpr int SBLOCK = 256;
constexpr int SWAVES = SBLOCK / 64;
constexpr int RADIX = 256;

__device__ __forceinline__ int slane() { return threadIdx.x & 63; }
__device__ __forceinline__ int swave() { return threadIdx.x >> 6; }

// ordered 128-bit view of a key register: signed integers as is; f64 through the total-order map;
// packed strings with the top bit flipped so that signed compare == bytewise compare
__device__ __forceinline__ i128 sort_view(u64 lo, u64 hi, int kind) {
  if (kind == 1) { const i64 k = f64_total_key(lo); return (i128)k; }
  if (kind == 2) hi ^= 0x8000000000000000ull;
  return mk128(lo, hi);
}

// ------------------------------------------------------------------ min / max per key
// out per block: [block][key] {min_lo,min_hi,max_lo,max_hi,flags(bit0 any non-null, bit1 any null)}
struct MinMax { u64 mn_lo, mn_hi, mx_lo, mx_hi; uint32_t fl; };
__device__ __forceinline__ void mm_init(MinMax& m) { m.mn_lo = ~0ull; m.mn_hi = 0x7FFFFFFFFFFFFFFFull; m.mx_lo = 0; m.mx_hi = 0x8000000000000000ull; m.fl = 0; }
__device__ __forceinline__ void mm_update(MinMax& m, i128 v) {
  if (v < mk128(m.mn_lo, m.mn_hi)) { m.mn_lo = (u64)v; m.mn_hi = (u64)((u128)v >> 64); }
  if (v > mk128(m.mx_lo, m.mx_hi)) { m.mx_lo = (u64)v; m.mx_hi = (u64)((u128)v >> 64); }
}
__device__ __forceinline__ void mm_row(MinMax& m, const SortSpec& S, int k, GPUQ_REGS_CPARAM) {
  if (k < S.n_keys) {
    const int r = __builtin_amdgcn_readfirstlane(S.reg[k]);
    const bool isn = (rnulls >> r) & 1;
    if (isn) m.fl |= 2u;
    else {
      m.fl |= 1u; mm_update(m, sort_view(rlo[r], rhi[r], S.kind[k]));
      if (S.kind[k] == 2) { const uint32_t len = (uint32_t)(rlo[r] & 0xFF) << 8; if (len > (m.fl & 0xFF00u)) m.fl = (m.fl & ~0xFF00u) | len; }   // longest string, bits 8..15
    }
  }
}
__device__ __forceinline__ void mm_reduce_store(MinMax& m, int k, u64 (*red)[MAX_SORT_KEYS][5]) {
  for (int off = 32; off > 0; off >>= 1) {
    const i128 omn = mk128(__shfl_xor(m.mn_lo, off), __shfl_xor(m.mn_hi, off));
    const i128 omx = mk128(__shfl_xor(m.mx_lo, off), __shfl_xor(m.mx_hi, off));
    if (omn < mk128(m.mn_lo, m.mn_hi)) { m.mn_lo = (u64)omn; m.mn_hi = (u64)((u128)omn >> 64); }
    if (omx > mk128(m.mx_lo, m.mx_hi)) { m.mx_lo = (u64)omx; m.mx_hi = (u64)((u128)omx >> 64); }
    { const uint32_t of = __shfl_xor(m.fl, off); const uint32_t ml = (of & 0xFF00u) > (m.fl & 0xFF00u) ? (of & 0xFF00u) : (m.fl & 0xFF00u); m.fl = ((m.fl | of) & 0xFFu) | ml; }
  }
  if (slane() == 0) { red[swave()][k][0] = m.mn_lo; red[swave()][k][1] = m.mn_hi; red[swave()][k][2] = m.mx_lo; red[swave()][k][3] = m.mx_hi; red[swave()][k][4] = m.fl; }
}
template <int MAXC>
__device__ __forceinline__ void k_sort_minmax_body(const DevProgram P, const i64 n_arg, const SortSpec S, u64* __restrict__ out, const i64 wstep) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  __shared__ u64 red[SWAVES][MAX_SORT_KEYS][5];
  MinMax m0, m1, m2, m3;   // scalars on purpose: an indexed array here competes with the register file for promotion
  mm_init(m0); mm_init(m1); mm_init(m2); mm_init(m3);
  const i64 nw_all = (n + 63) >> 6;
  const i64 nwords = wstep > 1 ? (nw_all + wstep - 1) / wstep + 1 : nw_all;      // wstep > 1: a strided sample of 64-row words + the last word
  for (i64 w = (i64)blockIdx.x * SWAVES + swave(); w < nwords; w += (i64)gridDim.x * SWAVES) {
    const i64 wi = w * wstep < nw_all ? w * wstep : nw_all - 1;
    const i64 pos = (wi << 6) + slane();
    if (pos >= n) continue;
    GPUQ_REGS_DECL;
    (void)GPUQ_EVAL(MAXC, P, pos);
    mm_row(m0, S, 0, GPUQ_REGS); mm_row(m1, S, 1, GPUQ_REGS); mm_row(m2, S, 2, GPUQ_REGS); mm_row(m3, S, 3, GPUQ_REGS);
  }
  // a lane that saw no value keeps (max,min) sentinels, which never win a reduction
  mm_reduce_store(m0, 0, red); mm_reduce_store(m1, 1, red); mm_reduce_store(m2, 2, red); mm_reduce_store(m3, 3, red);
  __syncthreads();
  if (threadIdx.x < S.n_keys) {
    const int k = threadIdx.x;
    i128 a = mk128(red[0][k][0], red[0][k][1]), b = mk128(red[0][k][2], red[0][k][3]); u64 f = red[0][k][4];
    for (int q = 1; q < SWAVES; ++q) {
      const i128 oa = mk128(red[q][k][0], red[q][k][1]), ob = mk128(red[q][k][2], red[q][k][3]);
      if (oa < a) a = oa;
      if (ob > b) b = ob;
      { const u64 of = red[q][k][4]; const u64 ml = (of & 0xFF00u) > (f & 0xFF00u) ? (of & 0xFF00u) : (f & 0xFF00u); f = ((f | of) & 0xFFu) | ml; }
    }
    u64* o = out + ((size_t)blockIdx.x * MAX_SORT_KEYS + k) * 5;
    o[0] = (u64)a; o[1] = (u64)((u128)a >> 64); o[2] = (u64)b; o[3] = (u64)((u128)b >> 64); o[4] = f;
  }
}
#ifndef GPUQ_JIT
template <int MAXC>
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(SBLOCK) k_sort_minmax(const DevProgram P, const i64 n, const SortSpec S, u64* __restrict__ out, const i64 wstep) { k_sort_minmax_body<MAXC>(P, n, S, out, wstep); }
#endif
#elif GPUQ_JIT_KERNEL == 8
extern "C" __global__ void __launch_bounds__(SBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const SortSpec S, u64* __restrict__ out, const i64 wstep) { k_sort_minmax_body<0>(P, n, S, out, wstep); }
#endif

// ------------------------------------------------------------------ composite key
// hist (or NULL): hist_passes x 256 u64 counters, pass p = bits [8p, 8p+8) of the composite key -- the digit counts of EVERY radix
// pass, accumulated on the way (a count does not depend on the order the keys are in when a pass runs; this kernel is memory-bound
// and has the ALU slots, the passes do not)
template <int MAXC>
__device__ __forceinline__ void k_sort_pack_body(const DevProgram P, const i64 n, const SortSpec S, const SortPack K,
                                                      u64* __restrict__ key_lo, u64* __restrict__ key_hi, uint32_t* __restrict__ ids, u64* __restrict__ hist, const int hist_passes) {
  // deferred execution (P.n_dev): only rows [0, n_real) exist; the positions up to the bound n become PADDING records whose composite key is
  // all ones -- the passes are stable, so they end up behind every real row (a real row with that very key came first) and the first
  // n_real entries of the permutation are the sort of the real rows
  const i64 n_real = rows_of(P, n);
  const u128 pad_key = K.total_bits >= 128 ? ~(u128)0 : ((((u128)1) << K.total_bits) - 1);
  __shared__ uint32_t h0[SORT_MAX_PASSES][256];
  if (hist) { for (int p = 0; p < hist_passes; ++p) h0[p][threadIdx.x] = 0; __syncthreads(); }
  bool bad = false;      // K.check: a row the guessed layout does not hold
  // (taking two or four 64-row words per wave and step with all their column loads issued first, as the join and aggregate kernels
  // do, was measured here and changes nothing: 2.74 / 2.80 / 2.77 ms per sort of 2^27 rows with 1 / 2 / 4 words)
  const i64 nwords = (n + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * SWAVES + swave(); w < nwords; w += (i64)gridDim.x * SWAVES) {
    const i64 pos = (w << 6) + slane();
    if (pos >= n) continue;
    const bool real = pos < n_real;
    GPUQ_REGS_DECL;
    if (real) (void)GPUQ_EVAL(MAXC, P, pos);
    u128 comp = real ? (u128)0 : pad_key;
#pragma unroll
    for (int k = 0; k < MAX_SORT_KEYS; ++k) {
      if (k < S.n_keys && real) {
        const int r = __builtin_amdgcn_readfirstlane(S.reg[k]);
        const bool isn = (rnulls >> r) & 1;
        u128 field = 0;
        if (!isn) {
          const i128 v = sort_view(rlo[r], rhi[r], S.kind[k]) >> K.rshift[k];   // arithmetic shift keeps the order
          const i128 base = mk128(K.base_lo[k], K.base_hi[k]);
          field = S.desc[k] ? (u128)(base - v) : (u128)(v - base);
          if (K.check && K.vbits[k] < 128 && (field >> K.vbits[k]) != 0) bad = true;      // below the base wraps to a huge field: caught too
        } else if (K.check && K.null_bit[k] < 0) bad = true;
        if (K.null_bit[k] >= 0) {
          // nulls_first: NULL -> 0, value -> 1 in the bit above the value field
          const u128 flag = (isn != (bool)S.nulls_first[k]) ? 1 : 0;
          field |= flag << K.null_bit[k];
        }
        comp |= field << K.shift[k];
      }
    }
    if (hist) {
      // clustered inputs put the same high digit in every lane of a wave: 64 LDS atomics on one counter serialise (measured:
      // +0.8 ms on 2^27 rows sorted by (orderkey, date)); when the whole wave agrees one lane adds the count
      const u64 live = __ballot(true);
      for (int p = 0; p < hist_passes; ++p) {
        const uint32_t d = (uint32_t)(comp >> (8 * p)) & 0xFFu;
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        if (__ballot(d == d0) == live) { if (slane() == (int)__builtin_ctzll(live)) atomicAdd(&h0[p][d0], (uint32_t)__popcll(live)); }
        else atomicAdd(&h0[p][d], 1u);
      }
    }
    if (!ids) { key_lo[pos] = ((u64)comp << 32) | (u64)(uint32_t)pos; continue; }      // <= 32 key bits: one 8-byte (key, row) record
    key_lo[pos] = (u64)comp;
    if (key_hi) key_hi[pos] = (u64)(comp >> 64);
    ids[pos] = (uint32_t)pos;
  }
  if (hist) { __syncthreads(); for (int p = 0; p < hist_passes; ++p) if (h0[p][threadIdx.x]) atomicAdd((unsigned long long*)&hist[p * 256 + threadIdx.x], (unsigned long long)h0[p][threadIdx.x]); }
  if (K.check == 1 && hist && bad) atomicOr((unsigned long long*)&hist[SORT_MAX_PASSES * 256], 1ull);
  if (K.check == 2 && bad) atomicOr(P.flags, FLAG_SORT_LAYOUT);
}
#ifndef GPUQ_JIT
template <int MAXC>
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(SBLOCK) k_sort_pack(const DevProgram P, const i64 n, const SortSpec S, const SortPack K,
                                                      u64* __restrict__ key_lo, u64* __restrict__ key_hi, uint32_t* __restrict__ ids, u64* __restrict__ hist, const int hist_passes) { k_sort_pack_body<MAXC>(P, n, S, K, key_lo, key_hi, ids, hist, hist_passes); }
#endif
#elif GPUQ_JIT_KERNEL == 9
extern "C" __global__ void __launch_bounds__(SBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const SortSpec S, const SortPack K,
                                                      u64* __restrict__ key_lo, u64* __restrict__ key_hi, uint32_t* __restrict__ ids, u64* __restrict__ hist, const int hist_passes) { k_sort_pack_body<0>(P, n, S, K, key_lo, key_hi, ids, hist, hist_passes); }
#endif

// ------------------------------------------------------------------ small inputs, end to end in one block
// <= SORT_DIRECT_MAX rows: evaluate the keys, normalise each to an unsigned 128-bit value whose order is the requested
// order (ASC / DESC), keep a 2-bit rank per key for NULL placement, and run a bitonic network over (rank, value) x keys
// with the row id as the last tie-break (= the stable order).  No per-key min/max, hence no host round trip: sorting the
// handful of groups a final aggregate produces is otherwise one synchronisation and three launches.
constexpr int SORT_DIRECT_MAX = 512;
template <int MAXC>
__device__ __forceinline__ void k_sort_direct_body(const DevProgram P, const int n, const SortSpec S, uint32_t* __restrict__ perm) {
  __shared__ u64 vlo[MAX_SORT_KEYS][SORT_DIRECT_MAX], vhi[MAX_SORT_KEYS][SORT_DIRECT_MAX];
  __shared__ uint32_t rk[SORT_DIRECT_MAX], sid[SORT_DIRECT_MAX];      // rk: 2 bits per key (0 NULL first, 1 value, 2 NULL last, 3 padding)
  int m = 2; while (m < n) m <<= 1;
  const int i = threadIdx.x;
  const int n_real = (int)rows_of(P, (i64)n);      // deferred execution: positions [n_real, n) are padding like [n, m)
  if (i < m) {
    uint32_t ranks = 0xFFu; uint32_t id = 0xFFFFFFFFu;
    u64 lo[MAX_SORT_KEYS], hi[MAX_SORT_KEYS];
#pragma unroll
    for (int k = 0; k < MAX_SORT_KEYS; ++k) { lo[k] = 0; hi[k] = 0; }
    const bool act = i < n_real;
    GPUQ_REGS_DECL;
    if (act) (void)GPUQ_EVAL(MAXC, P, (i64)i);
    if (act) {
      ranks = 0; id = (uint32_t)i;
#pragma unroll
      for (int k = 0; k < MAX_SORT_KEYS; ++k) {
        if (k < S.n_keys) {
          const int r = __builtin_amdgcn_readfirstlane(S.reg[k]);
          const bool isn = (rnulls >> r) & 1;
          u128 u = 0;
          if (!isn) {
            u = (u128)sort_view(rlo[r], rhi[r], S.kind[k]) ^ ((u128)1 << 127);     // signed order -> unsigned order
            if (S.desc[k]) u = ~u;
          }
          lo[k] = (u64)u; hi[k] = (u64)(u >> 64);
          ranks |= (isn ? (S.nulls_first[k] ? 0u : 2u) : 1u) << (2 * k);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < MAX_SORT_KEYS; ++k) { vlo[k][i] = lo[k]; vhi[k][i] = hi[k]; }
    rk[i] = ranks; sid[i] = id;
  }
  __syncthreads();
  for (int kk = 2; kk <= m; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      const int x = i ^ j;
      if (i < m && x > i) {
        // a > b in (rank_0, value_0, rank_1, value_1, ..., id) order
        bool gt = false, decided = false;
        const uint32_t ra = rk[i], rb = rk[x];
#pragma unroll
        for (int k = 0; k < MAX_SORT_KEYS; ++k) {
          if (k < S.n_keys && !decided) {
            const uint32_t a2 = (ra >> (2 * k)) & 3u, b2 = (rb >> (2 * k)) & 3u;
            if (a2 != b2) { gt = a2 > b2; decided = true; }
            else if (a2 == 1u) {
              const u64 ah = vhi[k][i], bh = vhi[k][x];
              if (ah != bh) { gt = ah > bh; decided = true; }
              else { const u64 al = vlo[k][i], bl = vlo[k][x]; if (al != bl) { gt = al > bl; decided = true; } }
            }
          }
        }
        if (!decided) gt = sid[i] > sid[x];
        const bool up = (i & kk) == 0;
        if (gt == up) {
#pragma unroll
          for (int k = 0; k < MAX_SORT_KEYS; ++k) { const u64 tl = vlo[k][i], th = vhi[k][i]; vlo[k][i] = vlo[k][x]; vhi[k][i] = vhi[k][x]; vlo[k][x] = tl; vhi[k][x] = th; }
          const uint32_t tr = rk[i]; rk[i] = rk[x]; rk[x] = tr;
          const uint32_t ti = sid[i]; sid[i] = sid[x]; sid[x] = ti;
        }
      }
      __syncthreads();
    }
  }
  if (i < n) perm[i] = sid[i];
}
#ifndef GPUQ_JIT
template <int MAXC>
__global__ void __launch_bounds__(SORT_DIRECT_MAX) k_sort_direct(const DevProgram P, const int n, const SortSpec S, uint32_t* __restrict__ perm) { k_sort_direct_body<MAXC>(P, n, S, perm); }
#endif

// ------------------------------------------------------------------ partition ids
template <int MAXC>
__device__ __forceinline__ void k_part_pid_body(const DevProgram P, const i64 n_arg, const KeySpec K, const uint32_t nparts,
                                                     u64* __restrict__ pid_out, uint32_t* __restrict__ ids) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  const i64 nwords = (n + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * SWAVES + swave(); w < nwords; w += (i64)gridDim.x * SWAVES) {
    const i64 pos = (w << 6) + slane();
    if (pos >= n) continue;
    GPUQ_REGS_DECL;
    (void)GPUQ_EVAL(MAXC, P, pos);
    u64 h = 0x243F6A8885A308D3ull;
#pragma unroll
    for (int k = 0; k < MAX_KEYS; ++k) {
      if (k < K.n_keys) {
        const int r = __builtin_amdgcn_readfirstlane(K.key_reg[k]);
        const bool isn = (rnulls >> r) & 1;
        h = hash_combine(h, isn ? 0 : rlo[r], (K.key_wide[k] && !isn) ? rhi[r] : 0, isn);
      }
    }
    if (!ids) { pid_out[pos] = ((u64)(h % nparts) << 32) | (u64)(uint32_t)pos; continue; }      // one 8-byte (partition, row) record
    pid_out[pos] = h % nparts;
    ids[pos] = (uint32_t)pos;
  }
}
#ifndef GPUQ_JIT
template <int MAXC>
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(SBLOCK) k_part_pid(const DevProgram P, const i64 n, const KeySpec K, const uint32_t nparts,
                                                     u64* __restrict__ pid_out, uint32_t* __restrict__ ids) { k_part_pid_body<MAXC>(P, n, K, nparts, pid_out, ids); }
#endif
#elif GPUQ_JIT_KERNEL == 10
extern "C" __global__ void __launch_bounds__(SBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const uint32_t nparts,
                                                     u64* __restrict__ pid_out, uint32_t* __restrict__ ids) { k_part_pid_body<0>(P, n, K, nparts, pid_out, ids); }
#endif

// partition sizes: per-block LDS histogram (one global atomic per partition per block), then a
// one-block exclusive scan.  (Per-wave global atomics serialise on the handful of counter words: 26 ms
// for 2^27 rows into 8 partitions.)
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(SBLOCK) k_pid_count(const u64* __restrict__ pid, const i64 n, const uint32_t np, uint32_t* __restrict__ counts, const int shift) {
  extern __shared__ uint32_t hist[];
  const bool lds = np <= 8192;
  if (lds) { for (uint32_t d = threadIdx.x; d < np; d += SBLOCK) hist[d] = 0; __syncthreads(); }
  for (i64 i = (i64)blockIdx.x * SBLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * SBLOCK) {
    const uint32_t d = (uint32_t)(pid[i] >> shift);
    if (lds) atomicAdd(&hist[d], 1u); else atomicAdd(&counts[d], 1u);
  }
  if (lds) { __syncthreads(); for (uint32_t d = threadIdx.x; d < np; d += SBLOCK) { const uint32_t c = hist[d]; if (c) atomicAdd(&counts[d], c); } }
}
#endif
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(1024) k_part_scan(const uint32_t* __restrict__ counts, const uint32_t np, u64* __restrict__ offsets) {
  __shared__ u64 part[1024];
  const int t = threadIdx.x;
  const uint32_t per = (np + 1023) / 1024;
  const uint32_t a = t * per; uint32_t b = a + per; if (b > np) b = np;
  u64 s = 0; for (uint32_t i = a; i < b; ++i) s += counts[i];
  part[t] = s; __syncthreads();
  if (t == 0) { u64 run = 0; for (int k = 0; k < 1024; ++k) { const u64 v = part[k]; part[k] = run; run += v; } offsets[np] = run; }
  __syncthreads();
  u64 run = part[t];
  for (uint32_t i = a; i < b; ++i) { offsets[i] = run; run += counts[i]; }
}
#endif

#ifndef GPUQ_JIT
__global__ void __launch_bounds__(SBLOCK) k_gather_u64(const u64* __restrict__ src, const uint32_t* __restrict__ idx, const i64 n, u64* __restrict__ dst) {
  for (i64 i = (i64)blockIdx.x * SBLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * SBLOCK) dst[i] = src[idx[i]];
}
#endif

#ifndef GPUQ_JIT
// ------------------------------------------------------------------ single-read radix passes (decoupled look-back)
// A classic radix pass (histogram, scan, scatter) reads every key twice; round 1 shipped that form.  Here the digit counts of ALL
// passes are taken while the records are written (the pack kernel; k_radix_ghist for records that come from elsewhere: the count of
// a digit does not depend on the order of the keys), and a pass is ONE kernel that reads a tile once: every wave ranks its keys
// inside (wave, digit) sequences with match-any ballots, per-(wave, digit) counters in LDS carry the ranks across steps, one scan
// over the 256 digits gives tile-local positions, records go to their sorted place in LDS and are written out position by position
// (neighbouring lanes hold neighbouring records of one digit: coalesced runs).  The tile publishes its per-digit counts and
// learns the counts of the tiles before it by looking back over their published words (aggregate of one tile, or inclusive prefix
// of all tiles up to it) instead of waiting for a separate scan [UPSTREAM-KNOWLEDGE: Merrill & Garland's decoupled look-back /
// Adinets & Merrill's Onesweep].  Tiles are handed out by a ticket counter, so every tile a block waits for has already started:
// the look-back cannot deadlock whatever order the hardware schedules workgroups in.  With <= 32 key bits the record is one u64
// (key << 32 | row): a pass moves 16 B per row, the last one 12 B (row ids only).
#ifndef GPUQ_JIT
constexpr u64 LB_AGG = 1ull << 62, LB_PREFIX = 2ull << 62, LB_MASK = (1ull << 62) - 1;
constexpr int GHIST_MAX_PASSES = 8;
__global__ void __launch_bounds__(SBLOCK) k_radix_ghist(const u64* __restrict__ keys, const i64 n, const int shift0, const int npasses, u64* __restrict__ ghist) {
  __shared__ uint32_t cnt[GHIST_MAX_PASSES][RADIX];
  for (int i = threadIdx.x; i < GHIST_MAX_PASSES * RADIX; i += SBLOCK) (&cnt[0][0])[i] = 0;
  __syncthreads();
  for (i64 i = (i64)blockIdx.x * SBLOCK + threadIdx.x; i < n; i += (i64)gridDim.x * SBLOCK) {
    const u64 k = keys[i] >> shift0;
    for (int p = 0; p < npasses; ++p) atomicAdd(&cnt[p][(uint32_t)(k >> (8 * p)) & 0xFFu], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < npasses * RADIX; i += SBLOCK) { const uint32_t c = (&cnt[0][0])[i]; if (c) atomicAdd((unsigned long long*)&ghist[i], (unsigned long long)c); }
}
// counts -> exclusive bases, one block per pass
__global__ void __launch_bounds__(RADIX) k_radix_ghist_scan(u64* __restrict__ ghist) {
  __shared__ u64 ws[RADIX / 64];
  u64* h = ghist + (size_t)blockIdx.x * RADIX;
  const int t = threadIdx.x, l = t & 63;
  const u64 c = h[t]; u64 x = c;
  for (int off = 1; off < 64; off <<= 1) { const u64 y = __shfl_up(x, off); if (l >= off) x += y; }
  if (l == 63) ws[t >> 6] = x;
  __syncthreads();
  u64 pre = 0; for (int q = 0; q < (t >> 6); ++q) pre += ws[q];
  h[t] = pre + x - c;
}
// OROUNDS keys per thread: the tile is SBLOCK * OROUNDS records.  Longer tiles mean longer runs per digit in the write-out and fewer
// barriers / look-back words per key (measured on 2^27 packed records: 2048-record tiles 1.2 ms per pass, 4096: 0.75, 8192: 0.63); the
// (key, value) form uses 6144 because its 12 bytes per record would leave one block per CU at 8192.
template <bool HASVAL, int OROUNDS>
__global__ void __launch_bounds__(SBLOCK) k_onesweep(const u64* __restrict__ keys, const uint32_t* __restrict__ vals, const i64 n, const int shift,
                                                     const u64* __restrict__ gexcl, u64* __restrict__ look, uint32_t* __restrict__ ticket,
                                                     u64* __restrict__ keys_out, uint32_t* __restrict__ vals_out, const int ids_only) {
  __shared__ u64 sk[(SBLOCK * OROUNDS)];
  __shared__ uint32_t sv[HASVAL ? (SBLOCK * OROUNDS) : 1];
  __shared__ uint32_t wcnt[SWAVES][RADIX];
  __shared__ uint32_t dstart[RADIX];
  __shared__ u64 gbase[RADIX];
  __shared__ uint32_t wsum[SWAVES];
  __shared__ uint32_t s_tile;
  const int t = threadIdx.x, w = swave(), l = slane();
  const u64 lt = (1ull << l) - 1;
  if (t == 0) s_tile = atomicAdd(ticket, 1u);
  for (int i = t; i < SWAVES * RADIX; i += SBLOCK) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  const i64 T = (i64)s_tile;
  const i64 s0 = T * (SBLOCK * OROUNDS);
  if (s0 >= n) return;                                 // (grid == number of tiles: not reached)
  const i64 b = s0 + (SBLOCK * OROUNDS) < n ? s0 + (SBLOCK * OROUNDS) : n;
  // 1. load the wave's quarter, rank every key inside its (wave, digit) sequence
  const i64 w0 = s0 + (i64)w * ((SBLOCK * OROUNDS) / SWAVES);
  u64 k[OROUNDS]; uint32_t v[HASVAL ? OROUNDS : 1]; uint32_t pos[OROUNDS];
#pragma unroll
  for (int r = 0; r < OROUNDS; ++r) {
    const i64 i = w0 + r * 64 + l;
    const bool act = i < b;
    k[r] = act ? keys[i] : 0;
    if (HASVAL) v[r] = act ? vals[i] : 0;
  }
#pragma unroll
  for (int r = 0; r < OROUNDS; ++r) {
    const i64 i = w0 + r * 64 + l;
    const bool act = i < b;
    const uint32_t d = (uint32_t)(k[r] >> shift) & 0xFFu;
    // match-any: lanes whose digit equals mine.  Per bit the lane's sign-extended bit S (0 / ~0) is XORed into the ballot of that bit:
    // a lane differs from me in this bit exactly where (ballot ^ S) is set; the differences of two bits are OR-ed per instruction
    // (v_bitop3 on gfx950).  4-5 VALU instructions per bit instead of 7 for the select-and-AND form (62 % of this kernel's VALU work was here).
    uint32_t dlo = 0, dhi = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit += 2) {
      const uint32_t s0 = (uint32_t)__builtin_amdgcn_sbfe((int)d, bit, 1), s1 = (uint32_t)__builtin_amdgcn_sbfe((int)d, bit + 1, 1);
      const u64 m0 = __ballot(s0 != 0), m1 = __ballot(s1 != 0);
      // gfx950's three-input boolean op: d | (m ^ s) in one instruction (truth table 0xF0 | (0xCC ^ 0xAA) = 0xF6)
      dlo = __builtin_amdgcn_bitop3_b32(dlo, (uint32_t)m0, s0, 0xF6); dhi = __builtin_amdgcn_bitop3_b32(dhi, (uint32_t)(m0 >> 32), s0, 0xF6);
      dlo = __builtin_amdgcn_bitop3_b32(dlo, (uint32_t)m1, s1, 0xF6); dhi = __builtin_amdgcn_bitop3_b32(dhi, (uint32_t)(m1 >> 32), s1, 0xF6);
    }
    const u64 same = ~(((u64)dhi << 32) | (u64)dlo) & __ballot(act);
    const uint32_t before = act ? wcnt[w][d] : 0;
    pos[r] = before + (uint32_t)__popcll(same & lt);
    __builtin_amdgcn_wave_barrier();
    if (act && (same >> l) <= 1ull) wcnt[w][d] = before + (uint32_t)__popcll(same);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // 2. tile-local layout (t == digit); the digit's count is published before anything else happens
  uint32_t tot = 0;
  {
    uint32_t c[SWAVES];
#pragma unroll
    for (int q = 0; q < SWAVES; ++q) { c[q] = wcnt[q][t]; tot += c[q]; }
    __hip_atomic_store(&look[(size_t)T * RADIX + t], (T == 0 ? LB_PREFIX : LB_AGG) | (u64)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t x = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if (l >= off) x += y; }
    if (l == 63) wsum[w] = x;
    __syncthreads();
    uint32_t pre = 0;
#pragma unroll
    for (int q = 0; q < SWAVES; ++q) if (q < w) pre += wsum[q];
    const uint32_t ds = pre + x - tot;
    dstart[t] = ds;
    uint32_t run = ds;
#pragma unroll
    for (int q = 0; q < SWAVES; ++q) { wcnt[q][t] = run; run += c[q]; }
  }
  __syncthreads();
  // 3. keys to their sorted place in LDS
#pragma unroll
  for (int r = 0; r < OROUNDS; ++r) {
    const i64 i = w0 + r * 64 + l;
    if (i < b) { const uint32_t d = (uint32_t)(k[r] >> shift) & 0xFFu; const uint32_t j = wcnt[w][d] + pos[r]; sk[j] = k[r]; if (HASVAL) sv[j] = v[r]; }
  }
  // 4. look back over the tiles before this one (t == digit): sum aggregates until a tile that knows its inclusive prefix
  {
    u64 excl = 0;
    if (T > 0) {
      // LBW predecessors per step: their words are independent loads in flight together (a one-at-a-time walk pays a full
      // device-scope round trip per tile, and at any moment hundreds of tiles have published only their aggregate)
      constexpr int LBW = 4;
      bool done = false;
      for (i64 j = T - 1; !done; j -= LBW) {
        u64 x[LBW];
#pragma unroll
        for (int q = 0; q < LBW; ++q) x[q] = j - q >= 0 ? __hip_atomic_load(&look[(size_t)(j - q) * RADIX + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_PREFIX;
#pragma unroll
        for (int q = 0; q < LBW; ++q) {
          if (done) break;
          while ((x[q] >> 62) == 0) { __builtin_amdgcn_s_sleep(1); x[q] = __hip_atomic_load(&look[(size_t)(j - q) * RADIX + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          excl += x[q] & LB_MASK;
          done = (x[q] >> 62) == 2;
        }
      }
      __hip_atomic_store(&look[(size_t)T * RADIX + t], LB_PREFIX | (excl + (u64)tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gbase[t] = gexcl[t] + excl;
  }
  __syncthreads();
  // 5. out, position by position: neighbouring lanes hold neighbouring keys of the same digit
  const int cnt = (int)(b - s0);
#pragma unroll 4
  for (int j = t; j < cnt; j += SBLOCK) {
    const u64 kk = sk[j];
    const uint32_t d = (uint32_t)(kk >> shift) & 0xFFu;
    const u64 dst = gbase[d] + (u64)((uint32_t)j - dstart[d]);
    // ids_only: 0 = records (and values), 1 = row ids only (a sort's last pass), 2 = packed records AND their row ids
    if (ids_only != 1) { keys_out[dst] = kk; if (HASVAL) vals_out[dst] = sv[j]; }
    if (ids_only != 0 && (ids_only == 1 || !HASVAL)) vals_out[dst] = HASVAL ? sv[j] : (uint32_t)kk;
  }
}
#endif

// ------------------------------------------------------------------ ordered fan-in: merge of sorted runs
// CoalesceTasksExec with an order / SortPreservingMergeExec (coalesce_tasks.rs:162-170 `streaming_merge`): the inputs are k
// runs, each already in the requested order.  The composite keys of the concatenation are packed exactly as for a sort; the runs
// are then merged pairwise, log2(k) rounds, each round one launch that reads and writes every (key, row) record once.  A block
// produces MTILE consecutive outputs of one pair: two binary searches along the merge path [UPSTREAM-KNOWLEDGE: Green, McColl &
// Bader, "GPU Merge Path"] bound the slices of both runs that feed them, the slices are staged in LDS with coalesced loads, every
// thread finds its own start inside the tile the same way and merges MVT outputs sequentially.  Ties take the element of the
// LEFT run first: across rounds equal keys keep (run, row) order -- what a loser tree that prefers the lower-numbered stream yields.
#ifndef GPUQ_JIT
constexpr int MVT = 8;
constexpr int MTILE = SBLOCK * MVT;      // 2048 records per block
struct MKey { u64 hi, lo; };
__device__ __forceinline__ bool mk_le(const MKey a, const MKey b) { return a.hi < b.hi || (a.hi == b.hi && a.lo <= b.lo); }      // a may precede an equal b
// number of A elements among the first d outputs of merge(A[0..na), B[0..nb)); key(i) of A / B through the accessors
template <class FA, class FB>
__device__ __forceinline__ int merge_path(const int d, const int na, const int nb, FA keyA, FB keyB) {
  int lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (mk_le(keyA(mid), keyB(d - 1 - mid))) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// Tile boundaries along the merge path, ALL tiles of a round at once: one thread per (pair, tile) runs the ~26-step binary search
// over global memory.  Inside the merge kernel the same search was a serial 20 us in front of every tile (two threads, dependent
// loads): 1.3 of a round's 2.0 ms at 2^27 records.
template <bool WIDE>
__global__ void __launch_bounds__(SBLOCK) k_merge_partition(const u64* __restrict__ klo, const u64* __restrict__ khi, const i64* __restrict__ pairs, const int tiles_per_pair,
                                                            i64* __restrict__ splits) {
  const int t = blockIdx.x * SBLOCK + threadIdx.x;
  if (t > tiles_per_pair) return;                      // tiles_per_pair + 1 boundaries per pair
  const i64 a0 = pairs[3 * blockIdx.y], a1 = pairs[3 * blockIdx.y + 1], a2 = pairs[3 * blockIdx.y + 2];
  const i64 na = a1 - a0, nb = a2 - a1;
  i64 d = (i64)t * MTILE; if (d > na + nb) d = na + nb;
  i64 lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {
    const i64 mid = (lo + hi) >> 1;
    const MKey ka{WIDE ? khi[a0 + mid] : 0, klo[a0 + mid]}, kb{WIDE ? khi[a1 + d - 1 - mid] : 0, klo[a1 + d - 1 - mid]};
    if (mk_le(ka, kb)) lo = mid + 1; else hi = mid;
  }
  splits[(size_t)blockIdx.y * (size_t)(tiles_per_pair + 1) + t] = lo;
}
// pairs[p] = {a0, a1, a2}: runs [a0, a1) and [a1, a2) of the input become one run at the same place of the output (a1 == a2: copy)
template <bool WIDE>
__global__ void __launch_bounds__(SBLOCK) k_merge_pairs(const u64* __restrict__ klo, const u64* __restrict__ khi, const uint32_t* __restrict__ ids, const i64* __restrict__ pairs,
                                                        const i64* __restrict__ splits, const int tiles_per_pair,
                                                        u64* __restrict__ klo_out, u64* __restrict__ khi_out, uint32_t* __restrict__ ids_out) {
  __shared__ u64 slo[MTILE]; __shared__ u64 shi[WIDE ? MTILE : 1]; __shared__ uint32_t sid[MTILE];
  const i64 a0 = pairs[3 * blockIdx.y], a1 = pairs[3 * blockIdx.y + 1], a2 = pairs[3 * blockIdx.y + 2];
  const i64 na = a1 - a0, nb = a2 - a1;
  const i64 d0 = (i64)blockIdx.x * MTILE;
  if (d0 >= na + nb) return;
  const i64 d1 = d0 + MTILE < na + nb ? d0 + MTILE : na + nb;
  auto gA = [&](i64 i) -> MKey { return MKey{WIDE ? khi[a0 + i] : 0, klo[a0 + i]}; };
  auto gB = [&](i64 i) -> MKey { return MKey{WIDE ? khi[a1 + i] : 0, klo[a1 + i]}; };
  // 1. the tile's slices of both runs: the boundaries k_merge_partition found
  const i64* sp = splits + (size_t)blockIdx.y * (size_t)(tiles_per_pair + 1) + blockIdx.x;
  const i64 i0 = sp[0], i1 = sp[1];
  const i64 j0 = d0 - i0, j1 = d1 - i1;
  const int ca = (int)(i1 - i0), cb = (int)(j1 - j0);      // ca + cb == d1 - d0 <= MTILE
  // 2. stage: A's slice at [0, ca), B's at [ca, ca + cb)
  for (int q = threadIdx.x; q < ca + cb; q += SBLOCK) {
    const i64 src = q < ca ? a0 + i0 + q : a1 + j0 + (q - ca);
    slo[q] = klo[src]; if (WIDE) shi[q] = khi[src]; sid[q] = ids[src];
  }
  __syncthreads();
  // 3. every thread merges MVT outputs from LDS
  auto lA = [&](int i) -> MKey { return MKey{WIDE ? shi[i] : 0, slo[i]}; };
  auto lB = [&](int i) -> MKey { return MKey{WIDE ? shi[ca + i] : 0, slo[ca + i]}; };
  const int t0 = threadIdx.x * MVT;
  const int total = ca + cb;
  u64 olo[MVT], ohi[MVT]; uint32_t oid[MVT];
  if (t0 < total) {
    int i = merge_path(t0, ca, cb, lA, lB), j = t0 - i;
#pragma unroll
    for (int q = 0; q < MVT; ++q) {
      if (t0 + q >= total) break;
      const bool takeA = j >= cb || (i < ca && mk_le(lA(i), lB(j)));
      const int src = takeA ? i : ca + j;
      olo[q] = slo[src]; ohi[q] = WIDE ? shi[src] : 0; oid[q] = sid[src];
      if (takeA) ++i; else ++j;
    }
  }
  __syncthreads();
  // 4. through LDS again so that the global stores are coalesced
  if (t0 < total) {
#pragma unroll
    for (int q = 0; q < MVT; ++q) if (t0 + q < total) { slo[t0 + q] = olo[q]; if (WIDE) shi[t0 + q] = ohi[q]; sid[t0 + q] = oid[q]; }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < total; q += SBLOCK) {
    const i64 dst = a0 + d0 + q;
    klo_out[dst] = slo[q]; if (WIDE) khi_out[dst] = shi[q]; ids_out[dst] = sid[q];
  }
}
#endif

// ------------------------------------------------------------------ launchers
static int sgrid(i64 n, int blocks_per_cu) {
  const i64 nwords = (n + 63) >> 6;
  i64 need = (nwords + SWAVES - 1) / SWAVES; if (need < 1) need = 1;
  const i64 cap = (i64)num_cus() * blocks_per_cu;
  return (int)(need < cap ? need : cap);
}
int sort_minmax_blocks(i64 n) { return sgrid(n, 4); }
void launch_sort_minmax(hipStream_t s, const DevProgram& P, i64 n, const SortSpec& S, u64* out, int nblocks, i64 wstep) {
  if (wstep < 1) wstep = 1;
  if (jit_override().fn && jit_override().kernel_id == 8) {
    (void)jit_launch(jit_override().fn, dim3(nblocks), dim3(SBLOCK), 0, s, P, n, S, out, wstep);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_sort_minmax<M>, dim3(nblocks), dim3(SBLOCK), 0, s, P, n, S, out, wstep)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
int sort_max_passes() { return SORT_MAX_PASSES; }
void launch_sort_pack(hipStream_t s, const DevProgram& P, i64 n, const SortSpec& S, const SortPack& K, u64* key_lo, u64* key_hi, uint32_t* ids, u64* hist, int hist_passes) {
  if (n <= 0) return;
  if (hist) (void)hipMemsetAsync(hist, 0, (size_t)hist_passes * RADIX * 8, s);
  if (hist && K.check == 1) (void)hipMemsetAsync(hist + (size_t)SORT_MAX_PASSES * RADIX, 0, 8, s);      // the "layout does not hold" word
  if (jit_override().fn && jit_override().kernel_id == 9) {
    (void)jit_launch(jit_override().fn, dim3(sgrid(n, 8)), dim3(SBLOCK), 0, s, P, n, S, K, key_lo, key_hi, ids, hist, hist_passes);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_sort_pack<M>, dim3(sgrid(n, 8)), dim3(SBLOCK), 0, s, P, n, S, K, key_lo, key_hi, ids, hist, hist_passes)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
void launch_part_pid(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, uint32_t nparts, u64* pid_out, uint32_t* ids) {
  if (n <= 0) return;
  if (jit_override().fn && jit_override().kernel_id == 10) {
    (void)jit_launch(jit_override().fn, dim3(sgrid(n, 8)), dim3(SBLOCK), 0, s, P, n, K, nparts, pid_out, ids);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_part_pid<M>, dim3(sgrid(n, 8)), dim3(SBLOCK), 0, s, P, n, K, nparts, pid_out, ids)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
// counts (u32[np]) -> the 256 u64 digit counts of a single 8-bit pass (np <= 256: the digit IS the partition)
__global__ void __launch_bounds__(RADIX) k_counts_to_ghist(const uint32_t* __restrict__ counts, const uint32_t np, u64* __restrict__ ghist) {
  ghist[threadIdx.x] = threadIdx.x < np ? (u64)counts[threadIdx.x] : 0;
}
void launch_counts_to_ghist(hipStream_t s, const uint32_t* counts, uint32_t np, u64* ghist) {
  hipLaunchKernelGGL(k_counts_to_ghist, dim3(1), dim3(RADIX), 0, s, counts, np, ghist);
}
void launch_part_offsets(hipStream_t s, const u64* pid, i64 n, uint32_t nparts, uint32_t* counts_ws, u64* offsets_out, int shift) {
  (void)hipMemsetAsync(counts_ws, 0, (size_t)(nparts + 1) * 4, s);
  i64 need = (n + SBLOCK - 1) / SBLOCK; const i64 cap = (i64)num_cus() * 8; if (need < 1) need = 1;
  hipLaunchKernelGGL(k_pid_count, dim3((int)(need < cap ? need : cap)), dim3(SBLOCK), nparts <= 8192 ? (size_t)nparts * 4 : 0, s, pid, n, nparts, counts_ws, shift);
  hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), 0, s, (const uint32_t*)counts_ws, nparts, offsets_out);
}
void launch_gather_u64(hipStream_t s, const u64* src, const uint32_t* idx, i64 n, u64* dst) {
  if (n <= 0) return;
  i64 need = (n + SBLOCK - 1) / SBLOCK; const i64 cap = (i64)num_cus() * 16;
  hipLaunchKernelGGL(k_gather_u64, dim3((int)(need < cap ? need : cap)), dim3(SBLOCK), 0, s, src, idx, n, dst);
}
// ------------------------------------------------------------------ small inputs: one block, bitonic network in LDS
// Replaces the radix passes (5 launches per 8 key bits) when the whole input fits one block: the sort of a final
// aggregate's handful of groups (q1: 4 rows) is pure launch latency otherwise.  Order = (composite key, row id): the row
// id as the last tie-break makes the network's result the stable order the LSD radix produces.
constexpr int SMALL_SORT_MAX = 2048;
__global__ void __launch_bounds__(1024) k_sort_small(const u64* __restrict__ klo, const u64* __restrict__ khi, const uint32_t* __restrict__ ids,
                                                     const int n, uint32_t* __restrict__ out) {
  __shared__ u64 slo[SMALL_SORT_MAX], shi[SMALL_SORT_MAX];
  __shared__ uint32_t sid[SMALL_SORT_MAX];
  int m = 2; while (m < n) m <<= 1;
  for (int i = threadIdx.x; i < m; i += 1024) {
    if (i < n) { slo[i] = klo[i]; shi[i] = khi ? khi[i] : 0ull; sid[i] = ids[i]; }
    else { slo[i] = ~0ull; shi[i] = ~0ull; sid[i] = 0xFFFFFFFFu; }          // padding sorts behind every real row
  }
  __syncthreads();
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < m; i += 1024) {
        const int x = i ^ j;
        if (x > i) {
          const u64 ah = shi[i], al = slo[i], bh = shi[x], bl = slo[x];
          const uint32_t ai = sid[i], bi = sid[x];
          const bool a_gt_b = ah != bh ? ah > bh : (al != bl ? al > bl : ai > bi);
          const bool up = (i & k) == 0;
          if (a_gt_b == up) { shi[i] = bh; slo[i] = bl; sid[i] = bi; shi[x] = ah; slo[x] = al; sid[x] = ai; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += 1024) out[i] = sid[i];
}
int sort_small_max() { return SMALL_SORT_MAX; }
int sort_direct_max() { return SORT_DIRECT_MAX; }
void launch_sort_direct(hipStream_t s, const DevProgram& P, i64 n, const SortSpec& S, uint32_t* perm) {
  if (n <= 0) return;
#define CALL(M) hipLaunchKernelGGL(k_sort_direct<M>, dim3(1), dim3(SORT_DIRECT_MAX), 0, s, P, (int)n, S, perm)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
}
void launch_sort_small(hipStream_t s, const u64* klo, const u64* khi, const uint32_t* ids, i64 n, uint32_t* out) {
  if (n > 0) hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(1024), 0, s, klo, khi, ids, (int)n, out);
}

constexpr int OS_ROUNDS_PACKED = 32, OS_ROUNDS_KV = 24;      // 8192-record tiles for 8-byte records (6144: 3.4 ms instead of 3.15 on 2^27 rows), 6144 for (key, value) (4096: 6.55 ms instead of 5.9 over 5 passes; 8192 leaves one block per CU)
size_t onesweep_ws_bytes(i64 n) { const i64 tiles = (n + SBLOCK * OS_ROUNDS_KV - 1) / (SBLOCK * OS_ROUNDS_KV); return (size_t)tiles * RADIX * 8 + 256; }      // look-back words + ticket (the smallest tile: an upper bound)
int onesweep_max_passes() { return GHIST_MAX_PASSES; }
// ghist: npasses * 256 u64 digit counts (launch_onesweep_pass turns a pass's counts into bases)
void launch_radix_ghist(hipStream_t s, const u64* keys, i64 n, int shift0, int npasses, u64* ghist) {
  if (n <= 0 || npasses <= 0) return;
  (void)hipMemsetAsync(ghist, 0, (size_t)npasses * RADIX * 8, s);
  i64 need = (n + SBLOCK * 16 - 1) / (SBLOCK * 16); const i64 cap = (i64)num_cus() * 8;
  hipLaunchKernelGGL(k_radix_ghist, dim3((unsigned)std::max<i64>(1, std::min(need, cap))), dim3(SBLOCK), 0, s, keys, n, shift0, npasses, ghist);
}
// ORDER BY one integer-like key: the key column IN SORTED ORDER is rebuilt from the sorted records (value = base +- field) instead of gathered
// through the permutation -- a sequential read and write where the gather pays a 64-byte line per 16-byte value (SF300 sort shard: the
// l_extendedprice gather was ~6 of 16.6 ms).  recs: the last pass's output (packed: composite in the high 32 bits); width: bytes per value.
__global__ void __launch_bounds__(SBLOCK) k_sort_decode(const u64* __restrict__ recs, const int rec_shift, const i64 n, const SortPack K, const int desc, const int nulls_first,
                                                        const int width, uint8_t* __restrict__ out, u64* __restrict__ valid_out) {
  const i64 nwords = (n + 63) >> 6;
  const i128 base = mk128(K.base_lo[0], K.base_hi[0]);
  const u64 vmask = K.vbits[0] >= 64 ? ~0ull : ((1ull << K.vbits[0]) - 1);
  for (i64 w = (i64)blockIdx.x * SWAVES + swave(); w < nwords; w += (i64)gridDim.x * SWAVES) {
    const i64 i = (w << 6) + slane();
    bool valid = false;
    if (i < n) {
      const u64 comp = recs[i] >> rec_shift;
      bool isn = false;
      if (K.null_bit[0] >= 0) isn = (((comp >> K.null_bit[0]) & 1ull) != 0) != (nulls_first != 0);      // flag = isnull XOR nulls_first (k_sort_pack)
      const u64 field = comp & vmask;
      const i128 v = isn ? (i128)0 : (desc ? base - (i128)field : base + (i128)field);
      valid = !isn;
      switch (width) {
        case 1: out[i] = (uint8_t)v; break;
        case 2: ((uint16_t*)out)[i] = (uint16_t)v; break;
        case 4: ((uint32_t*)out)[i] = (uint32_t)v; break;
        case 8: ((u64*)out)[i] = (u64)v; break;
        default: ((ulonglong2*)out)[i] = make_ulonglong2((u64)v, (u64)((u128)v >> 64)); break;
      }
    }
    if (valid_out) { const u64 vb = __ballot(valid); if (slane() == 0) valid_out[w] = vb; }
  }
}
void launch_sort_decode(hipStream_t s, const u64* recs, int rec_shift, i64 n, const SortPack& K, int desc, int nulls_first, int width, void* out, u64* valid_out) {
  if (n <= 0) return;
  i64 need = ((n + 63) / 64 + SWAVES - 1) / SWAVES; const i64 cap = (i64)num_cus() * 16;
  hipLaunchKernelGGL(k_sort_decode, dim3((unsigned)(need < cap ? need : cap)), dim3(SBLOCK), 0, s, recs, rec_shift, n, K, desc, nulls_first, width, (uint8_t*)out, valid_out);
}

// one stable 8-bit pass; vals == NULL: packed (key << 32 | row) records; ids_only: only vals_out is written (the last pass)
void launch_onesweep_pass(hipStream_t s, const u64* keys, const uint32_t* vals, i64 n, int shift, u64* gexcl, void* ws, size_t ws_bytes,
                          u64* keys_out, uint32_t* vals_out, int ids_only) {
  if (n <= 0) return;
  const int rounds = vals ? OS_ROUNDS_KV : OS_ROUNDS_PACKED;
  const i64 tile = (i64)SBLOCK * rounds;
  const i64 tiles = (n + tile - 1) / tile;
  (void)hipMemsetAsync(ws, 0, ws_bytes, s);
  u64* look = (u64*)ws; uint32_t* ticket = (uint32_t*)((char*)ws + (size_t)tiles * RADIX * 8);
  hipLaunchKernelGGL(k_radix_ghist_scan, dim3(1), dim3(RADIX), 0, s, gexcl);      // this pass's counts (taken by the pack kernel) -> bases
  if (vals) hipLaunchKernelGGL((k_onesweep<true, OS_ROUNDS_KV>), dim3((unsigned)tiles), dim3(SBLOCK), 0, s, keys, vals, n, shift, (const u64*)gexcl, look, ticket, keys_out, vals_out, ids_only);
  else hipLaunchKernelGGL((k_onesweep<false, OS_ROUNDS_PACKED>), dim3((unsigned)tiles), dim3(SBLOCK), 0, s, keys, vals, n, shift, (const u64*)gexcl, look, ticket, keys_out, vals_out, ids_only);
}

// one round: n_pairs triples in `pairs` (device), the longest pair has max_len records
size_t merge_splits_entries(i64 max_len, int n_pairs) { return (size_t)n_pairs * (size_t)((max_len + MTILE - 1) / MTILE + 1); }
void launch_merge_pairs(hipStream_t s, const u64* klo, const u64* khi, const uint32_t* ids, const i64* pairs, int n_pairs, i64 max_len, i64* splits,
                        u64* klo_out, u64* khi_out, uint32_t* ids_out) {
  if (n_pairs <= 0 || max_len <= 0) return;
  const int tpp = (int)((max_len + MTILE - 1) / MTILE);
  const dim3 pg((unsigned)((tpp + 1 + SBLOCK - 1) / SBLOCK), (unsigned)n_pairs), mg((unsigned)tpp, (unsigned)n_pairs);
  if (khi) {
    hipLaunchKernelGGL(k_merge_partition<true>, pg, dim3(SBLOCK), 0, s, klo, khi, pairs, tpp, splits);
    hipLaunchKernelGGL(k_merge_pairs<true>, mg, dim3(SBLOCK), 0, s, klo, khi, ids, pairs, (const i64*)splits, tpp, klo_out, khi_out, ids_out);
  } else {
    hipLaunchKernelGGL(k_merge_partition<false>, pg, dim3(SBLOCK), 0, s, klo, khi, pairs, tpp, splits);
    hipLaunchKernelGGL(k_merge_pairs<false>, mg, dim3(SBLOCK), 0, s, klo, khi, ids, pairs, (const i64*)splits, tpp, klo_out, khi_out, ids_out);
  }
}

#endif  // GPUQ_JIT

}  // namespace gpuq
