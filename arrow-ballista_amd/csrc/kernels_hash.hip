// Hash-table operators for gfx950: high-cardinality AggregateExec and HashJoinExec build/probe.
//
// Replaces, on the reference's path: a6 AggregateExec (datafusion.proto:1405-1450)
// for large group counts and a7 HashJoinExec (datafusion.proto:1346-1360, join types :280-289;
// ctor surface ballista/core/src/physical_optimizer/task_group.rs:306-315).  Integer hash / compare /
// gather work, HBM- and atomic-bound; no MFMA.
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, "8-B agent atomics both sides"):
// every word of a slot that is read inside the kernel that writes it is accessed ONLY through
// 8-byte agent-scope atomics.  A slot is claimed by CAS(state: EMPTY->LOCKED); the winner stores
// the key words (write-through), drains them (s_waitcnt vmcnt(0)) and then publishes state=tag
// with one more atomic store.  Readers poll the state with a relaxed atomic load and read the key
// words with atomic loads only after they saw the tag.  Correctness never depends on dispatch
// order or XCD placement; every probe sequence is bounded by n_slots.
#include "gpuq_kernels.h"

namespace gpuq {

constexpr int HBLOCK = 256;
constexpr int HWAVES = HBLOCK / 64;
constexpr uint32_t NIL = 0xFFFFFFFFu;

__device__ __forceinline__ int hlane() { return threadIdx.x & 63; }
__device__ __forceinline__ int hwave() { return threadIdx.x >> 6; }

__device__ __forceinline__ u64 a_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void a_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 a_cas(u64* p, u64 expected, u64 desired) {
  __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expected;
}
__device__ __forceinline__ u64 a_add(u64* p, u64 v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Key words of the current row + their hash.  Returns true when any key is NULL.
__device__ __forceinline__ bool make_key(const KeySpec& K, GPUQ_REGS_CPARAM, u64 (&kw)[MAX_KW], u64& hash) {
  uint32_t knull = 0;
  u64 h = 0x243F6A8885A308D3ull;
#pragma unroll
  for (int k = 0; k < MAX_KEYS; ++k) {
    if (k < K.n_keys) {
      const int r = __builtin_amdgcn_readfirstlane(K.key_reg[k]);
      const bool isn = (rnulls >> r) & 1;
      const u64 lo = isn ? 0 : rlo[r], hi = isn ? 0 : rhi[r];
      knull |= (uint32_t)isn << k;
      h = hash_combine(h, lo, K.key_wide[k] ? hi : 0, isn);
    }
  }
  // kw[] is indexed statically only (it must stay in VGPRs); the source register is the dynamic part
#pragma unroll
  for (int q = 0; q < MAX_KW; ++q) {
    kw[q] = 0;
    if (q < K.key_words) {
      const int half = __builtin_amdgcn_readfirstlane(K.word_half[q]);
      if (half == 2) kw[q] = knull;
      else {
        const int r = __builtin_amdgcn_readfirstlane(K.word_reg[q]);
        const bool isn = (rnulls >> r) & 1;
        const u64 vl = rlo[r], vh = rhi[r];   // read both: a select of element pointers would demote the file to scratch
        const u64 v = (vl & ((u64)half - 1)) | (vh & (0 - (u64)half));
        kw[q] = isn ? 0 : v;
      }
    }
  }
  hash = h;
  return knull != 0;
}

__device__ __forceinline__ uint32_t tag_of(u64 h) { return (uint32_t)(h >> 32) | 2u; }

// Find the slot of a key or claim a new one.  Returns slot index, or ~0 when the table is full.
__device__ __forceinline__ u64 ht_find_or_insert(const HashTable& T, const u64 (&kw)[MAX_KW], u64 h, uint32_t payload, bool& inserted) {
  const u64 mask = T.n_slots - 1;
  u64 s = h & mask;
  const uint32_t tag = tag_of(h);
  inserted = false;
  u64 probes = 0;
  while (probes < T.n_slots) {
    u64* slot = T.slots + s * (u64)T.slot_words;
    u64 w0 = a_load(slot);
    uint32_t st = (uint32_t)w0;
    if (st == 0u) {
      const u64 old = a_cas(slot, 0ull, 1ull);
      if (old == 0ull) {
#pragma unroll
        for (int q = 0; q < MAX_KW; ++q) if (q < T.key_words) a_store(slot + 1 + q, kw[q]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // key words drained before the tag is visible
        a_store(slot, ((u64)payload << 32) | tag);
        inserted = true;
        return s;
      }
      st = (uint32_t)old;
    }
    if (st == 1u) { __builtin_amdgcn_s_sleep(1); continue; }  // being published: look again
    if (st == tag) {
      bool eq = true;
#pragma unroll
      for (int q = 0; q < MAX_KW; ++q) if (q < T.key_words) eq = eq && (a_load(slot + 1 + q) == kw[q]);
      if (eq) return s;
    }
    s = (s + 1) & mask;
    ++probes;
  }
  return ~0ull;
}

// Read-only lookup (table finished by an earlier kernel): plain loads.
__device__ __forceinline__ bool ht_find(const HashTable& T, const u64 (&kw)[MAX_KW], u64 h, uint32_t& payload) {
  const u64 mask = T.n_slots - 1;
  u64 s = h & mask;
  const uint32_t tag = tag_of(h);
  for (u64 probes = 0; probes < T.n_slots; ++probes) {
    const u64* slot = T.slots + s * (u64)T.slot_words;
    u64 w0, k0 = 0;
    if (T.slot_words == 2) { const ulonglong2 v = *(const ulonglong2*)slot; w0 = v.x; k0 = v.y; }
    else w0 = slot[0];
    const uint32_t st = (uint32_t)w0;
    if (st == 0u) return false;
    if (st == tag) {
      bool eq = true;
      if (T.slot_words == 2) eq = (k0 == kw[0]);
      else {
#pragma unroll
        for (int q = 0; q < MAX_KW; ++q) if (q < T.key_words) eq = eq && (slot[1 + q] == kw[q]);
      }
      if (eq) { payload = (uint32_t)(w0 >> 32); return true; }
    }
    s = (s + 1) & mask;
  }
  return false;
}

// ------------------------------------------------------------------ table init
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(HBLOCK) k_ht_init(const HashTable T, const AggSpec A, const int has_agg) {
  const u64 total = T.n_slots * (u64)T.slot_words;
  const int cell0 = 1 + T.key_words;
  for (u64 i = (u64)blockIdx.x * HBLOCK + threadIdx.x; i < total; i += (u64)gridDim.x * HBLOCK) {
    const int w = (int)(i % (u64)T.slot_words);
    u64 v = 0;
    if (has_agg && w >= cell0) {
      const int a = (w - cell0) >> 1, half = (w - cell0) & 1;
      if (a < A.n_accs) {
        switch (A.acc_kind[a]) {
          case ACC_MIN: v = half ? 0 : 0x7FFFFFFFFFFFFFFFull; break;
          case ACC_MAX: v = half ? ~0ull : 0x8000000000000000ull; break;
          case ACC_FMIN: v = half ? 0 : 0x7FF0000000000000ull; break;
          case ACC_FMAX: v = half ? 0 : 0xFFF0000000000000ull; break;
          default: break;
        }
      }
    }
    T.slots[i] = v;
  }
}
#endif

// ------------------------------------------------------------------ hash aggregate
// Rows-in-flight driver for the one-row-per-lane kernels (hash aggregate, key range, build, generic unique probe): a wave takes ROWS_U of its 64-row words per step.  With
// the generated evaluator its three stages are used so that every column load of all ROWS_U words is issued before any loaded
// value is looked at (one memory round trip per step instead of one per word: vmcnt is in order, and the one-shot evaluator waits
// for its record before the next word's loads can be issued); with the interpreter the words are simply taken one after the other.
// body(w, pos, active, regs...) runs wave-uniformly once per word (active = row exists and passes the fused predicate).
#ifndef GPUQ_ROWS_U
#define GPUQ_ROWS_U 2
#endif
template <int MAXC, class Body>
__device__ __forceinline__ void for_rows_in_flight(const DevProgram& P, const i64 n, const i64 w_first, const i64 w_stride, Body body, const i64 w_end = -1) {
  const i64 nw_all = (n + 63) >> 6;
  const i64 nwords = (w_end >= 0 && w_end < nw_all) ? w_end : nw_all;      // words [w_first, nwords) in steps of w_stride
#ifdef GPUQ_JIT
  constexpr int U = GPUQ_ROWS_U;
  for (i64 w0 = w_first; w0 < nwords; w0 += w_stride * U) {
    JitPre jq[U]; JitRaw jw[U]; i64 posc[U]; bool ex[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const i64 w = w0 + (i64)u * w_stride;
      const i64 pos = (w << 6) + hlane();
      ex[u] = w < nwords && pos < n;
      posc[u] = ex[u] ? pos : n - 1;      // clamped, masked afterwards: no exec-masked load regions (n > 0 here: nwords > 0)
      gpuq_jit_pre(P, posc[u], jq[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) gpuq_jit_load(P, posc[u], jq[u], jw[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const i64 w = w0 + (i64)u * w_stride;
      if (w >= nwords) break;             // wave-uniform
      GPUQ_REGS_DECL;
      const bool pass = gpuq_jit_compute(P, posc[u], jw[u], GPUQ_REGS);
      body(w, (w << 6) + hlane(), ex[u] && pass, GPUQ_REGS);
    }
  }
#else
  for (i64 w = w_first; w < nwords; w += w_stride) {
    const i64 pos = (w << 6) + hlane();
    bool active = pos < n;
    GPUQ_REGS_DECL;
    if (active) active = GPUQ_EVAL(MAXC, P, pos);
    body(w, pos, active, GPUQ_REGS);
  }
#endif
}

template <int MAXC>
__device__ __forceinline__ void k_agg_hash_body(const DevProgram P, const i64 n_arg, const KeySpec K, const AggSpec A, const HashTable T) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  const i64 nwords = (n + 63) >> 6;
  const int cell0 = 1 + T.key_words;
  // Rows with equal keys in neighbouring lanes (clustered input: lineitem rows of one order, a join's probe-ordered
  // output) are combined inside the wave by a segmented scan; only the last lane of each run touches the table.  Every
  // table access is a device-scope transaction (~10 G/s on this part), so a run of r rows costs 1/r of the traffic.
  bool combine = true;
  for (int a = 0; a < A.n_accs; ++a) combine = combine && A.acc_kind[a] != ACC_FMIN && A.acc_kind[a] != ACC_FMAX;
  const int lane = hlane();
  for_rows_in_flight<MAXC>(P, n, (i64)blockIdx.x * HWAVES + hwave(), (i64)gridDim.x * HWAVES, [&](const i64, const i64, const bool active, GPUQ_REGS_PARAM) {
    if (__ballot(active) == 0) return;
    u64 kw[MAX_KW]; u64 h = 0;
#pragma unroll
    for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
    if (active) make_key(K, GPUQ_REGS, kw, h);
    // head[i]: lane i starts a run (its key differs from lane i-1's, or lane i-1 is not an active row)
    // every cross-lane read below is executed by ALL lanes and only then combined: a shuffle under a short-circuit
    // (`lane > 0 && shfl(..)`) runs with some lanes masked off, and a masked-off source lane reads as 0
    bool head = true;
    if (combine) {
      const int pa = __shfl_up((int)active, 1);
      const u64 ph = __shfl_up(h, 1);
      bool eq = (pa != 0) & (lane > 0) & (ph == h);
#pragma unroll
      for (int q = 0; q < MAX_KW; ++q) { const u64 pk = __shfl_up(kw[q], 1); if (q < T.key_words) eq = eq & (pk == kw[q]); }
      head = !(active & eq);
    }
    const int nh = __shfl_down((int)head, 1);
    const bool next_head = (lane == 63) | (nh != 0);
    const bool tail = active & next_head;
    u64 s = ~0ull;
    if (tail) {
      bool inserted;
      s = ht_find_or_insert(T, kw, h, 0u, inserted);
      if (s == ~0ull) atomicOr(P.flags, FLAG_TABLE_FULL);
    }
    u64* cells = T.slots + (s == ~0ull ? 0 : s) * (u64)T.slot_words + cell0;
    const bool upd = tail && s != ~0ull;
    for (int a = 0; a < A.n_accs; ++a) {
      const int kind = A.acc_kind[a];
      u64 vlo = 1, vhi = 0; bool vnull = false;
      if (kind != ACC_COUNT_STAR) {
        const int r = __builtin_amdgcn_readfirstlane(A.acc_reg[a]);
        vlo = rlo[r]; vhi = rhi[r]; vnull = (rnulls >> r) & 1;
      }
      if (!active) vnull = true;
      bool wide = false;      // MIN/MAX over a value outside int64
      // identity for rows that do not contribute
      switch (kind) {
        case ACC_COUNT: case ACC_COUNT_STAR: vlo = vnull ? 0 : 1; vhi = 0; break;
        case ACC_SUM: if (vnull) { vlo = 0; vhi = 0; } break;
        case ACC_MIN: wide = !vnull && (i64)vhi != ((i64)vlo >> 63); if (vnull) vlo = 0x7FFFFFFFFFFFFFFFull; break;
        case ACC_MAX: wide = !vnull && (i64)vhi != ((i64)vlo >> 63); if (vnull) vlo = 0x8000000000000000ull; break;
        case ACC_FSUM: if (vnull) vlo = 0; break;
        default: break;
      }
      if (__ballot(wide)) { if (wide) atomicOr(P.flags, FLAG_WIDE_MINMAX); continue; }
      bool any = !vnull;        // does the run hold at least one contributing row
      if (combine) {
        bool f = head;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const u64 ulo = __shfl_up(vlo, off), uhi = __shfl_up(vhi, off);
          const bool uf = __shfl_up((int)f, off) != 0, uany = __shfl_up((int)any, off) != 0;
          if (lane >= off && !f) {
            switch (kind) {
              case ACC_COUNT: case ACC_COUNT_STAR: vlo += ulo; break;
              case ACC_SUM: { const u64 sl = vlo + ulo; vhi = vhi + uhi + (sl < vlo ? 1 : 0); vlo = sl; break; }
              case ACC_MIN: if ((i64)ulo < (i64)vlo) vlo = ulo; break;
              case ACC_MAX: if ((i64)ulo > (i64)vlo) vlo = ulo; break;
              case ACC_FSUM: vlo = (u64)__double_as_longlong(__longlong_as_double((i64)ulo) + __longlong_as_double((i64)vlo)); break;
              default: break;
            }
            any = any || uany;
            f = uf;
          }
        }
      }
      if (!upd || !any) continue;
      u64* c = cells + 2 * a;
      switch (kind) {
        case ACC_COUNT: case ACC_COUNT_STAR: a_add(c, vlo); break;
        case ACC_SUM: {
          const u64 old = a_add(c, vlo);
          const u64 carry = (old + vlo < old) ? 1 : 0;
          if (vhi + carry) a_add(c + 1, vhi + carry);
          break;
        }
        case ACC_MIN: __hip_atomic_fetch_min((i64*)c, (i64)vlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case ACC_MAX: __hip_atomic_fetch_max((i64*)c, (i64)vlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case ACC_FSUM: unsafeAtomicAdd((double*)c, __longlong_as_double((i64)vlo)); break;
        case ACC_FMIN: case ACC_FMAX: {
          // total-order min/max through a CAS loop on the bit pattern (never combined: every row is its own run)
          u64 cur = a_load(c);
          for (;;) {
            const bool better = (kind == ACC_FMIN) ? (f64_total_key(vlo) < f64_total_key(cur)) : (f64_total_key(vlo) > f64_total_key(cur));
            if (!better) break;
            const u64 seen = a_cas(c, cur, vlo);
            if (seen == cur) break;
            cur = seen;
          }
          break;
        }
        default: break;
      }
    }
  });
}
#ifndef GPUQ_JIT
template <int MAXC>
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(HBLOCK) k_agg_hash(const DevProgram P, const i64 n, const KeySpec K, const AggSpec A, const HashTable T) { k_agg_hash_body<MAXC>(P, n, K, A, T); }
#endif
#elif GPUQ_JIT_KERNEL == 4
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const AggSpec A, const HashTable T) { k_agg_hash_body<0>(P, n, K, A, T); }
#endif

#ifndef GPUQ_JIT
__global__ void __launch_bounds__(HBLOCK) k_agg_hash_extract(const KeySpec K, const AggSpec A, const HashTable T, const AggOut out,
                                                             uint32_t* __restrict__ flags) {
  // A wave claims output rows for XSUB x 64 slots with ONE atomic on the group counter (one atomic per 64 slots made a
  // 2^26-slot table cost 18 ms: ~17 ns per serialised device-scope atomic).  Output order is arbitrary anyway.
  constexpr int XSUB = 16;
  const int cell0 = 1 + T.key_words;
  const int kstride = K.n_keys > 0 ? K.n_keys : 1;
  const u64 chunk_slots = 64ull * XSUB;
  const u64 nchunks = (T.n_slots + chunk_slots - 1) / chunk_slots;
  const u64 wave0 = (u64)blockIdx.x * HWAVES + hwave(), nwaves = (u64)gridDim.x * HWAVES;
  const u64 ltmask = (1ull << hlane()) - 1;
  for (u64 c = wave0; c < nchunks; c += nwaves) {
    u64 masks[XSUB]; uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < XSUB; ++j) {
      const u64 s = c * chunk_slots + (u64)j * 64 + hlane();
      const bool live = s < T.n_slots && ((uint32_t)T.slots[s * (u64)T.slot_words]) >= 2u;
      masks[j] = __ballot(live); total += (uint32_t)__popcll(masks[j]);
    }
    if (total == 0) continue;
    uint32_t base = 0;
    if (hlane() == 0) base = atomicAdd(out.n_groups, total);
    base = __shfl(base, 0);
    if (out.cap == 0) continue;           // counting pass (the caller sizes the result from n_groups): nothing to write, nothing to flag
    if (base + total > (uint32_t)out.cap && hlane() == 0) atomicOr(flags, FLAG_GROUP_OVERFLOW);      // once per wave, not once per row
#pragma unroll
    for (int j = 0; j < XSUB; ++j) {
      const u64 m = masks[j];
      if ((m >> hlane()) & 1) {
        const u64 s = c * chunk_slots + (u64)j * 64 + hlane();
        const u64* slot = T.slots + s * (u64)T.slot_words;
        const uint32_t g = base + (uint32_t)__popcll(m & ltmask);
        if (g < (uint32_t)out.cap) {
          int w = 0;
          for (int k = 0; k < K.n_keys; ++k) {
            const u64 lo = slot[1 + w]; ++w;
            u64 hi = (u64)((i64)lo >> 63);
            if (K.key_wide[k]) { hi = slot[1 + w]; ++w; }
            out.keys[((size_t)g * kstride + k) * 2] = lo;
            out.keys[((size_t)g * kstride + k) * 2 + 1] = hi;
          }
          out.key_nulls[g] = K.null_word ? (uint32_t)slot[1 + w] : 0u;
          for (int a = 0; a < A.n_accs; ++a) {
            u64 lo = slot[cell0 + 2 * a], hi = slot[cell0 + 2 * a + 1];
            const int kind = A.acc_kind[a];
            if (kind == ACC_MIN || kind == ACC_MAX) hi = (u64)((i64)lo >> 63);
            out.cells[((size_t)g * A.n_accs + a) * 2] = lo;
            out.cells[((size_t)g * A.n_accs + a) * 2 + 1] = hi;
          }
        }
      }
      base += (uint32_t)__popcll(m);
    }
  }
}
#endif

// ------------------------------------------------------------------ radix-partitioned aggregate (high cardinality)
// The global-table aggregate is bound by device-scope transactions (~10 G/s: every probe, CAS and accumulate of every row
// is one).  For tens of millions of groups the rows are first partitioned by key hash into buckets small enough for an LDS
// hash table (bucket id + row id through the coalesced radix passes of kernels_sort.hip); one block then aggregates a bucket
// entirely in LDS -- workgroup-scope atomics only -- and appends its groups to the result.  Three kernels:
//   k_agg_bucket_id     rows -> (bucket = hash & mask, row id | NIL for rows the predicate drops)
//   k_bucket_bounds     lower bound of every bucket in the sorted bucket-id array
//   k_agg_bucket        per bucket: LDS find-or-insert + accumulate over the bucket's rows (re-evaluated by row id), extract
template <int MAXC>
__device__ __forceinline__ void k_agg_bucket_id_body(const DevProgram P, const i64 n_arg, const KeySpec K, const u64 bucket_mask,
                                                         u64* __restrict__ bid, uint32_t* __restrict__ ids) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  const i64 nwords = (n + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * HWAVES + hwave(); w < nwords; w += (i64)gridDim.x * HWAVES) {
    const i64 pos = (w << 6) + hlane();
    if (pos >= n) continue;
    GPUQ_REGS_DECL;
    const bool pass = GPUQ_EVAL(MAXC, P, pos);
    u64 kw[MAX_KW]; u64 h = 0;
#pragma unroll
    for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
    if (pass) make_key(K, GPUQ_REGS, kw, h);
    if (!ids) bid[pos] = ((h & bucket_mask) << 32) | (u64)(pass ? (uint32_t)pos : NIL);      // one 8-byte (bucket, row) record
    else { bid[pos] = h & bucket_mask; ids[pos] = pass ? (uint32_t)pos : NIL; }
  }
}
#ifndef GPUQ_JIT
template <int MAXC>
__global__ void __launch_bounds__(HBLOCK) k_agg_bucket_id(const DevProgram P, const i64 n, const KeySpec K, const u64 bucket_mask,
                                                          u64* __restrict__ bid, uint32_t* __restrict__ ids) { k_agg_bucket_id_body<MAXC>(P, n, K, bucket_mask, bid, ids); }
__global__ void __launch_bounds__(HBLOCK) k_bucket_bounds(const u64* __restrict__ sorted_bid, const i64 n, const u64 nbuckets, uint32_t* __restrict__ bounds, const int shift) {
  for (u64 b = (u64)blockIdx.x * HBLOCK + threadIdx.x; b <= nbuckets; b += (u64)gridDim.x * HBLOCK) {
    i64 lo = 0, hi = n;                     // first position whose bucket id is >= b
    while (lo < hi) { const i64 mid = (lo + hi) >> 1; if ((sorted_bid[mid] >> shift) < b) lo = mid + 1; else hi = mid; }
    bounds[b] = (uint32_t)lo;
  }
}
#elif GPUQ_JIT_KERNEL == 11
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const u64 bucket_mask,
                                                          u64* __restrict__ bid, uint32_t* __restrict__ ids) { k_agg_bucket_id_body<0>(P, n, K, bucket_mask, bid, ids); }
#endif

// LDS table slot: word 0 = state (0 empty, 1 being written, else tag), then key words, then 2 words per accumulator
__device__ __forceinline__ uint32_t lds_slot_find_or_insert(u64* slots, const uint32_t cap, const int slot_words, const int key_words, const u64 (&kw)[MAX_KW], const u64 h) {
  const uint32_t mask = cap - 1;
  uint32_t s = (uint32_t)(h >> 20) & mask;          // bits the bucket id did not use
  const u64 tag = (u64)tag_of(h);
  for (uint32_t probes = 0; probes < cap;) {
    u64* slot = slots + (size_t)s * slot_words;
    u64 st = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (st == 0) {
      u64 expected = 0;
      if (__hip_atomic_compare_exchange_strong(slot, &expected, 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
        for (int q = 0; q < MAX_KW; ++q) if (q < key_words) __hip_atomic_store(slot + 1 + q, kw[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __hip_atomic_store(slot, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return s;
      }
      st = expected;
    }
    if (st == 1) continue;                           // being published by another lane: look again
    if (st == tag) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      bool eq = true;
#pragma unroll
      for (int q = 0; q < MAX_KW; ++q) if (q < key_words) eq = eq && (__hip_atomic_load(slot + 1 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kw[q]);
      if (eq) return s;
    }
    s = (s + 1) & mask; ++probes;
  }
  return 0xFFFFFFFFu;
}

// ------------------------------------------------------------------ cardinality estimate
// Which aggregate kernel is right depends on the number of groups, which nobody tells a first run.  A strided sample of the
// input (every stride-th row, so clustered keys do not fool it) is hashed into a bitmap (linear counting: with far more bits
// than samples the popcount IS the number of distinct keys seen); `passed` counts the sampled rows the predicate kept.
#ifndef GPUQ_JIT
template <int MAXC>
__global__ void __launch_bounds__(HBLOCK) k_key_sample(const DevProgram P, const i64 n, const KeySpec K, const i64 stride, const i64 nsample,
                                                       unsigned int* __restrict__ bitmap, const u64 bit_mask, unsigned long long* __restrict__ passed) {
  __shared__ unsigned int block_passed;
  if (threadIdx.x == 0) block_passed = 0;
  __syncthreads();
  const i64 nwords = (nsample + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * HWAVES + hwave(); w < nwords; w += (i64)gridDim.x * HWAVES) {
    const i64 i = (w << 6) + hlane();
    i64 pos = i * stride; if (pos >= n) pos = n - 1;
    bool active = i < nsample;
    GPUQ_REGS_DECL;
    if (active) active = GPUQ_EVAL(MAXC, P, pos);
    const u64 m = __ballot(active);
    if (m && hlane() == 0) atomicAdd(&block_passed, (unsigned int)__popcll(m));
    if (!active) continue;
    u64 kw[MAX_KW]; u64 h = 0;
#pragma unroll
    for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
    make_key(K, GPUQ_REGS, kw, h);
    const u64 bit = mix64(h) & bit_mask;
    // few groups = a million samples on a handful of words: look before the atomic (q1's 4 groups: 5.9 ms of same-address
    // atomics without the check)
    const unsigned int want = 1u << (bit & 31);
    if (!(__hip_atomic_load(bitmap + (bit >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & want)) atomicOr(bitmap + (bit >> 5), want);
  }
  __syncthreads();
  if (threadIdx.x == 0 && block_passed) atomicAdd(passed, (unsigned long long)block_passed);
}
#endif

// ------------------------------------------------------------------ block-local pre-aggregation (medium cardinality)
// A group-by with a few hundred to a few thousand groups (too many for k_agg_tiny's register-cached dictionary, few enough
// that every row of a 10 M-row input hits one of a handful of hot slots) makes the global table a contention point: device-
// scope atomics run at ~10 G/s spread over addresses, far less on a hundred hot ones (h2o q1: 14 ms for 10 M rows; three
// float sums per row: 540 ms).  Here every block folds its rows into an LDS table first (LDS atomics are per-CU) and only
// the table's entries go to the global table: #blocks x #groups global operations instead of #rows.  LDS slot = [state/tag,
// hash, key words, 2 words per accumulator]; a block whose table runs full sends its remaining rows straight to the global
// table, so the kernel is correct for any cardinality -- the host only picks it when the group count is known to be small.
constexpr uint32_t LDS_PROBES = 32;
__device__ __forceinline__ uint32_t lds_group_slot(u64* slots, const uint32_t cap, const int lw, const int key_words, const u64 (&kw)[MAX_KW], const u64 h) {
  const uint32_t mask = cap - 1;
  uint32_t s = (uint32_t)(h >> 17) & mask;
  const u64 tag = (u64)tag_of(h);
  for (uint32_t probes = 0; probes < LDS_PROBES;) {
    u64* slot = slots + (size_t)s * lw;
    u64 st = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (st == 0) {
      u64 expected = 0;
      if (__hip_atomic_compare_exchange_strong(slot, &expected, 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        __hip_atomic_store(slot + 1, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int q = 0; q < MAX_KW; ++q) if (q < key_words) __hip_atomic_store(slot + 2 + q, kw[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __hip_atomic_store(slot, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return s;
      }
      st = expected;
    }
    if (st == 1) continue;                           // being published by another lane: look again
    if (st == tag) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      bool eq = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == h;
#pragma unroll
      for (int q = 0; q < MAX_KW; ++q) if (q < key_words) eq = eq && (__hip_atomic_load(slot + 2 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kw[q]);
      if (eq) return s;
    }
    s = (s + 1) & mask; ++probes;
  }
  return NIL;
}
// one accumulator value folded into a GLOBAL cell (device-scope atomics); FMIN/FMAX through a CAS loop on the total order
__device__ __forceinline__ void global_fold(u64* c, const int kind, const u64 vlo, const u64 vhi) {
  switch (kind) {
    case ACC_COUNT: case ACC_COUNT_STAR: if (vlo) a_add(c, vlo); break;
    case ACC_SUM: {
      if (!(vlo | vhi)) break;
      const u64 old = a_add(c, vlo);
      const u64 carry = (old + vlo < old) ? 1 : 0;
      if (vhi + carry) a_add(c + 1, vhi + carry);
      break;
    }
    case ACC_MIN: __hip_atomic_fetch_min((i64*)c, (i64)vlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
    case ACC_MAX: __hip_atomic_fetch_max((i64*)c, (i64)vlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
    case ACC_FSUM: unsafeAtomicAdd((double*)c, __longlong_as_double((i64)vlo)); break;
    case ACC_FMIN: case ACC_FMAX: {
      u64 cur = a_load(c);
      for (;;) {
        const bool better = (kind == ACC_FMIN) ? (f64_total_key(vlo) < f64_total_key(cur)) : (f64_total_key(vlo) > f64_total_key(cur));
        if (!better) break;
        const u64 seen = a_cas(c, cur, vlo);
        if (seen == cur) break;
        cur = seen;
      }
      break;
    }
    default: break;
  }
}

template <int MAXC>
__device__ __forceinline__ void k_agg_lds_body(const DevProgram P, const i64 n_arg, const KeySpec K, const AggSpec A, const HashTable T, const uint32_t lcap,
                                               u64* __restrict__ fstage, const int n_fsum) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  extern __shared__ __attribute__((aligned(16))) u64 lslots[];
  __shared__ uint32_t lfull;
  const int key_words = K.key_words;
  const int lw = T.slot_words + 1, lcell0 = 2 + key_words, gcell0 = 1 + key_words;
  for (uint32_t i = threadIdx.x; i < lcap * (uint32_t)lw; i += HBLOCK) {
    const int w = (int)(i % (uint32_t)lw);
    u64 v = 0;
    if (w >= lcell0) {
      const int a = (w - lcell0) >> 1, half = (w - lcell0) & 1;
      if (a < A.n_accs) {
        switch (A.acc_kind[a]) {
          case ACC_MIN: v = half ? 0 : 0x7FFFFFFFFFFFFFFFull; break;
          case ACC_MAX: v = half ? ~0ull : 0x8000000000000000ull; break;
          case ACC_FMIN: v = half ? 0 : 0x7FF0000000000000ull; break;
          case ACC_FMAX: v = half ? 0 : 0xFFF0000000000000ull; break;
          default: break;
        }
      }
    }
    lslots[i] = v;
  }
  if (threadIdx.x == 0) lfull = 0;
  __syncthreads();
  const i64 nwords = (n + 63) >> 6;
  for (i64 w = (i64)blockIdx.x * HWAVES + hwave(); w < nwords; w += (i64)gridDim.x * HWAVES) {
    const i64 pos = (w << 6) + hlane();
    bool active = pos < n;
    GPUQ_REGS_DECL;
    if (active) active = GPUQ_EVAL(MAXC, P, pos);
    if (!active) continue;
    u64 kw[MAX_KW]; u64 h = 0;
#pragma unroll
    for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
    make_key(K, GPUQ_REGS, kw, h);
    uint32_t ls = NIL;
    if (!__hip_atomic_load(&lfull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
      ls = lds_group_slot(lslots, lcap, lw, key_words, kw, h);
      if (ls == NIL) __hip_atomic_store(&lfull, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    u64* gcells = nullptr;
    if (ls == NIL) {                                  // this block's table is full: the row goes to the global table directly
      bool inserted;
      const u64 gs = ht_find_or_insert(T, kw, h, 0u, inserted);
      if (gs == ~0ull) { atomicOr(P.flags, FLAG_TABLE_FULL); continue; }
      gcells = T.slots + gs * (u64)T.slot_words + gcell0;
    }
    u64* cells = lslots + (size_t)(ls == NIL ? 0 : ls) * lw + lcell0;
    for (int a = 0; a < A.n_accs; ++a) {
      const int kind = A.acc_kind[a];
      u64 vlo = 1, vhi = 0; bool vnull = false;
      if (kind != ACC_COUNT_STAR) {
        const int r = __builtin_amdgcn_readfirstlane(A.acc_reg[a]);
        vlo = rlo[r]; vhi = rhi[r]; vnull = (rnulls >> r) & 1;
      }
      if (vnull) continue;
      if (kind == ACC_COUNT) { vlo = 1; vhi = 0; }
      if ((kind == ACC_MIN || kind == ACC_MAX) && (i64)vhi != ((i64)vlo >> 63)) { atomicOr(P.flags, FLAG_WIDE_MINMAX); continue; }
      if (gcells) { global_fold(gcells + 2 * a, kind, vlo, vhi); continue; }
      u64* c = cells + 2 * a;
      switch (kind) {
        case ACC_COUNT: case ACC_COUNT_STAR: atomicAdd((unsigned long long*)c, 1ull); break;
        case ACC_SUM: {
          const u64 old = atomicAdd((unsigned long long*)c, (unsigned long long)vlo);
          const u64 carry = (old + vlo < old) ? 1 : 0;
          if (vhi + carry) atomicAdd((unsigned long long*)(c + 1), (unsigned long long)(vhi + carry));
          break;
        }
        case ACC_MIN: atomicMin((long long*)c, (long long)vlo); break;
        case ACC_MAX: atomicMax((long long*)c, (long long)vlo); break;
        case ACC_FSUM: unsafeAtomicAdd((double*)c, __longlong_as_double((i64)vlo)); break;      // (a CAS loop or a fetch_add cost the same: measured)
        case ACC_FMIN: case ACC_FMAX: {
          u64 cur = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          for (;;) {
            const bool better = (kind == ACC_FMIN) ? (f64_total_key(vlo) < f64_total_key(cur)) : (f64_total_key(vlo) > f64_total_key(cur));
            if (!better) break;
            const u64 seen = atomicCAS((unsigned long long*)c, (unsigned long long)cur, (unsigned long long)vlo);
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
  // flush: every live LDS slot is folded into the global table
  for (uint32_t sl = threadIdx.x; sl < lcap; sl += HBLOCK) {
    const u64* slot = lslots + (size_t)sl * lw;
    if (slot[0] < 2) continue;
    u64 kw[MAX_KW];
#pragma unroll
    for (int q = 0; q < MAX_KW; ++q) kw[q] = (q < key_words) ? slot[2 + q] : 0;
    bool inserted;
    const u64 gs = ht_find_or_insert(T, kw, slot[1], 0u, inserted);
    if (gs == ~0ull) { atomicOr(P.flags, FLAG_TABLE_FULL); continue; }
    u64* g = T.slots + gs * (u64)T.slot_words + gcell0;
    int kf = 0;
    for (int a = 0; a < A.n_accs; ++a) {
      const int kind = A.acc_kind[a];
      // A float partial sum does not go to the table cell: thousands of blocks adding doubles to the same hundred addresses
      // serialise at 2-7 us per add (that alone was 6 of q4's 6.5 ms).  It is parked in its own word [slot][sum][block]
      // and k_fsum_stage_reduce adds the blocks up in block order afterwards -- which also makes the sum reproducible.
      if (kind == ACC_FSUM && fstage) { fstage[((gs * (u64)n_fsum + (u64)kf) * gridDim.x) + blockIdx.x] = slot[lcell0 + 2 * a]; ++kf; continue; }
      if (kind == ACC_FSUM) ++kf;
      global_fold(g + 2 * a, kind, slot[lcell0 + 2 * a], slot[lcell0 + 2 * a + 1]);
    }
  }
}
#ifndef GPUQ_JIT
// one wave per (table slot, float sum): the blocks' partial sums in block order
__global__ void __launch_bounds__(HBLOCK) k_fsum_stage_reduce(const HashTable T, const AggSpec A, const u64* __restrict__ fstage, const int n_fsum, const uint32_t nblk) {
  const int gcell0 = 1 + T.key_words;
  const u64 items = T.n_slots * (u64)n_fsum;
  for (u64 it = (u64)blockIdx.x * HWAVES + hwave(); it < items; it += (u64)gridDim.x * HWAVES) {
    const u64 gs = it / (u64)n_fsum; const int kf = (int)(it % (u64)n_fsum);
    if ((uint32_t)T.slots[gs * (u64)T.slot_words] < 2u) continue;
    const u64* src = fstage + it * nblk;
    double acc = 0.0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 64) {
      const uint32_t b = b0 + hlane();
      double v = b < nblk ? __longlong_as_double((i64)src[b]) : 0.0;
      // fixed tree inside the 64 blocks, then in order across groups of 64
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      acc += v;
    }
    if (hlane() == 0) {
      int k = 0, a = 0;
      for (; a < A.n_accs; ++a) if (A.acc_kind[a] == ACC_FSUM) { if (k == kf) break; ++k; }
      double* cell = (double*)(T.slots + gs * (u64)T.slot_words + gcell0 + 2 * a);
      *cell += acc;          // rows that bypassed a full LDS table have already added theirs here
    }
  }
}
#endif
#ifndef GPUQ_JIT
template <int MAXC>
__global__ void __launch_bounds__(HBLOCK) k_agg_lds(const DevProgram P, const i64 n, const KeySpec K, const AggSpec A, const HashTable T, const uint32_t lcap,
                                                    u64* fstage, const int n_fsum) { k_agg_lds_body<MAXC>(P, n, K, A, T, lcap, fstage, n_fsum); }
#elif GPUQ_JIT_KERNEL == 13
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const AggSpec A, const HashTable T, const uint32_t lcap,
                                                                     u64* fstage, const int n_fsum) { k_agg_lds_body<0>(P, n, K, A, T, lcap, fstage, n_fsum); }
#endif

template <int MAXC>
__device__ __forceinline__ void k_agg_bucket_body(const DevProgram P, const KeySpec K, const AggSpec A, const uint32_t* __restrict__ ids,
                                                      const uint32_t* __restrict__ bounds, const uint32_t nbuckets, const uint32_t cap, const int slot_words,
                                                      const AggOut out) {
  extern __shared__ __attribute__((aligned(16))) u64 bslots[];
  __shared__ uint32_t cnt[2]; __shared__ uint32_t gbase;
  const int key_words = K.key_words;
  const int cell0 = 1 + key_words;
  const int kstride = K.n_keys > 0 ? K.n_keys : 1;
  for (uint32_t bk = blockIdx.x; bk < nbuckets; bk += gridDim.x) {
    const uint32_t b0 = bounds[bk], b1 = bounds[bk + 1];
    if (b0 == b1) continue;                          // uniform for the block
    // table init
    for (uint32_t i = threadIdx.x; i < cap * (uint32_t)slot_words; i += HBLOCK) {
      const int w = (int)(i % (uint32_t)slot_words);
      u64 v = 0;
      if (w >= cell0) {
        const int a = (w - cell0) >> 1, half = (w - cell0) & 1;
        if (a < A.n_accs) {
          switch (A.acc_kind[a]) {
            case ACC_MIN: v = half ? 0 : 0x7FFFFFFFFFFFFFFFull; break;
            case ACC_MAX: v = half ? ~0ull : 0x8000000000000000ull; break;
            case ACC_FMIN: v = half ? 0 : 0x7FF0000000000000ull; break;
            case ACC_FMAX: v = half ? 0 : 0xFFF0000000000000ull; break;
            default: break;
          }
        }
      }
      bslots[i] = v;
    }
    if (threadIdx.x == 0) { cnt[0] = 0; cnt[1] = 0; }
    __syncthreads();
    // aggregate the bucket's rows
    for (uint32_t i0 = b0; i0 < b1; i0 += HBLOCK) {
      const uint32_t i = i0 + threadIdx.x;
      uint32_t row = NIL;
      if (i < b1) row = ids[i];
      bool active = row != NIL;
      GPUQ_REGS_DECL;
      if (active) active = GPUQ_EVAL(MAXC, P, (i64)row);
      if (!active) continue;
      u64 kw[MAX_KW]; u64 h;
#pragma unroll
      for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
      make_key(K, GPUQ_REGS, kw, h);
      const uint32_t s = lds_slot_find_or_insert(bslots, cap, slot_words, key_words, kw, h);
      if (s == 0xFFFFFFFFu) { atomicOr(P.flags, FLAG_TABLE_FULL); continue; }
      u64* cells = bslots + (size_t)s * slot_words + cell0;
      for (int a = 0; a < A.n_accs; ++a) {
        const int kind = A.acc_kind[a];
        u64 vlo = 1, vhi = 0; bool vnull = false;
        if (kind != ACC_COUNT_STAR) {
          const int r = __builtin_amdgcn_readfirstlane(A.acc_reg[a]);
          vlo = rlo[r]; vhi = rhi[r]; vnull = (rnulls >> r) & 1;
        }
        if (vnull) continue;
        u64* c = cells + 2 * a;
        switch (kind) {
          case ACC_COUNT: case ACC_COUNT_STAR: atomicAdd((unsigned long long*)c, 1ull); break;
          case ACC_SUM: {
            const u64 old = atomicAdd((unsigned long long*)c, (unsigned long long)vlo);
            const u64 carry = (old + vlo < old) ? 1 : 0;
            if (vhi + carry) atomicAdd((unsigned long long*)(c + 1), (unsigned long long)(vhi + carry));
            break;
          }
          case ACC_MIN: case ACC_MAX: {
            if ((i64)vhi != ((i64)vlo >> 63)) { atomicOr(P.flags, FLAG_WIDE_MINMAX); break; }
            if (kind == ACC_MIN) atomicMin((long long*)c, (long long)vlo); else atomicMax((long long*)c, (long long)vlo);
            break;
          }
          case ACC_FSUM: unsafeAtomicAdd((double*)c, __longlong_as_double((i64)vlo)); break;
          case ACC_FMIN: case ACC_FMAX: {
            u64 cur = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (;;) {
              const bool better = (kind == ACC_FMIN) ? (f64_total_key(vlo) < f64_total_key(cur)) : (f64_total_key(vlo) > f64_total_key(cur));
              if (!better) break;
              const u64 seen = atomicCAS((unsigned long long*)c, (unsigned long long)cur, (unsigned long long)vlo);
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
    // extract: count, claim a range of the result with one device atomic per bucket, write
    for (int pass = 0; pass < 2; ++pass) {
      for (uint32_t s0 = 0; s0 < cap; s0 += HBLOCK) {
        const uint32_t s = s0 + threadIdx.x;
        const u64* slot = bslots + (size_t)s * slot_words;
        const bool live = s < cap && (uint32_t)slot[0] >= 2u;
        const u64 m = __ballot(live);
        uint32_t wbase = 0;
        if (hlane() == 0 && m) wbase = atomicAdd(&cnt[pass], (uint32_t)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (pass == 1 && live) {
          const uint32_t g = gbase + wbase + (uint32_t)__popcll(m & ((1ull << hlane()) - 1));
          if (g >= (uint32_t)out.cap) atomicOr(P.flags, FLAG_GROUP_OVERFLOW);
          else {
            int w = 0;
            for (int k = 0; k < K.n_keys; ++k) {
              const u64 lo = slot[1 + w]; ++w;
              u64 hi = (u64)((i64)lo >> 63);
              if (K.key_wide[k]) { hi = slot[1 + w]; ++w; }
              out.keys[((size_t)g * kstride + k) * 2] = lo; out.keys[((size_t)g * kstride + k) * 2 + 1] = hi;
            }
            out.key_nulls[g] = K.null_word ? (uint32_t)slot[1 + w] : 0u;
            for (int a = 0; a < A.n_accs; ++a) {
              u64 lo = slot[cell0 + 2 * a], hi = slot[cell0 + 2 * a + 1];
              const int kind = A.acc_kind[a];
              if (kind == ACC_MIN || kind == ACC_MAX) hi = (u64)((i64)lo >> 63);
              out.cells[((size_t)g * A.n_accs + a) * 2] = lo; out.cells[((size_t)g * A.n_accs + a) * 2 + 1] = hi;
            }
          }
        }
      }
      __syncthreads();
      if (pass == 0) {
        if (threadIdx.x == 0) gbase = atomicAdd(out.n_groups, cnt[0]);
        __syncthreads();
      }
    }
  }
}
#ifndef GPUQ_JIT
template <int MAXC>
__global__ void __launch_bounds__(HBLOCK) k_agg_bucket(const DevProgram P, const KeySpec K, const AggSpec A, const uint32_t* __restrict__ ids,
                                                       const uint32_t* __restrict__ bounds, const uint32_t nbuckets, const uint32_t cap, const int slot_words,
                                                       const AggOut out) { k_agg_bucket_body<MAXC>(P, K, A, ids, bounds, nbuckets, cap, slot_words, out); }
#elif GPUQ_JIT_KERNEL == 12
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const KeySpec K, const AggSpec A, const uint32_t* __restrict__ ids,
                                                       const uint32_t* __restrict__ bounds, const uint32_t nbuckets, const uint32_t cap, const int slot_words,
                                                       const AggOut out) { k_agg_bucket_body<0>(P, K, A, ids, bounds, nbuckets, cap, slot_words, out); }
#endif

// ------------------------------------------------------------------ join build
// Key range of the rows the build would insert (single narrow key): decides between the direct-addressed table and
// open addressing.  out = {min, max, count}, pre-set by the host to {INT64_MAX, INT64_MIN, 0}.
template <int MAXC>
__device__ __forceinline__ void k_join_keyrange_body(const DevProgram P, const i64 n_arg, const KeySpec K, const int null_eq, u64* __restrict__ out, const i64 wstep) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  __shared__ i64 smn[HWAVES], smx[HWAVES]; __shared__ u64 scn[HWAVES];
  i64 mn = 0x7FFFFFFFFFFFFFFFll, mx = (i64)0x8000000000000000ull; u64 cn = 0;
  const int kr = __builtin_amdgcn_readfirstlane(K.key_reg[0]);
  // wstep > 1: every wstep-th 64-row word only (a sample: the host widens what comes out and the build checks every row against it)
  for_rows_in_flight<MAXC>(P, n, ((i64)blockIdx.x * HWAVES + hwave()) * wstep, (i64)gridDim.x * HWAVES * wstep, [&](const i64, const i64, const bool active, GPUQ_REGS_PARAM) {
    if (active && !((rnulls >> kr) & 1)) {
      const i64 v = (i64)rlo[kr];
      mn = v < mn ? v : mn; mx = v > mx ? v : mx; ++cn;
    }
  });
  (void)null_eq;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const i64 a = __shfl_xor(mn, o), b = __shfl_xor(mx, o); const u64 c = __shfl_xor(cn, o);
    mn = a < mn ? a : mn; mx = b > mx ? b : mx; cn += c;
  }
  if (hlane() == 0) { smn[hwave()] = mn; smx[hwave()] = mx; scn[hwave()] = cn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < HWAVES; ++k) { mn = smn[k] < mn ? smn[k] : mn; mx = smx[k] > mx ? smx[k] : mx; cn += scn[k]; }
    if (cn) { atomicMin((long long*)out, (long long)mn); atomicMax((long long*)out + 1, (long long)mx); atomicAdd((unsigned long long*)out + 2, (unsigned long long)cn); }
  }
}
#ifndef GPUQ_JIT
template <int MAXC>
__global__ void __launch_bounds__(HBLOCK) k_join_keyrange(const DevProgram P, const i64 n, const KeySpec K, const int null_eq, u64* __restrict__ out, const i64 wstep) { k_join_keyrange_body<MAXC>(P, n, K, null_eq, out, wstep); }
#elif GPUQ_JIT_KERNEL == 14
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const int null_eq, u64* __restrict__ out, const i64 wstep) { k_join_keyrange_body<0>(P, n, K, null_eq, out, wstep); }
#endif

// key lookup shared by the probes: chain head row of the key, or NIL
__device__ __forceinline__ uint32_t join_lookup(const HashTable& T, const u64 (&kw)[MAX_KW], const u64 h) {
  if (T.dense) {
    const u64 idx = kw[0] - (u64)T.dense_min;
    if (idx >= T.dense_range) return NIL;
    if (T.dense_bits && !((T.dense_bits[idx >> 5] >> (idx & 31)) & 1u)) return NIL;
    return T.dense[idx];
  }
  uint32_t payload;
  return ht_find(T, kw, h, payload) ? payload : NIL;
}

// Chain fusion (A |x| B) |x| C with A's keys unique: the build side of the second join is "the rows of B that find their key in A's
// table".  Instead of probing A with B, writing the pairs, and building from the pairs through an index vector, the BUILD kernel of
// the second join looks every row of B up in A's table (S.T, key registers S.K) and inserts the survivors: one pass over B, B's row
// position is the build row.  hit_out[pos] (optional) = A's row for the surviving position (A's columns are then read through it);
// rows_out (optional) += number of surviving rows (one atomic per wave at the end).
// one row into the direct-addressed table (idx = key - dense_min, checked by the caller)
__device__ __forceinline__ void build_insert_dense(const DevProgram& P, const HashTable& T, uint32_t* __restrict__ next, const u64 idx, const uint32_t row) {
  bool inserted; uint32_t old = NIL;
  if (T.dense_bits) {
    // presence bitmap + uninitialised row array: valid for unique keys only; a duplicate raises the flag and the host rebuilds
    // with the initialised array (chains need a defined head)
    const uint32_t bit = 1u << (idx & 31);
    const uint32_t was = atomicOr(T.dense_bits + (idx >> 5), bit);
    T.dense[idx] = row;
    inserted = !(was & bit);
  } else {
    // the table word is the chain head; an exchange both claims the key and links a duplicate
    old = __hip_atomic_exchange(T.dense + idx, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    inserted = old == NIL;
  }
  if (inserted) {
    if (next) next[row] = NIL;
  } else {
    // duplicate key: remember it (the host then uses the chained probe) -- test first, one word must not be hammered
    if (!(__hip_atomic_load(P.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FLAG_DUP_BUILD_KEY)) atomicOr(P.flags, FLAG_DUP_BUILD_KEY);
    if (next) next[row] = old;
  }
}
constexpr int BUILD_QCAP = 128;      // survivor queue per wave: < 64 waiting + <= 64 from one more word
// SEMI: compiled with the chain-fusion front (the plain build keeps its register budget: the 16-column interpreter instantiation
// spilled its register file with both key sets live)
template <int MAXC, bool SEMI>
__device__ __forceinline__ void k_join_build_body(const DevProgram P, const i64 n_arg, const KeySpec K, const HashTable T,
                                                       uint32_t* __restrict__ next, uint32_t* __restrict__ present, const int payload_via,
                                                       const int null_eq, const SemiProbe S) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  // Chain fusion over a direct-addressed table: only a fraction of the rows survives the lookup in the other join's table (SF100 q3:
  // one order in ten), and an insert is a chain of dependent memory operations (atomic on the bitmap, store of the row, the other
  // table's row for the hit vector).  Run per 64-row word, that chain is paid 2.3 M times with six lanes busy (measured: 2.27 ms for
  // the orders side, worse than the two-step form).  So the survivors of a wave are QUEUED in LDS -- (position, table index, the other
  // table's index) -- and inserted 64 at a time with every lane busy: the chain is paid once per 64 survivors.
  __shared__ uint32_t q_pos[SEMI ? HWAVES : 1][SEMI ? BUILD_QCAP : 1], q_idx[SEMI ? HWAVES : 1][SEMI ? BUILD_QCAP : 1], q_aux[SEMI ? HWAVES : 1][SEMI ? BUILD_QCAP : 1];
  const bool queued = SEMI && S.on && T.dense != nullptr;
  const bool s_bits = S.T.dense != nullptr && S.T.dense_bits != nullptr;      // the other table answers "is it there" from its bitmap alone
  uint32_t qn = 0, survivors = 0;
  const int wv = SEMI ? hwave() : 0;
  auto flush = [&](const uint32_t from, const uint32_t cnt) {      // entries [from, from + cnt) of the queue, cnt <= 64
    const bool on = (uint32_t)hlane() < cnt;
    const uint32_t j = from + (on ? (uint32_t)hlane() : 0u);
    const uint32_t pos = q_pos[wv][j], idx = q_idx[wv][j], aux = q_aux[wv][j];
    if (on) {
      if (S.hit_out) S.hit_out[pos] = s_bits ? S.T.dense[aux] : aux;
      build_insert_dense(P, T, next, (u64)idx, pos);
    }
  };
  for_rows_in_flight<MAXC>(P, n, (i64)blockIdx.x * HWAVES + hwave(), (i64)gridDim.x * HWAVES, [&](const i64 w, const i64 pos, bool active, GPUQ_REGS_PARAM) {
    uint32_t aux = NIL;
    if (SEMI && S.on) {
      bool found = false;
      if (active) {
        u64 kw1[MAX_KW]; u64 h1;
#pragma unroll
        for (int q = 0; q < MAX_KW; ++q) kw1[q] = 0;
        const bool null1 = make_key(S.K, GPUQ_REGS, kw1, h1);
        if (!(null1 && !S.null_eq)) {
          if (s_bits && queued) {
            const u64 i1 = kw1[0] - (u64)S.T.dense_min;
            if (i1 < S.T.dense_range && ((S.T.dense_bits[i1 >> 5] >> (i1 & 31)) & 1u)) { found = true; aux = (uint32_t)i1; }      // the row itself is read at the flush
          } else { aux = join_lookup(S.T, kw1, h1); found = aux != NIL; }
        }
      }
      active = active && found;
      if (!queued && active && S.hit_out) S.hit_out[pos] = aux;
      if (S.rows_out) survivors += active ? 1u : 0u;
    }
    // `present` = every build-side row that passes the side's predicate, NULL keys included (outer joins emit them).  When rows
    // are positions, the 64 rows of this step ARE word w of the bitmap: one plain 8-byte store (64 lanes OR-ing into two words
    // serialise in the L2's atomic unit -- it was most of the build's time: 1.4 ms for 14.6 M rows)
    if (present && payload_via == 0) { const u64 am = __ballot(active); if (hlane() == 0) ((u64*)present)[w] = am; }
    if (queued) {
      u64 idx = 0; bool ins = false;
      if (active) {
        u64 kw[MAX_KW]; u64 h;
#pragma unroll
        for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
        const bool any_null = make_key(K, GPUQ_REGS, kw, h);
        idx = kw[0] - (u64)T.dense_min;
        ins = !(any_null && !null_eq);
        if (ins && idx >= T.dense_range) { atomicOr(P.flags, FLAG_TABLE_FULL); ins = false; }
        if (!ins && S.hit_out) S.hit_out[pos] = s_bits ? S.T.dense[aux] : aux;      // a surviving row without a usable key is still a row of the build side
      }
      const u64 m = __ballot(ins);
      if (ins) {
        const uint32_t j = qn + (uint32_t)__popcll(m & ((1ull << hlane()) - 1));
        q_pos[wv][j] = (uint32_t)pos; q_idx[wv][j] = (uint32_t)idx; q_aux[wv][j] = aux;
      }
      qn += (uint32_t)__popcll(m);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (qn >= 64) { qn -= 64; flush(qn, 64); __builtin_amdgcn_wave_barrier(); }
      return;
    }
    if (!active) return;
    u64 kw[MAX_KW]; u64 h;
#pragma unroll
    for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
    const bool any_null = make_key(K, GPUQ_REGS, kw, h);
    uint32_t row = (uint32_t)pos;
    if (payload_via > 0) { row = P.via[payload_via - 1][pos]; if (present) atomicOr(&present[row >> 5], 1u << (row & 31)); }
    if (any_null && !null_eq) return;   // a NULL key never matches (SQL equi-join)
    if (T.dense) {
      const u64 idx = kw[0] - (u64)T.dense_min;
      if (idx >= T.dense_range) { atomicOr(P.flags, FLAG_TABLE_FULL); return; }      // cannot happen: the range was measured on these rows
      build_insert_dense(P, T, next, idx, row);
      return;
    }
    bool inserted; uint32_t old = NIL;
    const u64 s = ht_find_or_insert(T, kw, h, row, inserted);
    if (s == ~0ull) { atomicOr(P.flags, FLAG_TABLE_FULL); return; }
    if (!inserted && next) {
      uint32_t* head = (uint32_t*)(T.slots + s * (u64)T.slot_words) + 1;   // high half of word 0
      old = __hip_atomic_exchange(head, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (inserted) {
      if (next) next[row] = NIL;
    } else {
      if (!(__hip_atomic_load(P.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FLAG_DUP_BUILD_KEY)) atomicOr(P.flags, FLAG_DUP_BUILD_KEY);
      if (next) next[row] = old;
    }
  });
  if (queued && qn > 0) flush(0, qn);
  if (SEMI && S.on && S.rows_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) survivors += __shfl_xor(survivors, o);
    if (hlane() == 0 && survivors) atomicAdd((unsigned long long*)S.rows_out, (unsigned long long)survivors);
  }
}
#if defined(GPUQ_JIT) && defined(GPUQ_JIT_SEMI) && GPUQ_JIT_SEMI == 2
// Chain fusion, specialised (run-time compiled) for the PK/FK shape: ONE narrow key on each side (registers JIT_KEY_REG0 = the key
// the rows are looked up with in the other join's table, JIT_KEY2_REG = this build's key), a direct-addressed table to build, no
// `present` bitmap.  The front is the unique probe's (k_join_probe_unique_body): U words of rows per step, every column load of all
// U issued before any value is looked at, then the U lookups in the other table back to back; survivors go to the wave's LDS queue
// and are inserted 64 at a time (see k_join_build_body).  The per-word form above waits for one lookup per word: 1.9 ms for the
// orders side of SF100 q3 against 0.84 ms for the unique probe over the same rows.
template <int MAXC>
__device__ __forceinline__ void k_join_build_semi1_body(const DevProgram P, const i64 n_arg, const HashTable T, uint32_t* __restrict__ next, const SemiProbe S) {
#ifndef GPUQ_SEMI_ROWS
#define GPUQ_SEMI_ROWS 4
#endif
  constexpr int U = GPUQ_SEMI_ROWS;
  constexpr int QC = 64 * (U + 1);
  __shared__ uint32_t q_pos[HWAVES][QC], q_idx[HWAVES][QC], q_aux[HWAVES][QC];
  const i64 n = rows_of(P, n_arg);
  const i64 nwords = (n + 63) >> 6;
  const int wv = hwave();
  const bool s_dense = S.T.dense != nullptr;
  const uint32_t* __restrict__ sbits = S.T.dense_bits;
  const bool late_row = s_dense && sbits != nullptr;      // the other table's row is only read for the survivors, at the flush
  uint32_t qn = 0, survivors = 0;
  auto flush = [&](const uint32_t from, const uint32_t cnt) {
    const bool on = (uint32_t)hlane() < cnt;
    const uint32_t j = from + (on ? (uint32_t)hlane() : 0u);
    const uint32_t pos = q_pos[wv][j], idx = q_idx[wv][j], aux = q_aux[wv][j];
    if (on) {
      if (S.hit_out) S.hit_out[pos] = late_row ? S.T.dense[aux] : aux;
      build_insert_dense(P, T, next, (u64)idx, pos);
    }
  };
  const i64 wave0 = ((i64)blockIdx.x * HWAVES + hwave()) * U, wstride = (i64)gridDim.x * HWAVES * U;
  for (i64 wb = wave0; wb < nwords; wb += wstride) {
    bool act[U]; u64 key1[U], key2[U]; bool n1[U], n2[U]; i64 posc[U];
    {
      JitPre jq[U]; JitRaw jw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const i64 pos = ((wb + u) << 6) + hlane();
        act[u] = (wb + u) < nwords && pos < n;
        posc[u] = act[u] ? pos : n - 1;      // clamped, masked afterwards (n > 0 here)
        gpuq_jit_pre(P, posc[u], jq[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) gpuq_jit_load(P, posc[u], jq[u], jw[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        GPUQ_REGS_DECL;
        const bool pass = gpuq_jit_compute(P, posc[u], jw[u], GPUQ_REGS);
        act[u] = act[u] && pass;
        n1[u] = (rnulls >> JIT_KEY_REG0) & 1; n2[u] = (rnulls >> JIT_KEY2_REG) & 1;
        key1[u] = rlo[JIT_KEY_REG0]; key2[u] = rlo[JIT_KEY2_REG];
      }
    }
    // the other table: all U lookups in flight
    uint32_t aux[U]; bool found[U];
    if (s_dense) {
      uint32_t bw[U]; u64 i1[U]; bool in[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        i1[u] = key1[u] - (u64)S.T.dense_min;
        in[u] = act[u] && !n1[u] && i1[u] < S.T.dense_range;
        bw[u] = 0;
        if (in[u]) bw[u] = sbits ? sbits[i1[u] >> 5] : S.T.dense[i1[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (sbits) { found[u] = in[u] && ((bw[u] >> (i1[u] & 31)) & 1u); aux[u] = (uint32_t)i1[u]; }
        else { found[u] = in[u] && bw[u] != NIL; aux[u] = bw[u]; }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        aux[u] = NIL;
        if (act[u] && !n1[u]) { u64 kw1[MAX_KW]; for (int q = 0; q < MAX_KW; ++q) kw1[q] = 0; kw1[0] = key1[u]; aux[u] = join_lookup(S.T, kw1, hash_combine(0x243F6A8885A308D3ull, key1[u], 0, false)); }
        found[u] = aux[u] != NIL;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool surv = act[u] && found[u];
      survivors += surv ? 1u : 0u;
      const u64 idx = key2[u] - (u64)T.dense_min;
      bool ins = surv && !n2[u];
      if (ins && idx >= T.dense_range) { atomicOr(P.flags, FLAG_TABLE_FULL); ins = false; }
      if (surv && !ins && S.hit_out) S.hit_out[posc[u]] = late_row ? S.T.dense[aux[u]] : aux[u];
      const u64 m = __ballot(ins);
      if (ins) {
        const uint32_t j = qn + (uint32_t)__popcll(m & ((1ull << hlane()) - 1));
        q_pos[wv][j] = (uint32_t)posc[u]; q_idx[wv][j] = (uint32_t)idx; q_aux[wv][j] = aux[u];
      }
      qn += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    while (qn >= 64) { qn -= 64; flush(qn, 64); }
    __builtin_amdgcn_wave_barrier();
  }
  if (qn > 0) flush(0, qn);
  if (S.rows_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) survivors += __shfl_xor(survivors, o);
    if (hlane() == 0 && survivors) atomicAdd((unsigned long long*)S.rows_out, (unsigned long long)survivors);
  }
}
#endif
#ifndef GPUQ_JIT
template <int MAXC, bool SEMI>
__global__ void __launch_bounds__(HBLOCK) k_join_build(const DevProgram P, const i64 n, const KeySpec K, const HashTable T,
                                                       uint32_t* __restrict__ next, uint32_t* __restrict__ present, const int payload_via,
                                                       const int null_eq, const SemiProbe S) { k_join_build_body<MAXC, SEMI>(P, n, K, T, next, present, payload_via, null_eq, S); }
#elif GPUQ_JIT_KERNEL == 5
#ifndef GPUQ_JIT_SEMI
#define GPUQ_JIT_SEMI 0
#endif
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const HashTable T,
                                                       uint32_t* __restrict__ next, uint32_t* __restrict__ present, const int payload_via,
                                                       const int null_eq, const SemiProbe S) {
#if GPUQ_JIT_SEMI == 2
  k_join_build_semi1_body<0>(P, n, T, next, S);
#else
  k_join_build_body<0, GPUQ_JIT_SEMI != 0>(P, n, K, T, next, present, payload_via, null_eq, S);
#endif
}
#endif

// ------------------------------------------------------------------ join probe
// Emits (build_row, probe_row) pairs with wave-ballot compaction: one global atomic per wave per
// chain step.  Pair order is not input order (DataFusion's is batch-local and unspecified across
// partitions); the SET of pairs is deterministic.
template <int MAXC>
__device__ __forceinline__ void k_join_probe_body(const DevProgram P, const i64 n_arg, const KeySpec K, const HashTable T,
                                                       const uint32_t* __restrict__ next, const int join_type, const int payload_via,
                                                       const int null_eq, uint32_t* __restrict__ out_build, uint32_t* __restrict__ out_probe,
                                                       const u64 out_cap, u64* __restrict__ out_count, uint32_t* __restrict__ visited) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  const i64 nwords = (n + 63) >> 6;
  const bool emit_pairs = (join_type == JT_INNER || join_type == JT_LEFT || join_type == JT_RIGHT || join_type == JT_FULL);
  const bool probe_outer = (join_type == JT_RIGHT || join_type == JT_FULL);
  const bool mark = (visited != nullptr);
  for (i64 w = (i64)blockIdx.x * HWAVES + hwave(); w < nwords; w += (i64)gridDim.x * HWAVES) {
    const i64 pos = (w << 6) + hlane();
    bool active = pos < n;
    GPUQ_REGS_DECL;
    if (active) active = GPUQ_EVAL(MAXC, P, pos);
    uint32_t cur = NIL;
    uint32_t prow = (uint32_t)pos;
    if (active) {
      if (payload_via > 0) prow = P.via[payload_via - 1][pos];
      u64 kw[MAX_KW]; u64 h;
#pragma unroll
      for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
      const bool any_null = make_key(K, GPUQ_REGS, kw, h);
      if (!(any_null && !null_eq)) cur = join_lookup(T, kw, h);
    }
    if (join_type == JT_RIGHT_SEMI || join_type == JT_RIGHT_ANTI) {
      const bool emit = active && ((join_type == JT_RIGHT_SEMI) == (cur != NIL));
      const u64 m = __ballot(emit);
      if (m) {
        u64 base = 0;
        if (hlane() == 0) base = a_add(out_count, (u64)__popcll(m));
        base = __shfl(base, 0);
        if (emit) {
          const u64 idx = base + (u64)__popcll(m & ((1ull << hlane()) - 1));
          if (idx < out_cap) { if (out_build) out_build[idx] = NULL_ROW; out_probe[idx] = prow; }
          else atomicOr(P.flags, FLAG_OUT_OVERFLOW);
        }
      }
      continue;
    }
    bool first = true;
    for (;;) {
      const bool has = (cur != NIL);
      const bool emit = emit_pairs && active && (has || (first && probe_outer));
      const u64 m = __ballot(emit);
      if (m) {
        u64 base = 0;
        if (hlane() == 0) base = a_add(out_count, (u64)__popcll(m));
        base = __shfl(base, 0);
        if (emit) {
          const u64 idx = base + (u64)__popcll(m & ((1ull << hlane()) - 1));
          if (idx < out_cap) { out_build[idx] = has ? cur : NULL_ROW; out_probe[idx] = prow; }
          else atomicOr(P.flags, FLAG_OUT_OVERFLOW);
        }
      }
      if (has) {
        if (mark) atomicOr(&visited[cur >> 5], 1u << (cur & 31));
        cur = next ? next[cur] : NIL;
      }
      first = false;
      if (__ballot(cur != NIL) == 0) break;
    }
  }
}
#ifndef GPUQ_JIT
template <int MAXC>
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(HBLOCK) k_join_probe(const DevProgram P, const i64 n, const KeySpec K, const HashTable T,
                                                       const uint32_t* __restrict__ next, const int join_type, const int payload_via,
                                                       const int null_eq, uint32_t* __restrict__ out_build, uint32_t* __restrict__ out_probe,
                                                       const u64 out_cap, u64* __restrict__ out_count, uint32_t* __restrict__ visited) { k_join_probe_body<MAXC>(P, n, K, T, next, join_type, payload_via, null_eq, out_build, out_probe, out_cap, out_count, visited); }
#endif
#elif GPUQ_JIT_KERNEL == 6
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const HashTable T,
                                                       const uint32_t* __restrict__ next, const int join_type, const int payload_via,
                                                       const int null_eq, uint32_t* __restrict__ out_build, uint32_t* __restrict__ out_probe,
                                                       const u64 out_cap, u64* __restrict__ out_count, uint32_t* __restrict__ visited) { k_join_probe_body<0>(P, n, K, T, next, join_type, payload_via, null_eq, out_build, out_probe, out_cap, out_count, visited); }
#endif

// ------------------------------------------------------------------ join probe, unique build keys
// No output atomics and no per-row match vector: every wave owns a SEGMENT of `wpw` consecutive 64-row words, probes it in
// order and appends its (build_row, probe_row) pairs to seg_build / seg_probe starting at the segment's first row (a segment
// of r rows emits at most r pairs, so segments never collide); seg_counts[g] = pairs of segment g.  A scan of the counts and
// k_copy_segments then move the segments to their final, PROBE-ORDERED places.  A selective join (TPC-H q3: 14.6 M pairs
// from 600 M probe rows) writes and re-reads only its pairs; the first version wrote a 4-byte match word per probe ROW and
// read all of them back in the compaction pass (4.8 GB of the kernel pair's 14 GB at SF100).  A single global counter would
// serialise the whole probe (measured: 51 ms for 2^28 probes whatever the table size -- one device-scope atomic per wave step).
__device__ __forceinline__ uint32_t emit_pairs(const bool emit, const uint32_t hit, const uint32_t prow, const u64 seg_base, const uint32_t cnt,
                                               uint32_t* __restrict__ seg_build, uint32_t* __restrict__ seg_probe) {
  const u64 m = __ballot(emit);
  if (emit) {
    const u64 j = seg_base + cnt + (u64)__popcll(m & ((1ull << hlane()) - 1));
    if (seg_build) seg_build[j] = hit;
    seg_probe[j] = prow;
  }
  return (uint32_t)__popcll(m);
}
#ifdef GPUQ_JIT_PROBE1
// JIT specialisation for the common PK/FK shape: ONE narrow (<= 64-bit) non-null-word key (16-byte slots, or the
// direct-addressed table).  Each lane keeps U probe rows in flight: U key evaluations, then U table loads issued back to back,
// then U resolutions -- the random table access is the long pole of a probe, and one outstanding access per lane cannot
// cover its latency (measured: 26-38 G probes/s with U = 1 whatever the table size).
template <int MAXC>
__device__ __forceinline__ void k_join_probe_unique_body(const DevProgram P, const i64 n_arg, const KeySpec K, const HashTable T,
                                                              const int join_type, const int null_eq, const int payload_via,
                                                              uint32_t* __restrict__ seg_build, uint32_t* __restrict__ seg_probe,
                                                              uint32_t* __restrict__ seg_counts, const int nsegs, const i64 wpw,
                                                              uint32_t* __restrict__ visited) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
#ifndef GPUQ_PROBE_ROWS
#define GPUQ_PROBE_ROWS 4
#endif
  constexpr int U = GPUQ_PROBE_ROWS;
  // Hit queue (sparse domain, Inner / RightSemi): a probe that finds its key's bit set still has to fetch the build row from the row
  // array and write the pair -- a third dependent memory round trip per step that one row in twenty needs (SF100 q3: 14.6 M hits in
  // 323 M probes).  Hits go to a FIFO in LDS (table index, probe row: first in, first out keeps probe order) and are resolved 64 at a
  // time with every lane busy and full-wave coalesced pair stores; the per-step chain is columns -> bitmap.
  constexpr int QC = 64 * (U + 1);
  __shared__ uint32_t hq_idx[HWAVES][QC], hq_row[HWAVES][QC];
  const i64 seg = (i64)blockIdx.x * HWAVES + hwave();
  if (seg >= nsegs) return;
  const i64 nwords = (n + 63) >> 6;
  const i64 w0 = seg * wpw;
  i64 w1 = w0 + wpw; if (w1 > nwords) w1 = nwords;
  const u64 seg_base = (u64)w0 << 6;
  const bool probe_outer = (join_type == JT_RIGHT || join_type == JT_FULL);
  const bool want_pairs = (join_type == JT_INNER || join_type == JT_LEFT || join_type == JT_RIGHT || join_type == JT_FULL);
  const u64 mask = T.n_slots - 1;
  const ulonglong2* __restrict__ slots = (const ulonglong2*)T.slots;
  const uint32_t* __restrict__ dense = T.dense;
  uint32_t cnt = 0;
  const bool hit_queue = dense != nullptr && T.dense_bits != nullptr && visited == nullptr && (join_type == JT_INNER || join_type == JT_RIGHT_SEMI);
  uint32_t qhead = 0, qtail = 0;      // wave-uniform; positions count up, slots are taken modulo QC
  const int wvq = hwave();
  auto drain = [&](const uint32_t n_out) {      // the n_out (<= 64) oldest hits -> pairs at seg_base + cnt
    const bool on = (uint32_t)hlane() < n_out;
    const uint32_t slot = (qhead + (on ? (uint32_t)hlane() : 0u)) % (uint32_t)QC;
    const uint32_t idx = hq_idx[wvq][slot], pr = hq_row[wvq][slot];
    if (on) {
      const u64 pos = seg_base + cnt + (u64)hlane();
      if (seg_build) seg_build[pos] = dense[idx];
      seg_probe[pos] = pr;
    }
    cnt += n_out; qhead += n_out;
  };
  for (i64 wb = w0; wb < w1; wb += U) {
    bool act[U]; u64 key[U]; u64 hs[U]; bool isn[U]; uint32_t prow[U];
    // stage 1: evaluate U rows.  The generated evaluator is used in its three stages (expr_compile.cpp: pre / load / compute) so that
    // EVERY column load of all U rows is issued before any loaded value is looked at: calling the one-shot evaluator per row puts
    // each row's loads behind the previous row's predicate (vmcnt is in order), i.e. 2 U dependent memory round trips per step
    // instead of 2.  Rows beyond the end are clamped to the last row and masked afterwards (no exec-masked load regions).
#ifndef GPUQ_PROBE_STAGED
#define GPUQ_PROBE_STAGED 1
#endif
#if GPUQ_PROBE_STAGED
    {
      JitPre jq[U]; JitRaw jw[U]; i64 posc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const i64 pos = ((wb + u) << 6) + hlane();
        act[u] = (wb + u) < w1 && pos < n;
        posc[u] = act[u] ? pos : (n > 0 ? n - 1 : 0);
        prow[u] = (uint32_t)pos;
        gpuq_jit_pre(P, posc[u], jq[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) gpuq_jit_load(P, posc[u], jq[u], jw[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        GPUQ_REGS_DECL;
        const bool pass = gpuq_jit_compute(P, posc[u], jw[u], GPUQ_REGS);
        act[u] = act[u] && pass;
        isn[u] = pass && ((rnulls >> JIT_KEY_REG0) & 1);
        key[u] = (pass && !isn[u]) ? rlo[JIT_KEY_REG0] : 0;
        hs[u] = 0;
        if (act[u] && payload_via > 0) prow[u] = P.via[payload_via - 1][posc[u]];
      }
    }
#else
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const i64 pos = ((wb + u) << 6) + hlane();
      act[u] = (wb + u) < w1 && pos < n;
      key[u] = 0; hs[u] = 0; isn[u] = false; prow[u] = (uint32_t)pos;
      if (act[u]) {
        GPUQ_REGS_DECL;
        act[u] = GPUQ_EVAL(MAXC, P, pos);
        isn[u] = (rnulls >> JIT_KEY_REG0) & 1;
        key[u] = isn[u] ? 0 : rlo[JIT_KEY_REG0];
        if (payload_via > 0) prow[u] = P.via[payload_via - 1][pos];
      }
    }
#endif
    uint32_t hit[U];
    if (dense) {
      // direct addressing: one load per row, all U in flight; neighbouring lanes with equal or adjacent keys share cache lines
      const uint32_t* __restrict__ dbits = T.dense_bits;
      if (dbits) {
        // sparse domain: presence bits first (all U loads in flight), the row array only for the hits
        uint32_t bw[U]; bool in[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const u64 idx = key[u] - (u64)T.dense_min;
          in[u] = act[u] && !isn[u] && idx < T.dense_range;
          bw[u] = 0;
          if (in[u]) bw[u] = dbits[idx >> 5];
        }
        if (hit_queue) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const u64 idx = key[u] - (u64)T.dense_min;
            const bool h = in[u] && ((bw[u] >> (idx & 31)) & 1u);
            const u64 m = __ballot(h);
            if (h) { const uint32_t slot = (qtail + (uint32_t)__popcll(m & ((1ull << hlane()) - 1))) % (uint32_t)QC; hq_idx[wvq][slot] = (uint32_t)idx; hq_row[wvq][slot] = prow[u]; }
            qtail += (uint32_t)__popcll(m);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          while (qtail - qhead >= 64u) drain(64u);
          __builtin_amdgcn_wave_barrier();
          continue;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const u64 idx = key[u] - (u64)T.dense_min;
          hit[u] = NIL;
          if (in[u] && ((bw[u] >> (idx & 31)) & 1u)) hit[u] = dense[idx];
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const u64 idx = key[u] - (u64)T.dense_min;
          hit[u] = NIL;
          if (act[u] && !isn[u] && idx < T.dense_range) hit[u] = dense[idx];
        }
      }
    } else {
      // Neighbouring lanes with the same key (clustered foreign keys: the lines of one order) look the key up once: only the
      // first lane of a run ("head") touches the table, the others copy its answer.  All cross-lane reads are executed by
      // every lane (a masked-off source lane reads as 0).
      bool head[U]; int src[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        hs[u] = hash_combine(0x243F6A8885A308D3ull, key[u], 0, isn[u]);
        const bool probe = act[u] && !(isn[u] && !null_eq);
        const int pp = __shfl_up((int)probe, 1);
        const u64 pk = __shfl_up(key[u], 1);
        const int pn = __shfl_up((int)isn[u], 1);
        const bool same = (hlane() > 0) & (pp != 0) & (pk == key[u]) & ((pn != 0) == isn[u]);
        head[u] = probe & !same;
        const u64 hm = __ballot(head[u]) & ((2ull << hlane()) - 1);          // heads at or below this lane
        src[u] = (probe && hm) ? (63 - __clzll((long long)hm)) : hlane();
      }
      // stage 2: first slot of every run
      ulonglong2 sv[U]; u64 si[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { si[u] = hs[u] & mask; sv[u] = make_ulonglong2(0, 0); if (head[u]) sv[u] = slots[si[u]]; }
      // stage 3: resolve (linear probing continues per row only on a tag/key mismatch)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t h_ = NIL;
        if (head[u]) {
          const uint32_t tag = tag_of(hs[u]);
          ulonglong2 v = sv[u]; u64 sidx = si[u];
          for (u64 probes = 0; probes < T.n_slots; ++probes) {
            const uint32_t st = (uint32_t)v.x;
            if (st == 0u) break;
            if (st == tag && v.y == key[u]) { h_ = (uint32_t)(v.x >> 32); break; }
            sidx = (sidx + 1) & mask; v = slots[sidx];
          }
        }
        hit[u] = (uint32_t)__shfl((int)h_, src[u]);      // run members take their head's answer (heads and idle lanes read themselves)
        if (!act[u]) hit[u] = NIL;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (visited && hit[u] != NIL) atomicOr(&visited[hit[u] >> 5], 1u << (hit[u] & 31));
      bool emit;
      if (join_type == JT_RIGHT_SEMI) emit = act[u] && hit[u] != NIL;
      else if (join_type == JT_RIGHT_ANTI) emit = act[u] && hit[u] == NIL;
      else emit = want_pairs && act[u] && (hit[u] != NIL || probe_outer);
      cnt += emit_pairs(emit, hit[u], prow[u], seg_base, cnt, seg_build, seg_probe);
    }
  }
  if (hit_queue && qtail != qhead) drain(qtail - qhead);
  if (hlane() == 0) seg_counts[seg] = cnt;
}
#else
template <int MAXC>
__device__ __forceinline__ void k_join_probe_unique_body(const DevProgram P, const i64 n_arg, const KeySpec K, const HashTable T,
                                                              const int join_type, const int null_eq, const int payload_via,
                                                              uint32_t* __restrict__ seg_build, uint32_t* __restrict__ seg_probe,
                                                              uint32_t* __restrict__ seg_counts, const int nsegs, const i64 wpw,
                                                              uint32_t* __restrict__ visited) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  const i64 seg = (i64)blockIdx.x * HWAVES + hwave();
  if (seg >= nsegs) return;
  const i64 nwords = (n + 63) >> 6;
  const i64 w0 = seg * wpw;
  i64 w1 = w0 + wpw; if (w1 > nwords) w1 = nwords;
  const u64 seg_base = (u64)w0 << 6;
  const bool probe_outer = (join_type == JT_RIGHT || join_type == JT_FULL);
  const bool want_pairs = (join_type == JT_INNER || join_type == JT_LEFT || join_type == JT_RIGHT || join_type == JT_FULL);
  uint32_t cnt = 0;
  // the segment's words in order (pair positions depend on it), two of them with all column loads in flight per step
  for_rows_in_flight<MAXC>(P, n, w0, 1, [&](const i64, const i64 pos, bool active, GPUQ_REGS_PARAM) {
    uint32_t hit = NIL, prow = (uint32_t)pos;
    if (active) {
      if (payload_via > 0) prow = P.via[payload_via - 1][pos];
      u64 kw[MAX_KW]; u64 h;
#pragma unroll
      for (int q = 0; q < MAX_KW; ++q) kw[q] = 0;
      const bool any_null = make_key(K, GPUQ_REGS, kw, h);
      if (!(any_null && !null_eq)) hit = join_lookup(T, kw, h);
      if (visited && hit != NIL) atomicOr(&visited[hit >> 5], 1u << (hit & 31));
    }
    bool emit;
    if (join_type == JT_RIGHT_SEMI) emit = active && hit != NIL;
    else if (join_type == JT_RIGHT_ANTI) emit = active && hit == NIL;
    else emit = want_pairs && active && (hit != NIL || probe_outer);
    cnt += emit_pairs(emit, hit, prow, seg_base, cnt, seg_build, seg_probe);
  }, w1);
  if (hlane() == 0) seg_counts[seg] = cnt;
}
#endif  // GPUQ_JIT_PROBE1
#ifndef GPUQ_JIT
template <int MAXC>
__global__ void __launch_bounds__(HBLOCK) k_join_probe_unique(const DevProgram P, const i64 n, const KeySpec K, const HashTable T,
                                                              const int join_type, const int null_eq, const int payload_via,
                                                              uint32_t* __restrict__ seg_build, uint32_t* __restrict__ seg_probe,
                                                              uint32_t* __restrict__ seg_counts, const int nsegs, const i64 wpw,
                                                              uint32_t* __restrict__ visited) { k_join_probe_unique_body<MAXC>(P, n, K, T, join_type, null_eq, payload_via, seg_build, seg_probe, seg_counts, nsegs, wpw, visited); }
#elif GPUQ_JIT_KERNEL == 7
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const HashTable T,
                                                              const int join_type, const int null_eq, const int payload_via,
                                                              uint32_t* __restrict__ seg_build, uint32_t* __restrict__ seg_probe,
                                                              uint32_t* __restrict__ seg_counts, const int nsegs, const i64 wpw,
                                                              uint32_t* __restrict__ visited) { k_join_probe_unique_body<0>(P, n, K, T, join_type, null_eq, payload_via, seg_build, seg_probe, seg_counts, nsegs, wpw, visited); }
#endif

#ifndef GPUQ_JIT
// one wave per segment: its pairs from the segment's scratch place to their final place (offsets = exclusive scan of the counts)
__global__ void __launch_bounds__(HBLOCK) k_copy_segments(const uint32_t* __restrict__ seg_build, const uint32_t* __restrict__ seg_probe,
                                                          const uint32_t* __restrict__ seg_offsets, const int nsegs, const i64 wpw, const u64* __restrict__ total,
                                                          uint32_t* __restrict__ out_build, uint32_t* __restrict__ out_probe, const u64 out_cap,
                                                          uint32_t* __restrict__ flags) {
  const u64 tot = *total;
  for (i64 seg = (i64)blockIdx.x * HWAVES + hwave(); seg < nsegs; seg += (i64)gridDim.x * HWAVES) {
    const u64 off = seg_offsets[seg];
    const u64 end = (seg + 1 < nsegs) ? (u64)seg_offsets[seg + 1] : tot;
    const u64 src = (u64)(seg * wpw) << 6;
    for (u64 i = hlane(); off + i < end; i += 64) {
      if (off + i < out_cap) {
        if (out_build) out_build[off + i] = seg_build[src + i];
        out_probe[off + i] = seg_probe[src + i];
      } else { atomicOr(flags, FLAG_OUT_OVERFLOW); break; }
    }
  }
}
#endif

// ------------------------------------------------------------------ partitioned probe (radix join over a direct-addressed table)
// A probe side whose keys arrive in random order touches one random cache line of the table per row; beyond the Infinity
// Cache every such touch moves a whole line across the fabric (measured: 2^28 probes of a 2^24-key table = 19.8 GB of fetches
// for 9.7 GB of algorithmic traffic, 42 G probes/s whatever the table layout).  The partitioned probe makes the accesses
// local instead: the probe rows are range-partitioned on the HIGH bits of (key - min) with ONE LDS-staged scatter pass
// (records of 8 bytes: table index << 32 | probe row), so that the records of one partition hit one slice of the table
// small enough for an XCD's L2; the lookups then run over the partition-ordered records with the segments dealt to the
// XCDs in contiguous runs (blocks that share an XCD work on neighbouring records, i.e. on the same slice).  The build side
// needs no partitioning at all: slice p of the direct-addressed table IS partition p's table.
//   k_rj_pack      rows -> records (front-end: predicate, key, range check) + per-block histogram of the partition digits
//   k_rj_scatter   one pass: tile sorted by digit in LDS (LDS atomics give the ranks; order inside a partition is free),
//                  runs written to consecutive addresses
//   k_rj_probe     records -> pairs, per-wave segments as in the direct probe
// Pair order is partition order, not probe order (DataFusion's is unspecified across batches); Inner / RightSemi only.
constexpr uint32_t RJ_MAX_PARTS = 1024;
constexpr int RJ_TILE = 4096;                 // records sorted in LDS per step: 32 KB
constexpr int RJ_ROUNDS = RJ_TILE / HBLOCK;
constexpr u64 RJ_DROPPED = ~0ull;             // a row that cannot match (predicate, NULL key, key outside the table's range)
struct RjGeom { uint32_t nparts; uint32_t shift; i64 tile; int32_t nblocks; int32_t pad; };      // digit = index >> shift; digit nparts = dropped rows

template <int MAXC>
__device__ __forceinline__ void k_rj_pack_body(const DevProgram P, const i64 n_arg, const KeySpec K, const HashTable T, const int payload_via, const RjGeom G,
                                                   u64* __restrict__ rec, int32_t* __restrict__ hist) {
  const i64 n = rows_of(P, n_arg);      // deferred execution: the row count is a device word, n_arg its host-side bound
  __shared__ uint32_t lcnt[RJ_MAX_PARTS + 1];
  for (uint32_t d = threadIdx.x; d <= G.nparts; d += HBLOCK) lcnt[d] = 0;
  __syncthreads();
  const int kr = __builtin_amdgcn_readfirstlane(K.key_reg[0]);
  const i64 a = (i64)blockIdx.x * G.tile;
  i64 b = a + G.tile; if (b > n) b = n;
  for (i64 p0 = a + (i64)hwave() * 64; p0 < b; p0 += HBLOCK) {
    const i64 pos = p0 + hlane();
    bool active = pos < b;
    GPUQ_REGS_DECL;
    if (active) active = GPUQ_EVAL(MAXC, P, pos);
    u64 r = RJ_DROPPED; uint32_t d = G.nparts;
    if (active && !((rnulls >> kr) & 1)) {
      const u64 idx = rlo[kr] - (u64)T.dense_min;
      if (idx < T.dense_range) {
        uint32_t prow = (uint32_t)pos;
        if (payload_via > 0) prow = P.via[payload_via - 1][pos];
        r = (idx << 32) | prow; d = (uint32_t)(idx >> G.shift);
      }
    }
    if (pos < b) { rec[pos] = r; atomicAdd(&lcnt[d], 1u); }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d <= G.nparts; d += HBLOCK) hist[(size_t)d * G.nblocks + blockIdx.x] = (int32_t)lcnt[d];
}
#ifndef GPUQ_JIT
template <int MAXC>
__global__ void __launch_bounds__(HBLOCK) k_rj_pack(const DevProgram P, const i64 n, const KeySpec K, const HashTable T, const int payload_via, const RjGeom G,
                                                    u64* __restrict__ rec, int32_t* __restrict__ hist) { k_rj_pack_body<MAXC>(P, n, K, T, payload_via, G, rec, hist); }
#elif GPUQ_JIT_KERNEL == 15
extern "C" __global__ void __launch_bounds__(HBLOCK) gpuq_jit_entry(const DevProgram P, const i64 n, const KeySpec K, const HashTable T, const int payload_via, const RjGeom G,
                                                    u64* __restrict__ rec, int32_t* __restrict__ hist) { k_rj_pack_body<0>(P, n, K, T, payload_via, G, rec, hist); }
#endif

#ifndef GPUQ_JIT
__device__ __forceinline__ uint32_t rj_digit(const u64 r, const RjGeom& G) { return r == RJ_DROPPED ? G.nparts : (uint32_t)(r >> (32 + G.shift)); }
// offsets: exclusive scan of hist ([digit][block]); dropped rows (digit nparts) are not written
__global__ void __launch_bounds__(HBLOCK) k_rj_scatter(const u64* __restrict__ rec, const i64 n, const RjGeom G, const int32_t* __restrict__ offsets, u64* __restrict__ out) {
  __shared__ u64 sk[RJ_TILE];
  __shared__ uint32_t lcnt[RJ_MAX_PARTS + 1];     // records of the digit in this step, then the digit's first slot in sk
  __shared__ uint32_t gbase[RJ_MAX_PARTS + 1];    // global position of the digit's next record
  __shared__ uint32_t wsum[HWAVES];
  const int t = threadIdx.x, l = hlane(), w = hwave();
  const uint32_t D = G.nparts;                    // digits 0..D-1 are written; digit D (dropped) is counted and skipped
  for (uint32_t d = t; d <= D; d += HBLOCK) gbase[d] = (uint32_t)offsets[(size_t)d * G.nblocks + blockIdx.x];
  const uint32_t per = (D + 1 + HBLOCK - 1) / HBLOCK;       // digits per thread in the scan (<= 5)
  const i64 a = (i64)blockIdx.x * G.tile;
  i64 b = a + G.tile; if (b > n) b = n;
  for (i64 s0 = a; s0 < b; s0 += RJ_TILE) {
    for (uint32_t d = t; d <= D; d += HBLOCK) lcnt[d] = 0;
    __syncthreads();
    u64 r[RJ_ROUNDS]; uint32_t rank[RJ_ROUNDS];
#pragma unroll
    for (int q = 0; q < RJ_ROUNDS; ++q) { const i64 i = s0 + q * HBLOCK + t; r[q] = i < b ? rec[i] : RJ_DROPPED; }
#pragma unroll
    for (int q = 0; q < RJ_ROUNDS; ++q) { const i64 i = s0 + q * HBLOCK + t; rank[q] = 0; if (i < b) rank[q] = atomicAdd(&lcnt[rj_digit(r[q], G)], 1u); }
    __syncthreads();
    // exclusive scan of the digit counts (thread t owns digits [t*per, t*per + per))
    uint32_t c[5]; uint32_t tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) { const uint32_t d = t * per + k; c[k] = (k < per && d <= D) ? lcnt[d] : 0; tot += c[k]; }
    uint32_t x = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if (l >= off) x += y; }
    if (l == 63) wsum[w] = x;
    __syncthreads();
    uint32_t pre = 0;
#pragma unroll
    for (int q = 0; q < HWAVES; ++q) if (q < w) pre += wsum[q];
    uint32_t run = pre + x - tot;
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) { const uint32_t d = t * per + k; if (k < per && d <= D) { lcnt[d] = run; run += c[k]; } }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RJ_ROUNDS; ++q) { const i64 i = s0 + q * HBLOCK + t; if (i < b) sk[lcnt[rj_digit(r[q], G)] + rank[q]] = r[q]; }
    __syncthreads();
    const uint32_t cnt = (uint32_t)((b - s0) < RJ_TILE ? (b - s0) : RJ_TILE);
    const uint32_t live = lcnt[D];                // dropped records sort last: slots [live, cnt) are skipped
    for (uint32_t j = t; j < live; j += HBLOCK) {
      const u64 v = sk[j];
      const uint32_t d = rj_digit(v, G);
      out[gbase[d] + (j - lcnt[d])] = v;
    }
    __syncthreads();
    for (uint32_t d = t; d <= D; d += HBLOCK) { const uint32_t nxt = d < D ? lcnt[d + 1] : cnt; gbase[d] += nxt - lcnt[d]; }
    __syncthreads();
  }
}

// XCD-aware work order.  Blocks b, b+8, b+16, ... share an XCD (observed round-robin placement; speed only, any placement is
// correct).  Group x = blockIdx % 8 owns the x-th eighth of the records, and ALL waves of the group stride through that
// eighth together, 256 records per wave and step, so that at any moment the group's waves read neighbouring records, i.e. hit
// the same table slice, and the slice stays in the group's L2.  (Giving every wave its own contiguous segment, as the direct
// probe does, puts ~1000 segments = tens of slices in flight per XCD: 2.3 / 4.3 ms for 2^28 records over 2^24 / 2^27 keys.)
// The grid is sized to what is resident at once (host side); a wave's pairs go to its own scratch segment of `wpw` words.
__global__ void __launch_bounds__(HBLOCK) k_rj_probe(const u64* __restrict__ rec, const int32_t* __restrict__ n_live_p, const HashTable T, const int join_type,
                                                     uint32_t* __restrict__ seg_build, uint32_t* __restrict__ seg_probe, uint32_t* __restrict__ seg_counts,
                                                     const int nsegs, const i64 wpw) {
  constexpr int U = 4;
  const i64 seg = (i64)blockIdx.x * HWAVES + hwave();
  if (seg >= nsegs) return;
  const i64 n = (i64)*n_live_p;
  const i64 nwords = (n + 63) >> 6;
  const i64 x = blockIdx.x % 8, nbx = ((i64)gridDim.x - x + 7) / 8;          // blocks of this group
  const i64 g0 = nwords * x / 8, g1 = nwords * (x + 1) / 8;                   // the group's words
  const i64 nwx = nbx * HWAVES, wl = (i64)(blockIdx.x / 8) * HWAVES + hwave();  // waves of the group, this wave's index among them
  const u64 seg_base = (u64)(seg * wpw) << 6;
  const uint32_t* __restrict__ dense = T.dense; const uint32_t* __restrict__ dbits = T.dense_bits;
  uint32_t cnt = 0;
  for (i64 wb = g0 + wl * U; wb < g1; wb += nwx * U) {
    u64 r[U]; bool act[U]; uint32_t hit[U]; uint32_t bw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const i64 pos = ((wb + u) << 6) + hlane(); act[u] = (wb + u) < g1 && pos < n; r[u] = act[u] ? rec[pos] : 0; }
    if (dbits) {
#pragma unroll
      for (int u = 0; u < U; ++u) { const u64 idx = r[u] >> 32; bw[u] = act[u] ? dbits[idx >> 5] : 0u; }
#pragma unroll
      for (int u = 0; u < U; ++u) { const u64 idx = r[u] >> 32; hit[u] = NIL; if (act[u] && ((bw[u] >> (idx & 31)) & 1u)) hit[u] = dense[idx]; }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) { hit[u] = NIL; if (act[u]) hit[u] = dense[r[u] >> 32]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) cnt += emit_pairs(act[u] && hit[u] != NIL, hit[u], (uint32_t)r[u], seg_base, cnt, join_type == JT_RIGHT_SEMI ? nullptr : seg_build, seg_probe);
  }
  if (hlane() == 0) seg_counts[seg] = cnt;
}
#endif

#ifndef GPUQ_JIT
// sample of the probe side's key locality: out[0] += adjacent pairs of live rows, out[1] += those within 2^14 table entries
template <int MAXC>
__global__ void __launch_bounds__(HBLOCK) k_join_locality(const DevProgram P, const i64 n, const KeySpec K, const HashTable T, const i64 stride, const i64 nsample,
                                                          u64* __restrict__ out) {
  const int kr = __builtin_amdgcn_readfirstlane(K.key_reg[0]);
  u64 pairs = 0, near = 0;
  for (i64 sidx = (i64)blockIdx.x * HWAVES + hwave(); sidx < nsample; sidx += (i64)gridDim.x * HWAVES) {
    const i64 pos = ((sidx * stride) << 6) + hlane();
    bool active = pos < n;
    GPUQ_REGS_DECL;
    if (active) active = GPUQ_EVAL(MAXC, P, pos);
    active = active && !((rnulls >> kr) & 1);
    const u64 idx = active ? rlo[kr] - (u64)T.dense_min : 0;
    active = active && idx < T.dense_range;
    const u64 nidx = __shfl_down(idx, 1);
    const int nact = __shfl_down((int)active, 1);
    const bool pair = active && nact && hlane() < 63;
    const u64 d = nidx > idx ? nidx - idx : idx - nidx;
    pairs += (u64)__popcll(__ballot(pair));
    near += (u64)__popcll(__ballot(pair && d < (1ull << 14)));
  }
  if (hlane() == 0 && pairs) { atomicAdd((unsigned long long*)out, (unsigned long long)pairs); atomicAdd((unsigned long long*)out + 1, (unsigned long long)near); }
}
#endif

#ifndef GPUQ_JIT
// ------------------------------------------------------------------ launchers
static int hgrid(i64 n, int blocks_per_cu) {
  const i64 nwords = (n + 63) >> 6;
  i64 need = (nwords + HWAVES - 1) / HWAVES;
  if (need < 1) need = 1;
  const i64 cap = (i64)num_cus() * blocks_per_cu;
  return (int)(need < cap ? need : cap);
}

void launch_ht_init(hipStream_t s, const HashTable& T, const AggSpec* A) {
  AggSpec dummy{};
  const u64 total = T.n_slots * (u64)T.slot_words;
  u64 need = (total + HBLOCK - 1) / HBLOCK;
  const u64 cap = (u64)num_cus() * 16;
  const int grid = (int)(need < cap ? (need ? need : 1) : cap);
  hipLaunchKernelGGL(k_ht_init, dim3(grid), dim3(HBLOCK), 0, s, T, A ? *A : dummy, A ? 1 : 0);
}
void launch_agg_hash(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const AggSpec& A, const HashTable& T) {
  if (n <= 0) return;
  if (jit_override().fn && jit_override().kernel_id == 4) {
    (void)jit_launch(jit_override().fn, dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, A, T);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_agg_hash<M>, dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, A, T)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
void launch_key_sample(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, i64 stride, i64 nsample, uint32_t* bitmap, u64 nbits, unsigned long long* passed) {
  if (n <= 0 || nsample <= 0) return;
#define CALL(M) hipLaunchKernelGGL(k_key_sample<M>, dim3(hgrid(nsample, 8)), dim3(HBLOCK), 0, s, P, n, K, stride, nsample, bitmap, nbits - 1, passed)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
}
uint32_t agg_lds_slots(const HashTable& T) {      // LDS table size (slots) of the pre-aggregating kernel, 0 = a slot is too wide for it
  const size_t bytes = (size_t)(T.slot_words + 1) * 8;
  uint32_t cap = 1; while ((size_t)cap * 2 * bytes <= 32768) cap *= 2;
  return cap >= 128 ? cap : 0;
}
int agg_lds_grid(i64 n) { return hgrid(n, 4); }
// fstage: nullptr, or T.n_slots * n_fsum * agg_lds_grid(n) zeroed words for the blocks' float partial sums
void launch_agg_lds(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const AggSpec& A, const HashTable& T, u64* fstage, int n_fsum) {
  if (n <= 0) return;
  const uint32_t lcap = agg_lds_slots(T);
  const size_t lds = (size_t)lcap * (T.slot_words + 1) * 8;
  const int grid = agg_lds_grid(n);
  if (jit_override().fn && jit_override().kernel_id == 13) {
    (void)jit_launch(jit_override().fn, dim3(grid), dim3(HBLOCK), lds, s, P, n, K, A, T, lcap, fstage, n_fsum);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_agg_lds<M>, dim3(grid), dim3(HBLOCK), lds, s, P, n, K, A, T, lcap, fstage, n_fsum)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
  if (fstage && n_fsum > 0) {
    const u64 items = T.n_slots * (u64)n_fsum;
    const u64 need = (items + HWAVES - 1) / HWAVES, cap = (u64)num_cus() * 8;
    hipLaunchKernelGGL(k_fsum_stage_reduce, dim3((unsigned)(need < cap ? (need ? need : 1) : cap)), dim3(HBLOCK), 0, s, T, A, (const u64*)fstage, n_fsum, (uint32_t)grid);
  }
}
void launch_agg_hash_extract(hipStream_t s, const KeySpec& K, const AggSpec& A, const HashTable& T, const AggOut& out, uint32_t* flags) {
  u64 need = (T.n_slots + (u64)HBLOCK * 16 - 1) / ((u64)HBLOCK * 16);      // a wave takes 16 x 64 slots per step
  const u64 cap = (u64)num_cus() * 16;
  const int grid = (int)(need < cap ? (need ? need : 1) : cap);
  hipLaunchKernelGGL(k_agg_hash_extract, dim3(grid), dim3(HBLOCK), 0, s, K, A, T, out, flags);
}
void launch_agg_bucket_id(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, u64 bucket_mask, u64* bid, uint32_t* ids) {
  if (n <= 0) return;
  if (jit_override().fn && jit_override().kernel_id == 11) {
    (void)jit_launch(jit_override().fn, dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, bucket_mask, bid, ids);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_agg_bucket_id<M>, dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, bucket_mask, bid, ids)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
void launch_bucket_bounds(hipStream_t s, const u64* sorted_bid, i64 n, u64 nbuckets, uint32_t* bounds, int shift) {
  const u64 need = (nbuckets + 1 + HBLOCK - 1) / HBLOCK;
  hipLaunchKernelGGL(k_bucket_bounds, dim3((unsigned)(need < 4096 ? need : 4096)), dim3(HBLOCK), 0, s, sorted_bid, n, nbuckets, bounds, shift);
}
void launch_agg_bucket(hipStream_t s, const DevProgram& P, const KeySpec& K, const AggSpec& A, const uint32_t* ids, const uint32_t* bounds, uint32_t nbuckets,
                       uint32_t cap, int slot_words, const AggOut& out) {
  const size_t lds = (size_t)cap * slot_words * 8;
  u64 grid = nbuckets; const u64 gcap = (u64)num_cus() * 8; if (grid > gcap) grid = gcap; if (grid < 1) grid = 1;
  if (jit_override().fn && jit_override().kernel_id == 12) {
    (void)jit_launch(jit_override().fn, dim3((unsigned)grid), dim3(HBLOCK), lds, s, P, K, A, ids, bounds, nbuckets, cap, slot_words, out);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_agg_bucket<M>, dim3((unsigned)grid), dim3(HBLOCK), lds, s, P, K, A, ids, bounds, nbuckets, cap, slot_words, out)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
bool launch_join_build(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, uint32_t* next, uint32_t* present,
                       int payload_via, int null_equals_null, const SemiProbe* semi) {
  if (n <= 0) return true;
  SemiProbe S{}; if (semi) S = *semi;
  if (jit_override().fn && jit_override().kernel_id == 5) {
    (void)jit_launch(jit_override().fn, dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, T, next, present, payload_via, null_equals_null, S);
  } else {
    if (semi) {      // chain fusion: the interpreter form exists for up to 8 input columns (the specialised form for any)
      if (P.n_cols > 8) return false;
#define CALLS(M) hipLaunchKernelGGL((k_join_build<M, true>), dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, T, next, present, payload_via, null_equals_null, S)
      if (P.n_cols <= 2) { CALLS(2); } else if (P.n_cols <= 4) { CALLS(4); } else { CALLS(8); }
#undef CALLS
      return true;
    }
#define CALL(M) hipLaunchKernelGGL((k_join_build<M, false>), dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, T, next, present, payload_via, null_equals_null, S)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
  return true;
}

// build-side row selection for Left/Full/LeftSemi/LeftAnti: present & (visited | ~visited)
#ifndef GPUQ_JIT
__global__ void __launch_bounds__(HBLOCK) k_bitmap_select(const u64* __restrict__ present, const u64* __restrict__ visited, const int matched,
                                                          const i64 nwords, const i64 n, u64* __restrict__ bitmap,
                                                          uint32_t* __restrict__ block_counts, const i64 wpb) {
  __shared__ uint32_t wc[HWAVES];
  const i64 w0 = (i64)blockIdx.x * wpb;
  i64 w1 = w0 + wpb; if (w1 > nwords) w1 = nwords;
  uint32_t cnt = 0;
  for (i64 w = w0 + threadIdx.x; w < w1; w += HBLOCK) {
    u64 m = present[w] & (matched ? visited[w] : ~visited[w]);
    if (w == nwords - 1 && (n & 63)) m &= (1ull << (n & 63)) - 1;
    bitmap[w] = m; cnt += (uint32_t)__popcll(m);
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (hlane() == 0) wc[hwave()] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < HWAVES; ++k) t += wc[k]; block_counts[blockIdx.x] = t; }
}
#endif
void launch_bitmap_select(hipStream_t s, const u64* present, const u64* visited, int matched, i64 nwords, i64 n, u64* bitmap,
                          uint32_t* block_counts, int nblocks, i64 wpb) {
  hipLaunchKernelGGL(k_bitmap_select, dim3(nblocks), dim3(HBLOCK), 0, s, present, visited, matched, nwords, n, bitmap, block_counts, wpb);
}
void launch_join_probe(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, const uint32_t* next,
                       int join_type, int payload_via, int null_equals_null, uint32_t* out_build, uint32_t* out_probe,
                       u64 out_cap, u64* out_count, uint32_t* visited) {
  if (n <= 0) return;
  if (jit_override().fn && jit_override().kernel_id == 6) {
    (void)jit_launch(jit_override().fn, dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, T, next, join_type, payload_via, null_equals_null, out_build, out_probe, out_cap, out_count, visited);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_join_probe<M>, dim3(hgrid(n, 8)), dim3(HBLOCK), 0, s, P, n, K, T, next, join_type, payload_via, null_equals_null,                                     out_build, out_probe, out_cap, out_count, visited)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}

void launch_join_keyrange(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, int null_equals_null, u64* out, i64 wstep) {
  if (n <= 0) return;
  if (wstep < 1) wstep = 1;
  const int grid = hgrid((n + wstep - 1) / wstep, 8);
  if (jit_override().fn && jit_override().kernel_id == 14) {
    (void)jit_launch(jit_override().fn, dim3(grid), dim3(HBLOCK), 0, s, P, n, K, null_equals_null, out, wstep);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_join_keyrange<M>, dim3(grid), dim3(HBLOCK), 0, s, P, n, K, null_equals_null, out, wstep)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
void launch_join_probe_unique(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, int join_type, int null_equals_null, int payload_via,
                              uint32_t* seg_build, uint32_t* seg_probe, uint32_t* seg_counts, int nsegs, i64 wpw, uint32_t* visited) {
  const int nblocks = (nsegs + HWAVES - 1) / HWAVES;
  if (jit_override().fn && jit_override().kernel_id == 7) {
    (void)jit_launch(jit_override().fn, dim3(nblocks), dim3(HBLOCK), 0, s, P, n, K, T, join_type, null_equals_null, payload_via, seg_build, seg_probe, seg_counts, nsegs, wpw, visited);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_join_probe_unique<M>, dim3(nblocks), dim3(HBLOCK), 0, s, P, n, K, T, join_type, null_equals_null, payload_via, seg_build, seg_probe, seg_counts, nsegs, wpw, visited)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
}
void launch_copy_segments(hipStream_t s, const uint32_t* seg_build, const uint32_t* seg_probe, const uint32_t* seg_offsets, int nsegs, i64 wpw, i64 n,
                          const u64* total, uint32_t* out_build, uint32_t* out_probe, u64 out_cap, uint32_t* flags) {
  (void)n;
  i64 need = ((i64)nsegs + HWAVES - 1) / HWAVES; const i64 cap = (i64)num_cus() * 8;
  hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)(need < cap ? (need ? need : 1) : cap)), dim3(HBLOCK), 0, s, seg_build, seg_probe, seg_offsets, nsegs, wpw, total,
                     out_build, out_probe, out_cap, flags);
}

void launch_join_locality(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, i64 stride, i64 nsample, u64* out) {
  if (n <= 0 || nsample <= 0) return;
  const int grid = (int)std::min<i64>((nsample + HWAVES - 1) / HWAVES, (i64)num_cus() * 4);
#define CALL(M) hipLaunchKernelGGL(k_join_locality<M>, dim3(grid), dim3(HBLOCK), 0, s, P, n, K, T, stride < 1 ? 1 : stride, nsample, out)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
}
// ---- partitioned probe
void rj_geometry(i64 n, u64 range, int slice_log2, RjGeomHost* g) {
  // slices of 2^slice_log2 table entries, at most RJ_MAX_PARTS of them
  int sh = slice_log2; while (((range + ((1ull << sh) - 1)) >> sh) > RJ_MAX_PARTS) ++sh;
  g->shift = (uint32_t)sh; g->nparts = (uint32_t)((range + ((1ull << sh) - 1)) >> sh);
  i64 t = RJ_TILE; const i64 maxb = (i64)num_cus() * 8;
  while ((n + t - 1) / t > maxb) t += RJ_TILE;
  g->tile = t; g->nblocks = (int32_t)((n + t - 1) / t); if (g->nblocks < 1) g->nblocks = 1;
}
size_t rj_hist_entries(const RjGeomHost& g) { return (size_t)(g.nparts + 1) * g.nblocks + 1; }
void launch_rj_partition(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, int payload_via, const RjGeomHost& g,
                         u64* rec, u64* rec_out, int32_t* hist, void* scan_ws, size_t scan_ws_bytes) {
  RjGeom G{g.nparts, g.shift, g.tile, g.nblocks, 0};
  if (jit_override().fn && jit_override().kernel_id == 15) {
    (void)jit_launch(jit_override().fn, dim3(g.nblocks), dim3(HBLOCK), 0, s, P, n, K, T, payload_via, G, rec, hist);
  } else {
#define CALL(M) hipLaunchKernelGGL(k_rj_pack<M>, dim3(g.nblocks), dim3(HBLOCK), 0, s, P, n, K, T, payload_via, G, rec, hist)
  GPUQ_DISPATCH_MAXC(P.n_cols, CALL);
#undef CALL
  }
  launch_exclusive_scan_i32(s, hist, (i64)(g.nparts + 1) * g.nblocks, scan_ws, scan_ws_bytes);
  hipLaunchKernelGGL(k_rj_scatter, dim3(g.nblocks), dim3(HBLOCK), 0, s, (const u64*)rec, n, G, (const int32_t*)hist, rec_out);
}
// grid = what is resident at once (the group's waves must advance together); *wpw_out = scratch words per wave
int rj_probe_geometry(i64 n, i64* wpw_out) {
  const i64 nwords = (n + 63) >> 6;
  i64 nblocks = (i64)num_cus() * 6;                    // 6 blocks of 256 threads per CU fit the kernel's registers / SGPRs
  const i64 need = (nwords + 4 * HWAVES - 1) / (4 * HWAVES);
  if (nblocks > need) nblocks = need < 8 ? 8 : need;
  const i64 nbx_min = nblocks / 8 > 0 ? nblocks / 8 : 1;                       // smallest group
  const i64 gw = nwords / 8 + 1;                                                // words of the largest group
  const i64 steps = (gw + nbx_min * HWAVES * 4 - 1) / (nbx_min * HWAVES * 4);   // 4 words per wave and step
  *wpw_out = steps * 4;
  return (int)nblocks;
}
void launch_rj_probe(hipStream_t s, const u64* rec, const int32_t* n_live, const HashTable& T, int join_type, uint32_t* seg_build, uint32_t* seg_probe,
                     uint32_t* seg_counts, int nblocks, i64 wpw) {
  hipLaunchKernelGGL(k_rj_probe, dim3(nblocks), dim3(HBLOCK), 0, s, rec, n_live, T, join_type, seg_build, seg_probe, seg_counts, nblocks * HWAVES, wpw);
}

#endif  // GPUQ_JIT

}  // namespace gpuq
