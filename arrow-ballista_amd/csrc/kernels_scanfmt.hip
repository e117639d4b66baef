// Scan-side decode on the device: delimited text (CsvExec) and Parquet pages (ParquetExec).
//
// What it replaces on the reference's path: the leaves of every TPC-H plan -- `register_csv` / `register_parquet` tables in
// benchmarks/src/bin/tpch.rs:801-862, i.e. DataFusion's CsvExec / ParquetExec decoding files into RecordBatches on the host.
// With those on the host every query pays PCIe for the DECODED columns; here the FILE bytes cross PCIe once (they are 2-4 x
// smaller than Arrow columns for text numerics and dictionary-encoded Parquet) and the columns are produced in HBM.
// Byte / integer work, latency- and PCIe-bound: no roofline claim beyond "faster than the link feeds it".
#include "gpuq_kernels.h"

namespace gpuq {

constexpr int FBLOCK = 256;
constexpr int FWAVES = FBLOCK / 64;
__device__ __forceinline__ int flane() { return threadIdx.x & 63; }
__device__ __forceinline__ int fwave() { return threadIdx.x >> 6; }

// ------------------------------------------------------------------ delimited text
// 1. line ends: every block counts the '\n' bytes of its chunk; after a scan of the counts k_csv_line_starts writes the start
//    offset of every line (ordered ballot compaction inside a wave, waves of a block in order).
// 0x80 in every byte of w that equals '\n' (exact per byte: no borrow between bytes)
__device__ __forceinline__ uint32_t nl_mask(uint32_t w) {
  const uint32_t x = w ^ 0x0A0A0A0Au;
  const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}
// the 16 bytes at text[i .. i+16) of a 16-byte aligned i, bytes at or beyond `b` (and before `a`) masked out; bit k = byte k is '\n'
__device__ __forceinline__ uint32_t nl_bits16(const uint8_t* __restrict__ text, i64 i, i64 a, i64 b) {
  const uint4 v = *(const uint4*)(text + i);
  const uint32_t m0 = nl_mask(v.x), m1 = nl_mask(v.y), m2 = nl_mask(v.z), m3 = nl_mask(v.w);
  // gather the 0x80 flags of each word into 4 bits
  auto pack = [](uint32_t m) -> uint32_t { return ((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u); };
  uint32_t bits = pack(m0) | (pack(m1) << 4) | (pack(m2) << 8) | (pack(m3) << 12);
  if (i < a) bits &= ~((1u << (int)(a - i)) - 1u);
  if (i + 16 > b) bits &= (b > i) ? ((1u << (int)(b - i)) - 1u) : 0u;
  return bits;
}
// chunks start on 16-byte boundaries (chunk is a multiple of 16, the text buffer is 256-byte aligned with >= 64 bytes of slack)
__global__ void __launch_bounds__(FBLOCK) k_csv_count_lines(const uint8_t* __restrict__ text, const i64 n, const i64 chunk, uint32_t* __restrict__ counts) {
  __shared__ uint32_t wc[FWAVES];
  const i64 a = (i64)blockIdx.x * chunk; i64 b = a + chunk; if (b > n) b = n;
  uint32_t c = 0;
  for (i64 i = a + (i64)threadIdx.x * 16; i < b; i += FBLOCK * 16) c += (uint32_t)__popc(nl_bits16(text, i, a, b));
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (flane() == 0) wc[fwave()] = c;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < FWAVES; ++k) t += wc[k]; counts[blockIdx.x] = t; }
}
// starts[r + 1] = offset of the byte after the r-th '\n' (starts[0] = 0 is set by the host): one wave walks the block's chunk,
// 16 bytes per lane and step; the lanes' newline counts are prefix-summed across the wave to place each line start in order
__global__ void __launch_bounds__(64) k_csv_line_starts(const uint8_t* __restrict__ text, const i64 n, const i64 chunk, const uint32_t* __restrict__ block_offsets,
                                                        i64* __restrict__ starts) {
  const i64 a = (i64)blockIdx.x * chunk; i64 b = a + chunk; if (b > n) b = n;
  u64 out = (u64)block_offsets[blockIdx.x] + 1;
  for (i64 i0 = a; i0 < b; i0 += 64 * 16) {
    const i64 i = i0 + (i64)flane() * 16;
    uint32_t bits = i < b ? nl_bits16(text, i, a, b) : 0u;
    const uint32_t cnt = (uint32_t)__popc(bits);
    uint32_t incl = cnt;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (flane() >= o) incl += y; }
    u64 at = out + (u64)(incl - cnt);
    while (bits) { const int k = __ffs((int)bits) - 1; bits &= bits - 1; starts[at++] = i + k + 1; }
    out += (u64)__shfl(incl, 63);
  }
}

// days since 1970-01-01 of a proleptic Gregorian date [UPSTREAM-KNOWLEDGE: chrono / Howard Hinnant's days_from_civil]
__device__ __forceinline__ int32_t days_from_civil(int y, int m, int d) {
  y -= m <= 2;
  const int era = (y >= 0 ? y : y - 399) / 400;
  const unsigned yoe = (unsigned)(y - era * 400);
  const unsigned doy = (153u * (unsigned)(m + (m > 2 ? -3 : 9)) + 2u) / 5u + (unsigned)d - 1u;
  const unsigned doe = yoe * 365u + yoe / 4u - yoe / 100u + doy;
  return era * 146097 + (int)doe - 719468;
}

// 2. one lane per line: walk the line's bytes once, parse the projected fields in place.  Strings only record (start, length);
//    their bytes are copied after the lengths have been scanned into offsets.
//    The 256 lines of a block are one contiguous byte range: it is staged into LDS with coalesced 16-byte loads (lane-per-line
//    byte loads from global memory touch a different cache line per lane and step); ranges beyond the LDS budget are read in place.
constexpr int CSV_LDS_BYTES = 40 * 1024;
struct CsvText {
  const uint8_t* g; const uint8_t* l; i64 base; bool staged;
  __device__ __forceinline__ uint8_t operator[](i64 i) const { return staged ? l[i - base] : g[i]; }
};
__global__ void __launch_bounds__(FBLOCK) k_csv_parse(const uint8_t* __restrict__ gtext, const i64 n_bytes, const i64* __restrict__ starts, const i64 row0, const i64 n_rows,
                                                      const CsvSpec S, const CsvOut O, uint32_t* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[CSV_LDS_BYTES];
  for (i64 rb = (i64)blockIdx.x * FBLOCK; rb < n_rows; rb += (i64)gridDim.x * FBLOCK) {
    const i64 r = rb + threadIdx.x;
    const bool live = r < n_rows;
    const i64 rlast = rb + FBLOCK < n_rows ? rb + FBLOCK : n_rows;
    const i64 lo = starts[row0 + rb] & ~(i64)15, hi = starts[row0 + rlast];      // block-uniform
    CsvText text{gtext, lds, lo, hi - lo <= (i64)CSV_LDS_BYTES};
    __syncthreads();                                                              // the previous iteration's readers are done
    if (text.staged) {
      for (i64 i = lo + (i64)threadIdx.x * 16; i < hi; i += FBLOCK * 16) *(uint4*)(lds + (i - lo)) = *(const uint4*)(gtext + i);
      __syncthreads();
    }
    i64 p = 0, end = 0;
    if (live) {
      p = starts[row0 + r]; end = starts[row0 + r + 1] - 1;            // end = position of the '\n' (or of the end of the text)
      if (end > p && text[end - 1] == '\r') --end;
    }
    uint32_t myflags = 0;
    for (int f = 0; f < S.n_fields; ++f) {
      // field = [p, q)
      i64 q = p;
      if (live) {
        while (q < end && text[q] != S.delim) { if (text[q] == S.quote) myflags |= CSVF_QUOTE; ++q; }
      }
      const int kind = S.kind[f];
      if (kind != CSV_SKIP) {
        const int oc = S.out[f];
        const bool empty = q == p;
        bool valid = live && !(empty && (kind != CSV_UTF8 || S.nullable[f]));      // arrow-csv: an empty field of a nullable column is NULL
        if (live && empty && kind != CSV_UTF8 && !S.nullable[f]) myflags |= CSVF_NULL_IN_REQUIRED;
        if (live) {
          switch (kind) {
            case CSV_I32: case CSV_I64: {
              i64 v = 0; bool neg = false; i64 k = p;
              if (k < q && (text[k] == '-' || text[k] == '+')) { neg = text[k] == '-'; ++k; }
              if (k == q && !empty) myflags |= CSVF_BAD_NUMBER;
              // accumulated as a magnitude with an overflow check: a field that does not fit its column's type is an error
              // (arrow-csv and pyarrow refuse it too), not a wrapped value
              u64 mag = 0; bool ovf = false;
              for (; k < q; ++k) {
                const int dg = (int)text[k] - '0'; if (dg < 0 || dg > 9) { myflags |= CSVF_BAD_NUMBER; break; }
                if (mag > (0xFFFFFFFFFFFFFFFFull - (u64)dg) / 10) ovf = true;
                mag = mag * 10 + (u64)dg;
              }
              const u64 lim = kind == CSV_I32 ? (neg ? 0x80000000ull : 0x7FFFFFFFull) : (neg ? 0x8000000000000000ull : 0x7FFFFFFFFFFFFFFFull);
              if (ovf || mag > lim) myflags |= CSVF_BAD_NUMBER;
              v = neg ? (i64)(0 - mag) : (i64)mag;
              if (kind == CSV_I32) ((int32_t*)O.data[oc])[r] = (int32_t)v; else ((i64*)O.data[oc])[r] = v;
              break;
            }
            case CSV_DATE32: {      // yyyy-mm-dd
              int32_t days = 0;
              if (!empty) {
                if (q - p != 10 || text[p + 4] != '-' || text[p + 7] != '-') myflags |= CSVF_BAD_NUMBER;
                else {
                  const int y = (text[p] - '0') * 1000 + (text[p + 1] - '0') * 100 + (text[p + 2] - '0') * 10 + (text[p + 3] - '0');
                  const int m = (text[p + 5] - '0') * 10 + (text[p + 6] - '0'), d = (text[p + 8] - '0') * 10 + (text[p + 9] - '0');
                  days = days_from_civil(y, m, d);
                }
              }
              ((int32_t*)O.data[oc])[r] = days;
              break;
            }
            case CSV_DEC128: {      // [-]digits[.digits] -> unscaled integer at the column's scale (extra fraction digits are an error, missing ones are zeros)
              i128 v = 0; bool neg = false; i64 k = p; int frac = -1;
              if (k < q && (text[k] == '-' || text[k] == '+')) { neg = text[k] == '-'; ++k; }
              for (; k < q; ++k) {
                if (text[k] == '.') { if (frac >= 0) myflags |= CSVF_BAD_NUMBER; frac = 0; continue; }
                const int dg = (int)text[k] - '0'; if (dg < 0 || dg > 9) { myflags |= CSVF_BAD_NUMBER; break; }
                v = v * 10 + dg; if (frac >= 0) ++frac;
              }
              if (frac < 0) frac = 0;
              if (frac > S.scale[f]) myflags |= CSVF_BAD_NUMBER;
              for (int z = frac; z < S.scale[f]; ++z) v *= 10;
              // more digits than the column's precision (or than 38: the accumulator would have wrapped) is an error
              { i128 lim10 = 1; for (int z = 0; z < S.prec[f]; ++z) lim10 *= 10; if (q - p > 40 || v >= lim10) myflags |= CSVF_BAD_NUMBER; }
              if (neg) v = -v;
              ((u64*)O.data[oc])[2 * r] = (u64)v; ((u64*)O.data[oc])[2 * r + 1] = (u64)((u128)v >> 64);
              break;
            }
            case CSV_F64: {
              // Clinger's exact fast path only: <= 15 significant digits and |decimal exponent| <= 22 give the correctly rounded
              // double with one multiplication or division; anything longer raises CSVF_FLOAT_PRECISION (refused loudly)
              u64 w = 0; int digits = 0, e10 = 0; bool neg = false, seen_dot = false; i64 k = p;
              if (k < q && (text[k] == '-' || text[k] == '+')) { neg = text[k] == '-'; ++k; }
              for (; k < q; ++k) {
                const uint8_t ch = text[k];
                if (ch == '.') { if (seen_dot) myflags |= CSVF_BAD_NUMBER; seen_dot = true; continue; }
                if (ch == 'e' || ch == 'E') break;
                const int dg = (int)ch - '0'; if (dg < 0 || dg > 9) { myflags |= CSVF_BAD_NUMBER; break; }
                if (w != 0 || dg != 0) { if (digits < 19) { w = w * 10 + (u64)dg; ++digits; } else if (!seen_dot) ++e10, myflags |= CSVF_FLOAT_PRECISION; else myflags |= (dg ? CSVF_FLOAT_PRECISION : 0u); }
                if (seen_dot) --e10;
              }
              if (k < q && (text[k] == 'e' || text[k] == 'E')) {
                ++k; bool eneg = false; int ev = 0;
                if (k < q && (text[k] == '-' || text[k] == '+')) { eneg = text[k] == '-'; ++k; }
                for (; k < q; ++k) { const int dg = (int)text[k] - '0'; if (dg < 0 || dg > 9) { myflags |= CSVF_BAD_NUMBER; break; } ev = ev * 10 + dg; }
                e10 += eneg ? -ev : ev;
              }
              double v = (double)w;
              if (w >= (1ull << 53) || e10 > 22 || e10 < -22) { if (w != 0) myflags |= CSVF_FLOAT_PRECISION; }
              else {
                double p10 = 1.0; for (int z = 0; z < (e10 < 0 ? -e10 : e10); ++z) p10 *= 10.0;      // exact: 10^k for k <= 22
                v = e10 < 0 ? v / p10 : v * p10;
              }
              if (neg) v = -v;
              ((double*)O.data[oc])[r] = v;
              break;
            }
            case CSV_UTF8: O.str_start[oc][r] = (uint32_t)p; O.str_len[oc][r] = (int32_t)(q - p); break;
            default: break;
          }
        }
        if (kind == CSV_BOOL) {
          const bool t = live && (q - p == 4) && (text[p] == 't' || text[p] == 'T');
          if (live && !empty && !t && !((q - p == 5) && (text[p] == 'f' || text[p] == 'F'))) myflags |= CSVF_BAD_NUMBER;
          const u64 m = __ballot(t);
          if (flane() == 0 && live) ((u64*)O.data[oc])[r >> 6] = m;
        }
        if (O.valid[oc]) { const u64 m = __ballot(valid); if (flane() == 0 && live) O.valid[oc][r >> 6] = m; }
      }
      if (live) {
        if (q < end) p = q + 1;
        else { if (f + 1 < S.n_fields) myflags |= CSVF_FIELD_COUNT; p = q; }
      }
    }
    if (live && p < end) myflags |= CSVF_FIELD_COUNT;      // more fields than the schema has (a trailing delimiter counts as consumed above)
    if (myflags) atomicOr(flags, myflags);
  }
}
__global__ void __launch_bounds__(FBLOCK) k_csv_copy_strings(const uint8_t* __restrict__ text, const uint32_t* __restrict__ start, const int32_t* __restrict__ offsets,
                                                             const i64 n, uint8_t* __restrict__ out) {
  for (i64 r = (i64)blockIdx.x * FBLOCK + threadIdx.x; r < n; r += (i64)gridDim.x * FBLOCK) {
    const int32_t o = offsets[r], len = offsets[r + 1] - o;
    const uint8_t* s = text + start[r];
    for (int32_t k = 0; k < len; ++k) out[o + k] = s[k];
  }
}

// ------------------------------------------------------------------ Parquet pages
// The host walks the Thrift metadata (footer, page headers: a few hundred bytes each) and hands the device one descriptor per
// data page; ONE WAVE decodes one page.  Pages of a column chunk are independent once the running value / row offsets are
// known (the host sums the page headers' value counts), so a file's thousands of pages decode concurrently.
struct BitReader { const uint8_t* p; const uint8_t* end; };
__device__ __forceinline__ uint32_t pq_varint(BitReader& b) {
  uint32_t v = 0; int sh = 0;
  while (b.p < b.end) { const uint8_t c = *b.p++; v |= (uint32_t)(c & 0x7F) << sh; if (!(c & 0x80)) break; sh += 7; if (sh > 28) break; }
  return v;
}
// value `i` of a bit-packed run (LSB first), `bw` bits per value
__device__ __forceinline__ uint32_t pq_unpack(const uint8_t* base, const uint8_t* end, int bw, uint32_t i) {
  const u64 bit = (u64)i * (u64)bw; const uint8_t* q = base + (bit >> 3);
  u64 w = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) if (q + k < end) w |= (u64)q[k] << (8 * k);
  return (uint32_t)((w >> (bit & 7)) & ((bw >= 32) ? 0xFFFFFFFFull : ((1ull << bw) - 1)));
}
// Decode an RLE / bit-packed hybrid stream of `n` values into out[0..n) (LDS or global), the whole wave cooperating on each run.
template <class Sink>
__device__ __forceinline__ bool pq_hybrid(const uint8_t* p, const uint8_t* end, int bw, int n, Sink sink) {
  BitReader b{p, end};
  int done = 0;
  while (done < n) {
    if (b.p >= b.end) return false;
    const uint32_t h = pq_varint(b);
    if (h & 1) {      // bit-packed: (h >> 1) groups of 8 values
      const i64 cnt = (i64)(h >> 1) * 8;      // 64-bit: a corrupt header must not wrap into a negative count
      if (cnt == 0) return false;
      const int take = cnt < (i64)(n - done) ? (int)cnt : n - done;
      for (int i = flane(); i < take; i += 64) sink(done + i, pq_unpack(b.p, b.end, bw, (uint32_t)i));
      const i64 adv = (cnt * bw + 7) / 8;
      b.p = adv < b.end - b.p ? b.p + adv : b.end; done += take;
    } else {          // run of one value, (bw + 7) / 8 bytes little-endian
      const i64 cnt = (i64)(h >> 1);
      if (cnt == 0) return false;
      uint32_t v = 0; const int nb = (bw + 7) / 8;
      for (int k = 0; k < nb; ++k) if (b.p + k < b.end) v |= (uint32_t)b.p[k] << (8 * k);
      b.p += nb;
      const int take = cnt < (i64)(n - done) ? (int)cnt : n - done;
      for (int i = flane(); i < take; i += 64) sink(done + i, v);
      done += take;
    }
  }
  return true;
}

// one wave per page
// INT96 (Impala / Spark timestamps): 8 bytes nanoseconds of the day, 4 bytes Julian day, little-endian -> nanoseconds since 1970-01-01
// (Julian day 2440588), as parquet-rs' Int96::to_nanos [UPSTREAM-KNOWLEDGE]
__device__ __forceinline__ i64 pq_int96_ns(const uint8_t* q) {
  i64 nanos; int32_t jd; __builtin_memcpy(&nanos, q, 8); __builtin_memcpy(&jd, q + 8, 4);
  return ((i64)jd - 2440588) * 86400000000000ll + nanos;
}
__global__ void __launch_bounds__(64) k_pq_decode(const uint8_t* __restrict__ file, const i64 file_bytes, const PqPage* __restrict__ pages, const int n_pages, const PqCol C,
                                                  const PqDict* __restrict__ dicts, const uint8_t* __restrict__ dict_values, const int32_t* __restrict__ dict_str_offsets,
                                                  uint32_t* __restrict__ scratch /* per page: n_values u32 (value index of every row, or NIL) */, const i64 scratch_stride,
                                                  uint32_t* __restrict__ flags) {
  const int pg = blockIdx.x;
  if (pg >= n_pages) return;
  const PqPage P = pages[pg];
  if (P.src < 0 || P.src + P.bytes > file_bytes) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
  const uint8_t* p = file + P.src; const uint8_t* end = p + P.bytes;
  uint32_t* vidx = scratch + (size_t)pg * (size_t)scratch_stride;      // row -> index among the page's non-null values
  const int n = P.n_values;
  // ---- definition levels -> validity + value index
  int n_present = n;
  if (C.optional) {
    const uint8_t* dp = p; const uint8_t* dend;
    if (P.def_v2 > 0) { dend = p + P.def_v2; p = dend; }
    else {
      if (end - p < 4) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
      const uint32_t len = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
      dp = p + 4; dend = dp + len; p = dend;
    }
    if (dend > end) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
    const bool ok = pq_hybrid(dp, dend, 1, n, [&](int i, uint32_t v) { vidx[i] = v; });
    if (!ok) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    // exclusive prefix of the levels = value index; validity bits by ballot (rows of a page start at any bit: atomicOr)
    uint32_t run = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + flane();
      const uint32_t lv = i < n ? vidx[i] : 0u;
      const u64 m = __ballot(lv != 0);
      const uint32_t mine = run + (uint32_t)__popcll(m & ((1ull << flane()) - 1));
      if (i < n) vidx[i] = lv ? mine : NULL_ROW;
      if (C.valid && m) {
        const i64 bit0 = P.row0 + i0; const int sh = (int)(bit0 & 63);
        if (flane() == 0) { atomicOr((unsigned long long*)&C.valid[bit0 >> 6], (unsigned long long)(m << sh)); if (sh) atomicOr((unsigned long long*)&C.valid[(bit0 >> 6) + 1], (unsigned long long)(m >> (64 - sh))); }
      }
      run += (uint32_t)__popcll(m);
    }
    n_present = (int)run;
    __builtin_amdgcn_wave_barrier();
  }
  // ---- values
  const i64 voff = p - file;      // payload position of the values
  if (P.enc == PQE_RLE) {      // v2 pages write BOOLEAN values as a length-prefixed hybrid stream of 1-bit values
    if (C.phys != PQ_BOOL || end - p < 4) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
    const uint32_t len = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    p += 4;
    if ((i64)len > end - p) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
    uint32_t* bits = vidx + n;
    if (!pq_hybrid(p, p + len, 1, n_present, [&](int i, uint32_t v) { bits[i] = v; })) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int i = flane(); i < n; i += 64) {
      const uint32_t vi = C.optional ? vidx[i] : (uint32_t)i;
      if (vi == NULL_ROW) continue;
      const i64 r = P.row0 + i;
      if (bits[vi] & 1u) atomicOr((unsigned long long*)&((u64*)C.data)[r >> 6], 1ull << (r & 63));
    }
    return;
  }
  if (P.enc == PQE_DICT) {
    if (P.dict < 0 || p >= end) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
    const PqDict D = dicts[P.dict];
    const int bw = (int)*p++;
    if (bw > 32) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
    // indices of the present values, decoded into the tail of the scratch row (n_present <= n)
    uint32_t* ix = vidx + n;
    const bool ok = bw == 0 ? true : pq_hybrid(p, end, bw, n_present, [&](int i, uint32_t v) { ix[i] = v; });
    if (bw == 0) for (int i = flane(); i < n_present; i += 64) ix[i] = 0;
    if (!ok) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int i = flane(); i < n; i += 64) {
      const uint32_t vi = C.optional ? vidx[i] : (uint32_t)i;
      const i64 r = P.row0 + i;
      if (vi == NULL_ROW) { if (C.phys == PQ_BYTE_ARRAY) { C.str_len[r] = 0; C.str_src[r] = -1; } continue; }
      const uint32_t k = ix[vi];
      if ((int)k >= D.n) { atomicOr(flags, PQF_MALFORMED); continue; }
      if (C.phys == PQ_BYTE_ARRAY) { const int32_t* so = dict_str_offsets + D.str_offsets; C.str_len[r] = so[k + 1] - so[k]; C.str_src[r] = -(2 + (D.values + so[k])); }   // negative: position in the dictionary bytes
      else if (C.width == 1) ((uint8_t*)C.data)[r] = ((const uint8_t*)(dict_values + D.values))[k];
      else if (C.width == 2) ((uint16_t*)C.data)[r] = ((const uint16_t*)(dict_values + D.values))[k];
      else if (C.width == 4) ((uint32_t*)C.data)[r] = ((const uint32_t*)(dict_values + D.values))[k];
      else if (C.width == 8) ((u64*)C.data)[r] = ((const u64*)(dict_values + D.values))[k];
      else { ((u64*)C.data)[2 * r] = ((const u64*)(dict_values + D.values))[2 * k]; ((u64*)C.data)[2 * r + 1] = ((const u64*)(dict_values + D.values))[2 * k + 1]; }
    }
    return;
  }
  // PLAIN
  if (C.phys == PQ_BYTE_ARRAY) {
    // length-prefixed values: a serial walk (lane 0) over the page's present values records where each one starts
    i64* starts = (i64*)(vidx + n + (n & 1));      // 8-byte aligned tail of the scratch row: n_present positions
    int walk_bad = 0;
    if (flane() == 0) {
      const uint8_t* q = p;
      for (int i = 0; i < n_present; ++i) {
        if (end - q < 4) { walk_bad = 1; break; }
        const uint32_t len = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        starts[i] = (q + 4) - file; q += 4 + (i64)len;
        if (q > end) { walk_bad = 1; break; }
      }
      if (walk_bad) atomicOr(flags, PQF_MALFORMED);
    }
    if (__shfl(walk_bad, 0)) return;      // the positions beyond the break are not written: nobody may follow them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int i = flane(); i < n; i += 64) {
      const uint32_t vi = C.optional ? vidx[i] : (uint32_t)i;
      const i64 r = P.row0 + i;
      if (vi == NULL_ROW) { C.str_len[r] = 0; C.str_src[r] = -1; continue; }
      const i64 s0 = starts[vi];
      const uint8_t* q = file + s0 - 4;
      C.str_len[r] = (int32_t)((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24));
      C.str_src[r] = s0;
    }
    return;
  }
  {      // the page must hold every value its header and levels announce: nothing below reads past `end`
    const i64 have = end - p;
    const i64 need = C.phys == PQ_BOOL ? ((i64)n_present + 7) / 8 : (i64)n_present * ((C.phys == PQ_I32 || C.phys == PQ_F32) ? 4 : (C.phys == PQ_FLBA ? C.flba_len : (C.phys == PQ_I96 ? 12 : 8)));
    if (need > have) { if (flane() == 0) atomicOr(flags, PQF_MALFORMED); return; }
  }
  for (int i = flane(); i < n; i += 64) {
    const uint32_t vi = C.optional ? vidx[i] : (uint32_t)i;
    if (vi == NULL_ROW) continue;
    const i64 r = P.row0 + i;
    switch (C.phys) {
      case PQ_BOOL: { const uint8_t byte = file[voff + (vi >> 3)]; if ((byte >> (vi & 7)) & 1) atomicOr((unsigned long long*)&((u64*)C.data)[r >> 6], 1ull << (r & 63)); break; }
      case PQ_I96: ((i64*)C.data)[r] = pq_int96_ns(file + voff + (i64)vi * 12); break;
      case PQ_F32: { uint32_t v; __builtin_memcpy(&v, file + voff + (i64)vi * 4, 4); ((uint32_t*)C.data)[r] = v; break; }
      case PQ_I32: { uint32_t v; __builtin_memcpy(&v, file + voff + (i64)vi * 4, 4); if (C.width == 4) ((uint32_t*)C.data)[r] = v; else if (C.width == 1) ((uint8_t*)C.data)[r] = (uint8_t)v; else if (C.width == 2) ((uint16_t*)C.data)[r] = (uint16_t)v; else { const i64 w = (int32_t)v; if (C.width == 8) ((i64*)C.data)[r] = w; else { ((u64*)C.data)[2 * r] = (u64)w; ((u64*)C.data)[2 * r + 1] = (u64)(w >> 63); } } break; }
      case PQ_I64: case PQ_F64: { u64 v; __builtin_memcpy(&v, file + voff + (i64)vi * 8, 8); if (C.width == 8) ((u64*)C.data)[r] = v; else { ((u64*)C.data)[2 * r] = v; ((u64*)C.data)[2 * r + 1] = (u64)((i64)v >> 63); } break; }
      case PQ_FLBA: {      // big-endian two's complement decimal of flba_len bytes -> little-endian 128 bits
        const uint8_t* q = file + voff + (i64)vi * C.flba_len;
        u128 v = (q[0] & 0x80) ? ~(u128)0 : 0;
        for (int k = 0; k < C.flba_len; ++k) v = (v << 8) | q[k];
        ((u64*)C.data)[2 * r] = (u64)v; ((u64*)C.data)[2 * r + 1] = (u64)(v >> 64);
        break;
      }
      default: atomicOr(flags, PQF_UNSUPPORTED); break;
    }
  }
}
// strings: bytes from the file copy (src >= 0) or from the decoded dictionary (src <= -2: position = -(src + 2))
__global__ void __launch_bounds__(FBLOCK) k_pq_copy_strings(const uint8_t* __restrict__ file, const uint8_t* __restrict__ dict_values, const i64* __restrict__ src,
                                                            const int32_t* __restrict__ offsets, const i64 n, uint8_t* __restrict__ out) {
  for (i64 r = (i64)blockIdx.x * FBLOCK + threadIdx.x; r < n; r += (i64)gridDim.x * FBLOCK) {
    const int32_t o = offsets[r], len = offsets[r + 1] - o;
    if (len <= 0) continue;
    const i64 s = src[r];
    const uint8_t* q = s >= 0 ? file + s : dict_values + (-(s + 2));
    for (int32_t k = 0; k < len; ++k) out[o + k] = q[k];
  }
}
// a PLAIN dictionary page of BYTE_ARRAY values: serial walk by one lane -> offsets (n + 1) and contiguous bytes
__global__ void __launch_bounds__(64) k_pq_dict_strings(const uint8_t* __restrict__ file, const i64 src, const int32_t bytes, const int32_t n, int32_t* __restrict__ offsets,
                                                        uint8_t* __restrict__ out, uint32_t* __restrict__ flags) {
  if (threadIdx.x != 0) return;
  const uint8_t* q = file + src; const uint8_t* end = q + bytes;
  int32_t at = 0;
  for (int i = 0; i < n; ++i) {
    if (end - q < 4) { atomicOr(flags, PQF_MALFORMED); break; }
    const uint32_t len = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    q += 4;
    if ((i64)len > end - q) { atomicOr(flags, PQF_MALFORMED); break; }
    offsets[i] = at;
    for (uint32_t k = 0; k < len; ++k) out[at + (int32_t)k] = q[k];
    at += (int32_t)len; q += len;
  }
  offsets[n] = at;
}
// fixed-width dictionary values -> the output width (INT32 -> 4 / 8 / 16, INT64 -> 8 / 16, FLBA -> 16)
__global__ void __launch_bounds__(FBLOCK) k_pq_dict_fixed(const uint8_t* __restrict__ file, const i64 src, const int32_t n, const int32_t phys, const int32_t flba_len,
                                                          const int32_t width, uint8_t* __restrict__ out) {
  for (int i = blockIdx.x * FBLOCK + threadIdx.x; i < n; i += gridDim.x * FBLOCK) {
    u128 v = 0;
    if (phys == PQ_I32 || phys == PQ_F32) { int32_t x; __builtin_memcpy(&x, file + src + (i64)i * 4, 4); v = (u128)(i128)x; }
    else if (phys == PQ_I96) v = (u128)(i128)pq_int96_ns(file + src + (i64)i * 12);
    else if (phys == PQ_I64 || phys == PQ_F64) { i64 x; __builtin_memcpy(&x, file + src + (i64)i * 8, 8); v = (u128)(i128)x; }
    else { const uint8_t* q = file + src + (i64)i * flba_len; v = (q[0] & 0x80) ? ~(u128)0 : 0; for (int k = 0; k < flba_len; ++k) v = (v << 8) | q[k]; }
    if (width == 4) ((uint32_t*)out)[i] = (uint32_t)v;
    else if (width == 1) out[i] = (uint8_t)v;
    else if (width == 2) ((uint16_t*)out)[i] = (uint16_t)v;
    else if (width == 8) ((u64*)out)[i] = (u64)v;
    else { ((u64*)out)[2 * i] = (u64)v; ((u64*)out)[2 * i + 1] = (u64)(v >> 64); }
  }
}

// ------------------------------------------------------------------ launchers
void launch_csv_count_lines(hipStream_t s, const uint8_t* text, i64 n, i64 chunk, int nblocks, uint32_t* counts) {
  hipLaunchKernelGGL(k_csv_count_lines, dim3(nblocks), dim3(FBLOCK), 0, s, text, n, chunk, counts);
}
void launch_csv_line_starts(hipStream_t s, const uint8_t* text, i64 n, i64 chunk, int nblocks, const uint32_t* block_offsets, i64* starts) {
  hipLaunchKernelGGL(k_csv_line_starts, dim3(nblocks), dim3(64), 0, s, text, n, chunk, block_offsets, starts);
}
void launch_csv_parse(hipStream_t s, const uint8_t* text, i64 n_bytes, const i64* starts, i64 row0, i64 n_rows, const CsvSpec& S, const CsvOut& O, uint32_t* flags) {
  if (n_rows <= 0) return;
  i64 need = (n_rows + FBLOCK - 1) / FBLOCK; const i64 cap = (i64)num_cus() * 8;
  hipLaunchKernelGGL(k_csv_parse, dim3((unsigned)(need < cap ? need : cap)), dim3(FBLOCK), 0, s, text, n_bytes, starts, row0, n_rows, S, O, flags);
}
void launch_csv_copy_strings(hipStream_t s, const uint8_t* text, const uint32_t* start, const int32_t* offsets, i64 n, uint8_t* out) {
  if (n <= 0) return;
  i64 need = (n + FBLOCK - 1) / FBLOCK; const i64 cap = (i64)num_cus() * 8;
  hipLaunchKernelGGL(k_csv_copy_strings, dim3((unsigned)(need < cap ? need : cap)), dim3(FBLOCK), 0, s, text, start, offsets, n, out);
}
void launch_pq_decode(hipStream_t s, const uint8_t* file, i64 file_bytes, const PqPage* pages, int n_pages, const PqCol& C, const PqDict* dicts, const uint8_t* dict_values,
                      const int32_t* dict_str_offsets, uint32_t* scratch, i64 scratch_stride, uint32_t* flags) {
  if (n_pages > 0) hipLaunchKernelGGL(k_pq_decode, dim3(n_pages), dim3(64), 0, s, file, file_bytes, pages, n_pages, C, dicts, dict_values, dict_str_offsets, scratch, scratch_stride, flags);
}
void launch_pq_copy_strings(hipStream_t s, const uint8_t* file, const uint8_t* dict_values, const i64* src, const int32_t* offsets, i64 n, uint8_t* out) {
  if (n <= 0) return;
  i64 need = (n + FBLOCK - 1) / FBLOCK; const i64 cap = (i64)num_cus() * 8;
  hipLaunchKernelGGL(k_pq_copy_strings, dim3((unsigned)(need < cap ? need : cap)), dim3(FBLOCK), 0, s, file, dict_values, src, offsets, n, out);
}
void launch_pq_dict_strings(hipStream_t s, const uint8_t* file, i64 src, int32_t bytes, int32_t n, int32_t* offsets, uint8_t* out, uint32_t* flags) {
  hipLaunchKernelGGL(k_pq_dict_strings, dim3(1), dim3(64), 0, s, file, src, bytes, n, offsets, out, flags);
}
void launch_pq_dict_fixed(hipStream_t s, const uint8_t* file, i64 src, int32_t n, int32_t phys, int32_t flba_len, int32_t width, uint8_t* out) {
  if (n > 0) hipLaunchKernelGGL(k_pq_dict_fixed, dim3((n + FBLOCK - 1) / FBLOCK), dim3(FBLOCK), 0, s, file, src, n, phys, flba_len, width, out);
}

}  // namespace gpuq
