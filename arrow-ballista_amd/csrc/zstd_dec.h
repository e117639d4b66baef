// gpuq -- Zstandard frame decoder for Parquet pages (RFC 8878), one wave per page.
//
// The reference's `tpch convert` writes its Parquet files with ZSTD unless told otherwise
// (/root/reference/benchmarks/src/bin/tpch.rs:225-226, 777); the CPU path hands those pages to the `zstd` crate.  This is the
// device side of that: `gpuq_k_zstd_pages` (kernels_zstd_dev.hip) runs `decode_frames` with one 64-lane wave per page.
//
// What runs where inside the wave:
//   * headers, the FSE table descriptions, table builds and the Huffman tree  -- lane 0, tables in LDS (`Shared`, ~11 KB);
//   * Huffman literals                                                         -- the four streams of a block on four lanes;
//   * the sequence bitstream (three interleaved FSE states, read backwards)    -- every lane computes the same values;
//   * literal and match copies                                                 -- all 64 lanes, one byte each per step.
// The chain of sequences is serial per page, like the element chain of the serial Snappy decoder: throughput is the number of
// pages in flight.
//
// The same text compiles as plain C++ (one "lane"): tests/test_cpu_zstd_host.py builds it with g++ and checks it against
// libzstd's output.  That build is a test of the decoder's logic only -- the library never decodes on the host.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
// (Forcing everything inline -- so that no LDS access goes through a generic pointer and a FLAT instruction -- was measured: 256 VGPRs,
// twice the code, 10 % slower.  The compiler's own choice stays.)
#define ZS_FN __device__ static inline
#define ZS_M __device__ inline
#define ZS_LANE ((int)threadIdx.x)
#define ZS_NL 64
#define ZS_SYNC() __syncthreads()
#define ZS_GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads(); } while (0)
// LDS written by some lanes of the wave, read by others: LDS operations of one wave complete in order, the compiler must not move them
#define ZS_LSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define ZS_FN static inline
#define ZS_M inline
#define ZS_LANE 0
#define ZS_NL 1
#define ZS_SYNC() ((void)0)
#define ZS_GSYNC() ((void)0)
#define ZS_LSYNC() ((void)0)
#endif

namespace gpuq {
namespace zs {

constexpr int64_t BLOCK_MAX = 131072;      // Block_Maximum_Size: literals and the output of one block never exceed it
// The device keeps the last RING bytes of the page's output and a window of the block's literals in LDS: a match or a literal run
// fetched from global memory costs a full L2 round trip per sequence (and a match needs the stores before it to have landed).
// So does the sequence bitstream: on gfx9 a load's data is waited for with vmcnt, which also counts the stores in front of it -- a
// bitstream refill from global memory would wait for every output byte stored so far (measured: ~1 us per sequence).
// Each of the four lanes that decode a block's Huffman streams reads its stream through a private HUFW-byte window, too.
constexpr int64_t RING = 32768, LITW = 4096, BITW = 4096, HUFW = 256;
struct Lds { uint8_t* ring; uint8_t* litw; uint64_t* bitw; uint64_t* hufw; };      // ring[RING], litw[LITW], bitw[BITW / 8 + 2], hufw[4][HUFW / 8 + 2]; all null in the host build

struct FseEnt { uint8_t sym, nbits; uint16_t base; };
struct Shared {
  FseEnt ll[512], of[256], ml[512], wt[64];      // literal-length / offset / match-length tables, and the Huffman weights' own
  uint16_t huf[2048];                            // Huffman decode table: symbol | number of bits << 8, indexed by the next huf_bits bits
  int16_t norm[256];
  uint16_t sdesc[256];
  uint8_t weights[256];
  uint32_t rank_count[16], rank_idx[16];
  int32_t ll_al, of_al, ml_al, huf_bits;
  int32_t ll_ok, of_ok, ml_ok, huf_ok;
  int32_t err;
  int64_t used;                                  // bytes a lane-0 parse consumed
};

ZS_FN int highbit(uint32_t v) { return 31 - __builtin_clz(v); }
ZS_FN uint64_t mask64(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }

// eight bytes at p[off ...), zeros beyond len
ZS_FN uint64_t ld64(const uint8_t* p, int64_t off, int64_t len) {
  uint64_t v = 0;
  if (off + 8 <= len) { __builtin_memcpy(&v, p + off, 8); return v; }
  for (int k = 0; k < 8; ++k) if (off + k < len) v |= (uint64_t)p[off + k] << (8 * k);
  return v;
}

// a bitstream written forwards and read backwards: the last byte's highest set bit marks the end
struct Back {
  const uint8_t* p; int64_t len; int64_t pos;      // pos: bits not read yet; negative once more was read than there is
  uint64_t c; int cb;                              // the cb bits right below pos (one load serves several reads)
  ZS_M bool init(const uint8_t* q, int64_t n) {
    p = q; len = n; pos = 0; c = 0; cb = 0; win = nullptr; wb = -BITW - 64;
    if (n < 1 || q[n - 1] == 0) return false;
    pos = (n - 1) * 8 + highbit(q[n - 1]);
    return true;
  }
  uint64_t* win; int64_t wb;                       // an LDS window over p[wb, wb + BITW + 16) (the sequence bitstream; every lane calls alike)
  ZS_M uint64_t load(int64_t q) {                  // 57+ bits from bit q on
    const int64_t b = q >> 3;
    if (!win) return ld64(p, b, len) >> (q & 7);
    if (b < wb || b + 8 > wb + BITW + 8) {
      wb = b + 8 - BITW; if (wb < 0) wb = 0; wb &= ~(int64_t)7;
      ZS_LSYNC();
      for (int64_t j = (int64_t)ZS_LANE * 16; j < BITW + 16; j += ZS_NL * 16) {
        uint64_t lo = 0, hi = 0;
        if (wb + j + 16 <= len) { __builtin_memcpy(&lo, p + wb + j, 8); __builtin_memcpy(&hi, p + wb + j + 8, 8); }
        else { lo = ld64(p, wb + j, len); hi = ld64(p, wb + j + 8, len); }
        win[j >> 3] = lo; win[(j >> 3) + 1] = hi;
      }
      ZS_LSYNC();
    }
    const int64_t r = b - wb; const int sh = (int)((r & 7) * 8 + (q & 7));
    const uint64_t lo = win[r >> 3], hi = win[(r >> 3) + 1];
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
  }
  ZS_M void refill() {
    if (pos >= 57) { c = load(pos - 57) & mask64(57); cb = 57; }
    else if (pos > 0) { c = load(0) & mask64((int)pos); cb = (int)pos; }
    else { c = 0; cb = 0; }
  }
  ZS_M uint64_t read(int n) {      // n <= 32; bits in front of the stream's first byte read as zeros
    if (n == 0) return 0;
    if (cb < n) refill();
    pos -= n;
    const uint64_t m = (1ull << n) - 1;
    if (cb >= n) { cb -= n; return (c >> cb) & m; }
    const uint64_t v = (c << (n - cb)) & m;      // fewer than n bits are left in the whole stream
    cb = 0;
    return v;
  }
};

// ---- FSE -------------------------------------------------------------------------------------------------------------------
ZS_FN bool fse_build(const int16_t* norm, int nsym, int al, FseEnt* T, uint16_t* sdesc) {
  const int size = 1 << al; int high = size;
  for (int s = 0; s < nsym; ++s) if (norm[s] == -1) { T[--high].sym = (uint8_t)s; sdesc[s] = 1; }
  const int step = (size >> 1) + (size >> 3) + 3, msk = size - 1; int pos = 0;
  for (int s = 0; s < nsym; ++s) {
    if (norm[s] <= 0) continue;
    sdesc[s] = (uint16_t)norm[s];
    for (int i = 0; i < norm[s]; ++i) { T[pos].sym = (uint8_t)s; do pos = (pos + step) & msk; while (pos >= high); }
  }
  if (pos != 0) return false;
  for (int i = 0; i < size; ++i) {
    const int s = T[i].sym; const uint32_t d = sdesc[s]++;
    const int nb = al - highbit(d);
    T[i].nbits = (uint8_t)nb; T[i].base = (uint16_t)((d << nb) - (uint32_t)size);
  }
  return true;
}

// an FSE table description at p[0, len): -> bytes consumed, or -1
ZS_FN int64_t fse_read_table(const uint8_t* p, int64_t len, int max_al, int max_sym, FseEnt* T, int32_t* al_out, int16_t* norm, uint16_t* sdesc) {
  int64_t bit = 0; const int64_t nbits_total = len * 8;
  auto rd = [&](int n) -> uint32_t { const uint32_t v = (uint32_t)((ld64(p, bit >> 3, len) >> (bit & 7)) & mask64(n)); bit += n; return v; };
  if (len < 1) return -1;
  const int al = 5 + (int)rd(4);
  if (al > max_al) return -1;
  int remaining = 1 << al, symb = 0;
  while (remaining > 0 && symb < max_sym) {
    const int bits = highbit((uint32_t)remaining + 1) + 1;
    uint32_t val = rd(bits);
    const uint32_t lower = (1u << (bits - 1)) - 1, thr = (1u << bits) - 1 - ((uint32_t)remaining + 1);
    if ((val & lower) < thr) { bit -= 1; val &= lower; }
    else if (val > lower) val -= thr;
    const int proba = (int)val - 1;
    remaining -= proba < 0 ? -proba : proba;
    norm[symb++] = (int16_t)proba;
    if (proba == 0) {
      uint32_t rep = rd(2);
      for (;;) {
        for (uint32_t i = 0; i < rep && symb < max_sym; ++i) norm[symb++] = 0;
        if (rep == 3) rep = rd(2); else break;
        if (bit > nbits_total) return -1;
      }
    }
    if (bit > nbits_total) return -1;
  }
  if (remaining != 0 || bit > nbits_total) return -1;
  if (!fse_build(norm, symb, al, T, sdesc)) return -1;
  *al_out = al;
  return (bit + 7) >> 3;
}

// ---- Huffman ---------------------------------------------------------------------------------------------------------------
// the tree description at p[0, len) -> S.huf / S.huf_bits; returns bytes consumed or -1   (lane 0)
ZS_FN int64_t huf_read_tree(const uint8_t* p, int64_t len, Shared& S) {
  if (len < 1) return -1;
  const int hb = p[0]; int n = 0; int64_t used;
  uint8_t* w = S.weights;
  if (hb >= 128) {
    n = hb - 127; const int nb = (n + 1) / 2;
    if (1 + nb > len) return -1;
    for (int i = 0; i < n; ++i) w[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
    used = 1 + nb;
  } else {
    if (hb == 0 || 1 + hb > len) return -1;
    int32_t al = 0;
    const int64_t t = fse_read_table(p + 1, hb, 6, 256, S.wt, &al, S.norm, S.sdesc);
    if (t < 0 || t >= hb) return -1;
    Back b; if (!b.init(p + 1 + t, hb - t)) return -1;
    uint32_t s1 = (uint32_t)b.read(al), s2 = (uint32_t)b.read(al);
    if (b.pos < 0) return -1;
    for (;;) {
      if (n >= 254) return -1;
      w[n++] = S.wt[s1].sym; s1 = S.wt[s1].base + (uint32_t)b.read(S.wt[s1].nbits);
      if (b.pos < 0) { w[n++] = S.wt[s2].sym; break; }
      if (n >= 254) return -1;
      w[n++] = S.wt[s2].sym; s2 = S.wt[s2].base + (uint32_t)b.read(S.wt[s2].nbits);
      if (b.pos < 0) { w[n++] = S.wt[s1].sym; break; }
    }
    used = 1 + hb;
  }
  uint32_t sum = 0;
  for (int i = 0; i < n; ++i) { if (w[i] > 11) return -1; if (w[i]) sum += 1u << (w[i] - 1); }
  if (sum == 0) return -1;
  const int mb = highbit(sum) + 1;
  if (mb > 11) return -1;
  const uint32_t left = (1u << mb) - sum;
  if (left & (left - 1)) return -1;
  w[n++] = (uint8_t)(highbit(left) + 1);
  for (int i = 0; i < 16; ++i) S.rank_count[i] = 0;
  for (int i = 0; i < n; ++i) if (w[i]) S.rank_count[mb + 1 - w[i]]++;
  S.rank_idx[mb] = 0;
  for (int i = mb; i >= 1; --i) S.rank_idx[i - 1] = S.rank_idx[i] + S.rank_count[i] * (1u << (mb - i));
  if (S.rank_idx[0] != (1u << mb)) return -1;
  for (int i = 0; i < n; ++i) {
    if (!w[i]) continue;
    const int b = mb + 1 - w[i]; const uint32_t code = S.rank_idx[b], cnt = 1u << (mb - b);
    for (uint32_t k = 0; k < cnt; ++k) S.huf[code + k] = (uint16_t)(i | (b << 8));
    S.rank_idx[b] += cnt;
  }
  S.huf_bits = mb;
  return used;
}

// one Huffman stream: n_out symbols, and the stream must be used up exactly.  hw: this lane's own window (HUFW + 16 bytes) or null
ZS_FN bool huf_stream(const uint8_t* p, int64_t len, const uint16_t* tab, int mb, uint8_t* out, int64_t n_out, uint64_t* hw) {
  Back b; if (!b.init(p, len)) return false;
  int64_t pos = b.pos, wb = -HUFW - 64; uint64_t c = 0; int cb = 0;      // c: the cb bits right below pos
  auto load = [&](int64_t q) -> uint64_t {      // 57+ bits from bit q on
    const int64_t at = q >> 3;
    if (!hw) return ld64(p, at, len) >> (q & 7);
    if (at < wb || at + 8 > wb + HUFW + 8) {
      wb = at + 8 - HUFW; if (wb < 0) wb = 0; wb &= ~(int64_t)7;
      constexpr int NW = (int)(HUFW / 8) + 2;
      uint64_t t[NW];
      if (wb + HUFW + 16 <= len) {
#pragma unroll
        for (int k = 0; k < NW; ++k) __builtin_memcpy(&t[k], p + wb + 8 * k, 8);      // (all loads in flight together)
      } else {
#pragma unroll
        for (int k = 0; k < NW; ++k) t[k] = ld64(p, wb + 8 * k, len);
      }
#pragma unroll
      for (int k = 0; k < NW; ++k) hw[k] = t[k];
    }
    const int64_t r = at - wb; const int sh = (int)((r & 7) * 8 + (q & 7));
    const uint64_t lo = hw[r >> 3], hi = hw[(r >> 3) + 1];
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
  };
  const uint32_t m = (1u << mb) - 1;
  uint64_t acc = 0;      // eight symbols leave in one store
  for (int64_t i = 0; i < n_out; ++i) {
    if (cb < mb) {
      if (pos >= 57) { c = load(pos - 57) & mask64(57); cb = 57; }
      else if (pos > 0) { c = load(0) & mask64((int)pos); cb = (int)pos; }
      else { c = 0; cb = 0; }
    }
    const uint32_t v = cb >= mb ? (uint32_t)(c >> (cb - mb)) & m : (uint32_t)(c << (mb - cb)) & m;
    const uint16_t e = tab[v];
    acc |= (uint64_t)(e & 255) << (8 * (int)(i & 7));
    if ((i & 7) == 7) { __builtin_memcpy(out + i - 7, &acc, 8); acc = 0; }
    const int nb = e >> 8;
    pos -= nb; cb -= nb; if (cb < 0) cb = 0;
  }
  for (int64_t i = n_out & ~(int64_t)7; i < n_out; ++i) out[i] = (uint8_t)(acc >> (8 * (int)(i & 7)));
  return pos == 0;
}

// ---- one compressed block --------------------------------------------------------------------------------------------------
ZS_FN bool seq_table(int mode, const uint8_t* sp, int64_t sl, int64_t& k, int which, Shared& S) {      // lane 0
  static constexpr int8_t LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
  static constexpr int8_t OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
  static constexpr int8_t ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
  FseEnt* T = which == 0 ? S.ll : (which == 1 ? S.of : S.ml);
  int32_t* al = which == 0 ? &S.ll_al : (which == 1 ? &S.of_al : &S.ml_al);
  int32_t* ok = which == 0 ? &S.ll_ok : (which == 1 ? &S.of_ok : &S.ml_ok);
  const int max_al = which == 1 ? 8 : 9, max_sym = which == 0 ? 36 : (which == 1 ? 32 : 53);
  if (mode == 0) {
    const int n = which == 0 ? 36 : (which == 1 ? 29 : 53);
    for (int i = 0; i < n; ++i) S.norm[i] = which == 0 ? LL_DEF[i] : (which == 1 ? OF_DEF[i] : ML_DEF[i]);
    *al = which == 1 ? 5 : 6;
    if (!fse_build(S.norm, n, *al, T, S.sdesc)) return false;
  } else if (mode == 1) {
    if (k >= sl) return false;
    const int sym = sp[k++];
    if (sym >= max_sym) return false;
    T[0].sym = (uint8_t)sym; T[0].nbits = 0; T[0].base = 0; *al = 0;
  } else if (mode == 2) {
    const int64_t t = fse_read_table(sp + k, sl - k, max_al, max_sym, T, al, S.norm, S.sdesc);
    if (t < 0) return false;
    k += t;
  } else if (!*ok) return false;
  *ok = 1;
  return true;
}

// block bp[0, bsz) -> out[op ...); rep: the frame's three repeat offsets; base: bytes of this frame in front of op
ZS_FN bool decode_block(const uint8_t* bp, int64_t bsz, uint8_t* out, int64_t& op, int64_t out_len, int64_t frame_start, uint32_t* rep, uint8_t* scratch, Shared& S, const Lds W) {
  static constexpr uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
  static constexpr uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  static constexpr uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
  static constexpr uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  const int lane = ZS_LANE;
  if (bsz < 1) return false;
  // -- literals
  const uint32_t b0 = bp[0]; const int lt = (int)(b0 & 3), sf = (int)((b0 >> 2) & 3);
  int64_t regen, lsec; const uint8_t* lit;
  if (lt < 2) {
    int h;
    if ((sf & 1) == 0) { regen = b0 >> 3; h = 1; }
    else if (sf == 1) { if (bsz < 2) return false; regen = (b0 >> 4) | ((uint32_t)bp[1] << 4); h = 2; }
    else { if (bsz < 3) return false; regen = (b0 >> 4) | ((uint32_t)bp[1] << 4) | ((uint32_t)bp[2] << 12); h = 3; }
    if (regen > BLOCK_MAX) return false;
    if (lt == 0) { if (h + regen > bsz) return false; lit = bp + h; lsec = h + regen; }
    else {
      if (h + 1 > bsz) return false;
      const uint8_t v = bp[h];
      for (int64_t i = lane; i < regen; i += ZS_NL) scratch[i] = v;
      ZS_GSYNC();
      lit = scratch; lsec = h + 1;
    }
  } else {
    const int h = sf < 2 ? 3 : (sf == 2 ? 4 : 5), n = sf < 2 ? 10 : (sf == 2 ? 14 : 18), streams = sf == 0 ? 1 : 4;
    if (bsz < h) return false;
    uint64_t hdr = 0; for (int i = 0; i < h; ++i) hdr |= (uint64_t)bp[i] << (8 * i);
    regen = (int64_t)((hdr >> 4) & mask64(n)); const int64_t comp = (int64_t)((hdr >> (4 + n)) & mask64(n));
    if (regen > BLOCK_MAX || h + comp > bsz) return false;
    const uint8_t* q = bp + h; int64_t qlen = comp;
    ZS_SYNC();      // (every lane is done with the tables of the block before)
    if (lt == 2) {
      if (lane == 0) { S.used = huf_read_tree(q, qlen, S); if (S.used >= 0) S.huf_ok = 1; }
      ZS_SYNC();
      if (S.used < 0) return false;
      q += S.used; qlen -= S.used;
    } else if (!S.huf_ok) return false;
    const int mb = S.huf_bits;
    if (lane == 0) S.err = 0;
    ZS_SYNC();
    if (streams == 1) {
      if (lane == 0 && !huf_stream(q, qlen, S.huf, mb, scratch, regen, W.hufw)) S.err = 1;
    } else {
      if (qlen < 6) return false;
      const int64_t s1 = q[0] | (q[1] << 8), s2 = q[2] | (q[3] << 8), s3 = q[4] | (q[5] << 8), s4 = qlen - 6 - s1 - s2 - s3;
      const int64_t per = (regen + 3) / 4, lastn = regen - 3 * per;
      if (s4 < 1 || lastn < 0) return false;
      for (int st = lane; st < 4; st += ZS_NL) {
        const int64_t so = st == 0 ? 0 : (st == 1 ? s1 : (st == 2 ? s1 + s2 : s1 + s2 + s3)), sn = st == 0 ? s1 : (st == 1 ? s2 : (st == 2 ? s3 : s4));
        if (!huf_stream(q + 6 + so, sn, S.huf, mb, scratch + st * per, st == 3 ? lastn : per, W.hufw ? W.hufw + st * (HUFW / 8 + 2) : nullptr)) S.err = 1;
      }
    }
    ZS_GSYNC();
    if (S.err) return false;
    lit = scratch; lsec = h + comp;
  }
  // -- sequences
  const uint8_t* sp = bp + lsec; const int64_t sl = bsz - lsec;
  if (sl < 1) return false;
  int64_t nseq, k;
  { const uint32_t c = sp[0];
    if (c < 128) { nseq = c; k = 1; }
    else if (c < 255) { if (sl < 2) return false; nseq = ((int64_t)(c - 128) << 8) + sp[1]; k = 2; }
    else { if (sl < 3) return false; nseq = (int64_t)sp[1] + ((int64_t)sp[2] << 8) + 0x7F00; k = 3; } }
  int64_t lp = 0, lw0 = -LITW - 1;      // lw0: the literal position W.litw[0] holds
  if (nseq > 0) {
    if (k >= sl) return false;
    const uint32_t modes = sp[k++];
    if (modes & 3) return false;
    ZS_SYNC();
    if (lane == 0) {
      int64_t kk = k; bool good = true;
      good = good && seq_table((int)(modes >> 6) & 3, sp, sl, kk, 0, S);
      good = good && seq_table((int)(modes >> 4) & 3, sp, sl, kk, 1, S);
      good = good && seq_table((int)(modes >> 2) & 3, sp, sl, kk, 2, S);
      S.used = good ? kk : -1;
    }
    ZS_SYNC();
    if (S.used < 0) return false;
    k = S.used;
    Back b; if (!b.init(sp + k, sl - k)) return false;
    b.win = W.bitw;
    uint32_t lls = (uint32_t)b.read(S.ll_al), ofs = (uint32_t)b.read(S.of_al), mls = (uint32_t)b.read(S.ml_al);
    if (b.pos < 0) return false;
    for (int64_t i = 0; i < nseq; ++i) {
      const FseEnt el = S.ll[lls], eo = S.of[ofs], em = S.ml[mls];
      if (eo.sym > 31 || el.sym > 35 || em.sym > 52) return false;
      const uint64_t ov = (1ull << eo.sym) + b.read(eo.sym);
      const int64_t mlen = (int64_t)ML_BASE[em.sym] + (int64_t)b.read(ML_BITS[em.sym]);
      const int64_t llen = (int64_t)LL_BASE[el.sym] + (int64_t)b.read(LL_BITS[el.sym]);
      uint64_t off;
      if (ov > 3) { off = ov - 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = (uint32_t)off; if (off > 0xFFFFFFFFull) return false; }
      else {
        const int idx = (int)ov + (llen == 0 ? 1 : 0);
        if (idx == 1) off = rep[0];
        else {
          off = idx == 4 ? (uint64_t)rep[0] - 1 : rep[idx - 1];
          if (off == 0 || off > 0xFFFFFFFFull) return false;
          if (idx > 2) rep[2] = rep[1];
          rep[1] = rep[0]; rep[0] = (uint32_t)off;
        }
      }
      if (i + 1 < nseq) {
        lls = el.base + (uint32_t)b.read(el.nbits); mls = em.base + (uint32_t)b.read(em.nbits); ofs = eo.base + (uint32_t)b.read(eo.nbits);
      }
      if (b.pos < 0) return false;
      if (lp + llen > regen || op + llen + mlen > out_len || (int64_t)off > op + llen - frame_start) return false;
      // (a lone wave issues an instruction every few cycles: the copies below index with 32-bit values, and the 64-bit modulo of an
      // overlapping match -- a couple of hundred instructions -- is a 32-bit one, taken only when the match overlaps itself)
      const uint32_t ll32 = (uint32_t)llen, ml32 = (uint32_t)mlen, rop = (uint32_t)op & (uint32_t)(RING - 1);
      uint8_t* const dst = out + op;
      if (W.litw && llen <= LITW) {
        if (llen > 0 && (lp < lw0 || lp + llen > lw0 + LITW)) {
          lw0 = lp;
          for (int64_t j = (int64_t)lane * 16; j < LITW; j += ZS_NL * 16) {
            if (lw0 + j + 16 <= regen) { uint64_t a, b2; __builtin_memcpy(&a, lit + lw0 + j, 8); __builtin_memcpy(&b2, lit + lw0 + j + 8, 8); *(uint64_t*)(W.litw + j) = a; *(uint64_t*)(W.litw + j + 8) = b2; }
            else for (int q = 0; q < 16; ++q) if (lw0 + j + q < regen) W.litw[j + q] = lit[lw0 + j + q];
          }
          ZS_LSYNC();
        }
        const uint8_t* const lw = W.litw + (uint32_t)(lp - lw0);
        for (uint32_t j = (uint32_t)lane; j < ll32; j += ZS_NL) { const uint8_t v = lw[j]; dst[j] = v; if (W.ring) W.ring[(rop + j) & (uint32_t)(RING - 1)] = v; }
      } else {
        const uint8_t* const ls = lit + lp;
        for (uint32_t j = (uint32_t)lane; j < ll32; j += ZS_NL) { const uint8_t v = ls[j]; dst[j] = v; if (W.ring) W.ring[(rop + j) & (uint32_t)(RING - 1)] = v; }
      }
      lp += llen; op += llen;
      const int64_t o = (int64_t)off;
      uint8_t* const md = out + op;
      const uint32_t mrop = (uint32_t)op & (uint32_t)(RING - 1);
      if (W.ring && o + mlen <= RING) {      // every source byte is still in the ring, and this copy's own bytes land elsewhere in it
        ZS_LSYNC();
        const uint32_t o32 = (uint32_t)o, sbase = mrop + (uint32_t)RING - o32;
        if (o32 >= ml32) for (uint32_t j = (uint32_t)lane; j < ml32; j += ZS_NL) { const uint8_t v = W.ring[(sbase + j) & (uint32_t)(RING - 1)]; md[j] = v; W.ring[(mrop + j) & (uint32_t)(RING - 1)] = v; }
        else for (uint32_t j = (uint32_t)lane; j < ml32; j += ZS_NL) { const uint8_t v = W.ring[(sbase + j % o32) & (uint32_t)(RING - 1)]; md[j] = v; W.ring[(mrop + j) & (uint32_t)(RING - 1)] = v; }
      } else {
        ZS_GSYNC();      // bytes stored by other lanes are this copy's source
        const uint8_t* const ms = md - o;
        if (o >= mlen) for (uint32_t j = (uint32_t)lane; j < ml32; j += ZS_NL) { const uint8_t v = ms[j]; md[j] = v; if (W.ring) W.ring[(mrop + j) & (uint32_t)(RING - 1)] = v; }
        else { const uint32_t o32 = (uint32_t)o; for (uint32_t j = (uint32_t)lane; j < ml32; j += ZS_NL) { const uint8_t v = ms[j % o32]; md[j] = v; if (W.ring) W.ring[(mrop + j) & (uint32_t)(RING - 1)] = v; } }
      }
      op += mlen;
    }
    if (b.pos != 0) return false;
  }
  const int64_t rest = regen - lp;
  if (op + rest > out_len) return false;
  for (int64_t j = lane; j < rest; j += ZS_NL) { const uint8_t v = lit[lp + j]; out[op + j] = v; if (W.ring) W.ring[(op + j) & (RING - 1)] = v; }
  op += rest;
  return true;
}

// every frame of in[0, in_len) -> out[0, out_len), exactly; scratch: BLOCK_MAX bytes of this wave's own
ZS_FN bool decode_frames(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_len, uint8_t* scratch, Shared& S, const Lds W = Lds{nullptr, nullptr, nullptr, nullptr}) {
  const int lane = ZS_LANE;
  int64_t ip = 0, op = 0;
  auto le = [&](int64_t at, int n) -> uint64_t { uint64_t v = 0; for (int i = 0; i < n; ++i) v |= (uint64_t)in[at + i] << (8 * i); return v; };
  while (ip < in_len) {
    if (in_len - ip < 4) return false;
    const uint32_t magic = (uint32_t)le(ip, 4);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { if (in_len - ip < 8) return false; const int64_t sz = (int64_t)le(ip + 4, 4); if (sz > in_len - ip - 8) return false; ip += 8 + sz; continue; }
    if (magic != 0xFD2FB528u || in_len - ip < 5) return false;
    ip += 4;
    const uint32_t fhd = in[ip++];
    const int fcs_flag = (int)(fhd >> 6), single = (int)((fhd >> 5) & 1), checksum = (int)((fhd >> 2) & 1), did = (int)(fhd & 3);
    if (fhd & 0x08) return false;
    const int did_bytes = did == 3 ? 4 : did, fcs_bytes = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    if (in_len - ip < (single ? 0 : 1) + did_bytes + fcs_bytes) return false;
    if (!single) ip += 1;
    if (did_bytes && le(ip, did_bytes) != 0) return false;      // a dictionary: not something a Parquet page carries
    ip += did_bytes;
    int64_t fcs = -1;
    if (fcs_bytes) { fcs = (int64_t)le(ip, fcs_bytes); if (fcs_bytes == 2) fcs += 256; ip += fcs_bytes; }
    const int64_t frame_start = op;
    uint32_t rep[3] = {1, 4, 8};
    ZS_SYNC();
    if (lane == 0) { S.ll_ok = S.of_ok = S.ml_ok = S.huf_ok = 0; }
    ZS_SYNC();
    for (;;) {
      if (in_len - ip < 3) return false;
      const uint32_t bh = (uint32_t)le(ip, 3); ip += 3;
      const int last = (int)(bh & 1), type = (int)((bh >> 1) & 3); const int64_t bsz = bh >> 3;
      if (type == 0) {
        if (bsz > in_len - ip || bsz > out_len - op) return false;
        for (int64_t j = lane; j < bsz; j += ZS_NL) { const uint8_t v = in[ip + j]; out[op + j] = v; if (W.ring) W.ring[(op + j) & (RING - 1)] = v; }
        ip += bsz; op += bsz;
      } else if (type == 1) {
        if (in_len - ip < 1 || bsz > out_len - op) return false;
        const uint8_t v = in[ip++];
        for (int64_t j = lane; j < bsz; j += ZS_NL) { out[op + j] = v; if (W.ring) W.ring[(op + j) & (RING - 1)] = v; }
        op += bsz;
      } else if (type == 2) {
        if (bsz > in_len - ip || bsz > BLOCK_MAX) return false;
        if (!decode_block(in + ip, bsz, out, op, out_len, frame_start, rep, scratch, S, W)) return false;
        ip += bsz;
      } else return false;
      if (last) break;
    }
    if (checksum) { if (in_len - ip < 4) return false; ip += 4; }      // (XXH64 of the content: not verified)
    if (fcs >= 0 && op - frame_start != fcs) return false;
  }
  return op == out_len;
}

}  // namespace zs
}  // namespace gpuq
