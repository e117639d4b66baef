// LZ4 block codec for the shuffle sink / source: the reference writes its shuffle partitions as Arrow
// IPC streams whose buffers are LZ4 frames (ballista/core/src/execution_plans/shuffle_writer.rs:365-378,
// ballista/core/src/utils.rs:179-219; the codec itself is the `lz4_flex` crate behind arrow-ipc 49 -- not in the tree, the
// format is the public LZ4 block / frame specification).  One WAVE per unit of independent work:
//   * compress: one 64 KiB block (frames are written with independent blocks, so every block is a unit);
//   * decompress: one independent block, or one whole frame with linked blocks (what Arrow C++ writes).
// Everything is byte / integer work bound by dependent-latency chains, not by MFMA or even HBM; the parallelism is across
// blocks (a 1 GB shuffle partition has 16 Ki of them).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "gpuq_kernels.h"

namespace gpuq {

typedef unsigned long long u64;

constexpr int LZ4_HASH_BITS = 12;
constexpr int LZ4_MFLIMIT = 12;        // a match must start at least 12 bytes before the end of the block
constexpr int LZ4_LASTLITERALS = 5;    // the last 5 bytes of a block are literals

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// n bytes, all lanes of the wave; src and dst do not overlap
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, int n) {
  const int lane = lane_id();
  if (n >= 512) {           // long runs: 4 bytes per lane through unaligned dword loads, stores aligned after a byte head
    const int head = (int)((4 - ((uintptr_t)dst & 3)) & 3);
    if (lane < head) dst[lane] = src[lane];
    const int words = (n - head) >> 2;
    for (int w = lane; w < words; w += 64) *(uint32_t*)(dst + head + 4 * w) = ld32u(src + head + 4 * w);
    const int done = head + 4 * words;
    if (lane < n - done) dst[done + lane] = src[done + lane];
    return;
  }
  for (int i = lane; i < n; i += 64) dst[i] = src[i];
}

// ---------------------------------------------------------------- compress
// length field continuation: v >= 15 -> bytes of 255 then the remainder
__device__ __forceinline__ int put_len_ext(uint8_t* out, int o, int v /* value - 15, >= 0 */) {
  const int nb = v / 255 + 1, lane = lane_id();
  for (int i = lane; i < nb; i += 64) out[o + i] = (uint8_t)((i < nb - 1) ? 255 : (v % 255));
  return o + nb;
}

__global__ __launch_bounds__(64) void k_lz4_compress(const uint8_t* __restrict__ src_base, uint8_t* __restrict__ dst_base, const Lz4Block* __restrict__ blocks,
                                                     int n_blocks, int32_t* __restrict__ csize_out) {
  __shared__ uint16_t ht[1 << LZ4_HASH_BITS];
  const int b = (int)blockIdx.x;
  if (b >= n_blocks) return;
  const int lane = lane_id();
  const uint8_t* src = src_base + blocks[b].src;
  uint8_t* out = dst_base + blocks[b].slot;
  const int len = blocks[b].len;
  for (int i = lane; i < (1 << LZ4_HASH_BITS); i += 64) ht[i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  int o = 0, anchor = 0;
  const int mlast = len - LZ4_MFLIMIT;            // last position a match may start at
  const int mend = len - LZ4_LASTLITERALS;        // matches end at or before this
  int base = 1;                                    // position 0 can never be a match start's candidate target < p anyway
  int cur = 0;                                     // first position not yet covered by an emitted sequence
  while (base <= mlast) {
    const int p = base + lane;
    const bool valid = p <= mlast;
    const uint32_t v = valid ? ld32u(src + p) : 0u;
    const uint32_t h = (v * 2654435761u) >> (32 - LZ4_HASH_BITS);
    int cand = valid ? (int)ht[h] : 0;
    bool ok = valid && cand < p && ld32u(src + cand) == v;
    // columnar data repeats with the period of its element width: look there before giving up (the hash table only
    // knows positions of EARLIER windows)
    if (__ballot(valid && !ok)) {
#pragma unroll
      for (int k = 16; k >= 1; k >>= 1) {
        if (k == 2) continue;
        if (valid && !ok && p >= k && ld32u(src + p - k) == v) { ok = true; cand = p - k; }
      }
    }
    if (valid) ht[h] = (uint16_t)p;
    u64 m = __ballot(ok);
    while (m) {
      const int L = __builtin_ctzll(m);
      const int pL = base + L;
      if (pL < cur) { m &= m - 1; continue; }
      const int cL = __builtin_amdgcn_readlane(cand, L);
      // extend: 4 bytes per lane, 256 per round
      int mlen = 4;
      {
        const int limit = mend - (pL + 4);          // bytes that may still be added
        int done = 0;
        while (done < limit) {
          const int off = done + 4 * lane;
          const int rem = limit - off;               // bytes this lane may compare (<=0: none)
          uint32_t x = 0; bool stop = rem < 4;
          if (rem > 0) {
            x = ld32u(src + pL + 4 + off) ^ ld32u(src + cL + 4 + off);
            if (rem < 4) x |= 0xFFFFFFFFu << (8 * rem);   // bytes past the limit count as mismatches
          } else x = 0xFFFFFFFFu;
          stop = x != 0;
          const u64 sm = __ballot(stop);
          if (sm) {
            const int S = __builtin_ctzll(sm);
            const uint32_t xs = (uint32_t)__builtin_amdgcn_readlane((int)x, S);
            done += 4 * S + (__builtin_ctz(xs) >> 3);
            break;
          }
          done += 256;
        }
        if (done > limit) done = limit;
        if (done < 0) done = 0;
        mlen += done;
      }
      // emit: token, literal length, literals, offset, match length
      const int lit = pL - anchor, ml = mlen - 4;
      if (lane == 0) out[o] = (uint8_t)(((lit < 15 ? lit : 15) << 4) | (ml < 15 ? ml : 15));
      o += 1;
      if (lit >= 15) o = put_len_ext(out, o, lit - 15);
      wave_copy(out + o, src + anchor, lit);
      o += lit;
      if (lane == 0) { const int d = pL - cL; out[o] = (uint8_t)d; out[o + 1] = (uint8_t)(d >> 8); }
      o += 2;
      if (ml >= 15) o = put_len_ext(out, o, ml - 15);
      cur = pL + mlen; anchor = cur;
      const int rel = cur - base;
      m = rel >= 64 ? 0ull : (m & ~((1ull << rel) - 1ull));
    }
    base = (base + 64 > cur) ? base + 64 : cur;
  }
  // last sequence: literals only
  {
    const int lit = len - anchor;
    if (lane == 0) out[o] = (uint8_t)((lit < 15 ? lit : 15) << 4);
    o += 1;
    if (lit >= 15) o = put_len_ext(out, o, lit - 15);
    wave_copy(out + o, src + anchor, lit);
    o += lit;
  }
  if (lane == 0) csize_out[b] = o;
}

// ---------------------------------------------------------------- frame layout + pack
// frame = [u64 uncompressed length][magic FLG BD HC][u32 block size | data]...[u32 0]; a block that did not shrink is stored raw
// with bit 31 of its size set.  A buffer whose frame would not be smaller than the data is written as [-1][raw bytes]
// (arrow-ipc's LENGTH_NO_COMPRESSED_DATA); an empty buffer has no bytes at all.
// Layout in three small launches (a single thread per buffer walking thousands of block sizes took 0.9 ms for 11 Ki blocks):
// A: one workgroup per buffer -- exclusive scan of its blocks' stored sizes (4-byte header + min(csize, len)) into blk_dst
//    (relative to the first block header) and the frame length; B: one thread -- buffer offsets (8-byte aligned);
// C: one workgroup per buffer -- blk_dst made absolute (negative = the buffer goes raw: -(destination of the bytes + 1)).
__global__ void __launch_bounds__(256) k_lz4_layout_sizes(const Lz4Block* __restrict__ blocks, const int32_t* __restrict__ csize, const int32_t* __restrict__ buf_first_block,
                                                          int64_t* __restrict__ buf_len, int64_t* __restrict__ blk_dst, int64_t* __restrict__ raw_len) {
  __shared__ int64_t wsum[4]; __shared__ int64_t carry, rawc;
  const int b = (int)blockIdx.x, f = buf_first_block[b], l = buf_first_block[b + 1], t = (int)threadIdx.x;
  if (t == 0) { carry = 0; rawc = 0; }
  __syncthreads();
  for (int k0 = f; k0 < l; k0 += 256) {
    const int k = k0 + t;
    int64_t v = 0, r = 0;
    if (k < l) { r = blocks[k].len; v = 4 + (csize[k] < blocks[k].len ? csize[k] : blocks[k].len); }
    int64_t x = v, rr = r;                                      // inclusive scan inside the wave, raw sum alongside
    for (int o = 1; o < 64; o <<= 1) { const int64_t y = __shfl_up(x, o); if ((t & 63) >= o) x += y; }
    for (int o = 32; o > 0; o >>= 1) rr += __shfl_xor(rr, o);
    if ((t & 63) == 63) wsum[t >> 6] = x;
    __syncthreads();
    int64_t before = carry;
    for (int w = 0; w < (t >> 6); ++w) before += wsum[w];
    if (k < l) blk_dst[k] = before + x - v;                     // exclusive
    __syncthreads();
    if (t == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if ((t & 63) == 0) atomicAdd((unsigned long long*)&rawc, (unsigned long long)rr);
    __syncthreads();
  }
  if (t == 0) {
    const int64_t raw = rawc, fr = 8 + 7 + carry + 4;
    raw_len[b] = raw;
    buf_len[b] = (raw == 0) ? 0 : (fr < raw + 8 ? fr : -(raw + 8));
  }
}
__global__ void k_lz4_layout_offsets(int n_buffers, const int64_t* __restrict__ buf_len, int64_t* __restrict__ buf_off) {
  if (threadIdx.x || blockIdx.x) return;
  int64_t off = 0;
  for (int b = 0; b < n_buffers; ++b) { buf_off[b] = off; const int64_t l = buf_len[b] < 0 ? -buf_len[b] : buf_len[b]; off += (l + 7) & ~(int64_t)7; }
  buf_off[n_buffers] = off;
}
__global__ void __launch_bounds__(256) k_lz4_layout_place(const Lz4Block* __restrict__ blocks, const int32_t* __restrict__ buf_first_block, const int64_t* __restrict__ buf_off,
                                                          const int64_t* __restrict__ buf_len, int64_t* __restrict__ blk_dst) {
  const int b = (int)blockIdx.x, f = buf_first_block[b], l = buf_first_block[b + 1], t = (int)threadIdx.x;
  if (buf_len[b] >= 0) {
    const int64_t base = buf_off[b] + 8 + 7;
    for (int k = f + t; k < l; k += 256) blk_dst[k] += base;
    return;
  }
  // raw buffer: blocks are full 64 KiB except the last, so the byte position of block k is (k - f) * 64 KiB
  const int64_t base = buf_off[b] + 8;
  for (int k = f + t; k < l; k += 256) blk_dst[k] = -(base + (int64_t)(k - f) * LZ4_BLOCK_BYTES + 1);
}

__device__ __forceinline__ void block_copy(uint8_t* dst, const uint8_t* src, int n) {
  const int t = (int)threadIdx.x, T = (int)blockDim.x;
  const int head = (int)((4 - ((uintptr_t)dst & 3)) & 3) < n ? (int)((4 - ((uintptr_t)dst & 3)) & 3) : n;
  if (t < head) dst[t] = src[t];
  const int words = (n - head) >> 2;
  for (int w = t; w < words; w += T) *(uint32_t*)(dst + head + 4 * w) = ld32u(src + head + 4 * w);
  const int done = head + 4 * words;
  if (t < n - done) dst[done + t] = src[done + t];
}

__global__ __launch_bounds__(256) void k_lz4_pack(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ slots, const Lz4Block* __restrict__ blocks,
                                                  const int32_t* __restrict__ csize, const int32_t* __restrict__ blk_buffer, const int32_t* __restrict__ buf_first_block,
                                                  const int64_t* __restrict__ buf_off, const int64_t* __restrict__ buf_len, const int64_t* __restrict__ blk_dst,
                                                  uint8_t* __restrict__ body) {
  const int k = (int)blockIdx.x, t = (int)threadIdx.x;
  const int b = blk_buffer[k];
  const bool first = buf_first_block[b] == k, last = buf_first_block[b + 1] == k + 1;
  const int len = blocks[k].len;
  const int64_t bl = buf_len[b];
  const int64_t total = bl < 0 ? -bl : bl;
  uint8_t* const bufp = body + buf_off[b];
  if (first && t < 8) {
    int64_t raw = 0;
    if (bl < 0) raw = -1; else for (int j = buf_first_block[b]; j < buf_first_block[b + 1]; ++j) raw += blocks[j].len;
    bufp[t] = (uint8_t)((u64)raw >> (8 * t));
  }
  if (last) { const int64_t padded = (total + 7) & ~(int64_t)7; if (t < (int)(padded - total)) bufp[total + t] = 0; }
  if (bl < 0) { block_copy(body + (-blk_dst[k] - 1), src_base + blocks[k].src, len); return; }
  if (first && t == 0) { uint8_t* h = bufp + 8; h[0] = 0x04; h[1] = 0x22; h[2] = 0x4D; h[3] = 0x18; h[4] = 0x60; h[5] = 0x40; h[6] = 0x82; }
  uint8_t* d = body + blk_dst[k];
  const int cs = csize[k];
  const bool stored = cs >= len;
  const uint32_t hdr = stored ? ((uint32_t)len | 0x80000000u) : (uint32_t)cs;
  if (t < 4) d[t] = (uint8_t)(hdr >> (8 * t));
  block_copy(d + 4, stored ? src_base + blocks[k].src : slots + blocks[k].slot, stored ? len : cs);
  if (last && t < 4) d[4 + (stored ? len : cs) + t] = 0;      // EndMark
}

// ---------------------------------------------------------------- decompress
struct InWin {      // 256-byte window of the compressed stream held in registers (one dword per lane)
  const uint32_t* base; int64_t w0; int64_t wend; uint32_t w;
};
__device__ __forceinline__ void win_load(InWin& W, int64_t dw) {
  W.w0 = dw;
  const int64_t i = dw + lane_id();
  W.w = (i < W.wend) ? W.base[i] : 0u;
}
__device__ __forceinline__ int win_byte(InWin& W, int64_t a /* byte address relative to W.base */) {
  const int64_t dw = a >> 2;
  if (dw < W.w0 || dw >= W.w0 + 64) win_load(W, dw);
  const int l = rfl((int)(dw - W.w0));
  return (int)((((uint32_t)__builtin_amdgcn_readlane((int)W.w, l)) >> (8 * (int)(a & 3))) & 0xFF);
}

// one LZ4 block: in [ip, iend) -> out at op; returns the new op or -1 on a malformed stream
__device__ __forceinline__ int64_t lz4_decode_block(InWin& W, const uint8_t* in8, int64_t ip, int64_t iend, uint8_t* out, int64_t op, int64_t out_lo, int64_t out_hi) {
  const int lane = lane_id();
  while (ip < iend) {
    const int token = win_byte(W, ip); ip += 1;
    int64_t lit = token >> 4;      // 64-bit, and bounded while it grows: a long run of 0xFF length bytes must not wrap past the checks below
    if (lit == 15) { int x; do { if (ip >= iend || lit > out_hi - op) return -1; x = win_byte(W, ip); ip += 1; lit += x; } while (x == 255); }
    if (lit > iend - ip || lit > out_hi - op) return -1;
    if (lit) { wave_copy(out + op, in8 + ip, (int)lit); ip += lit; op += lit; }
    if (ip >= iend) break;                            // the last sequence has no match
    if (ip + 2 > iend) return -1;
    const int offset = win_byte(W, ip) | (win_byte(W, ip + 1) << 8); ip += 2;
    int64_t ml = token & 15;
    if (ml == 15) { int x; do { if (ip >= iend || ml > out_hi - op) return -1; x = win_byte(W, ip); ip += 1; ml += x; } while (x == 255); }
    ml += 4;
    if (offset == 0 || offset > op - out_lo || ml > out_hi - op) return -1;
    // the source lies entirely before op (bytes of a period shorter than the match repeat): every byte is independent
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const uint8_t* s = out + op - offset;
    if (offset >= ml) { for (int64_t i = lane; i < ml; i += 64) out[op + i] = s[i]; }
    else { for (int64_t i = lane; i < ml; i += 64) out[op + i] = s[i % offset]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    op += ml;
  }
  return op;
}

__global__ __launch_bounds__(64) void k_lz4_decode(const uint8_t* __restrict__ src_base, uint8_t* dst_base, const Lz4Unit* __restrict__ units, int n_units,
                                                   int64_t src_bytes, uint32_t* __restrict__ status) {
  const int u = (int)blockIdx.x;
  if (u >= n_units) return;
  const Lz4Unit U = units[u];
  InWin W; W.base = (const uint32_t*)src_base; W.wend = (src_bytes + 3) >> 2; W.w0 = -1000; W.w = 0;
  uint8_t* out = dst_base + U.dst;
  bool bad = false;
  if (U.mode == LZ4_UNIT_BLOCK) {
    const int64_t end = lz4_decode_block(W, src_base, U.src, U.src + U.src_len, out, 0, 0, U.dst_len);
    bad = end != U.dst_len;
  } else if (U.mode == LZ4_UNIT_STORED) {
    if (U.src_len != U.dst_len) bad = true; else wave_copy(out, src_base + U.src, (int)U.dst_len);
  } else {      // LZ4_UNIT_FRAME_BLOCKS: walk the block headers of one frame (linked blocks: matches reach into earlier blocks)
    int64_t ip = U.src, op = 0;
    const int64_t iend = U.src + U.src_len;
    while (true) {
      if (ip + 4 > iend) { bad = true; break; }
      const uint32_t hdr = (uint32_t)win_byte(W, ip) | ((uint32_t)win_byte(W, ip + 1) << 8) | ((uint32_t)win_byte(W, ip + 2) << 16) | ((uint32_t)win_byte(W, ip + 3) << 24);
      ip += 4;
      if (hdr == 0) break;                                         // EndMark
      const int64_t bs = (int64_t)(hdr & 0x7FFFFFFFu);
      if (bs > iend - ip) { bad = true; break; }
      if (hdr & 0x80000000u) {
        if (bs > U.dst_len - op) { bad = true; break; }
        wave_copy(out + op, src_base + ip, (int)bs); op += bs;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      } else {
        // the window a match may reach back into is the 64 KiB before op, across block boundaries
        op = lz4_decode_block(W, src_base, ip, ip + bs, out, op, 0, U.dst_len);
        if (op < 0) { bad = true; break; }
      }
      ip += bs + ((U.flags & LZ4_UNIT_BLOCK_CHECKSUM) ? 4 : 0);
    }
    if (!bad && op != U.dst_len) bad = true;
  }
  if (bad && lane_id() == 0) atomicOr(status, 1u);
}

__global__ void k_popcount_bits(const u64* __restrict__ bits, int64_t n_bits, unsigned long long* __restrict__ out) {
  const int64_t words = (n_bits + 63) >> 6;
  unsigned long long c = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) {
    u64 x = bits[w];
    if (w == words - 1 && (n_bits & 63)) x &= (1ull << (n_bits & 63)) - 1ull;
    c += (unsigned long long)__popcll(x);
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ---------------------------------------------------------------- Utf8 columns of a multi-batch stream
// Every batch brings its own offsets (decoded into a temp) and its own data buffer (decoded at a host-known position `at`,
// the sum of the buffers before it).  Producers may pad a data buffer beyond the bytes its offsets use (Arrow C++ rounds a
// sliced buffer up to 8 bytes) or leave the offsets un-rebased: the merged column needs `start` = the bytes really used before
// the batch.  k_utf8_piece_starts computes that and raises `gap` when the data is not already contiguous (then
// k_utf8_piece_compact moves it); k_utf8_piece_offsets writes the merged offsets.
__global__ void k_utf8_piece_starts(Utf8Piece* __restrict__ pieces, int n_pieces, uint32_t* __restrict__ gap) {
  for (int k = (int)threadIdx.x; k < n_pieces; k += (int)blockDim.x) {
    const int32_t a = pieces[k].tmp[0], b = pieces[k].tmp[pieces[k].n];
    pieces[k].first = a; pieces[k].used = (int64_t)b - a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t start = 0; bool g = false;
    for (int k = 0; k < n_pieces; ++k) {
      pieces[k].start = start;
      g |= (start != pieces[k].at + pieces[k].first) || pieces[k].used < 0 || pieces[k].used + pieces[k].first > pieces[k].dl;
      start += pieces[k].used;
    }
    if (g) atomicOr(gap, 2u);
  }
}
// A peer's shuffle file is untrusted input: offsets that run backwards, start below zero or end beyond the bytes their data
// buffer holds would make every later kernel read out of bounds.  Raises bit 2 of the decode status ("malformed").
__global__ __launch_bounds__(256) void k_utf8_piece_validate(const Utf8Piece* __restrict__ pieces, uint32_t* __restrict__ status) {
  const Utf8Piece P = pieces[blockIdx.y];
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * 256) bad |= P.tmp[i] > P.tmp[i + 1];
  if (blockIdx.x == 0 && threadIdx.x == 0) bad |= P.tmp[0] < 0 || (int64_t)P.tmp[P.n] > P.dl;
  if (bad) atomicOr(status, 4u);
}
__global__ __launch_bounds__(256) void k_utf8_piece_offsets(const Utf8Piece* __restrict__ pieces) {
  const Utf8Piece P = pieces[blockIdx.y];
  const int32_t delta = (int32_t)(P.start - P.first);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= P.n; i += (int64_t)gridDim.x * 256) P.dst[i] = P.tmp[i] + delta;
}
__global__ __launch_bounds__(256) void k_utf8_piece_compact(const Utf8Piece* __restrict__ pieces, const uint8_t* __restrict__ from, uint8_t* __restrict__ to) {
  const Utf8Piece P = pieces[blockIdx.y];
  if (P.used <= 0 || P.used + P.first > P.dl) return;
  const uint8_t* s = from + P.at + P.first; uint8_t* d = to + P.start;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.used; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}

// ---------------------------------------------------------------- Snappy (Parquet pages; kernels_scanfmt.hip decodes what this unpacks)
// Raw Snappy block format [UPSTREAM-KNOWLEDGE: google/snappy format_description.txt]: varint uncompressed length, then elements --
// tag & 3 == 0 literal (length (tag >> 2) + 1, or 1-4 length bytes when that is 61-64), 1 copy with 11-bit offset and length 4-11,
// 2 copy with 16-bit offset, 3 copy with 32-bit offset (length (tag >> 2) + 1).  One wave per page: every lane reads the same tag
// bytes (uniform loads), literals and copies are moved by all lanes; a copy whose offset is shorter than its length repeats the
// period (source byte i % offset lies before the copy).  mode 0 = stored (plain copy).  `raw_prefix` bytes in front of the
// compressed stream are copied verbatim (a v2 data page's levels).
// The element stream is read through a 4 KiB window in LDS (refilled with coalesced 16-byte loads) and the last 32 KiB of output
// are mirrored in an LDS ring: a tag byte or a short back-reference fetched from global memory costs a full L2 round trip
// (~0.7 us, and a copy's source was written only moments ago).  Copies that reach further back than the ring read global memory.
// What remains is the serial element chain itself: sorted / low-entropy numeric columns compress into one or two elements per VALUE
// (a 1 MB page of ascending int64 keys is ~260 K elements), a few LDS round trips each: ~10 MB/s per wave, i.e. the decoder's
// throughput is the number of pages in flight (SF1 lineitem, 110 pages of 1 MB: 117 ms -- the host's 93 ms on 256 threads;
// a wide copy for literals made it slower, the literals are short).
constexpr int SNAPPY_INWIN = 4096, SNAPPY_RING = 32768;
__global__ __launch_bounds__(64) void k_unpack_pages(const uint8_t* __restrict__ src_base, uint8_t* __restrict__ dst_base, const UnpackJob* __restrict__ jobs, int n_jobs,
                                                     uint32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t inw[SNAPPY_INWIN + 16];
  __shared__ uint8_t ring[SNAPPY_RING];
  const int u = (int)blockIdx.x;
  if (u >= n_jobs) return;
  const UnpackJob J = jobs[u];
  const int lane = lane_id();
  const uint8_t* in = src_base + J.src; uint8_t* out = dst_base + J.dst;
  if (J.raw_prefix > J.src_len || J.raw_prefix > J.dst_len) { if (lane == 0) atomicOr(status, 1u); return; }
  for (int64_t i = lane; i < J.raw_prefix; i += 64) out[i] = in[i];
  in += J.raw_prefix; out += J.raw_prefix;
  if (J.mode == 4) return;      // ZSTD frames: k_zstd_pages
  const int64_t in_len = J.src_len - J.raw_prefix, out_len = J.dst_len - J.raw_prefix;
  if (J.mode == 0) {
    if (in_len != out_len) { if (lane == 0) atomicOr(status, 1u); return; }
    for (int64_t i = lane; i < in_len; i += 64) out[i] = in[i];
    return;
  }
  // input window: bytes [wbase, wbase + SNAPPY_INWIN) of `in`
  int64_t wbase = -(int64_t)SNAPPY_INWIN - 16;
  auto want = [&](int64_t pos, int nbytes) {      // wave-uniform
    if (pos >= wbase && pos + nbytes <= wbase + SNAPPY_INWIN) return;
    wbase = pos;
    for (int j = lane * 16; j < SNAPPY_INWIN; j += 64 * 16) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (wbase + j + 16 <= in_len) __builtin_memcpy(&v, in + wbase + j, 16);
      else for (int q = 0; q < 16; ++q) if (wbase + j + q < in_len) ((uint8_t*)&v)[q] = in[wbase + j + q];
      *(uint4*)(inw + j) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  };
  auto ib = [&](int64_t pos) -> uint32_t { return inw[pos - wbase]; };
  int64_t ip = 0, op = 0;
  bool bad = false;
  // preamble: uncompressed length
  want(0, 8);
  uint64_t ulen = 0; int sh = 0;
  for (;;) { if (ip >= in_len || sh > 35) { bad = true; break; } const uint32_t c = ib(ip++); ulen |= (uint64_t)(c & 0x7F) << sh; if (!(c & 0x80)) break; sh += 7; }
  if (bad || (int64_t)ulen != out_len) { if (lane == 0) atomicOr(status, 1u); return; }
  while (ip < in_len) {
    want(ip, 5);                                       // a tag and up to four length / offset bytes
    const uint32_t tag = ib(ip++);
    int64_t len; int64_t off = 0;
    if ((tag & 3) == 0) {
      len = (int64_t)(tag >> 2) + 1;
      if (len > 60) {
        const int nb = (int)len - 60;
        if (ip + nb > in_len) { bad = true; break; }
        uint32_t v = 0; for (int k = 0; k < nb; ++k) v |= ib(ip + k) << (8 * k);
        ip += nb; len = (int64_t)v + 1;
      }
      if (len > in_len - ip || len > out_len - op) { bad = true; break; }
      for (int64_t i = lane; i < len; i += 64) { const uint8_t b = in[ip + i]; out[op + i] = b; ring[(op + i) & (SNAPPY_RING - 1)] = b; }
      ip += len; op += len;
      continue;
    }
    if ((tag & 3) == 1) { if (ip + 1 > in_len) { bad = true; break; } len = 4 + ((tag >> 2) & 7); off = (int64_t)((tag >> 5) << 8) | ib(ip); ip += 1; }
    else if ((tag & 3) == 2) { if (ip + 2 > in_len) { bad = true; break; } len = (int64_t)(tag >> 2) + 1; off = (int64_t)ib(ip) | ((int64_t)ib(ip + 1) << 8); ip += 2; }
    else { if (ip + 4 > in_len) { bad = true; break; } len = (int64_t)(tag >> 2) + 1; off = (int64_t)ib(ip) | ((int64_t)ib(ip + 1) << 8) | ((int64_t)ib(ip + 2) << 16) | ((int64_t)ib(ip + 3) << 24); ip += 4; }
    if (off == 0 || off > op || len > out_len - op) { bad = true; break; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();      // earlier stores of this wave are visible to its loads
    const bool near = off + 64 <= SNAPPY_RING;         // every source byte is still in the ring (a copy is at most 64 bytes long)
    for (int64_t i = lane; i < len; i += 64) {
      const int64_t sp = op - off + (off >= len ? i : i % off);
      const uint8_t b = near ? ring[sp & (SNAPPY_RING - 1)] : out[sp];
      out[op + i] = b; ring[(op + i) & (SNAPPY_RING - 1)] = b;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    op += len;
  }
  if (bad || op != out_len) { if (lane == 0) atomicOr(status, 1u); }
}

// ---------------------------------------------------------------- launchers
void launch_utf8_piece_starts(hipStream_t s, Utf8Piece* pieces, int n_pieces, uint32_t* gap) {
  if (n_pieces > 0) hipLaunchKernelGGL(k_utf8_piece_starts, dim3(1), dim3(256), 0, s, pieces, n_pieces, gap);
}
void launch_utf8_piece_validate(hipStream_t s, const Utf8Piece* pieces, int n_pieces, int64_t max_rows, uint32_t* status) {
  if (n_pieces <= 0) return;
  const unsigned gx = (unsigned)std::min<int64_t>(64, (max_rows + 256) / 256);
  hipLaunchKernelGGL(k_utf8_piece_validate, dim3(gx, (unsigned)n_pieces), dim3(256), 0, s, pieces, status);
}
void launch_utf8_piece_offsets(hipStream_t s, const Utf8Piece* pieces, int n_pieces, int64_t max_rows) {
  if (n_pieces <= 0) return;
  const unsigned gx = (unsigned)std::min<int64_t>(64, (max_rows + 256) / 256);
  hipLaunchKernelGGL(k_utf8_piece_offsets, dim3(gx, (unsigned)n_pieces), dim3(256), 0, s, pieces);
}
void launch_utf8_piece_compact(hipStream_t s, const Utf8Piece* pieces, int n_pieces, int64_t max_bytes, const uint8_t* from, uint8_t* to) {
  if (n_pieces <= 0) return;
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(256, (max_bytes + 4095) / 4096));
  hipLaunchKernelGGL(k_utf8_piece_compact, dim3(gx, (unsigned)n_pieces), dim3(256), 0, s, pieces, from, to);
}
void launch_lz4_compress(hipStream_t s, const uint8_t* src, uint8_t* slots, const Lz4Block* blocks, int n_blocks, int32_t* csize) {
  if (n_blocks > 0) hipLaunchKernelGGL(k_lz4_compress, dim3((unsigned)n_blocks), dim3(64), 0, s, src, slots, blocks, n_blocks, csize);
}
void launch_lz4_layout(hipStream_t s, const Lz4Block* blocks, const int32_t* csize, const int32_t* buf_first_block, int n_buffers, int64_t* buf_off, int64_t* buf_len,
                       int64_t* blk_dst) {
  if (n_buffers <= 0) return;
  // (buf_off doubles as the scratch for the raw lengths of step A; step B overwrites it with the offsets)
  hipLaunchKernelGGL(k_lz4_layout_sizes, dim3((unsigned)n_buffers), dim3(256), 0, s, blocks, csize, buf_first_block, buf_len, blk_dst, buf_off);
  hipLaunchKernelGGL(k_lz4_layout_offsets, dim3(1), dim3(64), 0, s, n_buffers, (const int64_t*)buf_len, buf_off);
  hipLaunchKernelGGL(k_lz4_layout_place, dim3((unsigned)n_buffers), dim3(256), 0, s, blocks, buf_first_block, (const int64_t*)buf_off, (const int64_t*)buf_len, blk_dst);
}
void launch_lz4_pack(hipStream_t s, const uint8_t* src, const uint8_t* slots, const Lz4Block* blocks, int n_blocks, const int32_t* csize, const int32_t* blk_buffer,
                     const int32_t* buf_first_block, const int64_t* buf_off, const int64_t* buf_len, const int64_t* blk_dst, uint8_t* body) {
  if (n_blocks > 0)
    hipLaunchKernelGGL(k_lz4_pack, dim3((unsigned)n_blocks), dim3(256), 0, s, src, slots, blocks, csize, blk_buffer, buf_first_block, buf_off, buf_len, blk_dst, body);
}
void launch_lz4_decode(hipStream_t s, const uint8_t* src, int64_t src_bytes, uint8_t* dst, const Lz4Unit* units, int n_units, uint32_t* status) {
  if (n_units > 0) hipLaunchKernelGGL(k_lz4_decode, dim3((unsigned)n_units), dim3(64), 0, s, src, dst, units, n_units, src_bytes, status);
}
void launch_popcount_bits(hipStream_t s, const uint8_t* bits, int64_t n_bits, unsigned long long* out) {
  const int64_t words = (n_bits + 63) >> 6;
  if (words <= 0) return;
  const int blocks = (int)std::min<int64_t>(1024, (words + 255) / 256);
  hipLaunchKernelGGL(k_popcount_bits, dim3((unsigned)blocks), dim3(256), 0, s, (const u64*)bits, n_bits, out);
}

// ---------------------------------------------------------------- Snappy by pointer jumping
// The element chain of k_unpack_pages is serial AND moves the bytes: ~1000 cycles per element (LDS ring round trips, two wave barriers per
// copy), 10 MB/s per page.  Here the serial part only WALKS the elements: every output byte gets a 32-bit word -- bit 31 set: resolved, the
// byte's value in the low 8 bits (literals); clear: the position in the page it is a copy of (always earlier) -- and the copies are
// resolved afterwards by pointer jumping, all bytes of all pages in parallel: word[p] = word[word[p]] until bit 31 is set.  A chain of d
// copies (sorted 64-bit keys: every value's high bytes are a copy of the previous value's, d = rows per page) resolves in log2(d) rounds;
// updates in place only ever shorten a chain, and a resolved word carries the value itself, so there is nothing to order between
// threads.  The last kernel writes the bytes.  Malformed input: the same checks as k_unpack_pages (an offset of 0 or beyond the output so
// far, lengths past either end); a word still unresolved after ceil(log2(page bytes)) + 1 rounds cannot happen for a checked stream and
// is reported as malformed.
constexpr uint32_t PJ_DONE = 0x80000000u;
constexpr int PJ_BLOCK_BYTES = 4096;
// -- the front half, parallel too.  An element's size and output length depend only on the bytes at its own position, so next(p) and
// olen(p) are computed for EVERY position of the compressed stream as if an element started there; the real elements are the positions on
// the path start -> next(start) -> ... -> end, found by doubling: marked positions mark jump(p), then jump = jump o jump (ping-pong arrays; a
// round that marks nothing new has closed the path).  An exclusive scan of olen over the marked positions gives every element its place
// in the output, and the elements are expanded into resolve words all at once.  (Walking the elements one after the other, as
// k_unpack_pages and a first version of this path did, costs ~1000 cycles per element on a lone wave: 112 ms for a 1 MB page of sorted
// keys, whatever the copying costs.)
struct SnElem { int64_t size; uint32_t olen; int kind; uint32_t off; int64_t lit; };      // kind 0 literal (lit = first data byte), 1 copy
__device__ __forceinline__ SnElem sn_element(const uint8_t* __restrict__ in, const int64_t p, const int64_t L) {
  SnElem e{1, 0u, 0, 0u, 0};
  uint32_t b[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) b[k] = p + k < L ? in[p + k] : 0u;
  const uint32_t tag = b[0];
  if ((tag & 3) == 0) {
    int64_t len = (int64_t)(tag >> 2) + 1; int nb = 0;
    if (len > 60) { nb = (int)len - 60; uint32_t v = 0; for (int k = 0; k < nb; ++k) v |= b[1 + k] << (8 * k); len = (int64_t)v + 1; }
    e.size = 1 + nb + len; e.olen = (uint32_t)len; e.kind = 0; e.lit = p + 1 + nb;
  } else if ((tag & 3) == 1) { e.size = 2; e.olen = 4 + ((tag >> 2) & 7); e.kind = 1; e.off = ((tag >> 5) << 8) | b[1]; }
  else if ((tag & 3) == 2) { e.size = 3; e.olen = (tag >> 2) + 1; e.kind = 1; e.off = b[1] | (b[2] << 8); }
  else { e.size = 5; e.olen = (tag >> 2) + 1; e.kind = 1; e.off = b[1] | (b[2] << 8) | (b[3] << 16) | (b[4] << 24); }
  return e;
}
// one LZ4 sequence starting at p of a block [0, L): token, literal length (+255-runs), literals, then -- unless the literals end the block -- a
// 2-byte offset and the match length (+255-runs, + 4).  lit = first literal byte, olen = literals + match
struct Lz4Elem { int64_t size; int64_t lit_len, match_len; uint32_t off; int64_t lit; bool ok; };
__device__ __forceinline__ Lz4Elem lz4_element(const uint8_t* __restrict__ in, const int64_t p, const int64_t L) {
  Lz4Elem e{0, 0, 0, 0u, 0, false};
  int64_t q = p;
  const uint32_t token = in[q++];
  int64_t lit = token >> 4;
  // (a length cannot legitimately run longer than the block; positions that are NOT sequence starts -- inside a run of 0xFF bytes, say --
  // must not walk it to its end each: the bound keeps the all-positions pass linear in practice)
  const int max_run = (int)(L / 255) + 2;
  if (lit == 15) { uint32_t x; int guard = 0; do { if (q >= L || ++guard > max_run) return e; x = in[q++]; lit += x; } while (x == 255); }
  e.lit = q; e.lit_len = lit;
  if (lit > L - q) return e;
  q += lit;
  if (q == L) { e.size = q - p; e.ok = true; return e; }      // the last sequence of a block: literals only
  if (q + 2 > L) return e;
  e.off = (uint32_t)in[q] | ((uint32_t)in[q + 1] << 8); q += 2;
  int64_t ml = token & 15;
  if (ml == 15) { uint32_t x; int guard = 0; do { if (q >= L) return e; x = in[q++]; ml += x; (void)guard; } while (x == 255); }      // (bounded by the block's end; positions inside a 0xFF run leave through the literal bound above)
  e.match_len = ml + 4; e.size = q - p; e.ok = true;
  return e;
}
constexpr uint32_t SN_BAD = 0xFFFFFFFFu;
// per job (one wave): the raw prefix and stored pages are copied; a Snappy page's length preamble is read and its first element marked
__global__ __launch_bounds__(64) void k_sn_head(const uint8_t* __restrict__ src_base, uint8_t* __restrict__ dst_base, const UnpackJob* __restrict__ jobs, int n_jobs,
                                                uint8_t* __restrict__ mark, uint32_t* __restrict__ resolve, uint32_t* __restrict__ status) {
  const int u = (int)blockIdx.x;
  if (u >= n_jobs) return;
  const UnpackJob J = jobs[u];
  const int lane = lane_id();
  const uint8_t* in = src_base + J.src; uint8_t* out = dst_base + J.dst;
  if (J.raw_prefix > J.src_len || J.raw_prefix > J.dst_len) { if (lane == 0) atomicOr(status, 1u); return; }
  for (int64_t i = lane; i < J.raw_prefix; i += 64) out[i] = in[i];
  in += J.raw_prefix; out += J.raw_prefix;
  const int64_t L = J.src_len - J.raw_prefix, out_len = J.dst_len - J.raw_prefix;
  if (J.mode == 4) return;      // ZSTD frames: k_zstd_pages
  if (J.mode == 0) {
    if (L != out_len) { if (lane == 0) atomicOr(status, 1u); return; }
    for (int64_t i = lane; i < L; i += 64) out[i] = in[i];
    return;
  }
  if (J.mode == 3) {      // stored block of a linked frame: every byte a resolved word (k_snappy_emit writes the bytes)
    if (L != out_len) { if (lane == 0) atomicOr(status, 1u); return; }
    uint32_t* S = resolve + J.s_off;
    for (int64_t i = lane; i < L; i += 64) S[i] = PJ_DONE | (uint32_t)in[i];
    return;
  }
  if (J.mode == 2) { if (lane == 0) mark[J.c_off] = 1; return; }      // an LZ4 block starts with its first sequence
  if (lane == 0) {
    uint64_t ulen = 0; int sh = 0; int64_t ip = 0; bool bad = false;
    for (;;) { if (ip >= L || sh > 35) { bad = true; break; } const uint32_t c = in[ip++]; ulen |= (uint64_t)(c & 0x7F) << sh; if (!(c & 0x80)) break; sh += 7; }
    if (bad || (int64_t)ulen != out_len || out_len >= (int64_t)PJ_DONE) atomicOr(status, 1u);
    else mark[J.c_off + ip] = 1;      // (an empty page: ip == L, the terminal slot)
  }
}
// every position: where the element that would start here ends, and how many bytes it would produce
__global__ __launch_bounds__(256) void k_sn_next(const uint8_t* __restrict__ src_base, const uint2* __restrict__ cmap, const UnpackJob* __restrict__ jobs, uint32_t* __restrict__ jump, uint32_t* __restrict__ olen) {
  const uint2 m = cmap[blockIdx.x];
  const UnpackJob J = jobs[m.x];
  const uint8_t* in = src_base + J.src + J.raw_prefix;
  const int64_t L = J.src_len - J.raw_prefix;
  for (int k = 0; k < PJ_BLOCK_BYTES / 256; ++k) {
    const int64_t p = (int64_t)m.y + k * 256 + threadIdx.x;
    if (p > L) continue;
    uint32_t nx = (uint32_t)L, ol = 0;
    if (p < L) {
      if (J.mode == 2) {
        const Lz4Elem e = lz4_element(in, p, L);
        if (!e.ok || p + e.size > L || e.lit_len + e.match_len >= (int64_t)PJ_DONE) ol = SN_BAD; else { nx = (uint32_t)(p + e.size); ol = (uint32_t)(e.lit_len + e.match_len); }
      } else {
        const SnElem e = sn_element(in, p, L);
        if (p + e.size > L) ol = SN_BAD; else { nx = (uint32_t)(p + e.size); ol = e.olen; }
      }
    }
    jump[J.c_off + p] = nx; olen[J.c_off + p] = ol;
  }
}
// one doubling round: marked positions mark where they jump to; jump_out = jump_in o jump_in
__global__ __launch_bounds__(256) void k_sn_mark(const uint2* __restrict__ cmap, const UnpackJob* __restrict__ jobs, const uint32_t* __restrict__ jin, uint32_t* __restrict__ jout, uint8_t* __restrict__ mark,
                                                 const uint32_t* __restrict__ cnt_prev, uint32_t* __restrict__ cnt_cur, const int first) {
  const uint2 m = cmap[blockIdx.x];
  if (!first && cnt_prev[m.x] == 0) return;
  const UnpackJob J = jobs[m.x];
  const int64_t L = J.src_len - J.raw_prefix;
  uint32_t fresh = 0;
  for (int k = 0; k < PJ_BLOCK_BYTES / 256; ++k) {
    const int64_t p = (int64_t)m.y + k * 256 + threadIdx.x;
    if (p > L) continue;
    const uint32_t j = jin[J.c_off + p];
    if (mark[J.c_off + p] && !mark[J.c_off + j]) { mark[J.c_off + j] = 1; ++fresh; }
    jout[J.c_off + p] = jin[J.c_off + j];
  }
  for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
  if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&cnt_cur[m.x], fresh);
}
// scan input: the output length of every real element
__global__ __launch_bounds__(256) void k_sn_lens(const uint2* __restrict__ cmap, const UnpackJob* __restrict__ jobs, const uint8_t* __restrict__ mark, const uint32_t* __restrict__ olen,
                                                 int32_t* __restrict__ pos, uint32_t* __restrict__ status) {
  const uint2 m = cmap[blockIdx.x];
  const UnpackJob J = jobs[m.x];
  const int64_t L = J.src_len - J.raw_prefix;
  for (int k = 0; k < PJ_BLOCK_BYTES / 256; ++k) {
    const int64_t p = (int64_t)m.y + k * 256 + threadIdx.x;
    if (p > L) continue;
    uint32_t v = 0;
    if (mark[J.c_off + p] && p < L) { v = olen[J.c_off + p]; if (v == SN_BAD || v >= PJ_DONE) { v = 0; atomicOr(status, 1u); } }      // an element that runs past the end of the page
    if (p == L && !mark[J.c_off + p]) atomicOr(status, 1u);                                                                             // the path must end exactly at the end
    pos[J.c_off + p] = (int32_t)v;
  }
}
// every real element into its resolve words
__global__ __launch_bounds__(256) void k_sn_fill(const uint8_t* __restrict__ src_base, const uint2* __restrict__ cmap, const UnpackJob* __restrict__ jobs, const uint8_t* __restrict__ mark,
                                                 const int32_t* __restrict__ pos, uint32_t* __restrict__ resolve, uint32_t* __restrict__ status) {
  const uint2 m = cmap[blockIdx.x];
  const UnpackJob J = jobs[m.x];
  const uint8_t* in = src_base + J.src + J.raw_prefix;
  const int64_t L = J.src_len - J.raw_prefix, out_len = J.dst_len - J.raw_prefix;
  uint32_t* S = resolve + J.s_off;
  const int64_t base = pos[J.c_off];
  for (int k = 0; k < PJ_BLOCK_BYTES / 256; ++k) {
    const int64_t p = (int64_t)m.y + k * 256 + threadIdx.x;
    if (p > L || !mark[J.c_off + p]) continue;
    const int64_t o = (int64_t)pos[J.c_off + p] - base;
    if (p == L) { if (o != out_len) atomicOr(status, 1u); continue; }      // the elements must produce exactly the announced length
    if (J.mode == 2) {
      const Lz4Elem e = lz4_element(in, p, L);
      if (!e.ok || p + e.size > L || o < 0 || o + e.lit_len + e.match_len > out_len) { atomicOr(status, 1u); continue; }
      for (int64_t i = 0; i < e.lit_len; ++i) S[o + i] = PJ_DONE | (uint32_t)in[e.lit + i];
      if (e.match_len) {
        const int64_t at = J.p_base + o + e.lit_len;      // position in the FRAME's output: a match may reach back into earlier blocks
        if (e.off == 0 || (int64_t)e.off > at) { atomicOr(status, 1u); continue; }
        const uint32_t from = (uint32_t)(at - e.off);
        uint32_t* M = S + o + e.lit_len;
        for (int64_t i = 0; i < e.match_len; ++i) M[i] = from + (uint32_t)((int64_t)e.off >= e.match_len ? i : i % (int64_t)e.off);
      }
      continue;
    }
    const SnElem e = sn_element(in, p, L);
    if (p + e.size > L || o < 0 || o + (int64_t)e.olen > out_len) { atomicOr(status, 1u); continue; }      // (o < 0: a corrupted stream whose announced lengths wrapped the 32-bit scan)
    if (e.kind == 0) { for (uint32_t i = 0; i < e.olen; ++i) S[o + i] = PJ_DONE | (uint32_t)in[e.lit + i]; }
    else {
      if (e.off == 0 || (int64_t)e.off > o) { atomicOr(status, 1u); continue; }
      const uint32_t from = (uint32_t)(o - e.off);
      for (uint32_t i = 0; i < e.olen; ++i) S[o + i] = from + (e.off >= e.olen ? i : i % e.off);
    }
  }
}
// one round: word[p] = word[word[p]] for every unresolved word of every page that still had one after the previous round
__global__ __launch_bounds__(256) void k_snappy_round(uint32_t* __restrict__ resolve, const uint2* __restrict__ blkmap, const UnpackJob* __restrict__ jobs, const uint32_t* __restrict__ cnt_prev,
                                                      uint32_t* __restrict__ cnt_cur, const int first) {
  const uint2 m = blkmap[blockIdx.x];
  if (!first && cnt_prev[m.x] == 0) return;
  const UnpackJob J = jobs[m.x];
  uint32_t* S = resolve + J.f_off;      // positions are the frame's (a page is its own frame: f_off = s_off, p_base = 0)
  const int64_t len = J.dst_len - J.raw_prefix;
  uint32_t left = 0;
#pragma unroll 4
  for (int k = 0; k < PJ_BLOCK_BYTES / 256; ++k) {
    const int64_t p = (int64_t)m.y + k * 256 + threadIdx.x;
    if (p < len) {
      const int64_t a = J.p_base + p;
      const uint32_t v = S[a];
      if (!(v & PJ_DONE)) {
        const uint32_t w = (int64_t)v < a ? S[v] : PJ_DONE;      // (v < a always for a checked stream; anything else ends here, harmlessly)
        S[a] = w;
        left += (w & PJ_DONE) ? 0u : 1u;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) left += __shfl_xor(left, o);
  if ((threadIdx.x & 63) == 0 && left) atomicAdd(&cnt_cur[m.x], left);
}
// the bytes: four resolved words -> one 32-bit store (the pages start 64-byte aligned)
__global__ __launch_bounds__(256) void k_snappy_emit(const uint32_t* __restrict__ resolve, const uint2* __restrict__ blkmap, const UnpackJob* __restrict__ jobs, uint8_t* __restrict__ dst_base,
                                                     uint32_t* __restrict__ status) {
  const uint2 m = blkmap[blockIdx.x];
  const UnpackJob J = jobs[m.x];
  const uint32_t* S = resolve + J.s_off;
  uint8_t* out = dst_base + J.dst + J.raw_prefix;
  const int64_t len = J.dst_len - J.raw_prefix;
  bool bad = false;
  for (int k = 0; k < PJ_BLOCK_BYTES / 1024; ++k) {
    const int64_t p = (int64_t)m.y + (int64_t)(k * 256 + threadIdx.x) * 4;
    if (p >= len) continue;
    if (p + 4 <= len && (((uintptr_t)(out + p)) & 3) == 0) {
      const uint4 v = *(const uint4*)(S + p);      // s_off and m.y are multiples of 4 words: aligned
      bad = bad || !((v.x & v.y & v.z & v.w) & PJ_DONE);
      *(uint32_t*)(out + p) = (v.x & 0xFF) | ((v.y & 0xFF) << 8) | ((v.z & 0xFF) << 16) | ((v.w & 0xFF) << 24);
    } else for (int64_t q = p; q < p + 4 && q < len; ++q) { const uint32_t v = S[q]; bad = bad || !(v & PJ_DONE); out[q] = (uint8_t)v; }
  }
  if (bad) atomicOr(status, 1u);
}
void launch_unpack_pages_pj(hipStream_t s, const uint8_t* src, uint8_t* dst, const UnpackJob* jobs, int n_jobs, const SnappyPjBuffers& B, uint32_t* status) {
  if (n_jobs <= 0) return;
  // front: which positions of the compressed streams are elements, and where their output goes
  hipLaunchKernelGGL(k_sn_head, dim3((unsigned)n_jobs), dim3(64), 0, s, src, dst, jobs, n_jobs, B.mark, B.resolve, status);
  if (B.n_cblocks <= 0 || B.n_blocks <= 0) return;
  hipLaunchKernelGGL(k_sn_next, dim3((unsigned)B.n_cblocks), dim3(256), 0, s, src, B.cmap, jobs, B.jump_a, B.olen);
  uint32_t* ja = B.jump_a; uint32_t* jb = B.jump_b;
  for (int r = 0; r < B.mark_rounds; ++r) {
    hipLaunchKernelGGL(k_sn_mark, dim3((unsigned)B.n_cblocks), dim3(256), 0, s, B.cmap, jobs, (const uint32_t*)ja, jb, B.mark, B.counts + (size_t)r * n_jobs, B.counts + (size_t)(r + 1) * n_jobs, r == 0 ? 1 : 0);
    std::swap(ja, jb);
  }
  int32_t* pos = (int32_t*)B.jump_b;      // (both jump arrays are free now)
  hipLaunchKernelGGL(k_sn_lens, dim3((unsigned)B.n_cblocks), dim3(256), 0, s, B.cmap, jobs, (const uint8_t*)B.mark, (const uint32_t*)B.olen, pos, status);
  launch_exclusive_scan_i32(s, pos, B.c_slots, B.scan_ws, B.scan_ws_bytes);
  hipLaunchKernelGGL(k_sn_fill, dim3((unsigned)B.n_cblocks), dim3(256), 0, s, src, B.cmap, jobs, (const uint8_t*)B.mark, (const int32_t*)pos, B.resolve, status);
  // back: copies resolved by pointer jumping, then the bytes
  uint32_t* c2 = B.counts + (size_t)(B.mark_rounds + 1) * n_jobs;
  for (int r = 0; r < B.rounds; ++r)
    hipLaunchKernelGGL(k_snappy_round, dim3((unsigned)B.n_blocks), dim3(256), 0, s, B.resolve, B.blkmap, jobs, c2 + (size_t)r * n_jobs, c2 + (size_t)(r + 1) * n_jobs, r == 0 ? 1 : 0);
  hipLaunchKernelGGL(k_snappy_emit, dim3((unsigned)B.n_blocks), dim3(256), 0, s, (const uint32_t*)B.resolve, B.blkmap, jobs, dst, status);
}

void launch_unpack_pages(hipStream_t s, const uint8_t* src, uint8_t* dst, const UnpackJob* jobs, int n_jobs, uint32_t* status) {
  if (n_jobs > 0) hipLaunchKernelGGL(k_unpack_pages, dim3((unsigned)n_jobs), dim3(64), 0, s, src, dst, jobs, n_jobs, status);
}

}  // namespace gpuq
