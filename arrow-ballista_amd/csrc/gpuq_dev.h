// gpuq device-side core: column views, the expression bytecode and its
// wave-uniform interpreter.  gfx950 (MI355X) only.
//
// What this replaces in the reference: DataFusion's PhysicalExpr::evaluate
// (column-at-a-time arrow kernels, one materialised array per expression node;
// parameter surface pinned by ballista/core/proto/datafusion.proto:1142-1180).
// Here an expression tree is compiled on the host (expr_compile.cpp) to a short
// register program that every lane runs on its own row; control flow is uniform
// across the wave, operands live in a VGPR register file indexed through
// s_set_gpr_idx (never scratch), intermediates never touch HBM.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
// hiprtc keeps the fixed-width integer names inside __hip_internal
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef signed int int32_t;
typedef unsigned int uint32_t;
#endif

namespace gpuq {

typedef unsigned long long u64;
typedef long long i64;
typedef __int128 i128;
typedef unsigned __int128 u128;

constexpr int MAX_COLS = 16;   // input columns referenced by one program
constexpr int NREG = 16;       // 128-bit virtual registers per lane
constexpr int MAX_INSNS = 192;   // (q19's JoinFilter: three OR-ed groups of six conjuncts over twelve container names)
constexpr int MAX_IMMS = 48;
constexpr int MAX_VIA = 3;     // index vectors (selection / join pair sides)
constexpr uint32_t NULL_ROW = 0xFFFFFFFFu;  // index-vector entry meaning "no row" (outer join)

// How a column's bytes become a 128-bit register value.
enum ColClass : int32_t {
  CC_I32 = 0,   // 4-byte signed (Int32, Date32) sign-extended
  CC_I64 = 1,   // 8-byte (Int64; Float64 bit pattern)
  CC_I128 = 2,  // 16-byte little-endian two's complement (Decimal128)
  CC_STR = 3,   // Utf8: first <=15 bytes big-endian in bits 127..8, length in bits 7..0
  CC_BIT = 4,   // Boolean (Arrow bit-packed) -> 0/1
  CC_U32 = 5,   // 4-byte unsigned (row ids)
  CC_I8 = 7, CC_I16 = 8, CC_U8 = 9, CC_U16 = 10,   // narrow integers: 1 / 2 bytes in memory, sign- / zero-extended on load, truncated on store
  CC_F32 = 11,  // Float32: 4 bytes in memory, the double of the same value in a register
  CC_STRQ = 6,  // Utf8 as CC_STR, for a column the program only compares for (in)equality with literals or tests for NULL: the length
                // byte alone tells a value beyond 15 bytes from every literal a register can hold, so such a value is not an error
};

struct DevCol {
  const void* data;
  const int32_t* offsets;   // CC_STR only
  const uint8_t* validity;  // Arrow validity bitmap or nullptr (all valid)
  int32_t cls;
  int32_t via;              // 0 = driving position, k>0 = row comes from via[k-1][pos]
};

enum Op : uint8_t {
  OP_NOP = 0,
  OP_IMM,    // dst <- imm[imm]
  OP_MOV,    // dst <- a
  OP_ADD, OP_SUB, OP_MUL, OP_MULW /* i64*i64 -> i128 */, OP_NEG, OP_DIV /* trunc */, OP_MOD,
  OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE,            // signed 128-bit compares -> 0/1
  OP_FADD, OP_FSUB, OP_FMUL, OP_FDIV, OP_FNEG, OP_FSQRT,  // f64 in lo
  OP_FEQ, OP_FNE, OP_FLT, OP_FLE, OP_FGT, OP_FGE,       // f64 total-order compares
  OP_I2F, OP_F2I,
  OP_AND, OP_OR, OP_NOT,                                // Kleene logic on 0/1 + null
  OP_ISNULL, OP_ISNOTNULL,
  OP_SELECT,  // dst <- (a is true) ? b : reg[imm]
  OP_SHL,     // dst <- a << imm   (imm 0..127)
  OP_BOR,     // dst <- a | b      (bitwise; key packing)
  OP_NULLIF0, // dst <- a, NULL when b == 0 (guards OP_DIV by zero -> NULL)
  OP_COALESCE0, // dst <- a, or 0 (non-null) when a is NULL
  OP_DATEPART,  // dst <- field imm (0 year, 1 month, 2 day) of the Date32 a (days since 1970-01-01), as an integer
  OP_SUBSTR,    // dst <- substr(a, start, len) of a packed Utf8 value, ASCII only; imm = (start - 1) | len << 8 (len 255 = to the end)
  OP_F32R,      // dst <- (double)(float)a: a double rounded to the nearest Float32 (Float32 arithmetic and casts)
};

struct DevInsn { uint8_t op, dst, a, b; uint32_t imm; };

// The instruction stream and immediates are uploaded once per compiled operator and read through a
// uniform pointer (scalar loads).  They must NOT sit in the by-value kernel argument: hipcc copies a
// dynamically indexed kernarg array into scratch.
struct DevCode {
  DevInsn insns[MAX_INSNS];
  u64 imm_lo[MAX_IMMS];
  u64 imm_hi[MAX_IMMS];
};

struct DevProgram {
  int32_t n_cols;
  int32_t n_insns;
  int32_t pred_reg;   // register holding the row predicate, -1 = keep every row
  int32_t n_via;
  DevCol cols[MAX_COLS];
  const uint32_t* via[MAX_VIA];
  const DevCode* code;   // device memory
  uint32_t* flags;       // device status word (FLAG_* bits)
  const u64* n_dev;      // deferred execution: when set, the row count is min(*n_dev, n) and the launch's n is only a bound -- a producer's
                         // count (join pairs, filter survivors, groups) is handed over on the device, the host never reads it between operators
};
// Loads of columns that are read by position, once, front to back.  GPUQ_NT_STREAM=1 (a tuning switch, GPUQ_JIT_DEFINES) makes them
// non-temporal so that a multi-GB scan does not wash a join table's bitmap out of the L2 / Infinity Cache.
#ifdef GPUQ_NT_STREAM
#define GPUQ_LD_STREAM(p) __builtin_nontemporal_load(p)
#else
#define GPUQ_LD_STREAM(p) (*(p))
#endif
__device__ __forceinline__ i64 rows_of(const DevProgram& P, const i64 n_bound) {
  if (P.n_dev) { const i64 nd = (i64)*P.n_dev; return nd < n_bound ? nd : n_bound; }
  return n_bound;
}

// launcher-side dispatch over the column-slot template parameter
#define GPUQ_DISPATCH_MAXC(ncols, CALL)          \
  do {                                           \
    if ((ncols) <= 2) { CALL(2); }               \
    else if ((ncols) <= 4) { CALL(4); }          \
    else if ((ncols) <= 8) { CALL(8); }          \
    else { CALL(16); }                           \
  } while (0)

constexpr uint32_t FLAG_STR_TRUNC = 1u;
constexpr uint32_t FLAG_GROUP_OVERFLOW = 2u;
constexpr uint32_t FLAG_TABLE_FULL = 4u;
constexpr uint32_t FLAG_DUP_BUILD_KEY = 8u;
constexpr uint32_t FLAG_OUT_OVERFLOW = 16u;
constexpr uint32_t FLAG_WIDE_MINMAX = 32u;  // MIN/MAX over a value outside the int64 range
constexpr uint32_t FLAG_SORT_LAYOUT = 64u;  // a row does not fit the composite sort key layout remembered from the previous run (deferred execution)

// Per-lane register file.  lo/hi MUST be separate plain u64 arrays local to the kernel: hipcc then
// keeps them in VGPRs and lowers wave-uniform dynamic indexing to s_set_gpr_idx_on.  Wrapped in a
// struct, or as an __int128 array, they are demoted to scratch (build() asserts ScratchSize == 0).
#define GPUQ_REGS_DECL ::gpuq::u64 rlo[::gpuq::NREG]; ::gpuq::u64 rhi[::gpuq::NREG]; uint32_t rnulls = 0
#define GPUQ_REGS_PARAM ::gpuq::u64 (&rlo)[::gpuq::NREG], ::gpuq::u64 (&rhi)[::gpuq::NREG], uint32_t& rnulls
#define GPUQ_REGS_CPARAM const ::gpuq::u64 (&rlo)[::gpuq::NREG], const ::gpuq::u64 (&rhi)[::gpuq::NREG], const uint32_t& rnulls
#define GPUQ_REGS rlo, rhi, rnulls

__device__ __forceinline__ i128 mk128(u64 lo, u64 hi) { return (i128)(((u128)hi << 64) | (u128)lo); }

__device__ __forceinline__ i64 f64_total_key(u64 bits) {
  // IEEE-754 totalOrder as a signed integer compare [UPSTREAM-KNOWLEDGE: arrow-ord 49 cmp kernels]
  i64 s = (i64)bits;
  return s ^ (i64)((u64)(s >> 63) >> 1);
}

// The first n (<= 15) bytes of a string as the packed form's two halves (byte k big-endian: bits 63-8k.. of hi for k < 8, of lo for
// k >= 8; the low byte of lo stays free for the length) from the ALIGNED 8-byte words that hold them.  A byte loop costs one dependent
// memory round trip per byte (a customer-segment compare at SF100 was 0.29 ms for 315 MB); at most three aligned words do it in one,
// and an aligned word that holds at least one byte of the string never leaves the string's own pages.
__device__ __forceinline__ void str15_assemble(u64 w0, u64 w1, u64 w2, uint32_t byte_off, int32_t n, u64& hi, u64& lo) {
  hi = 0; lo = 0;
  if (n <= 0) return;
  const uint32_t sh = byte_off * 8u;
  u64 s0 = sh ? ((w0 >> sh) | (w1 << (64u - sh))) : w0;      // little-endian byte stream from the string's first byte
  u64 s1 = sh ? ((w1 >> sh) | (w2 << (64u - sh))) : w1;
  if (n < 8) { s0 &= (1ull << (8 * n)) - 1; s1 = 0; }
  else if (n == 8) s1 = 0;
  else s1 &= (1ull << (8 * (n - 8))) - 1;                   // n <= 15: at most seven bytes of the second half
  hi = __builtin_bswap64(s0); lo = __builtin_bswap64(s1);
}
__device__ __forceinline__ void load_str15(const uint8_t* data, int32_t o0, int32_t len, u64& hi, u64& lo) {
  const int32_t n = len < 15 ? len : 15;
  if (n <= 0) { hi = 0; lo = 0; return; }
  const unsigned long long addr = (unsigned long long)data + (unsigned long long)(uint32_t)o0;
  const u64* w = (const u64*)(addr & ~7ull);
  const uint32_t off = (uint32_t)(addr & 7ull), span = off + (uint32_t)n;
  const u64 w0 = w[0], w1 = span > 8u ? w[1] : 0ull, w2 = span > 16u ? w[2] : 0ull;
  str15_assemble(w0, w1, w2, off, n, hi, lo);
}
// Calendar field of a day number (proleptic Gregorian; the days-from-civil inverse) [UPSTREAM-KNOWLEDGE: arrow-arith 49 temporal
// kernels behind datafusion's date_part]: which = 0 year, 1 month (1-12), 2 day of month (1-31).
__device__ __forceinline__ i64 date_part_of_days(i64 days, int which) {
  const i64 z = days + 719468;
  const i64 era = (z >= 0 ? z : z - 146096) / 146097;
  const i64 doe = z - era * 146097;                                           // [0, 146096]
  const i64 yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;      // [0, 399]
  const i64 doy = doe - (365 * yoe + yoe / 4 - yoe / 100);                    // [0, 365]
  const i64 mp = (5 * doy + 2) / 153;                                         // [0, 11]
  const i64 d = doy - (153 * mp + 2) / 5 + 1;
  const i64 m = mp < 10 ? mp + 3 : mp - 9;
  const i64 y = yoe + era * 400 + (m <= 2 ? 1 : 0);
  return which == 0 ? y : (which == 1 ? m : d);
}
// substr over the packed form (<= 15 bytes big-endian in bits 127..8, true length in bits 7..0): `skip` leading bytes dropped, at most
// `len` kept (255 = the rest).  Exact whenever skip + kept <= 15 even for a longer value (its first 15 bytes are all there); `bad`
// is raised when bytes beyond the packed prefix would be needed, or when a kept byte is not ASCII (SQL counts characters).
__device__ __forceinline__ u128 substr_packed(u128 x, uint32_t skip, uint32_t len, bool& bad) {
  const uint32_t L = (uint32_t)x & 0xFFu;
  uint32_t keep = L > skip ? L - skip : 0u;
  if (len != 255u && keep > len) keep = len;
  if (skip + keep > 15u) { bad = true; keep = skip < 15u ? 15u - skip : 0u; }
  const u128 body = (x >> 8) << 8;                                   // bytes only
  const u128 shifted = skip >= 16u ? (u128)0 : (body << (8u * skip));
  const u128 mask = keep ? (~(u128)0 << (128u - 8u * keep)) : (u128)0;
  const u128 out = shifted & mask;
  if ((out >> 8) & ((u128)0x80808080808080ull << 64 | (u128)0x8080808080808080ull)) bad = true;
  return out | (u128)keep;
}

// Truncating signed 128-bit division (hardware has none; the AMDGPU backend has no __divti3).
// Only used on tiny outputs (AVG finalisation) and explicit '/' expressions.
__device__ inline void divmod128(i128 n, i128 d, i128& q, i128& r) {
  bool nn = n < 0, dn = d < 0;
  u128 un = nn ? (u128)(-n) : (u128)n;
  u128 ud = dn ? (u128)(-d) : (u128)d;
  u128 uq = 0, ur = 0;
  if (ud != 0) {
    if ((ud >> 64) == 0 && (un >> 64) == 0) {
      uq = (u64)un / (u64)ud; ur = (u64)un % (u64)ud;
    } else {
      for (int i = 127; i >= 0; --i) {
        ur = (ur << 1) | ((un >> i) & 1);
        if (ur >= ud) { ur -= ud; uq |= ((u128)1 << i); }
      }
    }
  }
  q = (nn != dn) ? -(i128)uq : (i128)uq;
  r = nn ? -(i128)ur : (i128)ur;
}

#ifndef GPUQ_JIT
// ---------------------------------------------------------------- column load
// Three phases, so that a wave has every column's load in flight before it consumes any of them
// (one HBM round trip per 64-row step instead of one per column):
//   A   issue the raw first-level loads into per-slot scalar registers with NO ALU work on loaded data.
//       Any use -- a select against a default, a copy into an array element -- makes the compiler put
//       s_waitcnt vmcnt(0) right behind the load and serialises the columns.  The slots therefore live
//       in a recursive struct of scalars addressed by compile-time indices (arrays get vectorised and
//       loaded values are then *copied* into them).
//   B1  Utf8 only: issue the load of byte 0 of every string column (second level, again all in flight)
//   B2  convert (sign-extend / pack strings / apply validity) and move into the register file, one
//       assignment per register outside any branch.
// MAXC = number of column slots (2/4/8/16, chosen by the launcher from P.n_cols); every slot costs
// instructions whether or not a column sits in it.
struct RawSlot { uint32_t r0, r1, r2, r3, v, b; };
template <int N> struct RawSlots { RawSlot s; RawSlots<N - 1> rest; };
template <> struct RawSlots<0> {};
template <int C, int N> __device__ __forceinline__ RawSlot& raw_slot(RawSlots<N>& r) {
  if constexpr (C == 0) return r.s; else return raw_slot<C - 1>(r.rest);
}
// rows are passed as a struct of scalars: an array indexed by the (uniform) via number is turned into a
// dynamically indexed private array, i.e. scratch.
struct RowIdx { uint32_t r0, r1, r2, r3; };
__device__ __forceinline__ uint32_t slot_row(const DevCol& col, const RowIdx& rows) {
  const int v = col.via;
  return v == 1 ? rows.r1 : (v == 2 ? rows.r2 : (v == 3 ? rows.r3 : rows.r0));
}

template <int C, int MAXC>
__device__ __forceinline__ void load_phase_a(const DevProgram& P, const RowIdx& rows, RawSlots<MAXC>& raw) {
  if constexpr (C < MAXC) {
    RawSlot& r = raw_slot<C>(raw);
    if (C < P.n_cols) {
      const DevCol col = P.cols[C];
      const uint32_t row = slot_row(col, rows);
      const bool ok = (col.via == 0) || (row != NULL_ROW);
      if (ok) {
        if (col.validity) r.v = col.validity[row >> 3];
        switch (col.cls) {
          case CC_I32: case CC_U32: case CC_F32: r.r0 = ((const uint32_t*)col.data)[row]; break;
          case CC_I8: case CC_U8: r.r0 = ((const uint8_t*)col.data)[row]; break;
          case CC_I16: case CC_U16: r.r0 = ((const uint16_t*)col.data)[row]; break;
          case CC_I64: { const uint2 v = ((const uint2*)col.data)[row]; r.r0 = v.x; r.r1 = v.y; break; }
          case CC_I128: { const uint4 v = ((const uint4*)col.data)[row]; r.r0 = v.x; r.r1 = v.y; r.r2 = v.z; r.r3 = v.w; break; }
          case CC_BIT: r.r0 = ((const uint8_t*)col.data)[row >> 3]; break;
          case CC_STR: case CC_STRQ: r.r0 = (uint32_t)col.offsets[row]; r.r1 = (uint32_t)col.offsets[row + 1]; break;
          default: break;
        }
      }
    }
    load_phase_a<C + 1, MAXC>(P, rows, raw);
  }
}
template <int C, int MAXC>
__device__ __forceinline__ void load_phase_b1(const DevProgram& P, const RowIdx& rows, RawSlots<MAXC>& raw) {
  if constexpr (C < MAXC) {
    RawSlot& r = raw_slot<C>(raw);
    if (C < P.n_cols) {
      const DevCol col = P.cols[C];
      if (col.cls == CC_STR || col.cls == CC_STRQ) {
        const uint32_t row = slot_row(col, rows);
        const bool ok = (col.via == 0) || (row != NULL_ROW);
        if (ok && r.r1 != r.r0) r.b = ((const uint8_t*)col.data)[(int32_t)r.r0];
      }
    }
    load_phase_b1<C + 1, MAXC>(P, rows, raw);
  }
}
template <int C, int MAXC>
__device__ __forceinline__ void load_phase_b2(const DevProgram& P, const RowIdx& rows, RawSlots<MAXC>& raw, GPUQ_REGS_PARAM) {
  if constexpr (C < MAXC) {
    RawSlot& r = raw_slot<C>(raw);
    u64 lo = 0, hi = 0;
    if (C < P.n_cols) {
      const DevCol col = P.cols[C];
      const uint32_t row = slot_row(col, rows);
      const bool ok = (col.via == 0) || (row != NULL_ROW);
      bool isnull = !ok;
      if (ok && col.validity) isnull = !((r.v >> (row & 7)) & 1u);
      if (isnull) rnulls |= (1u << C);
      else {
        switch (col.cls) {
          case CC_I32: { const i64 v = (int32_t)r.r0; lo = (u64)v; hi = (u64)(v >> 63); break; }
          case CC_U32: case CC_U8: case CC_U16: lo = r.r0; break;
          case CC_I8: { const i64 v = (int8_t)r.r0; lo = (u64)v; hi = (u64)(v >> 63); break; }
          case CC_I16: { const i64 v = (int16_t)r.r0; lo = (u64)v; hi = (u64)(v >> 63); break; }
          case CC_F32: { lo = (u64)__double_as_longlong((double)__uint_as_float(r.r0)); hi = (u64)((i64)lo >> 63); break; }
          case CC_I64: lo = (u64)r.r0 | ((u64)r.r1 << 32); hi = (u64)((i64)lo >> 63); break;
          case CC_I128: lo = (u64)r.r0 | ((u64)r.r1 << 32); hi = (u64)r.r2 | ((u64)r.r3 << 32); break;
          case CC_BIT: lo = (r.r0 >> (row & 7)) & 1u; break;
          case CC_STR: case CC_STRQ: {
            const int32_t o0 = (int32_t)r.r0, o1 = (int32_t)r.r1;
            const int32_t len = o1 - o0;
            if (len > 15 && col.cls == CC_STR) { if (P.flags) atomicOr(P.flags, FLAG_STR_TRUNC); }
            u64 h, l;
            load_str15((const uint8_t*)col.data, o0, len, h, l);
            hi = h; lo = l | (u64)(len < 255 ? len : 255);
            break;
          }
          default: break;
        }
      }
    }
    rlo[C] = lo; rhi[C] = hi;
    load_phase_b2<C + 1, MAXC>(P, rows, raw, GPUQ_REGS);
  }
}

template <int MAXC>
__device__ __forceinline__ void load_columns(const DevProgram& P, i64 pos, GPUQ_REGS_PARAM) {
  RowIdx rows;
  rows.r0 = (uint32_t)pos;
  rows.r1 = (P.n_via > 0) ? P.via[0][pos] : 0u;
  rows.r2 = (P.n_via > 1) ? P.via[1][pos] : 0u;
  rows.r3 = (P.n_via > 2) ? P.via[2][pos] : 0u;
  rnulls = 0;
  RawSlots<MAXC> raw;
  load_phase_a<0, MAXC>(P, rows, raw);
  load_phase_b1<0, MAXC>(P, rows, raw);
  load_phase_b2<0, MAXC>(P, rows, raw, GPUQ_REGS);
}

// ---------------------------------------------------------------- interpreter
// The instruction stream is read through a CONSTANT-address-space pointer: with a wave-uniform index
// that makes every fetch an s_load (scalar cache) instead of a per-lane global_load + vmcnt wait,
// which would put one L2 round trip in front of every interpreted instruction.
typedef const DevCode __attribute__((address_space(4))) * ConstCode;
__device__ __forceinline__ ConstCode const_code(const DevProgram& P) {
  return (ConstCode)(unsigned long long)(const void*)P.code;
}
__device__ __forceinline__ void run_program(const DevProgram& P, GPUQ_REGS_PARAM) {
  ConstCode code = const_code(P);
  for (int p = 0; p < P.n_insns; ++p) {
    const unsigned long long raw = *(const unsigned long long __attribute__((address_space(4)))*)&code->insns[p];
    DevInsn in;
    in.op = (uint8_t)raw; in.dst = (uint8_t)(raw >> 8); in.a = (uint8_t)(raw >> 16); in.b = (uint8_t)(raw >> 24); in.imm = (uint32_t)(raw >> 32);
    const int op = __builtin_amdgcn_readfirstlane((int)in.op);
    const int d = __builtin_amdgcn_readfirstlane((int)in.dst);
    const int a = __builtin_amdgcn_readfirstlane((int)in.a);
    const int b = __builtin_amdgcn_readfirstlane((int)in.b);
    const uint32_t imm = (uint32_t)__builtin_amdgcn_readfirstlane((int)in.imm);
    const u64 alo = rlo[a], ahi = rhi[a], blo = rlo[b], bhi = rhi[b];
    const bool an = (rnulls >> a) & 1, bn = (rnulls >> b) & 1;
    u64 zlo = 0, zhi = 0;
    bool zn = an || bn;  // default null propagation for binary ops
    switch (op) {
      case OP_IMM: zlo = code->imm_lo[imm]; zhi = code->imm_hi[imm]; zn = false; break;
      case OP_MOV: zlo = alo; zhi = ahi; zn = an; break;
      case OP_ADD: { i128 z = mk128(alo, ahi) + mk128(blo, bhi); zlo = (u64)z; zhi = (u64)((u128)z >> 64); break; }
      case OP_SUB: { i128 z = mk128(alo, ahi) - mk128(blo, bhi); zlo = (u64)z; zhi = (u64)((u128)z >> 64); break; }
      case OP_MUL: { i128 z = (i128)((u128)mk128(alo, ahi) * (u128)mk128(blo, bhi)); zlo = (u64)z; zhi = (u64)((u128)z >> 64); break; }
      case OP_MULW: { i128 z = (i128)(i64)alo * (i128)(i64)blo; zlo = (u64)z; zhi = (u64)((u128)z >> 64); break; }
      case OP_NEG: { i128 z = -mk128(alo, ahi); zlo = (u64)z; zhi = (u64)((u128)z >> 64); zn = an; break; }
      case OP_DIV: case OP_MOD: {
        i128 q, r; i128 den = mk128(blo, bhi);
        if (den == 0) { zn = true; q = 0; r = 0; } else divmod128(mk128(alo, ahi), den, q, r);
        i128 z = (op == OP_DIV) ? q : r; zlo = (u64)z; zhi = (u64)((u128)z >> 64); break;
      }
      case OP_EQ: zlo = (alo == blo) & (ahi == bhi); break;
      case OP_NE: zlo = (alo != blo) | (ahi != bhi); break;
      case OP_LT: zlo = mk128(alo, ahi) < mk128(blo, bhi); break;
      case OP_LE: zlo = mk128(alo, ahi) <= mk128(blo, bhi); break;
      case OP_GT: zlo = mk128(alo, ahi) > mk128(blo, bhi); break;
      case OP_GE: zlo = mk128(alo, ahi) >= mk128(blo, bhi); break;
      case OP_FADD: zlo = (u64)__double_as_longlong(__longlong_as_double((i64)alo) + __longlong_as_double((i64)blo)); break;
      case OP_FSUB: zlo = (u64)__double_as_longlong(__longlong_as_double((i64)alo) - __longlong_as_double((i64)blo)); break;
      case OP_FMUL: zlo = (u64)__double_as_longlong(__longlong_as_double((i64)alo) * __longlong_as_double((i64)blo)); break;
      case OP_FDIV: zlo = (u64)__double_as_longlong(__longlong_as_double((i64)alo) / __longlong_as_double((i64)blo)); break;
      case OP_FNEG: zlo = alo ^ 0x8000000000000000ull; zn = an; break;
      case OP_FSQRT: zlo = (u64)__double_as_longlong(sqrt(__longlong_as_double((i64)alo))); zn = an; break;
      case OP_FEQ: zlo = f64_total_key(alo) == f64_total_key(blo); break;
      case OP_FNE: zlo = f64_total_key(alo) != f64_total_key(blo); break;
      case OP_FLT: zlo = f64_total_key(alo) < f64_total_key(blo); break;
      case OP_FLE: zlo = f64_total_key(alo) <= f64_total_key(blo); break;
      case OP_FGT: zlo = f64_total_key(alo) > f64_total_key(blo); break;
      case OP_FGE: zlo = f64_total_key(alo) >= f64_total_key(blo); break;
      case OP_I2F: {
        // i128 -> f64, correctly rounded for |a| < 2^64 (hi is sign extension), else via two halves
        double v = ((i64)ahi == ((i64)alo >> 63)) ? (double)(i64)alo
                                                  : ((double)(i64)ahi * 18446744073709551616.0 + (double)alo);
        zlo = (u64)__double_as_longlong(v); zn = an; break;
      }
      case OP_F2I: { i64 v = (i64)__longlong_as_double((i64)alo); zlo = (u64)v; zhi = (u64)(v >> 63); zn = an; break; }
      case OP_F32R: { zlo = (u64)__double_as_longlong((double)(float)__longlong_as_double((i64)alo)); zhi = (u64)((i64)zlo >> 63); zn = an; break; }
      case OP_AND: {  // Kleene: false AND x = false
        bool af = !an && alo == 0, bf = !bn && blo == 0;
        zn = !(af || bf) && (an || bn);
        zlo = (af || bf) ? 0 : 1; break;
      }
      case OP_OR: {
        bool at = !an && alo != 0, bt = !bn && blo != 0;
        zn = !(at || bt) && (an || bn);
        zlo = (at || bt) ? 1 : 0; break;
      }
      case OP_NOT: zlo = alo ? 0 : 1; zn = an; break;
      case OP_ISNULL: zlo = an; zn = false; break;
      case OP_ISNOTNULL: zlo = !an; zn = false; break;
      case OP_SELECT: {
        bool t = !an && alo != 0;
        const int e = (int)imm;
        zlo = t ? blo : rlo[e]; zhi = t ? bhi : rhi[e];
        zn = t ? bn : (bool)((rnulls >> e) & 1); break;
      }
      case OP_SHL: { u128 z = (u128)mk128(alo, ahi) << imm; zlo = (u64)z; zhi = (u64)(z >> 64); zn = an; break; }
      case OP_DATEPART: { const i64 z = date_part_of_days((i64)alo, (int)imm); zlo = (u64)z; zhi = (u64)(z >> 63); zn = an; break; }
      case OP_SUBSTR: { bool bad = false; const u128 z = substr_packed((u128)mk128(alo, ahi), imm & 0xFFu, (imm >> 8) & 0xFFu, bad); zlo = (u64)z; zhi = (u64)(z >> 64); zn = an;
                        if (bad && !an && P.flags) atomicOr(P.flags, FLAG_STR_TRUNC); break; }
      case OP_BOR: zlo = alo | blo; zhi = ahi | bhi; break;
      case OP_NULLIF0: zlo = alo; zhi = ahi; zn = an || bn || (blo == 0 && bhi == 0); break;
      case OP_COALESCE0: zlo = an ? 0 : alo; zhi = an ? 0 : ahi; zn = false; break;
      default: zn = false; break;
    }
    rlo[d] = zlo; rhi[d] = zhi;
    rnulls = (rnulls & ~(1u << d)) | ((uint32_t)zn << d);
  }
}

// Row predicate after run_program: NULL counts as false (SQL WHERE; FilterExec drops null).
__device__ __forceinline__ bool row_passes(const DevProgram& P, GPUQ_REGS_CPARAM) {
  if (P.pred_reg < 0) return true;
  const int r = __builtin_amdgcn_readfirstlane(P.pred_reg);
  return rlo[r] != 0 && !((rnulls >> r) & 1);
}

#define GPUQ_EVAL(MAXC, P, pos) (load_columns<MAXC>(P, pos, GPUQ_REGS), run_program(P, GPUQ_REGS), row_passes(P, GPUQ_REGS))
#else
// JIT build: the row front-end is a generated, typed, straight-line function (jit_codegen.cpp) with the
// same contract: fills the registers the sink reads and returns the row predicate.
__device__ __forceinline__ bool gpuq_jit_eval(const DevProgram& P, i64 pos, GPUQ_REGS_PARAM);
#define GPUQ_EVAL(MAXC, P, pos) gpuq_jit_eval(P, pos, GPUQ_REGS)
#endif

// ---------------------------------------------------------------- hashing
// 64-bit mixer (splitmix64 finaliser).  This is gpuq's own partition/hash function;
// DataFusion's ahash(RandomState::with_seeds(0,0,0,0)) is CPU-feature dependent and is
// not a portable contract -- both sides of an exchange use this one.
__device__ __host__ __forceinline__ u64 mix64(u64 x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31; return x;
}
__device__ __host__ __forceinline__ u64 hash_combine(u64 h, u64 lo, u64 hi, bool isnull) {
  // per-column combine: h' = mix(h ^ mix(value)); NULL hashes as a fixed constant
  u64 v = isnull ? 0x9E3779B97F4A7C15ull : mix64(lo ^ mix64(hi + 0x632BE59BD9B4E019ull));
  return mix64(h * 31 + v + 0x9E3779B97F4A7C15ull);
}

}  // namespace gpuq
