// TPC-H-shaped synthetic table generator: bench / test input only.  Its own shared library (benchmarks/libgpuq_tpchgen.so, built by
// arrow-ballista_amd/build.py): nothing of it is part of libgpuq.so or include/gpuq.h.
// Distributions follow SURVEY.md §8(d); the schema is the reference's
// (benchmarks/src/bin/tpch.rs:864-957): Int64 keys, Decimal128(15,2) money, Date32 dates, Utf8 flags.
// Counter-based: value = f(seed, column id, row), so any row range can be produced independently on
// any GPU and restated bit-for-bit on the CPU (oracle/oracle.c: oracle_gen_lineitem / _orders / _customer / _supplier).
//
// Deviation from dbgen, stated once: every order has exactly 4 lineitems (dbgen: 1..7, mean 4), so
// lineitem row i belongs to order index i>>2 and offsets have a closed form.
#include "gpuq_kernels.h"      // mix64, the integer typedefs (csrc/, on the include path)
#include "gpuq_tpchgen.h"
#include <cstdio>

namespace gpuq {

struct LineitemCols {
  i64* l_orderkey; i64* l_suppkey;
  u64* l_quantity; u64* l_extendedprice; u64* l_discount; u64* l_tax;   // Decimal128 as (lo,hi)
  int32_t* l_shipdate;
  uint8_t* l_returnflag; int32_t* l_returnflag_off;
  uint8_t* l_linestatus; int32_t* l_linestatus_off;
};
struct OrdersCols { i64* o_orderkey; i64* o_custkey; int32_t* o_orderdate; int32_t* o_shippriority; };
struct CustomerCols { i64* c_custkey; i64* c_nationkey; uint8_t* c_mktsegment; int32_t* c_mktsegment_off; };
struct SupplierCols { i64* s_suppkey; i64* s_nationkey; };
static int gen_cus() { static int n = []() { int d = 0, v = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d); return v > 0 ? v : 256; }(); return n; }

__device__ __host__ __forceinline__ u64 gen_u64(u64 seed, u64 col, u64 row) {
  return mix64((seed + col * 0xD1B54A32D192ED03ull) ^ (row * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull));
}
// Range reduction on a folded 32-bit hash with a 32-bit modulus.  (Deliberate: hipcc 7.2 miscompiles
// `u64 % even_constant` when the remainder feeds 64-bit arithmetic -- the magic-number division keeps
// only 28 bits of the quotient's low word -- so no 64-bit modulo by a constant appears in this file.)
__device__ __host__ __forceinline__ uint32_t gen_mod(u64 seed, u64 col, u64 row, uint32_t m) {
  const u64 x = gen_u64(seed, col, row);
  return ((uint32_t)(x >> 32) ^ (uint32_t)x) % m;
}
__device__ __host__ __forceinline__ i64 order_key(i64 o) { return (o >> 3) * 32 + (o & 7) + 1; }   // sparse: 8 of every 32 ids
__device__ __host__ __forceinline__ int32_t order_date(u64 seed_orders, i64 o) { return 8035 + (int32_t)gen_mod(seed_orders, 4, (u64)o, 2406u); }

// column ids: lineitem 1..9, orders 1..5, customer 1..3, supplier 1..2
__global__ void __launch_bounds__(256) k_gen_lineitem(const u64 seed, const u64 seed_orders, const i64 row0, const i64 n, const i64 n_supp,
                                                      const LineitemCols c) {
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < n; j += (i64)gridDim.x * 256) {
    const i64 i = row0 + j;
    const i64 o = i >> 2;
    const int32_t odate = order_date(seed_orders, o);
    const int32_t ship = odate + 1 + (int32_t)gen_mod(seed, 1, (u64)i, 121u);
    const int32_t receipt = ship + 1 + (int32_t)gen_mod(seed, 2, (u64)i, 30u);
    if (c.l_orderkey) c.l_orderkey[j] = order_key(o);
    if (c.l_suppkey) c.l_suppkey[j] = 1 + (i64)gen_mod(seed, 3, (u64)i, (uint32_t)n_supp);
    if (c.l_quantity) { c.l_quantity[2 * j] = (u64)((gen_mod(seed, 4, (u64)i, 50u) + 1u) * 100u); c.l_quantity[2 * j + 1] = 0; }
    if (c.l_extendedprice) { c.l_extendedprice[2 * j] = (u64)(90100u + gen_mod(seed, 5, (u64)i, 10404851u)); c.l_extendedprice[2 * j + 1] = 0; }
    if (c.l_discount) { c.l_discount[2 * j] = (u64)gen_mod(seed, 6, (u64)i, 11u); c.l_discount[2 * j + 1] = 0; }
    if (c.l_tax) { c.l_tax[2 * j] = (u64)gen_mod(seed, 7, (u64)i, 9u); c.l_tax[2 * j + 1] = 0; }
    if (c.l_shipdate) c.l_shipdate[j] = ship;
    if (c.l_returnflag) {
      c.l_returnflag[j] = (receipt <= 9298) ? ((gen_u64(seed, 8, (u64)i) & 1) ? 'R' : 'A') : 'N';
      c.l_returnflag_off[j] = (int32_t)j;
      if (j == n - 1) c.l_returnflag_off[n] = (int32_t)n;
    }
    if (c.l_linestatus) {
      c.l_linestatus[j] = (ship > 9298) ? 'O' : 'F';
      c.l_linestatus_off[j] = (int32_t)j;
      if (j == n - 1) c.l_linestatus_off[n] = (int32_t)n;
    }
  }
}

__global__ void __launch_bounds__(256) k_gen_orders(const u64 seed, const i64 row0, const i64 n, const i64 n_cust, const OrdersCols c) {
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < n; j += (i64)gridDim.x * 256) {
    const i64 o = row0 + j;
    if (c.o_orderkey) c.o_orderkey[j] = order_key(o);
    if (c.o_custkey) {
      // custkey never a multiple of 3 (one third of customers place no orders)
      c.o_custkey[j] = 3 * (i64)gen_mod(seed, 2, (u64)o, (uint32_t)(n_cust / 3)) + 1 + (i64)((gen_u64(seed, 2, (u64)o) >> 40) & 1);
    }
    if (c.o_orderdate) c.o_orderdate[j] = order_date(seed, o);
    if (c.o_shippriority) c.o_shippriority[j] = 0;
  }
}

__constant__ const char kSegments[5][11] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"};
__constant__ const int kSegLen[5] = {10, 8, 9, 9, 9};

// Each run of 5 consecutive customers holds a random permutation of the 5 segments, so the Utf8
// offsets have a closed form (45 bytes per run) and the segment selectivity is exactly 1/5.
__device__ __forceinline__ void seg_perm(u64 x64, int (&perm)[5]) {
  uint32_t x = (uint32_t)(x64 >> 32) ^ (uint32_t)x64;
  int pool[5] = {0, 1, 2, 3, 4};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int r = (int)(x % (uint32_t)(5 - k)); x /= (uint32_t)(5 - k);
    int pick = 0, seen = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) { if (pool[q] >= 0) { if (seen == r) pick = q; ++seen; } }
    perm[k] = pool[pick];
#pragma unroll
    for (int q = 0; q < 5; ++q) if (q == pick) pool[q] = -1;
  }
}

__global__ void __launch_bounds__(256) k_gen_customer(const u64 seed, const i64 row0, const i64 n, const CustomerCols c) {
  // row0 must be a multiple of 5
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < n; j += (i64)gridDim.x * 256) {
    const i64 i = row0 + j;
    if (c.c_custkey) c.c_custkey[j] = i + 1;
    if (c.c_nationkey) c.c_nationkey[j] = (i64)gen_mod(seed, 2, (u64)i, 25u);
    if (c.c_mktsegment) {
      const i64 run = i / 5; const int m = (int)(i % 5);
      int perm[5]; seg_perm(gen_u64(seed, 3, (u64)run), perm);
      int off = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) if (k < m) off += kSegLen[perm[k]];
      int seg = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) if (k == m) seg = perm[k];
      const i64 base = (run - row0 / 5) * 45 + off;
      c.c_mktsegment_off[j] = (int32_t)base;
      for (int k = 0; k < kSegLen[seg]; ++k) c.c_mktsegment[base + k] = (uint8_t)kSegments[seg][k];
      if (j == n - 1) c.c_mktsegment_off[n] = (int32_t)(base + kSegLen[seg]);
    }
  }
}

__global__ void __launch_bounds__(256) k_gen_supplier(const u64 seed, const i64 row0, const i64 n, const SupplierCols c) {
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < n; j += (i64)gridDim.x * 256) {
    const i64 i = row0 + j;
    if (c.s_suppkey) c.s_suppkey[j] = i + 1;
    if (c.s_nationkey) c.s_nationkey[j] = (i64)gen_mod(seed, 2, (u64)i, 25u);
  }
}

static int ggrid(i64 n) {
  i64 need = (n + 255) / 256; if (need < 1) need = 1;
  const i64 cap = (i64)gen_cus() * 16;
  return (int)(need < cap ? need : cap);
}
void launch_gen_lineitem(hipStream_t s, u64 seed, u64 seed_orders, i64 row0, i64 n, i64 n_supp, const LineitemCols& c) {
  if (n > 0) hipLaunchKernelGGL(k_gen_lineitem, dim3(ggrid(n)), dim3(256), 0, s, seed, seed_orders, row0, n, n_supp, c);
}
void launch_gen_orders(hipStream_t s, u64 seed, i64 row0, i64 n, i64 n_cust, const OrdersCols& c) {
  if (n > 0) hipLaunchKernelGGL(k_gen_orders, dim3(ggrid(n)), dim3(256), 0, s, seed, row0, n, n_cust, c);
}
void launch_gen_customer(hipStream_t s, u64 seed, i64 row0, i64 n, const CustomerCols& c) {
  if (n > 0) hipLaunchKernelGGL(k_gen_customer, dim3(ggrid(n)), dim3(256), 0, s, seed, row0, n, c);
}
void launch_gen_supplier(hipStream_t s, u64 seed, i64 row0, i64 n, const SupplierCols& c) {
  if (n > 0) hipLaunchKernelGGL(k_gen_supplier, dim3(ggrid(n)), dim3(256), 0, s, seed, row0, n, c);
}


}  // namespace gpuq

// ---- the C entry points (benchmarks/tpchgen/gpuq_tpchgen.h): the current device, the caller's stream
extern "C" {
static thread_local char g_gen_err[256] = "";
const char* gpuq_tpchgen_last_error(void) { return g_gen_err; }
static int gen_done(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  std::snprintf(g_gen_err, sizeof(g_gen_err), "%s: %s", what, hipGetErrorString(e));
  return 2;
}
int gpuq_tpchgen_lineitem(void* stream, uint64_t seed, uint64_t seed_orders, int64_t row0, int64_t n, int64_t n_supp, const gpuq_lineitem_cols* c) {
  if (!c || n < 0 || n_supp < 1) { std::snprintf(g_gen_err, sizeof(g_gen_err), "bad arguments"); return 1; }
  gpuq::LineitemCols d{c->l_orderkey ? (gpuq::i64*)c->l_orderkey : nullptr, (gpuq::i64*)c->l_suppkey, (gpuq::u64*)c->l_quantity, (gpuq::u64*)c->l_extendedprice, (gpuq::u64*)c->l_discount,
                       (gpuq::u64*)c->l_tax, c->l_shipdate, c->l_returnflag, c->l_returnflag_off, c->l_linestatus, c->l_linestatus_off};
  gpuq::launch_gen_lineitem((hipStream_t)stream, seed, seed_orders, row0, n, n_supp, d);
  return gen_done("gpuq_tpchgen_lineitem");
}
int gpuq_tpchgen_orders(void* stream, uint64_t seed, int64_t row0, int64_t n, int64_t n_cust, const gpuq_orders_cols* c) {
  if (!c || n < 0 || n_cust < 3) { std::snprintf(g_gen_err, sizeof(g_gen_err), "bad arguments"); return 1; }
  gpuq::OrdersCols d{(gpuq::i64*)c->o_orderkey, (gpuq::i64*)c->o_custkey, c->o_orderdate, c->o_shippriority};
  gpuq::launch_gen_orders((hipStream_t)stream, seed, row0, n, n_cust, d);
  return gen_done("gpuq_tpchgen_orders");
}
int gpuq_tpchgen_customer(void* stream, uint64_t seed, int64_t row0, int64_t n, const gpuq_customer_cols* c) {
  if (!c || n < 0 || row0 % 5 != 0) { std::snprintf(g_gen_err, sizeof(g_gen_err), "bad arguments (row0 must be a multiple of 5)"); return 1; }
  gpuq::CustomerCols d{(gpuq::i64*)c->c_custkey, (gpuq::i64*)c->c_nationkey, c->c_mktsegment, c->c_mktsegment_off};
  gpuq::launch_gen_customer((hipStream_t)stream, seed, row0, n, d);
  return gen_done("gpuq_tpchgen_customer");
}
int gpuq_tpchgen_supplier(void* stream, uint64_t seed, int64_t row0, int64_t n, const gpuq_supplier_cols* c) {
  if (!c || n < 0) { std::snprintf(g_gen_err, sizeof(g_gen_err), "bad arguments"); return 1; }
  gpuq::SupplierCols d{(gpuq::i64*)c->s_suppkey, (gpuq::i64*)c->s_nationkey};
  gpuq::launch_gen_supplier((hipStream_t)stream, seed, row0, n, d);
  return gen_done("gpuq_tpchgen_supplier");
}
}  // extern "C"
