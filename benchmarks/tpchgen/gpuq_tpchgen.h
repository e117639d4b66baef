/* Synthetic TPC-H-shaped input for benches and tests: device-resident columns in Arrow layout, produced by
   benchmarks/libgpuq_tpchgen.so (benchmarks/tpchgen/tpchgen.hip).  NOT part of the product: libgpuq.so and include/gpuq.h know nothing
   of it.  Runs on the current HIP device, on the caller's stream; returns 0, or non-zero with gpuq_tpchgen_last_error(). */
#ifndef GPUQ_TPCHGEN_H
#define GPUQ_TPCHGEN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct gpuq_lineitem_cols {
  int64_t* l_orderkey; int64_t* l_suppkey;
  void* l_quantity; void* l_extendedprice; void* l_discount; void* l_tax; /* Decimal128(15,2), 16 B/row */
  int32_t* l_shipdate;
  uint8_t* l_returnflag; int32_t* l_returnflag_off;  /* Utf8: n bytes, n+1 offsets */
  uint8_t* l_linestatus; int32_t* l_linestatus_off;
} gpuq_lineitem_cols;
typedef struct gpuq_orders_cols { int64_t* o_orderkey; int64_t* o_custkey; int32_t* o_orderdate; int32_t* o_shippriority; } gpuq_orders_cols;
typedef struct gpuq_customer_cols { int64_t* c_custkey; int64_t* c_nationkey; uint8_t* c_mktsegment; int32_t* c_mktsegment_off; } gpuq_customer_cols;
typedef struct gpuq_supplier_cols { int64_t* s_suppkey; int64_t* s_nationkey; } gpuq_supplier_cols;
/* Any pointer may be NULL (column skipped).  Rows [row0, row0+n) of the table. */
int gpuq_tpchgen_lineitem(void* stream, uint64_t seed, uint64_t seed_orders, int64_t row0, int64_t n, int64_t n_supp,
                      const gpuq_lineitem_cols* cols);
int gpuq_tpchgen_orders(void* stream, uint64_t seed, int64_t row0, int64_t n, int64_t n_cust, const gpuq_orders_cols* cols);
int gpuq_tpchgen_customer(void* stream, uint64_t seed, int64_t row0, int64_t n, const gpuq_customer_cols* cols);
int gpuq_tpchgen_supplier(void* stream, uint64_t seed, int64_t row0, int64_t n, const gpuq_supplier_cols* cols);

const char* gpuq_tpchgen_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
