/* TPC-H q1 through the C ABI alone (C99, no Python, no torch): what a cgo / JNI / Rust FFI binding of libgpuq does.
 *
 *   q1_native PLAN.json N_ROWS [SEED]
 *
 * 1. create a device context, allocate the seven lineitem columns q1 reads in device memory (Arrow physical layout) and fill
 *    them with the benches' synthetic TPC-H-shaped generator (benchmarks/libgpuq_tpchgen.so: a library of its own, not part of libgpuq);
 * 2. hand the stage plan (JSON mirror of the reference's PhysicalPlanNode tree, see include/gpuq.h) to the native plan
 *    executor: gpuq_plan_create + gpuq_plan_execute -- the counterpart of `plan.execute(0, ctx)` in
 *    ballista/core/src/execution_plans/shuffle_writer.rs:255;
 * 3. copy the result columns back and print one line per group (decimals as unscaled integers).
 * tests/test_gpu_c_example.py runs it on the GPU box and compares the lines with the oracle. */
#include "../include/gpuq.h"
#include "gpuq_tpchgen.h"      /* benchmarks/tpchgen/ (-I) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                                                          \
  do {                                                                                                       \
    int rc_ = (call);                                                                                        \
    if (rc_ != GPUQ_OK) { fprintf(stderr, "%s failed (%d): %s / %s\n", #call, rc_, gpuq_last_error(ctx), gpuq_plan_last_error()); return 1; } \
  } while (0)

static char* read_file(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  char* b = (char*)malloc((size_t)n + 1);
  if (fread(b, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(b); return NULL; }
  b[n] = 0; fclose(f); return b;
}

/* 128-bit little-endian two's complement -> decimal text (q1's sums stay far below 2^63 at these sizes, but print exactly) */
static void print_i128(const unsigned char* p) {
  unsigned __int128 v = 0; int neg;
  for (int i = 15; i >= 0; --i) v = (v << 8) | p[i];
  neg = (p[15] & 0x80) != 0;
  if (neg) v = ~v + 1;
  char buf[48]; int k = 47; buf[k] = 0;
  do { buf[--k] = (char)('0' + (int)(v % 10)); v /= 10; } while (v);
  if (neg) buf[--k] = '-';
  fputs(buf + k, stdout);
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s PLAN.json N_ROWS [SEED]\n", argv[0]); return 2; }
  const long long n = atoll(argv[2]);
  const unsigned long long seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 1;
  char* plan_json = read_file(argv[1]);
  if (!plan_json) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  gpuq_ctx* ctx = gpuq_ctx_create(0, NULL);
  if (!ctx) { fprintf(stderr, "gpuq_ctx_create: %s\n", gpuq_last_error(NULL)); return 1; }

  /* ---- input: l_quantity, l_extendedprice, l_discount, l_tax (Decimal128(15,2)), l_returnflag, l_linestatus (Utf8), l_shipdate (Date32) */
  void *qty, *ext, *disc, *tax, *ship, *rf, *rfo, *ls, *lso;
  const size_t pad = 64;
  CHECK(gpuq_buffer_alloc(ctx, (size_t)n * 16 + pad, &qty));  CHECK(gpuq_buffer_alloc(ctx, (size_t)n * 16 + pad, &ext));
  CHECK(gpuq_buffer_alloc(ctx, (size_t)n * 16 + pad, &disc)); CHECK(gpuq_buffer_alloc(ctx, (size_t)n * 16 + pad, &tax));
  CHECK(gpuq_buffer_alloc(ctx, (size_t)n * 4 + pad, &ship));
  CHECK(gpuq_buffer_alloc(ctx, (size_t)n + pad, &rf)); CHECK(gpuq_buffer_alloc(ctx, (size_t)(n + 1) * 4 + pad, &rfo));
  CHECK(gpuq_buffer_alloc(ctx, (size_t)n + pad, &ls)); CHECK(gpuq_buffer_alloc(ctx, (size_t)(n + 1) * 4 + pad, &lso));
  gpuq_lineitem_cols gen; memset(&gen, 0, sizeof gen);
  gen.l_quantity = qty; gen.l_extendedprice = ext; gen.l_discount = disc; gen.l_tax = tax; gen.l_shipdate = (int32_t*)ship;
  gen.l_returnflag = (uint8_t*)rf; gen.l_returnflag_off = (int32_t*)rfo; gen.l_linestatus = (uint8_t*)ls; gen.l_linestatus_off = (int32_t*)lso;
  if (gpuq_tpchgen_lineitem(NULL, seed, 2 /* orders seed */, 0, n, 10000, &gen) != 0) { fprintf(stderr, "generator: %s\n", gpuq_tpchgen_last_error()); return 1; }

  /* column order = the MemoryExec schema of the plan file */
  gpuq_column cols[7]; memset(cols, 0, sizeof cols);
  void* dec[4] = {qty, ext, disc, tax};
  for (int i = 0; i < 4; ++i) { cols[i].type = GPUQ_DECIMAL128; cols[i].precision = 15; cols[i].scale = 2; cols[i].data = dec[i]; cols[i].length = n; }
  cols[4].type = GPUQ_UTF8; cols[4].data = rf; cols[4].offsets = (const int32_t*)rfo; cols[4].length = n;
  cols[5].type = GPUQ_UTF8; cols[5].data = ls; cols[5].offsets = (const int32_t*)lso; cols[5].length = n;
  cols[6].type = GPUQ_DATE32; cols[6].data = ship; cols[6].length = n;
  gpuq_input in; memset(&in, 0, sizeof in);
  in.cols = cols; in.n_cols = 7; in.n_rows = n; in.n_via = 0;

  /* ---- the stage plan, executed inside the library */
  gpuq_plan* plan = NULL; gpuq_result* res = NULL;
  CHECK(gpuq_plan_create(ctx, plan_json, &plan));
  CHECK(gpuq_plan_execute(plan, NULL, 0, &in, 1, &res));

  /* ---- result: 2 PACKED15 strings, 4 + 3 Decimal128, 1 Int64 */
  const long long rows = (long long)gpuq_result_num_rows(res);
  const int nc = gpuq_result_num_columns(res);
  unsigned char** host = (unsigned char**)calloc((size_t)nc, sizeof(unsigned char*));
  gpuq_field_info* fi = (gpuq_field_info*)calloc((size_t)nc, sizeof(gpuq_field_info));
  for (int c = 0; c < nc; ++c) {
    gpuq_column col;
    CHECK(gpuq_result_column(res, c, &col, &fi[c]));
    const size_t bytes = (size_t)rows * (size_t)fi[c].width;
    host[c] = (unsigned char*)malloc(bytes + 16);
    if (bytes) CHECK(gpuq_copy_d2h(ctx, NULL, host[c], col.data, bytes));
  }
  for (long long r = 0; r < rows; ++r) {
    for (int c = 0; c < nc; ++c) {
      const unsigned char* p = host[c] + (size_t)r * (size_t)fi[c].width;
      if (c) putchar('|');
      if (fi[c].type == GPUQ_UTF8) {          /* PACKED15: length in byte 0, characters big-endian from byte 15 down */
        const int len = p[0];
        for (int k = 0; k < len && k < 15; ++k) putchar(p[15 - k]);
      } else if (fi[c].type == GPUQ_DECIMAL128) print_i128(p);
      else if (fi[c].type == GPUQ_INT64 || fi[c].type == GPUQ_UINT64) { long long v; memcpy(&v, p, 8); printf("%lld", v); }
      else if (fi[c].type == GPUQ_INT32 || fi[c].type == GPUQ_DATE32) { int v; memcpy(&v, p, 4); printf("%d", v); }
      else printf("?");
    }
    putchar('\n');
  }
  char metrics[8192];
  if (gpuq_plan_metrics(plan, metrics, sizeof metrics) == GPUQ_OK) fprintf(stderr, "metrics: %s\n", metrics);
  gpuq_result_free(res); gpuq_plan_free(plan);
  void* all[9] = {qty, ext, disc, tax, ship, rf, rfo, ls, lso};
  for (int i = 0; i < 9; ++i) gpuq_buffer_free(ctx, all[i]);
  gpuq_jit_quiesce();      /* no background compile may outlive main (include/gpuq.h) */
  gpuq_ctx_free(ctx);
  free(plan_json);
  return 0;
}
