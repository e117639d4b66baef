// Handle types shared by the translation units of libgpuq.so (capi.cpp, exchange.cpp): not part of the C ABI.
#pragma once
#include "../../include/gpuq.h"
#include "devbuf.h"
#include "gpuq_dev.h"
#include <memory>
#include <string>
#include <vector>

using gpuq::i64;

struct gpuq_ctx {
  int device = 0; int cus = 256; size_t hbm = 0; std::string name, arch;
  int jit_mode = 1;                 // 0 off, 1 auto, 2 force
  bool jit_wait = false;            // auto: a large input (>= jit_min_rows) WAITS for its compile (jit = "wait"); default: the worker thread compiles, the interpreter runs meanwhile
  i64 jit_min_rows = 1ll << 21;
  std::string last_jit_error;       // auto mode: why the last specialisation fell back to the interpreter kernels
  int jit_launches = 0;
  int join_dense = 1;               // direct-addressed join tables for one narrow key of bounded range (gpuq_ctx_set_option "join_dense")
  i64 join_dense_ratio = 4096;      // ... while range <= ratio x keys (sparse domains sit behind a presence bitmap: nothing but the bitmap is initialised)
  int join_radix = 0;               // partitioned probe over a direct-addressed table: 0 off (default: measured 1.0-1.16x, pairs
                                    // leave probe order -- DESIGN.md section 3), 1 auto, 2 force ("join_radix")
  int join_radix_slice_log2 = 18;   // table entries per partition slice (2^18 x 4 B = 1 MiB: an XCD's L2 holds a few)
};

// a table whose buffers the library owns (gpuq_table_import_arrow, gpuq_exchange_*): freed by gpuq_table_free
struct ImportedCol { gpuq_column col{}; gpuq_field_info field{}; gpuq::DevBuf data, offsets, validity; };
struct gpuq_table {
  gpuq_ctx* ctx = nullptr; int64_t n_rows = 0; std::vector<std::unique_ptr<ImportedCol>> cols;
  std::vector<int64_t> piece_rows;      // a table that came out of an exchange: rows received from rank 0, 1, ... (in this order)
};
