// libgpuq.so C ABI (include/gpuq.h): operator descriptors -> compiled device programs,
// workspace management, strategy selection, launches.  Host logic only; all arithmetic on the
// data path runs in the HIP kernels (kernels_*.hip).  There is deliberately no CPU fallback.
#include "../../include/gpuq.h"
#include "expr_compile.h"
#include "gpuq_kernels.h"
#include "jit_runtime.h"
#include "devbuf.h"
#include "gpuq_internal.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <memory>
#include <mutex>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <map>
#include <deque>
#include <functional>
#include <vector>

using namespace gpuq;

namespace {

thread_local std::string g_last_error;

u64 next_pow2(u64 v) { u64 r = 1; while (r < v) r <<= 1; return r; }

}  // namespace


struct gpuq_timer { hipEvent_t a = nullptr, b = nullptr; };

enum OpKind { K_FILTER, K_PROJECT, K_AGG, K_JOIN_BUILD, K_JOIN_PROBE, K_SORT, K_PARTITION };

struct gpuq_op {
  gpuq_ctx* ctx = nullptr;
  OpKind kind = K_FILTER;
  Schema in_schema;
  CompiledProgram prog;
  DevBuf code_dev, flags_dev;
  std::vector<gpuq_field_info> out_fields;
  // aggregate
  std::string mode = "Single", strategy = "auto";
  AggSpec agg{}; KeySpec keys{};
  std::vector<DType> key_types, acc_types;
  std::vector<int> acc_bits;        // |argument| < 2^bits per accumulator (type-derived): lets the specialised kernel drop range checks
  std::map<std::pair<const void*, int>, int> jit_runs;      // (program, sink kernel) -> small-input runs so far (background tier)
  std::map<std::tuple<const void*, int, size_t>, const JitFn*> jit_fns;      // (program, sink kernel, specialisation) -> compiled function
  struct PostChunk { CompiledProgram prog; DevBuf code; int first_out = 0; };
  std::deque<PostChunk> posts; Schema post_schema;
  i64 expected_groups = 0;
  i64 last_groups = -1;             // groups of this operator's previous run
  int last_path = 0;                // ... and the way it took: 1 LDS dictionary, 2 global hash table, 3 radix-partitioned
  // join
  int join_type = JT_INNER; int null_eq = 0; bool build_side_rows = true;
  bool has_semi = false; KeySpec semi_keys{};      // chain fusion: the keys this build's rows are looked up with in another join's table
  // sort
  SortSpec sort{}; i64 fetch = -1; bool sort_guess_failed = false, join_guess_failed = false;
  // partition
  uint32_t nparts = 0;
  // deferred execution (include/gpuq.h): what the last completed synchronous run learned, and what a deferred run may leave behind
  struct { void* data = nullptr; u64* valid = nullptr; bool done = false; } sort_dec;      // gpuq_sort_run_keys: where the sorted key column goes
  DType sort_key0;                  // sort: type of the first key expression
  std::string refuse;               // the operator compiles (its output types are known) but cannot run: why
  std::string label;                // descriptor "label": appended to the run-time compiled kernels' names (a plan node id: profiles tell call sites apart)
  bool deferred = false, defer_client = false;
  uint32_t expect_flags = 0;        // status bits a deferred run is allowed to raise (a build side known to hold duplicate keys)
  struct { bool valid = false, dense = false, sparse_bits = false, has_dups = false; i64 kmin = 0; u64 krange = 0; u64 n_slots = 0; } jb;
  struct { bool valid = false; int path = 0, gmax = 0; u64 est = 0; bool use_lds = false; i64 groups = 0; } ag;      // path: 1 LDS dictionary, 2 global hash table
  struct { bool valid = false; SortPack K{}; int total = 0; } so;
  // scratch
  DevBuf ws[10];
  // pinned host words for the small device->host reads (flags, counts): a pageable destination makes every such copy a
  // staged, blocking transfer
  uint32_t* pin = nullptr;
  uint32_t* pinned() { if (!pin) { if (hipHostMalloc((void**)&pin, 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pin = nullptr; } } return pin; }
  // profiling of the dominant kernel
  bool profile = false; hipEvent_t ev0 = nullptr, ev1 = nullptr; float kernel_ms = 0; int launches = 0; bool ev_pending = false;
  // ... and of everything one call queues (the dominant kernel plus table initialisation, scans, compaction, result projection)
  hipEvent_t ev2 = nullptr, ev3 = nullptr; float total_ms = 0, last_total_ms = 0; bool ev_total_pending = false;
  ~gpuq_op() { for (hipEvent_t e : {ev0, ev1, ev2, ev3}) if (e) (void)hipEventDestroy(e); if (pin) (void)hipHostFree(pin); }
};

struct gpuq_join_table {
  gpuq_ctx* ctx = nullptr;
  KeySpec keys{}; HashTable T{}; int null_eq = 0; bool has_present = true;
  DevBuf slots, dense, dense_bits, next, visited, present, ws_bitmap, ws_counts;
  i64 bound = 0;
  bool visited_ready = false;
  bool has_dups = false;   // some key occurs on more than one build row -> chained probe
};

namespace {

void set_err(gpuq_ctx*, const std::string& m) { g_last_error = m; }      // per calling thread, like errno: contexts are shared between task threads

template <class F> int guarded(gpuq_ctx* ctx, F&& f) {
  try { f(); return GPUQ_OK; }
  catch (const HipError& e) { set_err(ctx, e.what()); return GPUQ_ERR_HIP; }
  catch (const Unsupported& e) { set_err(ctx, e.what()); return GPUQ_ERR_UNSUPPORTED; }
  catch (const Capacity& e) { set_err(ctx, e.what()); return GPUQ_ERR_CAPACITY; }
  catch (const Retry& e) { set_err(ctx, e.what()); return GPUQ_ERR_RETRY; }
  catch (const std::bad_alloc&) { set_err(ctx, "out of host memory"); return GPUQ_ERR_INTERNAL; }
  catch (const std::exception& e) { set_err(ctx, e.what()); return GPUQ_ERR_INVALID; }
}

gpuq_field_info make_field(const std::string& name, const DType& t, bool nullable) {
  gpuq_field_info f{};
  std::snprintf(f.name, sizeof(f.name), "%s", name.c_str());
  f.type = t.id; f.precision = t.p; f.scale = t.s; f.nullable = nullable;
  f.repr = (t.id == T_UTF8) ? GPUQ_REPR_PACKED15 : GPUQ_REPR_ARROW;
  f.width = (t.id == T_BOOL) ? 0 : type_width(t);
  return f;
}

thread_local bool g_upload = true;   // false inside gpuq_compile_check (no device); per thread: other threads create operators meanwhile
void upload_code(const CompiledProgram& p, DevBuf& dst) {
  if (!g_upload) return;
  dst.ensure(sizeof(DevCode));
  HIPCHECK(hipMemcpy(dst.p, &p.code, sizeof(DevCode), hipMemcpyHostToDevice));
}

// Bind the call's column pointers to a compiled program.
DevProgram bind_program(const CompiledProgram& cp, const Schema& schema, const DevCode* code_dev, uint32_t* flags_dev, const gpuq_input* in) {
  if (!in) throw std::runtime_error("input is NULL");
  if (in->n_cols != (int)schema.fields.size())
    throw std::runtime_error("input has " + std::to_string(in->n_cols) + " columns, operator expects " + std::to_string(schema.fields.size()));
  if (in->n_via < 0 || in->n_via > MAX_VIA) throw std::runtime_error("n_via out of range");
  if (in->n_rows < 0 || in->n_rows > 0xFFFFFFFEll) throw std::runtime_error("n_rows out of range (max 2^32-2 positions per call)");
  DevProgram P{};
  P.n_cols = (int)cp.col_field.size(); P.n_insns = cp.n_insns; P.pred_reg = cp.pred_reg; P.n_via = in->n_via;
  for (int k = 0; k < in->n_via; ++k) { if (!in->via[k] && in->n_rows > 0) throw std::runtime_error("index vector is NULL"); P.via[k] = in->via[k]; }
  P.code = code_dev; P.flags = flags_dev; P.n_dev = (const u64*)in->n_rows_dev;
  for (size_t c = 0; c < cp.col_field.size(); ++c) {
    const int fi = cp.col_field[c];
    const Field& f = schema.fields[fi];
    const gpuq_column& col = in->cols[fi];
    if (col.type != f.type.id || (f.type.id == T_DECIMAL128 && (col.precision != f.type.p || col.scale != f.type.s)))
      throw std::runtime_error("column '" + f.name + "': type does not match the operator's input schema (" + f.type.to_string() + ")");
    if (f.side > in->n_via) throw std::runtime_error("column '" + f.name + "' needs index vector " + std::to_string(f.side));
    if (!col.data && col.length > 0 && in->n_rows > 0) throw std::runtime_error("column '" + f.name + "': data is NULL");
    if (!f.nullable && f.side == 0 && col.validity) { /* declared non-nullable but carries a bitmap: honour the bitmap */ }
    DevCol& d = P.cols[c];
    d.data = col.data; d.offsets = col.offsets; d.validity = col.validity;
    d.cls = f.raw128 ? CC_I128 : col_class_for(f.type);
    if (d.cls == CC_STR && !f.raw128 && !col.offsets && col.length > 0) throw std::runtime_error("Utf8 column '" + f.name + "' has no offsets");
    if (f.raw128 && f.type.id == T_UTF8) d.cls = CC_I128;
    if (d.cls == CC_STR && c < cp.col_loose.size() && cp.col_loose[c]) d.cls = CC_STRQ;
    d.via = f.side;
  }
  return P;
}

// entry points that decide things on the host from the exact row count cannot take a device-side one
void need_exact_rows(const gpuq_input* in, const char* what) {
  if (in && in->n_rows_dev) throw std::runtime_error(std::string(what) + ": the input carries a device-side row count (n_rows_dev); this entry point needs the exact count");
}
void reset_flags(gpuq_op* op, hipStream_t s) { HIPCHECK(hipMemsetAsync(op->flags_dev.p, 0, 4, s)); }
// words [0, n) of the op's status block (flags, pad, n_groups, pad) -> host
void read_status(gpuq_op* op, hipStream_t s, uint32_t* out, int n) {
  uint32_t* h = op->pinned();
  HIPCHECK(hipMemcpyAsync(h ? h : out, op->flags_dev.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  if (h) for (int i = 0; i < n; ++i) out[i] = h[i];
}
uint32_t read_flags(gpuq_op* op, hipStream_t s) { uint32_t f = 0; read_status(op, s, &f, 1); return f; }
void raise_flags(uint32_t f) {
  if (f & FLAG_STR_TRUNC) throw Unsupported("a Utf8 value longer than 15 bytes reached a device string comparison/key (PACKED15 limit)");
  if (f & FLAG_WIDE_MINMAX) throw Unsupported("MIN/MAX over a value outside the int64 range is not supported on device");
  if (f & FLAG_DUP_BUILD_KEY) throw Unsupported("duplicate build keys without a chain buffer");
  if (f & FLAG_OUT_OVERFLOW) throw Capacity("join output capacity exceeded; see the pair count for the required size");
  if (f & FLAG_TABLE_FULL) throw std::runtime_error("hash table full");
  if (f & FLAG_GROUP_OVERFLOW) throw Capacity("group capacity exceeded");
  if (f & FLAG_SORT_LAYOUT) throw std::runtime_error("a row does not fit the remembered sort key layout");
}

struct ProfScope {
  gpuq_op* op; hipStream_t s;
  ProfScope(gpuq_op* o, hipStream_t st) : op(o), s(st) {
    if (op->profile) {
      if (!op->ev0) { HIPCHECK(hipEventCreate(&op->ev0)); HIPCHECK(hipEventCreate(&op->ev1)); }
      if (op->ev_pending) { float ms = 0; HIPCHECK(hipEventSynchronize(op->ev1)); HIPCHECK(hipEventElapsedTime(&ms, op->ev0, op->ev1)); op->kernel_ms += ms; op->ev_pending = false; }
      HIPCHECK(hipEventRecord(op->ev0, s));
    }
  }
  ~ProfScope() { if (op->profile) { (void)hipEventRecord(op->ev1, s); op->ev_pending = true; op->launches++; } }
};

// events around EVERYTHING a run call queues on its stream (gpuq_op_profile_total)
struct ProfTotal {
  gpuq_op* op; hipStream_t s; bool on = false;
  ProfTotal(gpuq_op* o, hipStream_t st) : op(o), s(st) {
    if (op->profile) {
      if (!op->ev2) { HIPCHECK(hipEventCreate(&op->ev2)); HIPCHECK(hipEventCreate(&op->ev3)); }
      if (op->ev_total_pending) {
        // never wait here (a deferred step must not meet a host round trip because it is being measured): a previous interval that
        // has not finished yet -- the same operator twice in a row, e.g. the take projections -- leaves this call unmeasured
        if (hipEventQuery(op->ev3) != hipSuccess) { (void)hipGetLastError(); return; }
        float ms = 0; HIPCHECK(hipEventElapsedTime(&ms, op->ev2, op->ev3)); op->total_ms += ms; op->ev_total_pending = false;
      }
      HIPCHECK(hipEventRecord(op->ev2, s)); on = true;
    }
  }
  ~ProfTotal() { if (on) { (void)hipEventRecord(op->ev3, s); op->ev_total_pending = true; } }
};

// Route the next launch of sink kernel `kernel_id` to the hiprtc-specialised function when the context's
// policy asks for it.  force: failures are errors.  auto: fall back to the (always present) interpreter kernels.
struct JitScope {
  bool active = false;
  JitScope(gpuq_op* op, const CompiledProgram& cp, int kernel_id, i64 n, const std::string& spec = std::string()) {
    gpuq_ctx* c = op->ctx;
    if (cp.jit_src.empty() || c->jit_mode == 0) return;
    const bool big = c->jit_mode == 1 && n >= c->jit_min_rows;
    // force: wait for the compile (failures are errors).  wait (jit = "wait"): a large input waits for it too (steady-state benches).
    // auto (the default): NOBODY waits -- a large input hands its source to the worker thread on its first run and runs the interpreter
    // kernels meanwhile (a one-shot SF100 q3 used to spend 1.3 s in hiprtc for a 4 ms query); a small input is only worth a compile
    // when its program keeps coming back (third run on).
    const bool use = c->jit_mode == 2 || (big && c->jit_wait);
    const bool hot = !use && (big || ++op->jit_runs[{(const void*)&cp, kernel_id}] > 2);
    if (!use && !hot) return;
    try {
      // the process-wide cache is keyed by the whole generated source (kilobytes of text to concatenate and compare): an
      // operator remembers the functions it has resolved
      const auto fkey = std::make_tuple((const void*)&cp, kernel_id, std::hash<std::string>()(spec));
      const JitFn* f = nullptr;
      auto hit = op->jit_fns.find(fkey);
      if (hit != op->jit_fns.end()) { f = hit->second; if (!f && c->jit_mode == 2) throw std::runtime_error("the run-time source of kernel " + std::to_string(kernel_id) + " failed to compile earlier"); }
      else {
        const std::string lab = op->label.empty() ? std::string() : "//@label " + op->label + "\n";
        try { f = use ? jit_get(cp.jit_src + spec + lab, kernel_id) : jit_try_get(cp.jit_src + spec + lab, kernel_id); }
        catch (...) { op->jit_fns[fkey] = nullptr; throw; }      // a source that does not compile is not compiled again on every call
        if (f) op->jit_fns[fkey] = f;
      }
      if (!f) return;
      jit_override().fn = f->fn; jit_override().kernel_id = kernel_id; active = true; c->jit_launches++;
    } catch (const std::exception& e) {
      if (c->jit_mode == 2) throw Unsupported(e.what());
      c->last_jit_error = e.what();
    }
  }
  ~JitScope() { jit_override().fn = nullptr; jit_override().kernel_id = 0; }
};

// ---------------------------------------------------------------- key layout
KeySpec make_keyspec(const std::vector<int>& regs, const std::vector<DType>& types, bool null_word) {
  if (regs.size() > (size_t)MAX_KEYS) throw Unsupported("more than " + std::to_string(MAX_KEYS) + " key columns");
  KeySpec K{};
  K.n_keys = (int)regs.size(); K.null_word = null_word ? 1 : 0;
  int w = 0;
  for (size_t k = 0; k < regs.size(); ++k) {
    K.key_reg[k] = regs[k];
    const bool wide = types[k].id == T_DECIMAL128 || types[k].id == T_UTF8;
    K.key_wide[k] = wide;
    K.word_reg[w] = regs[k]; K.word_half[w] = 0; ++w;
    if (wide) { K.word_reg[w] = regs[k]; K.word_half[w] = 1; ++w; }
  }
  if (null_word) { K.word_reg[w] = 0; K.word_half[w] = 2; ++w; }
  K.key_words = w;
  return K;
}

// ---------------------------------------------------------------- aggregate compilation
struct AccDef { int kind; NodeP arg; DType type; };

int find_or_add_acc(std::vector<AccDef>& accs, int kind, NodeP arg, const DType& type) {
  for (size_t i = 0; i < accs.size(); ++i)
    if (accs[i].kind == kind && ((!arg && !accs[i].arg) || (arg && accs[i].arg && arg->key == accs[i].arg->key))) return (int)i;
  if ((int)accs.size() >= MAX_ACCS) throw Unsupported("aggregate needs more than " + std::to_string(MAX_ACCS) + " accumulators");
  accs.push_back({kind, arg, type}); return (int)accs.size() - 1;
}
DType t_of(int id) { DType t; t.id = id; return t; }
DType dec_t(int p, int s) { DType t; t.id = T_DECIMAL128; t.p = std::min(p, 38); t.s = std::min(s, 38); return t; }

// How one SQL aggregate maps to accumulators (indices into accs) and to output columns.
struct AggPlan {
  std::string fn, name;
  DType arg_type; bool arg_nullable = false;
  int acc_sum = -1, acc_cnt = -1, acc_mm = -1;   // sum / count / min-max accumulators
  int acc_sx = -1, acc_sy = -1, acc_sxx = -1, acc_syy = -1, acc_sxy = -1;   // variance family: f64 power sums
  bool is_float = false;
};

// VARIANCE / STDDEV / COVARIANCE / CORRELATION (datafusion.proto:639-645).  The reference keeps Welford-style
// running (count, mean, m2[, algo_const]) states [UPSTREAM-KNOWLEDGE]; a data-parallel device cannot follow a
// row order, so the accumulators are the order-free power sums n, Sx, Sy, Sxx, Syy, Sxy in f64 and the
// reference's state columns are derived from them (mean = Sx/n, m2 = Sxx - Sx^2/n, algo = Sxy - Sx*Sy/n).
// Results agree with the reference to f64 rounding (tolerance stated in tests/test_gpu_operators.py).
int var_family(const std::string& fn) {   // 1: one-argument (x), 2: two-argument (x, y)
  if (fn == "VARIANCE" || fn == "VAR" || fn == "VAR_SAMP" || fn == "VARIANCE_POP" || fn == "VAR_POP" ||
      fn == "STDDEV" || fn == "STDDEV_SAMP" || fn == "STDDEV_POP") return 1;
  if (fn == "COVARIANCE" || fn == "COVAR" || fn == "COVAR_SAMP" || fn == "COVARIANCE_POP" || fn == "COVAR_POP" ||
      fn == "CORRELATION" || fn == "CORR") return 2;
  return 0;
}
bool var_is_pop(const std::string& fn) { return fn.size() > 4 && fn.compare(fn.size() - 4, 4, "_POP") == 0; }
bool var_is_corr(const std::string& fn) { return fn == "CORRELATION" || fn == "CORR"; }
bool var_is_stddev(const std::string& fn) { return fn.compare(0, 6, "STDDEV") == 0; }

void compile_aggregate(gpuq_op* op, const Json& d) {
  op->mode = d.get_str("mode", "Single");
  const bool is_final = (op->mode == "Final" || op->mode == "FinalPartitioned");
  const bool emit_state = (op->mode == "Partial");
  if (!is_final && !emit_state && op->mode != "Single") throw std::runtime_error("unknown aggregate mode '" + op->mode + "'");
  op->strategy = d.get_str("strategy", "auto");
  op->expected_groups = d.get_i64("expected_groups", 0);
  ExprCompiler ec(op->in_schema);
  if (d.has("predicate")) ec.add_predicate(ec.from_json(d.at("predicate")));
  std::vector<NodeP> key_nodes; std::vector<std::string> key_names;
  if (d.has("group_expr")) for (const Json& g : d.at("group_expr").a) {
    NodeP n = ec.from_json(g.at("expr"));
    key_nodes.push_back(n); key_names.push_back(g.get_str("name", "group" + std::to_string(key_nodes.size() - 1)));
  }
  std::vector<AccDef> accs; std::vector<AggPlan> plans;
  // More than MAX_KEYS group columns (q10 groups by seven, q18 by five): the table still holds at most MAX_KEYS keys of up to 128 bits, so
  // narrow keys are PACKED -- each biased to a non-negative number of `bits + 1` bits (+ 1 bit "is NULL"), shifted and OR-ed into 126-bit
  // composites -- and the table groups by the composites.  The declared key columns are unpacked again over the GROUPS (the result
  // projection divides by powers of two), so the table carries no extra state.  Utf8 / float keys take a slot of their own (the
  // executor hands long or many Utf8 keys over as dictionary codes, which pack well).
  std::vector<NodeP> orig_keys = key_nodes; std::vector<std::string> orig_names = key_names;
  struct Packed { int slot = -1, shift = 0, width = 0; bool own = false; };      // own: the key IS column `slot` of the result
  std::vector<Packed> pk(orig_keys.size());
  const bool packed = orig_keys.size() > (size_t)MAX_KEYS;
  if (packed) {
    struct Slot { std::vector<size_t> ks; int bits = 0; bool solo = false; };
    std::vector<Slot> slots;
    for (size_t k = 0; k < orig_keys.size(); ++k) {
      const NodeP& n = orig_keys[k];
      const int b = (n->type.id == T_UTF8 || n->type.is_float() || n->type.id == T_BOOL) ? 0 : n->bits + 1 + (n->nullable ? 1 : 0);
      if (b == 0 || b > 126) { Slot sl; sl.ks.push_back(k); sl.solo = true; slots.push_back(sl); continue; }
      bool placed = false;
      for (auto& sl : slots) if (!sl.solo && sl.bits + b <= 126) { sl.ks.push_back(k); sl.bits += b; placed = true; break; }
      if (!placed) { Slot sl; sl.ks.push_back(k); sl.bits = b; slots.push_back(sl); }
    }
    if (slots.size() > (size_t)MAX_KEYS) op->refuse = std::to_string(orig_keys.size()) + " group-by columns need " + std::to_string(slots.size()) + " packed keys (" + std::to_string(MAX_KEYS) + " are held)";
    key_nodes.clear(); key_names.clear();
    const DType wide = dec_t(38, 0);
    for (size_t si = 0; si < slots.size() && si < (size_t)MAX_KEYS; ++si) {
      const Slot& sl = slots[si];
      if (sl.solo) { pk[sl.ks[0]].slot = (int)key_nodes.size(); pk[sl.ks[0]].own = true; key_nodes.push_back(orig_keys[sl.ks[0]]); key_names.push_back(orig_names[sl.ks[0]]); continue; }
      NodeP acc; int shift = 0;
      for (size_t k : sl.ks) {
        const NodeP& n = orig_keys[k];
        NodeP v = ec.raw(OP_ADD, wide, n->nullable, n->bits + 1, {n, ec.lit_int(wide, (i128)1 << n->bits)});      // |value| < 2^bits  ->  [0, 2^(bits+1))
        if (n->nullable) {
          v = ec.coalesce0(v);
          NodeP flag = ec.raw(OP_SHL, wide, false, n->bits + 2, {ec.raw(OP_MOV, wide, false, 1, {ec.is_null(n, false)})}, (uint32_t)(n->bits + 1));
          v = ec.raw(OP_BOR, wide, false, n->bits + 2, {v, flag});
        }
        NodeP sh = shift ? ec.raw(OP_SHL, wide, false, 127, {v}, (uint32_t)shift) : v;
        acc = acc ? ec.raw(OP_BOR, wide, false, 127, {acc, sh}) : sh;
        pk[k].slot = (int)key_nodes.size(); pk[k].shift = shift; pk[k].width = n->bits + 1 + (n->nullable ? 1 : 0);
        shift += pk[k].width;
      }
      key_nodes.push_back(acc); key_names.push_back("__packed" + std::to_string(si));
    }
  }
  size_t state_col = orig_keys.size();   // Final modes: state columns follow the group columns positionally
  const Json& aggs = d.at("aggr_expr");
  for (const Json& a : aggs.a) {
    AggPlan pl; pl.fn = a.at("fn").str(); pl.name = a.get_str("name", pl.fn);
    for (auto& ch : pl.fn) ch = (char)std::toupper(ch);
    if (a.get_bool("distinct", false)) throw Unsupported("DISTINCT aggregates are not supported on device");
    if (!is_final) {
      NodeP arg = a.has("expr") ? ec.from_json(a.at("expr")) : nullptr;
      // per-aggregate FILTER (AggregateExecNode.filter_expr, datafusion.proto:1437-1450): agg(x) FILTER (WHERE p) is agg over the rows
      // where p is true, i.e. agg(CASE WHEN p THEN x END) -- every accumulator here skips NULL arguments; COUNT(*) counts the 1s
      if (a.has("filter") && !a.at("filter").is_null()) {
        NodeP p = ec.from_json(a.at("filter"));
        if (p->type.id != T_BOOL) throw std::runtime_error("aggregate FILTER must be boolean");
        if (!arg) arg = ec.lit_int(t_of(T_INT64), 1);
        arg = ec.select(p, arg, ec.lit_null(arg->type));
        if (a.has("expr2")) throw Unsupported("FILTER on a two-argument aggregate");
      }
      if (pl.fn == "COUNT") {
        if (!arg || !arg->nullable) pl.acc_cnt = find_or_add_acc(accs, ACC_COUNT_STAR, nullptr, t_of(T_INT64));
        else pl.acc_cnt = find_or_add_acc(accs, ACC_COUNT, arg, t_of(T_INT64));
        pl.arg_type = t_of(T_INT64);
      } else {
        if (!arg) throw std::runtime_error(pl.fn + " needs an argument");
        pl.arg_type = arg->type; pl.arg_nullable = arg->nullable || key_nodes.empty();   // ungrouped: zero input rows -> NULL
        auto count_of = [&](NodeP x) { return x->nullable ? find_or_add_acc(accs, ACC_COUNT, x, t_of(T_INT64)) : find_or_add_acc(accs, ACC_COUNT_STAR, nullptr, t_of(T_INT64)); };
        if (pl.fn == "SUM" || pl.fn == "AVG") {
          if (arg->type.is_decimal()) {
            pl.acc_sum = find_or_add_acc(accs, ACC_SUM, arg, dec_t(arg->type.p + 10, arg->type.s));
          } else if (arg->type.is_int() && pl.fn == "SUM") {
            // sum_return_type [UPSTREAM-KNOWLEDGE]: signed integers of any width -> Int64, unsigned -> UInt64
            const DType st = t_of(arg->type.is_unsigned() ? T_UINT64 : T_INT64);
            NodeP x = ec.cast(arg, st);
            pl.acc_sum = find_or_add_acc(accs, ACC_SUM, x, st);
          } else if (arg->type.is_float() || arg->type.is_int()) {
            NodeP x = ec.cast(arg, t_of(T_FLOAT64)); pl.is_float = true;
            pl.acc_sum = find_or_add_acc(accs, ACC_FSUM, x, t_of(T_FLOAT64));
          } else throw Unsupported(pl.fn + " over " + arg->type.to_string());
          if (pl.fn == "AVG" || pl.arg_nullable) pl.acc_cnt = count_of(arg);
        } else if (pl.fn == "MIN" || pl.fn == "MAX") {
          const bool mn = pl.fn == "MIN";
          if (arg->type.is_float()) { pl.is_float = true; pl.acc_mm = find_or_add_acc(accs, mn ? ACC_FMIN : ACC_FMAX, arg, arg->type); }
          else if (arg->type.is_int() || arg->type.is_decimal() || arg->type.is_temporal()) pl.acc_mm = find_or_add_acc(accs, mn ? ACC_MIN : ACC_MAX, arg, arg->type);
          else throw Unsupported(pl.fn + " over " + arg->type.to_string());
          if (pl.arg_nullable) pl.acc_cnt = count_of(arg);
        } else if (var_family(pl.fn)) {
          const DType f64 = t_of(T_FLOAT64);
          if (!(arg->type.is_float() || arg->type.is_int() || arg->type.is_decimal())) throw Unsupported(pl.fn + " over " + arg->type.to_string());
          NodeP x = ec.cast(arg, f64), y;
          pl.is_float = true;
          if (var_family(pl.fn) == 2) {
            if (!a.has("expr2")) throw std::runtime_error(pl.fn + " needs two arguments (expr, expr2)");
            y = ec.cast(ec.from_json(a.at("expr2")), f64);
            // rows where either argument is NULL are skipped for every sum
            NodeP both = ec.binary("AND", ec.is_null(x, true), ec.is_null(y, true));
            if (x->nullable || y->nullable) { NodeP x2 = ec.select(both, x, ec.lit_null(f64)); NodeP y2 = ec.select(both, y, ec.lit_null(f64)); x = x2; y = y2; }
          }
          pl.acc_cnt = count_of(x);
          pl.acc_sx = find_or_add_acc(accs, ACC_FSUM, x, f64);
          if (var_family(pl.fn) == 1 || var_is_corr(pl.fn)) pl.acc_sxx = find_or_add_acc(accs, ACC_FSUM, ec.binary("*", x, x), f64);
          if (y) {
            pl.acc_sy = find_or_add_acc(accs, ACC_FSUM, y, f64);
            pl.acc_sxy = find_or_add_acc(accs, ACC_FSUM, ec.binary("*", x, y), f64);
            if (var_is_corr(pl.fn)) pl.acc_syy = find_or_add_acc(accs, ACC_FSUM, ec.binary("*", y, y), f64);
          }
        } else throw Unsupported("aggregate function " + pl.fn);
      }
    } else {
      // merge of partial states: columns arrive positionally after the group columns
      auto state = [&](void) { if (state_col >= op->in_schema.fields.size()) throw std::runtime_error("Final aggregate: input has too few state columns"); return ec.column((int)state_col++); };
      if (pl.fn == "COUNT") { NodeP c = state(); pl.acc_cnt = find_or_add_acc(accs, ACC_SUM, ec.cast(c, t_of(T_INT64)), t_of(T_INT64)); pl.arg_type = t_of(T_INT64); }
      else if (pl.fn == "SUM") {
        NodeP s = state(); pl.arg_type = s->type; pl.arg_nullable = s->nullable || key_nodes.empty();
        if (s->type.is_float()) { pl.is_float = true; pl.acc_sum = find_or_add_acc(accs, ACC_FSUM, ec.coalesce0(s), s->type); }
        else pl.acc_sum = find_or_add_acc(accs, ACC_SUM, s, s->type);
        if (pl.arg_nullable) pl.acc_cnt = s->nullable ? find_or_add_acc(accs, ACC_COUNT, s, t_of(T_INT64)) : find_or_add_acc(accs, ACC_COUNT_STAR, nullptr, t_of(T_INT64));
      } else if (pl.fn == "AVG") {
        NodeP c = state(); NodeP s = state(); pl.arg_type = s->type;
        pl.acc_cnt = find_or_add_acc(accs, ACC_SUM, ec.cast(c, t_of(T_INT64)), t_of(T_INT64));
        if (s->type.is_float()) { pl.is_float = true; pl.acc_sum = find_or_add_acc(accs, ACC_FSUM, ec.coalesce0(s), s->type); }
        else pl.acc_sum = find_or_add_acc(accs, ACC_SUM, s, s->type);
      } else if (pl.fn == "MIN" || pl.fn == "MAX") {
        NodeP s = state(); const bool mn = pl.fn == "MIN"; pl.arg_type = s->type; pl.arg_nullable = s->nullable || key_nodes.empty();
        if (s->type.is_float()) { pl.is_float = true; pl.acc_mm = find_or_add_acc(accs, mn ? ACC_FMIN : ACC_FMAX, s, s->type); }
        else pl.acc_mm = find_or_add_acc(accs, mn ? ACC_MIN : ACC_MAX, s, s->type);
        if (pl.arg_nullable) pl.acc_cnt = s->nullable ? find_or_add_acc(accs, ACC_COUNT, s, t_of(T_INT64)) : find_or_add_acc(accs, ACC_COUNT_STAR, nullptr, t_of(T_INT64));
      } else if (var_family(pl.fn)) {
        // reference state columns: VAR/STDDEV [count, mean, m2]; COVAR [count, mean1, mean2, algo_const];
        // CORR [count, mean1, m2_1, mean2, m2_2, algo_const] [UPSTREAM-KNOWLEDGE].  Back to power sums, then add.
        const DType f64 = t_of(T_FLOAT64); pl.is_float = true;
        NodeP c = state(); NodeP cf = ec.cast(c, f64);
        auto fs = [&](NodeP e) { return find_or_add_acc(accs, ACC_FSUM, ec.coalesce0(e), f64); };
        auto mul = [&](NodeP l, NodeP r) { return ec.binary("*", l, r); };
        auto add = [&](NodeP l, NodeP r) { return ec.binary("+", l, r); };
        pl.acc_cnt = find_or_add_acc(accs, ACC_SUM, ec.cast(c, t_of(T_INT64)), t_of(T_INT64));
        if (var_family(pl.fn) == 1) {
          NodeP mean = ec.cast(state(), f64), m2 = ec.cast(state(), f64);
          pl.acc_sx = fs(mul(cf, mean)); pl.acc_sxx = fs(add(m2, mul(cf, mul(mean, mean))));
        } else if (!var_is_corr(pl.fn)) {
          NodeP m1 = ec.cast(state(), f64), m2 = ec.cast(state(), f64), al = ec.cast(state(), f64);
          pl.acc_sx = fs(mul(cf, m1)); pl.acc_sy = fs(mul(cf, m2)); pl.acc_sxy = fs(add(al, mul(cf, mul(m1, m2))));
        } else {
          NodeP m1 = ec.cast(state(), f64), v1 = ec.cast(state(), f64), m2 = ec.cast(state(), f64), v2 = ec.cast(state(), f64), al = ec.cast(state(), f64);
          pl.acc_sx = fs(mul(cf, m1)); pl.acc_sxx = fs(add(v1, mul(cf, mul(m1, m1))));
          pl.acc_sy = fs(mul(cf, m2)); pl.acc_syy = fs(add(v2, mul(cf, mul(m2, m2))));
          pl.acc_sxy = fs(add(al, mul(cf, mul(m1, m2))));
        }
      } else throw Unsupported("aggregate function " + pl.fn);
    }
    plans.push_back(pl);
  }
  if (accs.empty()) find_or_add_acc(accs, ACC_COUNT_STAR, nullptr, t_of(T_INT64));   // GROUP BY without aggregates still needs a cell
  // scan program outputs: keys then accumulator arguments
  std::vector<int> key_slots, acc_slots(accs.size(), -1);
  for (auto& k : key_nodes) key_slots.push_back(ec.add_output(k));
  for (size_t i = 0; i < accs.size(); ++i) if (accs[i].arg) acc_slots[i] = ec.add_output(accs[i].arg);
  op->prog = ec.finish();
  upload_code(op->prog, op->code_dev);
  op->agg = AggSpec{};
  op->agg.n_keys = (int)key_nodes.size(); op->agg.n_accs = (int)accs.size();
  std::vector<int> kregs; bool any_null_key = false;
  for (size_t k = 0; k < key_nodes.size(); ++k) {
    op->agg.key_reg[k] = op->prog.out_reg[key_slots[k]]; kregs.push_back(op->agg.key_reg[k]);
    op->key_types.push_back(key_nodes[k]->type); any_null_key = any_null_key || key_nodes[k]->nullable;
    if (key_nodes[k]->type.id == T_BOOL || key_nodes[k]->type.id == T_NULL) throw Unsupported("group-by key of type " + key_nodes[k]->type.to_string());
  }
  for (size_t i = 0; i < accs.size(); ++i) {
    op->agg.acc_kind[i] = accs[i].kind; op->agg.acc_reg[i] = accs[i].arg ? op->prog.out_reg[acc_slots[i]] : 0;
    op->acc_types.push_back(accs[i].type);
    op->acc_bits.push_back(accs[i].arg ? op->prog.out_bits[acc_slots[i]] : 1);
  }
  op->keys = make_keyspec(kregs, op->key_types, any_null_key);

  // post program over the SoA result: [key_0.., acc_0..] as raw (lo,hi) columns
  for (size_t k = 0; k < key_nodes.size(); ++k) { Field f; f.name = key_names[k]; f.type = key_nodes[k]->type; f.nullable = key_nodes[k]->nullable; f.raw128 = 1; op->post_schema.fields.push_back(f); }
  for (size_t i = 0; i < accs.size(); ++i) { Field f; f.name = "acc" + std::to_string(i); f.type = accs[i].type; f.nullable = false; f.raw128 = 1; op->post_schema.fields.push_back(f); }
  const int nk = (int)key_nodes.size();
  // The post program runs in the same 16-register machine: when all outputs do not fit, the plan list is split and
  // each chunk becomes its own program over the same SoA columns (a few extra launches over <= n_groups rows).
  auto build_chunk = [&](ExprCompiler& pc, size_t lo, size_t hi, bool with_keys, std::vector<std::string>& out_names) {
  if (with_keys && !packed) for (int k = 0; k < nk; ++k) { pc.add_output(pc.column(k)); out_names.push_back(key_names[k]); }
  if (with_keys && packed) for (size_t k = 0; k < orig_keys.size(); ++k) {      // the group columns in their declared order, unpacked
    const NodeP& n = orig_keys[k];
    if (pk[k].slot < 0) { pc.add_output(pc.lit_null(n->type)); out_names.push_back(orig_names[k]); continue; }      // (refused operator: types only)
    if (pk[k].own) { pc.add_output(pc.column(pk[k].slot)); out_names.push_back(orig_names[k]); continue; }
    const DType wide = dec_t(38, 0);
    NodeP f = pc.column(pk[k].slot);      // non-negative, < 2^126: truncating division is the shift
    if (pk[k].shift) f = pc.raw(OP_DIV, wide, false, 127, {f, pc.lit_int(wide, (i128)1 << pk[k].shift)});
    f = pc.raw(OP_MOD, wide, false, pk[k].width, {f, pc.lit_int(wide, (i128)1 << pk[k].width)});
    NodeP v = pc.raw(OP_SUB, n->type, false, n->bits, {f, pc.lit_int(wide, (i128)1 << n->bits)});
    if (n->nullable) v = pc.select(pc.raw(OP_GE, t_of(T_BOOL), false, 2, {f, pc.lit_int(wide, (i128)1 << (n->bits + 1))}), pc.lit_null(n->type), v);
    pc.add_output(v); out_names.push_back(orig_names[k]);
  }
  for (size_t pi = lo; pi < hi; ++pi) {
    const AggPlan& pl = plans[pi];
    auto acc = [&](int i) { return pc.column(nk + i); };
    auto guard = [&](NodeP v) { return pl.acc_cnt >= 0 && (pl.arg_nullable) ? pc.nullif0(v, acc(pl.acc_cnt)) : v; };
    if (pl.fn == "COUNT") { pc.add_output(acc(pl.acc_cnt)); out_names.push_back(emit_state ? pl.name + "[count]" : pl.name); }
    else if (pl.fn == "SUM") { pc.add_output(guard(acc(pl.acc_sum))); out_names.push_back(emit_state ? pl.name + "[sum]" : pl.name); }
    else if (pl.fn == "MIN" || pl.fn == "MAX") { pc.add_output(guard(acc(pl.acc_mm))); out_names.push_back(emit_state ? pl.name + (pl.fn == "MIN" ? "[min]" : "[max]") : pl.name); }
    else if (var_family(pl.fn)) {
      const DType f64 = t_of(T_FLOAT64);
      auto F = [&](int op_, NodeP l, NodeP r) { return pc.raw(op_, f64, false, 127, {l, r}); };
      NodeP n = acc(pl.acc_cnt), nf = pc.cast(n, f64), zero = pc.lit_f64(0.0);
      NodeP n_is0 = pc.binary("=", n, pc.lit_int(t_of(T_INT64), 0));
      NodeP n_le1 = pc.binary("<=", n, pc.lit_int(t_of(T_INT64), 1));
      auto when0 = [&](NodeP v) { return pc.select(n_is0, zero, v); };                    // state columns of an empty group are 0
      auto centred = [&](int sab, int sa, int sb) { return when0(F(OP_FSUB, acc(sab), F(OP_FDIV, F(OP_FMUL, acc(sa), acc(sb)), nf))); };
      auto nonneg = [&](NodeP v) { return pc.select(pc.raw(OP_FLT, t_of(T_BOOL), false, 1, {v, zero}), zero, v); };   // rounding can leave -eps
      NodeP mean_x = when0(F(OP_FDIV, acc(pl.acc_sx), nf));
      NodeP m2x = pl.acc_sxx >= 0 ? nonneg(centred(pl.acc_sxx, pl.acc_sx, pl.acc_sx)) : nullptr;
      NodeP mean_y = pl.acc_sy >= 0 ? when0(F(OP_FDIV, acc(pl.acc_sy), nf)) : nullptr;
      NodeP m2y = pl.acc_syy >= 0 ? nonneg(centred(pl.acc_syy, pl.acc_sy, pl.acc_sy)) : nullptr;
      NodeP cxy = pl.acc_sxy >= 0 ? centred(pl.acc_sxy, pl.acc_sx, pl.acc_sy) : nullptr;
      auto out = [&](NodeP v, const std::string& nm) { pc.add_output(v); out_names.push_back(nm); };
      if (emit_state) {
        out(pc.cast(n, t_of(T_UINT64)), pl.name + "[count]");
        if (var_family(pl.fn) == 1) { out(mean_x, pl.name + "[mean]"); out(m2x, pl.name + "[m2]"); }
        else if (!var_is_corr(pl.fn)) { out(mean_x, pl.name + "[mean1]"); out(mean_y, pl.name + "[mean2]"); out(cxy, pl.name + "[algoConst]"); }
        else { out(mean_x, pl.name + "[mean1]"); out(m2x, pl.name + "[m2_1]"); out(mean_y, pl.name + "[mean2]"); out(m2y, pl.name + "[m2_2]"); out(cxy, pl.name + "[algoConst]"); }
      } else {
        const bool pop = var_is_pop(pl.fn);
        NodeP null64 = pc.lit_null(f64);
        NodeP denom = pop ? nf : F(OP_FSUB, nf, pc.lit_f64(1.0));
        NodeP undefined = pop ? n_is0 : n_le1;                                            // sample statistics need n >= 2, population n >= 1
        if (var_family(pl.fn) == 1) {
          NodeP v = F(OP_FDIV, m2x, denom);
          if (var_is_stddev(pl.fn)) v = pc.raw(OP_FSQRT, f64, false, 127, {v});
          out(pc.select(undefined, null64, v), pl.name);
        } else if (!var_is_corr(pl.fn)) {
          out(pc.select(undefined, null64, F(OP_FDIV, cxy, denom)), pl.name);
        } else {
          // corr = cov_pop / (sd_pop_x * sd_pop_y); 0 when either deviation is 0; NULL over no rows
          NodeP sx = pc.raw(OP_FSQRT, f64, false, 127, {F(OP_FDIV, m2x, nf)}), sy = pc.raw(OP_FSQRT, f64, false, 127, {F(OP_FDIV, m2y, nf)});
          NodeP flat = pc.binary("OR", pc.raw(OP_FEQ, t_of(T_BOOL), false, 1, {sx, zero}), pc.raw(OP_FEQ, t_of(T_BOOL), false, 1, {sy, zero}));
          NodeP v = F(OP_FDIV, F(OP_FDIV, F(OP_FDIV, cxy, nf), sx), sy);
          out(pc.select(n_is0, null64, pc.select(flat, zero, v)), pl.name);
        }
      }
    }
    else if (pl.fn == "AVG") {
      if (emit_state) {
        pc.add_output(pc.cast(acc(pl.acc_cnt), t_of(T_UINT64))); out_names.push_back(pl.name + "[count]");
        pc.add_output(guard(acc(pl.acc_sum))); out_names.push_back(pl.name + "[sum]");
      } else if (pl.is_float) {
        NodeP cnt = pc.cast(acc(pl.acc_cnt), t_of(T_FLOAT64));
        pc.add_output(pc.nullif0(pc.raw(OP_FDIV, t_of(T_FLOAT64), true, 127, {acc(pl.acc_sum), cnt}), acc(pl.acc_cnt))); out_names.push_back(pl.name);
      } else {
        // Decimal AVG: sum * 10^(s_avg - s_sum) / count, truncating; count == 0 -> NULL (OP_DIV by zero)
        const DType st = op->acc_types[pl.acc_sum];
        // the sum state is Decimal(min(38,p+10), s); the argument was Decimal(p, s)
        const int arg_p = is_final ? std::max(1, st.p - 10) : pl.arg_type.p;
        const DType rt = dec_t(arg_p + 4, st.s + 4);
        NodeP scaled = pc.raw(OP_MUL, rt, false, 127, {acc(pl.acc_sum), pc.lit_int(dec_t(38, 0), pow10_i128(rt.s - st.s))});
        pc.add_output(pc.raw(OP_DIV, rt, true, 127, {scaled, acc(pl.acc_cnt)})); out_names.push_back(pl.name);
      }
    }
  }
  };
  std::function<void(size_t, size_t, bool)> emit = [&](size_t lo, size_t hi, bool with_keys) {
    ExprCompiler pc(op->post_schema);
    std::vector<std::string> names; CompiledProgram cp;
    try { build_chunk(pc, lo, hi, with_keys, names); cp = pc.finish(); }
    catch (const Unsupported&) { throw; }
    catch (const std::runtime_error&) {
      if (hi - lo + (with_keys ? 1 : 0) <= 1) throw;
      if (with_keys && hi > lo) { emit(lo, lo, true); emit(lo, hi, false); }
      else { const size_t mid = lo + (hi - lo) / 2; emit(lo, mid, with_keys); emit(mid, hi, false); }
      return;
    }
    if (cp.out_reg.empty()) return;
    op->posts.emplace_back(); gpuq_op::PostChunk& pcn = op->posts.back();
    pcn.prog = cp; upload_code(pcn.prog, pcn.code); pcn.first_out = (int)op->out_fields.size();
    for (size_t i = 0; i < cp.out_type.size(); ++i) op->out_fields.push_back(make_field(names[i], cp.out_type[i], cp.out_nullable[i]));
  };
  emit(0, plans.size(), true);
}

OutSpec make_outspec(const CompiledProgram& cp, gpuq_column* outs, int n_outs, const std::vector<gpuq_field_info>& fields) {
  if (n_outs != (int)cp.out_reg.size()) throw std::runtime_error("expected " + std::to_string(cp.out_reg.size()) + " output columns, got " + std::to_string(n_outs));
  if (n_outs > MAX_OUTS) throw Unsupported("more than " + std::to_string(MAX_OUTS) + " output columns in one call");
  OutSpec O{}; O.n_out = n_outs;
  for (int i = 0; i < n_outs; ++i) {
    if (!outs[i].data) throw std::runtime_error("output column " + std::to_string(i) + " has no data buffer");
    O.cols[i].data = const_cast<void*>(outs[i].data);
    O.cols[i].validity = (u64*)const_cast<uint8_t*>(outs[i].validity);
    O.cols[i].reg = cp.out_reg[i];
    O.cols[i].cls = col_class_for(cp.out_type[i]);
    outs[i].type = fields[i].type; outs[i].precision = fields[i].precision; outs[i].scale = fields[i].scale; outs[i].repr = fields[i].repr;
  }
  return O;
}

// the LDS aggregate is specialised on its whole shape (keys, accumulators, group capacity)
std::string agg_tiny_spec(const gpuq_op* op, int gmax) {
  const int nk = op->agg.n_keys, na = op->agg.n_accs;
  std::string spec = "#define GPUQ_JIT_SPEC 1\nconstexpr int JIT_NKEYS = " + std::to_string(nk) + ", JIT_NACCS = " + std::to_string(na) +
                     ", JIT_GMAX = " + std::to_string(gmax) + ", JIT_NKC = " + std::to_string(nk > 0 ? nk : 1) + ";\n";
  auto arr = [](const char* name, const int32_t* v, int n_) { std::string r = std::string("constexpr int ") + name + "[" + std::to_string(n_) + "] = {";
                                                             for (int i = 0; i < n_; ++i) r += std::to_string(v[i]) + (i + 1 < n_ ? "," : ""); return r + "};\n"; };
  int32_t bits[MAX_ACCS]; for (int i = 0; i < MAX_ACCS; ++i) bits[i] = i < (int)op->acc_bits.size() ? op->acc_bits[(size_t)i] : 127;
  return spec + arr("JIT_KEY_REG", op->agg.key_reg, MAX_KEYS) + arr("JIT_ACC_KIND", op->agg.acc_kind, MAX_ACCS) + arr("JIT_ACC_REG", op->agg.acc_reg, MAX_ACCS) +
         arr("JIT_ACC_BITS", bits, MAX_ACCS);
}

void check_ctx(gpuq_ctx* c) { if (!c) throw std::runtime_error("ctx is NULL"); HIPCHECK(hipSetDevice(c->device)); }

}  // namespace

// =====================================================================================================
extern "C" {

int gpuq_abi_version(void) { return GPUQ_ABI_VERSION; }

gpuq_ctx* gpuq_ctx_create(int device_ordinal, const char* json_opts) {
  gpuq_ctx* c = nullptr;
  int rc = guarded(nullptr, [&]() {
    int dev = device_ordinal;
    if (json_opts && *json_opts) { Json o = JsonParser(json_opts).parse(); dev = (int)o.get_i64("device", dev); }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) throw HipError("no usable HIP device (gpuq has no CPU fallback): " + std::string(hipGetErrorString(e)));
    if (dev < 0 || dev >= count) throw std::runtime_error("device ordinal out of range");
    HIPCHECK(hipSetDevice(dev));
    hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, dev));
    c = new gpuq_ctx();
    c->device = dev; c->cus = prop.multiProcessorCount; c->hbm = prop.totalGlobalMem; c->name = prop.name; c->arch = prop.gcnArchName;
    set_num_cus(c->cus);
    std::string jm = std::getenv("GPUQ_JIT") ? std::getenv("GPUQ_JIT") : "";
    if (json_opts && *json_opts) { Json o = JsonParser(json_opts).parse(); jm = o.get_str("jit", jm); c->jit_min_rows = o.get_i64("jit_min_rows", c->jit_min_rows); }
    if (jm == "off" || jm == "0") c->jit_mode = 0; else if (jm == "force" || jm == "2") c->jit_mode = 2; else c->jit_mode = 1;
    c->jit_wait = jm == "wait";
    if (!jit_available() && c->jit_mode == 1) c->jit_mode = 0;
    if (const char* e = std::getenv("GPUQ_JOIN_DENSE")) c->join_dense = std::atoi(e) != 0;
  });
  if (rc != GPUQ_OK) { delete c; return nullptr; }
  return c;
}
void gpuq_ctx_free(gpuq_ctx* ctx) { delete ctx; }
const char* gpuq_last_error(gpuq_ctx*) { return g_last_error.c_str(); }
int gpuq_ctx_device_info(gpuq_ctx* ctx, char* buf, size_t cap) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    std::snprintf(buf, cap, "{\"name\":\"%s\",\"arch\":\"%s\",\"cus\":%d,\"hbm_bytes\":%zu}", ctx->name.c_str(), ctx->arch.c_str(), ctx->cus, ctx->hbm);
  });
}

// ---------------------------------------------------------------- op create
static void compile_op(gpuq_op* op, const Json& d) {
    const std::string kind = d.at("op").str();
    for (char c : d.get_str("label", "")) if (std::isalnum((unsigned char)c) || c == '_') op->label += c;
    op->in_schema = schema_from_json(d.at("input"));
    if (kind == "filter") {
      op->kind = K_FILTER;
      ExprCompiler ec(op->in_schema); ec.add_predicate(ec.from_json(d.at("predicate")));
      op->prog = ec.finish(); upload_code(op->prog, op->code_dev);
    } else if (kind == "project") {
      op->kind = K_PROJECT;
      ExprCompiler ec(op->in_schema);
      std::vector<std::string> names;
      for (const Json& e : d.at("exprs").a) { ec.add_output(ec.from_json(e.at("expr"))); names.push_back(e.get_str("name", "col" + std::to_string(names.size()))); }
      op->prog = ec.finish(); upload_code(op->prog, op->code_dev);
      for (size_t i = 0; i < names.size(); ++i) op->out_fields.push_back(make_field(names[i], op->prog.out_type[i], op->prog.out_nullable[i]));
    } else if (kind == "aggregate") {
      op->kind = K_AGG; compile_aggregate(op, d);
    } else if (kind == "join_build" || kind == "join_probe") {
      op->kind = kind == "join_build" ? K_JOIN_BUILD : K_JOIN_PROBE;
      ExprCompiler ec(op->in_schema);
      if (d.has("predicate")) ec.add_predicate(ec.from_json(d.at("predicate")));
      std::vector<NodeP> ks;
      for (const Json& e : d.at("on").a) { NodeP n = ec.from_json(e); ks.push_back(n); ec.add_output(n); }
      if (ks.empty()) throw std::runtime_error("join needs at least one key");
      // "semi_on" (join_build only): keys these rows are first looked up with in ANOTHER join's table (gpuq_join_build_run_semi)
      std::vector<NodeP> semi;
      if (d.has("semi_on")) {
        if (op->kind != K_JOIN_BUILD) throw std::runtime_error("semi_on belongs to a join_build operator");
        for (const Json& e : d.at("semi_on").a) { NodeP n = ec.from_json(e); semi.push_back(n); ec.add_output(n); }
      }
      op->prog = ec.finish(); upload_code(op->prog, op->code_dev);
      op->null_eq = d.get_bool("null_equals_null", false) ? 1 : 0;
      op->build_side_rows = d.get_bool("build_side_rows", true);
      std::vector<int> regs; for (size_t k = 0; k < ks.size(); ++k) { regs.push_back(op->prog.out_reg[k]); op->key_types.push_back(ks[k]->type); }
      op->keys = make_keyspec(regs, op->key_types, op->null_eq != 0);
      op->has_semi = !semi.empty();
      if (op->has_semi) {
        std::vector<int> sregs; std::vector<DType> stypes;
        for (size_t k = 0; k < semi.size(); ++k) { sregs.push_back(op->prog.out_reg[ks.size() + k]); stypes.push_back(semi[k]->type); }
        op->semi_keys = make_keyspec(sregs, stypes, op->null_eq != 0);
      }
      if (op->kind == K_JOIN_PROBE) {
        static const std::map<std::string, int> jt = {{"Inner", JT_INNER}, {"Left", JT_LEFT}, {"Right", JT_RIGHT}, {"Full", JT_FULL},
            {"LeftSemi", JT_LEFT_SEMI}, {"LeftAnti", JT_LEFT_ANTI}, {"RightSemi", JT_RIGHT_SEMI}, {"RightAnti", JT_RIGHT_ANTI}};
        auto it = jt.find(d.get_str("join_type", "Inner"));
        if (it == jt.end()) throw std::runtime_error("unknown join_type");
        op->join_type = it->second;
      }
    } else if (kind == "sort") {
      op->kind = K_SORT;
      ExprCompiler ec(op->in_schema);
      const auto& es = d.at("expr").a;
      if (es.empty() || es.size() > (size_t)MAX_SORT_KEYS) throw Unsupported("sort needs 1.." + std::to_string(MAX_SORT_KEYS) + " keys");
      std::vector<NodeP> ks;
      for (const Json& e : es) { NodeP n = ec.from_json(e.at("expr")); ks.push_back(n); ec.add_output(n); }
      op->prog = ec.finish(); upload_code(op->prog, op->code_dev);
      op->sort.n_keys = (int)ks.size();
      for (size_t k = 0; k < ks.size(); ++k) {
        const bool asc = es[k].get_bool("asc", true);
        op->sort.reg[k] = op->prog.out_reg[k]; op->sort.desc[k] = asc ? 0 : 1;
        op->sort.nulls_first[k] = es[k].get_bool("nulls_first", !asc) ? 1 : 0;
        op->sort.kind[k] = ks[k]->type.is_float() ? 1 : (ks[k]->type.id == T_UTF8 ? 2 : 0);
      }
      op->sort_key0 = ks[0]->type;
      op->fetch = d.get_i64("fetch", -1);
    } else if (kind == "partition") {
      op->kind = K_PARTITION;
      ExprCompiler ec(op->in_schema);
      std::vector<NodeP> ks;
      for (const Json& e : d.at("hash_expr").a) { NodeP n = ec.from_json(e); ks.push_back(n); ec.add_output(n); }
      if (ks.empty()) throw std::runtime_error("partition needs at least one hash expression");
      op->prog = ec.finish(); upload_code(op->prog, op->code_dev);
      std::vector<int> regs; for (size_t k = 0; k < ks.size(); ++k) { regs.push_back(op->prog.out_reg[k]); op->key_types.push_back(ks[k]->type); }
      op->keys = make_keyspec(regs, op->key_types, false);
      const i64 np = d.at("partition_count").i64();
      if (np < 1 || np > 65536) throw std::runtime_error("partition_count out of range (1..65536)");
      op->nparts = (uint32_t)np;
    } else throw std::runtime_error("unknown op '" + kind + "'");
}

int gpuq_op_create(gpuq_ctx* ctx, const char* json, gpuq_op** out) {
  gpuq_op* op = nullptr;
  int rc = guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!json || !out) throw std::runtime_error("json/out is NULL");
    Json d = JsonParser(json).parse();
    op = new gpuq_op(); op->ctx = ctx;
    op->flags_dev.ensure(256); HIPCHECK(hipMemset(op->flags_dev.p, 0, 256));
    g_upload = true;
    compile_op(op, d);
    *out = op;
  });
  if (rc != GPUQ_OK) { delete op; if (out) *out = nullptr; }
  return rc;
}

static const char* op_name(int op) {
  static const char* n[] = {"NOP","IMM","MOV","ADD","SUB","MUL","MULW","NEG","DIV","MOD","EQ","NE","LT","LE","GT","GE","FADD","FSUB","FMUL","FDIV","FNEG","FSQRT",
    "FEQ","FNE","FLT","FLE","FGT","FGE","I2F","F2I","AND","OR","NOT","ISNULL","ISNOTNULL","SELECT","SHL","BOR","NULLIF0","COALESCE0"};
  return (op >= 0 && op < (int)(sizeof(n) / sizeof(n[0]))) ? n[op] : "?";
}
static std::string describe_program(const CompiledProgram& p, const Schema& sc) {
  std::string r = "{\"columns\":[";
  for (size_t i = 0; i < p.col_field.size(); ++i) { if (i) r += ","; r += "\"" + sc.fields[p.col_field[i]].name + "\""; }
  r += "],\"pred_reg\":" + std::to_string(p.pred_reg) + ",\"insns\":[";
  for (int i = 0; i < p.n_insns; ++i) {
    const DevInsn& in = p.code.insns[i];
    if (i) r += ",";
    r += std::string("\"") + op_name(in.op) + " r" + std::to_string(in.dst) + " r" + std::to_string(in.a) + " r" + std::to_string(in.b) + " #" + std::to_string(in.imm) + "\"";
  }
  r += "],\"out_reg\":[";
  for (size_t i = 0; i < p.out_reg.size(); ++i) { if (i) r += ","; r += std::to_string(p.out_reg[i]); }
  r += "],\"out_type\":[";
  for (size_t i = 0; i < p.out_type.size(); ++i) { if (i) r += ","; r += "\"" + p.out_type[i].to_string() + "\""; }
  return r + "]}";
}

// Host-only: compile a descriptor without touching a device and describe the result (CPU tests,
// plan validation in a scheduler-side process).  buf receives JSON.
int gpuq_compile_check(const char* json, char* buf, size_t cap) {
  gpuq_op* op = nullptr;
  int rc = guarded(nullptr, [&]() {
    if (!json) throw std::runtime_error("json is NULL");
    Json d = JsonParser(json).parse();
    op = new gpuq_op();
    g_upload = false;
    compile_op(op, d);
    g_upload = true;
    std::string r = "{\"program\":" + describe_program(op->prog, op->in_schema);
    if (op->kind == K_AGG) {
      r += ",\"post\":" + describe_program(op->posts.front().prog, op->post_schema) + ",\"post_programs\":" + std::to_string(op->posts.size()) + ",\"acc_kinds\":[";
      for (int a = 0; a < op->agg.n_accs; ++a) { if (a) r += ","; r += std::to_string(op->agg.acc_kind[a]); }
      r += "]";
    }
    r += ",\"outputs\":[";
    for (size_t i = 0; i < op->out_fields.size(); ++i) {
      const gpuq_field_info& f = op->out_fields[i];
      if (i) r += ",";
      DType t; t.id = f.type; t.p = f.precision; t.s = f.scale;
      r += std::string("{\"name\":\"") + f.name + "\",\"type\":\"" + t.to_string() + "\",\"nullable\":" + (f.nullable ? "true" : "false") + "}";
    }
    r += "]}";
    if (buf && cap) { std::snprintf(buf, cap, "%s", r.c_str()); if (r.size() + 1 > cap) throw Capacity("describe buffer too small"); }
  });
  g_upload = true;
  delete op;
  return rc;
}
// Host-only: the hiprtc input for a descriptor + sink kernel, without a device (build-time check of the JIT path).
int gpuq_compile_jit_source(const char* json, int kernel_id, char* buf, size_t cap) {
  gpuq_op* op = nullptr;
  int rc = guarded(nullptr, [&]() {
    if (!json) throw std::runtime_error("json is NULL");
    Json d = JsonParser(json).parse();
    op = new gpuq_op();
    g_upload = false;
    compile_op(op, d);
    g_upload = true;
    std::string eval = op->prog.jit_src;
    if (kernel_id == 3 && op->kind == K_AGG) eval += agg_tiny_spec(op, (int)d.get_i64("jit_gmax", 4));   // what the run-time path appends
    const std::string src = jit_full_source(eval, kernel_id);
    if (src.size() + 1 > cap) throw Capacity("source needs " + std::to_string(src.size() + 1) + " bytes");
    std::memcpy(buf, src.c_str(), src.size() + 1);
  });
  g_upload = true;
  delete op;
  return rc;
}
void gpuq_op_free(gpuq_op* op) { delete op; }
int gpuq_op_num_outputs(gpuq_op* op) { return op ? (int)op->out_fields.size() : 0; }
int gpuq_op_output_field(gpuq_op* op, int i, gpuq_field_info* out) {
  if (!op || !out || i < 0 || i >= (int)op->out_fields.size()) return GPUQ_ERR_INVALID;
  *out = op->out_fields[i]; return GPUQ_OK;
}
int gpuq_op_check(gpuq_op* op, void* stream) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() { check_ctx(op->ctx); const uint32_t f = read_flags(op, use_stream(stream)); if (f) { reset_flags(op, use_stream(stream)); raise_flags(f); } });
}
int gpuq_op_set_deferred(gpuq_op* op, int on) { if (!op) return GPUQ_ERR_INVALID; op->deferred = on != 0; op->defer_client = true; return GPUQ_OK; }
int gpuq_op_can_defer(gpuq_op* op) {
  if (!op) return 0;
  switch (op->kind) {
    case K_FILTER: case K_PROJECT: case K_JOIN_PROBE: return 1;      // nothing is read back inside these calls
    case K_JOIN_BUILD: return op->jb.valid ? 1 : 0;
    case K_AGG: return op->ag.valid ? 1 : 0;
    case K_SORT: return op->so.valid ? 1 : 0;
    default: return 0;
  }
}
int gpuq_ops_settle(gpuq_ctx* ctx, void* stream, gpuq_op* const* ops, int n_ops, const uint64_t* const* words, int n_words, uint64_t* words_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (n_ops < 0 || n_words < 0 || (n_ops && !ops) || (n_words && (!words || !words_out))) throw std::runtime_error("settle: bad arguments");
    hipStream_t s = use_stream(stream);
    const int total = n_ops + n_words;
    // one pinned block and one device block per calling thread (plans of several task threads settle concurrently)
    struct Blocks { u64* host = nullptr; DevBuf dev; size_t cap = 0; };
    thread_local Blocks B;
    if ((size_t)total > B.cap) {
      if (B.host) (void)hipHostFree(B.host);
      B.cap = (size_t)std::max(total, 256); B.host = nullptr;
      HIPCHECK(hipHostMalloc((void**)&B.host, B.cap * 8, hipHostMallocDefault));
    }
    u64* dev = (u64*)B.dev.ensure(B.cap * 8);
    for (int base = 0; base < total; base += 64) {
      GatherWords g{}; g.n = std::min(64, total - base);
      for (int i = 0; i < g.n; ++i) { const int k = base + i; g.src[i] = k < n_ops ? (const u64*)ops[k]->flags_dev.p : (const u64*)words[k - n_ops]; if (!g.src[i]) throw std::runtime_error("settle: NULL word"); }
      launch_gather_words(s, g, dev + base);
    }
    HIPCHECK(hipGetLastError());
    if (total > 0) HIPCHECK(hipMemcpyAsync(B.host, dev, (size_t)total * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    for (int i = 0; i < n_words; ++i) words_out[i] = B.host[n_ops + i];
    bool retry = false;
    for (int i = 0; i < n_ops; ++i) {
      const uint32_t f = (uint32_t)B.host[i];
      if (f) reset_flags(ops[i], s);
      static const bool trace = getenv("GPUQ_TRACE_DEFER") != nullptr;
      if (trace && (f & ~ops[i]->expect_flags)) fprintf(stderr, "[gpuq] settle: operator %d of %d (kind %d) raised status 0x%x (expected 0x%x)\n", i, n_ops, (int)ops[i]->kind, f, ops[i]->expect_flags);
      if (f & ~ops[i]->expect_flags) { retry = true; ops[i]->jb.valid = false; ops[i]->ag.valid = false; ops[i]->so.valid = false; }
    }
    if (retry) throw Retry("an assumption of a deferred run did not hold");
  });
}
int gpuq_op_profile(gpuq_op* op, int enable, float* kernel_ms_out, int* launches_out) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() {
    if (op->ev_pending) { float ms = 0; HIPCHECK(hipEventSynchronize(op->ev1)); HIPCHECK(hipEventElapsedTime(&ms, op->ev0, op->ev1)); op->kernel_ms += ms; op->ev_pending = false; }
    if (op->ev_total_pending) { float ms = 0; HIPCHECK(hipEventSynchronize(op->ev3)); HIPCHECK(hipEventElapsedTime(&ms, op->ev2, op->ev3)); op->total_ms += ms; op->ev_total_pending = false; }
    if (kernel_ms_out) *kernel_ms_out = op->kernel_ms;
    if (launches_out) *launches_out = op->launches;
    op->last_total_ms = op->total_ms;
    op->kernel_ms = 0; op->total_ms = 0; op->launches = 0; if (enable >= 0) op->profile = enable != 0;
  });
}
int gpuq_op_profile_total(gpuq_op* op, float* total_ms_out) {      // of the interval the last gpuq_op_profile call closed
  if (!op || !total_ms_out) return GPUQ_ERR_INVALID;
  *total_ms_out = op->last_total_ms;
  return GPUQ_OK;
}

// ---------------------------------------------------------------- filter
int gpuq_filter_run(gpuq_op* op, void* stream, const gpuq_input* in, int payload_via, uint32_t* sel_out, uint64_t* count_out) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() {
    check_ctx(op->ctx);
    if (op->kind != K_FILTER) throw std::runtime_error("not a filter operator");
    hipStream_t s = use_stream(stream);
    ProfTotal ptot(op, s);
    DevProgram P = bind_program(op->prog, op->in_schema, op->code_dev.as<DevCode>(), op->flags_dev.as<uint32_t>(), in);
    if (payload_via < 0 || payload_via > in->n_via) throw std::runtime_error("payload_via out of range");
    const i64 n = in->n_rows;
    if (n == 0) { if (count_out) HIPCHECK(hipMemsetAsync(count_out, 0, 8, s)); return; }
    const i64 nwords = (n + 63) >> 6;
    const i64 maxb = (i64)op->ctx->cus * 8;
    i64 wpb = (nwords + maxb - 1) / maxb; if (wpb < 16) wpb = 16;
    const int nblocks = (int)((nwords + wpb - 1) / wpb);
    u64* bitmap = (u64*)op->ws[0].ensure((size_t)nwords * 8);
    uint32_t* counts = (uint32_t*)op->ws[1].ensure((size_t)nblocks * 4 + 16);
    u64* total = count_out ? (u64*)count_out : (u64*)op->ws[2].ensure(8);
    { JitScope js(op, op->prog, 1, n); ProfScope ps(op, s); launch_filter_bitmap(s, P, n, bitmap, counts, nblocks, wpb); }
    launch_scan_block_counts(s, counts, nblocks, total);
    if (sel_out) launch_compact(s, bitmap, counts, nblocks, wpb, n, payload_via > 0 ? in->via[payload_via - 1] : nullptr, sel_out);
    HIPCHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------- project
int gpuq_project_run(gpuq_op* op, void* stream, const gpuq_input* in, gpuq_column* outs, int n_outs) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() {
    check_ctx(op->ctx);
    if (op->kind != K_PROJECT) throw std::runtime_error("not a project operator");
    hipStream_t s = use_stream(stream);
    ProfTotal ptot(op, s);
    DevProgram P = bind_program(op->prog, op->in_schema, op->code_dev.as<DevCode>(), op->flags_dev.as<uint32_t>(), in);
    OutSpec O = make_outspec(op->prog, outs, n_outs, op->out_fields);
    for (int i = 0; i < n_outs; ++i) outs[i].length = in->n_rows;
    { JitScope js(op, op->prog, 2, in->n_rows); ProfScope ps(op, s); launch_project(s, P, in->n_rows, O); }
    HIPCHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------- aggregate
// ndev_out != NULL: the deferred form is allowed (gpuq_aggregate_run_deferred)
// Slots per estimated group, in percent, before rounding up to a power of two.  An estimate that is the operator's own last group count plus a
// quarter (the deferred runs of a plan, the partitions of a stage) is trusted with 130: the true load ends up in (0.31, 0.62], and the
// table is half the size 200 gave -- SF100 q3's 1.2 M groups: 2 Mi slots instead of 4 Mi, aggregate 0.66 -> 0.57 ms (initialise + extract halve,
// the hash kernel keeps its time).  A hint or a sampled estimate keeps 200.  GPUQ_AGG_SLOT_PCT overrides both (tuning).
static u64 agg_slot_pct(const bool from_last_run) {
  static const long env = []() { const char* e = std::getenv("GPUQ_AGG_SLOT_PCT"); const long x = e ? std::atol(e) : 0; return (x >= 110 && x <= 800) ? x : 0; }();
  return env ? (u64)env : (from_last_run ? 130u : 200u);
}
static int aggregate_run_impl(gpuq_op* op, void* stream, const gpuq_input* in, gpuq_column* outs, int n_outs, int64_t cap, int64_t* n_groups_out, const uint64_t** ndev_out) {
  if (!op) return GPUQ_ERR_INVALID;
  if (ndev_out) *ndev_out = nullptr;
  return guarded(op->ctx, [&]() {
    check_ctx(op->ctx);
    if (op->kind != K_AGG) throw std::runtime_error("not an aggregate operator");
    if (!op->refuse.empty()) throw Unsupported(op->refuse);
    hipStream_t s = use_stream(stream);
    ProfTotal ptot(op, s);
    DevProgram P = bind_program(op->prog, op->in_schema, op->code_dev.as<DevCode>(), op->flags_dev.as<uint32_t>(), in);
    const i64 n = in->n_rows;
    const int nk = op->agg.n_keys, na = op->agg.n_accs, kstride = nk > 0 ? nk : 1;
    uint32_t ng = 0;
    AggOut raw{};
    auto alloc_raw = [&](i64 rcap) {
      raw.cap = (int32_t)rcap;
      raw.keys = (u64*)op->ws[0].ensure((size_t)rcap * kstride * 16);
      raw.key_nulls = (uint32_t*)op->ws[1].ensure((size_t)rcap * 4);
      raw.cells = (u64*)op->ws[2].ensure((size_t)rcap * na * 16);
      raw.n_groups = op->flags_dev.as<uint32_t>() + 2;     // next to the flags word: one copy reads both
      HIPCHECK(hipMemsetAsync(raw.n_groups, 0, 8, s));
    };
    auto read_ng = [&]() { uint32_t v[3] = {0, 0, 0}; read_status(op, s, v, 3); return v[2]; };
    if (n_outs != (int)op->out_fields.size()) throw std::runtime_error("expected " + std::to_string(op->out_fields.size()) + " output columns, got " + std::to_string(n_outs));
    bool posted = false;
    // AoS -> SoA, then the final projection into the caller's typed columns.  rows: host-side bound on the group count;
    // rows_dev (optional): the device word holding the actual count.
    auto run_post = [&](uint32_t rows, const uint32_t* rows_dev) {
      const size_t colbytes = (size_t)std::max<uint32_t>(rows, 1) * 16, vbytes = ((size_t)rows + 63) / 64 * 8 + 8;
      char* soa_mem = (char*)op->ws[6].ensure((size_t)(nk + na) * colbytes + (size_t)nk * vbytes);
      AggSoA soa{};
      std::vector<gpuq_column> pcols(nk + na);
      for (int k = 0; k < nk; ++k) {
        soa.key_col[k] = (ulonglong2*)(soa_mem + (size_t)k * colbytes);
        soa.key_valid[k] = (u64*)(soa_mem + (size_t)(nk + na) * colbytes + (size_t)k * vbytes);
        const DType& t = op->key_types[k];
        pcols[k] = gpuq_column{t.id, t.p, t.s, 0, soa.key_col[k], nullptr, op->post_schema.fields[k].nullable ? (const uint8_t*)soa.key_valid[k] : nullptr, (int64_t)rows};
      }
      for (int a = 0; a < na; ++a) {
        soa.acc_col[a] = (ulonglong2*)(soa_mem + (size_t)(nk + a) * colbytes);
        const DType& t = op->acc_types[a];
        pcols[nk + a] = gpuq_column{t.id, t.p, t.s, 0, soa.acc_col[a], nullptr, nullptr, (int64_t)rows};
      }
      launch_agg_emit(s, raw, nk, na, rows, soa, rows_dev);
      gpuq_input pin{}; pin.cols = pcols.data(); pin.n_cols = nk + na; pin.n_rows = rows; pin.n_via = 0;
      for (auto& pc : op->posts) {
        DevProgram PP = bind_program(pc.prog, op->post_schema, pc.code.as<DevCode>(), op->flags_dev.as<uint32_t>(), &pin);
        PP.n_dev = (const u64*)rows_dev;      // (the 8 bytes at raw.n_groups are zeroed together: the u32 count reads as a u64)
        const int no = (int)pc.prog.out_reg.size();
        std::vector<gpuq_field_info> fi(op->out_fields.begin() + pc.first_out, op->out_fields.begin() + pc.first_out + no);
        OutSpec O = make_outspec(pc.prog, outs + pc.first_out, no, fi);
        { JitScope js(op, pc.prog, 2, rows); launch_project(s, PP, rows, O); }
      }
    };
    // ---- deferred form: the strategy, table size and output capacity of the last completed synchronous run, nothing read back; a
    // table or an output that turns out too small raises the status word gpuq_ops_settle reads
    if (ndev_out && op->deferred && op->ag.valid) {
      if (op->ag.path == 1) {
        const int gmax = op->ag.gmax;
        int nb = 0; const size_t wsb = agg_tiny_workspace_bytes(gmax, nk, na, &nb);
        void* wsp = op->ws[4].ensure(wsb);
        alloc_raw(64);
        if ((i64)raw.cap <= cap) {
          const std::string spec = agg_tiny_spec(op, gmax);
          { JitScope js(op, op->prog, 3, n, spec); ProfScope ps(op, s); launch_agg_tiny(s, P, n, op->agg, gmax, wsp); }
          launch_agg_tiny_merge(s, P, n, op->agg, gmax, wsp, raw);
          run_post((uint32_t)raw.cap, raw.n_groups);
          HIPCHECK(hipGetLastError());
          for (int i = 0; i < n_outs; ++i) outs[i].length = raw.cap;
          if (n_groups_out) *n_groups_out = raw.cap;
          *ndev_out = (const uint64_t*)raw.n_groups;
          return;
        }
      } else if (op->ag.path == 2) {
        HashTable T{};
        T.key_words = op->keys.key_words; T.slot_words = 1 + T.key_words + 2 * na;
        // the table follows the group count of the last run (+ 1/4), not the size that run happened to use (its first run sizes from a sample)
        const u64 est = std::min<u64>(std::max<u64>(op->ag.est, 64), (u64)(op->ag.groups + op->ag.groups / 4 + 64));
        i64 rcap = op->ag.groups + op->ag.groups / 4 + 1024; if (rcap > cap) rcap = cap;
        if (rcap >= 1 && rcap <= 0x7FFFFFFFll) {
          T.n_slots = next_pow2(est * agg_slot_pct(true) / 100);
          T.slots = (u64*)op->ws[5].ensure((size_t)T.n_slots * T.slot_words * 8);
          launch_ht_init(s, T, &op->agg);
          if (op->ag.use_lds) {
            int n_fsum = 0; for (int a = 0; a < na; ++a) n_fsum += op->agg.acc_kind[a] == ACC_FSUM;
            u64* fstage = nullptr;
            const size_t fbytes = (size_t)T.n_slots * (size_t)n_fsum * (size_t)agg_lds_grid(n) * 8;
            if (n_fsum > 0 && fbytes <= ((size_t)256 << 20)) { fstage = (u64*)op->ws[8].ensure(fbytes); HIPCHECK(hipMemsetAsync(fstage, 0, fbytes, s)); }
            JitScope js(op, op->prog, 13, n); ProfScope ps(op, s); launch_agg_lds(s, P, n, op->keys, op->agg, T, fstage, n_fsum);
          }
          else { JitScope js(op, op->prog, 4, n); ProfScope ps(op, s); launch_agg_hash(s, P, n, op->keys, op->agg, T); }
          alloc_raw(rcap);
          launch_agg_hash_extract(s, op->keys, op->agg, T, raw, op->flags_dev.as<uint32_t>());
          run_post((uint32_t)raw.cap, raw.n_groups);
          HIPCHECK(hipGetLastError());
          for (int i = 0; i < n_outs; ++i) outs[i].length = raw.cap;
          if (n_groups_out) *n_groups_out = raw.cap;
          *ndev_out = (const uint64_t*)raw.n_groups;
          return;
        }
      }
    }
    if (in->n_rows_dev) throw std::runtime_error("aggregate: a device-side row count needs a deferred operator with a completed synchronous run (gpuq_op_can_defer)");
    op->ag.valid = false;
    int path_done = 0, gmax_done = 0; u64 est_done = 0; bool lds_done = false;
    bool done = false;
    std::string strat = op->strategy;
    if (nk == 0) strat = "tiny";
    const int exact_gmax = 0;
    // tiny input: a 2n-slot table beats the LDS kernel's fixed per-block cost (measured: 60 us for the five small hash-path
    // launches against 83 us for one cold block of the 16-slot LDS kernel on a 4-row input)
    if (strat == "auto" && n <= 16384) strat = "hash";
    // groups this operator is expected to produce: the caller's hint, else what its previous run produced (another partition
    // of the same stage, the same query again)
    i64 known_groups = op->expected_groups > 0 ? op->expected_groups : op->last_groups;
    bool many_groups = false;          // the sample says: far more groups than an LDS dictionary, count unknown
    if (strat == "auto" && nk > 0 && known_groups < 0 && n >= (1ll << 20)) {
      // a first run over a large input: estimate the cardinality from a strided sample of 2^20 rows (linear counting into 2^24
      // bits), 20-40 us against the milliseconds a wrong strategy costs
      const i64 nsample = 1ll << 20, stride = std::max<i64>(1, n / nsample);
      const u64 nbits = 1ull << 24;
      uint32_t* bm = (uint32_t*)op->ws[7].ensure(nbits / 8 + 64);
      unsigned long long* cnt = (unsigned long long*)((char*)bm + nbits / 8);
      HIPCHECK(hipMemsetAsync(bm, 0, nbits / 8 + 64, s));
      launch_key_sample(s, P, n, op->keys, stride, nsample, bm, nbits, cnt + 1);
      launch_popcount_bits(s, (const uint8_t*)bm, (i64)nbits, cnt);
      unsigned long long* pd = (unsigned long long*)op->pinned();
      unsigned long long host2[2] = {0, 0};
      unsigned long long* dst = pd ? pd : host2;
      HIPCHECK(hipMemcpyAsync(dst, cnt, 16, hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      const i64 distinct = (i64)dst[0], passed = (i64)dst[1];
      if (passed >= 4096) {
        if (distinct * 4 <= passed) known_groups = std::max<i64>(1, distinct);      // every group seen several times: the sample covers the key domain (users of the number add their own margin)
        else many_groups = true;
      }
    }
    if (strat == "auto" && nk > 0 && (many_groups || known_groups > (i64)agg_tiny_max_groups(na))) {
      // more groups than the LDS dictionary holds: do not even try it
    } else if (strat == "auto" || strat == "tiny") {
      const int fit_big = agg_tiny_max_groups(na);
      if (fit_big < 1) { if (strat == "tiny") throw Unsupported("too many accumulators for the LDS aggregate"); }
      else {
        int fit_small = 0; for (int g = 1; g <= fit_big; ++g) { int nb; (void)nb; if ((size_t)g * na * 2048 + 4096 <= 64 * 1024) fit_small = g; }
        std::vector<int> tries;
        if (nk == 0) tries = {1};
        else if (exact_gmax > 0) tries = {exact_gmax};
        else { if (fit_small > 4) { tries.push_back(4); } if (fit_small >= 2) tries.push_back(fit_small); if (fit_big > fit_small) tries.push_back(fit_big); }
        // a known group count goes straight to the smallest capacity that holds it
        if (nk > 0 && known_groups > 0) while (tries.size() > 1 && (i64)tries.front() < known_groups) tries.erase(tries.begin());
        for (int gmax : tries) {
          int nb = 0; const size_t wsb = agg_tiny_workspace_bytes(gmax, nk, na, &nb);
          void* wsp = op->ws[4].ensure(wsb);
          alloc_raw(64);
          reset_flags(op, s);
          // the LDS aggregate is specialised on its whole shape (keys, accumulators, group capacity)
          const std::string spec = agg_tiny_spec(op, gmax);
          { JitScope js(op, op->prog, 3, n, spec); ProfScope ps(op, s); launch_agg_tiny(s, P, n, op->agg, gmax, wsp); }
          launch_agg_tiny_merge(s, P, n, op->agg, gmax, wsp, raw);
          HIPCHECK(hipGetLastError());
          // The result projection is queued before the host knows whether this try held all groups (it reads the group
          // count on the device): one host round trip per aggregate instead of two.  An overflowing try is simply redone.
          const bool ahead = (i64)raw.cap <= cap;
          if (ahead) run_post((uint32_t)raw.cap, raw.n_groups);
          uint32_t fw[4] = {0, 0, 0, 0};
          read_status(op, s, fw, 4);
          const uint32_t f = fw[0];
          if (f & ~FLAG_GROUP_OVERFLOW) { reset_flags(op, s); raise_flags(f & ~FLAG_GROUP_OVERFLOW); }
          if (!(f & FLAG_GROUP_OVERFLOW)) { ng = fw[2]; done = true; posted = ahead; path_done = 1; gmax_done = gmax; break; }
        }
        if (!done && strat == "tiny") throw Capacity("more groups than the LDS aggregate holds; use strategy hash/auto");
      }
    }
    // (a count that is merely REMEMBERED says nothing about how the keys lie: the run that produced it went through the global table --
    // e.g. because its keys are clustered, which the table's wave-level run combining turns into one touch per run -- so the same
    // path is taken again; the partitioned form is for a caller's hint or a sample that saw every group several times)
    const bool remembered_hash = op->expected_groups <= 0 && op->last_path == 2;
    if (!done && (strat == "radix" || (strat == "auto" && !remembered_hash && ((n >= (1ll << 22) && op->expected_groups >= (1ll << 20)) ||
                                                            (n >= (1ll << 20) && known_groups >= 4096))))) {
      // High cardinality: partition the rows by key hash into buckets whose groups fit an LDS table, aggregate every bucket
      // inside one block (kernels_hash.hip).  Falls through to the global table when a bucket overflows (skew, or more
      // groups than the hint promised).
      const int slot_words = 1 + op->keys.key_words + 2 * na;
      uint32_t capslots = 64; while ((size_t)capslots * 2 * slot_words * 8 <= 60 * 1024) capslots *= 2;
      if ((size_t)capslots * slot_words * 8 <= 60 * 1024 && n < (1ll << 31)) {
        const u64 est = op->expected_groups > 0 ? (u64)op->expected_groups : (known_groups > 0 ? (u64)known_groups + (u64)known_groups / 4 : (u64)n);
        const u64 per_bucket = std::max<u64>(1, (u64)capslots / 2);             // mean load 0.5: buckets are Poisson-even (hash bits), sqrt(mean) of spread
        u64 nbk = next_pow2(std::max<u64>(1, (est + per_bucket - 1) / per_bucket)); if (nbk > (1ull << 24)) nbk = 1ull << 24;
        if (n >= (1ll << 20) && nbk < 2048) nbk = 2048;                          // one block per bucket: enough of them to fill the chip
        int bits = 0; while ((1ull << bits) < nbk) ++bits;
        // 8-byte (bucket << 32 | row) records, bucketed by single-read radix passes (kernels_sort.hip); the last pass also writes the
        // row ids the bucket kernel walks
        DevBuf b_bid, b_bid2, b_ids, b_hist, b_look, b_bounds;
        u64* bid = (u64*)b_bid.ensure((size_t)n * 8); u64* bid2 = (u64*)b_bid2.ensure((size_t)n * 8);
        uint32_t* ids = (uint32_t*)b_ids.ensure((size_t)n * 4 + 16);
        u64* ghist = (u64*)b_hist.ensure((size_t)sort_max_passes() * 256 * 8);
        const size_t lwb = onesweep_ws_bytes(n);
        void* lws = b_look.ensure(lwb);
        uint32_t* bounds = (uint32_t*)b_bounds.ensure((size_t)(nbk + 2) * 4);
        reset_flags(op, s);
        { JitScope js(op, op->prog, 11, n); launch_agg_bucket_id(s, P, n, op->keys, nbk - 1, bid, nullptr); }
        const int npass = std::max(1, (bits + 7) / 8);
        launch_radix_ghist(s, bid, n, 32, npass, ghist);
        for (int p = 0; p < npass; ++p) {
          launch_onesweep_pass(s, bid, nullptr, n, 32 + 8 * p, ghist + (size_t)p * 256, lws, lwb, bid2, ids, p + 1 == npass ? 2 : 0);
          std::swap(bid, bid2);
        }
        launch_bucket_bounds(s, bid, n, nbk, bounds, 32);
        alloc_raw((i64)std::min<u64>((u64)std::max<i64>(n, 1), std::max<u64>(est + est / 4, 1ull << 20)));
        { JitScope js(op, op->prog, 12, n); ProfScope ps(op, s); launch_agg_bucket(s, P, op->keys, op->agg, ids, bounds, (uint32_t)nbk, capslots, slot_words, raw); }
        HIPCHECK(hipGetLastError());
        uint32_t fw[4] = {0, 0, 0, 0};
        read_status(op, s, fw, 4);
        if (fw[0] & ~(FLAG_TABLE_FULL | FLAG_GROUP_OVERFLOW)) { reset_flags(op, s); raise_flags(fw[0] & ~(FLAG_TABLE_FULL | FLAG_GROUP_OVERFLOW)); }
        if (!(fw[0] & (FLAG_TABLE_FULL | FLAG_GROUP_OVERFLOW))) { ng = fw[2]; done = true; path_done = 3; }
        else if (strat == "radix" && op->expected_groups > 0 && est < (u64)n) throw Capacity("radix aggregate: a bucket overflowed; raise expected_groups or use strategy hash/auto");
        else reset_flags(op, s);
      } else if (strat == "radix") throw Unsupported("radix aggregate: the group state does not fit an LDS table");
    }
    if (!done) {
      // global hash table; grow on FLAG_TABLE_FULL
      // the group count this operator produced last time (another partition of the same stage, the same query again) stands in
      // for a missing expected_groups: it sizes the table and decides on block-local pre-aggregation
      // (a quarter of head-room over the last run's count: the table is twice that, rounded up to a power of two -- SF100 q3's 1.2 M groups sat
      // in 8 Mi slots of 56 bytes when the count was doubled first: 0.09 ms to initialise and 0.26 ms to extract from, now half of both)
      const i64 known = known_groups >= 0 ? (op->expected_groups > 0 ? known_groups : std::max<i64>(known_groups + known_groups / 4, 64)) : -1;
      u64 est = known > 0 ? (u64)std::min<i64>(known, std::max<i64>(n, 1)) : (u64)std::min<i64>(std::max<i64>(n, 1), 1ll << 24);
      if (est < 64) est = 64;
      HashTable T{};
      T.key_words = op->keys.key_words; T.slot_words = 1 + T.key_words + 2 * na;
      // medium cardinality (known, and a block's LDS table holds a good share of the groups): fold rows inside the block first
      const uint32_t lslots = agg_lds_slots(T);
      const bool use_lds = strat == "lds" || (strat == "auto" && lslots > 0 && n >= (1ll << 17) && known > 0 && known <= (i64)lslots * 4);
      if (strat == "lds" && !lslots) throw Unsupported("lds aggregate: the group state does not fit an LDS table");
      for (;;) {
        T.n_slots = next_pow2(est * agg_slot_pct(known_groups >= 0 && op->expected_groups <= 0 && est == (u64)known) / 100);
        T.slots = (u64*)op->ws[5].ensure((size_t)T.n_slots * T.slot_words * 8);
        launch_ht_init(s, T, &op->agg);
        reset_flags(op, s);
        if (use_lds) {
          // float partial sums are parked per (table slot, sum, block) and added up in block order afterwards
          int n_fsum = 0; for (int a = 0; a < na; ++a) n_fsum += op->agg.acc_kind[a] == ACC_FSUM;
          u64* fstage = nullptr;
          const size_t fbytes = (size_t)T.n_slots * (size_t)n_fsum * (size_t)agg_lds_grid(n) * 8;
          if (n_fsum > 0 && fbytes <= ((size_t)256 << 20)) { fstage = (u64*)op->ws[8].ensure(fbytes); HIPCHECK(hipMemsetAsync(fstage, 0, fbytes, s)); }
          JitScope js(op, op->prog, 13, n); ProfScope ps(op, s); launch_agg_lds(s, P, n, op->keys, op->agg, T, fstage, n_fsum);
        }
        else { JitScope js(op, op->prog, 4, n); ProfScope ps(op, s); launch_agg_hash(s, P, n, op->keys, op->agg, T); }
        HIPCHECK(hipGetLastError());
        const uint32_t f = (n <= (1ll << 20) && est >= (u64)n) ? 0u : read_flags(op, s);   // a 2n-slot table cannot fill up; other flags surface after extract
        if (f & FLAG_TABLE_FULL) { if (est >= (u64)std::max<i64>(n, 1024)) throw std::runtime_error("hash aggregate: table full at maximum size"); est = std::min<u64>(est * 4, (u64)std::max<i64>(n, 1024)); continue; }
        if (f) { reset_flags(op, s); raise_flags(f); }
        break;
      }
      path_done = 2; est_done = est; lds_done = use_lds;
      if (n <= (1ll << 20)) {
        // small input: groups <= rows, so size the raw result by n and extract once (one sync instead of three)
        alloc_raw(std::max<i64>(n, 1));
        launch_agg_hash_extract(s, op->keys, op->agg, T, raw, op->flags_dev.as<uint32_t>());
        HIPCHECK(hipGetLastError());
        // as on the LDS path: queue the result projection before the group count is known on the host (it reads the count on
        // the device) when the caller's columns can hold any outcome -- one host round trip instead of two
        const bool ahead = (i64)raw.cap <= cap;
        if (ahead) run_post((uint32_t)raw.cap, raw.n_groups);
        uint32_t fw[4] = {0, 0, 0, 0};
        read_status(op, s, fw, 4);
        if (fw[0]) { reset_flags(op, s); raise_flags(fw[0]); }
        ng = fw[2];
        posted = ahead;
      } else {
        // Size the raw result from what this operator produced last time (the partitions of a stage, or the same query again,
        // have similar cardinalities) and extract once; an unknown or outgrown count costs a counting pass first.
        bool done = false;
        if (op->last_groups > 0) {
          alloc_raw(op->last_groups + op->last_groups / 4 + 1024);
          launch_agg_hash_extract(s, op->keys, op->agg, T, raw, op->flags_dev.as<uint32_t>());
          HIPCHECK(hipGetLastError());
          uint32_t fw[4] = {0, 0, 0, 0};
          read_status(op, s, fw, 4);
          ng = fw[2];
          if (fw[0] & ~FLAG_GROUP_OVERFLOW) { reset_flags(op, s); raise_flags(fw[0]); }
          done = !(fw[0] & FLAG_GROUP_OVERFLOW);
          if (!done) { reset_flags(op, s); HIPCHECK(hipMemsetAsync(raw.n_groups, 0, 4, s)); }
        } else {
          alloc_raw(1);
          raw.cap = 0;      // counting pass
          launch_agg_hash_extract(s, op->keys, op->agg, T, raw, op->flags_dev.as<uint32_t>());
          ng = read_ng();
          reset_flags(op, s);
        }
        if (!done) {
          alloc_raw(std::max<i64>(ng, 1));
          launch_agg_hash_extract(s, op->keys, op->agg, T, raw, op->flags_dev.as<uint32_t>());
          HIPCHECK(hipGetLastError());
          ng = read_ng();
        }
      }
    }
    op->last_groups = (i64)ng;
    op->last_path = path_done;
    if (path_done == 1 || path_done == 2) { op->ag.valid = true; op->ag.path = path_done; op->ag.gmax = gmax_done; op->ag.est = est_done; op->ag.use_lds = lds_done; op->ag.groups = (i64)ng; op->expect_flags = 0; }
    if (n_groups_out) *n_groups_out = ng;
    if ((i64)ng > cap) throw Capacity("aggregate produced " + std::to_string(ng) + " groups, output capacity is " + std::to_string(cap));
    for (int i = 0; i < n_outs; ++i) outs[i].length = ng;
    if (!posted) {
      run_post(ng, nullptr);
      HIPCHECK(hipGetLastError());
      HIPCHECK(hipStreamSynchronize(s));
    }
  });
}

int gpuq_aggregate_run(gpuq_op* op, void* stream, const gpuq_input* in, gpuq_column* outs, int n_outs, int64_t cap, int64_t* n_groups_out) {
  return aggregate_run_impl(op, stream, in, outs, n_outs, cap, n_groups_out, nullptr);
}
int gpuq_aggregate_run_deferred(gpuq_op* op, void* stream, const gpuq_input* in, gpuq_column* outs, int n_outs, int64_t cap, int64_t* n_bound_out, const uint64_t** n_groups_dev_out) {
  if (!n_groups_dev_out) return GPUQ_ERR_INVALID;
  return aggregate_run_impl(op, stream, in, outs, n_outs, cap, n_bound_out, n_groups_dev_out);
}

// ---------------------------------------------------------------- join
namespace {
// Do neighbouring probe rows hit neighbouring table entries?  A strided sample of 64-row words; "local" when most adjacent
// pairs of live rows are within 2^14 entries (64 KiB of table) of each other -- clustered / sorted foreign keys.
bool probe_keys_local(gpuq_op* op, hipStream_t s, const DevProgram& P, i64 n, const HashTable& T) {
  u64* c = (u64*)op->ws[7].ensure(32);
  HIPCHECK(hipMemsetAsync(c, 0, 16, s));
  const i64 nwords = (n + 63) >> 6;
  const i64 nsample = std::min<i64>(nwords, 4096);
  launch_join_locality(s, P, n, op->keys, T, nwords / nsample, nsample, c);
  HIPCHECK(hipGetLastError());
  u64 got[2] = {0, 0};
  HIPCHECK(hipMemcpyAsync(got, c, 16, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  return got[0] == 0 || got[1] * 2 >= got[0];
}
}  // namespace
static int join_build_impl(gpuq_op* op, void* stream, const gpuq_input* in, int payload_via, int64_t build_rows_bound, gpuq_join_table* semi_table,
                           uint32_t* semi_hits_out, uint64_t* rows_out, gpuq_join_table** out) {
  if (!op) return GPUQ_ERR_INVALID;
  gpuq_join_table* t = nullptr;
  int rc = guarded(op->ctx, [&]() {
    check_ctx(op->ctx);
    if (op->kind != K_JOIN_BUILD) throw std::runtime_error("not a join_build operator");
    if (!out) throw std::runtime_error("out is NULL");
    hipStream_t s = use_stream(stream);
    ProfTotal ptot(op, s);
    SemiProbe semi{}; const SemiProbe* semi_p = nullptr;
    if (semi_table) {
      if (!op->has_semi) throw std::runtime_error("join build: the descriptor has no \"semi_on\" keys");
      if (semi_table->has_dups) throw std::runtime_error("join build: the semi table holds duplicate keys (a row could survive more than once)");
      if (op->semi_keys.n_keys != semi_table->keys.n_keys || op->semi_keys.key_words != semi_table->keys.key_words) throw std::runtime_error("semi keys do not match the semi table's keys (count / width)");
      for (int k = 0; k < op->semi_keys.n_keys; ++k) if (op->semi_keys.key_wide[k] != semi_table->keys.key_wide[k]) throw std::runtime_error("semi key " + std::to_string(k) + " width class differs from the table's key; cast one side");
      if (payload_via != 0) throw std::runtime_error("join build over a semi table: payload_via must be 0 (the build row is the position)");
      semi.T = semi_table->T; semi.K = op->semi_keys; semi.null_eq = semi_table->null_eq; semi.on = 1; semi.hit_out = semi_hits_out; semi.rows_out = (u64*)rows_out;
      semi_p = &semi;
    } else if (op->has_semi) throw std::runtime_error("join build: this operator was compiled with \"semi_on\" keys: call gpuq_join_build_run_semi");
    DevProgram P = bind_program(op->prog, op->in_schema, op->code_dev.as<DevCode>(), op->flags_dev.as<uint32_t>(), in);
    if (payload_via < 0 || payload_via > in->n_via) throw std::runtime_error("payload_via out of range");
    const i64 n = in->n_rows;
    if (build_rows_bound < n && payload_via == 0) build_rows_bound = n;
    if (build_rows_bound < 0 || build_rows_bound > 0xFFFFFFFEll) throw std::runtime_error("build_rows_bound out of range");
    t = new gpuq_join_table(); t->ctx = op->ctx; t->keys = op->keys; t->null_eq = op->null_eq; t->bound = build_rows_bound;
    t->T.key_words = op->keys.key_words; t->T.slot_words = 1 + t->T.key_words;
    // One narrow key whose values span a bounded range: direct addressing (dense[key - min] = chain head).  The range of the rows
    // that will be inserted is measured first (one more pass over the build input; the build is synchronous anyway).  Direct
    // addressing wins as long as the array fits: a sparse domain (range > 4 x count) is guarded by a presence bitmap, so only
    // range / 8 bytes are initialised and a miss costs one bit (SF100 q5: 4.5 M order keys in a range of 600 M -- 133 x -- probed
    // by 600 M lineitem rows: 5.2 ms through the hash table, 1.9 ms through the array).  The range may be up to 4096 x count
    // (gpuq_ctx_set_option "join_dense" / "join_dense_ratio") and the array at most an eighth of HBM.
    // From 2^21 build rows on the range is GUESSED: min / max / count of ~2^18 sampled rows (every k-th 64-row word), the range
    // widened by 1/32 on both sides; the build kernel checks every key against the array's bounds anyway and raises a flag, in
    // which case the build is redone with the measured range and the operator stops guessing (SF100 q3: the two measuring passes
    // were 0.44 of 5.65 ms per run).  env GPUQ_JOIN_SPECULATE=0: always measure.
    static const bool spec_on = []() { const char* e = getenv("GPUQ_JOIN_SPECULATE"); return !(e && e[0] == '0'); }();
    const int dense_mode = op->ctx->join_dense; const i64 dense_ratio = op->ctx->join_dense_ratio;
    const bool narrow_key = dense_mode && n > 0 && op->keys.n_keys == 1 && !op->keys.key_wide[0] && !op->keys.null_word;
    // mode 0: the key range is measured, 1: guessed from a sample, 2: deferred -- the layout of the last completed synchronous run is
    // taken as it is (no pass over the keys, nothing read back; the build kernel's own bounds / duplicate tests raise the status word
    // that gpuq_ops_settle reads)
    // run-time specialisation of the chain-fusion build: the PK/FK shape (one narrow key on each side, a direct-addressed table to
    // build, no `present` bitmap) gets the probe-style front (kernels_hash.hip k_join_build_semi1_body), anything else the generic one
    auto semi_spec = [&](const HashTable& T2, const uint32_t* present_) -> std::string {
      if (!semi_p) return std::string();
      const KeySpec& A = op->semi_keys; const KeySpec& B = op->keys;
      const bool narrow = A.n_keys == 1 && A.key_words == 1 && !A.null_word && B.n_keys == 1 && B.key_words == 1 && !B.null_word;
      if (narrow && T2.dense && !present_ && payload_via == 0)
        return "#define GPUQ_JIT_SEMI 2\nconstexpr int JIT_KEY_REG0 = " + std::to_string(A.key_reg[0]) + ";\nconstexpr int JIT_KEY2_REG = " + std::to_string(B.key_reg[0]) + ";\n";
      return "#define GPUQ_JIT_SEMI 1\n";
    };
    auto attempt = [&](const int mode) -> bool {
    const bool guess = mode == 1, memo = mode == 2;
    bool dense = false; i64 kmin = 0; u64 krange = 0, kcount = 0;
    if (memo) { dense = op->jb.dense; kmin = op->jb.kmin; krange = op->jb.krange; }
    else if (narrow_key) {
      u64* kr = (u64*)op->ws[0].ensure(32);
      const u64 init[3] = {0x7FFFFFFFFFFFFFFFull, 0x8000000000000000ull, 0};
      const i64 wstep = guess ? std::max<i64>(1, ((n + 63) >> 6) >> 12) : 1;
      HIPCHECK(hipMemcpyAsync(kr, init, sizeof(init), hipMemcpyHostToDevice, s));
      { JitScope js(op, op->prog, 14, n); launch_join_keyrange(s, P, n, op->keys, op->null_eq, kr, wstep); }
      HIPCHECK(hipGetLastError());
      u64 got[3];
      HIPCHECK(hipMemcpyAsync(got, kr, sizeof(got), hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      u64 cnt = got[2];
      if (guess && cnt == 0) return false;      // the sample saw no row: nothing to guess from
      if (cnt > 0) {
        u64 span = got[1] - got[0];      // unsigned difference of two's complement values: exact for max >= min
        i64 lo = (i64)got[0];
        if (guess) {
          const u64 margin = span / 32 + 64;
          if (span >= (1ull << 31) || lo < (i64)0x8000000000000000ull + (i64)margin || (i64)got[1] > 0x7FFFFFFFFFFFFFFFll - (i64)margin) return false;
          lo -= (i64)margin; span += 2 * margin;
          cnt = std::min<u64>(cnt * (u64)wstep, (u64)n);
        }
        const u64 lim = std::max<u64>((u64)cnt * (u64)dense_ratio, 1ull << 16);
        if (span < (1ull << 31) && span < lim && (span + 1) * 4 <= op->ctx->hbm / 8) { dense = true; kmin = lo; krange = span + 1; kcount = cnt; }
      }
    }
    bool sparse_bits = false;
    t->T.dense = nullptr; t->T.dense_bits = nullptr; t->T.dense_min = 0; t->T.dense_range = 0;      // (a second attempt starts from a clean descriptor)
    if (dense) {
      t->T.n_slots = 0; t->T.slots = nullptr;
      t->T.dense = (uint32_t*)t->dense.ensure((size_t)krange * 4 + 16); t->T.dense_min = kmin; t->T.dense_range = krange;
      // sparse domain (fewer than one value in four is a key): presence bitmap + uninitialised row array (gpuq_kernels.h)
      sparse_bits = memo ? op->jb.sparse_bits : kcount * 4 < krange;
      if (sparse_bits) {
        const size_t bb = ((size_t)krange + 63) / 64 * 8 + 8;
        t->T.dense_bits = (uint32_t*)t->dense_bits.ensure(bb);
        HIPCHECK(hipMemsetAsync(t->T.dense_bits, 0, bb, s));
      } else HIPCHECK(hipMemsetAsync(t->T.dense, 0xFF, (size_t)krange * 4, s));
    } else {
      t->T.n_slots = next_pow2(std::max<u64>((u64)n * 3, 1024));   // load factor in (0.17, 0.33]: a miss ends after ~1.5 slot visits (2.5 at 0.5)
      t->T.slots = (u64*)t->slots.ensure((size_t)t->T.n_slots * t->T.slot_words * 8);
    }
    if (memo) {
      // unique keys remembered: no chain array (a duplicate raises the status word and the run is redone with one)
      uint32_t* next = op->jb.has_dups ? (uint32_t*)t->next.ensure((size_t)std::max<i64>(build_rows_bound, 1) * 4) : nullptr;
      const size_t bm = ((size_t)build_rows_bound + 63) / 64 * 8 + 8;
      uint32_t* present = nullptr;
      if (op->build_side_rows) { present = (uint32_t*)t->present.ensure(bm); HIPCHECK(hipMemsetAsync(present, 0, bm, s)); }
      t->has_present = op->build_side_rows;
      if (!dense) launch_ht_init(s, t->T, nullptr);
      if (rows_out) HIPCHECK(hipMemsetAsync(rows_out, 0, 8, s));      // (a build that is redone counts its survivors again)
      { JitScope js(op, op->prog, 5, n, semi_spec(t->T, present)); ProfScope ps(op, s); if (!launch_join_build(s, P, n, op->keys, t->T, next, present, payload_via, op->null_eq, semi_p)) throw Unsupported("join build over a semi table: more than 8 input columns without the specialised kernel"); }
      HIPCHECK(hipGetLastError());
      t->has_dups = op->jb.has_dups;
      return true;
    }
    uint32_t* next = (uint32_t*)t->next.ensure((size_t)std::max<i64>(build_rows_bound, 1) * 4);
    const size_t bm = ((size_t)build_rows_bound + 63) / 64 * 8 + 8;
    // `present` (which build rows passed the side's predicate) only serves gpuq_join_build_side_rows: a build whose descriptor says
    // "build_side_rows": false (Inner / Right / RightSemi / RightAnti joins) skips it -- one device-scope atomic per row less when
    // the rows come through an index vector (SF100 q3: 14.6 M of them)
    uint32_t* present = nullptr;
    if (op->build_side_rows) { present = (uint32_t*)t->present.ensure(bm); HIPCHECK(hipMemsetAsync(present, 0, bm, s)); }
    t->has_present = op->build_side_rows;
    if (!dense) launch_ht_init(s, t->T, nullptr);
    reset_flags(op, s);
    if (rows_out) HIPCHECK(hipMemsetAsync(rows_out, 0, 8, s));      // (a build that is redone counts its survivors again)
      { JitScope js(op, op->prog, 5, n, semi_spec(t->T, present)); ProfScope ps(op, s); if (!launch_join_build(s, P, n, op->keys, t->T, next, present, payload_via, op->null_eq, semi_p)) throw Unsupported("join build over a semi table: more than 8 input columns without the specialised kernel"); }
    HIPCHECK(hipGetLastError());
    uint32_t f = read_flags(op, s);
    if (guess && (f & FLAG_TABLE_FULL)) { reset_flags(op, s); return false; }      // a key outside the guessed range
    if (sparse_bits && (f & FLAG_DUP_BUILD_KEY)) {
      // duplicate keys: chains need defined heads -- rebuild over the initialised array
      t->T.dense_bits = nullptr;
      HIPCHECK(hipMemsetAsync(t->T.dense, 0xFF, (size_t)krange * 4, s));
      if (present) HIPCHECK(hipMemsetAsync(present, 0, bm, s));
      reset_flags(op, s);
      if (rows_out) HIPCHECK(hipMemsetAsync(rows_out, 0, 8, s));      // (a build that is redone counts its survivors again)
      { JitScope js(op, op->prog, 5, n, semi_spec(t->T, present)); ProfScope ps(op, s); if (!launch_join_build(s, P, n, op->keys, t->T, next, present, payload_via, op->null_eq, semi_p)) throw Unsupported("join build over a semi table: more than 8 input columns without the specialised kernel"); }
      HIPCHECK(hipGetLastError());
      f = read_flags(op, s);
    }
    if (f & ~FLAG_DUP_BUILD_KEY) { reset_flags(op, s); raise_flags(f & ~FLAG_DUP_BUILD_KEY); }
    if (f) reset_flags(op, s);
    t->has_dups = (f & FLAG_DUP_BUILD_KEY) != 0;
    // what a deferred run goes by next time (a guessed range is remembered widened, as it was used)
    op->jb.valid = true; op->jb.dense = dense; op->jb.sparse_bits = t->T.dense_bits != nullptr; op->jb.has_dups = t->has_dups; op->jb.kmin = kmin; op->jb.krange = krange;
    op->expect_flags = t->has_dups ? FLAG_DUP_BUILD_KEY : 0u;
    return true;
    };
    bool built = false;
    static const bool trace = getenv("GPUQ_TRACE_JOIN_BUILD") != nullptr;
    if (op->deferred && op->jb.valid) { built = attempt(2); if (trace) fprintf(stderr, "[gpuq] join build: %lld rows (bound), deferred with the remembered layout\n", (long long)n); }
    else if (in->n_rows_dev) throw std::runtime_error("join build: a device-side row count needs a deferred operator with a completed synchronous run (gpuq_op_can_defer)");
    if (!built && spec_on && narrow_key && n >= (1ll << 21) && !op->join_guess_failed) { built = attempt(true); if (!built) op->join_guess_failed = true; if (trace) fprintf(stderr, "[gpuq] join build: %lld rows, guessed range %s\n", (long long)n, built ? "held" : "did not hold"); }
    if (!built) { attempt(false); if (trace) fprintf(stderr, "[gpuq] join build: %lld rows, measured range, %s\n", (long long)n, t->T.dense ? (t->T.dense_bits ? "array + bitmap" : "array") : "hash table"); }
    *out = t;
  });
  if (rc != GPUQ_OK) { delete t; if (out) *out = nullptr; }
  return rc;
}
int gpuq_join_build_run(gpuq_op* op, void* stream, const gpuq_input* in, int payload_via, int64_t build_rows_bound, gpuq_join_table** out) {
  return join_build_impl(op, stream, in, payload_via, build_rows_bound, nullptr, nullptr, nullptr, out);
}
int gpuq_join_build_run_semi(gpuq_op* op, void* stream, const gpuq_input* in, int payload_via, int64_t build_rows_bound, gpuq_join_table* semi_table,
                             uint32_t* semi_hits_out, uint64_t* rows_out, gpuq_join_table** out) {
  if (!semi_table) return GPUQ_ERR_INVALID;
  return join_build_impl(op, stream, in, payload_via, build_rows_bound, semi_table, semi_hits_out, rows_out, out);
}
int gpuq_join_table_has_duplicates(const gpuq_join_table* t) { return t && t->has_dups ? 1 : 0; }
void gpuq_join_table_free(gpuq_join_table* t) { delete t; }

int gpuq_join_probe_run(gpuq_op* op, void* stream, gpuq_join_table* t, const gpuq_input* in, int payload_via, uint32_t* out_build,
                        uint32_t* out_probe, uint64_t out_cap, uint64_t* count_out) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() {
    check_ctx(op->ctx);
    if (op->kind != K_JOIN_PROBE) throw std::runtime_error("not a join_probe operator");
    if (!t || !count_out) throw std::runtime_error("table/count_out is NULL");
    if (op->keys.n_keys != t->keys.n_keys || op->keys.key_words != t->keys.key_words) throw std::runtime_error("probe keys do not match the build keys (count / width)");
    for (int k = 0; k < op->keys.n_keys; ++k) if (op->keys.key_wide[k] != t->keys.key_wide[k]) throw std::runtime_error("probe key " + std::to_string(k) + " width class differs from the build key; cast one side");
    hipStream_t s = use_stream(stream);
    ProfTotal ptot(op, s);
    DevProgram P = bind_program(op->prog, op->in_schema, op->code_dev.as<DevCode>(), op->flags_dev.as<uint32_t>(), in);
    if (payload_via < 0 || payload_via > in->n_via) throw std::runtime_error("payload_via out of range");
    const int jt = op->join_type;
    const bool need_pairs = (jt == JT_INNER || jt == JT_LEFT || jt == JT_RIGHT || jt == JT_FULL);
    if (need_pairs && (!out_build || !out_probe) && out_cap > 0) throw std::runtime_error("pair outputs are NULL");
    if ((jt == JT_RIGHT_SEMI || jt == JT_RIGHT_ANTI) && !out_probe && out_cap > 0) throw std::runtime_error("out_probe is NULL");
    uint32_t* visited = nullptr;
    if (jt == JT_LEFT || jt == JT_FULL || jt == JT_LEFT_SEMI || jt == JT_LEFT_ANTI) {
      const size_t bm = ((size_t)t->bound + 63) / 64 * 8 + 8;
      visited = (uint32_t*)t->visited.ensure(bm);
      if (!t->visited_ready) { HIPCHECK(hipMemsetAsync(visited, 0, bm, s)); t->visited_ready = true; }
    }
    if (!t->has_dups) {
      // unique build keys: atomic-free two-pass probe, output in probe order
      const i64 n = in->n_rows;
      if (n == 0) { HIPCHECK(hipMemsetAsync(count_out, 0, 8, s)); return; }
      const i64 nwords = (n + 63) >> 6;
      // segments: one wave each, small enough that the scheduler evens out waves that finish early (>= 32 segments per resident
      // wave slot would be wasted scan work; 4 per slot keeps the tail short)
      const i64 target = (i64)op->ctx->cus * 8 * 4 * 4;
      i64 wpw = (nwords + target - 1) / target; if (wpw < 16) wpw = 16;
      wpw = (wpw + 3) & ~(i64)3;
      const int nsegs = (int)((nwords + wpw - 1) / wpw);
      const bool want_build = out_build != nullptr;
      uint32_t* seg_build = want_build ? (uint32_t*)op->ws[0].ensure((size_t)nwords * 256 + 16) : nullptr;
      uint32_t* seg_probe = (uint32_t*)op->ws[1].ensure((size_t)nwords * 256 + 16);
      uint32_t* counts = (uint32_t*)op->ws[2].ensure((size_t)nsegs * 4 + 16);
      // Partitioned probe (kernels_hash.hip): pays when the table is beyond the caches AND the probe keys arrive in random
      // order; the pairs then come out in partition order, so whatever reads probe-side columns through them gathers at random
      // -- "auto" therefore asks for a big table, a big probe side and keys without locality (sampled), "force" is for measurements.
      const int rjm = op->ctx->join_radix;
      bool use_rj = rjm != 0 && t->T.dense && (jt == JT_INNER || jt == JT_RIGHT_SEMI) && n < (1ll << 31) && out_probe;
      if (use_rj && rjm == 1) use_rj = n >= (1ll << 24) && t->T.dense_range * 4 >= ((u64)64 << 20) && !probe_keys_local(op, s, P, n, t->T);
      if (use_rj) {
        RjGeomHost g; rj_geometry(n, t->T.dense_range, op->ctx->join_radix_slice_log2, &g);
        u64* rec = (u64*)op->ws[3].ensure((size_t)n * 8 + 16);
        u64* rec2 = (u64*)op->ws[4].ensure((size_t)n * 8 + 16);
        int32_t* hist = (int32_t*)op->ws[5].ensure(rj_hist_entries(g) * 4 + 16);
        const size_t swb = exclusive_scan_ws_bytes((i64)rj_hist_entries(g));
        void* sws = op->ws[6].ensure(swb);
        i64 rwpw = 0; const int rblocks = rj_probe_geometry(n, &rwpw); const int rsegs = rblocks * 4;
        if (want_build) seg_build = (uint32_t*)op->ws[0].ensure((size_t)rsegs * rwpw * 256 + 16);
        seg_probe = (uint32_t*)op->ws[1].ensure((size_t)rsegs * rwpw * 256 + 16);
        counts = (uint32_t*)op->ws[2].ensure((size_t)rsegs * 4 + 16);
        JitScope js(op, op->prog, 15, n); ProfScope ps(op, s);
        launch_rj_partition(s, P, n, op->keys, t->T, payload_via, g, rec, rec2, hist, sws, swb);
        launch_rj_probe(s, rec2, hist + (size_t)g.nparts * g.nblocks, t->T, jt, seg_build, seg_probe, counts, rblocks, rwpw);
        launch_scan_block_counts(s, counts, rsegs, (u64*)count_out);
        launch_copy_segments(s, seg_build, seg_probe, counts, rsegs, rwpw, n, (const u64*)count_out, out_build, out_probe, out_cap, op->flags_dev.as<uint32_t>());
        HIPCHECK(hipGetLastError());
        return;
      }
      std::string spec;
      if (op->keys.n_keys == 1 && op->keys.key_words == 1 && (t->T.slot_words == 2 || t->T.dense))     // one narrow key: 16-byte slots or direct addressing
        spec = "#define GPUQ_JIT_PROBE1 1\nconstexpr int JIT_KEY_REG0 = " + std::to_string(op->keys.key_reg[0]) + ";\n";
      { JitScope js(op, op->prog, 7, n, spec);
        // the profile events bracket the probe kernel alone (what rocprofv3 reports for it); the 30 us scan of the segment counts and
        // the compaction of the segments (20-50 us) follow outside the bracket
        { ProfScope ps(op, s); launch_join_probe_unique(s, P, n, op->keys, t->T, jt, op->null_eq, payload_via, seg_build, seg_probe, counts, nsegs, wpw, visited); }
        launch_scan_block_counts(s, counts, nsegs, (u64*)count_out);
        if (out_probe) launch_copy_segments(s, seg_build, seg_probe, counts, nsegs, wpw, n, (const u64*)count_out, out_build, out_probe, out_cap, op->flags_dev.as<uint32_t>()); }
      HIPCHECK(hipGetLastError());
      return;
    }
    HIPCHECK(hipMemsetAsync(count_out, 0, 8, s));
    { JitScope js(op, op->prog, 6, in->n_rows); ProfScope ps(op, s);
      launch_join_probe(s, P, in->n_rows, op->keys, t->T, t->next.as<uint32_t>(), jt, payload_via, op->null_eq, out_build, out_probe, out_cap, (u64*)count_out, visited); }
    HIPCHECK(hipGetLastError());
  });
}

int gpuq_join_build_side_rows(gpuq_join_table* t, void* stream, int matched, uint32_t* rows_out, uint64_t* count_out) {
  if (!t) return GPUQ_ERR_INVALID;
  return guarded(t->ctx, [&]() {
    check_ctx(t->ctx);
    hipStream_t s = use_stream(stream);
    const i64 n = t->bound;
    if (!t->has_present) throw std::runtime_error("this join table was built with \"build_side_rows\": false: it does not know its build side's rows");
    if (n == 0) { if (count_out) HIPCHECK(hipMemsetAsync(count_out, 0, 8, s)); return; }
    const size_t bm = ((size_t)n + 63) / 64 * 8 + 8;
    if (!t->visited_ready) { HIPCHECK(hipMemsetAsync(t->visited.ensure(bm), 0, bm, s)); t->visited_ready = true; }
    const i64 nwords = (n + 63) >> 6;
    const i64 maxb = (i64)t->ctx->cus * 8;
    i64 wpb = (nwords + maxb - 1) / maxb; if (wpb < 16) wpb = 16;
    const int nblocks = (int)((nwords + wpb - 1) / wpb);
    u64* bitmap = (u64*)t->ws_bitmap.ensure((size_t)nwords * 8);
    uint32_t* counts = (uint32_t*)t->ws_counts.ensure((size_t)nblocks * 4 + 16 + 8);
    u64* total = count_out ? (u64*)count_out : (u64*)((char*)counts + (size_t)nblocks * 4 + 8 - ((size_t)nblocks * 4) % 8);
    launch_bitmap_select(s, (const u64*)t->present.p, (const u64*)t->visited.p, matched, nwords, n, bitmap, counts, nblocks, wpb);
    launch_scan_block_counts(s, counts, nblocks, total);
    if (rows_out) launch_compact(s, bitmap, counts, nblocks, wpb, n, nullptr, rows_out);
    HIPCHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------- sort
static int bitlen128(u128 v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

// Per-key min/max in the ordered view (one pass + read-back) -> the bit layout of the composite key (SortExec and the ordered merge).
// wstep > 1: the min/max of a strided SAMPLE (every wstep-th 64-row word), widened into a GUESS of the layout that the pack kernel
// then verifies row by row (K.check; gpuq_sort_run falls back to the exact pass when it does not hold).  The guess keeps the cost
// class of the sample's own layout (number of passes, packed 8-byte records or not): it is widened by 1/64 of the range on both
// sides if that is free, by 1/4096 if not, and always spread over the whole 2^bits its width has anyway.
static SortPack sort_key_plan(gpuq_op* op, hipStream_t s, const DevProgram& P, i64 n, int* total_out, i64 wstep = 1) {
    const SortSpec& S = op->sort;
    const i64 n_seen = wstep > 1 ? ((((n + 63) >> 6) + wstep - 1) / wstep + 1) << 6 : n;
    const int mb = sort_minmax_blocks(n_seen);
    u64* mm = (u64*)op->ws[0].ensure((size_t)mb * MAX_SORT_KEYS * 5 * 8);
    { JitScope js(op, op->prog, 8, n); launch_sort_minmax(s, P, n, S, mm, mb, wstep); }
    std::vector<u64> hmm((size_t)mb * MAX_SORT_KEYS * 5);
    HIPCHECK(hipMemcpyAsync(hmm.data(), mm, hmm.size() * 8, hipMemcpyDeviceToHost, s));
    // the pass has evaluated the key expressions of every row it saw: what they could not do (a Utf8 value beyond the 15 bytes a
    // packed key holds) is reported with the same read-back instead of sorting by a truncated key
    uint32_t eflags = 0;
    HIPCHECK(hipMemcpyAsync(&eflags, op->flags_dev.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (eflags) { reset_flags(op, s); raise_flags(eflags); }
    const i128 I128_MAX = ((i128)0x7FFFFFFFFFFFFFFFll << 64) | (i128)0xFFFFFFFFFFFFFFFFull, I128_MIN = -I128_MAX - 1;
    i128 mn[MAX_SORT_KEYS], mx[MAX_SORT_KEYS]; u64 fl[MAX_SORT_KEYS] = {0, 0, 0, 0}; int rshift[MAX_SORT_KEYS] = {0, 0, 0, 0};
    for (int k = 0; k < S.n_keys; ++k) {
      mn[k] = I128_MAX; mx[k] = I128_MIN;
      for (int b = 0; b < mb; ++b) {
        const u64* o = &hmm[((size_t)b * MAX_SORT_KEYS + k) * 5];
        if (!(o[4] & 1)) { fl[k] |= (o[4] & 0xFF); continue; }
        const i128 a = (i128)(((u128)o[1] << 64) | o[0]), c = (i128)(((u128)o[3] << 64) | o[2]);
        if (a < mn[k]) mn[k] = a;
        if (c > mx[k]) mx[k] = c;
        fl[k] = ((fl[k] | o[4]) & 0xFF) | std::max<u64>(fl[k] & 0xFF00, o[4] & 0xFF00);
      }
      if (S.kind[k] == 2) { const int maxlen = std::min<int>((int)((fl[k] >> 8) & 0xFF), 15); rshift[k] = 8 * (15 - maxlen) + 8; }
      if (fl[k] & 1) { mn[k] >>= rshift[k]; mx[k] >>= rshift[k]; }
    }
    // layout for the ranges [mn - range >> mshift, mx + range >> mshift] (mshift < 0: the ranges as they are)
    auto layout = [&](int mshift, bool spread, SortPack& K) {
      int width[MAX_SORT_KEYS] = {0, 0, 0, 0}; int total = 0;
      K = SortPack{};
      for (int k = 0; k < S.n_keys; ++k) {
        int vb = 0; K.rshift[k] = rshift[k];
        if (fl[k] & 1) {
          i128 lo = mn[k], hi = mx[k];
          if (mshift >= 0) {
            const u128 m = (((u128)(hi - lo)) >> mshift) + 1;
            lo = (u128)(lo - I128_MIN) > m ? lo - (i128)m : I128_MIN;
            hi = (u128)(I128_MAX - hi) > m ? hi + (i128)m : I128_MAX;
          }
          vb = bitlen128((u128)(hi - lo));
          if (spread && vb < 127) {      // the field holds 2^vb values whatever the range: centre the range in it
            const u128 spare = (((u128)1 << vb) - 1) - (u128)(hi - lo), down = spare / 2;
            lo = (u128)(lo - I128_MIN) > down ? lo - (i128)down : I128_MIN;
            hi = (u128)(I128_MAX - lo) > (((u128)1 << vb) - 1) ? lo + (i128)(((u128)1 << vb) - 1) : I128_MAX;
          }
          const i128 base = S.desc[k] ? hi : lo; K.base_lo[k] = (u64)base; K.base_hi[k] = (u64)((u128)base >> 64);
        }
        K.vbits[k] = vb;
        K.null_bit[k] = (fl[k] & 2) ? vb : -1;
        width[k] = vb + ((fl[k] & 2) ? 1 : 0);
        total += width[k];
      }
      int sh = 0; for (int k = S.n_keys - 1; k >= 0; --k) { K.shift[k] = sh; sh += width[k]; }
      return total;
    };
    auto cost_class = [](int total) { return ((total + 7) / 8) * 4 + (total <= 32 ? 0 : total <= 64 ? 1 : 2); };
    SortPack K{};
    int total = layout(-1, false, K);
    if (wstep > 1) {
      const int cls = cost_class(total);
      SortPack G{}; int t = layout(6, true, G);
      if (cost_class(t) != cls) t = layout(12, true, G);
      if (cost_class(t) != cls) t = layout(-1, true, G);
      K = G; total = t; K.check = 1;
    }
    if (total > 128) throw Unsupported("composite sort key needs " + std::to_string(total) + " bits (max 128)");
    *total_out = total;
    return K;
}

static void sort_with_plan(gpuq_op* op, hipStream_t s, const DevProgram& P, const i64 n, const SortPack& K, const int total, uint32_t* perm_out);
int gpuq_sort_run_keys(gpuq_op* op, void* stream, const gpuq_input* in, uint32_t* perm_out, void* key_data_out, uint8_t* key_validity_out, int* decoded_out) {
  if (decoded_out) *decoded_out = 0;
  if (!op) return GPUQ_ERR_INVALID;
  op->sort_dec.data = key_data_out; op->sort_dec.valid = (u64*)key_validity_out; op->sort_dec.done = false;
  const int rc = gpuq_sort_run(op, stream, in, perm_out);
  if (decoded_out) *decoded_out = (rc == GPUQ_OK && op->sort_dec.done) ? 1 : 0;
  op->sort_dec.data = nullptr; op->sort_dec.valid = nullptr;
  return rc;
}
int gpuq_sort_run(gpuq_op* op, void* stream, const gpuq_input* in, uint32_t* perm_out) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() {
    check_ctx(op->ctx);
    if (op->kind != K_SORT) throw std::runtime_error("not a sort operator");
    hipStream_t s = use_stream(stream);
    ProfTotal ptot(op, s);
    DevProgram P = bind_program(op->prog, op->in_schema, op->code_dev.as<DevCode>(), op->flags_dev.as<uint32_t>(), in);
    const i64 n = in->n_rows;
    if (n == 0) return;
    if (n >= (1ll << 31)) throw Unsupported("sort of >= 2^31 rows in one call");
    if (!perm_out) throw std::runtime_error("perm_out is NULL");
    const SortSpec& S = op->sort;
    bool string_key = false;
    for (int k = 0; k < S.n_keys; ++k) string_key = string_key || S.kind[k] == 2;
    if (n <= sort_direct_max()) {      // one block, no min/max read-back
      launch_sort_direct(s, P, n, S, perm_out);
      HIPCHECK(hipGetLastError());
      if (op->deferred) return;      // (the status word is read by gpuq_ops_settle)
      if (in->n_rows_dev) throw std::runtime_error("sort: a device-side row count needs a deferred operator (gpuq_op_set_deferred)");
      if (string_key) { const uint32_t f = read_flags(op, s); if (f) { reset_flags(op, s); raise_flags(f); } }      // a value beyond 15 bytes: refuse, do not sort by a prefix
      // a client that defers (gpuq_op_set_deferred was called on this operator) will come back with a BOUND instead of a count, and
      // the bound of a handful of groups is easily beyond what one block sorts: learn the key layout now, while reading back is allowed
      if (op->defer_client) { int total = 0; const SortPack K = sort_key_plan(op, s, P, n, &total); op->so.valid = true; op->so.K = K; op->so.total = total; op->expect_flags = 0; }
      return;
    }
    int total = 0;
    // deferred: the key layout of the last completed synchronous run, verified row by row by the pack kernel (FLAG_SORT_LAYOUT); rows
    // beyond the device-side count become padding records that sort behind everything
    if (op->deferred && op->so.valid) {
      SortPack K = op->so.K; K.check = 2;
      sort_with_plan(op, s, P, n, K, op->so.total, perm_out);
      return;
    }
    if (in->n_rows_dev) throw std::runtime_error("sort: a device-side row count needs a deferred operator with a completed synchronous run (gpuq_op_can_defer)");
    op->so.valid = false;
    // Large inputs: the exact min/max pass reads every key once more (0.48 of 3.15 ms at 2^27 Decimal128 keys).  Guess the layout from
    // 2^18 sampled rows instead, let the pack kernel verify it on the way, and read one word back at the end; a guess that does not
    // hold (outliers beyond the margin, a NULL the sample did not see) costs the pack + passes again, so an operator whose guess
    // failed stops guessing.  Packed strings are left out (their layout depends on the longest value).
    static const bool spec_on = []() { const char* e = getenv("GPUQ_SORT_SPECULATE"); return !(e && e[0] == '0'); }();
    bool spec = spec_on && n >= (1ll << 22) && !op->sort_guess_failed;
    for (int k = 0; k < S.n_keys; ++k) if (S.kind[k] == 2) spec = false;
    if (spec) {
      const i64 wstep = std::max<i64>(1, ((n + 63) >> 6) >> 12);      // ~4096 words of 64 rows
      const SortPack G = sort_key_plan(op, s, P, n, &total, wstep);
      if (total > 0) {      // (no key bits: the pack kernel takes no counts and has nowhere to report to)
        sort_with_plan(op, s, P, n, G, total, perm_out);
        u64 failed = 0; uint32_t eflags = 0;
        HIPCHECK(hipMemcpyAsync(&failed, (const u64*)op->ws[6].p + (size_t)sort_max_passes() * 256, 8, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(&eflags, op->flags_dev.p, 4, hipMemcpyDeviceToHost, s));      // the pack kernel evaluated every row
        HIPCHECK(hipStreamSynchronize(s));
        if (eflags) { reset_flags(op, s); raise_flags(eflags); }
        if (!failed) { op->so.valid = true; op->so.K = G; op->so.total = total; op->expect_flags = 0; return; }
        op->sort_guess_failed = true;
      }
    }
    const SortPack K = sort_key_plan(op, s, P, n, &total);
    sort_with_plan(op, s, P, n, K, total, perm_out);
    op->so.valid = true; op->so.K = K; op->so.total = total; op->expect_flags = 0;
  });
}

// pack + LSD radix passes over a composite key whose layout is known (SortExec; the ordered fan-in when passes are cheaper than rounds)
static void sort_with_plan(gpuq_op* op, hipStream_t s, const DevProgram& P, const i64 n, const SortPack& K_in, const int total, uint32_t* perm_out) {
  {
    SortPack K = K_in; K.total_bits = total;
    const SortSpec& S = op->sort;
    // 2. pack + LSD radix passes.  <= 32 key bits: one u64 (key << 32 | row) record per row, no separate id array.
    const bool packed = total >= 1 && total <= 32 && n > sort_small_max();
    u64* klo = (u64*)op->ws[1].ensure((size_t)n * 8);
    u64* klo2 = (u64*)op->ws[2].ensure((size_t)n * 8);
    u64* khi = total > 64 ? (u64*)op->ws[3].ensure((size_t)n * 8) : nullptr;
    uint32_t* ids = packed ? nullptr : (uint32_t*)op->ws[4].ensure((size_t)n * 4);
    uint32_t* ids2 = packed ? nullptr : (uint32_t*)op->ws[5].ensure((size_t)n * 4);
    ProfScope ps(op, s);
    // digit counts of every pass (256 u64 each), taken by the pack kernel on the way
    const int np_all = (total + 7) / 8;
    u64* ghist = (u64*)op->ws[6].ensure(((size_t)sort_max_passes() * 256 + 1) * 8);      // + the "guessed layout does not hold" word
    const bool small = n <= sort_small_max();
    // gpuq_sort_run_keys: one integer-like key whose field IS value - base (no string shift), at most one 64-bit word of composite
    const int dec_width = (op->sort_dec.data && S.n_keys == 1 && S.kind[0] == 0 && K.rshift[0] == 0 && total >= 1 && total <= 64 && !small) ? type_width(op->sort_key0) : 0;
    const bool decode = dec_width == 1 || dec_width == 2 || dec_width == 4 || dec_width == 8 || (dec_width == 16 && op->sort_key0.id == T_DECIMAL128);
    { JitScope js(op, op->prog, 9, n); launch_sort_pack(s, P, n, S, K, klo, khi, ids, small || total == 0 ? nullptr : ghist, np_all); }
    if (small) {      // one block sorts it in LDS: no histogram / scan / scatter launches
      launch_sort_small(s, klo, khi, ids, n, perm_out);
      HIPCHECK(hipGetLastError());
      return;
    }
    // single-read passes (kernels_sort.hip): one look-back kernel per 8 key bits, the last one writes the row ids straight into perm_out
    const size_t lwb = onesweep_ws_bytes(n);
    void* lws = op->ws[7].ensure(lwb);
    auto run_passes = [&](int bits, int shift0, int pass0, bool last_word) {
      const int np = (bits + 7) / 8;
      if (np == 0) { if (last_word) HIPCHECK(hipMemcpyAsync(perm_out, ids, (size_t)n * 4, hipMemcpyDeviceToDevice, s)); return; }
      for (int p = 0; p < np; ++p) {
        const bool final_pass = last_word && p + 1 == np;
        // (decode: the last pass keeps its records as well -- packed: records AND ids, otherwise keys and ids -- the key column is rebuilt from them)
        launch_onesweep_pass(s, klo, ids, n, shift0 + 8 * p, ghist + (size_t)(pass0 + p) * 256, lws, lwb, klo2, final_pass ? perm_out : ids2, final_pass ? (decode ? (packed ? 2 : 0) : 1) : 0);
        std::swap(klo, klo2); if (!final_pass) std::swap(ids, ids2);
      }
    };
    if (packed) {
      run_passes(total, 32, 0, true);
      if (decode) { launch_sort_decode(s, klo, 32, n, K, S.desc[0], S.nulls_first[0], dec_width, op->sort_dec.data, op->sort_dec.valid); op->sort_dec.done = true; }
    } else if (decode) {
      run_passes(total, 0, 0, true);
      launch_sort_decode(s, klo, 0, n, K, S.desc[0], S.nulls_first[0], dec_width, op->sort_dec.data, op->sort_dec.valid); op->sort_dec.done = true;
    } else {
      run_passes(std::min(total, 64), 0, 0, total <= 64);
      if (total > 64) {
        launch_gather_u64(s, khi, ids, n, klo);   // hi words in the current order (their digit counts are passes 8.. of the pack kernel's)
        run_passes(total - 64, 0, 8, true);
      }
    }
    HIPCHECK(hipGetLastError());
  }
}

// ---------------------------------------------------------------- ordered fan-in
int gpuq_merge_run(gpuq_op* op, void* stream, const gpuq_input* in, const int64_t* run_offsets, int n_runs, uint32_t* perm_out) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() {
    check_ctx(op->ctx);
    if (op->kind != K_SORT) throw std::runtime_error("not a sort operator");
    if (!run_offsets || n_runs < 1) throw std::runtime_error("run_offsets is NULL / no runs");
    hipStream_t s = use_stream(stream);
    need_exact_rows(in, "merge");
    const i64 n = in->n_rows;
    if (run_offsets[0] != 0 || run_offsets[n_runs] != n) throw std::runtime_error("run_offsets must start at 0 and end at the row count");
    for (int r = 0; r < n_runs; ++r) if (run_offsets[r] > run_offsets[r + 1]) throw std::runtime_error("run_offsets must not decrease");
    if (n == 0) return;
    if (n >= (1ll << 31)) throw Unsupported("merge of >= 2^31 rows in one call");
    if (!perm_out) throw std::runtime_error("perm_out is NULL");
    DevProgram P = bind_program(op->prog, op->in_schema, op->code_dev.as<DevCode>(), op->flags_dev.as<uint32_t>(), in);
    int total = 0;
    const SortPack K = sort_key_plan(op, s, P, n, &total);
    // Rounds or passes?  Both give the same permutation (the stable sort of the concatenation IS the merge that prefers the lower
    // run on ties).  Measured at 2^27 records: a merge-path round 1.45 ms (est. 1.9 with a two-word key), a radix pass 0.65 ms on packed
    // 8-byte records (<= 32 key bits), 1.0 ms otherwise.  Few runs and wide keys merge; many runs of narrow keys take the passes.
    {
      int live = 0; for (int r = 0; r < n_runs; ++r) live += run_offsets[r + 1] > run_offsets[r];
      int rounds = 0; while ((1 << rounds) < live) ++rounds;
      const double est_merge = rounds * (total > 64 ? 1.9 : 1.45);
      const double est_passes = ((total + 7) / 8) * (total <= 32 ? 0.65 : 1.0) + (total > 64 ? 1.0 : 0.0);
      static const bool force_rounds = []() { const char* e = std::getenv("GPUQ_MERGE_ROUNDS"); return e && std::string(e) == "force"; }();      // measurement switch
      if (live > 1 && est_passes < est_merge && !force_rounds) { sort_with_plan(op, s, P, n, K, total, perm_out); return; }
    }
    u64* klo = (u64*)op->ws[1].ensure((size_t)n * 8);
    u64* klo2 = (u64*)op->ws[2].ensure((size_t)n * 8);
    u64* khi = total > 64 ? (u64*)op->ws[3].ensure((size_t)n * 8) : nullptr;
    u64* khi2 = total > 64 ? (u64*)op->ws[8].ensure((size_t)n * 8) : nullptr;
    uint32_t* ids = (uint32_t*)op->ws[4].ensure((size_t)n * 4);
    uint32_t* ids2 = (uint32_t*)op->ws[5].ensure((size_t)n * 4);
    ProfScope ps(op, s);
    { JitScope js(op, op->prog, 9, n); launch_sort_pack(s, P, n, op->sort, K, klo, khi, ids, nullptr, 0); }
    // runs -> pairs, round by round (empty runs drop out; an odd run is carried as a pair with an empty right side)
    std::vector<i64> bounds; bounds.push_back(0);
    for (int r = 0; r < n_runs; ++r) if (run_offsets[r + 1] > run_offsets[r]) bounds.push_back(run_offsets[r + 1]);
    i64* dpairs = (i64*)op->ws[9].ensure((size_t)(bounds.size() + 2) * 3 * 8);
    while (bounds.size() > 2) {
      std::vector<i64> pairs, next; next.push_back(0); i64 max_len = 0;
      for (size_t r = 0; r + 1 < bounds.size(); r += 2) {
        const i64 a0 = bounds[r], a1 = bounds[r + 1], a2 = r + 2 < bounds.size() ? bounds[r + 2] : a1;
        pairs.push_back(a0); pairs.push_back(a1); pairs.push_back(a2);
        max_len = std::max(max_len, a2 - a0); next.push_back(a2);
      }
      HIPCHECK(hipStreamSynchronize(s));      // the previous round has read its descriptors
      HIPCHECK(hipMemcpyAsync(dpairs, pairs.data(), pairs.size() * 8, hipMemcpyHostToDevice, s));
      HIPCHECK(hipStreamSynchronize(s));      // (pageable source)
      i64* splits = (i64*)op->ws[0].ensure(merge_splits_entries(max_len, (int)(pairs.size() / 3)) * 8 + 64);
      launch_merge_pairs(s, klo, khi, ids, dpairs, (int)(pairs.size() / 3), max_len, splits, klo2, khi2, ids2);
      std::swap(klo, klo2); std::swap(khi, khi2); std::swap(ids, ids2);
      bounds.swap(next);
    }
    HIPCHECK(hipMemcpyAsync(perm_out, ids, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------- partition
int gpuq_partition_run(gpuq_op* op, void* stream, const gpuq_input* in, uint32_t* perm_out, uint64_t* part_offsets_out) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() {
    check_ctx(op->ctx);
    if (op->kind != K_PARTITION) throw std::runtime_error("not a partition operator");
    need_exact_rows(in, "partition");
    hipStream_t s = use_stream(stream);
    DevProgram P = bind_program(op->prog, op->in_schema, op->code_dev.as<DevCode>(), op->flags_dev.as<uint32_t>(), in);
    const i64 n = in->n_rows;
    const uint32_t np = op->nparts;
    if (!part_offsets_out) throw std::runtime_error("part_offsets_out is NULL");
    if (n == 0) { HIPCHECK(hipMemsetAsync(part_offsets_out, 0, (size_t)(np + 1) * 8, s)); return; }
    if (n >= (1ll << 31)) throw Unsupported("partition of >= 2^31 rows in one call");
    if (!perm_out) throw std::runtime_error("perm_out is NULL");
    // one 8-byte (partition << 32 | row) record per row; the partition sizes double as the digit counts of the single-read pass
    // (<= 256 partitions: one pass that reads 8 and writes 4 bytes per row; more: one pass per 8 bits of the partition id)
    u64* pid = (u64*)op->ws[1].ensure((size_t)n * 8);
    u64* pid2 = (u64*)op->ws[2].ensure((size_t)n * 8);
    uint32_t* counts = (uint32_t*)op->ws[8].ensure((size_t)(np + 1) * 4 + 16);
    u64* ghist = (u64*)op->ws[6].ensure((size_t)sort_max_passes() * 256 * 8);
    const size_t lwb = onesweep_ws_bytes(n);
    void* lws = op->ws[7].ensure(lwb);
    ProfScope ps(op, s);
    { JitScope js(op, op->prog, 10, n); launch_part_pid(s, P, n, op->keys, np, pid, nullptr); }
    launch_part_offsets(s, pid, n, np, counts, (u64*)part_offsets_out, 32);
    int bits = 0; while ((1u << bits) < np) ++bits;
    const int npass = std::max(1, (bits + 7) / 8);
    if (npass == 1) launch_counts_to_ghist(s, counts, np, ghist);
    else launch_radix_ghist(s, pid, n, 32, npass, ghist);
    for (int p = 0; p < npass; ++p) {
      const bool final_pass = p + 1 == npass;
      launch_onesweep_pass(s, pid, nullptr, n, 32 + 8 * p, ghist + (size_t)p * 256, lws, lwb, pid2, final_pass ? perm_out : nullptr, final_pass ? 1 : 0);
      std::swap(pid, pid2);
    }
    HIPCHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------- fan-in support
int gpuq_concat_bitmap(gpuq_ctx* ctx, void* stream, uint8_t* dst, int64_t dst_bit_offset, const uint8_t* src, int64_t n_bits) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!dst || dst_bit_offset < 0 || n_bits < 0) throw std::runtime_error("bad arguments");
    if (((uintptr_t)dst & 7) != 0) throw std::runtime_error("dst bitmap must be 8-byte aligned");
    launch_concat_bitmap(use_stream(stream), (u64*)dst, dst_bit_offset, src, 0, n_bits);
    HIPCHECK(hipGetLastError());
  });
}
int gpuq_copy_bits(gpuq_ctx* ctx, void* stream, uint8_t* dst, int64_t dst_bit_offset, const uint8_t* src, int64_t src_bit_offset, int64_t n_bits) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!dst || dst_bit_offset < 0 || src_bit_offset < 0 || n_bits < 0) throw std::runtime_error("bad arguments");
    if (((uintptr_t)dst & 7) != 0) throw std::runtime_error("dst bitmap must be 8-byte aligned");
    launch_concat_bitmap(use_stream(stream), (u64*)dst, dst_bit_offset, src, src_bit_offset, n_bits);
    HIPCHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------- utf8 take (payload strings of any length)
int gpuq_take_utf8(gpuq_ctx* ctx, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, int32_t* offsets_out, uint8_t* validity_out,
                   uint8_t* data_out, int64_t data_cap, int64_t* data_len_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    hipStream_t s = use_stream(stream);
    if (!col || n < 0 || !offsets_out) throw std::runtime_error("bad arguments");
    if (col->type != T_UTF8 || col->repr != GPUQ_REPR_ARROW || (!col->offsets && col->length > 0)) throw std::runtime_error("gpuq_take_utf8 takes an Arrow-layout Utf8 column (offsets + bytes)");
    if (validity_out && ((uintptr_t)validity_out & 7)) throw std::runtime_error("validity_out must be 8-byte aligned");
    if (n == 0) { HIPCHECK(hipMemsetAsync(offsets_out, 0, 4, s)); if (data_len_out) *data_len_out = 0; return; }
    DevBuf ws; ws.ensure(exclusive_scan_ws_bytes(n));
    launch_take_utf8_lengths(s, col->offsets, col->validity, idx, n, offsets_out, (u64*)validity_out);
    launch_exclusive_scan_i32(s, offsets_out, n, ws.p, ws.cap);
    int32_t total = 0;
    HIPCHECK(hipMemcpyAsync(&total, offsets_out + n, 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (data_len_out) *data_len_out = total;
    if (total > data_cap) throw Capacity("utf8 data needs " + std::to_string(total) + " bytes");
    if (total > 0) {
      if (!data_out) throw std::runtime_error("data_out is NULL");
      launch_take_utf8_bytes(s, (const uint8_t*)col->data, col->offsets, idx, n, offsets_out, data_out, (i64)total);
    }
    HIPCHECK(hipGetLastError());
  });
}

int gpuq_offsets_rebase(gpuq_ctx* ctx, void* stream, const int32_t* src, int64_t n, int32_t delta, int32_t* dst) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (n < 0 || (n > 0 && (!src || !dst))) throw std::runtime_error("bad arguments");
    launch_offsets_rebase(use_stream(stream), src, n, delta, dst);
    HIPCHECK(hipGetLastError());
  });
}

int gpuq_mark_rows(gpuq_ctx* ctx, void* stream, const uint32_t* rows, int64_t n, uint8_t* bitmap) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (n < 0 || (n > 0 && (!rows || !bitmap))) throw std::runtime_error("bad arguments");
    launch_mark_rows(use_stream(stream), rows, n, bitmap);
    HIPCHECK(hipGetLastError());
  });
}

int gpuq_memory_limit(int64_t bytes) {
  if (bytes < 0) return GPUQ_ERR_INVALID;
  DevPool& P = DevPool::get();
  std::lock_guard<std::mutex> lk(P.mu);
  P.limit = (size_t)bytes;
  return GPUQ_OK;
}
int gpuq_memory_stats(int64_t* in_use_out, int64_t* peak_out, int64_t* cached_out, int64_t* limit_out, int reset_peak) {
  DevPool& P = DevPool::get();
  std::lock_guard<std::mutex> lk(P.mu);
  if (in_use_out) *in_use_out = (int64_t)P.in_use;
  if (peak_out) *peak_out = (int64_t)P.peak;
  if (cached_out) *cached_out = (int64_t)P.held;
  if (limit_out) *limit_out = (int64_t)P.limit;
  if (reset_peak) P.peak = P.in_use;
  return GPUQ_OK;
}
int gpuq_cross_pairs(gpuq_ctx* ctx, void* stream, int64_t n_left, int64_t n_right, uint32_t* left_rows_out, uint32_t* right_rows_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (n_left < 0 || n_right < 0) throw std::runtime_error("bad arguments");
    if (n_left > 0 && n_right > 0 && n_left > (int64_t)0xFFFFFFFEll / n_right) throw Capacity("cross join of " + std::to_string(n_left) + " x " + std::to_string(n_right) + " rows exceeds 2^32 pairs");
    if (n_left * n_right > 0 && (!left_rows_out || !right_rows_out)) throw std::runtime_error("output vectors are NULL");
    launch_cross_pairs(use_stream(stream), n_left, n_right, left_rows_out, right_rows_out);
    HIPCHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------- Utf8 dictionary codes (keys longer than 15 bytes)
struct gpuq_utf8_dict { gpuq_ctx* ctx; DevBuf table; u64 mask = 0; gpuq_column dict_col{}; bool filled = false; };

int gpuq_utf8_max_len(gpuq_ctx* ctx, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, int32_t* max_len_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!col || !max_len_out) throw std::runtime_error("col / max_len_out is NULL");
    if (col->type != T_UTF8 || col->repr != GPUQ_REPR_ARROW) throw Unsupported("gpuq_utf8_max_len needs a Utf8 column in Arrow layout");
    *max_len_out = 0;
    if (n <= 0) return;
    hipStream_t s = use_stream(stream);
    DevBuf out; out.ensure(16); HIPCHECK(hipMemsetAsync(out.p, 0, 16, s));
    launch_utf8_max_len(s, col->offsets, col->validity, idx, n, (int32_t*)out.p);
    HIPCHECK(hipMemcpyAsync(max_len_out, out.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
  });
}
int gpuq_utf8_sort_piece(gpuq_ctx* ctx, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, int piece, void* keys_out, uint8_t* validity_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!col || !keys_out) throw std::runtime_error("col / keys_out is NULL");
    if (col->type != T_UTF8 || col->repr != GPUQ_REPR_ARROW) throw Unsupported("gpuq_utf8_sort_piece needs a Utf8 column in Arrow layout");
    if (piece < 0 || piece > (1 << 27)) throw std::runtime_error("piece out of range");
    if (n <= 0) return;
    launch_utf8_sort_piece(use_stream(stream), (const uint8_t*)col->data, col->offsets, col->validity, idx, n, piece, keys_out, (u64*)validity_out);
    HIPCHECK(hipGetLastError());
  });
}
int gpuq_utf8_dict_create(gpuq_ctx* ctx, void* stream, int64_t capacity_rows, gpuq_utf8_dict** out) {
  if (out) *out = nullptr;
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!out || capacity_rows < 0) throw std::runtime_error("out is NULL / negative capacity");
    std::unique_ptr<gpuq_utf8_dict> d(new gpuq_utf8_dict()); d->ctx = ctx;
    u64 slots = 1024; while (slots < (u64)capacity_rows * 2) slots <<= 1;
    d->mask = slots - 1;
    d->table.ensure((size_t)slots * 8);
    HIPCHECK(hipMemsetAsync(d->table.p, 0, (size_t)slots * 8, use_stream(stream)));
    *out = d.release();
  });
}
void gpuq_utf8_dict_free(gpuq_utf8_dict* d) { delete d; }
int gpuq_utf8_code_rows(gpuq_ctx* ctx, void* stream, const gpuq_column* codes, int64_t n, uint32_t* rows_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!codes || (n > 0 && !rows_out)) throw std::runtime_error("codes / rows_out is NULL");
    if (codes->type != T_INT64 && codes->type != T_UINT32) throw std::runtime_error("gpuq_utf8_code_rows: the code column is Int64 (or UInt32: a code is a row id)");
    launch_utf8_code_rows(use_stream(stream), codes->data, codes->type == T_INT64 ? 8 : 4, codes->validity, n, rows_out);
    HIPCHECK(hipGetLastError());
  });
}
int gpuq_utf8_intern(gpuq_utf8_dict* d, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, int insert, int64_t* codes_out, uint8_t* validity_out) {
  if (!d) return GPUQ_ERR_INVALID;
  return guarded(d->ctx, [&]() {
    check_ctx(d->ctx);
    if (!col || (n > 0 && (!codes_out || !validity_out))) throw std::runtime_error("col / codes_out / validity_out is NULL");
    if (col->type != T_UTF8 || col->repr != GPUQ_REPR_ARROW || (col->length > 0 && !col->offsets)) throw Unsupported("gpuq_utf8_intern needs a Utf8 column in Arrow layout (offsets + bytes)");
    if (insert) {
      if (d->filled && (d->dict_col.data != col->data || d->dict_col.offsets != col->offsets)) throw std::runtime_error("a dictionary is filled from ONE column (its codes are that column's row ids)");
      if ((u64)std::min<int64_t>(n, col->length) * 2 > d->mask + 1) throw Capacity("dictionary created for fewer rows than are inserted");
      d->dict_col = *col; d->filled = true;
    } else if (!d->filled) throw std::runtime_error("lookup in a dictionary nothing was inserted into");
    if (n <= 0) return;
    hipStream_t s = use_stream(stream);
    DevBuf flags; flags.ensure(16); HIPCHECK(hipMemsetAsync(flags.p, 0, 16, s));
    launch_utf8_intern(s, (const uint8_t*)d->dict_col.data, d->dict_col.offsets, (const uint8_t*)col->data, col->offsets, col->validity, idx, n,
                       (u64*)d->table.p, d->mask, insert ? 1 : 0, (i64*)codes_out, (u64*)validity_out, (uint32_t*)flags.p);
    uint32_t fl = 0; HIPCHECK(hipMemcpyAsync(&fl, flags.p, 4, hipMemcpyDeviceToHost, s)); HIPCHECK(hipStreamSynchronize(s));
    if (fl) throw Capacity("string dictionary is full");
  });
}

// ---------------------------------------------------------------- LIKE
int gpuq_utf8_compare(gpuq_ctx* ctx, void* stream, const gpuq_column* a, const uint32_t* idx_a, const gpuq_column* b, const uint32_t* idx_b, const char* literal,
                      int64_t literal_len, int64_t n, int op, uint8_t* bits_out, uint8_t* validity_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!a || (n > 0 && !bits_out) || op < 0 || op > 5) throw std::runtime_error("gpuq_utf8_compare: bad arguments");
    auto arrow = [&](const gpuq_column* c) { if (c->type != T_UTF8 || c->repr != GPUQ_REPR_ARROW || (n > 0 && !c->offsets)) throw Unsupported("gpuq_utf8_compare needs Utf8 columns in Arrow layout (offsets + bytes)"); };
    arrow(a); if (b) arrow(b);
    if (!b && (!literal || literal_len < 0 || literal_len > 0x7FFFFFFFll)) throw std::runtime_error("gpuq_utf8_compare: neither a second column nor a literal");
    hipStream_t s = use_stream(stream);
    DevBuf lit;      // released to the pool behind the launch on this stream
    if (!b) { lit.ensure((size_t)literal_len + 16); if (literal_len) HIPCHECK(hipMemcpyAsync(lit.p, literal, (size_t)literal_len, hipMemcpyHostToDevice, s)); }      // (pageable source: staged before the call returns)
    launch_utf8_compare(s, (const uint8_t*)a->data, a->offsets, a->validity, idx_a, b ? (const uint8_t*)b->data : (const uint8_t*)lit.p, b ? b->offsets : nullptr,
                        b ? b->validity : nullptr, idx_b, (int32_t)literal_len, n, op, (u64*)bits_out, (u64*)validity_out);
    HIPCHECK(hipGetLastError());
  });
}
int gpuq_like_utf8(gpuq_ctx* ctx, void* stream, const gpuq_column* col, const uint32_t* idx, int64_t n, const char* pattern, int negated, int case_insensitive,
                   uint8_t* bits_out, uint8_t* validity_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!col || !pattern || (n > 0 && !bits_out)) throw std::runtime_error("col/pattern/bits_out is NULL");
    if (col->type != T_UTF8 || col->repr != GPUQ_REPR_ARROW || (n > 0 && !col->offsets)) throw Unsupported("LIKE needs a Utf8 column in Arrow layout (offsets + bytes)");
    if (case_insensitive) throw Unsupported("ILIKE (case_insensitive) is not supported on device: arrow upper-cases both sides with full Unicode case mapping");
    const std::string r = pattern;
    auto wild = [](char c) { return c == '%' || c == '_'; };
    auto has_wild = [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) if (wild(r[i])) return true; return false; };
    auto ends_with_esc = [&]() { return r.size() >= 2 && r[r.size() - 2] == '\\' && r.back() == '%'; };
    // arrow-string 49 like.rs op_scalar: the shapes served without a regex [UPSTREAM-KNOWLEDGE]
    bool fast = false;
    const size_t L = r.size();
    if (!has_wild(0, L)) fast = true;                                                                            // equality
    else if (r.back() == '%' && !ends_with_esc() && !has_wild(0, L - 1)) fast = true;                            // starts_with
    else if (r.front() == '%' && !has_wild(1, L)) fast = true;                                                   // ends_with
    else if (L >= 2 && r.front() == '%' && r.back() == '%' && !ends_with_esc() && !has_wild(1, L - 1)) fast = true;   // contains
    LikePattern pat{}; pat.regex_mode = fast ? 0 : 1;
    auto push = [&](int t) { if (pat.n >= LIKE_MAX_TOKENS) throw Unsupported("LIKE pattern longer than 256 tokens"); pat.tok[pat.n++] = (uint16_t)t; };
    for (size_t i = 0; i < L; ++i) {
      const char c = r[i];
      if (c == '\\' && i + 1 < L && wild(r[i + 1])) { push((uint8_t)r[i + 1]); ++i; }      // \% and \_ are literals; any other backslash is itself
      else if (c == '%') { if (!(pat.n > 0 && pat.tok[pat.n - 1] == 257)) push(257); }
      else if (c == '_') push(256);
      else push((uint8_t)c);
    }
    // literal segments between '%' (the segment search of k_like_utf8); '_' anywhere, a literal newline under the regex rule, or more
    // segments than the descriptor holds leave the general matcher in charge
    pat.n_seg = 0; pat.anchored_start = pat.n > 0 && pat.tok[0] != 257; pat.anchored_end = pat.n > 0 && pat.tok[pat.n - 1] != 257;
    if (pat.n == 0) { pat.anchored_start = pat.anchored_end = 1; }
    for (int i = 0; i < pat.n && pat.n_seg >= 0;) {
      if (pat.tok[i] == 257) { ++i; continue; }
      int j = i; bool bad = false;
      while (j < pat.n && pat.tok[j] != 257) { if (pat.tok[j] == 256 || (pat.regex_mode && pat.tok[j] == (uint16_t)'\n')) bad = true; ++j; }
      if (bad || pat.n_seg >= LIKE_MAX_SEGS) { pat.n_seg = -1; break; }
      const int k = pat.n_seg++;
      pat.seg_off[k] = i; pat.seg_len[k] = j - i;
      unsigned long long f8 = 0, m8 = 0;
      for (int q = 0; q < j - i && q < 8; ++q) { f8 |= (unsigned long long)(pat.tok[i + q] & 0xFF) << (8 * q); m8 |= 0xFFull << (8 * q); }
      pat.seg_first8[k] = f8; pat.seg_mask8[k] = m8;
      i = j;
    }
    launch_like_utf8(use_stream(stream), (const uint8_t*)col->data, col->offsets, col->validity, idx, n, pat, negated ? 1 : 0, (u64*)bits_out, (u64*)validity_out);
    HIPCHECK(hipGetLastError());
  });
}

// ---------------------------------------------------------------- utf8 unpack
int gpuq_unpack_utf8(gpuq_ctx* ctx, void* stream, const void* packed, int64_t n, int32_t* offsets_out, uint8_t* data_out, int64_t data_cap,
                     int64_t* data_len_out) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    hipStream_t s = use_stream(stream);
    if (n < 0 || !offsets_out) throw std::runtime_error("bad arguments");
    if (n == 0) { HIPCHECK(hipMemsetAsync(offsets_out, 0, 4, s)); if (data_len_out) *data_len_out = 0; return; }
    DevBuf ws; ws.ensure(exclusive_scan_ws_bytes(n));
    DevBuf flag; flag.ensure(16);
    HIPCHECK(hipMemsetAsync(flag.p, 0, 4, s));
    launch_unpack_utf8_lengths(s, (const ulonglong2*)packed, n, offsets_out, (uint32_t*)flag.p);
    launch_exclusive_scan_i32(s, offsets_out, n, ws.p, ws.cap);
    int32_t total = 0; uint32_t too_long = 0;
    HIPCHECK(hipMemcpyAsync(&total, offsets_out + n, 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(&too_long, flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (too_long) throw Unsupported("a Utf8 value longer than 15 bytes went through a device-side materialisation (PACKED15 limit): project it away or keep it on the host side");
    if (data_len_out) *data_len_out = total;
    if (total > data_cap) throw Capacity("utf8 data needs " + std::to_string(total) + " bytes");
    if (total > 0) { if (!data_out) throw std::runtime_error("data_out is NULL"); launch_unpack_utf8_bytes(s, (const ulonglong2*)packed, n, offsets_out, data_out); }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(s));
  });
}

}  // extern "C"

// ---------------------------------------------------------------- device memory + Arrow C Data Interface
namespace {
struct Staging {   // two pinned buffers, copies alternate between them (H2D overlaps the next memcpy into pinned memory)
  void* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; size_t cap = 0; int next = 0;
  ~Staging() { for (int i = 0; i < 2; ++i) { if (buf[i]) (void)hipHostFree(buf[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); } }
  void ensure() {
    if (cap) return;
    cap = 32u << 20;
    for (int i = 0; i < 2; ++i) { HIPCHECK(hipHostMalloc(&buf[i], cap, hipHostMallocDefault)); HIPCHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
  }
  void h2d(hipStream_t s, void* dst, const void* src, size_t bytes) {
    ensure();
    size_t done = 0;
    while (done < bytes) {
      const size_t n = std::min(cap, bytes - done);
      const int k = next; next ^= 1;
      if (used[k]) HIPCHECK(hipEventSynchronize(ev[k]));
      std::memcpy(buf[k], (const char*)src + done, n);
      HIPCHECK(hipMemcpyAsync((char*)dst + done, buf[k], n, hipMemcpyHostToDevice, s));
      HIPCHECK(hipEventRecord(ev[k], s)); used[k] = true;
      done += n;
    }
  }
  void drain() { for (int k = 0; k < 2; ++k) if (used[k]) { HIPCHECK(hipEventSynchronize(ev[k])); used[k] = false; } }
};
thread_local Staging g_staging;


DType dtype_from_format(const char* f, bool* large) { return dtype_from_arrow_format(f, large); }
std::string format_of(const gpuq_field_info& f) { DType t; t.id = f.type; t.p = f.precision; t.s = f.scale; return arrow_format_of(t); }
// copy `nbits` bits starting at bit `off` of src into a fresh LSB-aligned bitmap
std::vector<uint8_t> realign_bits(const uint8_t* src, int64_t off, int64_t nbits) {
  std::vector<uint8_t> out((size_t)(nbits + 7) / 8 + 8, 0);
  for (int64_t i = 0; i < nbits; ++i) if ((src[(off + i) >> 3] >> ((off + i) & 7)) & 1) out[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
  return out;
}
}  // namespace


extern "C" {
int gpuq_buffer_alloc(gpuq_ctx* ctx, size_t bytes, void** dev_out) {
  return guarded(ctx, [&]() { check_ctx(ctx); if (!dev_out) throw std::runtime_error("dev_out is NULL"); HIPCHECK(hipMalloc(dev_out, bytes ? bytes : 1)); });
}
int gpuq_buffer_free(gpuq_ctx* ctx, void* dev) { return guarded(ctx, [&]() { check_ctx(ctx); if (dev) HIPCHECK(hipFree(dev)); }); }
int gpuq_copy_h2d(gpuq_ctx* ctx, void* stream, void* dst_dev, const void* src_host, size_t bytes) {
  return guarded(ctx, [&]() { check_ctx(ctx); if (bytes) { g_staging.h2d(use_stream(stream), dst_dev, src_host, bytes); g_staging.drain(); } });
}
int gpuq_copy_d2h(gpuq_ctx* ctx, void* stream, void* dst_host, const void* src_dev, size_t bytes) {
  return guarded(ctx, [&]() { check_ctx(ctx); if (bytes) { HIPCHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, use_stream(stream))); HIPCHECK(hipStreamSynchronize(use_stream(stream))); } });
}

int gpuq_table_import_arrow(gpuq_ctx* ctx, void* stream, const struct ArrowArray* batch, const struct ArrowSchema* schema, gpuq_table** out) {
  gpuq_table* t = nullptr;
  int rc = guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!batch || !schema || !out) throw std::runtime_error("batch/schema/out is NULL");
    if (std::string(schema->format ? schema->format : "") != "+s") throw std::runtime_error("expected a struct-typed ArrowArray (RecordBatch)");
    if (batch->n_children != schema->n_children) throw std::runtime_error("array/schema children mismatch");
    hipStream_t s = use_stream(stream);
    t = new gpuq_table(); t->ctx = ctx; t->n_rows = batch->length;
    for (int64_t c = 0; c < batch->n_children; ++c) {
      const ArrowArray* a = batch->children[c]; const ArrowSchema* f = schema->children[c];
      // Dictionary(K, V) (datafusion.proto Dictionary{key, value}): decoded while it is staged -- inside, the column has its value type.
      // The indices may be any integer type; a NULL index or a NULL dictionary entry is a NULL value.
      const ArrowArray* idx = nullptr; const ArrowSchema* idx_f = nullptr;
      if (a->dictionary) {
        if (!f->dictionary) throw std::runtime_error("dictionary array without a dictionary schema");
        idx = a; idx_f = f; a = idx->dictionary; f = idx_f->dictionary;
      }
      bool large = false;
      const DType ty = dtype_from_format(f->format, &large);
      auto ic = std::make_unique<ImportedCol>();
      const ArrowArray* top = idx ? idx : a;
      const int64_t n = top->length, off = top->offset + batch->offset;
      const char* nm = (idx ? idx_f : f)->name;
      ic->field = make_field(nm ? nm : "", ty, (((idx ? idx_f : f)->flags) & 2) != 0);
      ic->field.repr = GPUQ_REPR_ARROW;
      ic->col.type = ty.id; ic->col.precision = ty.p; ic->col.scale = ty.s; ic->col.repr = GPUQ_REPR_ARROW; ic->col.length = n;
      // the values of column rows [0, n): for a dictionary column picked through the indices on the host (this entry point stages
      // through host memory anyway), otherwise the array's own buffers
      std::vector<uint8_t> v_valid, v_data; std::vector<int64_t> v_offs;
      const uint8_t* valid_src = nullptr; int64_t valid_off = 0;
      const int w = (ty.id == T_UTF8 || ty.id == T_BOOL) ? 0 : type_width(ty);
      auto utf8_off = [&](const ArrowArray* arr, int64_t i) -> int64_t { return large ? ((const int64_t*)arr->buffers[1])[i] : (int64_t)((const int32_t*)arr->buffers[1])[i]; };
      if (idx) {
        const std::string kf = idx_f->format ? idx_f->format : "";
        auto key_at = [&](int64_t i) -> int64_t {
          const void* kb = idx->buffers[1];
          if (kf == "c") return ((const int8_t*)kb)[i]; if (kf == "C") return ((const uint8_t*)kb)[i]; if (kf == "s") return ((const int16_t*)kb)[i];
          if (kf == "S") return ((const uint16_t*)kb)[i]; if (kf == "i") return ((const int32_t*)kb)[i]; if (kf == "I") return ((const uint32_t*)kb)[i];
          if (kf == "l") return ((const int64_t*)kb)[i]; if (kf == "L") return (int64_t)((const uint64_t*)kb)[i];
          throw Unsupported("dictionary index format '" + kf + "'");
        };
        const uint8_t* kval = (idx->null_count != 0 && idx->n_buffers > 0) ? (const uint8_t*)idx->buffers[0] : nullptr;
        const uint8_t* dval = (a->null_count != 0 && a->n_buffers > 0) ? (const uint8_t*)a->buffers[0] : nullptr;
        const int64_t doff = a->offset, dn = a->length;
        v_valid.assign((size_t)(n + 7) / 8 + 8, 0); bool any_null = false;
        if (ty.id == T_UTF8) v_offs.assign((size_t)n + 1, 0); else if (ty.id == T_BOOL) v_data.assign((size_t)(n + 7) / 8 + 8, 0); else v_data.assign((size_t)n * (size_t)w + 16, 0);
        for (int64_t i = 0; i < n; ++i) {
          bool ok = !kval || ((kval[(off + i) >> 3] >> ((off + i) & 7)) & 1);
          int64_t k = 0;
          if (ok) { k = key_at(off + i); if (k < 0 || k >= dn) throw std::runtime_error("dictionary index out of range"); ok = !dval || ((dval[(doff + k) >> 3] >> ((doff + k) & 7)) & 1); }
          if (ok) v_valid[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7)); else any_null = true;
          if (ty.id == T_UTF8) {
            const int64_t len = ok ? utf8_off(a, doff + k + 1) - utf8_off(a, doff + k) : 0;
            if (len > 0) { const uint8_t* src = (const uint8_t*)a->buffers[2] + utf8_off(a, doff + k); v_data.insert(v_data.end(), src, src + len); }
            v_offs[(size_t)i + 1] = (int64_t)v_data.size();
          } else if (ty.id == T_BOOL) { if (ok && ((((const uint8_t*)a->buffers[1])[(doff + k) >> 3] >> ((doff + k) & 7)) & 1)) v_data[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7)); }
          else if (ok) std::memcpy(v_data.data() + (size_t)i * (size_t)w, (const char*)a->buffers[1] + (size_t)(doff + k) * (size_t)w, (size_t)w);
        }
        if (any_null) { valid_src = v_valid.data(); valid_off = 0; }
      } else if (a->null_count != 0 && a->n_buffers > 0 && a->buffers[0]) { valid_src = (const uint8_t*)a->buffers[0]; valid_off = off; }
      if (valid_src) {
        std::vector<uint8_t> bits = realign_bits(valid_src, valid_off, n);
        ic->validity.ensure(bits.size()); g_staging.h2d(s, ic->validity.p, bits.data(), bits.size()); g_staging.drain();
        ic->col.validity = (const uint8_t*)ic->validity.p;
      }
      if (ty.id == T_UTF8) {
        // 32-bit offsets on the device, starting at 0 (LargeUtf8 and dictionary values are re-based here)
        std::vector<int32_t> o32((size_t)n + 1);
        const int64_t base = idx ? 0 : utf8_off(a, off);
        const int64_t last = idx ? v_offs[(size_t)n] : (n > 0 ? utf8_off(a, off + n) - base : 0);
        if (last > 0x7FFFFFFFll) throw Unsupported("a Utf8 column of more than 2^31 bytes (split the batch)");
        if (last < 0) throw std::runtime_error("Utf8 offsets of column '" + std::string(nm ? nm : "") + "' decrease");
        int64_t prev = 0;
        for (int64_t i = 0; i <= n; ++i) {      // (the offsets are the producer's: a decreasing or negative one must not become a copy length)
          const int64_t o = idx ? v_offs[(size_t)i] : utf8_off(a, off + i) - base;
          if (o < prev || o > last) throw std::runtime_error("Utf8 offsets of column '" + std::string(nm ? nm : "") + "' are not non-decreasing");
          o32[(size_t)i] = (int32_t)o; prev = o;
        }
        ic->offsets.ensure((size_t)(n + 1) * 4 + 16); g_staging.h2d(s, ic->offsets.p, o32.data(), (size_t)(n + 1) * 4); g_staging.drain();
        ic->data.ensure((size_t)last + 16);
        if (last > 0) { g_staging.h2d(s, ic->data.p, idx ? (const void*)v_data.data() : (const void*)((const char*)a->buffers[2] + base), (size_t)last); g_staging.drain(); }
        ic->col.offsets = (const int32_t*)ic->offsets.p;
      } else if (ty.id == T_BOOL) {
        std::vector<uint8_t> bits = idx ? v_data : realign_bits((const uint8_t*)a->buffers[1], off, n);
        ic->data.ensure(bits.size()); g_staging.h2d(s, ic->data.p, bits.data(), bits.size()); g_staging.drain();
      } else {
        ic->data.ensure((size_t)n * w + 16);
        if (n > 0) { g_staging.h2d(s, ic->data.p, idx ? (const char*)v_data.data() : (const char*)a->buffers[1] + (size_t)off * w, (size_t)n * w); if (idx) g_staging.drain(); }
      }
      ic->col.data = ic->data.p;
      t->cols.push_back(std::move(ic));
    }
    g_staging.drain();
    *out = t;
  });
  if (rc != GPUQ_OK) { delete t; if (out) *out = nullptr; }
  return rc;
}
int64_t gpuq_table_num_rows(const gpuq_table* t) { return t ? t->n_rows : 0; }
int gpuq_table_num_columns(const gpuq_table* t) { return t ? (int)t->cols.size() : 0; }
int gpuq_table_column(const gpuq_table* t, int i, gpuq_column* col_out, gpuq_field_info* field_out) {
  if (!t || i < 0 || i >= (int)t->cols.size()) return GPUQ_ERR_INVALID;
  if (col_out) *col_out = t->cols[i]->col;
  if (field_out) *field_out = t->cols[i]->field;
  return GPUQ_OK;
}
void gpuq_table_free(gpuq_table* t) { delete t; }

namespace {
struct ExportPriv { std::vector<std::vector<uint8_t>> bufs; std::vector<const void*> ptrs; std::vector<ArrowArray> kids; std::vector<ArrowArray*> kid_ptrs;
                    std::vector<std::vector<const void*>> kid_bufs; };
struct ExportSchemaPriv { std::vector<std::string> strs; std::vector<ArrowSchema> kids; std::vector<ArrowSchema*> kid_ptrs; };
void release_array(ArrowArray* a) { if (a && a->release) { delete (ExportPriv*)a->private_data; a->release = nullptr; } }
void release_child(ArrowArray* a) { if (a) a->release = nullptr; }
void release_schema(ArrowSchema* s) { if (s && s->release) { delete (ExportSchemaPriv*)s->private_data; s->release = nullptr; } }
void release_schema_child(ArrowSchema* s) { if (s) s->release = nullptr; }
}  // namespace

int gpuq_export_arrow(gpuq_ctx* ctx, void* stream, const gpuq_column* cols, const gpuq_field_info* fields, int n_cols, int64_t n_rows,
                      struct ArrowArray* out, struct ArrowSchema* out_schema) {
  return guarded(ctx, [&]() {
    check_ctx(ctx);
    if (!cols || !fields || !out || !out_schema || n_cols < 0 || n_rows < 0) throw std::runtime_error("bad arguments");
    hipStream_t s = use_stream(stream);
    auto* P = new ExportPriv(); auto* S = new ExportSchemaPriv();
    std::unique_ptr<ExportPriv> gp(P); std::unique_ptr<ExportSchemaPriv> gs(S);
    P->kids.resize(n_cols); P->kid_bufs.resize(n_cols); S->kids.resize(n_cols); S->strs.reserve((size_t)n_cols * 2 + 2);
    auto fetch = [&](const void* dev, size_t bytes) -> const void* {
      P->bufs.emplace_back(bytes + 8, 0);
      if (bytes) HIPCHECK(hipMemcpyAsync(P->bufs.back().data(), dev, bytes, hipMemcpyDeviceToHost, s));
      return P->bufs.back().data();
    };
    for (int c = 0; c < n_cols; ++c) {
      const gpuq_column& col = cols[c]; const gpuq_field_info& f = fields[c];
      ArrowArray& a = P->kids[c]; std::memset(&a, 0, sizeof(a));
      a.length = n_rows; a.null_count = col.validity ? -1 : 0; a.release = release_child;
      std::vector<const void*>& b = P->kid_bufs[c];
      b.push_back(col.validity ? fetch(col.validity, (size_t)(n_rows + 7) / 8) : nullptr);
      DType ty; ty.id = f.type; ty.p = f.precision; ty.s = f.scale;
      if (f.type == T_UTF8) {
        if (col.repr == GPUQ_REPR_PACKED15) {
          DevBuf offs, data; offs.ensure((size_t)(n_rows + 1) * 4 + 16); data.ensure((size_t)n_rows * 15 + 16);
          int64_t dl = 0;
          int rc2 = gpuq_unpack_utf8(ctx, stream, col.data, n_rows, (int32_t*)offs.p, (uint8_t*)data.p, n_rows * 15 + 16, &dl);
          if (rc2 != GPUQ_OK) throw std::runtime_error(g_last_error);
          b.push_back(fetch(offs.p, (size_t)(n_rows + 1) * 4)); b.push_back(fetch(data.p, (size_t)dl));
          HIPCHECK(hipStreamSynchronize(s));
        } else {
          int32_t last = 0;
          if (n_rows > 0) { HIPCHECK(hipMemcpyAsync(&last, col.offsets + n_rows, 4, hipMemcpyDeviceToHost, s)); HIPCHECK(hipStreamSynchronize(s)); }
          b.push_back(fetch(col.offsets, (size_t)(n_rows + 1) * 4)); b.push_back(fetch(col.data, (size_t)last));
        }
      } else if (f.type == T_BOOL) b.push_back(fetch(col.data, (size_t)(n_rows + 7) / 8));
      else b.push_back(fetch(col.data, (size_t)n_rows * type_width(ty)));
      a.n_buffers = (int64_t)b.size(); a.buffers = b.data();
      ArrowSchema& sc = S->kids[c]; std::memset(&sc, 0, sizeof(sc));
      S->strs.push_back(format_of(f)); sc.format = S->strs.back().c_str();
      S->strs.push_back(f.name); sc.name = S->strs.back().c_str();
      sc.flags = f.nullable ? 2 : 0; sc.release = release_schema_child;
    }
    HIPCHECK(hipStreamSynchronize(s));
    for (auto& k : P->kids) P->kid_ptrs.push_back(&k);
    for (auto& k : S->kids) S->kid_ptrs.push_back(&k);
    std::memset(out, 0, sizeof(*out));
    P->ptrs.push_back(nullptr);
    out->length = n_rows; out->n_buffers = 1; out->buffers = P->ptrs.data(); out->n_children = n_cols; out->children = P->kid_ptrs.data();
    out->release = release_array; out->private_data = gp.release();
    std::memset(out_schema, 0, sizeof(*out_schema));
    out_schema->format = "+s"; out_schema->name = ""; out_schema->n_children = n_cols; out_schema->children = S->kid_ptrs.data();
    out_schema->release = release_schema; out_schema->private_data = gs.release();
  });
}
}  // extern "C"

extern "C" {
// ---------------------------------------------------------------- JIT control / introspection
int gpuq_ctx_set_jit(gpuq_ctx* ctx, const char* mode, int64_t min_rows) {
  return guarded(ctx, [&]() {
    if (!ctx || !mode) throw std::runtime_error("ctx/mode is NULL");
    const std::string m = mode;
    if (m == "off") ctx->jit_mode = 0; else if (m == "auto" || m == "wait") ctx->jit_mode = jit_available() ? 1 : 0; else if (m == "force") ctx->jit_mode = 2;
    else throw std::runtime_error("jit mode must be off|auto|wait|force");
    ctx->jit_wait = m == "wait";
    if (min_rows >= 0) ctx->jit_min_rows = min_rows;
  });
}
int gpuq_ctx_set_option(gpuq_ctx* ctx, const char* key, const char* value) {
  return guarded(ctx, [&]() {
    if (!ctx || !key || !value) throw std::runtime_error("ctx/key/value is NULL");
    const std::string k = key;
    if (k == "join_dense") ctx->join_dense = std::atoi(value) != 0;
    else if (k == "join_radix") { const std::string v = value; ctx->join_radix = v == "off" || v == "0" ? 0 : (v == "force" || v == "2" ? 2 : 1); }
    else if (k == "join_radix_slice_log2") { const int b = std::atoi(value); if (b < 10 || b > 26) throw std::runtime_error("join_radix_slice_log2 must be in [10, 26]"); ctx->join_radix_slice_log2 = b; }
    else if (k == "join_dense_ratio") { const long long r = std::atoll(value); if (r < 1) throw std::runtime_error("join_dense_ratio must be >= 1"); ctx->join_dense_ratio = r; }
    else throw std::runtime_error("unknown option '" + k + "'");
  });
}
int gpuq_ctx_jit_wait(gpuq_ctx* ctx) { return guarded(ctx, [&]() { check_ctx(ctx); jit_drain(); }); }
void gpuq_jit_quiesce(void) { try { jit_drain(); } catch (...) {} }
int gpuq_ctx_jit_stats(gpuq_ctx* ctx, int* available, int* launches, char* last_error, size_t cap) {
  if (!ctx) return GPUQ_ERR_INVALID;
  if (available) *available = jit_available() ? 1 : 0;
  if (launches) *launches = ctx->jit_launches;
  if (last_error && cap) std::snprintf(last_error, cap, "%s", ctx->last_jit_error.c_str());
  return GPUQ_OK;
}
int gpuq_jit_cache_stats(int* disk_hits, int* compiles) { jit_cache_stats(disk_hits, compiles); return GPUQ_OK; }
int gpuq_op_jit_source(gpuq_op* op, int kernel_id, char* buf, size_t cap) {
  if (!op) return GPUQ_ERR_INVALID;
  return guarded(op->ctx, [&]() {
    const std::string src = jit_full_source(op->prog.jit_src, kernel_id);
    if (buf && cap) std::snprintf(buf, cap, "%s", src.c_str());
    if (src.size() + 1 > cap) throw Capacity("source needs " + std::to_string(src.size() + 1) + " bytes");
  });
}

// ---------------------------------------------------------------- timers
int gpuq_timer_create(gpuq_ctx* ctx, gpuq_timer** out) {
  return guarded(ctx, [&]() { check_ctx(ctx); auto* t = new gpuq_timer(); HIPCHECK(hipEventCreate(&t->a)); HIPCHECK(hipEventCreate(&t->b)); *out = t; });
}
int gpuq_timer_start(gpuq_timer* t, void* stream) { return guarded(nullptr, [&]() { HIPCHECK(hipEventRecord(t->a, use_stream(stream))); }); }
int gpuq_timer_stop(gpuq_timer* t, void* stream) { return guarded(nullptr, [&]() { HIPCHECK(hipEventRecord(t->b, use_stream(stream))); }); }
int gpuq_timer_elapsed_ms(gpuq_timer* t, float* ms_out) {
  return guarded(nullptr, [&]() { HIPCHECK(hipEventSynchronize(t->b)); HIPCHECK(hipEventElapsedTime(ms_out, t->a, t->b)); });
}
void gpuq_timer_free(gpuq_timer* t) { if (t) { if (t->a) (void)hipEventDestroy(t->a); if (t->b) (void)hipEventDestroy(t->b); delete t; } }

}  // extern "C"
