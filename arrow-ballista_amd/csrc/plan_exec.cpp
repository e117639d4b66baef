// Native plan executor: the host side of the engine above the operator ABI, in C++.
//
// What it stands in for in the reference: the walk an ExecutionEngine makes over the stage plan it is
// handed (ballista/executor/src/execution_engine.rs:34-60: create_query_stage_exec(plan) ->
// QueryStageExecutor::execute_query_stage) -- DataFusion's ExecutionPlan::execute() calls down the tree
// (shuffle_writer.rs:255 `plan.execute(0, ctx)`).  PNode names, fields and modes mirror the protobuf
// (ballista/core/proto/datafusion.proto: FilterExecNode :1291, ProjectionExecNode :1399, AggregateExecNode
// :1405, HashJoinExecNode :1346, SortExecNode :1465, CoalesceBatchesExecNode :1487, GlobalLimit/LocalLimit
// :1453-1463, UnionExecNode :1319, CoalescePartitionsExecNode :1492), carried as JSON.
//
// This file is a CLIENT of the C ABI in include/gpuq.h (operators are created from descriptors and run
// through gpuq_*_run exactly as a Rust shim would) plus the planning that keeps data on the device between
// operators: Filter -> Projection -> consumer chains are fused into the consumer's descriptor, filters /
// joins / sorts produce index vectors and downstream operators read columns through them (late
// materialisation), outputs of one operator live in one pooled device allocation.
// The Python classes of the same names in arrow-ballista_amd/plan.py are the test-side mirror of this logic.
#include "../../include/gpuq.h"
#include "devbuf.h"
#include "expr_compile.h"
#include "json.h"
#include <algorithm>
#include <atomic>
#include <thread>
#include <cerrno>
#include <chrono>
#include <random>
#include <sys/stat.h>
#include <functional>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

using namespace gpuq;

namespace {

constexpr uint32_t NULL_ROW_ID = 0xFFFFFFFFu;
typedef std::shared_ptr<DevBuf> BufP;

// ---------------------------------------------------------------- JSON building
Json jstr(const std::string& s) { Json j; j.kind = Json::STR; j.s = s; return j; }
Json jnum(long long v) { Json j; j.kind = Json::NUM; j.s = std::to_string(v); return j; }
Json jbool(bool b) { Json j; j.kind = Json::BOOL; j.b = b; return j; }
Json jarr(std::vector<Json> a = {}) { Json j; j.kind = Json::ARR; j.a = std::move(a); return j; }
Json jobj(std::vector<std::pair<std::string, Json>> o = {}) { Json j; j.kind = Json::OBJ; j.o = std::move(o); return j; }
Json jcol(const std::string& name, int index) { return jobj({{"column", jobj({{"name", jstr(name)}, {"index", jnum(index)}})}}); }
Json jand(const Json& l, const Json& r) { return jobj({{"binary_expr", jobj({{"l", l}, {"r", r}, {"op", jstr("AND")}})}}); }

// ---------------------------------------------------------------- schema / tables
struct PField { std::string name; Json type; bool nullable = true; };
typedef std::vector<PField> PSchema;

int type_id_of(const Json& t, int& p, int& s) { DType d = dtype_from_json(t); p = d.p; s = d.s; return d.id; }
Json type_json_of(int tid, int p, int s) {
  switch (tid) {
    case T_BOOL: return jstr("Boolean"); case T_INT32: return jstr("Int32"); case T_INT64: return jstr("Int64"); case T_DATE32: return jstr("Date32");
    case T_FLOAT64: return jstr("Float64"); case T_UTF8: return jstr("Utf8"); case T_UINT32: return jstr("UInt32"); case T_UINT64: return jstr("UInt64");
    case T_INT8: return jstr("Int8"); case T_INT16: return jstr("Int16"); case T_UINT8: return jstr("UInt8"); case T_UINT16: return jstr("UInt16");
    case T_FLOAT32: return jstr("Float32"); case T_DATE64: return jstr("Date64");
    case T_DECIMAL128: return jobj({{"Decimal128", jarr({jnum(p), jnum(s)})}});
    case T_TIMESTAMP: { static const char* u[4] = {"Second", "Millisecond", "Microsecond", "Nanosecond"}; return jobj({{"Timestamp", jarr({jstr(u[p & 3]), Json()})}}); }
  }
  throw std::runtime_error("plan: type id " + std::to_string(tid) + " has no name");
}

struct PCol { std::string name; Json type; bool nullable = true; gpuq_column c{}; };
struct PTable {
  std::vector<PCol> cols; int64_t n = 0;
  std::vector<const uint32_t*> via; std::vector<int> sides; bool dense = false;
  std::vector<BufP> keep;          // owners of every buffer the table points into
  int64_t record_cap = 0;          // > 0: all columns live in keep[0], laid out for this row capacity (alloc_outputs)
  // Deferred execution (include/gpuq.h): n_dev != nullptr means `n` is only an upper BOUND of the row count and the actual count is
  // the device u64 *n_dev (a join's pair count, a filter's survivors, an aggregate's groups), owned by n_keep.  Operators that
  // can run deferred pass both on (gpuq_input.n_rows_dev); everything else calls resolve() first, which settles the plan's
  // pending operators and makes n exact.
  const uint64_t* n_dev = nullptr; BufP n_keep;
  bool is_view() const { return !via.empty(); }
  void own(const PTable& o) { keep.insert(keep.end(), o.keep.begin(), o.keep.end()); }
  void count_from(const PTable& o) { n_dev = o.n_dev; n_keep = o.n_keep; }
};

// field list of a table as an operator sees it (side / raw128 / dense annotations)
Json table_fields(const PTable& t) {
  Json f = jarr();
  for (size_t i = 0; i < t.cols.size(); ++i) {
    const PCol& c = t.cols[i];
    std::vector<std::pair<std::string, Json>> o = {{"name", jstr(c.name)}, {"type", c.type}, {"nullable", jbool(c.nullable)}, {"side", jnum(t.sides[i])}};
    if (c.c.repr == GPUQ_REPR_PACKED15) o.push_back({"raw128", jnum(1)});
    if (t.sides[i] > 0 && t.dense) o.push_back({"dense", jnum(1)});
    f.a.push_back(jobj(o));
  }
  return f;
}
PSchema plain_schema(const PTable& t) {
  PSchema s;
  for (size_t i = 0; i < t.cols.size(); ++i) s.push_back({t.cols[i].name, t.cols[i].type, t.cols[i].nullable || (t.sides[i] > 0 && !t.dense)});
  return s;
}
std::string table_sig(const PTable& t) {
  std::string r = std::to_string(t.via.size()) + (t.dense ? "d" : "s");
  for (size_t i = 0; i < t.cols.size(); ++i)
    r += "|" + t.cols[i].name + ":" + t.cols[i].type.dump() + (t.cols[i].nullable ? "?" : "") + std::to_string(t.sides[i]) + "r" + std::to_string(t.cols[i].c.repr);
  return r;
}

// ---------------------------------------------------------------- expression trees (PhysicalExprNode mirror)
Json rewrite_columns(const Json& e, const std::function<Json(const Json&)>& fn) {
  if (e.is_obj()) {
    if (e.o.size() == 1 && e.o[0].first == "column" && e.o[0].second.is_obj() && e.o[0].second.find("name")) return fn(e.o[0].second);
    Json r = jobj();
    for (auto& kv : e.o) r.o.emplace_back(kv.first, rewrite_columns(kv.second, fn));
    return r;
  }
  if (e.is_arr()) { Json r = jarr(); for (auto& v : e.a) r.a.push_back(rewrite_columns(v, fn)); return r; }
  return e;
}
typedef std::map<std::string, Json> ColMap;      // output column of a ProjectionExec -> expression over its input
Json inline_projection(const Json& e, const ColMap* m) {
  if (!m) return e;
  return rewrite_columns(e, [&](const Json& c) { auto it = m->find(c.at("name").str()); return it != m->end() ? it->second : jobj({{"column", c}}); });
}
template <class Names> Json rebind(const Json& e, const Names& names) {
  return rewrite_columns(e, [&](const Json& c) {
    const std::string& n = c.at("name").str();
    for (size_t i = 0; i < names.size(); ++i) if (names[i] == n) return jcol(n, (int)i);
    throw std::runtime_error("plan: column '" + n + "' not found in the input schema");
  });
}
std::vector<std::string> names_of(const PTable& t) { std::vector<std::string> v; for (auto& c : t.cols) v.push_back(c.name); return v; }
bool has_like(const Json& e) {
  if (e.is_obj()) { for (auto& kv : e.o) if (kv.first == "like_expr" || has_like(kv.second)) return true; return false; }
  if (e.is_arr()) { for (auto& v : e.a) if (has_like(v)) return true; }
  return false;
}
// `column = 'literal'` / `column != 'literal'` with a Utf8 literal beyond the 15 bytes a register holds (q19's l_shipinstruct =
// 'DELIVER IN PERSON'): equality with a literal is LIKE without wildcards, and LIKE runs over the Arrow-layout bytes at any length
// (gpuq_like_utf8) -- so the comparison becomes a like_expr whose pattern is the literal with its % and _ escaped.  NULL semantics
// are the same (NULL operand -> NULL).  Applied to the whole plan when it is created; shorter literals are left alone.
Json lower_long_string_eq(const Json& e) {
  if (e.is_obj()) {
    if (e.o.size() == 1 && e.o[0].first == "binary_expr" && e.o[0].second.is_obj()) {
      const Json& b = e.o[0].second;
      const std::string op = b.get_str("op", "");
      if ((op == "=" || op == "!=" || op == "Eq" || op == "NotEq") && b.find("l") && b.find("r")) {
        auto is_col = [](const Json& v) { return v.is_obj() && v.o.size() == 1 && v.o[0].first == "column"; };
        auto long_lit = [](const Json& v) {
          if (!(v.is_obj() && v.o.size() == 1 && v.o[0].first == "literal" && v.o[0].second.is_obj())) return false;
          const Json& l = v.o[0].second;
          return l.get_str("type", "") == "Utf8" && l.find("value") && !l.at("value").is_null() && l.at("value").str().size() > 15;
        };
        const Json* c = nullptr; const Json* l = nullptr;
        if (is_col(b.at("l")) && long_lit(b.at("r"))) { c = &b.at("l"); l = &b.at("r"); }
        else if (is_col(b.at("r")) && long_lit(b.at("l"))) { c = &b.at("r"); l = &b.at("l"); }
        if (c) {
          std::string pat;
          for (char ch : l->o[0].second.at("value").str()) { if (ch == '%' || ch == '_') pat += '\\'; pat += ch; }
          return jobj({{"like_expr", jobj({{"negated", jbool(op == "!=" || op == "NotEq")}, {"case_insensitive", jbool(false)}, {"expr", *c},
                                           {"pattern", jobj({{"literal", jobj({{"type", jstr("Utf8")}, {"value", jstr(pat)}})}})}})}});
        }
      }
    }
    Json r = jobj();
    for (auto& kv : e.o) r.o.emplace_back(kv.first, lower_long_string_eq(kv.second));
    return r;
  }
  if (e.is_arr()) { Json r = jarr(); for (auto& v : e.a) r.a.push_back(lower_long_string_eq(v)); return r; }
  return e;
}
Json rewrite_like(const Json& e, const std::function<Json(const Json&)>& fn) {
  if (e.is_obj()) {
    if (e.o.size() == 1 && e.o[0].first == "like_expr") return fn(e.o[0].second);
    Json r = jobj();
    for (auto& kv : e.o) r.o.emplace_back(kv.first, rewrite_like(kv.second, fn));
    return r;
  }
  if (e.is_arr()) { Json r = jarr(); for (auto& v : e.a) r.a.push_back(rewrite_like(v, fn)); return r; }
  return e;
}

// ---------------------------------------------------------------- execution context
struct Exec {
  gpuq_ctx* ctx = nullptr; void* stream = nullptr;
  std::map<std::string, gpuq_op*>* ops = nullptr;        // compiled operators, keyed by descriptor text (owned by the plan)
  const gpuq_input* inputs = nullptr; int n_inputs = 0;
  uint64_t* pin = nullptr;                                // pinned host words for count read-backs
  std::map<std::string, gpuq_op*>* memo = nullptr;       // (call site, input layout) -> operator: skips rebuilding the descriptor
  gpuq_comm* comm = nullptr;                              // ranks of the node (gpuq_plan_set_comm); nullptr = a single-GPU plan
  const std::atomic<int>* cancel = nullptr;               // set by gpuq_task_cancel: checked between operator calls
  // Deferred execution: from a plan's second run on, operators that remember a completed synchronous run queue their kernels
  // without reading anything back; row counts travel as device words (PTable::n_dev) and everything is settled in ONE host round
  // trip (settle(): at the plan's end, or earlier where a node needs an exact count -- an exchange, a concatenation, a file).
  bool deferred = false;
  std::vector<gpuq_op*> pending;                          // operators with deferred runs since the last settle
  struct CountFix { const uint64_t* word; int64_t bound; int64_t* rows; BufP keep; };
  std::vector<CountFix> fixes;                            // output_rows metrics booked with a bound: corrected at the settle
  int settles = 0, host_syncs = 0;                        // per execution: settles, and count / status read-backs of the synchronous paths
};
struct Cancelled : std::runtime_error { using std::runtime_error::runtime_error; };
struct DeferredRetry : std::runtime_error { using std::runtime_error::runtime_error; };      // an assumption of a deferred run did not hold: the plan runs again, synchronously
// ... and every rank of the node already knows (it came out of a collective, or was announced to the peers): not to be announced again
struct AgreedRetry : DeferredRetry { using DeferredRetry::DeferredRetry; };
struct AgreedFailure : std::runtime_error { int rc; AgreedFailure(const std::string& m, int rc_ = GPUQ_ERR_PEER) : std::runtime_error(m), rc(rc_) {} };
inline void check_cancel(const Exec& x) { if (x.cancel && x.cancel->load(std::memory_order_relaxed)) throw Cancelled("task cancelled"); }

void check(Exec& x, int rc) {
  check_cancel(x);      // every operator call passes through here: the granularity of cancellation is one operator
  if (rc == GPUQ_OK) return;
  const char* m = gpuq_last_error(x.ctx);
  const std::string msg = m ? m : "gpuq error";
  if (rc == GPUQ_ERR_UNSUPPORTED) throw Unsupported(msg);
  if (rc == GPUQ_ERR_CAPACITY) throw Capacity(msg);
  if (rc == GPUQ_ERR_HIP) throw HipError(msg);
  throw std::runtime_error(msg);
}
gpuq_op* get_op(Exec& x, const Json& desc) {
  const std::string key = desc.dump();
  auto it = x.ops->find(key);
  if (it != x.ops->end()) return it->second;
  gpuq_op* op = nullptr;
  check(x, gpuq_op_create(x.ctx, key.c_str(), &op));
  (*x.ops)[key] = op;
  return op;
}
// Operator for a call site and an input layout.  Building and serialising a descriptor costs tens of microseconds; a plan
// that runs repeatedly (or over many partitions) sees the same layouts again, so the descriptor is only built on a miss.
template <class MakeDesc> gpuq_op* cached_op(Exec& x, const void* site, int tag, const std::string& sig, MakeDesc&& mk) {
  std::string key = std::to_string((uintptr_t)site); key += '#'; key += std::to_string(tag); key += '#'; key += sig;
  auto it = x.memo->find(key);
  if (it != x.memo->end()) return it->second;
  gpuq_op* op = get_op(x, mk());
  (*x.memo)[key] = op;
  return op;
}
uint64_t read_u64(Exec& x, const void* dev) {
  HIPCHECK(hipMemcpyAsync(x.pin, dev, 8, hipMemcpyDeviceToHost, (hipStream_t)x.stream));
  HIPCHECK(hipStreamSynchronize((hipStream_t)x.stream));
  ++x.host_syncs;
  return x.pin[0];
}
// ---------------------------------------------------------------- deferred execution
// May `op` run deferred in this execution?  Marks it and books it for the next settle.
void settle(Exec& x, struct PTable* t);
// the operator runs synchronously: what deferred runs of it may have left in its status word (a synchronous run resets it) is looked at first
void use_sync(Exec& x, gpuq_op* op) {
  if (std::find(x.pending.begin(), x.pending.end(), op) != x.pending.end()) settle(x, nullptr);
  gpuq_op_set_deferred(op, 0);
}
bool use_deferred(Exec& x, gpuq_op* op) {
  if (!x.deferred || !gpuq_op_can_defer(op)) { use_sync(x, op); return false; }
  // (one operator may run deferred several times before a settle -- the take / materialise projections do: status bits are sticky,
  // workspaces are reused in stream order, and counts are copied out of the operator right after each run)
  gpuq_op_set_deferred(op, 1);
  if (std::find(x.pending.begin(), x.pending.end(), op) == x.pending.end()) x.pending.push_back(op);
  return true;
}
BufP dev_alloc(size_t bytes) { BufP b = std::make_shared<DevBuf>(); b->ensure(bytes ? bytes : 16); return b; }

// ONE host round trip for everything deferred so far: the status words of the pending operators, the count behind `t` (made exact)
// and the counts behind the metrics booked with a bound.
void settle(Exec& x, PTable* t) {
  if (x.pending.empty() && x.fixes.empty() && !(t && t->n_dev)) return;
  std::vector<const uint64_t*> words; std::vector<uint64_t> vals;
  for (auto& f : x.fixes) words.push_back(f.word);
  if (t && t->n_dev) words.push_back(t->n_dev);
  vals.assign(words.size() + 1, 0);
  const int rc = gpuq_ops_settle(x.ctx, x.stream, x.pending.data(), (int)x.pending.size(), words.data(), (int)words.size(), vals.data());
  ++x.settles;
  for (gpuq_op* op : x.pending) gpuq_op_set_deferred(op, 0);
  x.pending.clear();
  if (rc == GPUQ_ERR_RETRY) { x.fixes.clear(); throw DeferredRetry("deferred run did not hold"); }
  check(x, rc);
  for (size_t i = 0; i < x.fixes.size(); ++i) { const int64_t actual = std::min<int64_t>((int64_t)vals[i], x.fixes[i].bound); *x.fixes[i].rows += actual - x.fixes[i].bound; }
  if (t && t->n_dev) {
    const int64_t actual = std::min<int64_t>((int64_t)vals[x.fixes.size()], t->n);
    // plain (non-view) tables carry their length in every column too
    if (!t->is_view()) for (auto& c : t->cols) if (c.c.length == t->n) c.c.length = actual;
    t->n = actual; t->n_dev = nullptr; t->n_keep = nullptr;
  }
  x.fixes.clear();
}
// make t.n exact (a node that concatenates, slices, writes files, exchanges or hands the table to the caller)
inline void resolve(Exec& x, PTable& t) { if (t.n_dev) settle(x, &t); }
// does `op` run deferred over `t`?  When it does not, t's count is made exact first (synchronous entry points need it)
inline bool prep(Exec& x, gpuq_op* op, PTable& t) { const bool d = use_deferred(x, op); if (!d) resolve(x, t); return d; }

struct InputC { gpuq_input in{}; std::vector<gpuq_column> cols; };
void make_input(const PTable& t, InputC& ic) {
  ic.cols.clear();
  for (auto& c : t.cols) ic.cols.push_back(c.c);
  ic.in.cols = ic.cols.data(); ic.in.n_cols = (int)ic.cols.size(); ic.in.n_via = (int)t.via.size(); ic.in.n_rows = t.n; ic.in.n_rows_dev = t.n_dev;
  for (int k = 0; k < 3; ++k) ic.in.via[k] = k < (int)t.via.size() ? t.via[k] : nullptr;
}

// one pooled allocation for all output columns of an operator (same layout as table.record_layout in the Python mirror)
PTable alloc_outputs(gpuq_op* op, int64_t n, std::vector<gpuq_column>& carr) {
  const int nf = gpuq_op_num_outputs(op);
  std::vector<gpuq_field_info> f((size_t)nf);
  for (int i = 0; i < nf; ++i) if (gpuq_op_output_field(op, i, &f[(size_t)i]) != GPUQ_OK) throw std::runtime_error("plan: output field query failed");
  const size_t bm = (size_t)((n + 63) / 64) * 8 + 8;
  std::vector<size_t> doff((size_t)nf), voff((size_t)nf);
  size_t off = 256;
  for (int i = 0; i < nf; ++i) {
    const size_t dbytes = f[(size_t)i].type == T_BOOL ? bm : (size_t)std::max<int64_t>(n, 1) * (size_t)f[(size_t)i].width + 16;
    doff[(size_t)i] = off; off += (dbytes + 255) & ~(size_t)255;
    voff[(size_t)i] = 0;
    if (f[(size_t)i].nullable) { voff[(size_t)i] = off; off += (bm + 255) & ~(size_t)255; }
  }
  BufP buf = dev_alloc(off);
  PTable t; t.n = n; t.keep.push_back(buf); t.record_cap = std::max<int64_t>(n, 1);
  carr.assign((size_t)nf, gpuq_column{});
  for (int i = 0; i < nf; ++i) {
    PCol c; c.name = f[(size_t)i].name; c.type = type_json_of(f[(size_t)i].type, f[(size_t)i].precision, f[(size_t)i].scale); c.nullable = f[(size_t)i].nullable != 0;
    c.c.type = f[(size_t)i].type; c.c.precision = f[(size_t)i].precision; c.c.scale = f[(size_t)i].scale; c.c.repr = f[(size_t)i].repr;
    c.c.data = (char*)buf->p + doff[(size_t)i]; c.c.offsets = nullptr;
    c.c.validity = f[(size_t)i].nullable ? (const uint8_t*)buf->p + voff[(size_t)i] : nullptr;
    c.c.length = n;
    carr[(size_t)i] = c.c;
    t.cols.push_back(c); t.sides.push_back(0);
  }
  return t;
}

// LIKE runs as its own kernel over the Arrow-layout bytes (gpuq_like_utf8): every like_expr node becomes a reference to a Boolean
// column appended to a copy of the table.  The operand must be a column of `t`, the pattern a Utf8 literal.
PTable lower_like(Exec& x, const PTable& t, std::vector<Json>& exprs) {
  bool any = false; for (auto& e : exprs) any = any || has_like(e);
  if (!any) return t;
  struct Found { std::string col, pattern; bool negated, ci; };
  std::vector<Found> found;
  for (auto& e : exprs)
    e = rewrite_like(e, [&](const Json& v) {
      const Json& operand = v.at("expr"); const Json& pat = v.at("pattern");
      if (!(operand.is_obj() && operand.o.size() == 1 && operand.o[0].first == "column")) throw Unsupported("LIKE over a computed expression is not supported on device (operand must be a column)");
      if (!(pat.is_obj() && pat.find("literal") && pat.at("literal").get_str("type", "") == "Utf8" && pat.at("literal").find("value") && !pat.at("literal").at("value").is_null()))
        throw Unsupported("LIKE needs a non-NULL Utf8 literal pattern");
      found.push_back({operand.o[0].second.at("name").str(), pat.at("literal").at("value").str(), v.get_bool("negated", false), v.get_bool("case_insensitive", false)});
      return jobj({{"column", jobj({{"name", jstr("__like_" + std::to_string(found.size() - 1))}})}});
    });
  PTable out = t;
  hipStream_t s = (hipStream_t)x.stream;
  for (size_t k = 0; k < found.size(); ++k) {
    size_t i = 0; while (i < t.cols.size() && t.cols[i].name != found[k].col) ++i;
    if (i == t.cols.size()) throw std::runtime_error("plan: column '" + found[k].col + "' not found in the input schema");
    PCol c = t.cols[i]; const int sd = t.sides[i];
    if (c.c.type != T_UTF8) throw std::runtime_error("LIKE over a non-Utf8 column '" + c.name + "'");
    if (c.c.repr == GPUQ_REPR_PACKED15) {      // a string produced by an operator: back to offsets + bytes first
      const int64_t m = c.c.length;
      BufP offs = dev_alloc((size_t)(m + 4) * 4), data = dev_alloc((size_t)m * 15 + 16);
      int64_t dl = 0;
      check(x, gpuq_unpack_utf8(x.ctx, x.stream, c.c.data, m, (int32_t*)offs->p, (uint8_t*)data->p, m * 15 + 16, &dl));
      c.c.repr = GPUQ_REPR_ARROW; c.c.data = data->p; c.c.offsets = (const int32_t*)offs->p;
      out.keep.push_back(offs); out.keep.push_back(data);
    }
    const size_t nb = (size_t)((t.n + 63) / 64) * 8 + 8;
    const bool nullable = c.nullable || (sd > 0 && !t.dense);
    BufP bits = dev_alloc(nb), valid = nullable ? dev_alloc(nb) : nullptr;
    HIPCHECK(hipMemsetAsync(bits->p, 0, nb, s)); if (valid) HIPCHECK(hipMemsetAsync(valid->p, 0, nb, s));
    check(x, gpuq_like_utf8(x.ctx, x.stream, &c.c, sd > 0 ? t.via[(size_t)sd - 1] : nullptr, t.n, found[k].pattern.c_str(), found[k].negated, found[k].ci,
                            (uint8_t*)bits->p, valid ? (uint8_t*)valid->p : nullptr));
    PCol b; b.name = "__like_" + std::to_string(k); b.type = jstr("Boolean"); b.nullable = nullable;
    b.c.type = T_BOOL; b.c.repr = GPUQ_REPR_ARROW; b.c.data = bits->p; b.c.validity = valid ? (const uint8_t*)valid->p : nullptr; b.c.length = t.n;
    out.cols.push_back(b); out.sides.push_back(0);
    out.keep.push_back(bits); if (valid) out.keep.push_back(valid);
  }
  return out;
}

// Comparisons of Utf8 operands beyond what a register holds (15 bytes): `<` `<=` `>` `>=` between a column and a literal or another
// column, `=` / `!=` between two columns.  The register program refuses them at run time (a value longer than 15 bytes reached the
// comparison); the executor then runs the expression again with every such comparison lowered to a Boolean helper column computed over
// the Arrow-layout bytes (gpuq_utf8_compare) -- the same move LIKE makes.  Operands must be bare columns of `t` / Utf8 literals.
bool strcmp_operand(const PTable& t, const Json& e, int& col, std::string& lit) {
  col = -1;
  if (!(e.is_obj() && e.o.size() == 1)) return false;
  if (e.o[0].first == "column" && e.o[0].second.is_obj() && e.o[0].second.find("name")) {
    const std::string& n = e.o[0].second.at("name").str();
    for (size_t i = 0; i < t.cols.size(); ++i) if (t.cols[i].name == n) { if (t.cols[i].c.type == T_UTF8 && t.cols[i].c.repr == GPUQ_REPR_ARROW) { col = (int)i; return true; } return false; }
    return false;
  }
  if (e.o[0].first == "literal") {
    const Json& v = e.o[0].second;
    const Json* ty = v.find("type"); const Json* val = v.find("value");
    if (ty && ty->is_str() && (ty->s == "Utf8" || ty->s == "LargeUtf8") && val && !val->is_null()) { lit = val->str(); return true; }
  }
  return false;
}
PTable lower_strcmp(Exec& x, const PTable& t, std::vector<Json>& exprs) {
  struct Found { int a, b; std::string lit; int op; };
  std::vector<Found> found;
  std::function<Json(const Json&)> walk = [&](const Json& e) -> Json {
    if (e.is_obj()) {
      if (e.o.size() == 1 && e.o[0].first == "binary_expr") {
        const Json& v = e.o[0].second;
        static const std::map<std::string, int> ops = {{"=", 0}, {"Eq", 0}, {"==", 0}, {"!=", 1}, {"NotEq", 1}, {"<>", 1}, {"<", 2}, {"Lt", 2}, {"<=", 3}, {"LtEq", 3}, {">", 4}, {"Gt", 4}, {">=", 5}, {"GtEq", 5}};
        auto it = ops.find(v.get_str("op", ""));
        int ca = -1, cb = -1; std::string la, lb;
        if (it != ops.end() && strcmp_operand(t, v.at("l"), ca, la) && strcmp_operand(t, v.at("r"), cb, lb) && (ca >= 0 || cb >= 0)) {
          static const int mirror[6] = {0, 1, 4, 5, 2, 3};      // literal OP column  ==  column mirror(OP) literal
          Found f;
          if (ca >= 0) { f.a = ca; f.b = cb; f.lit = lb; f.op = it->second; } else { f.a = cb; f.b = -1; f.lit = la; f.op = mirror[it->second]; }
          found.push_back(f);
          return jobj({{"column", jobj({{"name", jstr("__cmp_" + std::to_string(found.size() - 1))}})}});
        }
      }
      Json r = jobj();
      for (auto& kv : e.o) r.o.emplace_back(kv.first, walk(kv.second));
      return r;
    }
    if (e.is_arr()) { Json r = jarr(); for (auto& v : e.a) r.a.push_back(walk(v)); return r; }
    return e;
  };
  for (auto& e : exprs) e = walk(e);
  if (found.empty()) return t;
  PTable out = t;
  hipStream_t s = (hipStream_t)x.stream;
  const size_t nb = (size_t)((t.n + 63) / 64) * 8 + 8;
  for (size_t k = 0; k < found.size(); ++k) {
    const Found& f = found[k];
    const PCol& a = t.cols[(size_t)f.a]; const int sa = t.sides[(size_t)f.a];
    const PCol* b = f.b >= 0 ? &t.cols[(size_t)f.b] : nullptr; const int sb = f.b >= 0 ? t.sides[(size_t)f.b] : 0;
    const bool nullable = a.nullable || (sa > 0 && !t.dense) || (b && (b->nullable || (sb > 0 && !t.dense)));
    BufP bits = dev_alloc(nb), valid = nullable ? dev_alloc(nb) : nullptr;
    HIPCHECK(hipMemsetAsync(bits->p, 0, nb, s)); if (valid) HIPCHECK(hipMemsetAsync(valid->p, 0, nb, s));
    check(x, gpuq_utf8_compare(x.ctx, x.stream, &a.c, sa > 0 ? t.via[(size_t)sa - 1] : nullptr, b ? &b->c : nullptr, sb > 0 ? t.via[(size_t)sb - 1] : nullptr,
                               b ? nullptr : f.lit.data(), b ? 0 : (int64_t)f.lit.size(), t.n, f.op, (uint8_t*)bits->p, valid ? (uint8_t*)valid->p : nullptr));
    PCol c; c.name = "__cmp_" + std::to_string(k); c.type = jstr("Boolean"); c.nullable = nullable;
    c.c.type = T_BOOL; c.c.repr = GPUQ_REPR_ARROW; c.c.data = bits->p; c.c.validity = valid ? (const uint8_t*)valid->p : nullptr; c.c.length = t.n;
    out.cols.push_back(c); out.sides.push_back(0);
    out.keep.push_back(bits); if (valid) out.keep.push_back(valid);
  }
  return out;
}

// ---------------------------------------------------------------- Utf8 keys longer than 15 bytes
// The expression programs carry a string as one 16-byte integer (<= 15 bytes).  When an aggregate or a join reports that a longer
// value reached a key, the executor runs the operator again over exact dictionary codes (gpuq_utf8_intern): a group / join key
// that is a plain Utf8 column in Arrow layout is replaced by an Int64 code column appended to (a copy of) the table; strings come
// back with a take through the codes.  Nothing is paid on the common path: the rewrite only happens after the loud failure.
bool is_long_string_failure(const std::exception& e) { return std::string(e.what()).find("longer than 15 bytes reached") != std::string::npos; }
// index of the column `e` names when `e` is a bare column reference to an Arrow-layout Utf8 column of `t`, else -1
int long_key_column(const PTable& t, const Json& e) {
  if (!(e.is_obj() && e.o.size() == 1 && e.o[0].first == "column" && e.o[0].second.is_obj() && e.o[0].second.find("name"))) return -1;
  const std::string& n = e.o[0].second.at("name").str();
  for (size_t i = 0; i < t.cols.size(); ++i)
    if (t.cols[i].name == n) return (t.cols[i].c.type == T_UTF8 && t.cols[i].c.repr == GPUQ_REPR_ARROW) ? (int)i : -1;
  return -1;
}
struct Utf8DictGuard { gpuq_utf8_dict* d = nullptr; ~Utf8DictGuard() { if (d) gpuq_utf8_dict_free(d); } };
// appends the code column of t.cols[ci] (insert: fills `dict` from it; else looks it up in `dict`) under `name`
void append_code_column(Exec& x, PTable& t, int ci, gpuq_utf8_dict* dict, bool insert, const std::string& name) {
  const PCol& c = t.cols[(size_t)ci]; const int sd = t.sides[(size_t)ci];
  const size_t nb = (size_t)((t.n + 63) / 64) * 8 + 8;
  BufP codes = dev_alloc((size_t)std::max<int64_t>(t.n, 1) * 8 + 16), valid = dev_alloc(nb);
  HIPCHECK(hipMemsetAsync(valid->p, 0, nb, (hipStream_t)x.stream));
  const int rc = gpuq_utf8_intern(dict, x.stream, &c.c, sd > 0 ? t.via[(size_t)sd - 1] : nullptr, t.n, insert ? 1 : 0, (int64_t*)codes->p, (uint8_t*)valid->p);
  check(x, rc);
  PCol k; k.name = name; k.type = jstr("Int64"); k.nullable = true;
  k.c.type = T_INT64; k.c.repr = GPUQ_REPR_ARROW; k.c.data = codes->p; k.c.validity = (const uint8_t*)valid->p; k.c.length = t.n;
  t.cols.push_back(k); t.sides.push_back(0); t.keep.push_back(codes); t.keep.push_back(valid); t.record_cap = 0;
}
void strip_code_columns(PTable& t) {
  for (size_t i = t.cols.size(); i-- > 0;)
    if (t.cols[i].name.rfind("__code_", 0) == 0) { t.cols.erase(t.cols.begin() + (long)i); t.sides.erase(t.sides.begin() + (long)i); t.record_cap = 0; }
}

static PTable project_impl(Exec& x, const PTable& t_in, const std::vector<Json>& exprs_in, const std::vector<std::string>& names, const void* site, int tag, bool long_cmp);
PTable project(Exec& x, const PTable& t_in, const std::vector<Json>& exprs_in, const std::vector<std::string>& names, const void* site, int tag) {
  try { return project_impl(x, t_in, exprs_in, names, site, tag, false); }
  catch (const Unsupported& e) { if (!is_long_string_failure(e)) throw; }
  return project_impl(x, t_in, exprs_in, names, site, tag + 64, true);      // string comparisons beyond 15 bytes: lowered to helper columns (lower_strcmp)
}
static PTable project_impl(Exec& x, const PTable& t_in, const std::vector<Json>& exprs_in, const std::vector<std::string>& names, const void* site, int tag, const bool long_cmp) {
  std::vector<Json> exprs = exprs_in;
  bool any_like = false; for (auto& e : exprs) any_like = any_like || has_like(e);
  PTable t_res = t_in; if (any_like || long_cmp) resolve(x, t_res);      // the LIKE / compare kernels take an exact row count
  PTable t = lower_like(x, t_res, exprs);
  if (long_cmp) t = lower_strcmp(x, t, exprs);
  gpuq_op* op = cached_op(x, site, tag, table_sig(t), [&]() {
    Json ex = jarr();
    const auto nm = names_of(t);
    for (size_t i = 0; i < exprs.size(); ++i) ex.a.push_back(jobj({{"expr", rebind(exprs[i], nm)}, {"name", jstr(names[i])}}));
    return jobj({{"op", jstr("project")}, {"input", jobj({{"fields", table_fields(t)}})}, {"exprs", ex}});
  });
  const bool deferred = prep(x, op, t);
  std::vector<gpuq_column> carr;
  PTable out = alloc_outputs(op, t.n, carr);
  out.count_from(t);
  InputC ic; make_input(t, ic);
  check(x, gpuq_project_run(op, x.stream, &ic.in, carr.data(), (int)carr.size()));
  if (deferred) return out;      // (the status word is read at the settle)
  // A computed expression over a table that holds strings: what the evaluator could not do (an ordering comparison of a value beyond
  // 15 bytes, ...) is in the operator's status word and nothing downstream would read it -- a bare column that is merely packed is
  // caught later, when its bytes are needed (gpuq_unpack_utf8).  Projections of numeric tables stay asynchronous.
  bool has_utf8 = false, computed = false;
  for (auto& c : t.cols) has_utf8 = has_utf8 || c.c.type == T_UTF8;
  for (auto& e : exprs) computed = computed || !(e.is_obj() && e.o.size() == 1 && (e.o[0].first == "column" || e.o[0].first == "literal"));
  if (has_utf8 && computed) check(x, gpuq_op_check(op, x.stream));
  return out;
}

// new[j] = vec[idx[j]] with NULL_ROW propagated
const uint32_t* take_u32(Exec& x, const uint32_t* vec, int64_t vec_len, const uint32_t* idx, int64_t n, std::vector<BufP>& keep, const PTable* count = nullptr) {
  PTable src; src.n = n; src.via.push_back(idx); src.sides.push_back(1);
  if (count) src.count_from(*count);
  PCol c; c.name = "v"; c.type = jstr("UInt32"); c.nullable = false; c.c.type = T_UINT32; c.c.data = vec; c.c.length = vec_len;
  src.cols.push_back(c);
  static const int take_site = 0;
  const Json v = jcol("v", 0);
  const Json e = jobj({{"case_", jobj({{"expr", Json()}, {"when_then_expr", jarr({jobj({{"when_expr", jobj({{"is_null_expr", jobj({{"expr", v}})}})},
                      {"then_expr", jobj({{"literal", jobj({{"type", jstr("UInt32")}, {"value", jstr(std::to_string(NULL_ROW_ID))}})}})}})})}, {"else_expr", v}})}});
  PTable out = project(x, src, {e}, {"v"}, &take_site, 0);
  keep.insert(keep.end(), out.keep.begin(), out.keep.end());
  return (const uint32_t*)out.cols[0].c.data;
}

PTable materialize(Exec& x, const PTable& t, bool force = false);

// address `t`'s rows through idx[0..n): a view
// drop the columns nothing above reads (PNode::require): what a join carries along is what its materialisations gather
void prune_columns(PTable& t, const std::set<std::string>& keep) {
  for (size_t i = t.cols.size(); i-- > 0;)
    if (!keep.count(t.cols[i].name)) { t.cols.erase(t.cols.begin() + (long)i); t.sides.erase(t.sides.begin() + (long)i); t.record_cap = 0; }
}
// count: the table whose (device-side) row count the n positions of idx have -- the view's count; nullptr = n is exact
PTable select_view(Exec& x, const PTable& t, const uint32_t* idx, int64_t n, const BufP& idx_owner, const PTable* count = nullptr) {
  PTable out; out.n = n; out.own(t); if (idx_owner) out.keep.push_back(idx_owner);
  if (count) out.count_from(*count);
  if (!t.is_view()) {
    out.cols = t.cols; out.via = {idx}; out.sides.assign(t.cols.size(), 1);
    return out;
  }
  bool has0 = false; for (int s : t.sides) has0 = has0 || s == 0;
  if ((int)t.via.size() + (has0 ? 1 : 0) > 3) {
    PTable m = materialize(x, t);
    out.own(m); out.cols = m.cols; out.via = {idx}; out.sides.assign(m.cols.size(), 1);
    return out;
  }
  out.cols = t.cols;
  if (has0) out.via.push_back(idx);
  for (const uint32_t* v : t.via) out.via.push_back(take_u32(x, v, t.n, idx, n, out.keep, count));
  const int shift = has0 ? 1 : 0;
  for (int s : t.sides) out.sides.push_back(s == 0 ? 1 : s + shift);
  return out;
}

// Arrow-layout Utf8 column read through an index vector (or as it lies) into a fresh Arrow-layout column: any string length
PCol take_utf8(Exec& x, const PCol& c, const uint32_t* idx, int64_t n, bool nullable, std::vector<BufP>& keep) {
  BufP offs = dev_alloc((size_t)(n + 4) * 4), valid = dev_alloc((size_t)((n + 63) / 64) * 8 + 8);
  int64_t dl = 0;
  int rc = gpuq_take_utf8(x.ctx, x.stream, &c.c, idx, n, (int32_t*)offs->p, (uint8_t*)valid->p, nullptr, 0, &dl);
  if (rc != GPUQ_OK && rc != GPUQ_ERR_CAPACITY) check(x, rc);
  BufP data = dev_alloc((size_t)dl + 16);
  if (dl > 0) check(x, gpuq_take_utf8(x.ctx, x.stream, &c.c, idx, n, (int32_t*)offs->p, (uint8_t*)valid->p, (uint8_t*)data->p, dl + 16, &dl));
  PCol o = c;
  o.nullable = nullable; o.c.repr = GPUQ_REPR_ARROW; o.c.data = data->p; o.c.offsets = (const int32_t*)offs->p;
  o.c.validity = nullable ? (const uint8_t*)valid->p : nullptr; o.c.length = n;
  keep.push_back(offs); keep.push_back(data); keep.push_back(valid);
  return o;
}

// force: also re-encode a plain table; strings then become fixed-width PACKED15 (concat / row ranges need fixed widths),
// which holds 15 bytes.  Otherwise Utf8 columns in Arrow layout are taken as they are, whatever their length.
PTable materialize(Exec& x, const PTable& t_in, bool force) {
  if (!t_in.is_view() && !force) return t_in;
  const bool pack_strings = force;
  PTable t = t_in;
  if (t.n_dev) {      // taking Arrow-layout strings is two passes with a size read-back in between; a forced re-encode feeds concatenations
    bool strings = force;
    for (auto& c : t.cols) strings = strings || (c.c.offsets && !pack_strings);
    if (strings) resolve(x, t);
  }
  PTable out; out.n = t.n; out.count_from(t);
  out.cols.resize(t.cols.size()); out.sides.assign(t.cols.size(), 0);
  std::vector<size_t> fixed;
  for (size_t i = 0; i < t.cols.size(); ++i) {
    const PCol& c = t.cols[i];
    if (c.c.offsets && !pack_strings) {
      if (!t.is_view()) out.cols[i] = c;
      else out.cols[i] = take_utf8(x, c, t.sides[i] > 0 ? t.via[(size_t)t.sides[i] - 1] : nullptr, t.n, c.nullable || (t.sides[i] > 0 && !t.dense), out.keep);
    } else fixed.push_back(i);
  }
  for (size_t a = 0; a < fixed.size(); a += 12) {
    PTable sub; sub.n = t.n; sub.via = t.via; sub.dense = t.dense; sub.count_from(t);
    std::vector<Json> ex; std::vector<std::string> nm;
    const size_t hi = std::min(a + 12, fixed.size());
    for (size_t k = a; k < hi; ++k) { sub.cols.push_back(t.cols[fixed[k]]); sub.sides.push_back(t.sides[fixed[k]]); }
    for (size_t i = 0; i < sub.cols.size(); ++i) { ex.push_back(jcol(sub.cols[i].name, (int)i)); nm.push_back(sub.cols[i].name); }
    // rebind by position, not by name: duplicate names (join outputs) must keep their own column
    static const int mat_site = 0;
    gpuq_op* op = cached_op(x, &mat_site, 0, table_sig(sub), [&]() {
      Json exj = jarr();
      for (size_t i = 0; i < ex.size(); ++i) exj.a.push_back(jobj({{"expr", ex[i]}, {"name", jstr(nm[i])}}));
      return jobj({{"op", jstr("project")}, {"input", jobj({{"fields", table_fields(sub)}})}, {"exprs", exj}});
    });
    prep(x, op, sub);
    if (!sub.n_dev && t.n_dev) { t.n = sub.n; t.count_from(sub); out.n = sub.n; out.count_from(sub); }      // (resolved on the way)
    std::vector<gpuq_column> carr;
    PTable part = alloc_outputs(op, sub.n, carr);
    InputC ic; make_input(sub, ic);
    check(x, gpuq_project_run(op, x.stream, &ic.in, carr.data(), (int)carr.size()));
    for (size_t k = a; k < hi; ++k) out.cols[fixed[k]] = part.cols[k - a];
    out.own(part);
  }
  // the kernels above read t's buffers asynchronously: keep them alive as long as the result
  out.own(t);
  return out;
}

// the passing driving positions of `source` (in order) and their number
static int64_t filter_sel_impl(Exec& x, const PTable& source_in, const Json& predicate_in, const void* site, int tag, BufP& sel_out, bool long_cmp);
int64_t filter_sel(Exec& x, const PTable& source_in, const Json& predicate_in, const void* site, int tag, BufP& sel_out) {
  try { return filter_sel_impl(x, source_in, predicate_in, site, tag, sel_out, false); }
  catch (const Unsupported& e) { if (!is_long_string_failure(e)) throw; }
  return filter_sel_impl(x, source_in, predicate_in, site, tag + 64, sel_out, true);
}
static int64_t filter_sel_impl(Exec& x, const PTable& source_in, const Json& predicate_in, const void* site, int tag, BufP& sel_out, const bool long_cmp) {
  std::vector<Json> pe{predicate_in};
  PTable source = source_in; resolve(x, source);      // (the callers of this form need the exact count back)
  PTable t = lower_like(x, source, pe);
  if (long_cmp) t = lower_strcmp(x, t, pe);
  const Json& predicate = pe[0];
  gpuq_op* op = cached_op(x, site, tag, table_sig(t), [&]() {
    return jobj({{"op", jstr("filter")}, {"input", jobj({{"fields", table_fields(t)}})}, {"predicate", rebind(predicate, names_of(t))}});
  });
  sel_out = dev_alloc((size_t)std::max<int64_t>(t.n, 1) * 4 + 16);
  BufP cnt = dev_alloc(16);
  InputC ic; make_input(t, ic);
  check(x, gpuq_filter_run(op, x.stream, &ic.in, 0, (uint32_t*)sel_out->p, (uint64_t*)cnt->p));
  const int64_t k = (int64_t)read_u64(x, cnt->p);
  check(x, gpuq_op_check(op, x.stream));
  return k;
}

static PTable filter_table_impl(Exec& x, const PTable& source_in, const Json& predicate_in, const void* site, int tag, bool long_cmp);
PTable filter_table(Exec& x, const PTable& source_in, const Json& predicate_in, const void* site, int tag) {
  try { return filter_table_impl(x, source_in, predicate_in, site, tag, false); }
  catch (const Unsupported& e) { if (!is_long_string_failure(e)) throw; }
  return filter_table_impl(x, source_in, predicate_in, site, tag + 64, true);      // string comparisons beyond 15 bytes: lowered to helper columns (lower_strcmp)
}
static PTable filter_table_impl(Exec& x, const PTable& source_in, const Json& predicate_in, const void* site, int tag, const bool long_cmp) {
  std::vector<Json> pe{predicate_in};
  PTable source = source_in;
  if (has_like(predicate_in) || long_cmp) resolve(x, source);      // the LIKE / compare kernels take an exact row count
  PTable t = lower_like(x, source, pe);          // helper columns are visible to the predicate only: the view is over `source`
  if (long_cmp) t = lower_strcmp(x, t, pe);
  const Json& predicate = pe[0];
  gpuq_op* op = cached_op(x, site, tag, table_sig(t), [&]() {
    return jobj({{"op", jstr("filter")}, {"input", jobj({{"fields", table_fields(t)}})}, {"predicate", rebind(predicate, names_of(t))}});
  });
  const bool deferred = prep(x, op, t);
  if (!t.n_dev && source.n_dev) { source.n = t.n; source.count_from(t); }
  BufP sel = dev_alloc((size_t)std::max<int64_t>(t.n, 1) * 4 + 16), cnt = dev_alloc(16);
  InputC ic; make_input(t, ic);
  check(x, gpuq_filter_run(op, x.stream, &ic.in, 0, (uint32_t*)sel->p, (uint64_t*)cnt->p));
  if (deferred && t.n > 0) {      // the survivors' count stays on the device: the view is as long as its input at most
    PTable c; c.n_dev = (const uint64_t*)cnt->p; c.n_keep = cnt;
    return select_view(x, source, (const uint32_t*)sel->p, t.n, sel, &c);
  }
  const int64_t k = (int64_t)read_u64(x, cnt->p);
  check(x, gpuq_op_check(op, x.stream));
  return select_view(x, source, (const uint32_t*)sel->p, k, sel);
}

// the stable permutation that puts `t` in `sort_exprs` order (one gpuq_sort_run: <= 4 keys whose composite fits 128 bits)
// ORDER BY one plain column of an integer-like type: the operator can hand that column back IN ORDER (rebuilt from its sorted records,
// gpuq_sort_run_keys) -- the view over the sorted table then reads it directly instead of through the permutation.
struct SortedKey { int col = -1; BufP data, valid; bool decoded = false; };
static BufP sort_perm(Exec& x, PTable& t, const Json& sort_exprs, const void* site, int tag, SortedKey* sk = nullptr) {      // (t's count is made exact when the operator cannot run deferred)
  gpuq_op* op = cached_op(x, site, tag, table_sig(t), [&]() {
    const auto nm = names_of(t);
    Json ex = jarr();
    for (auto& s : sort_exprs.a) {
      const bool asc = s.get_bool("asc", true);
      ex.a.push_back(jobj({{"expr", rebind(s.at("expr"), nm)}, {"asc", jbool(asc)}, {"nulls_first", jbool(s.get_bool("nulls_first", !asc))}}));
    }
    return jobj({{"op", jstr("sort")}, {"input", jobj({{"fields", table_fields(t)}})}, {"expr", ex}});
  });
  prep(x, op, t);
  BufP perm = dev_alloc((size_t)std::max<int64_t>(t.n, 1) * 4 + 16);
  InputC ic; make_input(t, ic);
  if (sk && sk->col >= 0) {
    const PCol& kc = t.cols[(size_t)sk->col];
    DType dt = dtype_from_json(kc.type);
    const size_t w = (size_t)type_width(dt);
    sk->data = dev_alloc((size_t)std::max<int64_t>(t.n, 1) * w + 16);
    const bool nullable = kc.nullable || (t.sides[(size_t)sk->col] > 0 && !t.dense);
    if (nullable) sk->valid = dev_alloc((size_t)((t.n + 63) / 64) * 8 + 8);
    int decoded = 0;
    check(x, gpuq_sort_run_keys(op, x.stream, &ic.in, (uint32_t*)perm->p, sk->data->p, sk->valid ? (uint8_t*)sk->valid->p : nullptr, &decoded));
    sk->decoded = decoded != 0;
    return perm;
  }
  check(x, gpuq_sort_run(op, x.stream, &ic.in, (uint32_t*)perm->p));
  return perm;
}

// Utf8 sort keys longer than 15 bytes (q2 / q21's s_name, q16's p_type, q18's c_name).  The packed key has refused; the order is then
// built from stable passes, least significant first (LSD over the ORDER BY list): a long string key becomes ceil(max_len / 14)
// pieces, each an order-preserving 16-byte integer column (gpuq_utf8_sort_piece) sorted in a pass of its own; the keys between
// long strings go through in groups of up to four, as one composite key per pass.  Every pass sorts the rows in the order the
// later keys left them, so ties keep that order.  Nothing is paid on the common path: this runs only after the loud failure.
struct PermOut { const uint32_t* p = nullptr; std::vector<BufP> keep; };
static PermOut sort_perm_long(Exec& x, const PTable& t, const Json& sort_exprs, const void* site, int tag, size_t max_group = 4) {
  PTable w = t;
  std::vector<Json> groups; Json cur = jarr();
  auto flush = [&]() { if (!cur.a.empty()) { groups.push_back(cur); cur = jarr(); } };
  int np = 0;
  for (auto& s : sort_exprs.a) {
    const int ci = long_key_column(t, s.at("expr"));
    int32_t maxlen = 0;
    if (ci >= 0) { const int sd = t.sides[(size_t)ci]; check(x, gpuq_utf8_max_len(x.ctx, x.stream, &t.cols[(size_t)ci].c, sd > 0 ? t.via[(size_t)sd - 1] : nullptr, t.n, &maxlen)); }
    if (ci < 0 || maxlen <= 15) { cur.a.push_back(s); if (cur.a.size() >= max_group) flush(); continue; }
    flush();
    const bool asc = s.get_bool("asc", true);
    for (int j = 0; j < (maxlen + 13) / 14; ++j) {
      const std::string name = "__sortpiece_" + std::to_string(np++);
      const PCol& c = w.cols[(size_t)ci]; const int sd = w.sides[(size_t)ci];
      const size_t nb = (size_t)((w.n + 63) / 64) * 8 + 8;
      BufP keys = dev_alloc((size_t)std::max<int64_t>(w.n, 1) * 16 + 16), valid = dev_alloc(nb);
      HIPCHECK(hipMemsetAsync(valid->p, 0, nb, (hipStream_t)x.stream));
      check(x, gpuq_utf8_sort_piece(x.ctx, x.stream, &c.c, sd > 0 ? w.via[(size_t)sd - 1] : nullptr, w.n, j, keys->p, (uint8_t*)valid->p));
      PCol k; k.name = name; k.type = jobj({{"Decimal128", jarr({jnum(38), jnum(0)})}}); k.nullable = true;
      k.c.type = T_DECIMAL128; k.c.precision = 38; k.c.scale = 0; k.c.repr = GPUQ_REPR_ARROW; k.c.data = keys->p; k.c.validity = (const uint8_t*)valid->p; k.c.length = w.n;
      w.cols.push_back(k); w.sides.push_back(0); w.keep.push_back(keys); w.keep.push_back(valid); w.record_cap = 0;
      groups.push_back(jarr({jobj({{"expr", jcol(name, (int)w.cols.size() - 1)}, {"asc", jbool(asc)}, {"nulls_first", jbool(s.get_bool("nulls_first", !asc))}})}));
    }
  }
  flush();
  PermOut out;
  for (size_t gi = groups.size(); gi-- > 0;) {
    if (!out.p) { BufP p = sort_perm(x, w, groups[gi], site, tag * 64 + 8 + (int)gi); out.p = (const uint32_t*)p->p; out.keep.push_back(p); continue; }      // (w is exact: see sort_table)
    PTable view = select_view(x, w, out.p, w.n, nullptr);
    BufP p2 = sort_perm(x, view, groups[gi], site, tag * 64 + 8 + (int)gi);
    out.keep.push_back(p2);
    for (auto& b : view.keep) out.keep.push_back(b);
    out.p = take_u32(x, out.p, w.n, (const uint32_t*)p2->p, w.n, out.keep);      // rows in the new order = old order read through the pass's permutation
  }
  for (auto& b : w.keep) out.keep.push_back(b);
  return out;
}

static bool is_wide_sort_key(const std::exception& e) { return std::string(e.what()).find("composite sort key needs") != std::string::npos; }
PTable sort_table(Exec& x, const PTable& t_in, const Json& sort_exprs, int64_t fetch, const void* site, int tag) {
  PTable t = t_in; bool wide = false;
  try {
    // one key that is a plain fixed-width column of a materialised table with enough rows for the radix passes: ask for it in order
    SortedKey sk;
    if (sort_exprs.a.size() == 1 && !t.is_view() && t.n >= (1 << 20)) {
      const Json& e = sort_exprs.a[0].at("expr");
      if (e.is_obj() && e.o.size() == 1 && e.o[0].first == "column" && e.o[0].second.find("name")) {
        const std::string& nm = e.o[0].second.at("name").str();
        for (size_t i = 0; i < t.cols.size(); ++i) if (t.cols[i].name == nm) {
          const int ty = t.cols[i].c.type;
          if (ty == T_INT8 || ty == T_INT16 || ty == T_INT32 || ty == T_INT64 || ty == T_UINT8 || ty == T_UINT16 || ty == T_UINT32 || ty == T_DATE32 || ty == T_DATE64 ||
              ty == T_TIMESTAMP || ty == T_DECIMAL128) sk.col = (int)i;
          break;
        }
      }
    }
    BufP perm = sort_perm(x, t, sort_exprs, site, tag, sk.col >= 0 ? &sk : nullptr);
    // deferred: the first *n_dev entries of the permutation are the sorted rows (padding sorts behind them); a fetch bounds both
    const int64_t k = (fetch < 0 || fetch > t.n) ? t.n : fetch;
    PTable out = select_view(x, t, (const uint32_t*)perm->p, k, perm, t.n_dev ? &t : nullptr);
    if (sk.decoded) {      // the key column lies in order already: read it as it lies (side 0), everything else through the permutation
      PCol& kc = out.cols[(size_t)sk.col];
      kc.c.data = sk.data->p; kc.c.validity = sk.valid ? (const uint8_t*)sk.valid->p : nullptr; kc.c.length = k;
      kc.nullable = sk.valid != nullptr;
      out.sides[(size_t)sk.col] = 0;
      out.keep.push_back(sk.data); if (sk.valid) out.keep.push_back(sk.valid);
      out.record_cap = 0;
    }
    return out;
  } catch (const Unsupported& e) { if (!is_long_string_failure(e) && !is_wide_sort_key(e)) throw; wide = is_wide_sort_key(e); }
  resolve(x, t);
  const int64_t k = (fetch < 0 || fetch > t.n) ? t.n : fetch;
  // strings beyond 15 bytes: stable passes over 14-byte pieces; a composite key beyond 128 bits (two packed strings and a Float64 whose bit
  // patterns span 2^40: q7's ORDER BY supp_nation, cust_nation, l_year): one stable pass per key, least significant first
  PermOut po = sort_perm_long(x, t, sort_exprs, site, tag, wide ? 1 : 4);
  PTable out = select_view(x, t, po.p, k, nullptr);
  for (auto& b : po.keep) out.keep.push_back(b);
  return out;
}

PTable concat_tables(Exec& x, std::vector<PTable> parts);

// ordered fan-in (gpuq_merge_run): `t` is the concatenation of runs that are each in `sort_exprs` order
PTable merge_table(Exec& x, const PTable& t_in, const std::vector<int64_t>& run_offsets, const Json& sort_exprs, int64_t fetch, const void* site, int tag) {
  if (run_offsets.size() <= 2) return sort_table(x, t_in, sort_exprs, fetch, site, tag);
  PTable t = t_in; resolve(x, t);      // one partition (e.g. gathered by BroadcastExec: several runs inside): sort
  gpuq_op* op = cached_op(x, site, tag, table_sig(t), [&]() {
    const auto nm = names_of(t);
    Json ex = jarr();
    for (auto& s : sort_exprs.a) {
      const bool asc = s.get_bool("asc", true);
      ex.a.push_back(jobj({{"expr", rebind(s.at("expr"), nm)}, {"asc", jbool(asc)}, {"nulls_first", jbool(s.get_bool("nulls_first", !asc))}}));
    }
    return jobj({{"op", jstr("sort")}, {"input", jobj({{"fields", table_fields(t)}})}, {"expr", ex}});
  });
  BufP perm = dev_alloc((size_t)std::max<int64_t>(t.n, 1) * 4 + 16);
  InputC ic; make_input(t, ic);
  check(x, gpuq_merge_run(op, x.stream, &ic.in, run_offsets.data(), (int)run_offsets.size() - 1, (uint32_t*)perm->p));
  const int64_t k = (fetch < 0 || fetch > t.n) ? t.n : fetch;
  return select_view(x, t, (const uint32_t*)perm->p, k, perm);
}

// ---------------------------------------------------------------- plan nodes
struct Metrics { int64_t output_rows = 0, elapsed_ns = 0; };
struct PNode {
  std::string kind; Metrics m; int id = 0;      // id: position in the plan, pre-order (labels the node's run-time compiled kernels)
  std::string label(const char* what) const { return std::string(what) + "_n" + std::to_string(id); }
  virtual ~PNode() {}
  virtual std::vector<PNode*> children() { return {}; }
  virtual int partitions() { auto c = children(); return c.empty() ? 1 : c[0]->partitions(); }
  // output schema, known before anything runs (QueryStageExecutor::schema(), execution_engine.rs:59; executor_server.rs:530-534
  // parses the output partitioning from it before the task executes).  Default: the input's.
  virtual PSchema schema() { auto c = children(); if (c.empty()) throw std::runtime_error("plan: node without a schema"); return c[0]->schema(); }
  virtual PTable execute(int part, Exec& x) = 0;
  // Which of this node's output columns does anything above it read?  (nullptr = all of them: the root, or a parent that does not
  // say.)  Walked once when the plan is created; a chain-fused join drops the inner build side's columns when nobody wants them.
  typedef std::set<std::string> Names;
  virtual void require(const Names* need) { (void)need; for (PNode* c : children()) c->require(nullptr); }
  PTable timed(Exec& x, std::chrono::steady_clock::time_point t0, PTable t) {
    m.elapsed_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    m.output_rows += t.n;
    if (t.n_dev) x.fixes.push_back({t.n_dev, t.n, &m.output_rows, t.n_keep});      // booked with the bound, corrected at the settle
    return t;
  }
};
typedef std::unique_ptr<PNode> PNodeP;
PNodeP build_node(const Json& j);

struct MemoryExec : PNode {        // leaf: partitions[p] = index into the inputs handed to gpuq_plan_execute
  PSchema schema_; std::vector<int> parts; bool dense = false;
  PSchema schema() override { return schema_; }
  int partitions() override { return (int)parts.size(); }
  PTable execute(int part, Exec& x) override {
    if (part < 0 || part >= (int)parts.size()) throw std::runtime_error("MemoryExec: partition out of range");
    const int slot = parts[(size_t)part];
    if (slot < 0 || slot >= x.n_inputs) throw std::runtime_error("MemoryExec: input " + std::to_string(slot) + " was not supplied");
    const gpuq_input& in = x.inputs[slot];
    if (in.n_cols != (int)schema_.size()) throw std::runtime_error("MemoryExec: input has " + std::to_string(in.n_cols) + " columns, schema_ has " + std::to_string(schema_.size()));
    PTable t; t.n = in.n_rows; t.dense = dense;
    for (int k = 0; k < in.n_via; ++k) t.via.push_back(in.via[k]);
    for (int i = 0; i < in.n_cols; ++i) {
      PCol c; c.name = schema_[(size_t)i].name; c.type = schema_[(size_t)i].type; c.nullable = schema_[(size_t)i].nullable; c.c = in.cols[i];
      int p, s; const int tid = type_id_of(c.type, p, s);
      if (c.c.type != tid) throw std::runtime_error("MemoryExec: column '" + c.name + "' does not have the declared type");
      t.cols.push_back(c);
    }
    t.sides = sides.empty() ? std::vector<int>(t.cols.size(), in.n_via > 0 ? 1 : 0) : sides;
    m.output_rows += t.n;
    return t;
  }
  std::vector<int> sides;       // per column: index vector it is read through (views handed in by the caller)
};

// ---------------------------------------------------------------- static schemas (no device: gpuq_compile_check types the expressions)
Json schema_fields(const PSchema& s) {
  Json f = jarr();
  for (auto& c : s) f.a.push_back(jobj({{"name", jstr(c.name)}, {"type", c.type}, {"nullable", jbool(c.nullable)}, {"side", jnum(0)}}));
  return f;
}
std::vector<std::string> schema_names(const PSchema& s) { std::vector<std::string> v; for (auto& c : s) v.push_back(c.name); return v; }
Json type_json_from_string(const std::string& t) {      // DType::to_string(): "Int64", "Decimal128(15, 2)" ...
  if (t.rfind("Decimal128", 0) == 0) {
    long long p = 38, sc = 0; std::sscanf(t.c_str(), "Decimal128(%lld, %lld)", &p, &sc);
    if (t.find(',') != std::string::npos && t.find(", ") == std::string::npos) std::sscanf(t.c_str(), "Decimal128(%lld,%lld)", &p, &sc);
    return jobj({{"Decimal128", jarr({jnum(p), jnum(sc)})}});
  }
  return jstr(t);
}
PSchema compiled_outputs(const Json& desc) {
  std::vector<char> buf(1 << 20);
  const int rc = gpuq_compile_check(desc.dump().c_str(), buf.data(), buf.size());
  if (rc != GPUQ_OK) { const char* m = gpuq_last_error(nullptr); throw std::runtime_error(std::string("plan schema: ") + (m ? m : "descriptor does not compile")); }
  const Json r = JsonParser(buf.data()).parse();
  PSchema out;
  for (auto& o : r.at("outputs").a) out.push_back({o.at("name").str(), type_json_from_string(o.at("type").str()), o.get_bool("nullable", true)});
  return out;
}

void collect_columns(const Json& e, std::set<std::string>& out) {
  (void)rewrite_columns(e, [&](const Json& c) { out.insert(c.at("name").str()); return jobj({{"column", c}}); });
}
struct PassThrough : PNode {       // CoalesceBatchesExec: whole partitions are already single tables
  PNodeP input;
  std::vector<PNode*> children() override { return {input.get()}; }
  void require(const Names* need) override { input->require(need); }
  PTable execute(int part, Exec& x) override { return input->execute(part, x); }
};

struct FilterExec : PNode {
  PNodeP input; Json predicate;
  std::vector<PNode*> children() override { return {input.get()}; }
  void require(const Names* need) override { if (!need) { input->require(nullptr); return; } Names n = *need; collect_columns(predicate, n); input->require(&n); }
  PTable execute(int part, Exec& x) override;
};
struct ProjectionExec : PNode {
  PNodeP input; std::vector<Json> exprs; std::vector<std::string> names;
  std::vector<PNode*> children() override { return {input.get()}; }
  void require(const Names*) override { Names n; for (auto& e : exprs) collect_columns(e, n); input->require(&n); }
  PSchema schema() override {
    const PSchema in = input->schema(); const auto nm = schema_names(in);
    std::vector<Json> ex = exprs;
    // LikeExpr is lowered to a Boolean column at run time: for typing, a like_expr is Boolean with its operand's nullability
    for (auto& e : ex) e = rewrite_like(e, [&](const Json& v) { return jobj({{"is_not_null_expr", jobj({{"expr", jobj({{"not_expr", jobj({{"expr", jobj({{"is_null_expr", jobj({{"expr", v.at("expr")}})}})}})}})}})}}); });
    Json exj = jarr();
    for (size_t i = 0; i < ex.size(); ++i) exj.a.push_back(jobj({{"expr", rebind(ex[i], nm)}, {"name", jstr(names[i])}}));
    PSchema out = compiled_outputs(jobj({{"op", jstr("project")}, {"input", jobj({{"fields", schema_fields(in)}})}, {"exprs", exj}}));
    for (size_t i = 0; i < exprs.size() && i < out.size(); ++i) if (has_like(exprs[i])) out[i].nullable = true;
    return out;
  }
  PTable execute(int part, Exec& x) override;
};

// Walk down through Filter / Projection / CoalesceBatches: (source node, AND-ed predicate, column map)
struct Fused { PNode* src = nullptr; bool has_pred = false; Json pred; bool has_map = false; ColMap map; };
Fused fuse(PNode* n) {
  if (auto* p = dynamic_cast<PassThrough*>(n)) return fuse(p->input.get());
  if (auto* f = dynamic_cast<FilterExec*>(n)) {
    if (has_like(f->predicate)) { Fused r; r.src = n; return r; }      // LIKE is its own kernel over this filter's input: not inlined into consumers
    Fused r = fuse(f->input.get());
    const Json mine = inline_projection(f->predicate, r.has_map ? &r.map : nullptr);
    r.pred = r.has_pred ? jand(r.pred, mine) : mine; r.has_pred = true;
    return r;
  }
  if (auto* p = dynamic_cast<ProjectionExec*>(n)) {
    for (auto& e : p->exprs) if (has_like(e)) { Fused r; r.src = n; return r; }
    Fused r = fuse(p->input.get());
    ColMap nm;
    for (size_t i = 0; i < p->exprs.size(); ++i) nm[p->names[i]] = inline_projection(p->exprs[i], r.has_map ? &r.map : nullptr);
    r.map = std::move(nm); r.has_map = true;
    return r;
  }
  Fused r; r.src = n; return r;
}

PTable FilterExec::execute(int part, Exec& x) {
  Fused f;
  if (has_like(predicate)) {       // fuse what lies below, evaluate this predicate (with its LIKE kernel) here
    f = fuse(input.get());
    const Json mine = inline_projection(predicate, f.has_map ? &f.map : nullptr);
    f.pred = f.has_pred ? jand(f.pred, mine) : mine; f.has_pred = true;
  } else f = fuse(this);
  PTable t = f.src->execute(part, x);
  auto t0 = std::chrono::steady_clock::now();
  if (!f.has_map) return timed(x, t0, filter_table(x, t, f.pred, this, 0));
  // a computed projection sits below: run it on the filtered rows
  PTable v = filter_table(x, t, f.pred, this, 0);
  std::vector<Json> ex; std::vector<std::string> nm;
  for (auto& kv : f.map) { nm.push_back(kv.first); ex.push_back(kv.second); }
  return timed(x, t0, project(x, v, ex, nm, this, 1));
}
PTable ProjectionExec::execute(int part, Exec& x) {
  Fused f = fuse(input.get());
  PTable t = f.src->execute(part, x);
  auto t0 = std::chrono::steady_clock::now();
  if (f.has_pred) t = filter_table(x, t, f.pred, this, 0);
  std::vector<Json> ex;
  for (auto& e : exprs) ex.push_back(inline_projection(e, f.has_map ? &f.map : nullptr));
  // a projection that only selects / renames columns is a view of its input: no kernel, and strings of any length pass through
  // (the project kernel writes Utf8 results as 16-byte PACKED15 values)
  std::vector<int> pick;
  for (auto& e : ex) {
    int ci = -1;
    if (e.is_obj() && e.o.size() == 1 && e.o[0].first == "column" && e.o[0].second.is_obj() && e.o[0].second.find("name")) {
      const std::string& n = e.o[0].second.at("name").str();
      for (size_t i = 0; i < t.cols.size(); ++i) if (t.cols[i].name == n) { ci = (int)i; break; }
    }
    if (ci < 0) { pick.clear(); break; }
    pick.push_back(ci);
  }
  if (!pick.empty() && pick.size() == ex.size()) {
    PTable out; out.n = t.n; out.via = t.via; out.dense = t.dense; out.keep = t.keep; out.count_from(t);
    for (size_t i = 0; i < pick.size(); ++i) { PCol c = t.cols[(size_t)pick[i]]; c.name = names[i]; out.cols.push_back(c); out.sides.push_back(t.sides[(size_t)pick[i]]); }
    return timed(x, t0, out);
  }
  return timed(x, t0, project(x, t, ex, names, this, 1));
}

struct AggregateExec : PNode {
  PNodeP input; std::string mode, strategy = "auto"; Json group_expr, aggr_expr; int64_t expected_groups = 0, output_capacity = 0;
  std::vector<PNode*> children() override { return {input.get()}; }
  void require(const Names*) override {
    if (mode == "Final" || mode == "FinalPartitioned") { input->require(nullptr); return; }      // reads its input's state columns by position
    Names n;
    for (auto& g : group_expr.a) collect_columns(g.at("expr"), n);
    for (auto& a : aggr_expr.a) for (const char* k : {"expr", "expr2", "filter"}) if (a.has(k)) collect_columns(a.at(k), n);
    input->require(&n);
  }
  PSchema schema() override {
    const PSchema in = input->schema(); const auto nm = schema_names(in);
    Json ge = jarr(), ae = jarr();
    for (auto& g : group_expr.a) ge.a.push_back(jobj({{"expr", rebind(g.at("expr"), nm)}, {"name", g.at("name")}}));
    for (auto& a : aggr_expr.a) {
      std::vector<std::pair<std::string, Json>> o = {{"fn", a.at("fn")}, {"name", a.at("name")}};
      for (const char* k : {"expr", "expr2", "filter"}) if (a.has(k)) o.push_back({k, rebind(a.at(k), nm)});
      ae.a.push_back(jobj(o));
    }
    PSchema out = compiled_outputs(jobj({{"op", jstr("aggregate")}, {"mode", jstr(mode)}, {"input", jobj({{"fields", schema_fields(in)}})}, {"strategy", jstr("hash")},
                                         {"group_expr", ge}, {"aggr_expr", ae}}));
    // DataFusion declares every aggregate but COUNT nullable (an empty group has no SUM / MIN / MAX / AVG) [UPSTREAM-KNOWLEDGE:
    // Sum::field etc.]; the operator may type a column tighter when its inputs cannot be NULL -- announce the declared form
    // (COUNT too: the merged count of a two-phase aggregate comes back through a nullable state column here; announcing
    // "nullable" for a column that never holds a NULL is harmless, the reverse is not)
    if (mode == "Final" || mode == "FinalPartitioned" || mode == "Single")
      for (size_t i = group_expr.a.size(); i < out.size(); ++i) out[i].nullable = true;
    return out;
  }
  PTable execute(int part, Exec& x) override {
    const bool final_ = mode == "Final" || mode == "FinalPartitioned";
    Fused f; if (final_) f.src = input.get(); else f = fuse(input.get());
    PTable t = f.src->execute(part, x);
    auto t0 = std::chrono::steady_clock::now();
    // (more group columns than the table holds keys, some of them strings: codes from the start -- a 128-bit packed string takes a key
    // slot of its own, a 32-bit code is packed with its neighbours, see compile_aggregate)
    bool many_string_keys = false;
    if (group_expr.a.size() > 4) {
      const ColMap* cm0 = f.has_map ? &f.map : nullptr;
      for (auto& g : group_expr.a) if (long_key_column(t, inline_projection(g.at("expr"), cm0)) >= 0) many_string_keys = true;
    }
    if (!many_string_keys) {
      try { return timed(x, t0, run(x, t, f, false)); }
      catch (const Unsupported& e) { if (!is_long_string_failure(e)) throw; }
    }
    resolve(x, t);
    try { return timed(x, t0, run(x, t, f, true)); }      // a group key holds strings of more than 15 bytes: again, over dictionary codes
    catch (const Unsupported& e) { if (!is_long_string_failure(e) || !f.has_pred) throw; }
    // it was the fused FILTER that compared long strings: apply it on its own (filter_table lowers such comparisons), then aggregate
    Fused f2 = f; f2.has_pred = false;
    PTable t2 = filter_table(x, t, f.pred, this, 9);
    return timed(x, t0, run(x, t2, f2, true));
  }
  PTable run(Exec& x, PTable t, const Fused& f, const bool long_keys) {
    const ColMap* cm0 = f.has_map ? &f.map : nullptr;
    // group expressions over the source table; with long_keys every plain Arrow-layout Utf8 key column is replaced by its codes
    std::vector<Json> gexprs; std::vector<int> coded;      // coded[k] = column of `t` key k's strings come from, or -1
    Utf8DictGuard dicts[8]; int nd = 0;
    const PTable t_src = t;
    for (auto& g : group_expr.a) {
      Json e = inline_projection(g.at("expr"), cm0); int ci = -1;
      if (long_keys && (ci = long_key_column(t_src, e)) >= 0 && nd < 8) {
        const std::string nm = "__code_" + std::to_string(gexprs.size());
        check(x, gpuq_utf8_dict_create(x.ctx, x.stream, std::min<int64_t>(t_src.n, t_src.cols[(size_t)ci].c.length), &dicts[nd].d));
        append_code_column(x, t, ci, dicts[nd].d, true, nm); ++nd;
        e = jobj({{"column", jobj({{"name", jstr(nm)}})}});
        // a code is a row id: as UInt32 it takes 35 bits of a packed key instead of 66 (more group columns than key slots, compile_aggregate)
        if (group_expr.a.size() > 4) e = jobj({{"cast", jobj({{"expr", e}, {"arrow_type", jstr("UInt32")}})}});
      } else ci = -1;
      gexprs.push_back(e); coded.push_back(ci);
    }
    gpuq_op* op = cached_op(x, this, long_keys ? 6 : 0, table_sig(t), [&]() {
    const auto nm = names_of(t);
    const ColMap* cm = f.has_map ? &f.map : nullptr;
    Json ge = jarr(), ae = jarr();
    for (size_t k = 0; k < group_expr.a.size(); ++k) ge.a.push_back(jobj({{"expr", rebind(gexprs[k], nm)}, {"name", group_expr.a[k].at("name")}}));
    for (auto& a : aggr_expr.a) {
      std::vector<std::pair<std::string, Json>> o = {{"fn", a.at("fn")}, {"name", a.at("name")}};
      for (const char* k : {"expr", "expr2", "filter"}) if (a.has(k)) o.push_back({k, rebind(inline_projection(a.at(k), cm), nm)});
      ae.a.push_back(jobj(o));
    }
    std::vector<std::pair<std::string, Json>> d = {{"op", jstr("aggregate")}, {"mode", jstr(mode)}, {"input", jobj({{"fields", table_fields(t)}})},
                                                   {"strategy", jstr(strategy)}, {"group_expr", ge}, {"aggr_expr", ae}};
    if (f.has_pred) d.push_back({"predicate", rebind(f.pred, nm)});
    if (expected_groups) d.push_back({"expected_groups", jnum(expected_groups)});
    return jobj(d);
    });
    const bool deferred = !long_keys && prep(x, op, t);
    if (!deferred) resolve(x, t);
    int64_t cap = output_capacity > 0 ? output_capacity : (group_expr.a.empty() ? 4096 : std::max<int64_t>(4096, std::min<int64_t>(t.n, 1ll << 22)));
    if (output_capacity <= 0 && expected_groups > 0) cap = std::max<int64_t>(cap, std::min<int64_t>(t.n, expected_groups + expected_groups / 4));   // a too-small capacity costs a second run
    InputC ic; make_input(t, ic);
    if (deferred) {
      // nothing is read back: the result is as long as the remembered group count allows, its actual length is a device word
      // (copied out of the operator's status block, which its next run reuses)
      std::vector<gpuq_column> carr;
      PTable out = alloc_outputs(op, cap, carr);
      int64_t bound = 0; const uint64_t* ndev = nullptr;
      check(x, gpuq_aggregate_run_deferred(op, x.stream, &ic.in, carr.data(), (int)carr.size(), cap, &bound, &ndev));
      out.n = bound; for (auto& c : out.cols) c.c.length = bound;
      if (ndev) {
        BufP cnt = dev_alloc(16);
        HIPCHECK(hipMemcpyAsync(cnt->p, ndev, 8, hipMemcpyDeviceToDevice, (hipStream_t)x.stream));
        out.n_dev = (const uint64_t*)cnt->p; out.n_keep = cnt;
      }
      return out;
    }
    for (;;) {
      std::vector<gpuq_column> carr;
      PTable out = alloc_outputs(op, cap, carr);
      int64_t ng = 0;
      const int rc = gpuq_aggregate_run(op, x.stream, &ic.in, carr.data(), (int)carr.size(), cap, &ng);
      ++x.host_syncs;
      if (rc == GPUQ_ERR_CAPACITY && ng > cap) { cap = ng; continue; }
      check(x, rc);
      out.n = ng; for (auto& c : out.cols) c.c.length = ng;
      // coded keys: the code IS the row of a representative string in the source column -- take the strings back
      for (size_t k = 0; k < coded.size(); ++k) {
        if (coded[k] < 0) continue;
        BufP rows = dev_alloc((size_t)std::max<int64_t>(ng, 1) * 4 + 16);
        check(x, gpuq_utf8_code_rows(x.ctx, x.stream, &out.cols[k].c, ng, (uint32_t*)rows->p));
        const PCol& src = t_src.cols[(size_t)coded[k]];
        PCol sc = take_utf8(x, src, (const uint32_t*)rows->p, ng, true, out.keep);
        sc.name = out.cols[k].name; out.cols[k] = sc; out.record_cap = 0;
        HIPCHECK(hipStreamSynchronize((hipStream_t)x.stream));      // `rows` dies here
      }
      return out;       // synchronous call: the input buffers are no longer referenced
    }
  }
};

struct SortExec : PNode {
  PNodeP input; Json expr; int64_t fetch = -1; bool merge_all = false;     // merge_all: SortPreservingMergeExec over a single partition
  std::vector<PNode*> children() override { return {input.get()}; }
  void require(const Names* need) override { if (!need) { input->require(nullptr); return; } Names n = *need; for (auto& s : expr.a) collect_columns(s.at("expr"), n); input->require(&n); }
  int partitions() override { return merge_all ? 1 : input->partitions(); }
  PTable execute(int part, Exec& x) override {
    PTable t;
    if (merge_all && input->partitions() != 1) {      // SortPreservingMergeExec: the partitions are sorted runs
      std::vector<PTable> in; std::vector<int64_t> offs{0};
      for (int p = 0; p < input->partitions(); ++p) { in.push_back(input->execute(p, x)); resolve(x, in.back()); offs.push_back(offs.back() + in.back().n); }
      auto t0 = std::chrono::steady_clock::now();
      t = concat_tables(x, std::move(in));
      return timed(x, t0, merge_table(x, t, offs, expr, fetch, this, 1));
    }
    t = input->execute(part, x);
    auto t0 = std::chrono::steady_clock::now();
    return timed(x, t0, sort_table(x, t, expr, fetch, this, 0));
  }
};

struct HashJoinExec : PNode {
  PNodeP left, right; Json on; std::string join_type = "Inner", partition_mode = "CollectLeft"; bool null_equals_null = false; bool has_filter = false; Json filter;
  std::vector<PNode*> children() override { return {left.get(), right.get()}; }
  bool out_need_all = true; Names out_need;      // what is read of this join's output (require())
  void require(const Names* need) override {
    out_need_all = need == nullptr; if (need) out_need = *need;
    if (!need) { left->require(nullptr); right->require(nullptr); return; }
    Names n = *need;
    for (auto& o : on.a) { collect_columns(o.at("left"), n); collect_columns(o.at("right"), n); }
    if (has_filter) collect_columns(filter, n);
    left->require(&n); right->require(&n);
  }
  int partitions() override { return right->partitions(); }
  PSchema schema() override {      // build_join_schema [UPSTREAM-KNOWLEDGE: datafusion joins/utils.rs]: left ++ right, the non-preserved side nullable
    PSchema l = left->schema(), r = right->schema();
    const std::string& jt = join_type;
    if (jt == "LeftSemi" || jt == "LeftAnti") return l;
    if (jt == "RightSemi" || jt == "RightAnti") return r;
    if (jt == "Right" || jt == "Full") for (auto& c : l) c.nullable = true;
    if (jt == "Left" || jt == "Full") for (auto& c : r) c.nullable = true;
    l.insert(l.end(), r.begin(), r.end());
    return l;
  }
  struct Side { PTable t; bool has_pred = false; Json pred; };
  Side side(PNode* plan, int part, Exec& x) {
    Fused f = fuse(plan);
    Side s;
    if (f.has_map) { s.t = plan->execute(part, x); return s; }      // computed projection below the join: materialised, not fused
    s.t = f.src->execute(part, x); s.has_pred = f.has_pred; s.pred = f.pred;
    return s;
  }
  int64_t last_pairs = -1;      // pairs the last synchronous run of this call site emitted: sizes the pair vectors of a deferred run
  PTable join_view(Exec& x, const PTable& lt, const PTable& rt, const uint32_t* ob, const uint32_t* opb, int64_t k, const BufP& ob_own, const BufP& opb_own, const PTable* count = nullptr) {
    PTable lv = select_view(x, lt, ob, k, ob_own, count), rv = select_view(x, rt, opb, k, opb_own, count);
    if (lv.via.size() + rv.via.size() > 3) {
      if (lv.via.size() >= rv.via.size()) lv = materialize(x, lv); else rv = materialize(x, rv);
    }
    PTable out; out.n = k; out.own(lv); out.own(rv);
    if (count) out.count_from(*count);
    out.cols = lv.cols; out.via = lv.via; out.sides = lv.sides;
    const int shift = (int)lv.via.size();
    for (size_t i = 0; i < rv.cols.size(); ++i) { out.cols.push_back(rv.cols[i]); out.sides.push_back(rv.sides[i] == 0 ? 0 : rv.sides[i] + shift); }
    for (auto v : rv.via) out.via.push_back(v);
    return out;
  }
  // positions of [0, n) that occur (want_marked) / do not occur in rows[0..k)
  int64_t marked_rows(Exec& x, const uint32_t* rows, int64_t k, int64_t n, bool want_marked, int tag, BufP& out) {
    const size_t nb = (size_t)((n + 63) / 64) * 8 + 8;
    BufP bits = dev_alloc(nb);
    HIPCHECK(hipMemsetAsync(bits->p, 0, nb, (hipStream_t)x.stream));
    check(x, gpuq_mark_rows(x.ctx, x.stream, rows, k, (uint8_t*)bits->p));
    PTable t; t.n = n; t.keep.push_back(bits);
    PCol c; c.name = "m"; c.type = jstr("Boolean"); c.nullable = false; c.c.type = T_BOOL; c.c.repr = GPUQ_REPR_ARROW; c.c.data = bits->p; c.c.length = n;
    t.cols.push_back(c); t.sides.push_back(0);
    const Json m = jcol("m", 0);
    return filter_sel(x, t, want_marked ? m : jobj({{"not_expr", jobj({{"expr", m}})}}), this, tag, out);
  }
  // JoinFilter on a non-inner join: `pairs` are the Inner matches; a pair that fails the filter is no match
  PTable residual_join(Exec& x, const PTable& lt, const PTable& rt, const PTable& pairs, const BufP& ob, const BufP& opb, int64_t k) {
    hipStream_t s = (hipStream_t)x.stream;
    BufP sel; const int64_t k2 = filter_sel(x, pairs, filter, this, 2, sel);
    std::vector<BufP> keep{ob, opb, sel};
    const uint32_t* ob2 = k2 ? take_u32(x, (const uint32_t*)ob->p, k, (const uint32_t*)sel->p, k2, keep) : (const uint32_t*)ob->p;
    const uint32_t* opb2 = k2 ? take_u32(x, (const uint32_t*)opb->p, k, (const uint32_t*)sel->p, k2, keep) : (const uint32_t*)opb->p;
    const std::string& jt = join_type;
    auto with_keep = [&](PTable t) { for (auto& b : keep) t.keep.push_back(b); return t; };
    if (jt == "LeftSemi" || jt == "LeftAnti") { BufP rows; const int64_t m = marked_rows(x, ob2, k2, lt.n, jt == "LeftSemi", 5, rows); return with_keep(select_view(x, lt, (const uint32_t*)rows->p, m, rows)); }
    if (jt == "RightSemi" || jt == "RightAnti") { BufP rows; const int64_t m = marked_rows(x, opb2, k2, rt.n, jt == "RightSemi", 6, rows); return with_keep(select_view(x, rt, (const uint32_t*)rows->p, m, rows)); }
    BufP lrows, rrows; int64_t ml = 0, mr = 0;
    if (jt == "Left" || jt == "Full") ml = marked_rows(x, ob2, k2, lt.n, false, 5, lrows);
    if (jt == "Right" || jt == "Full") mr = marked_rows(x, opb2, k2, rt.n, false, 6, rrows);
    const int64_t k3 = k2 + ml + mr;
    BufP ob3 = dev_alloc((size_t)std::max<int64_t>(k3, 1) * 4 + 16), opb3 = dev_alloc((size_t)std::max<int64_t>(k3, 1) * 4 + 16);
    uint32_t* a = (uint32_t*)ob3->p; uint32_t* b = (uint32_t*)opb3->p;
    if (k2) { HIPCHECK(hipMemcpyAsync(a, ob2, (size_t)k2 * 4, hipMemcpyDeviceToDevice, s)); HIPCHECK(hipMemcpyAsync(b, opb2, (size_t)k2 * 4, hipMemcpyDeviceToDevice, s)); }
    if (ml) { HIPCHECK(hipMemcpyAsync(a + k2, lrows->p, (size_t)ml * 4, hipMemcpyDeviceToDevice, s)); HIPCHECK(hipMemsetAsync(b + k2, 0xFF, (size_t)ml * 4, s)); }
    if (mr) { HIPCHECK(hipMemsetAsync(a + k2 + ml, 0xFF, (size_t)mr * 4, s)); HIPCHECK(hipMemcpyAsync(b + k2 + ml, rrows->p, (size_t)mr * 4, hipMemcpyDeviceToDevice, s)); }
    HIPCHECK(hipStreamSynchronize(s));      // the sources of the copies die with this frame
    return join_view(x, lt, rt, a, b, k3, ob3, opb3);
  }
  PTable execute(int part, Exec& x) override {
    // The two inputs are executed ONCE; what may run twice is the join over the tables they produced (a retry that went back to the
    // children re-entered the exchanges below them on this rank alone, re-read shuffle files and counted the children's metrics twice).
    Side L, R;
    HashJoinExec* inner = chain_inner();
    if (inner) {
      // (A |x| B) |x| C: this join's build side is built straight from B (see chain_build); the inner join's own output is never formed
      const int lpart = partition_mode == "Partitioned" ? part : 0;
      Side Li, Ri; inner->sides(lpart, x, Li, Ri);
      R = side(right.get(), part, x);
      auto t0 = std::chrono::steady_clock::now();
      Chain ch;
      if (chain_build(x, inner, Li, Ri, ch)) {
        struct Guard { gpuq_join_table* t; ~Guard() { if (t) gpuq_join_table_free(t); } } guard{ch.table};
        Side Lv; Lv.t = ch.view;
        return timed(x, t0, join_sides(x, Lv, R, false, &ch));
      }
      // the inner build side holds duplicate keys (or a layout the fused form does not take): the two-step form, over the same inputs
      auto t1 = std::chrono::steady_clock::now();
      L.t = inner->timed(x, t1, inner->join_sides(x, Li, Ri, false));
    } else sides(part, x, L, R);
    auto t0 = std::chrono::steady_clock::now();
    try { return timed(x, t0, join_sides(x, L, R, false)); }
    catch (const Unsupported& e) { if (!is_long_string_failure(e)) throw; }
    // a join key holds strings of more than 15 bytes: again, with those key columns replaced by exact dictionary codes (the
    // build side fills the dictionary, the probe side is looked up in it; a probe string that is not in it gets no code = no match)
    resolve(x, L.t); resolve(x, R.t);
    PTable out;
    try { out = join_sides(x, L, R, true); }
    catch (const Unsupported& e) {
      if (!is_long_string_failure(e) || !(L.has_pred || R.has_pred)) throw;
      // it was a filter fused into a side that compared long strings: apply the filters on their own (filter_table lowers such comparisons)
      if (L.has_pred) { L.t = filter_table(x, L.t, L.pred, this, 41); L.has_pred = false; }
      if (R.has_pred) { R.t = filter_table(x, R.t, R.pred, this, 42); R.has_pred = false; }
      strip_code_columns(L.t); strip_code_columns(R.t);
      out = join_sides(x, L, R, true);
    }
    strip_code_columns(out);
    return timed(x, t0, out);
  }
  void sides(int part, Exec& x, Side& L, Side& R) {
    const int lpart = partition_mode == "Partitioned" ? part : 0;
    if (partition_mode != "Partitioned" && left->partitions() != 1) {
      // CollectLeft: the build side is ALL partitions of the left input, collected into one table (what DataFusion's collect_left_input
      // does before it builds); each partition runs with its own filters / projections applied
      std::vector<PTable> in; for (int p = 0; p < left->partitions(); ++p) in.push_back(left->execute(p, x));
      L.t = concat_tables(x, std::move(in));
    } else L = side(left.get(), lpart, x);
    R = side(right.get(), part, x);
  }
  // ---- chain fusion (include/gpuq.h gpuq_join_build_run_semi).  This join is the outer one of (A |x| B) |x| C: its left input is an
  // Inner join without a JoinFilter, and its own build keys are columns of B.  When A's keys turn out unique, B's rows that find
  // their key in A's table ARE this join's build side, so the build kernel looks them up on the way: one pass over B instead of
  // probe + pair emit + compaction + a build through an index vector (SF100 q3: the orders side 0.83 + 0.08 + 0.71 ms -> one kernel).
  struct Chain { gpuq_join_table* table = nullptr; PTable view; bool deferred = false; };
  static bool is_plain_column(const Json& e, std::string& name) {
    if (!(e.is_obj() && e.o.size() == 1 && e.o[0].first == "column" && e.o[0].second.is_obj() && e.o[0].second.find("name"))) return false;
    name = e.o[0].second.at("name").str(); return true;
  }
  int chain_checked = 0;      // 0 not looked at yet, 1 candidate, -1 no
  HashJoinExec* chain_inner() {
    static const bool on_env = []() { const char* e = getenv("GPUQ_JOIN_CHAIN"); return !(e && e[0] == '0'); }();
    if (!on_env || chain_checked < 0) return nullptr;
    PNode* n = left.get();
    while (auto* p = dynamic_cast<PassThrough*>(n)) n = p->input.get();
    auto* in = dynamic_cast<HashJoinExec*>(n);
    if (chain_checked > 0) return in;
    chain_checked = -1;
    if (!in || has_filter || null_equals_null || in->has_filter || in->null_equals_null || in->join_type != "Inner") return nullptr;
    if (!(join_type == "Inner" || join_type == "Right" || join_type == "RightSemi" || join_type == "RightAnti")) return nullptr;
    if (partition_mode != "Partitioned" && left->partitions() != 1) return nullptr;
    // this join's build keys must be plain columns of the inner join's probe side (B), under names its build side (A) does not use
    std::vector<std::string> ln, rn;
    for (auto& f : in->left->schema()) ln.push_back(f.name);
    for (auto& f : in->right->schema()) rn.push_back(f.name);
    for (auto& o : on.a) {
      std::string nm;
      if (!is_plain_column(o.at("left"), nm)) return nullptr;
      if (std::count(rn.begin(), rn.end(), nm) != 1 || std::count(ln.begin(), ln.end(), nm) != 0) return nullptr;
    }
    chain_checked = 1;
    return in;
  }
  bool chain_build(Exec& x, HashJoinExec* in, Side& Li, Side& Ri, Chain& ch) {
    if (Ri.t.via.size() > 2) return false;      // B may bring up to two index vectors
    if (Li.t.is_view()) {
      // A's rows are addressed by position: a view (the output of a join below: q5's ASIA customers) is brought to plain columns first,
      // only the ones somebody reads (its keys, and what flows up)
      if (!in->out_need_all) {
        Names keep = in->out_need;
        for (auto& o : in->on.a) collect_columns(o.at("left"), keep);
        if (Li.has_pred) collect_columns(Li.pred, keep);
        prune_columns(Li.t, keep);
      }
      if (Li.has_pred) { Li.t = filter_table(x, Li.t, Li.pred, in, 13); Li.has_pred = false; }
      Li.t = materialize(x, Li.t);
    }
    // 1. A's table, exactly as the inner join would build it
    gpuq_op* bop1 = cached_op(x, in, 0, table_sig(Li.t), [&]() {
      const auto ln = names_of(Li.t);
      Json lk = jarr(); for (auto& o : in->on.a) lk.a.push_back(rebind(o.at("left"), ln));
      std::vector<std::pair<std::string, Json>> bd = {{"op", jstr("join_build")}, {"input", jobj({{"fields", table_fields(Li.t)}})}, {"on", lk}, {"null_equals_null", jbool(false)},
                                                     {"build_side_rows", jbool(false)}, {"label", jstr(in->label("build"))}};
      if (Li.has_pred) bd.push_back({"predicate", rebind(Li.pred, ln)});
      return jobj(bd);
    });
    // 2. this join's table from B's rows that are in A's table
    gpuq_op* bop2 = cached_op(x, this, 12, table_sig(Ri.t), [&]() {
      const auto rn = names_of(Ri.t);
      Json lk = jarr(), sk = jarr();
      for (auto& o : on.a) lk.a.push_back(rebind(o.at("left"), rn));
      for (auto& o : in->on.a) sk.a.push_back(rebind(o.at("right"), rn));
      std::vector<std::pair<std::string, Json>> bd = {{"op", jstr("join_build")}, {"input", jobj({{"fields", table_fields(Ri.t)}})}, {"on", lk}, {"semi_on", sk},
                                                     {"null_equals_null", jbool(false)}, {"build_side_rows", jbool(false)}, {"label", jstr(label("chain_build"))}};
      if (Ri.has_pred) bd.push_back({"predicate", rebind(Ri.pred, rn)});
      return jobj(bd);
    });
    bool deferred = x.deferred && last_pairs >= 0 && gpuq_op_can_defer(bop1) && gpuq_op_can_defer(bop2);
    if (deferred) deferred = use_deferred(x, bop1) && use_deferred(x, bop2);
    if (!deferred) { use_sync(x, bop1); use_sync(x, bop2); resolve(x, Li.t); resolve(x, Ri.t); }
    InputC lic, ric; make_input(Li.t, lic); make_input(Ri.t, ric);
    gpuq_join_table* t1 = nullptr;
    check(x, gpuq_join_build_run(bop1, x.stream, &lic.in, 0, Li.t.n, &t1));
    struct Guard { gpuq_join_table* t; ~Guard() { if (t) gpuq_join_table_free(t); } } g1{t1};
    if (!deferred) x.host_syncs += 2;
    if (gpuq_join_table_has_duplicates(t1)) return false;
    // A's row of every surviving position of B (A's columns are read through it) -- not formed at all when nothing above the inner
    // join reads a column of A (its keys have done their work in A's table): two scattered accesses per surviving row less
    bool want_hits = in->out_need_all;
    for (auto& c : Li.t.cols) want_hits = want_hits || in->out_need.count(c.name) > 0;
    BufP hits = want_hits ? dev_alloc((size_t)std::max<int64_t>(Ri.t.n, 1) * 4 + 16) : nullptr, rows = dev_alloc(16);
    gpuq_join_table* t2 = nullptr;
    check(x, gpuq_join_build_run_semi(bop2, x.stream, &ric.in, 0, Ri.t.n, t1, want_hits ? (uint32_t*)hits->p : nullptr, (uint64_t*)rows->p, &t2));
    ch.table = t2; ch.deferred = deferred;
    // the inner join's output_rows: the survivors (booked with the bound and corrected at the settle when nothing is read back)
    if (deferred) { in->m.output_rows += Ri.t.n; x.fixes.push_back({(const uint64_t*)rows->p, Ri.t.n, &in->m.output_rows, rows}); }
    else { in->m.output_rows += (int64_t)read_u64(x, rows->p); x.host_syncs += 2; }
    // the build side as a table: B's rows by position, A's columns through `hits`
    PTable v; v.n = Ri.t.n; v.count_from(Ri.t); v.own(Ri.t);
    if (want_hits) {
      v.own(Li.t); v.keep.push_back(hits);
      v.via.push_back((const uint32_t*)hits->p);
      for (auto p : Ri.t.via) v.via.push_back(p);
      for (size_t i = 0; i < Li.t.cols.size(); ++i) { v.cols.push_back(Li.t.cols[i]); v.sides.push_back(1); }
      for (size_t i = 0; i < Ri.t.cols.size(); ++i) { v.cols.push_back(Ri.t.cols[i]); v.sides.push_back(Ri.t.sides[i] == 0 ? 0 : Ri.t.sides[i] + 1); }
    } else { v.via = Ri.t.via; v.cols = Ri.t.cols; v.sides = Ri.t.sides; v.dense = Ri.t.dense; }      // B alone: A's columns are not part of what flows up
    ch.view = v;
    return true;
  }
  PTable join_sides(Exec& x, Side L, Side R, const bool long_keys, Chain* chain = nullptr) {
    // columns nobody above this join reads are dropped before anything is gathered through the pair vectors (q5: the region / nation /
    // customer columns that only served the joins below; SF100: the materialisation of the orders-side view took 0.7 ms with them)
    if (!out_need_all) {
      Names keep = out_need;
      for (auto& o : on.a) { collect_columns(o.at("left"), keep); collect_columns(o.at("right"), keep); }
      if (has_filter) collect_columns(filter, keep);
      if (L.has_pred) collect_columns(L.pred, keep);
      if (R.has_pred) collect_columns(R.pred, keep);
      prune_columns(L.t, keep); prune_columns(R.t, keep);
    }
    const bool residual = has_filter && join_type != "Inner";
    const std::string jt = residual ? std::string("Inner") : join_type;
    if (residual) {      // rows that fail a side's own predicate are not part of the join at all: apply those first
      resolve(x, L.t); resolve(x, R.t);
      if (L.has_pred) { L.t = filter_table(x, L.t, L.pred, this, 3); L.has_pred = false; }
      if (R.has_pred) { R.t = filter_table(x, R.t, R.pred, this, 4); R.has_pred = false; }
      resolve(x, L.t); resolve(x, R.t);
    }
    Json on_eff = on;
    Utf8DictGuard dicts[8]; int nd = 0;
    if (long_keys) {
      if (null_equals_null) throw Unsupported("null_equals_null with Utf8 join keys longer than 15 bytes");
      for (size_t k = 0; k < on_eff.a.size() && nd < 8; ++k) {
        const int li = long_key_column(L.t, on.a[k].at("left")), ri = long_key_column(R.t, on.a[k].at("right"));
        if (li < 0 || ri < 0) continue;
        const std::string nm = "__code_" + std::to_string(k);
        check(x, gpuq_utf8_dict_create(x.ctx, x.stream, std::min<int64_t>(L.t.n, L.t.cols[(size_t)li].c.length), &dicts[nd].d));
        append_code_column(x, L.t, li, dicts[nd].d, true, nm + "l");
        append_code_column(x, R.t, ri, dicts[nd].d, false, nm + "r");
        ++nd;
        on_eff.a[k] = jobj({{"left", jobj({{"column", jobj({{"name", jstr(nm + "l")}})}})}, {"right", jobj({{"column", jobj({{"name", jstr(nm + "r")}})}})}});
      }
    }
    gpuq_op* bop = chain ? nullptr : cached_op(x, this, long_keys ? 7 : 0, table_sig(L.t), [&]() {
      const auto ln = names_of(L.t);
      Json lk = jarr();
      for (auto& o : on_eff.a) lk.a.push_back(rebind(o.at("left"), ln));
      // only Left / Full / LeftSemi / LeftAnti ask the table for its build side's rows afterwards
      const bool side_rows = jt == "Left" || jt == "Full" || jt == "LeftSemi" || jt == "LeftAnti";
      std::vector<std::pair<std::string, Json>> bd = {{"op", jstr("join_build")}, {"input", jobj({{"fields", table_fields(L.t)}})}, {"on", lk}, {"null_equals_null", jbool(null_equals_null)},
                                                     {"build_side_rows", jbool(side_rows)}, {"label", jstr(label("build"))}};
      if (L.has_pred) bd.push_back({"predicate", rebind(L.pred, ln)});
      return jobj(bd);
    });
    gpuq_op* pop = cached_op(x, this, long_keys ? 8 : 1, table_sig(R.t), [&]() {
      const auto rn = names_of(R.t);
      Json rk = jarr();
      for (auto& o : on_eff.a) rk.a.push_back(rebind(o.at("right"), rn));
      std::vector<std::pair<std::string, Json>> pd = {{"op", jstr("join_probe")}, {"input", jobj({{"fields", table_fields(R.t)}})}, {"on", rk}, {"join_type", jstr(jt)},
                                                     {"null_equals_null", jbool(null_equals_null)}, {"label", jstr(label("probe"))}};
      if (R.has_pred) pd.push_back({"predicate", rebind(R.pred, rn)});
      return jobj(pd);
    });
    // Deferred (a plan's second run on): the build keeps the table layout it remembers, the probe's pair count stays on the device and
    // sizes nothing on the host -- the pair vectors are as long as the count this call site produced last time allows (+ 1/8), an
    // overflow raises the probe's status word.  Only the join types whose output is the pair list itself run this way.
    const bool plain = !long_keys && !residual && !has_filter && (jt == "Inner" || jt == "Right" || jt == "RightSemi" || jt == "RightAnti");
    bool deferred = false;
    if (chain) deferred = chain->deferred && plain && use_deferred(x, pop);      // (the table is there already: chain_build)
    else if (plain && x.deferred && last_pairs >= 0 && gpuq_op_can_defer(bop)) deferred = use_deferred(x, bop) && use_deferred(x, pop);
    if (!deferred) { if (bop) use_sync(x, bop); use_sync(x, pop); resolve(x, L.t); resolve(x, R.t); }
    InputC lic, ric; make_input(L.t, lic); make_input(R.t, ric);
    gpuq_join_table* jtab = chain ? chain->table : nullptr;
    if (!chain) {
      check(x, gpuq_join_build_run(bop, x.stream, &lic.in, 0, L.t.n, &jtab));
      if (!deferred) x.host_syncs += 2;
    }
    struct Guard { gpuq_join_table* t; ~Guard() { if (t) gpuq_join_table_free(t); } } guard{chain ? nullptr : jtab};
    const bool lout = jt == "Left" || jt == "Full";
    const int64_t extra_cap = lout ? L.t.n : 0;
    int64_t cap = std::max<int64_t>(R.t.n, 1) + extra_cap;
    BufP cnt = dev_alloc(16), ob, opb; int64_t k = 0;
    if (deferred) {
      // (unique build keys: no more pairs than probe rows; duplicate keys can multiply them)
      cap = gpuq_join_table_has_duplicates(jtab) ? last_pairs + last_pairs / 8 + 4096 : std::min<int64_t>(cap, last_pairs + last_pairs / 8 + 4096);
      { static const bool trace = getenv("GPUQ_TRACE_DEFER") != nullptr; if (trace) fprintf(stderr, "[gpuq] deferred %s join: build bound %lld, probe bound %lld, pair capacity %lld (last run %lld pairs), dups %d\n", jt.c_str(), (long long)L.t.n, (long long)R.t.n, (long long)cap, (long long)last_pairs, gpuq_join_table_has_duplicates(jtab)); }
      ob = dev_alloc((size_t)cap * 4 + 16); opb = dev_alloc((size_t)cap * 4 + 16);
      const bool pairs = jt == "Inner" || jt == "Right";
      check(x, gpuq_join_probe_run(pop, x.stream, jtab, &ric.in, 0, pairs ? (uint32_t*)ob->p : nullptr, (uint32_t*)opb->p, (uint64_t)cap, (uint64_t*)cnt->p));
      PTable c; c.n_dev = (const uint64_t*)cnt->p; c.n_keep = cnt;
      if (!pairs) return select_view(x, R.t, (const uint32_t*)opb->p, cap, opb, &c);
      return join_view(x, L.t, R.t, (const uint32_t*)ob->p, (const uint32_t*)opb->p, cap, ob, opb, &c);
    }
    for (;;) {
      ob = dev_alloc((size_t)cap * 4 + 16); opb = dev_alloc((size_t)cap * 4 + 16);
      check(x, gpuq_join_probe_run(pop, x.stream, jtab, &ric.in, 0, (uint32_t*)ob->p, (uint32_t*)opb->p, (uint64_t)(cap - extra_cap), (uint64_t*)cnt->p));
      k = (int64_t)read_u64(x, cnt->p);
      const int rc = gpuq_op_check(pop, x.stream);
      ++x.host_syncs;
      if (rc == GPUQ_ERR_CAPACITY) { cap = k + extra_cap + 1; continue; }
      check(x, rc);
      break;
    }
    if (plain) last_pairs = k;
    if (jt == "LeftSemi" || jt == "LeftAnti" || lout) {
      BufP extra = dev_alloc(16);
      if (!lout) {
        BufP rows = dev_alloc((size_t)std::max<int64_t>(L.t.n, 1) * 4 + 16);
        check(x, gpuq_join_build_side_rows(jtab, x.stream, jt == "LeftSemi" ? 1 : 0, (uint32_t*)rows->p, (uint64_t*)extra->p));
        const int64_t mrows = (int64_t)read_u64(x, extra->p);
        return select_view(x, L.t, (const uint32_t*)rows->p, mrows, rows);
      }
      check(x, gpuq_join_build_side_rows(jtab, x.stream, 0, (uint32_t*)ob->p + k, (uint64_t*)extra->p));
      const int64_t mrows = (int64_t)read_u64(x, extra->p);
      if (mrows > 0) HIPCHECK(hipMemsetAsync((uint32_t*)opb->p + k, 0xFF, (size_t)mrows * 4, (hipStream_t)x.stream));     // NULL_ROW on the probe side
      k += mrows;
    }
    if (jt == "RightSemi" || jt == "RightAnti") return select_view(x, R.t, (const uint32_t*)opb->p, k, opb);
    PTable out = join_view(x, L.t, R.t, (const uint32_t*)ob->p, (const uint32_t*)opb->p, k, ob, opb);
    if (residual) return residual_join(x, L.t, R.t, out, ob, opb, k);
    if (has_filter) out = filter_table(x, out, filter, this, 2);
    // the probe kernels read the build table asynchronously; results were read back (synchronised) above
    return out;
  }
};

struct LimitExec : PNode {       // LocalLimitExec: first `fetch` rows of each partition (a prefix needs no index vector for plain tables)
  PNodeP input; int64_t fetch = 0;
  std::vector<PNode*> children() override { return {input.get()}; }
  PTable execute(int part, Exec& x) override {
    PTable t = input->execute(part, x);
    resolve(x, t);
    if (fetch < t.n) { t.n = fetch; for (auto& c : t.cols) if (!t.is_view() && c.c.length > fetch) c.c.length = fetch; }
    m.output_rows += t.n;
    return t;
  }
};

// ---------------------------------------------------------------- fan-in (coalesce_tasks.rs:130-229) and row ranges
// Concatenation in partition order.  Pieces are first brought to the fixed-width layout (materialize(force): no views,
// Utf8 as PACKED15); data buffers are joined with device copies, validity / Boolean bitmaps at bit granularity.
PTable concat_tables(Exec& x, std::vector<PTable> parts) {
  for (auto& p : parts) resolve(x, p);
  std::vector<PTable> live;
  for (auto& p : parts) if (p.n > 0) live.push_back(p);
  if (live.empty() && !parts.empty()) live.push_back(parts[0]);
  if (live.empty()) throw std::runtime_error("concat: no input partitions");
  if (live.size() == 1 && !live[0].is_view()) return live[0];
  for (auto& p : live) p = materialize(x, p);        // views -> plain columns; Arrow-layout strings stay as they are
  int64_t n = 0; for (auto& p : live) n += p.n;
  const size_t nc = live[0].cols.size();
  const size_t bm = (size_t)((n + 63) / 64) * 8 + 8;
  std::vector<size_t> doff(nc), voff(nc, 0), dbytes(nc);
  size_t off = 256;
  for (size_t i = 0; i < nc; ++i) {
    const PCol& c0 = live[0].cols[i];
    for (auto& p : live) if (p.cols[i].c.type != c0.c.type || p.cols[i].c.repr != c0.c.repr) throw std::runtime_error("concat: column '" + c0.name + "' has different layouts across partitions");
    DType dt; dt.id = c0.c.type; dt.p = c0.c.precision; dt.s = c0.c.scale;
    for (auto& p : live) if ((p.cols[i].c.offsets != nullptr) != (c0.c.offsets != nullptr)) throw std::runtime_error("concat: column '" + c0.name + "' has different layouts across partitions");
    if (c0.c.offsets) dbytes[i] = (size_t)(n + 4) * 4;          // Arrow-layout Utf8: this slot holds the joined offsets, the bytes get their own buffer
    else dbytes[i] = c0.c.type == T_BOOL ? bm : (size_t)std::max<int64_t>(n, 1) * (size_t)type_width(dt) + 16;
    doff[i] = off; off += (dbytes[i] + 255) & ~(size_t)255;
    bool any_valid = false; for (auto& p : live) any_valid = any_valid || p.cols[i].c.validity != nullptr;
    if (any_valid) { voff[i] = off; off += (bm + 255) & ~(size_t)255; }
  }
  BufP buf = dev_alloc(off);
  hipStream_t s = (hipStream_t)x.stream;
  PTable out; out.n = n; out.keep.push_back(buf); out.record_cap = std::max<int64_t>(n, 1);
  for (size_t i = 0; i < nc; ++i) {
    PCol c = live[0].cols[i];
    DType dt; dt.id = c.c.type; dt.p = c.c.precision; dt.s = c.c.scale;
    char* d = (char*)buf->p + doff[i];
    uint8_t* v = voff[i] ? (uint8_t*)buf->p + voff[i] : nullptr;
    if (c.c.type == T_BOOL) HIPCHECK(hipMemsetAsync(d, 0, bm, s));
    if (v) HIPCHECK(hipMemsetAsync(v, 0, bm, s));
    int64_t row = 0; bool nullable = false;
    if (c.c.offsets) {
      // first / last offset of every piece (one small read-back per piece), then re-based offsets and joined bytes
      std::vector<int32_t> first(live.size()), last(live.size()); int64_t total = 0;
      for (size_t q = 0; q < live.size(); ++q) {
        const PCol& pc = live[q].cols[i];
        HIPCHECK(hipMemcpyAsync(&x.pin[0], pc.c.offsets, 4, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync((char*)&x.pin[0] + 4, pc.c.offsets + live[q].n, 4, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        std::memcpy(&first[q], &x.pin[0], 4); std::memcpy(&last[q], (char*)&x.pin[0] + 4, 4);
        total += last[q] - first[q];
      }
      BufP bytes = dev_alloc((size_t)total + 16);
      int64_t base = 0;
      for (size_t q = 0; q < live.size(); ++q) {
        const PCol& pc = live[q].cols[i];
        nullable = nullable || pc.nullable;
        check(x, gpuq_offsets_rebase(x.ctx, x.stream, pc.c.offsets, live[q].n + 1, (int32_t)(base - first[q]), (int32_t*)d + row));
        if (last[q] > first[q]) HIPCHECK(hipMemcpyAsync((char*)bytes->p + base, (const char*)pc.c.data + first[q], (size_t)(last[q] - first[q]), hipMemcpyDeviceToDevice, s));
        if (v && live[q].n > 0) check(x, gpuq_copy_bits(x.ctx, x.stream, v, row, pc.c.validity, 0, live[q].n));
        row += live[q].n; base += last[q] - first[q];
      }
      c.nullable = nullable; c.c.offsets = (const int32_t*)d; c.c.data = bytes->p; c.c.validity = v; c.c.length = n;
      out.keep.push_back(bytes);
      out.cols.push_back(c); out.sides.push_back(0);
      continue;
    }
    for (auto& p : live) {
      const PCol& pc = p.cols[i];
      nullable = nullable || pc.nullable;
      if (p.n > 0) {
        if (c.c.type == T_BOOL) check(x, gpuq_copy_bits(x.ctx, x.stream, (uint8_t*)d, row, (const uint8_t*)pc.c.data, 0, p.n));
        else { const size_t w = (size_t)type_width(dt); HIPCHECK(hipMemcpyAsync(d + (size_t)row * w, pc.c.data, (size_t)p.n * w, hipMemcpyDeviceToDevice, s)); }
        if (v) check(x, gpuq_copy_bits(x.ctx, x.stream, v, row, pc.c.validity, 0, p.n));      // NULL source = all valid
      }
      row += p.n;
    }
    c.nullable = nullable; c.c.data = d; c.c.validity = v; c.c.offsets = nullptr; c.c.length = n;
    out.cols.push_back(c); out.sides.push_back(0);
  }
  for (auto& p : live) out.own(p);       // the copies above are asynchronous
  return out;
}

// rows [skip, skip + count) of a table
PTable slice_table(Exec& x, PTable t, int64_t skip, int64_t count) {
  resolve(x, t);
  skip = std::min(std::max<int64_t>(skip, 0), t.n);
  count = (count < 0 || skip + count > t.n) ? t.n - skip : count;
  if (skip == 0) { t.n = count; if (!t.is_view()) for (auto& c : t.cols) c.c.length = count; return t; }
  bool has0 = false; if (t.is_view()) for (int sd : t.sides) has0 = has0 || sd == 0;
  if (t.is_view() && !has0) { for (auto& v : t.via) v += skip; t.n = count; return t; }      // every column goes through an index vector
  t = materialize(x, t, true);
  hipStream_t s = (hipStream_t)x.stream;
  const size_t bm = (size_t)((count + 63) / 64) * 8 + 8;
  for (auto& c : t.cols) {
    DType dt; dt.id = c.c.type; dt.p = c.c.precision; dt.s = c.c.scale;
    if (c.c.type == T_BOOL) {
      BufP b = dev_alloc(bm); HIPCHECK(hipMemsetAsync(b->p, 0, bm, s));
      check(x, gpuq_copy_bits(x.ctx, x.stream, (uint8_t*)b->p, 0, (const uint8_t*)c.c.data, skip, count));
      c.c.data = b->p; t.keep.push_back(b);
    } else c.c.data = (const char*)c.c.data + (size_t)skip * (size_t)type_width(dt);
    if (c.c.validity) {
      BufP b = dev_alloc(bm); HIPCHECK(hipMemsetAsync(b->p, 0, bm, s));
      check(x, gpuq_copy_bits(x.ctx, x.stream, (uint8_t*)b->p, 0, c.c.validity, skip, count));
      c.c.validity = (const uint8_t*)b->p; t.keep.push_back(b);
    }
    c.c.length = count;
  }
  t.n = count; t.record_cap = 0;
  return t;
}

// CrossJoinExec (datafusion.proto:1382-1385): every row of the (collected) left input with every row of the right partition.  What
// DataFusion plans for an uncorrelated scalar subquery -- the one-row side of q11 / q15 / q22's threshold -- followed by a FilterExec.
struct CrossJoinExec : HashJoinExec {
  void require(const Names* need) override { left->require(need); right->require(need); }
  PSchema schema() override { PSchema l = left->schema(), r = right->schema(); l.insert(l.end(), r.begin(), r.end()); return l; }
  PTable execute(int part, Exec& x) override {
    std::vector<PTable> ls;
    for (int p = 0; p < left->partitions(); ++p) { ls.push_back(left->execute(p, x)); resolve(x, ls.back()); }
    PTable lt = ls.size() == 1 ? ls[0] : concat_tables(x, std::move(ls));
    PTable rt = right->execute(part, x); resolve(x, rt);
    auto t0 = std::chrono::steady_clock::now();
    const int64_t k = lt.n * rt.n;
    BufP ob = dev_alloc((size_t)std::max<int64_t>(k, 1) * 4 + 16), opb = dev_alloc((size_t)std::max<int64_t>(k, 1) * 4 + 16);
    check(x, gpuq_cross_pairs(x.ctx, x.stream, lt.n, rt.n, (uint32_t*)ob->p, (uint32_t*)opb->p));
    return timed(x, t0, join_view(x, lt, rt, (const uint32_t*)ob->p, (const uint32_t*)opb->p, k, ob, opb));
  }
};

struct UnionExec : PNode {       // output partitions = the inputs' partitions, one input after another (UNION ALL)
  std::vector<PNodeP> inputs;
  std::vector<PNode*> children() override { std::vector<PNode*> v; for (auto& i : inputs) v.push_back(i.get()); return v; }
  int partitions() override { int k = 0; for (auto& i : inputs) k += i->partitions(); return k; }
  PSchema schema() override {      // first input's names and types; a column is nullable when it is in any input
    PSchema s0 = inputs.at(0)->schema();
    for (size_t k = 1; k < inputs.size(); ++k) { const PSchema sk = inputs[k]->schema(); for (size_t i = 0; i < s0.size() && i < sk.size(); ++i) s0[i].nullable = s0[i].nullable || sk[i].nullable; }
    return s0;
  }
  PTable execute(int part, Exec& x) override {
    for (auto& i : inputs) { const int k = i->partitions(); if (part < k) { PTable t = i->execute(part, x); return timed(x, std::chrono::steady_clock::now(), t); } part -= k; }
    throw std::runtime_error("UnionExec: partition out of range");
  }
};

struct CoalesceExec : PNode {    // CoalesceTasksExec / CoalescePartitionsExec: P partitions -> 1, optionally preserving an order
  PNodeP input; std::vector<int> parts; bool all = false; bool ordered = false; Json order_by;
  std::vector<PNode*> children() override { return {input.get()}; }
  int partitions() override { return 1; }
  PTable execute(int, Exec& x) override {
    if (!all && parts.size() == 1) return input->execute(parts[0], x);          // coalesce_tasks.rs:143-145
    std::vector<PTable> in;
    // with an order the reference merges ALL input partitions (coalesce_tasks.rs:151), not only the listed ones
    if (all || ordered) for (int p = 0; p < input->partitions(); ++p) in.push_back(input->execute(p, x));
    else for (int p : parts) in.push_back(input->execute(p, x));
    std::vector<int64_t> offs{0};
    for (auto& t : in) { resolve(x, t); offs.push_back(offs.back() + t.n); }
    auto t0 = std::chrono::steady_clock::now();
    PTable out = concat_tables(x, std::move(in));
    // k-way merge (coalesce_tasks.rs:162-170): pairwise merge-path rounds, ties keep (partition, row) order
    if (ordered) out = merge_table(x, out, offs, order_by, -1, this, 0);
    return timed(x, t0, out);
  }
};

struct GlobalLimitExec : PNode {
  PNodeP input; int64_t skip = 0, fetch = -1;
  std::vector<PNode*> children() override { return {input.get()}; }
  int partitions() override { return 1; }
  PTable execute(int, Exec& x) override {
    if (input->partitions() != 1) throw std::runtime_error("GlobalLimitExec requires a single input partition");
    PTable t = input->execute(0, x);
    auto t0 = std::chrono::steady_clock::now();
    return timed(x, t0, slice_table(x, t, skip, fetch));
  }
};

PNodeP build_child(const Json& v, const char* key) { return build_node(v.at(key)); }

// ---------------------------------------------------------------- stage driver: shuffle sink and source
// ShuffleWriterExec (ballista/core/src/execution_plans/shuffle_writer.rs:234-456): runs the child for one input partition,
// optionally hash-partitions its output (BatchPartitioner call site :336-391) and writes one Arrow IPC stream with LZ4_FRAME
// buffers per non-empty output partition (:358-378; lazily created writers: an empty partition gets no file, :329-334), in the
// reference's path layout: <work_dir>/<job_id>/<stage_id>/<output partition>/<uuid>.arrow, or .../<stage_id>/<uuid>/data.arrow
// when the stage is not repartitioned (utils.rs:179-219).  The node's result is the reference's result batch
// (shuffle_writer.rs:470-520): one row per file -- partition, path, num_rows, num_batches, num_bytes.
void mkdirs(const std::string& path) {
  for (size_t i = 1; i <= path.size(); ++i)
    if (i == path.size() || path[i] == '/') {
      const std::string d = path.substr(0, i);
      if (::mkdir(d.c_str(), 0777) != 0 && errno != EEXIST) throw std::runtime_error("cannot create directory " + d + ": " + std::strerror(errno));
    }
}
std::string uuid4() {
  static thread_local std::mt19937_64 rng{std::random_device{}()};
  uint64_t a = rng(), b = rng();
  a = (a & 0xFFFFFFFFFFFF0FFFull) | 0x0000000000004000ull; b = (b & 0x3FFFFFFFFFFFFFFFull) | 0x8000000000000000ull;
  char buf[40];
  std::snprintf(buf, sizeof(buf), "%08x-%04x-%04x-%04x-%012llx", (unsigned)(a >> 32), (unsigned)((a >> 16) & 0xFFFF), (unsigned)(a & 0xFFFF), (unsigned)(b >> 48),
                (unsigned long long)(b & 0xFFFFFFFFFFFFull));
  return buf;
}
void ipc_check(int rc) {
  if (rc == GPUQ_OK) return;
  const std::string msg = gpuq_ipc_last_error();
  if (rc == GPUQ_ERR_UNSUPPORTED) throw Unsupported(msg);
  if (rc == GPUQ_ERR_CAPACITY) throw Capacity(msg);
  if (rc == GPUQ_ERR_HIP) throw HipError(msg);
  throw std::runtime_error(msg);
}
struct PinnedBuf {
  uint8_t* p = nullptr; size_t cap = 0;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  uint8_t* ensure(size_t n) { if (n > cap) { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; HIPCHECK(hipHostMalloc((void**)&p, n, hipHostMallocDefault)); cap = n; } return p; }
};
thread_local PinnedBuf g_sink_buf;

// plain table with every Utf8 column in Arrow layout (PACKED15 columns are unpacked)
PTable arrow_layout(Exec& x, const PTable& in_) {
  PTable in = in_; resolve(x, in);
  PTable t = materialize(x, in);
  for (auto& c : t.cols) {
    if (c.c.repr != GPUQ_REPR_PACKED15) continue;
    const int64_t n = t.n;
    BufP offs = dev_alloc((size_t)(n + 4) * 4), data = dev_alloc((size_t)n * 15 + 16);
    int64_t dl = 0;
    check(x, gpuq_unpack_utf8(x.ctx, x.stream, c.c.data, n, (int32_t*)offs->p, (uint8_t*)data->p, n * 15 + 16, &dl));
    c.c.repr = GPUQ_REPR_ARROW; c.c.data = data->p; c.c.offsets = (const int32_t*)offs->p;
    t.keep.push_back(offs); t.keep.push_back(data);
  }
  return t;
}

struct ShuffleFile { int64_t partition; std::string path; int64_t rows, batches, bytes; };

struct ShuffleWriterExec : PNode {
  PNodeP input; std::string job_id, work_dir; int64_t stage_id = 0; bool hashed = false; Json hash_expr; int64_t partition_count = 0; int64_t batch_rows = 1 << 20;
  int64_t write_ns = 0, repart_ns = 0, input_rows = 0;      // ShuffleWriteMetrics, shuffle_writer.rs:139-160
  std::vector<int64_t> stage_partitions;                    // shuffle_writer.rs:118-119
  std::vector<PNode*> children() override { return {input.get()}; }
  PSchema schema() override {      // the result batch of shuffle_writer.rs:470-520
    return {{"partition", jstr("UInt32"), false}, {"path", jstr("Utf8"), false}, {"num_rows", jstr("UInt64"), false}, {"num_batches", jstr("UInt64"), false}, {"num_bytes", jstr("UInt64"), false}};
  }

  ShuffleFile write_file(Exec& x, const PTable& view, int64_t partition, const std::string& path) {
    PTable t = arrow_layout(x, view);
    std::vector<gpuq_field_info> fields(t.cols.size());
    size_t raw = 0;
    for (size_t i = 0; i < t.cols.size(); ++i) {
      gpuq_field_info& f = fields[i]; f = gpuq_field_info{};
      std::snprintf(f.name, sizeof(f.name), "%s", t.cols[i].name.c_str());
      f.type = t.cols[i].c.type; f.precision = t.cols[i].c.precision; f.scale = t.cols[i].c.scale; f.nullable = t.cols[i].nullable; f.repr = GPUQ_REPR_ARROW;
    }
    mkdirs(path.substr(0, path.rfind('/')));
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("cannot create " + path + ": " + std::strerror(errno));
    ShuffleFile sf{partition, path, t.n, 0, 0};
    try {
      auto put = [&](const void* p, size_t n) { if (n && std::fwrite(p, 1, n, fp) != n) throw std::runtime_error("short write to " + path); sf.bytes += (int64_t)n; };
      int64_t len = 0;
      // a hash-partitioned file says which partition function placed its rows (GPUQ_PARTITION_FN_KEY, include/gpuq.h)
      const char* mk[1] = {GPUQ_PARTITION_FN_KEY}; const char* mv[1] = {GPUQ_PARTITION_FN};
      const int nkv = hashed ? 1 : 0;
      ipc_check(gpuq_ipc_schema_message_kv(fields.data(), (int)fields.size(), mk, mv, nkv, nullptr, 0, &len));
      std::vector<uint8_t> sm((size_t)len);
      ipc_check(gpuq_ipc_schema_message_kv(fields.data(), (int)fields.size(), mk, mv, nkv, sm.data(), len, &len));
      put(sm.data(), sm.size());
      const int64_t step = std::max<int64_t>(64, (batch_rows + 63) / 64 * 64);      // batches start on bitmap words: row ranges are pointer arithmetic
      // Utf8 byte counts of the whole partition, for the output bound of a batch
      for (int64_t lo = 0; lo < t.n; lo += step) {
        const int64_t k = std::min(step, t.n - lo);
        std::vector<gpuq_column> cols;
        raw = 4096;
        for (auto& c : t.cols) {
          gpuq_column v = c.c; v.length = k;
          DType dt; dt.id = v.type; dt.p = v.precision; dt.s = v.scale;
          if (v.validity) v.validity += lo / 8;
          if (v.type == T_UTF8) { v.offsets += lo; raw += (size_t)(k + 1) * 4 + 64; }
          else if (v.type == T_BOOL) { v.data = (const uint8_t*)v.data + lo / 8; raw += (size_t)k / 8 + 64; }
          else { v.data = (const char*)v.data + (size_t)lo * (size_t)type_width(dt); raw += (size_t)k * (size_t)type_width(dt) + 64; }
          raw += (size_t)k / 8 + 64;
          cols.push_back(v);
        }
        // string bytes: not known on the host; ask for the size only when the first guess (32 B per string) is too small
        size_t guess = raw; for (auto& c : t.cols) if (c.c.type == T_UTF8) guess += (size_t)k * 32;
        uint8_t* hb = g_sink_buf.ensure(guess);
        int rc = gpuq_ipc_encode_batch(x.ctx, x.stream, cols.data(), (int)cols.size(), k, 0, hb, (int64_t)g_sink_buf.cap, &len);
        if (rc == GPUQ_ERR_CAPACITY) { hb = g_sink_buf.ensure((size_t)len); rc = gpuq_ipc_encode_batch(x.ctx, x.stream, cols.data(), (int)cols.size(), k, 0, hb, (int64_t)g_sink_buf.cap, &len); }
        ipc_check(rc);
        put(hb, (size_t)len);
        sf.batches += 1;
      }
      static const uint8_t eos[8] = {0xFF, 0xFF, 0xFF, 0xFF, 0, 0, 0, 0};
      put(eos, 8);
    } catch (...) { std::fclose(fp); throw; }
    if (std::fclose(fp) != 0) throw std::runtime_error("cannot close " + path);
    return sf;
  }

  PTable execute(int part, Exec& x) override {
    auto t0 = std::chrono::steady_clock::now();
    PTable t = input->execute(part, x);
    resolve(x, t);      // files are written from exact counts (and only after everything deferred below has held)
    input_rows += t.n;
    const std::string base = work_dir + "/" + job_id + "/" + std::to_string(stage_id);
    std::vector<ShuffleFile> files;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ns = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
    if (!hashed) {
      auto tw = now();
      files.push_back(write_file(x, t, part, base + "/" + uuid4() + "/data.arrow"));
      write_ns += ns(tw, now());
    } else {
      auto tr = now();
      PSchema ps = plain_schema(t);
      std::vector<std::string> names; for (auto& f : ps) names.push_back(f.name);
      gpuq_op* op = cached_op(x, this, 0, table_sig(t), [&]() {
        Json he = jarr(); for (auto& e : hash_expr.a) he.a.push_back(rebind(e, names));
        return jobj({{"op", jstr("partition")}, {"input", jobj({{"fields", table_fields(t)}})}, {"hash_expr", he}, {"partition_count", jnum(partition_count)}});
      });
      BufP perm = dev_alloc((size_t)std::max<int64_t>(1, t.n) * 4), offs = dev_alloc((size_t)(partition_count + 2) * 8);
      InputC ic; make_input(t, ic);
      check(x, gpuq_partition_run(op, x.stream, &ic.in, (uint32_t*)perm->p, (uint64_t*)offs->p));
      std::vector<uint64_t> o((size_t)partition_count + 1);
      HIPCHECK(hipMemcpyAsync(o.data(), offs->p, o.size() * 8, hipMemcpyDeviceToHost, (hipStream_t)x.stream));
      HIPCHECK(hipStreamSynchronize((hipStream_t)x.stream));
      repart_ns += ns(tr, now());
      auto tw = now();
      for (int64_t q = 0; q < partition_count; ++q) {
        const int64_t cnt = (int64_t)(o[(size_t)q + 1] - o[(size_t)q]);
        if (cnt == 0) continue;
        PTable v = select_view(x, t, (const uint32_t*)perm->p + o[(size_t)q], cnt, perm);
        files.push_back(write_file(x, v, q, base + "/" + std::to_string(q) + "/" + uuid4() + ".arrow"));
      }
      write_ns += ns(tw, now());
    }
    // the result batch: partition, path, num_rows, num_batches, num_bytes
    const int64_t n = (int64_t)files.size();
    std::vector<uint32_t> pid; std::vector<int32_t> poff{0}; std::string pdata; std::vector<uint64_t> rows, batches, bytes;
    for (auto& f : files) { pid.push_back((uint32_t)f.partition); pdata += f.path; poff.push_back((int32_t)pdata.size()); rows.push_back((uint64_t)f.rows); batches.push_back((uint64_t)f.batches); bytes.push_back((uint64_t)f.bytes); }
    PTable out; out.n = n;
    auto up = [&](const void* src, size_t nbytes) -> BufP {
      BufP b = dev_alloc(nbytes + 16);
      if (nbytes) HIPCHECK(hipMemcpyAsync(b->p, src, nbytes, hipMemcpyHostToDevice, (hipStream_t)x.stream));
      out.keep.push_back(b); return b;
    };
    auto fixed = [&](const char* name, int tid, const void* src, size_t w) {
      PCol c; c.name = name; c.type = type_json_of(tid, 0, 0); c.nullable = false; c.c.type = tid; c.c.repr = GPUQ_REPR_ARROW; c.c.length = n; c.c.data = up(src, (size_t)n * w)->p;
      out.cols.push_back(c); out.sides.push_back(0);
    };
    fixed("partition", T_UINT32, pid.data(), 4);
    { PCol c; c.name = "path"; c.type = jstr("Utf8"); c.nullable = false; c.c.type = T_UTF8; c.c.repr = GPUQ_REPR_ARROW; c.c.length = n;
      c.c.offsets = (const int32_t*)up(poff.data(), poff.size() * 4)->p; c.c.data = up(pdata.data(), pdata.size())->p; out.cols.push_back(c); out.sides.push_back(0); }
    fixed("num_rows", T_UINT64, rows.data(), 8); fixed("num_batches", T_UINT64, batches.data(), 8); fixed("num_bytes", T_UINT64, bytes.data(), 8);
    HIPCHECK(hipStreamSynchronize((hipStream_t)x.stream));      // the uploads read host vectors that die with this frame
    m.elapsed_ns += ns(t0, now());
    for (auto& f : files) m.output_rows += f.rows;
    return out;
  }
};

// ShuffleReaderExec (ballista/core/src/execution_plans/shuffle_reader.rs:149-177): output partition p = the IPC streams of its
// locations, decoded on the device in one launch per file.  Local files only (a remote location is the Flight client's job);
// a missing file is the reference's FetchFailed (:654), which makes the scheduler re-run the map stage.
struct ShuffleReaderExec : PNode {
  PSchema schema_; std::vector<std::vector<std::string>> locations;
  PSchema schema() override { return schema_; }
  int partitions() override { return (int)locations.size(); }
  PTable execute(int part, Exec& x) override {
    auto t0 = std::chrono::steady_clock::now();
    if (part < 0 || part >= (int)locations.size()) throw std::runtime_error("ShuffleReaderExec: partition out of range");
    std::vector<gpuq_field_info> fields(schema_.size());
    for (size_t i = 0; i < schema_.size(); ++i) {
      gpuq_field_info& f = fields[i]; f = gpuq_field_info{};
      std::snprintf(f.name, sizeof(f.name), "%s", schema_[i].name.c_str());
      int p, s; f.type = type_id_of(schema_[i].type, p, s); f.precision = p; f.scale = s; f.nullable = schema_[i].nullable; f.repr = GPUQ_REPR_ARROW;
    }
    std::vector<PTable> parts;
    std::vector<uint8_t> bytes;
    std::string part_fn;
    auto wrap = [&](gpuq_ipc_batch* b) {
      PTable t; t.n = gpuq_ipc_batch_num_rows(b);
      t.keep.push_back(BufP(new DevBuf(), [b](DevBuf* d) { delete d; gpuq_ipc_batch_free(b); }));
      for (size_t i = 0; i < schema_.size(); ++i) {
        PCol c; c.name = schema_[i].name; c.type = schema_[i].type; c.nullable = schema_[i].nullable;
        gpuq_ipc_batch_column(b, (int)i, &c.c);
        t.cols.push_back(c); t.sides.push_back(0);
      }
      return t;
    };
    for (auto& path : locations[(size_t)part]) {
      FILE* fp = std::fopen(path.c_str(), "rb");
      if (!fp) throw std::runtime_error("FetchFailed: shuffle partition file " + path + " cannot be opened: " + std::strerror(errno));
      std::fseek(fp, 0, SEEK_END); const long sz = std::ftell(fp); std::fseek(fp, 0, SEEK_SET);
      bytes.resize((size_t)std::max<long>(sz, 0));
      const size_t got = bytes.empty() ? 0 : std::fread(bytes.data(), 1, bytes.size(), fp);
      std::fclose(fp);
      if (got != bytes.size()) throw std::runtime_error("short read from " + path);
      {      // every file of one shuffle partition must have been partitioned by the same function (include/gpuq.h)
        char val[64] = {0}; int found = 0;
        if (gpuq_ipc_schema_metadata(bytes.data(), (int64_t)bytes.size(), GPUQ_PARTITION_FN_KEY, val, sizeof(val), &found) == GPUQ_OK) {
          const std::string fn = found ? std::string(val) : std::string("datafusion-ahash (no " GPUQ_PARTITION_FN_KEY " metadata)");
          if (part_fn.empty()) part_fn = fn;
          else if (part_fn != fn)
            throw std::runtime_error("ShuffleReaderExec: the files of shuffle partition " + std::to_string(part) + " were written with different partition functions (" + part_fn + " vs " + fn +
                                     " in " + path + "): all map tasks of a hash-partitioned stage must run on the same engine");
        }
      }
      gpuq_ipc_batch* b = nullptr;
      ipc_check(gpuq_ipc_decode_stream(x.ctx, x.stream, bytes.data(), (int64_t)bytes.size(), fields.data(), (int)fields.size(), &b));
      parts.push_back(wrap(b));
    }
    if (parts.empty()) {
      static const uint8_t eos[8] = {0xFF, 0xFF, 0xFF, 0xFF, 0, 0, 0, 0};
      gpuq_ipc_batch* b = nullptr;
      ipc_check(gpuq_ipc_decode_stream(x.ctx, x.stream, eos, 8, fields.data(), (int)fields.size(), &b));
      parts.push_back(wrap(b));
    }
    return timed(x, t0, parts.size() == 1 ? parts[0] : concat_tables(x, std::move(parts)));
  }
};

// ---------------------------------------------------------------- exchange between the GPUs of a node (exchange.cpp)
// The intra-node stand-ins for a stage boundary of the reference (ShuffleWriterExec -> files -> ShuffleReaderExec):
//   RepartitionExec {input, hash_expr, partition_count == ranks}: this rank's rows are hash-partitioned by destination rank
//     (same mix64 partition function on every rank), grouped with one take, and exchanged; the result is what every rank
//     sent here (RepartitionExecNode, datafusion.proto; the planner splits stages at exactly this node, planner.rs:137-151);
//   BroadcastExec {input}: every rank receives all ranks' rows -- a CollectLeft build side whose reduce task reads every
//     partition of the build stage (shuffle_reader.rs: all locations of the stage).
// What comes out of a collective call is the same on every rank (or the communicator is gone): nothing is left to announce.
void xcheck(int rc) {
  if (rc == GPUQ_OK) return;
  const std::string msg = gpuq_exchange_last_error();
  if (rc == GPUQ_ERR_RETRY) throw AgreedRetry(msg);
  throw AgreedFailure(msg, rc);
}
// The part of an exchange node that runs BEFORE its collective (the child plan, the grouping by destination): when it fails, the
// peers are -- or will be -- waiting in that collective's meta round.  They are told (status 1: failed; 2 while executing
// deferred: "everybody runs again, synchronously") through the same fixed-size round, and only then does the failure travel up.
template <class F> auto announce_failures(Exec& x, F&& local_part) -> decltype(local_part()) {
  try { return local_part(); }
  catch (const AgreedRetry&) { throw; }
  catch (const AgreedFailure&) { throw; }
  catch (const Cancelled&) { throw; }
  catch (const std::exception& e) {
    (void)gpuq_comm_set_status(x.comm, x.deferred ? 2 : 1);
    (void)gpuq_comm_announce(x.comm, x.stream);
    if (x.deferred) throw AgreedRetry(e.what());
    int rc = GPUQ_ERR_INVALID;
    if (dynamic_cast<const Unsupported*>(&e)) rc = GPUQ_ERR_UNSUPPORTED; else if (dynamic_cast<const Capacity*>(&e)) rc = GPUQ_ERR_CAPACITY; else if (dynamic_cast<const HipError*>(&e)) rc = GPUQ_ERR_HIP;
    throw AgreedFailure(e.what(), rc);
  }
}
void table_c_arrays(const PTable& t, std::vector<gpuq_column>& cols, std::vector<gpuq_field_info>& fields) {
  cols.clear(); fields.clear();
  for (auto& c : t.cols) {
    gpuq_field_info f{};
    std::snprintf(f.name, sizeof(f.name), "%s", c.name.c_str());
    f.type = c.c.type; f.precision = c.c.precision; f.scale = c.c.scale; f.nullable = c.nullable; f.repr = c.c.repr;
    cols.push_back(c.c); fields.push_back(f);
  }
}
PTable table_from_owned(gpuq_table* tab, const PTable& like) {
  PTable out; out.n = gpuq_table_num_rows(tab);
  out.keep.push_back(BufP(new DevBuf(), [tab](DevBuf* d) { delete d; gpuq_table_free(tab); }));
  for (size_t i = 0; i < like.cols.size(); ++i) {
    PCol c; c.name = like.cols[i].name; c.type = like.cols[i].type; c.nullable = like.cols[i].nullable;
    gpuq_table_column(tab, (int)i, &c.c, nullptr);
    out.cols.push_back(c); out.sides.push_back(0);
  }
  return out;
}
struct RepartitionExec : PNode {
  PNodeP input; Json hash_expr; int64_t partition_count = 0;
  std::vector<PNode*> children() override { return {input.get()}; }
  bool need_all = true; Names need;      // columns somebody above reads: the others do not cross the links
  void require(const Names* n) override {
    need_all = n == nullptr; if (n) { need = *n; for (auto& e : hash_expr.a) collect_columns(e, need); }
    input->require(n ? &need : nullptr);
  }
  int partitions() override { return input->partitions(); }
  PTable execute(int part, Exec& x) override {
    if (!x.comm) throw Unsupported("RepartitionExec inside a stage needs the ranks of the node (gpuq_plan_set_comm); without them the reference's planner splits the stage here");
    const int W = gpuq_comm_world(x.comm);
    if (partition_count != W) throw Unsupported("RepartitionExec: partition_count " + std::to_string(partition_count) + " != number of ranks " + std::to_string(W));
    std::chrono::steady_clock::time_point t0;
    PSchema ps; PTable grouped; std::vector<int64_t> doff;
    std::vector<gpuq_column> cols; std::vector<gpuq_field_info> fields;
    announce_failures(x, [&]() {
    PTable t = input->execute(part, x);
    resolve(x, t);      // (a settle: everything deferred below this exchange is looked at before a row leaves the rank)
    if (!need_all) prune_columns(t, need);
    t0 = std::chrono::steady_clock::now();
    ps = plain_schema(t);
    std::vector<std::string> names; for (auto& f : ps) names.push_back(f.name);
    gpuq_op* op = cached_op(x, this, 0, table_sig(t), [&]() {
      Json he = jarr(); for (auto& e : hash_expr.a) he.a.push_back(rebind(e, names));
      return jobj({{"op", jstr("partition")}, {"input", jobj({{"fields", table_fields(t)}})}, {"hash_expr", he}, {"partition_count", jnum(partition_count)}});
    });
    BufP perm = dev_alloc((size_t)std::max<int64_t>(1, t.n) * 4), offs = dev_alloc((size_t)(partition_count + 2) * 8);
    InputC ic; make_input(t, ic);
    check(x, gpuq_partition_run(op, x.stream, &ic.in, (uint32_t*)perm->p, (uint64_t*)offs->p));
    std::vector<uint64_t> o((size_t)partition_count + 1);
    HIPCHECK(hipMemcpyAsync(o.data(), offs->p, o.size() * 8, hipMemcpyDeviceToHost, (hipStream_t)x.stream));
    HIPCHECK(hipStreamSynchronize((hipStream_t)x.stream));
    ++x.host_syncs;
    // one take groups the rows by destination rank
    grouped = materialize(x, select_view(x, t, (const uint32_t*)perm->p, t.n, perm));
    resolve(x, grouped);
    table_c_arrays(grouped, cols, fields);
    // a column read through an index vector may carry NULLs: nullability as the plan sees it (identical on every rank)
    for (size_t i = 0; i < fields.size(); ++i) fields[i].nullable = ps[i].nullable ? 1 : 0;
    doff.assign(o.begin(), o.end());
    return 0;
    });
    gpuq_table* tab = nullptr;
    xcheck(gpuq_exchange_partitions(x.comm, x.stream, cols.data(), fields.data(), (int)cols.size(), doff.data(), &tab));
    PTable out = table_from_owned(tab, grouped);
    for (size_t i = 0; i < out.cols.size(); ++i) out.cols[i].nullable = ps[i].nullable;
    return timed(x, t0, out);
  }
};
struct BroadcastExec : PNode {
  PNodeP input;
  std::vector<PNode*> children() override { return {input.get()}; }
  bool need_all = true; Names need;
  void require(const Names* n) override { need_all = n == nullptr; if (n) need = *n; input->require(n); }
  int partitions() override { return input->partitions(); }
  PTable execute(int part, Exec& x) override {
    if (!x.comm) throw Unsupported("BroadcastExec needs the ranks of the node (gpuq_plan_set_comm)");
    std::chrono::steady_clock::time_point t0;
    PSchema ps; PTable plain;
    std::vector<gpuq_column> cols; std::vector<gpuq_field_info> fields;
    announce_failures(x, [&]() {
    PTable t = input->execute(part, x);
    resolve(x, t);
    if (!need_all) prune_columns(t, need);
    t0 = std::chrono::steady_clock::now();
    ps = plain_schema(t);
    plain = materialize(x, t);
    resolve(x, plain);
    table_c_arrays(plain, cols, fields);
    for (size_t i = 0; i < fields.size(); ++i) fields[i].nullable = ps[i].nullable ? 1 : 0;
    return 0;
    });
    gpuq_table* tab = nullptr;
    xcheck(gpuq_allgather_table(x.comm, x.stream, cols.data(), fields.data(), (int)cols.size(), plain.n, &tab));
    PTable out = table_from_owned(tab, plain);
    for (size_t i = 0; i < out.cols.size(); ++i) out.cols[i].nullable = ps[i].nullable;
    return timed(x, t0, out);
  }
};

// Distributed SortExec over the ranks of a node (SURVEY.md section 8e "Sort", BASELINE configs[4]: "range-partition over xGMI"): what the
// reference runs as a single-partition SortPreservingMergeExec stage (ballista/scheduler/src/planner.rs:120-136) becomes
//   1. every rank contributes a strided sample of its rows; the gathered samples are sorted (identically on every rank: same rows,
//      same stable sort) and world-1 of them become the splitters;
//   2. the local rows and the splitters are sorted TOGETHER by the engine's own SortExec (a tag column puts a splitter behind the rows
//      it equals): where the splitters land are the range boundaries of the locally sorted run -- no key is ever compared outside the
//      device sort, so every type / direction / NULL placement SortExec knows works here;
//   3. range r of every rank goes to rank r in ONE exchange; the received pieces are sorted runs and are MERGED (gpuq_merge_run).
// Output: this rank's range, sorted; ranks hold ascending, non-overlapping ranges (rank order = global order).
struct RangeRepartitionExec : PNode {
  PNodeP input; Json expr; int64_t partition_count = 0, samples = 1024;
  std::vector<PNode*> children() override { return {input.get()}; }
  int partitions() override { return input->partitions(); }
  static PCol tag_column(Exec& x, int64_t n, int32_t value, std::vector<BufP>& keep) {
    BufP b = dev_alloc((size_t)std::max<int64_t>(n, 1) * 4 + 16);
    if (value == 0) HIPCHECK(hipMemsetAsync(b->p, 0, (size_t)std::max<int64_t>(n, 1) * 4, (hipStream_t)x.stream));
    else { std::vector<int32_t> h((size_t)std::max<int64_t>(n, 1), value); HIPCHECK(hipMemcpyAsync(b->p, h.data(), h.size() * 4, hipMemcpyHostToDevice, (hipStream_t)x.stream)); HIPCHECK(hipStreamSynchronize((hipStream_t)x.stream)); }
    keep.push_back(b);
    PCol c; c.name = "__gpuq_splitter"; c.type = jstr("Int32"); c.nullable = false;
    c.c.type = T_INT32; c.c.repr = GPUQ_REPR_ARROW; c.c.data = b->p; c.c.length = n;
    return c;
  }
  static BufP upload_u32(Exec& x, const std::vector<uint32_t>& v) {
    BufP b = dev_alloc(std::max<size_t>(v.size(), 1) * 4 + 16);
    if (!v.empty()) { HIPCHECK(hipMemcpyAsync(b->p, v.data(), v.size() * 4, hipMemcpyHostToDevice, (hipStream_t)x.stream)); HIPCHECK(hipStreamSynchronize((hipStream_t)x.stream)); }
    return b;
  }
  PTable execute(int part, Exec& x) override {
    if (!x.comm) throw Unsupported("RangeRepartitionExec needs the ranks of the node (gpuq_plan_set_comm)");
    const int W = gpuq_comm_world(x.comm);
    if (partition_count != W) throw Unsupported("RangeRepartitionExec: partition_count " + std::to_string(partition_count) + " != number of ranks " + std::to_string(W));
    std::chrono::steady_clock::time_point t0;
    PTable t, samp; PSchema ps;
    std::vector<gpuq_column> cols; std::vector<gpuq_field_info> fields;
    // ---- 1. sample
    announce_failures(x, [&]() {
      t = input->execute(part, x);
      resolve(x, t);
      t0 = std::chrono::steady_clock::now();
      ps = plain_schema(t);
      t = materialize(x, t, true);      // fixed-width layout: the pieces below are concatenated and sliced
      const int64_t k = std::min<int64_t>(t.n, samples), stride = std::max<int64_t>(1, t.n / std::max<int64_t>(k, 1));
      std::vector<uint32_t> idx((size_t)k); for (int64_t i = 0; i < k; ++i) idx[(size_t)i] = (uint32_t)(i * stride);
      BufP di = upload_u32(x, idx);
      samp = materialize(x, select_view(x, t, (const uint32_t*)di->p, k, di), true);
      table_c_arrays(samp, cols, fields);
      for (size_t i = 0; i < fields.size(); ++i) fields[i].nullable = ps[i].nullable ? 1 : 0;
      return 0;
    });
    gpuq_table* gathered = nullptr;
    xcheck(gpuq_allgather_table(x.comm, x.stream, cols.data(), fields.data(), (int)cols.size(), samp.n, &gathered));
    // ---- 2. splitters -> range boundaries of the locally sorted run
    PTable grouped; std::vector<int64_t> doff;
    announce_failures(x, [&]() {
      PTable all = table_from_owned(gathered, samp);
      for (size_t i = 0; i < all.cols.size(); ++i) all.cols[i].nullable = ps[i].nullable;
      PTable ssorted = materialize(x, sort_table(x, all, expr, -1, this, 0), true);
      const int64_t m = ssorted.n;
      std::vector<uint32_t> cut;
      if (m > 0) for (int j = 1; j < W; ++j) cut.push_back((uint32_t)std::min<int64_t>(m - 1, std::max<int64_t>(0, (int64_t)j * m / W)));
      std::vector<PTable> parts;
      { PTable a = t; for (auto& c : a.cols) c.nullable = true; a.cols.push_back(tag_column(x, a.n, 0, a.keep)); a.sides.push_back(0); a.record_cap = 0; parts.push_back(a); }
      if (!cut.empty()) {
        BufP dc = upload_u32(x, cut);
        PTable spl = materialize(x, select_view(x, ssorted, (const uint32_t*)dc->p, (int64_t)cut.size(), dc), true);
        for (auto& c : spl.cols) c.nullable = true;
        spl.cols.push_back(tag_column(x, spl.n, 1, spl.keep)); spl.sides.push_back(0); spl.record_cap = 0;
        parts.push_back(spl);
      }
      PTable both = parts.size() == 1 ? parts[0] : concat_tables(x, parts);
      Json keys = expr; keys.a.push_back(jobj({{"expr", jobj({{"column", jobj({{"name", jstr("__gpuq_splitter")}})}})}, {"asc", jbool(true)}, {"nulls_first", jbool(false)}}));
      PTable run = sort_table(x, both, keys, -1, this, 1);
      // positions of the splitters in the sorted run: filter on the tag, read the W-1 positions back
      const Json tagc = jobj({{"column", jobj({{"name", jstr("__gpuq_splitter")}})}});
      auto lit_i32 = [](int v) { return jobj({{"literal", jobj({{"type", jstr("Int32")}, {"value", jstr(std::to_string(v))}})}}); };
      PTable run_m = materialize(x, run);
      BufP sel; const int64_t ns = filter_sel(x, run_m, jobj({{"binary_expr", jobj({{"l", tagc}, {"r", lit_i32(1)}, {"op", jstr("=")}})}}), this, 2, sel);
      std::vector<uint32_t> pos((size_t)ns);
      if (ns > 0) { HIPCHECK(hipMemcpyAsync(pos.data(), sel->p, (size_t)ns * 4, hipMemcpyDeviceToHost, (hipStream_t)x.stream)); HIPCHECK(hipStreamSynchronize((hipStream_t)x.stream)); }
      doff.assign(1, 0);
      for (int64_t j = 0; j < ns; ++j) doff.push_back((int64_t)pos[(size_t)j] - j);      // boundaries in the run with the splitters taken out
      while ((int)doff.size() < W) doff.push_back(t.n);                                  // (fewer splitters than ranks: empty ranges at the end)
      doff.push_back(t.n);
      PTable rows_only = filter_table(x, run_m, jobj({{"binary_expr", jobj({{"l", tagc}, {"r", lit_i32(0)}, {"op", jstr("=")}})}}), this, 3);
      rows_only.cols.pop_back(); rows_only.sides.pop_back();
      grouped = materialize(x, rows_only, true);
      resolve(x, grouped);
      table_c_arrays(grouped, cols, fields);
      for (size_t i = 0; i < fields.size(); ++i) fields[i].nullable = ps[i].nullable ? 1 : 0;
      return 0;
    });
    // ---- 3. one exchange of ranges, then an ordered fan-in of the received runs
    gpuq_table* tab = nullptr;
    xcheck(gpuq_exchange_partitions(x.comm, x.stream, cols.data(), fields.data(), (int)cols.size(), doff.data(), &tab));
    PTable mine = table_from_owned(tab, grouped);
    for (size_t i = 0; i < mine.cols.size(); ++i) mine.cols[i].nullable = ps[i].nullable;
    std::vector<int64_t> pr((size_t)W, 0); int np = 0;
    xcheck(gpuq_table_piece_rows(tab, pr.data(), W, &np));
    std::vector<int64_t> offs{0}; for (int i = 0; i < np; ++i) offs.push_back(offs.back() + pr[(size_t)i]);
    return timed(x, t0, merge_table(x, mine, offs, expr, -1, this, 4));
  }
};

PNodeP build_node(const Json& j) {
  if (!j.is_obj() || j.o.size() != 1) throw std::runtime_error("plan: a node is an object with one key (the node type): " + j.dump().substr(0, 80));
  const std::string& kind = j.o[0].first; const Json& v = j.o[0].second;
  PNodeP out;
  if (kind == "MemoryExec") {
    auto n = std::make_unique<MemoryExec>();
    for (auto& f : v.at("schema").a) n->schema_.push_back({f.at("name").str(), f.at("type"), f.get_bool("nullable", true)});
    for (auto& p : v.at("partitions").a) n->parts.push_back((int)p.i64());
    n->dense = v.get_bool("dense", false);
    if (v.has("sides")) for (auto& s : v.at("sides").a) n->sides.push_back((int)s.i64());
    out = std::move(n);
  } else if (kind == "CoalesceBatchesExec") {
    auto n = std::make_unique<PassThrough>(); n->input = build_child(v, "input"); out = std::move(n);
  } else if (kind == "FilterExec") {
    auto n = std::make_unique<FilterExec>(); n->input = build_child(v, "input"); n->predicate = v.at("expr"); out = std::move(n);
  } else if (kind == "ProjectionExec") {
    auto n = std::make_unique<ProjectionExec>(); n->input = build_child(v, "input");
    const Json& e = v.at("expr"); const Json& names = v.at("expr_name");
    if (e.a.size() != names.a.size()) throw std::runtime_error("ProjectionExec: expr and expr_name differ in length");
    for (size_t i = 0; i < e.a.size(); ++i) { n->exprs.push_back(e.a[i]); n->names.push_back(names.a[i].str()); }
    out = std::move(n);
  } else if (kind == "AggregateExec") {
    auto n = std::make_unique<AggregateExec>(); n->input = build_child(v, "input"); n->mode = v.get_str("mode", "Single"); n->strategy = v.get_str("strategy", "auto");
    n->group_expr = v.has("group_expr") ? v.at("group_expr") : jarr(); n->aggr_expr = v.at("aggr_expr");
    n->expected_groups = v.get_i64("expected_groups", 0); n->output_capacity = v.get_i64("output_capacity", 0);
    // agg(DISTINCT x) (AggregateExprNode.distinct; benchmarks/queries/q16.sql:5 count(distinct ps_suppkey)): when every aggregate of the
    // node is DISTINCT over the same argument the node becomes two -- GROUP BY (keys, x) throws the duplicates away, GROUP BY keys
    // aggregates what is left -- which is the rewrite DataFusion's SingleDistinctToGroupBy rule makes [UPSTREAM-KNOWLEDGE]; both levels
    // run on the existing hash aggregate.  Mixed DISTINCT / plain aggregates stay refused (gpuq_op_create says so).
    bool any_distinct = false, all_distinct = !n->aggr_expr.a.empty();
    for (auto& a : n->aggr_expr.a) { const bool d = a.get_bool("distinct", false); any_distinct = any_distinct || d; all_distinct = all_distinct && d; }
    if (any_distinct && all_distinct && n->mode == "Single") {
      const Json& arg = n->aggr_expr.a[0].at("expr");
      bool same = true;
      for (auto& a : n->aggr_expr.a) same = same && a.has("expr") && a.at("expr").dump() == arg.dump() && !a.has("filter") && !a.has("expr2");
      if (same && (int)n->group_expr.a.size() + 1 <= 4) {
        auto inner = std::make_unique<AggregateExec>(); inner->input = std::move(n->input); inner->mode = "Single"; inner->strategy = n->strategy;
        inner->group_expr = n->group_expr;
        inner->group_expr.a.push_back(jobj({{"expr", arg}, {"name", jstr("__distinct")}}));
        inner->aggr_expr = jarr({jobj({{"fn", jstr("COUNT")}, {"expr", jobj({{"literal", jobj({{"type", jstr("Int64")}, {"value", jstr("1")}})}})}, {"name", jstr("__n")}})});
        Json outer_groups = jarr(), outer_aggs = jarr();
        for (auto& g : n->group_expr.a) outer_groups.a.push_back(jobj({{"expr", jobj({{"column", jobj({{"name", g.at("name")}})}})}, {"name", g.at("name")}}));
        for (auto& a : n->aggr_expr.a) outer_aggs.a.push_back(jobj({{"fn", a.at("fn")}, {"expr", jobj({{"column", jobj({{"name", jstr("__distinct")}})}})}, {"name", a.at("name")}}));
        n->input = std::move(inner); n->group_expr = outer_groups; n->aggr_expr = outer_aggs;
      }
    }
    out = std::move(n);
  } else if (kind == "SortExec" || kind == "SortPreservingMergeExec") {
    auto n = std::make_unique<SortExec>(); n->input = build_child(v, "input"); n->expr = v.at("expr"); n->fetch = v.get_i64("fetch", -1); n->merge_all = kind != "SortExec";
    out = std::move(n);
  } else if (kind == "HashJoinExec") {
    auto n = std::make_unique<HashJoinExec>(); n->left = build_child(v, "left"); n->right = build_child(v, "right"); n->on = v.at("on");
    n->join_type = v.get_str("join_type", "Inner"); n->partition_mode = v.get_str("partition_mode", "CollectLeft"); n->null_equals_null = v.get_bool("null_equals_null", false);
    if (v.has("filter")) { n->has_filter = true; n->filter = v.at("filter"); }
    out = std::move(n);
  } else if (kind == "CrossJoinExec") {
    auto n = std::make_unique<CrossJoinExec>(); n->left = build_child(v, "left"); n->right = build_child(v, "right"); n->on = jarr(); out = std::move(n);
  } else if (kind == "UnionExec") {
    auto n = std::make_unique<UnionExec>(); for (auto& i : v.at("inputs").a) n->inputs.push_back(build_node(i)); out = std::move(n);
  } else if (kind == "CoalescePartitionsExec") {
    auto n = std::make_unique<CoalesceExec>(); n->input = build_child(v, "input"); n->all = true; out = std::move(n);
  } else if (kind == "CoalesceTasksExec") {
    auto n = std::make_unique<CoalesceExec>(); n->input = build_child(v, "input");
    for (auto& p : v.at("partitions").a) n->parts.push_back((int)p.i64());
    if (v.has("order_by")) { n->ordered = true; n->order_by = v.at("order_by"); }
    out = std::move(n);
  } else if (kind == "GlobalLimitExec") {
    auto n = std::make_unique<GlobalLimitExec>(); n->input = build_child(v, "input"); n->skip = v.get_i64("skip", 0); n->fetch = v.get_i64("fetch", -1); out = std::move(n);
  } else if (kind == "LocalLimitExec") {
    auto n = std::make_unique<LimitExec>(); n->input = build_child(v, "input"); n->fetch = v.at("fetch").i64(); out = std::move(n);
  } else if (kind == "RepartitionExec") {
    auto n = std::make_unique<RepartitionExec>(); n->input = build_child(v, "input"); n->hash_expr = v.at("hash_expr"); n->partition_count = v.at("partition_count").i64(); out = std::move(n);
  } else if (kind == "BroadcastExec") {
    auto n = std::make_unique<BroadcastExec>(); n->input = build_child(v, "input"); out = std::move(n);
  } else if (kind == "RangeRepartitionExec") {
    auto n = std::make_unique<RangeRepartitionExec>(); n->input = build_child(v, "input"); n->expr = v.at("expr"); n->partition_count = v.at("partition_count").i64();
    n->samples = v.get_i64("samples", 1024); out = std::move(n);
  } else if (kind == "ShuffleWriterExec") {
    auto n = std::make_unique<ShuffleWriterExec>(); n->input = build_child(v, "input");
    n->job_id = v.at("job_id").str(); n->stage_id = v.at("stage_id").i64(); n->work_dir = v.at("work_dir").str(); n->batch_rows = v.get_i64("batch_rows", 1 << 20);
    // the stage partitions this task executes (shuffle_writer.rs:118-119): the child's CoalesceTasksExec does the work, the writer
    // only reports them back in every ShuffleWritePartition (:319, :411) -- echoed in gpuq_plan_metrics for the shim to fill in
    if (v.has("partitions")) for (auto& e : v.at("partitions").a) { if (e.i64() < 0) throw std::runtime_error("ShuffleWriterExec: negative stage partition"); n->stage_partitions.push_back(e.i64()); }
    if (n->work_dir.empty()) throw std::runtime_error("ShuffleWriterExec: work_dir is empty (the decoded plan carries \"\"; the engine sets the executor's, serde/mod.rs:191)");
    if (v.has("output_partitioning")) {
      const Json& op = v.at("output_partitioning");
      n->hashed = true; n->hash_expr = op.at("hash_expr"); n->partition_count = op.at("partition_count").i64();
      if (n->partition_count < 1) throw std::runtime_error("ShuffleWriterExec: partition_count must be >= 1");
    }
    out = std::move(n);
  } else if (kind == "ShuffleReaderExec") {
    auto n = std::make_unique<ShuffleReaderExec>();
    for (auto& f : v.at("schema").a) n->schema_.push_back({f.at("name").str(), f.at("type"), f.get_bool("nullable", true)});
    for (auto& p : v.at("partition").a) { n->locations.emplace_back(); for (auto& l : p.a) n->locations.back().push_back(l.is_obj() ? l.at("path").str() : l.str()); }
    out = std::move(n);
  } else throw Unsupported("plan: node type '" + kind + "' is not executed natively");
  out->kind = kind;
  return out;
}

void collect(PNode* n, std::vector<PNode*>& out) { out.push_back(n); for (PNode* c : n->children()) collect(c, out); }

thread_local std::string g_plan_error;

}  // namespace

struct gpuq_plan {
  gpuq_ctx* ctx = nullptr; PNodeP root; std::map<std::string, gpuq_op*> ops, memo; uint64_t* pin = nullptr; gpuq_comm* comm = nullptr;
  // deferred execution: allowed once an execution has completed synchronously (the operators then remember what they need); switched
  // off for good after three deferred executions had to be redone (inputs that change from call to call)
  int completed = 0, retries = 0; bool defer_ok = true;
  int last_settles = 0, last_host_syncs = 0, last_deferred = 0;      // of the last execution (gpuq_plan_exec_stats)
  ~gpuq_plan() { for (auto& kv : ops) gpuq_op_free(kv.second); if (pin) (void)hipHostFree(pin); }
};
struct gpuq_result { PTable t; std::vector<gpuq_field_info> fields; };

namespace {
template <class F> int plan_guarded(F&& f) {
  try { f(); return GPUQ_OK; }
  catch (const Cancelled& e) { g_plan_error = e.what(); return GPUQ_ERR_CANCELLED; }
  catch (const AgreedFailure& e) { g_plan_error = e.what(); return e.rc; }
  catch (const HipError& e) { g_plan_error = e.what(); return GPUQ_ERR_HIP; }
  catch (const Unsupported& e) { g_plan_error = e.what(); return GPUQ_ERR_UNSUPPORTED; }
  catch (const Capacity& e) { g_plan_error = e.what(); return GPUQ_ERR_CAPACITY; }
  catch (const std::bad_alloc&) { g_plan_error = "out of host memory"; return GPUQ_ERR_INTERNAL; }
  catch (const std::exception& e) { g_plan_error = e.what(); return GPUQ_ERR_INVALID; }
}
}  // namespace

extern "C" {

const char* gpuq_plan_last_error(void) { return g_plan_error.c_str(); }

int gpuq_plan_create(gpuq_ctx* ctx, const char* plan_json, gpuq_plan** out) {
  if (!out) return GPUQ_ERR_INVALID;
  *out = nullptr;
  return plan_guarded([&]() {
    // ctx == NULL: a plan for validation only (gpuq_plan_schema / gpuq_plan_num_partitions work without a device; a
    // scheduler-side process can check and type a stage plan) -- gpuq_plan_execute then fails
    if (!plan_json) throw std::runtime_error("plan_json is NULL");
    std::unique_ptr<gpuq_plan> p(new gpuq_plan());
    p->ctx = ctx;
    p->root = build_node(lower_long_string_eq(JsonParser(plan_json).parse()));
    p->root->require(nullptr);
    int next_id = 0;
    std::function<void(PNode*)> number = [&](PNode* n) { n->id = next_id++; for (PNode* c : n->children()) number(c); };
    number(p->root.get());
    *out = p.release();
  });
}

void gpuq_plan_free(gpuq_plan* p) { delete p; }

int gpuq_plan_num_partitions(gpuq_plan* p) { return p ? p->root->partitions() : 0; }

static int plan_execute_impl(gpuq_plan* p, void* stream, int partition, const gpuq_input* inputs, int n_inputs, const std::atomic<int>* cancel, gpuq_result** out) {
  *out = nullptr;
  return plan_guarded([&]() {
    if (!p->ctx) throw std::runtime_error("this plan was created without a context (validation only): it cannot execute");
    if (!p->pin) HIPCHECK(hipHostMalloc((void**)&p->pin, 64, hipHostMallocDefault));
    (void)use_stream(stream);
    Exec x; x.ctx = p->ctx; x.stream = stream; x.ops = &p->ops; x.inputs = inputs; x.n_inputs = n_inputs; x.pin = p->pin; x.memo = &p->memo; x.comm = p->comm; x.cancel = cancel;
    static const bool defer_env = []() { const char* e = getenv("GPUQ_DEFER"); return !(e && e[0] == '0'); }();
    // a plan with exchanges inside runs synchronously: a rank that had to redo a deferred execution would re-enter the collectives alone
    // With the ranks of a node attached, "redo it synchronously" is a decision all ranks take together: it travels in the status word of
    // the exchanges' meta rounds (gpuq_comm_set_status: 2) and a deferred execution ends with one such round that says "nobody has to".
    x.deferred = defer_env && p->defer_ok && p->completed > 0;
    PTable t;
    auto run = [&]() {
      t = materialize(x, p->root->execute(partition, x)); check_cancel(x); settle(x, &t);
      if (x.comm && x.deferred) xcheck(gpuq_comm_announce(x.comm, x.stream));
    };
    auto drain = [&]() {
      // whatever was queued keeps running: drain it before the buffers it uses go back to the pool (unwinding frees them); status
      // words deferred runs may have raised are cleared with it
      (void)hipStreamSynchronize((hipStream_t)stream);
      if (!x.pending.empty()) { std::vector<uint64_t> none(1); (void)gpuq_ops_settle(x.ctx, stream, x.pending.data(), (int)x.pending.size(), nullptr, 0, none.data()); }
      for (gpuq_op* op : x.pending) gpuq_op_set_deferred(op, 0);
      x.pending.clear(); x.fixes.clear();
    };
    const bool was_deferred = x.deferred;
    try { run(); }
    catch (const Cancelled&) { drain(); throw; }
    catch (const std::exception& e) {
      static const bool trace = getenv("GPUQ_TRACE_DEFER") != nullptr;
      if (trace && x.deferred) fprintf(stderr, "[gpuq] deferred execution redone synchronously: %s\n", e.what());
      drain();
      if (!x.deferred) throw;
      // the peers are waiting in (or heading for) a collective: tell them this rank starts over, unless they already know
      if (x.comm && !dynamic_cast<const AgreedRetry*>(&e)) {
        if (dynamic_cast<const AgreedFailure*>(&e)) throw;      // (a failed collective: every rank has the error, nobody starts over)
        (void)gpuq_comm_set_status(x.comm, 2); (void)gpuq_comm_announce(x.comm, x.stream);
      }
      // a deferred execution that did not hold (or failed in any other way): the same plan again, synchronously -- that run either
      // succeeds and refreshes what the operators remember, or raises the error with its proper message
      if (++p->retries >= 3) p->defer_ok = false;
      x.deferred = false; t = PTable();
      try { run(); } catch (...) { drain(); throw; }
    }
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    ++p->completed;
    p->last_settles = x.settles; p->last_host_syncs = x.host_syncs; p->last_deferred = was_deferred && x.deferred ? 1 : 0;
    std::unique_ptr<gpuq_result> r(new gpuq_result());
    for (auto& c : t.cols) {
      gpuq_field_info f{};
      std::snprintf(f.name, sizeof(f.name), "%s", c.name.c_str());
      f.type = c.c.type; f.precision = c.c.precision; f.scale = c.c.scale; f.nullable = c.nullable; f.repr = c.c.repr;
      DType dt; dt.id = c.c.type; dt.p = c.c.precision; dt.s = c.c.scale;
      f.width = c.c.type == T_BOOL ? 0 : type_width(dt);
      r->fields.push_back(f);
    }
    // inputs are the caller's: only library-owned buffers are kept
    r->t = std::move(t);
    *out = r.release();
  });
}

int gpuq_plan_execute(gpuq_plan* p, void* stream, int partition, const gpuq_input* inputs, int n_inputs, gpuq_result** out) {
  if (!p || !out) return GPUQ_ERR_INVALID;
  return plan_execute_impl(p, stream, partition, inputs, n_inputs, nullptr, out);
}

// ---- asynchronous execution + cancellation.  The reference runs a task as a future on its task-runner pool and cancels it by
// dropping the future at an await point (ballista/executor/src/executor.rs:201-240: no callback; whatever holds resources cleans
// up in Drop).  Here a task is a worker thread running the plan on the caller's stream; cancel raises a flag the executor checks
// between operator calls, the worker then drains the stream and releases every pooled buffer it held.
struct gpuq_task {
  gpuq_plan* plan = nullptr; void* stream = nullptr; int partition = 0;
  std::vector<gpuq_input> inputs; std::vector<std::vector<gpuq_column>> cols;      // private copies: the caller's arrays may go away
  std::thread th; std::atomic<int> cancel{0}, done{0};
  int rc = GPUQ_OK; std::string err; gpuq_result* res = nullptr;
};

int gpuq_plan_execute_async(gpuq_plan* p, void* stream, int partition, const gpuq_input* inputs, int n_inputs, gpuq_task** out) {
  if (!p || !out || n_inputs < 0 || (n_inputs > 0 && !inputs)) return GPUQ_ERR_INVALID;
  *out = nullptr;
  return plan_guarded([&]() {
    std::unique_ptr<gpuq_task> t(new gpuq_task());
    t->plan = p; t->stream = stream; t->partition = partition;
    t->inputs.assign(inputs, inputs + n_inputs); t->cols.resize((size_t)n_inputs);
    for (int i = 0; i < n_inputs; ++i) {
      if (inputs[i].n_cols > 0 && inputs[i].cols) t->cols[(size_t)i].assign(inputs[i].cols, inputs[i].cols + inputs[i].n_cols);
      t->inputs[(size_t)i].cols = t->cols[(size_t)i].data();
    }
    gpuq_task* raw = t.get();
    int dev = 0; (void)hipGetDevice(&dev);      // the caller's current device (the context's: every context call makes it current)
    t->th = std::thread([raw, dev]() {
      (void)hipSetDevice(dev);
      raw->rc = plan_execute_impl(raw->plan, raw->stream, raw->partition, raw->inputs.data(), (int)raw->inputs.size(), &raw->cancel, &raw->res);
      if (raw->rc != GPUQ_OK) raw->err = g_plan_error;      // the error text is per thread: carry it to whoever waits
      raw->done.store(1, std::memory_order_release);
    });
    *out = t.release();
  });
}
int gpuq_task_poll(gpuq_task* t, int* done_out) { if (!t || !done_out) return GPUQ_ERR_INVALID; *done_out = t->done.load(std::memory_order_acquire); return GPUQ_OK; }
int gpuq_task_cancel(gpuq_task* t) { if (!t) return GPUQ_ERR_INVALID; t->cancel.store(1, std::memory_order_relaxed); return GPUQ_OK; }
int gpuq_task_wait(gpuq_task* t, gpuq_result** out) {
  if (!t) return GPUQ_ERR_INVALID;
  if (t->th.joinable()) t->th.join();
  if (out) { *out = t->res; t->res = nullptr; }
  if (t->rc != GPUQ_OK) g_plan_error = t->err;
  return t->rc;
}
void gpuq_task_free(gpuq_task* t) {
  if (!t) return;
  if (t->th.joinable()) { t->cancel.store(1, std::memory_order_relaxed); t->th.join(); }      // dropping a running task cancels it, as dropping the future does
  if (t->res) gpuq_result_free(t->res);
  delete t;
}

int gpuq_result_record(const gpuq_result* r, void** base_out, size_t* bytes_out, int64_t* cap_out) {
  if (!r || !base_out || !bytes_out || !cap_out) return GPUQ_ERR_INVALID;
  *base_out = nullptr; *bytes_out = 0; *cap_out = 0;
  if (r->t.record_cap > 0 && r->t.keep.size() == 1 && !r->t.is_view()) { *base_out = r->t.keep[0]->p; *bytes_out = r->t.keep[0]->cap; *cap_out = r->t.record_cap; }
  return GPUQ_OK;
}
int64_t gpuq_result_num_rows(const gpuq_result* r) { return r ? r->t.n : 0; }
int gpuq_result_num_columns(const gpuq_result* r) { return r ? (int)r->t.cols.size() : 0; }
int gpuq_result_column(const gpuq_result* r, int i, gpuq_column* col_out, gpuq_field_info* field_out) {
  if (!r || i < 0 || i >= (int)r->t.cols.size()) return GPUQ_ERR_INVALID;
  if (col_out) { *col_out = r->t.cols[(size_t)i].c; col_out->length = r->t.n; }
  if (field_out) *field_out = r->fields[(size_t)i];
  return GPUQ_OK;
}
void gpuq_result_free(gpuq_result* r) { delete r; }

// Device time of the dominant kernel of the plan's operators (gpuq_op_profile on every compiled operator): reports the
// operator that accumulated the most kernel time since profiling was enabled.
int gpuq_plan_schema(gpuq_plan* p, gpuq_field_info* fields_out, int cap, int* n_out) {
  if (!p || !n_out) return GPUQ_ERR_INVALID;
  return plan_guarded([&]() {
    const PSchema s = p->root->schema();
    *n_out = (int)s.size();
    if (!fields_out || cap < (int)s.size()) { if (!fields_out && cap == 0) return; throw Capacity("plan schema has " + std::to_string(s.size()) + " fields"); }
    for (size_t i = 0; i < s.size(); ++i) {
      gpuq_field_info f{};
      std::snprintf(f.name, sizeof(f.name), "%s", s[i].name.c_str());
      int pr = 0, sc = 0; f.type = type_id_of(s[i].type, pr, sc); f.precision = pr; f.scale = sc; f.nullable = s[i].nullable ? 1 : 0;
      f.repr = GPUQ_REPR_ARROW;
      DType dt; dt.id = f.type; dt.p = pr; dt.s = sc; f.width = (f.type == T_BOOL) ? 0 : type_width(dt);
      fields_out[i] = f;
    }
  });
}

int gpuq_plan_exec_stats(gpuq_plan* p, int* deferred_out, int* settles_out, int* host_syncs_out, int* retries_out) {
  if (!p) return GPUQ_ERR_INVALID;
  if (deferred_out) *deferred_out = p->last_deferred;
  if (settles_out) *settles_out = p->last_settles;
  if (host_syncs_out) *host_syncs_out = p->last_host_syncs;
  if (retries_out) *retries_out = p->retries;
  return GPUQ_OK;
}
int gpuq_plan_set_comm(gpuq_plan* p, gpuq_comm* comm) { if (!p) return GPUQ_ERR_INVALID; p->comm = comm; return GPUQ_OK; }

int gpuq_plan_profile(gpuq_plan* p, int enable, float* kernel_ms_out, int* launches_out, char* op_desc_out, size_t cap) {
  if (!p) return GPUQ_ERR_INVALID;
  float best = -1; int bl = 0; std::string bd;
  for (auto& kv : p->ops) {
    float ms = 0; int n = 0;
    if (gpuq_op_profile(kv.second, enable, &ms, &n) != GPUQ_OK) continue;
    if (ms > best) { best = ms; bl = n; bd = kv.first; }
  }
  if (kernel_ms_out) *kernel_ms_out = best < 0 ? 0 : best;
  if (launches_out) *launches_out = bl;
  if (op_desc_out && cap) { std::snprintf(op_desc_out, cap, "%s", bd.c_str()); }
  return GPUQ_OK;
}

int gpuq_plan_profile_all(gpuq_plan* p, char* json_out, size_t cap) {
  if (!p || !json_out || !cap) return GPUQ_ERR_INVALID;
  Json arr = jarr();
  for (auto& kv : p->ops) {
    float ms = 0; int n = 0;
    if (gpuq_op_profile(kv.second, -1, &ms, &n) != GPUQ_OK) continue;
    std::string kind;
    try { kind = JsonParser(kv.first.c_str()).parse().get_str("op", ""); } catch (const std::exception&) {}
    Json num; num.kind = Json::NUM; { char b[64]; std::snprintf(b, sizeof(b), "%.6f", (double)ms); num.s = b; }
    float tot = 0; (void)gpuq_op_profile_total(kv.second, &tot);
    Json tnum; tnum.kind = Json::NUM; { char b[64]; std::snprintf(b, sizeof(b), "%.6f", (double)tot); tnum.s = b; }
    arr.a.push_back(jobj({{"op", jstr(kind)}, {"kernel_ms", num}, {"op_ms", tnum}, {"launches", jnum(n)}, {"desc", jstr(kv.first)}}));
  }
  const std::string s = arr.dump();
  if (s.size() + 1 > cap) return GPUQ_ERR_CAPACITY;
  std::memcpy(json_out, s.c_str(), s.size() + 1);
  return GPUQ_OK;
}

int gpuq_plan_metrics(gpuq_plan* p, char* buf, size_t cap) {
  if (!p || !buf) return GPUQ_ERR_INVALID;
  std::vector<PNode*> nodes; collect(p->root.get(), nodes);
  std::string s = "[";
  for (size_t i = 0; i < nodes.size(); ++i)
  {
    s += std::string(i ? "," : "") + "{\"node\":\"" + nodes[i]->kind + "\",\"output_rows\":" + std::to_string(nodes[i]->m.output_rows) + ",\"elapsed_compute\":" + std::to_string(nodes[i]->m.elapsed_ns);
    if (auto* w = dynamic_cast<ShuffleWriterExec*>(nodes[i]))
    {
      s += ",\"write_time\":" + std::to_string(w->write_ns) + ",\"repart_time\":" + std::to_string(w->repart_ns) + ",\"input_rows\":" + std::to_string(w->input_rows) + ",\"partitions\":[";
      for (size_t k = 0; k < w->stage_partitions.size(); ++k) s += (k ? "," : "") + std::to_string(w->stage_partitions[k]);
      s += "]";
    }
    s += "}";
  }
  s += "]";
  if (s.size() + 1 > cap) return GPUQ_ERR_CAPACITY;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return GPUQ_OK;
}

}  // extern "C"
