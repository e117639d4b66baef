#include "expr_compile.h"
#include "devbuf.h"
#include <algorithm>
#include <cctype>
#include <cmath>
#include <functional>
#include <set>

namespace gpuq {

// ------------------------------------------------------------------ small helpers
i128 pow10_i128(int k) { i128 r = 1; for (int i = 0; i < k; ++i) r *= 10; return r; }
int bits_for_precision(int p) { int b = (int)std::ceil(p * 3.3219280948873626) + 1; return b > 127 ? 127 : b; }
static int bits_of_value(i128 v) {
  u128 m = v < 0 ? (u128)(-(v + 1)) + 1 : (u128)v;
  int b = 0; while (m) { ++b; m >>= 1; }
  return b + 1 > 127 ? 127 : b + 1;
}
i128 parse_i128(const std::string& s) {
  size_t i = 0; bool neg = false;
  while (i < s.size() && s[i] == ' ') ++i;
  if (i < s.size() && (s[i] == '-' || s[i] == '+')) { neg = s[i] == '-'; ++i; }
  i128 v = 0; bool any = false;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') throw std::runtime_error("bad integer literal '" + s + "'");
    v = v * 10 + (s[i] - '0'); any = true;
  }
  if (!any) throw std::runtime_error("bad integer literal '" + s + "'");
  return neg ? -v : v;
}
std::string i128_to_string(i128 v) {
  if (v == 0) return "0";
  bool neg = v < 0; u128 m = neg ? (u128)(-(v + 1)) + 1 : (u128)v;
  std::string r; while (m) { r += (char)('0' + (int)(m % 10)); m /= 10; }
  if (neg) r += '-';
  std::reverse(r.begin(), r.end()); return r;
}
bool pack_str15(const std::string& s, u64& lo, u64& hi) {
  if (s.size() > 15) return false;
  hi = 0; lo = 0;
  for (size_t k = 0; k < s.size(); ++k) {
    u64 b = (unsigned char)s[k];
    if (k < 8) hi |= b << (56 - 8 * k); else lo |= b << (56 - 8 * (k - 8));
  }
  lo |= (u64)s.size();
  return true;
}

std::string DType::to_string() const {
  switch (id) {
    case T_NULL: return "Null"; case T_BOOL: return "Boolean"; case T_INT32: return "Int32"; case T_INT64: return "Int64";
    case T_DATE32: return "Date32"; case T_FLOAT64: return "Float64"; case T_UTF8: return "Utf8";
    case T_UINT32: return "UInt32"; case T_UINT64: return "UInt64";
    case T_INT8: return "Int8"; case T_INT16: return "Int16"; case T_UINT8: return "UInt8"; case T_UINT16: return "UInt16";
    case T_FLOAT32: return "Float32"; case T_DATE64: return "Date64";
    case T_TIMESTAMP: { static const char* u[4] = {"Second", "Millisecond", "Microsecond", "Nanosecond"}; return std::string("Timestamp(") + u[p & 3] + ")"; }
    case T_DECIMAL128: return "Decimal128(" + std::to_string(p) + "," + std::to_string(s) + ")";
  }
  return "?";
}
DType dtype_from_json(const Json& j) {
  DType t;
  if (j.is_str()) {
    const std::string& s = j.s;
    if (s == "Int32") t.id = T_INT32; else if (s == "Int64") t.id = T_INT64; else if (s == "Date32") t.id = T_DATE32;
    else if (s == "Float64") t.id = T_FLOAT64; else if (s == "Utf8") t.id = T_UTF8; else if (s == "Boolean") t.id = T_BOOL;
    else if (s == "UInt32") t.id = T_UINT32; else if (s == "UInt64") t.id = T_UINT64; else if (s == "Null") t.id = T_NULL;
    else if (s == "Int8") t.id = T_INT8; else if (s == "Int16") t.id = T_INT16; else if (s == "UInt8") t.id = T_UINT8; else if (s == "UInt16") t.id = T_UINT16;
    else if (s == "Float32") t.id = T_FLOAT32; else if (s == "Date64") t.id = T_DATE64;
    else if (s == "LargeUtf8") t.id = T_UTF8;      // 64-bit offsets exist only at the boundary (gpuq_table_import_arrow / gpuq_ingest_push narrow them); the schema layer keeps the name
    else throw std::runtime_error("unsupported type '" + s + "'");
    return t;
  }
  if (j.is_obj() && j.has("Decimal128")) {
    const Json& a = j.at("Decimal128");
    t.id = T_DECIMAL128; t.p = (int)a.a.at(0).i64(); t.s = (int)a.a.at(1).i64();
    if (t.p < 1 || t.p > 38 || t.s < 0 || t.s > t.p) throw std::runtime_error("bad Decimal128 precision/scale");
    return t;
  }
  if (j.is_obj() && j.has("Timestamp")) {
    // {"Timestamp": [unit, tz]} (datafusion.proto Timestamp{time_unit, timezone}); the time zone does not change what is stored or how
    // values compare: it stays with the schema layer
    const Json& a = j.at("Timestamp");
    std::string u = a.is_arr() ? a.a.at(0).str() : a.str();
    t.id = T_TIMESTAMP;
    if (u == "Second" || u == "s") t.p = 0; else if (u == "Millisecond" || u == "ms") t.p = 1; else if (u == "Microsecond" || u == "us") t.p = 2;
    else if (u == "Nanosecond" || u == "ns") t.p = 3; else throw std::runtime_error("bad Timestamp unit '" + u + "'");
    return t;
  }
  if (j.is_obj() && j.has("Dictionary")) {
    // {"Dictionary": [key type, value type]}: dictionary-encoded columns are decoded where they enter (gpuq_table_import_arrow); inside, the
    // column has its value type
    return dtype_from_json(j.at("Dictionary").a.at(1));
  }
  throw std::runtime_error("unsupported type descriptor " + j.dump());
}
// Arrow C data interface format strings (the host boundary: gpuq_import_arrow / gpuq_export_arrow / gpuq_ingest_*).  large = "U"
// (LargeUtf8): 64-bit offsets on the host side, narrowed while staging.
DType dtype_from_arrow_format(const char* f, bool* large) {
  DType t; const std::string s = f ? f : "";
  if (large) *large = false;
  if (s == "i") t.id = T_INT32; else if (s == "l") t.id = T_INT64; else if (s == "tdD") t.id = T_DATE32; else if (s == "g") t.id = T_FLOAT64;
  else if (s == "u") t.id = T_UTF8; else if (s == "b") t.id = T_BOOL; else if (s == "I") t.id = T_UINT32; else if (s == "L") t.id = T_UINT64;
  else if (s == "c") t.id = T_INT8; else if (s == "C") t.id = T_UINT8; else if (s == "s") t.id = T_INT16; else if (s == "S") t.id = T_UINT16;
  else if (s == "f") t.id = T_FLOAT32; else if (s == "tdm") t.id = T_DATE64;
  else if (s == "U") { t.id = T_UTF8; if (large) *large = true; else throw Unsupported("LargeUtf8 where 32-bit offsets are required"); }
  else if (s.size() >= 4 && s.compare(0, 2, "ts") == 0 && s[3] == ':') {
    t.id = T_TIMESTAMP;
    switch (s[2]) { case 's': t.p = 0; break; case 'm': t.p = 1; break; case 'u': t.p = 2; break; case 'n': t.p = 3; break; default: throw Unsupported("Arrow format '" + s + "'"); }
  }
  else if (s.rfind("d:", 0) == 0) {
    int p = 0, sc = 0, bits = 128;
    if (std::sscanf(s.c_str(), "d:%d,%d,%d", &p, &sc, &bits) < 2 || bits != 128) throw Unsupported("decimal format '" + s + "'");
    t.id = T_DECIMAL128; t.p = p; t.s = sc;
  } else throw Unsupported("Arrow format '" + s + "' is not supported on device (supported: b c C s S i I l L f g tdD tdm ts{s,m,u,n}: d:p,s u U)");
  return t;
}
std::string arrow_format_of(const DType& t) {
  switch (t.id) {
    case T_INT32: return "i"; case T_INT64: return "l"; case T_DATE32: return "tdD"; case T_FLOAT64: return "g"; case T_UTF8: return "u";
    case T_BOOL: return "b"; case T_UINT32: return "I"; case T_UINT64: return "L";
    case T_INT8: return "c"; case T_UINT8: return "C"; case T_INT16: return "s"; case T_UINT16: return "S"; case T_FLOAT32: return "f"; case T_DATE64: return "tdm";
    case T_TIMESTAMP: return std::string("ts") + "smun"[t.p & 3] + ":";
    case T_DECIMAL128: return "d:" + std::to_string(t.p) + "," + std::to_string(t.s);
  }
  throw Unsupported("type has no Arrow format");
}
int col_class_for(const DType& t) {
  switch (t.id) {
    case T_INT32: case T_DATE32: return CC_I32;
    case T_UINT32: return CC_U32;
    case T_INT64: case T_UINT64: case T_FLOAT64: case T_TIMESTAMP: case T_DATE64: return CC_I64;
    case T_INT8: return CC_I8; case T_INT16: return CC_I16; case T_UINT8: return CC_U8; case T_UINT16: return CC_U16; case T_FLOAT32: return CC_F32;
    case T_DECIMAL128: return CC_I128;
    case T_UTF8: return CC_STR;
    case T_BOOL: return CC_BIT;
    default: throw std::runtime_error("type " + t.to_string() + " has no device column class");
  }
}
int type_width(const DType& t) {
  switch (col_class_for(t)) { case CC_I32: case CC_U32: case CC_F32: return 4; case CC_I64: return 8; case CC_I128: case CC_STR: return 16; case CC_I8: case CC_U8: return 1; case CC_I16: case CC_U16: return 2; default: return 0; }
}
int Schema::index_of(const std::string& name) const {
  for (size_t i = 0; i < fields.size(); ++i) if (fields[i].name == name) return (int)i;
  return -1;
}
Schema schema_from_json(const Json& j) {
  Schema s;
  const Json& fs = j.is_obj() ? j.at("fields") : j;
  for (const Json& f : fs.a) {
    Field fd; fd.name = f.at("name").str(); fd.type = dtype_from_json(f.at("type"));
    fd.nullable = f.get_bool("nullable", true); fd.side = (int)f.get_i64("side", 0); fd.raw128 = (int)f.get_i64("raw128", 0); fd.dense = (int)f.get_i64("dense", 0);
    s.fields.push_back(fd);
  }
  return s;
}

static int type_bits(const DType& t) {
  switch (t.id) {
    case T_BOOL: return 2; case T_INT32: case T_DATE32: return 32; case T_UINT32: return 33; case T_INT64: case T_TIMESTAMP: case T_DATE64: return 64;
    case T_INT8: return 8; case T_UINT8: return 9; case T_INT16: return 16; case T_UINT16: return 17;
    case T_UINT64: return 65; case T_DECIMAL128: return bits_for_precision(t.p); default: return 127;
  }
}

// ------------------------------------------------------------------ AST
NodeP ExprCompiler::intern(NodeP n) {
  auto it = interned_.find(n->key);
  if (it != interned_.end()) return it->second;
  interned_[n->key] = n; return n;
}
NodeP ExprCompiler::column(int fi) {
  if (fi < 0 || fi >= (int)schema_.fields.size()) throw std::runtime_error("column index " + std::to_string(fi) + " out of range");
  auto n = std::make_shared<Node>();
  n->kind = Node::COL; n->col = fi; n->type = schema_.fields[fi].type; n->nullable = schema_.fields[fi].nullable || (schema_.fields[fi].side > 0 && !schema_.fields[fi].dense);
  n->bits = type_bits(n->type); n->key = "c" + std::to_string(fi);
  return intern(n);
}
NodeP ExprCompiler::lit_int(DType t, i128 v) {
  auto n = std::make_shared<Node>();
  n->kind = Node::LIT; n->type = t; n->nullable = false; n->lit_lo = (u64)v; n->lit_hi = (u64)((u128)v >> 64);
  n->bits = bits_of_value(v); n->key = "l:" + t.to_string() + ":" + i128_to_string(v);
  return intern(n);
}
NodeP ExprCompiler::lit_f64(double v) {
  auto n = std::make_shared<Node>();
  n->kind = Node::LIT; n->type.id = T_FLOAT64; n->nullable = false;
  u64 b; std::memcpy(&b, &v, 8); n->lit_lo = b; n->key = "lf:" + std::to_string(b);
  return intern(n);
}
NodeP ExprCompiler::lit_f32(float v) {      // registers hold a Float32 as the double of the same value
  auto n = std::make_shared<Node>();
  n->kind = Node::LIT; n->type.id = T_FLOAT32; n->nullable = false;
  const double d = (double)v; u64 b; std::memcpy(&b, &d, 8); n->lit_lo = b; n->key = "lf32:" + std::to_string(b);
  return intern(n);
}
NodeP ExprCompiler::lit_null(DType t) {
  auto n = std::make_shared<Node>();
  n->kind = Node::LIT; n->type = t; n->nullable = true; n->lit_null = true; n->bits = 1; n->key = "ln:" + t.to_string();
  return intern(n);
}
NodeP ExprCompiler::lit_str(const std::string& s) {
  auto n = std::make_shared<Node>();
  n->kind = Node::LIT; n->type.id = T_UTF8; n->nullable = false;
  if (!pack_str15(s, n->lit_lo, n->lit_hi)) {
    // beyond what a register holds: the node stands for the literal in COMPARISONS only (binary() compares with its 15-byte prefix, which
    // is exact for every value a register can hold; longer values raise FLAG_STR_TRUNC when they are loaded and the executor then lowers the
    // comparison to gpuq_utf8_compare).  Anything else that consumes it fails when the program is generated.
    pack_str15(s.substr(0, 15), n->lit_lo, n->lit_hi); n->lit_long = true; n->key = "lsL:" + s;
    return intern(n);
  }
  n->key = "ls:" + s;
  return intern(n);
}
NodeP ExprCompiler::raw(int op, DType t, bool nullable, int bits, std::vector<NodeP> ch, uint32_t imm) {
  auto n = std::make_shared<Node>();
  n->kind = Node::OPN; n->op = op; n->type = t; n->nullable = nullable; n->bits = bits > 127 ? 127 : bits; n->ch = std::move(ch); n->imm = imm;
  std::string k = "o" + std::to_string(op) + ":" + t.to_string() + ":" + std::to_string(imm) + "(";
  for (auto& c : n->ch) k += c->key + ",";
  n->key = k + ")";
  return intern(n);
}

static DType dec_type(int p, int s) { DType t; t.id = T_DECIMAL128; t.p = p > 38 ? 38 : p; t.s = s > 38 ? 38 : s; return t; }
static DType as_decimal(const DType& t) {
  if (t.id == T_DECIMAL128) return t;
  if (t.id == T_INT8 || t.id == T_UINT8) return dec_type(3, 0);      // datafusion's coercion of integers to decimals [UPSTREAM-KNOWLEDGE]
  if (t.id == T_INT16 || t.id == T_UINT16) return dec_type(5, 0);
  if (t.id == T_INT32 || t.id == T_UINT32) return dec_type(10, 0);
  if (t.id == T_INT64 || t.id == T_UINT64) return dec_type(20, 0);
  throw std::runtime_error("cannot treat " + t.to_string() + " as decimal");
}
static DType mk(int id) { DType t; t.id = id; return t; }

NodeP ExprCompiler::rescale(NodeP e, int new_scale) {
  DType d = as_decimal(e->type);
  if (d.s == new_scale) {
    if (e->type.id == T_DECIMAL128) return e;
    return raw(OP_MOV, d, e->nullable, e->bits, {e});
  }
  if (new_scale > d.s) {
    const int k = new_scale - d.s;
    if (e->kind == Node::LIT && !e->lit_null && e->bits + bits_for_precision(k + 1) < 127) {
      // constant-fold the rescale of a literal
      const i128 v = (i128)(((u128)e->lit_hi << 64) | e->lit_lo);
      return lit_int(dec_type(d.p + k, new_scale), v * pow10_i128(k));
    }
    NodeP f = lit_int(dec_type(k + 1, 0), pow10_i128(k));
    const int op = (e->bits <= 63 && f->bits <= 63) ? OP_MULW : OP_MUL;
    return raw(op, dec_type(d.p + k, new_scale), e->nullable, e->bits + f->bits, {e, f});
  }
  // reducing scale: divide, round half away from zero (arrow cast_decimal_to_decimal) [UPSTREAM-KNOWLEDGE]
  const int k = d.s - new_scale;
  DType rt = dec_type(std::max(1, d.p - k), new_scale);
  NodeP div = lit_int(dec_type(38, 0), pow10_i128(k));
  NodeP half = lit_int(dec_type(38, 0), pow10_i128(k) / 2);
  NodeP nhalf = lit_int(dec_type(38, 0), -(pow10_i128(k) / 2));
  NodeP zero = lit_int(dec_type(38, 0), 0), one = lit_int(dec_type(38, 0), 1), mone = lit_int(dec_type(38, 0), -1);
  NodeP q = raw(OP_DIV, rt, e->nullable, e->bits, {e, div});
  NodeP r = raw(OP_MOD, rt, e->nullable, div->bits, {e, div});
  NodeP nonneg = raw(OP_GE, mk(T_BOOL), e->nullable, 2, {e, zero});
  NodeP up = raw(OP_SELECT, rt, e->nullable, 2, {raw(OP_GE, mk(T_BOOL), e->nullable, 2, {r, half}), one, zero});
  NodeP down = raw(OP_SELECT, rt, e->nullable, 2, {raw(OP_LE, mk(T_BOOL), e->nullable, 2, {r, nhalf}), mone, zero});
  NodeP adj = raw(OP_SELECT, rt, e->nullable, 2, {nonneg, up, down});
  return raw(OP_ADD, rt, e->nullable, e->bits, {q, adj});
}

static i64 units_per_second(int unit) { static const i64 k[4] = {1, 1000, 1000000, 1000000000}; return k[unit & 3]; }
// floor(e / d) for a positive constant d (days of a timestamp before the epoch round down, as chrono's date of a datetime does)
NodeP ExprCompiler::floor_div(NodeP e, i64 d, DType rt) {
  if (d == 1) return raw(OP_MOV, rt, e->nullable, e->bits, {e});
  const DType i64t = mk(T_INT64);
  NodeP dv = lit_int(i64t, d), zero = lit_int(i64t, 0), mone = lit_int(i64t, -1);
  NodeP q = raw(OP_DIV, i64t, e->nullable, e->bits, {e, dv});
  NodeP r = raw(OP_MOD, i64t, e->nullable, dv->bits, {e, dv});
  NodeP adj = raw(OP_SELECT, i64t, e->nullable, 2, {raw(OP_LT, mk(T_BOOL), e->nullable, 2, {r, zero}), mone, zero});
  return raw(OP_ADD, rt, e->nullable, e->bits, {q, adj});
}
// a count of `from_per_s` units per second as a count of `to_per_s` units (arrow-cast: multiply, or divide truncating toward zero)
NodeP ExprCompiler::rescale_time(NodeP e, i64 from_per_s, i64 to_per_s, DType rt) {
  if (from_per_s == to_per_s) return raw(OP_MOV, rt, e->nullable, e->bits, {e});
  if (to_per_s > from_per_s) { NodeP f = lit_int(mk(T_INT64), to_per_s / from_per_s); return raw(OP_MULW, rt, e->nullable, std::min(127, e->bits + f->bits), {e, f}); }
  return raw(OP_DIV, rt, e->nullable, e->bits, {e, lit_int(mk(T_INT64), from_per_s / to_per_s)});
}

NodeP ExprCompiler::cast(NodeP e, DType to) {
  const DType from = e->type;
  if (from == to) return e;
  if (from.id == T_NULL) return lit_null(to);
  // temporal <-> temporal (arrow-cast 49 cast_with_options [UPSTREAM-KNOWLEDGE]: Date32 = days, Date64 = milliseconds, Timestamp = count of its unit)
  if (from.is_temporal() && to.is_temporal()) {
    const i64 day_ms = 86400000;
    if (from.id == T_TIMESTAMP && to.id == T_TIMESTAMP) return rescale_time(e, units_per_second(from.p), units_per_second(to.p), to);
    if (from.id == T_TIMESTAMP && to.id == T_DATE32) return floor_div(e, units_per_second(from.p) * 86400, to);
    if (from.id == T_TIMESTAMP && to.id == T_DATE64) return rescale_time(e, units_per_second(from.p), 1000, to);
    if (from.id == T_DATE32 && to.id == T_TIMESTAMP) return raw(OP_MULW, to, e->nullable, 64, {e, lit_int(mk(T_INT64), units_per_second(to.p) * 86400)});
    if (from.id == T_DATE32 && to.id == T_DATE64) return raw(OP_MULW, to, e->nullable, 64, {e, lit_int(mk(T_INT64), day_ms)});
    if (from.id == T_DATE64 && to.id == T_DATE32) return raw(OP_DIV, to, e->nullable, 32, {e, lit_int(mk(T_INT64), day_ms)});
    if (from.id == T_DATE64 && to.id == T_TIMESTAMP) return rescale_time(e, 1000, units_per_second(to.p), to);
  }
  if (to.id == T_FLOAT32) {
    // one rounding to double and one to float; the second is exact in effect for every operand a double holds exactly (all narrow integers,
    // Int32, Date32, decimals of <= 15 digits), an Int64 beyond 2^53 can round twice
    if (from.id == T_FLOAT64) return raw(OP_F32R, to, e->nullable, 127, {e});
    return raw(OP_F32R, to, e->nullable, 127, {cast(e, mk(T_FLOAT64))});
  }
  if (to.id == T_FLOAT64) {
    if (from.id == T_FLOAT32) return raw(OP_MOV, to, e->nullable, 127, {e});      // the register already holds the double of that value
    if (from.is_int() || from.id == T_DATE32 || from.id == T_BOOL) return raw(OP_I2F, to, e->nullable, 127, {e});
    if (from.is_decimal()) {
      NodeP f = raw(OP_I2F, to, e->nullable, 127, {e});
      if (from.s == 0) return f;
      return raw(OP_FDIV, to, e->nullable, 127, {f, lit_f64(std::pow(10.0, from.s))});
    }
  }
  if (to.is_decimal()) {
    if (from.is_int() || from.is_decimal()) {
      NodeP r = rescale(e, to.s);
      if (r->type == to) return r;
      return raw(OP_MOV, to, e->nullable, std::min(r->bits, bits_for_precision(to.p)), {r});
    }
  }
  if (to.is_int() || to.is_temporal()) {
    // the storage integer of a temporal type casts to and from plain integers unchanged
    if ((from.is_int() || from.id == T_BOOL || from.is_temporal()) && (to.is_int() || from.is_int())) return raw(OP_MOV, to, e->nullable, std::min(e->bits, type_bits(to)), {e});
    if (from.is_decimal()) { NodeP r = rescale(e, 0); return raw(OP_MOV, to, e->nullable, std::min(r->bits, type_bits(to)), {r}); }
    if (from.is_float()) return raw(OP_F2I, to, e->nullable, type_bits(to), {e});
  }
  if (to.id == T_BOOL && from.is_int()) return raw(OP_NE, to, e->nullable, 2, {e, lit_int(from, 0)});
  if (to.id == T_BOOL && from.is_float()) return raw(OP_FNE, to, e->nullable, 2, {cast(e, mk(T_FLOAT64)), lit_f64(0.0)});
  throw std::runtime_error("unsupported cast " + from.to_string() + " -> " + to.to_string());
}

static std::string norm_op(const std::string& op) {
  static const std::map<std::string, std::string> m = {
      {"Plus", "+"}, {"Minus", "-"}, {"Multiply", "*"}, {"Divide", "/"}, {"Modulo", "%"}, {"Eq", "="}, {"NotEq", "!="},
      {"Lt", "<"}, {"LtEq", "<="}, {"Gt", ">"}, {"GtEq", ">="}, {"And", "AND"}, {"Or", "OR"}, {"and", "AND"}, {"or", "OR"}, {"<>", "!="}, {"==", "="}};
  auto it = m.find(op); return it == m.end() ? op : it->second;
}

NodeP ExprCompiler::binary(const std::string& op_in, NodeP l, NodeP r) {
  const std::string op = norm_op(op_in);
  const bool nullable = l->nullable || r->nullable;
  if (op == "AND" || op == "OR") {
    if (l->type.id != T_BOOL || r->type.id != T_BOOL) throw std::runtime_error(op + " needs boolean operands");
    return raw(op == "AND" ? OP_AND : OP_OR, mk(T_BOOL), nullable, 2, {l, r});
  }
  const bool is_cmp = (op == "=" || op == "!=" || op == "<" || op == "<=" || op == ">" || op == ">=");
  const bool is_arith = (op == "+" || op == "-" || op == "*" || op == "/" || op == "%");
  if (!is_cmp && !is_arith) throw std::runtime_error("unsupported binary operator '" + op_in + "'");
  // NULL literal operand adopts the other side's type
  if (l->type.id == T_NULL && r->type.id != T_NULL) l = lit_null(r->type);
  if (r->type.id == T_NULL && l->type.id != T_NULL) r = lit_null(l->type);
  if (is_cmp) {
    static const std::map<std::string, std::pair<int, int>> ops = {
        {"=", {OP_EQ, OP_FEQ}}, {"!=", {OP_NE, OP_FNE}}, {"<", {OP_LT, OP_FLT}}, {"<=", {OP_LE, OP_FLE}}, {">", {OP_GT, OP_FGT}}, {">=", {OP_GE, OP_FGE}}};
    const auto pr = ops.at(op);
    if (l->type.is_float() || r->type.is_float()) {
      return raw(pr.second, mk(T_BOOL), nullable, 2, {cast(l, mk(T_FLOAT64)), cast(r, mk(T_FLOAT64))});
    }
    if (l->type.is_decimal() || r->type.is_decimal()) {
      const int s = std::max(as_decimal(l->type).s, as_decimal(r->type).s);
      return raw(pr.first, mk(T_BOOL), nullable, 2, {rescale(l, s), rescale(r, s)});
    }
    const bool lu = l->type.id == T_UTF8, ru = r->type.id == T_UTF8;
    if (lu != ru) throw std::runtime_error("cannot compare " + l->type.to_string() + " with " + r->type.to_string());
    if (lu && (l->lit_long || r->lit_long) && !(l->lit_long && r->lit_long)) {
      // c OP L with L beyond 15 bytes, c a value of at most 15 (longer ones were flagged at their load): with P = L's first 15 bytes,
      // c == P means c is a proper prefix of L, hence c < L  <=>  c <= P,  c <= L  <=>  c <= P,  c > L  <=>  c > P,  c >= L  <=>  c > P,
      // c = L never, c != L always (NULL in -> NULL out in every case)
      const bool lit_left = l->lit_long;
      NodeP c = lit_left ? r : l;
      NodeP P = raw(OP_MOV, mk(T_UTF8), false, 127, {lit_left ? l : r}, 0xC0DEu);      // the packed prefix as an ordinary value (the marker: finish() lets only this use through)
      std::string o = op;
      if (lit_left) o = op == "<" ? ">" : op == "<=" ? ">=" : op == ">" ? "<" : op == ">=" ? "<=" : op;      // L OP c  ==  c mirror(OP) L
      if (o == "=") return raw(OP_LT, mk(T_BOOL), nullable, 2, {c, c});
      if (o == "!=") return raw(OP_EQ, mk(T_BOOL), nullable, 2, {c, c});
      return raw((o == "<" || o == "<=") ? OP_LE : OP_GT, mk(T_BOOL), nullable, 2, {c, P});
    }
    const bool ld = l->type.is_temporal(), rd = r->type.is_temporal();
    if ((ld && !(r->type == l->type || r->type.is_int())) || (rd && !(l->type == r->type || l->type.is_int())))
      throw std::runtime_error("cannot compare " + l->type.to_string() + " with " + r->type.to_string());
    return raw(pr.first, mk(T_BOOL), nullable, 2, {l, r});
  }
  // arithmetic
  if (l->type.is_float() || r->type.is_float()) {
    static const std::map<std::string, int> fo = {{"+", OP_FADD}, {"-", OP_FSUB}, {"*", OP_FMUL}, {"/", OP_FDIV}};
    auto it = fo.find(op); if (it == fo.end()) throw std::runtime_error("unsupported float operator " + op);
    NodeP z = raw(it->second, mk(T_FLOAT64), nullable, 127, {cast(l, mk(T_FLOAT64)), cast(r, mk(T_FLOAT64))});
    // Float32 op Float32: the double result rounded to float IS the correctly rounded float result for + - * / (53 >= 2 * 24 + 2 bits)
    if (l->type.id == T_FLOAT32 && r->type.id == T_FLOAT32) return raw(OP_F32R, mk(T_FLOAT32), nullable, 127, {z});
    return z;
  }
  if (l->type.is_decimal() || r->type.is_decimal()) {
    if (!(l->type.is_decimal() || l->type.is_int()) || !(r->type.is_decimal() || r->type.is_int()))
      throw std::runtime_error("unsupported decimal arithmetic operands " + l->type.to_string() + ", " + r->type.to_string());
    const DType dl = as_decimal(l->type), dr = as_decimal(r->type);
    if (op == "+" || op == "-") {
      const int s = std::max(dl.s, dr.s);
      const int p = std::min(38, std::max(dl.p - dl.s, dr.p - dr.s) + s + 1);
      NodeP a = rescale(l, s), b = rescale(r, s);
      return raw(op == "+" ? OP_ADD : OP_SUB, dec_type(p, s), nullable, std::max(a->bits, b->bits) + 1, {a, b});
    }
    if (op == "*") {
      const DType rt = dec_type(std::min(38, dl.p + dl.s * 0 + dr.p + 1), std::min(38, dl.s + dr.s));
      NodeP a = rescale(l, dl.s), b = rescale(r, dr.s);
      const int o = (a->bits <= 63 && b->bits <= 63) ? OP_MULW : OP_MUL;
      return raw(o, rt, nullable, a->bits + b->bits, {a, b});
    }
    // arrow-arith 49 numeric.rs decimal_op [UPSTREAM-KNOWLEDGE]: Div -> scale s1+4, precision p1+(4+s2), l*10^(4+s2) / r
    // truncated toward zero; Rem -> scale max(s1,s2), precision min(p1-s1,p2-s2)+scale.  x/0 -> NULL (see the integer case).
    if (op == "/") {
      const int rs = std::min(38, dl.s + 4), k = rs - dl.s + dr.s;
      NodeP a = rescale(l, dl.s + k), b = rescale(r, dr.s);
      if (a->bits > 127) throw std::runtime_error("decimal division " + l->type.to_string() + " / " + r->type.to_string() + " can overflow 128 bits on device");
      return raw(OP_DIV, dec_type(std::min(38, dl.p + k), rs), true, a->bits, {a, b});
    }
    if (op == "%") {
      const int s = std::max(dl.s, dr.s);
      NodeP a = rescale(l, s), b = rescale(r, s);
      if (a->bits > 127 || b->bits > 127) throw std::runtime_error("decimal modulo operands can overflow 128 bits on device");
      return raw(OP_MOD, dec_type(std::min(38, std::min(dl.p - dl.s, dr.p - dr.s) + s), s), true, b->bits, {a, b});
    }
    throw std::runtime_error("decimal operator '" + op + "' is not supported on device yet");
  }
  if ((l->type.is_int() || l->type.id == T_DATE32) && (r->type.is_int() || r->type.id == T_DATE32)) {
    DType rt = mk((l->type.id == T_INT64 || r->type.id == T_INT64 || l->type.id == T_UINT64 || r->type.id == T_UINT64) ? T_INT64 : T_INT32);
    if (l->type == r->type && l->type.is_int()) rt = l->type;      // the planner has coerced both sides: the result wraps to that width where it is stored
    if (l->type.id == T_DATE32 && r->type.id == T_DATE32 && op == "-") rt = mk(T_INT32);
    else if (l->type.id == T_DATE32 || r->type.id == T_DATE32) rt = mk(T_DATE32);
    if (op == "+") return raw(OP_ADD, rt, nullable, std::max(l->bits, r->bits) + 1, {l, r});
    if (op == "-") return raw(OP_SUB, rt, nullable, std::max(l->bits, r->bits) + 1, {l, r});
    if (op == "*") return raw((l->bits <= 63 && r->bits <= 63) ? OP_MULW : OP_MUL, rt, nullable, l->bits + r->bits, {l, r});
    if (op == "/") return raw(OP_DIV, rt, true, l->bits, {l, r});   // x/0 -> NULL (arrow raises DivideByZero)
    if (op == "%") return raw(OP_MOD, rt, true, r->bits, {l, r});
  }
  throw std::runtime_error("unsupported operands for '" + op + "': " + l->type.to_string() + ", " + r->type.to_string());
}

NodeP ExprCompiler::not_(NodeP e) {
  if (e->type.id != T_BOOL) throw std::runtime_error("NOT needs a boolean operand");
  return raw(OP_NOT, mk(T_BOOL), e->nullable, 2, {e});
}
NodeP ExprCompiler::is_null(NodeP e, bool negate) { return raw(negate ? OP_ISNOTNULL : OP_ISNULL, mk(T_BOOL), false, 2, {e}); }
NodeP ExprCompiler::negative(NodeP e) {
  if (e->type.is_float()) return raw(OP_FNEG, e->type, e->nullable, 127, {e});
  return raw(OP_NEG, e->type, e->nullable, e->bits + 1, {e});
}
NodeP ExprCompiler::select(NodeP c, NodeP t, NodeP f) {
  if (t->type.id == T_NULL) t = lit_null(f->type);
  if (f->type.id == T_NULL) f = lit_null(t->type);
  if (t->type != f->type) {
    if (t->type.is_float() || f->type.is_float()) { t = cast(t, mk(T_FLOAT64)); f = cast(f, mk(T_FLOAT64)); }
    else if (t->type.is_decimal() || f->type.is_decimal()) {
      const DType a = as_decimal(t->type), b = as_decimal(f->type);
      const int s = std::max(a.s, b.s); const DType rt = dec_type(std::min(38, std::max(a.p - a.s, b.p - b.s) + s), s);
      t = cast(t, rt); f = cast(f, rt);
    } else if (t->type.is_int() && f->type.is_int()) { t = cast(t, mk(T_INT64)); f = cast(f, mk(T_INT64)); }
    else throw std::runtime_error("CASE branches have incompatible types " + t->type.to_string() + ", " + f->type.to_string());
  }
  return raw(OP_SELECT, t->type, t->nullable || f->nullable, std::max(t->bits, f->bits), {c, t, f});
}
NodeP ExprCompiler::coalesce0(NodeP e) { return raw(OP_COALESCE0, e->type, false, e->bits, {e}); }
NodeP ExprCompiler::nullif0(NodeP e, NodeP guard) { return raw(OP_NULLIF0, e->type, true, e->bits, {e, guard}); }

NodeP ExprCompiler::from_json(const Json& e) {
  if (!e.is_obj() || e.o.size() != 1) throw std::runtime_error("expression must be an object with one key: " + e.dump());
  const std::string& kind = e.o[0].first;
  const Json& v = e.o[0].second;
  if (kind == "column") {
    int idx = (int)v.get_i64("index", -1);
    if (v.has("name")) {
      const int byname = schema_.index_of(v.at("name").str());
      if (idx < 0) idx = byname;
      else if (byname >= 0 && idx < (int)schema_.fields.size() && schema_.fields[idx].name != v.at("name").str()) idx = byname;
    }
    if (idx < 0) throw std::runtime_error("unknown column " + v.dump());
    return column(idx);
  }
  if (kind == "literal") {
    const DType t = dtype_from_json(v.at("type"));
    const Json* val = v.find("value");
    if (!val || val->is_null()) return lit_null(t);
    switch (t.id) {
      case T_UTF8: return lit_str(val->str());
      case T_FLOAT64: return lit_f64(val->f64());
      case T_FLOAT32: return lit_f32((float)val->f64());
      case T_BOOL: return lit_int(t, val->boolean() ? 1 : 0);
      default: return lit_int(t, parse_i128(val->is_str() ? val->s : val->s));
    }
  }
  if (kind == "binary_expr") return binary(v.at("op").str(), from_json(v.at("l")), from_json(v.at("r")));
  if (kind == "cast" || kind == "try_cast") return cast(from_json(v.at("expr")), dtype_from_json(v.at("arrow_type")));
  if (kind == "not_expr") return not_(from_json(v.at("expr")));
  if (kind == "is_null_expr") return is_null(from_json(v.at("expr")), false);
  if (kind == "is_not_null_expr") return is_null(from_json(v.at("expr")), true);
  if (kind == "negative") return negative(from_json(v.at("expr")));
  if (kind == "in_list") {
    NodeP x = from_json(v.at("expr"));
    NodeP acc;
    for (const Json& it : v.at("list").a) {
      NodeP eq = binary("=", x, from_json(it));
      acc = acc ? binary("OR", acc, eq) : eq;
    }
    if (!acc) acc = lit_int(mk(T_BOOL), 0);
    return v.get_bool("negated", false) ? not_(acc) : acc;
  }
  if (kind == "scalar_function") {
    // PhysicalScalarFunctionNode (datafusion.proto: name, fun, args, return_type).  Two of the harness's functions are built:
    //   date_part('YEAR' | 'MONTH' | 'DAY', Date32) -> Float64 (extract(year from l_shipdate): q7, q8, q9)
    //   substr(Utf8, start [, len]) with literal start >= 1 / len >= 0, ASCII (substring(c_phone from 1 for 2): q22)
    std::string nm = v.get_str("name", v.get_str("fun", ""));
    for (auto& ch : nm) ch = (char)std::tolower((unsigned char)ch);
    const auto& args = v.at("args").a;
    auto lit_arg = [&](const Json& a, const char* what) -> const Json& {
      if (!(a.is_obj() && a.o.size() == 1 && a.o[0].first == "literal" && a.o[0].second.find("value") && !a.o[0].second.at("value").is_null()))
        throw Unsupported(std::string("scalar function ") + nm + ": " + what + " must be a non-NULL literal");
      return a.o[0].second.at("value");
    };
    if (nm == "date_part" || nm == "datepart") {
      if (args.size() != 2) throw std::runtime_error("date_part takes (part, date)");
      std::string part = lit_arg(args[0], "the part").str();
      for (auto& ch : part) ch = (char)std::toupper((unsigned char)ch);
      const int which = part == "YEAR" ? 0 : part == "MONTH" ? 1 : part == "DAY" ? 2 : -1;
      if (which < 0) throw Unsupported("date_part('" + part + "', ..): YEAR, MONTH and DAY are built");
      NodeP x = from_json(args[1]);
      if (x->type.id == T_TIMESTAMP || x->type.id == T_DATE64) x = cast(x, mk(T_DATE32));      // the date of the instant (UTC), then as for a date
      if (x->type.id != T_DATE32) throw Unsupported("date_part over " + x->type.to_string() + " (Date32, Date64 and Timestamp are built)");
      return cast(raw(OP_DATEPART, mk(T_INT64), x->nullable, 24, {x}, (uint32_t)which), mk(T_FLOAT64));      // [UPSTREAM-KNOWLEDGE] datafusion 34 date_part returns Float64
    }
    if (nm == "substr" || nm == "substring") {
      if (args.size() != 2 && args.size() != 3) throw std::runtime_error("substr takes (string, start [, length])");
      NodeP x = from_json(args[0]);
      if (x->type.id != T_UTF8) throw Unsupported("substr over " + x->type.to_string());
      const i128 start = parse_i128(lit_arg(args[1], "start").s);
      const i128 len = args.size() == 3 ? parse_i128(lit_arg(args[2], "length").s) : (i128)255;
      if (start < 1 || start > 16) throw Unsupported("substr: a start outside 1..16 (the packed form holds 15 bytes)");
      if (len < 0 || (args.size() == 3 && len > 15)) throw Unsupported("substr: a length outside 0..15");
      return raw(OP_SUBSTR, mk(T_UTF8), x->nullable, 127, {x}, (uint32_t)(start - 1) | ((uint32_t)len << 8));
    }
    throw Unsupported("scalar function '" + nm + "' is not built on the device (date_part, substr are)");
  }
  if (kind == "case_") {
    NodeP base = v.has("expr") ? from_json(v.at("expr")) : nullptr;
    NodeP acc = v.has("else_expr") ? from_json(v.at("else_expr")) : lit_null(mk(T_NULL));
    const auto& wt = v.at("when_then_expr").a;
    for (size_t i = wt.size(); i-- > 0;) {
      NodeP w = from_json(wt[i].at("when_expr"));
      if (base) w = binary("=", base, w);
      acc = select(w, from_json(wt[i].at("then_expr")), acc);
    }
    return acc;
  }
  throw std::runtime_error("unsupported expression node '" + kind + "'");
}

// ------------------------------------------------------------------ program assembly
void ExprCompiler::add_predicate(NodeP e) {
  if (e->type.id != T_BOOL) throw std::runtime_error("predicate must be boolean, got " + e->type.to_string());
  pred_ = pred_ ? binary("AND", pred_, e) : e;
}
int ExprCompiler::add_output(NodeP e) { outs_.push_back(e); return (int)outs_.size() - 1; }

CompiledProgram ExprCompiler::finish() {
  CompiledProgram C;
  // OP_MOV only re-types a value (same 128-bit pattern): it aliases its operand's register
  auto rep = [](Node* n) { while (n->kind == Node::OPN && n->op == OP_MOV) n = n->ch[0].get(); return n; };
  // use counts over the DAG of representatives
  std::map<Node*, int> uses;
  std::set<Node*> seen;
  std::function<void(Node*)> visit = [&](Node* n0) {
    Node* n = rep(n0);
    uses[n]++;
    if (seen.count(n)) return;
    seen.insert(n);
    for (auto& c : n->ch) visit(c.get());
  };
  if (pred_) { visit(pred_.get()); uses[rep(pred_.get())] += 1000000; }
  for (auto& o : outs_) { visit(o.get()); uses[rep(o.get())] += 1000000; }

  std::map<Node*, int> reg;
  std::vector<Node*> order;
  std::set<Node*> ordered;
  std::function<void(Node*)> topo = [&](Node* n0) {
    Node* n = rep(n0);
    if (ordered.count(n)) return;
    ordered.insert(n);
    for (auto& c : n->ch) topo(c.get());
    order.push_back(n);
  };
  {      // a Utf8 literal beyond 15 bytes may only feed the prefix comparison binary() builds for it
    const char* msg = "Utf8 literal longer than 15 bytes is not supported on device outside a comparison";
    std::set<Node*> walked;
    std::function<void(Node*)> chk = [&](Node* n) {
      if (!walked.insert(n).second) return;
      for (auto& c : n->ch) {
        if (c->kind == Node::LIT && c->lit_long && !(n->kind == Node::OPN && n->op == OP_MOV && n->imm == 0xC0DEu)) throw std::runtime_error(msg);
        chk(c.get());
      }
    };
    if (pred_) { if (pred_->kind == Node::LIT && pred_->lit_long) throw std::runtime_error(msg); chk(pred_.get()); }
    for (auto& o : outs_) { if (o->kind == Node::LIT && o->lit_long) throw std::runtime_error(msg); chk(o.get()); }
  }
  if (pred_) topo(pred_.get());
  for (auto& o : outs_) topo(o.get());
  for (Node* n : order) if (n->kind == Node::COL) {
    if ((int)C.col_field.size() >= MAX_COLS) throw std::runtime_error("expression references more than " + std::to_string(MAX_COLS) + " columns");
    reg[n] = (int)C.col_field.size();
    C.col_field.push_back(n->col);
  }
  // Utf8 columns whose every use is `= literal` / `!= literal` (IN lists compile to those) or IS [NOT] NULL: the packed form carries
  // the true length in its low byte, so a value beyond 15 bytes differs from every literal a register can hold and needs no refusal
  C.col_loose.assign(C.col_field.size(), false);
  for (size_t c = 0; c < C.col_field.size(); ++c) {
    const Field& f = schema_.fields[C.col_field[c]];
    if (f.type.id != T_UTF8 || f.raw128) continue;
    bool loose = true;
    auto is_this = [&](Node* m) { return m->kind == Node::COL && m->col == C.col_field[c]; };
    if (pred_ && is_this(rep(pred_.get()))) loose = false;
    for (auto& o : outs_) if (is_this(rep(o.get()))) loose = false;
    for (Node* n : order) {
      if (n->kind != Node::OPN) continue;
      for (size_t k = 0; k < n->ch.size(); ++k) {
        if (!is_this(rep(n->ch[k].get()))) continue;
        if (n->op == OP_ISNULL || n->op == OP_ISNOTNULL) continue;
        const bool eq_lit = (n->op == OP_EQ || n->op == OP_NE) && n->ch.size() == 2 && rep(n->ch[1 - k].get())->kind == Node::LIT;
        // a substring that lies inside the first 15 bytes is exact whatever the value's length (the packed form holds them all)
        const bool prefix_substr = n->op == OP_SUBSTR && ((n->imm >> 8) & 0xFFu) != 255u && (n->imm & 0xFFu) + ((n->imm >> 8) & 0xFFu) <= 15u;
        if (!eq_lit && !prefix_substr) loose = false;
      }
    }
    C.col_loose[c] = loose;
  }
  std::vector<bool> busy(NREG, false);
  for (size_t i = 0; i < C.col_field.size(); ++i) busy[i] = true;
  auto alloc = [&]() { for (int r = 0; r < NREG; ++r) if (!busy[r]) { busy[r] = true; return r; } throw std::runtime_error("expression needs more than " + std::to_string(NREG) + " live registers"); return -1; };
  std::vector<std::pair<u64, u64>> imms;
  auto imm_index = [&](u64 lo, u64 hi) {
    for (size_t i = 0; i < imms.size(); ++i) if (imms[i].first == lo && imms[i].second == hi) return (int)i;
    if ((int)imms.size() >= MAX_IMMS) throw std::runtime_error("expression needs more than " + std::to_string(MAX_IMMS) + " immediates");
    imms.push_back({lo, hi}); return (int)imms.size() - 1;
  };
  auto emit = [&](int op, int d, int a, int b, uint32_t imm) {
    if (C.n_insns >= MAX_INSNS) throw std::runtime_error("expression program longer than " + std::to_string(MAX_INSNS) + " instructions");
    DevInsn& in = C.code.insns[C.n_insns++];
    in.op = (uint8_t)op; in.dst = (uint8_t)d; in.a = (uint8_t)a; in.b = (uint8_t)b; in.imm = imm;
  };
  std::map<Node*, int> remaining = uses;
  auto release = [&](Node* c0) { Node* c = rep(c0); if (--remaining[c] == 0) busy[reg[c]] = false; };
  for (Node* n : order) {
    if (n->kind == Node::COL) continue;
    if (n->kind == Node::LIT) {
      const int d = alloc(); reg[n] = d;
      if (n->lit_null) { emit(OP_IMM, d, 0, 0, (uint32_t)imm_index(0, 0)); emit(OP_NULLIF0, d, d, d, 0); }
      else emit(OP_IMM, d, 0, 0, (uint32_t)imm_index(n->lit_lo, n->lit_hi));
      continue;
    }
    const int a = n->ch.size() > 0 ? reg.at(rep(n->ch[0].get())) : 0;
    const int b = n->ch.size() > 1 ? reg.at(rep(n->ch[1].get())) : a;
    uint32_t imm = n->imm;
    if (n->op == OP_SELECT) imm = (uint32_t)reg.at(rep(n->ch[2].get()));
    for (auto& c : n->ch) release(c.get());     // operands are read before dst is written
    const int d = alloc(); reg[n] = d;
    emit(n->op, d, a, b, imm);
  }
  for (size_t i = 0; i < imms.size(); ++i) { C.code.imm_lo[i] = imms[i].first; C.code.imm_hi[i] = imms[i].second; }
  C.pred_reg = pred_ ? reg.at(rep(pred_.get())) : -1;
  for (auto& o : outs_) {
    C.out_reg.push_back(reg.at(rep(o.get()))); C.out_type.push_back(o->type); C.out_nullable.push_back(o->nullable); C.out_key.push_back(o->key); C.out_bits.push_back((o->type.is_int() || o->type.is_decimal() || o->type.id == T_DATE32) ? o->bits : 127);
  }
  C.jit_src = jit_source(C, order, reg);
  return C;
}

// ------------------------------------------------------------------ JIT source
// Same DAG, emitted as one straight-line typed function: every first-level load is issued before any
// loaded value is touched, integers stay 64-bit wherever the range analysis allows, the predicate
// returns early, and only the registers a sink reads are written.
std::string ExprCompiler::jit_source(const CompiledProgram& C, const std::vector<Node*>& order, const std::map<Node*, int>& /*reg*/) {
  auto rep = [](Node* n) { while (n->kind == Node::OPN && n->op == OP_MOV) n = n->ch[0].get(); return n; };
  std::string S;
  auto L = [&](const std::string& l) { S += "  " + l + "\n"; };
  std::map<Node*, std::string> name;       // value expression of a node (variable name)
  std::map<Node*, std::string> nul;        // null expression ("false" or variable)
  auto is_str = [](const Node* n) { return n->type.id == T_UTF8; };
  auto is_f = [](const Node* n) { return n->type.is_float(); };
  auto is_b = [](const Node* n) { return n->type.id == T_BOOL; };
  auto wide = [&](const Node* n) { return !is_str(n) && !is_f(n) && !is_b(n) && n->bits > 64; };
  auto ctype = [&](const Node* n) { return is_f(n) ? std::string("double") : (is_b(n) ? std::string("bool") : (wide(n) ? std::string("i128") : std::string("i64"))); };
  // Three stages so that a sink can software-pipeline rows (k_agg_tiny_body does): gpuq_jit_pre issues the loads other
  // loads depend on (index vectors, Utf8 offsets of directly addressed columns), gpuq_jit_load issues every remaining
  // first-touch load into a JitRaw record without looking at any loaded value, gpuq_jit_compute is pure arithmetic
  // (plus the bytes of strings longer than one byte).  gpuq_jit_eval is the three in sequence.
  std::string SP, SLD, FP, FR;       // stage bodies / struct fields
  auto LP = [&](const std::string& l) { SP += "  " + l + "\n"; };
  auto LL = [&](const std::string& l) { SLD += "  " + l + "\n"; };
  auto fieldP = [&](const std::string& ty, const std::string& nm) { FP += "  " + ty + " " + nm + ";\n"; };
  std::string touch;                 // every loaded record field is read once, unconditionally, at the top of the compute stage
  auto fieldR = [&](const std::string& ty, const std::string& nm) {
    FR += "  " + ty + " " + nm + ";\n";
    if (ty == "ulonglong2") touch += "  asm volatile(\"\" :: \"v\"(w." + nm + ".x), \"v\"(w." + nm + ".y));\n";
    else if (ty != "bool") touch += "  asm volatile(\"\" :: \"v\"(w." + nm + "));\n";
  };
  // ---- rows
  bool via_used[MAX_VIA + 1] = {false, false, false, false};
  for (size_t c = 0; c < C.col_field.size(); ++c) via_used[schema_.fields[C.col_field[c]].side] = true;
  LP("const uint32_t row0 = (uint32_t)pos; (void)row0;");
  LL("const uint32_t row0 = (uint32_t)pos; (void)row0;");
  L("const uint32_t row0 = (uint32_t)pos; (void)row0;");
  for (int k = 1; k <= MAX_VIA; ++k) if (via_used[k]) {
    const std::string ks = std::to_string(k);
    fieldP("uint32_t", "row" + ks);
    LP("q.row" + ks + " = P.via[" + std::to_string(k - 1) + "][pos];");
    LL("const uint32_t row" + ks + " = q.row" + ks + ";");
  }
  // 16-byte columns of which only the low 8 bytes are read (Decimal128 whose precision bounds |v| < 2^63, Float64 / Bool in raw cells)
  std::vector<bool> narrow128(C.col_field.size(), false);
  for (size_t c = 0; c < C.col_field.size(); ++c) {
    const Field& f = schema_.fields[C.col_field[c]];
    const int cls = f.raw128 ? (int)CC_I128 : col_class_for(f.type);
    if (cls != CC_I128 || f.type.id == T_UTF8) continue;
    bool all_narrow = true, seen = false;
    for (Node* n : order) if (n->kind == Node::COL && n->col == C.col_field[c]) { seen = true; if (wide(n)) all_narrow = false; }
    narrow128[c] = seen && all_narrow;
  }
  // ---- loads
  for (size_t c = 0; c < C.col_field.size(); ++c) {
    const Field& f = schema_.fields[C.col_field[c]];
    const std::string cs = std::to_string(c), row = "row" + std::to_string(f.side);
    LL("const DevCol& col" + cs + " = P.cols[" + cs + "];");
    L("const DevCol& col" + cs + " = P.cols[" + cs + "]; (void)col" + cs + ";");
    if (f.side > 0) {
      LL("const bool ok" + cs + " = " + row + " != NULL_ROW;"); LL("const uint32_t r" + cs + " = ok" + cs + " ? " + row + " : 0u;");
      fieldR("bool", "ok" + cs); LL("w.ok" + cs + " = ok" + cs + ";"); L("const bool ok" + cs + " = w.ok" + cs + ";");
      fieldR("uint32_t", "r" + cs); LL("w.r" + cs + " = r" + cs + ";"); L("const uint32_t r" + cs + " = w.r" + cs + "; (void)r" + cs + ";");
    } else { LL("const uint32_t r" + cs + " = row0;"); L("const uint32_t r" + cs + " = row0; (void)r" + cs + ";"); }
    if (f.nullable) {
      LL("w.vb" + cs + " = (col" + cs + ".validity ? col" + cs + ".validity : (const uint8_t*)P.code)[col" + cs + ".validity ? (r" + cs + " >> 3) : 0u];");
      fieldR("uint32_t", "vb" + cs); L("const uint32_t vb" + cs + " = w.vb" + cs + ";");
    }
    const int cls = f.raw128 ? (int)CC_I128 : col_class_for(f.type);
    const std::string A = "a" + cs;
    switch (cls) {
      // (columns read by position stream through once: GPUQ_LD_STREAM can make those loads non-temporal, gpuq_dev.h)
      case CC_I32: case CC_U32: fieldR("uint32_t", A); LL("w." + A + " = " + std::string(f.side == 0 ? "GPUQ_LD_STREAM" : "*") + "(((const uint32_t*)col" + cs + ".data) + r" + cs + ");"); L("const uint32_t " + A + " = w." + A + ";"); break;
      case CC_I64: fieldR("u64", A); LL("w." + A + " = " + std::string(f.side == 0 ? "GPUQ_LD_STREAM" : "*") + "(((const u64*)col" + cs + ".data) + r" + cs + ");"); L("const u64 " + A + " = w." + A + ";"); break;
      case CC_I128:
        if (narrow128[c]) {   // declared precision fits 64 bits: only the low half of the 16-byte value is ever used
          fieldR("u64", A); LL("w." + A + " = ((const u64*)col" + cs + ".data)[2 * (size_t)r" + cs + "];"); L("const ulonglong2 " + A + " = make_ulonglong2(w." + A + ", 0ull);");
        } else { fieldR("ulonglong2", A); LL("w." + A + " = ((const ulonglong2*)col" + cs + ".data)[r" + cs + "];"); L("const ulonglong2 " + A + " = w." + A + ";"); }
        break;
      case CC_BIT: fieldR("uint32_t", A); LL("w." + A + " = ((const uint8_t*)col" + cs + ".data)[r" + cs + " >> 3];"); L("const uint32_t " + A + " = w." + A + ";"); break;
      case CC_F32: fieldR("uint32_t", A); LL("w." + A + " = " + std::string(f.side == 0 ? "GPUQ_LD_STREAM" : "*") + "(((const uint32_t*)col" + cs + ".data) + r" + cs + ");"); L("const uint32_t " + A + " = w." + A + ";"); break;
      case CC_I8: case CC_U8: fieldR("uint32_t", A); LL("w." + A + " = ((const uint8_t*)col" + cs + ".data)[r" + cs + "];"); L("const uint32_t " + A + " = w." + A + ";"); break;
      case CC_I16: case CC_U16: fieldR("uint32_t", A); LL("w." + A + " = ((const uint16_t*)col" + cs + ".data)[r" + cs + "];"); L("const uint32_t " + A + " = w." + A + ";"); break;
      case CC_STR: {
        const std::string oa = "o" + cs + "a", ob = "o" + cs + "b";
        fieldR("int32_t", oa); fieldR("int32_t", ob);
        if (f.side == 0) {
          // offsets one stage early, first byte in the load stage
          fieldP("int32_t", oa); fieldP("int32_t", ob);
          LP("q." + oa + " = P.cols[" + cs + "].offsets[row0]; q." + ob + " = P.cols[" + cs + "].offsets[row0 + 1];");
          LL("w." + oa + " = q." + oa + "; w." + ob + " = q." + ob + ";");
          // the (up to three) aligned 8-byte words that hold the first 15 bytes, in the load stage: unconditional loads through selected
          // addresses (a word the string does not reach is read from the program block instead: a branch here would make the wave wait)
          fieldR("u64", "sw" + cs + "a"); fieldR("u64", "sw" + cs + "b"); fieldR("u64", "sw" + cs + "c");
          LL("{ const unsigned long long sa = (unsigned long long)col" + cs + ".data + (unsigned long long)(uint32_t)q." + oa + "; const int sl = q." + ob + " - q." + oa + "; const uint32_t sn = (uint32_t)(sl < 15 ? sl : 15), span = (uint32_t)(sa & 7ull) + sn;");
          LL("  const u64* swp = (const u64*)(sa & ~7ull); const u64* dummy = (const u64*)P.code;");
          LL("  w.sw" + cs + "a = *(sl > 0 ? swp : dummy); w.sw" + cs + "b = *(span > 8u ? swp + 1 : dummy); w.sw" + cs + "c = *(span > 16u ? swp + 2 : dummy); }");
        } else {
          LL("w." + oa + " = col" + cs + ".offsets[r" + cs + "]; w." + ob + " = col" + cs + ".offsets[r" + cs + " + 1];");
        }
        L("const int32_t " + oa + " = w." + oa + ", " + ob + " = w." + ob + ";");
        break;
      }
    }
  }
  // ---- first byte of every string column
  for (size_t c = 0; c < C.col_field.size(); ++c) {
    const Field& f = schema_.fields[C.col_field[c]];
    if (f.raw128 || f.type.id != T_UTF8) continue;
    const std::string cs = std::to_string(c);
    L("const int32_t len" + cs + " = o" + cs + "b - o" + cs + "a;");
    L("const uint8_t* sp" + cs + " = (const uint8_t*)col" + cs + ".data + o" + cs + "a;");
    (void)0;
  }
  // ---- phase B2: typed column values
  std::map<int, std::string> col_null;
  for (Node* n : order) {
    if (n->kind != Node::COL) continue;
    int c = -1; for (size_t k = 0; k < C.col_field.size(); ++k) if (C.col_field[k] == n->col) c = (int)k;
    const Field& f = schema_.fields[n->col];
    const std::string cs = std::to_string(c);
    std::string nn = "false";
    if (f.nullable || (f.side > 0 && !f.dense)) {
      std::string e;
      if (f.side > 0 && !f.dense) e = "!ok" + cs;
      if (f.nullable) e += std::string(e.empty() ? "" : " || ") + "(col" + cs + ".validity && !((vb" + cs + " >> (r" + cs + " & 7)) & 1u))";
      L("const bool n_c" + cs + " = " + e + ";");
      nn = "n_c" + cs;
    }
    nul[n] = nn;
    const int cls = f.raw128 ? (int)CC_I128 : col_class_for(f.type);
    if (is_str(n)) {
      if (f.raw128) { L("const u64 c" + cs + "_lo = a" + cs + ".x, c" + cs + "_hi = a" + cs + ".y;"); }
      else {
        if (!(c < C.col_loose.size() && C.col_loose[c])) L("if (len" + cs + " > 15 && P.flags) atomicOr(P.flags, FLAG_STR_TRUNC);");
        L("u64 c" + cs + "_hi, c" + cs + "_lo;");
        if (f.side == 0) L("str15_assemble(w.sw" + cs + "a, w.sw" + cs + "b, w.sw" + cs + "c, (uint32_t)(((unsigned long long)col" + cs + ".data + (unsigned long long)(uint32_t)o" + cs + "a) & 7ull), len" + cs + " < 15 ? len" + cs + " : 15, c" + cs + "_hi, c" + cs + "_lo);");
        else L("load_str15((const uint8_t*)col" + cs + ".data, o" + cs + "a, len" + cs + ", c" + cs + "_hi, c" + cs + "_lo);");
        L("c" + cs + "_lo |= (u64)(len" + cs + " < 255 ? len" + cs + " : 255);");
      }
      name[n] = "c" + cs;
      continue;
    }
    std::string v;
    switch (cls) {
      case CC_I32: v = "(i64)(int32_t)a" + cs; break;
      case CC_U32: case CC_U8: case CC_U16: v = "(i64)a" + cs; break;
      case CC_I8: v = "(i64)(int8_t)a" + cs; break;
      case CC_I16: v = "(i64)(int16_t)a" + cs; break;
      case CC_F32: v = "(double)__uint_as_float(a" + cs + ")"; break;
      case CC_I64: v = is_f(n) ? "__longlong_as_double((i64)a" + cs + ")" : "(i64)a" + cs; break;
      case CC_I128:
        if (is_f(n)) v = "__longlong_as_double((i64)a" + cs + ".x)";
        else if (is_b(n)) v = "(a" + cs + ".x != 0)";
        else v = wide(n) ? "mk128(a" + cs + ".x, a" + cs + ".y)" : "(i64)a" + cs + ".x";
        break;
      case CC_BIT: v = "(((a" + cs + " >> (r" + cs + " & 7)) & 1u) != 0)"; break;
    }
    L("const " + ctype(n) + " c" + cs + " = " + v + ";");
    name[n] = "c" + cs;
  }
  // ---- expressions
  int vid = 0;
  auto V = [&](const NodeP& x) { return name.at(rep(x.get())); };
  auto N = [&](const NodeP& x) { return nul.at(rep(x.get())); };
  auto orn = [&](std::initializer_list<std::string> xs) { std::string r; for (auto& x : xs) if (x != "false") r += (r.empty() ? "" : " || ") + x; return r.empty() ? std::string("false") : r; };
  auto as128 = [&](const NodeP& x) { Node* r = rep(x.get()); return is_str(r) ? "mk128(" + name.at(r) + "_lo, " + name.at(r) + "_hi)" : "(i128)" + name.at(r); };
  Node* predn = pred_ ? rep(pred_.get()) : nullptr;
  bool pred_done = false;
  auto emit_pred = [&]() { if (predn && !pred_done && name.count(predn)) { L("if (" + (nul.at(predn) == "false" ? std::string("") : "(" + nul.at(predn) + ") || ") + "!" + name.at(predn) + ") return false;"); pred_done = true; } };
  emit_pred();
  for (Node* n : order) {
    if (n->kind == Node::COL) continue;
    const std::string t = "t" + std::to_string(vid++);
    if (n->kind == Node::LIT) {
      if (n->lit_null) { nul[n] = "true"; if (is_str(n)) { L("const u64 " + t + "_lo = 0, " + t + "_hi = 0;"); } else L("const " + ctype(n) + " " + t + " = 0;"); name[n] = t; continue; }
      nul[n] = "false";
      if (is_str(n)) L("const u64 " + t + "_lo = " + std::to_string(n->lit_lo) + "ull, " + t + "_hi = " + std::to_string(n->lit_hi) + "ull;");
      else if (is_f(n)) L("const double " + t + " = __longlong_as_double((i64)" + std::to_string(n->lit_lo) + "ull);");
      else if (is_b(n)) L("const bool " + t + " = " + (n->lit_lo ? "true" : "false") + ";");
      else if (wide(n)) L("const i128 " + t + " = mk128(" + std::to_string(n->lit_lo) + "ull, " + std::to_string(n->lit_hi) + "ull);");
      else L("const i64 " + t + " = (i64)" + std::to_string(n->lit_lo) + "ull;");
      name[n] = t; continue;
    }
    const NodeP& a = n->ch[0];
    const NodeP& b = n->ch.size() > 1 ? n->ch[1] : n->ch[0];
    const std::string T = ctype(n);
    std::string e, ne = orn({N(a), n->ch.size() > 1 ? N(b) : std::string("false")});
    auto cmp = [&](const char* op) {
      Node* ra = rep(a.get());
      if (is_str(ra)) {
        if (std::string(op) == "==") return "(" + V(a) + "_lo == " + V(b) + "_lo && " + V(a) + "_hi == " + V(b) + "_hi)";
        if (std::string(op) == "!=") return "(" + V(a) + "_lo != " + V(b) + "_lo || " + V(a) + "_hi != " + V(b) + "_hi)";
        return "(" + as128(a) + " " + op + " " + as128(b) + ")";
      }
      if (is_b(ra)) return "((int)" + V(a) + " " + op + " (int)" + V(b) + ")";
      return "(" + V(a) + " " + op + " " + V(b) + ")";
    };
    auto fkey = [&](const NodeP& x) { return "f64_total_key((u64)__double_as_longlong(" + V(x) + "))"; };
    bool two_vars = false;
    switch (n->op) {
      case OP_ADD: e = wide(n) ? "(" + as128(a) + " + " + as128(b) + ")" : "(i64)((u64)" + V(a) + " + (u64)" + V(b) + ")"; break;
      case OP_SUB: e = wide(n) ? "(" + as128(a) + " - " + as128(b) + ")" : "(i64)((u64)" + V(a) + " - (u64)" + V(b) + ")"; break;
      case OP_MUL: case OP_MULW:
        e = wide(n) ? "(i128)((u128)" + as128(a) + " * (u128)" + as128(b) + ")" : "(i64)((u64)" + V(a) + " * (u64)" + V(b) + ")"; break;
      case OP_NEG: e = wide(n) ? "(-" + as128(a) + ")" : "(i64)(0ull - (u64)" + V(a) + ")"; ne = N(a); break;
      case OP_DIV: case OP_MOD: {
        const bool narrow = !wide(rep(a.get())) && !wide(rep(b.get()));
        L("const bool " + t + "_z = (" + V(b) + " == 0);");
        if (narrow) e = "(" + t + "_z ? (i64)0 : (i64)(" + V(a) + (n->op == OP_DIV ? " / " : " % ") + "(" + t + "_z ? (i64)1 : (i64)" + V(b) + ")))";
        else { L("i128 " + t + "_q = 0, " + t + "_r = 0; if (!" + t + "_z) divmod128(" + as128(a) + ", " + as128(b) + ", " + t + "_q, " + t + "_r);");
               e = std::string("(") + T + ")" + t + (n->op == OP_DIV ? "_q" : "_r"); }
        ne = orn({N(a), N(b), t + "_z"});
        break;
      }
      case OP_EQ: e = cmp("=="); break;  case OP_NE: e = cmp("!="); break;  case OP_LT: e = cmp("<"); break;
      case OP_LE: e = cmp("<="); break;  case OP_GT: e = cmp(">"); break;   case OP_GE: e = cmp(">="); break;
      case OP_FADD: e = "(" + V(a) + " + " + V(b) + ")"; break;  case OP_FSUB: e = "(" + V(a) + " - " + V(b) + ")"; break;
      case OP_FMUL: e = "(" + V(a) + " * " + V(b) + ")"; break;  case OP_FDIV: e = "(" + V(a) + " / " + V(b) + ")"; break;
      case OP_FNEG: e = "__longlong_as_double(__double_as_longlong(" + V(a) + ") ^ (i64)0x8000000000000000ull)"; ne = N(a); break;
      case OP_FSQRT: e = "sqrt(" + V(a) + ")"; ne = N(a); break;
      case OP_FEQ: e = "(" + fkey(a) + " == " + fkey(b) + ")"; break;  case OP_FNE: e = "(" + fkey(a) + " != " + fkey(b) + ")"; break;
      case OP_FLT: e = "(" + fkey(a) + " < " + fkey(b) + ")"; break;   case OP_FLE: e = "(" + fkey(a) + " <= " + fkey(b) + ")"; break;
      case OP_FGT: e = "(" + fkey(a) + " > " + fkey(b) + ")"; break;   case OP_FGE: e = "(" + fkey(a) + " >= " + fkey(b) + ")"; break;
      case OP_I2F: {
        Node* ra = rep(a.get());
        if (is_b(ra)) e = "(double)(int)" + V(a);
        else if (!wide(ra)) e = "(double)" + V(a);
        else e = "(((i64)((u128)" + V(a) + " >> 64) == ((i64)(u64)" + V(a) + " >> 63)) ? (double)(i64)(u64)" + V(a) + " : ((double)(i64)((u128)" + V(a) + " >> 64) * 18446744073709551616.0 + (double)(u64)" + V(a) + "))";
        ne = N(a); break;
      }
      case OP_F2I: e = "(i64)" + V(a); ne = N(a); break;
      case OP_F32R: e = "(double)(float)" + V(a); ne = N(a); break;
      case OP_AND: {
        const std::string af = "(!(" + N(a) + ") && !" + V(a) + ")", bf = "(!(" + N(b) + ") && !" + V(b) + ")";
        e = "!(" + af + " || " + bf + ")"; ne = "(!(" + af + " || " + bf + ") && (" + orn({N(a), N(b)}) + "))"; break;
      }
      case OP_OR: {
        const std::string at = "(!(" + N(a) + ") && " + V(a) + ")", bt = "(!(" + N(b) + ") && " + V(b) + ")";
        e = "(" + at + " || " + bt + ")"; ne = "(!(" + at + " || " + bt + ") && (" + orn({N(a), N(b)}) + "))"; break;
      }
      case OP_NOT: e = "!" + V(a); ne = N(a); break;
      case OP_ISNULL: e = "(" + N(a) + ")"; ne = "false"; break;
      case OP_ISNOTNULL: e = "!(" + N(a) + ")"; ne = "false"; break;
      case OP_SELECT: {
        const NodeP& f = n->ch[2];
        L("const bool " + t + "_c = !(" + N(a) + ") && " + V(a) + ";");
        if (is_str(n)) { L("const u64 " + t + "_lo = " + t + "_c ? " + V(b) + "_lo : " + V(f) + "_lo, " + t + "_hi = " + t + "_c ? " + V(b) + "_hi : " + V(f) + "_hi;"); two_vars = true; }
        else e = "(" + t + "_c ? (" + T + ")" + V(b) + " : (" + T + ")" + V(f) + ")";
        ne = (N(b) == "false" && N(f) == "false") ? "false" : "(" + t + "_c ? (" + N(b) + ") : (" + N(f) + "))";
        break;
      }
      case OP_SHL: e = "(i128)((u128)" + as128(a) + " << " + std::to_string(n->imm) + ")"; ne = N(a); break;
      case OP_DATEPART: e = "date_part_of_days((i64)" + V(a) + ", " + std::to_string(n->imm) + ")"; ne = N(a); break;
      case OP_SUBSTR: {
        L("bool " + t + "_bad = false; const u128 " + t + "_x = substr_packed(((u128)" + V(a) + "_hi << 64) | (u128)" + V(a) + "_lo, " + std::to_string(n->imm & 0xFFu) + "u, " +
          std::to_string((n->imm >> 8) & 0xFFu) + "u, " + t + "_bad);");
        L("if (" + t + "_bad && !(" + N(a) + ") && P.flags) atomicOr(P.flags, FLAG_STR_TRUNC);");
        L("const u64 " + t + "_lo = (u64)" + t + "_x, " + t + "_hi = (u64)(" + t + "_x >> 64);");
        two_vars = true; ne = N(a); break;
      }
      case OP_BOR: e = "(" + as128(a) + " | " + as128(b) + ")"; break;
      case OP_NULLIF0: {
        Node* rb = rep(b.get());
        const std::string bz = is_str(rb) ? "(" + V(b) + "_lo == 0 && " + V(b) + "_hi == 0)" : (is_f(rb) ? "(__double_as_longlong(" + V(b) + ") == 0)" : "(" + V(b) + " == 0)");
        if (is_str(n)) { L("const u64 " + t + "_lo = " + V(a) + "_lo, " + t + "_hi = " + V(a) + "_hi;"); two_vars = true; } else e = "(" + T + ")" + V(a);
        ne = orn({N(a), N(b), bz}); break;
      }
      case OP_COALESCE0:
        if (is_str(n)) { L("const u64 " + t + "_lo = (" + N(a) + ") ? 0 : " + V(a) + "_lo, " + t + "_hi = (" + N(a) + ") ? 0 : " + V(a) + "_hi;"); two_vars = true; }
        else e = "((" + N(a) + ") ? (" + T + ")0 : (" + T + ")" + V(a) + ")";
        ne = "false"; break;
      default: throw std::runtime_error("jit: unsupported op " + std::to_string(n->op));
    }
    if (!two_vars) L("const " + T + " " + t + " = (" + T + ")" + e + ";");
    if (!n->nullable || ne == "false") nul[n] = "false";
    else { L("const bool " + t + "_n = " + ne + ";"); nul[n] = t + "_n"; }
    name[n] = t;
    emit_pred();
  }
  emit_pred();
  // ---- outputs into the registers the sink reads
  L("uint32_t nm = 0;");
  std::set<int> done;
  for (size_t k = 0; k < outs_.size(); ++k) {
    Node* o = rep(outs_[k].get());
    const int r = C.out_reg[k];
    if (done.count(r)) continue;
    done.insert(r);
    const std::string rs = std::to_string(r);
    if (is_str(o)) L("rlo[" + rs + "] = " + name.at(o) + "_lo; rhi[" + rs + "] = " + name.at(o) + "_hi;");
    else if (is_f(o)) L("rlo[" + rs + "] = (u64)__double_as_longlong(" + name.at(o) + "); rhi[" + rs + "] = 0;");
    else if (is_b(o)) L("rlo[" + rs + "] = " + name.at(o) + " ? 1ull : 0ull; rhi[" + rs + "] = 0;");
    else if (wide(o)) L("rlo[" + rs + "] = (u64)" + name.at(o) + "; rhi[" + rs + "] = (u64)((u128)" + name.at(o) + " >> 64);");
    else L("rlo[" + rs + "] = (u64)" + name.at(o) + "; rhi[" + rs + "] = (u64)(" + name.at(o) + " >> 63);");
    if (nul.at(o) != "false") L("if (" + nul.at(o) + ") nm |= " + std::to_string(1u << r) + "u;");
  }
  L("rnulls = nm;");
  L("return true;");
  std::string R = "struct JitPre {\n" + FP + "  int32_t pad_;\n};\nstruct JitRaw {\n" + FR + "  int32_t pad_;\n};\n";
  R += "__device__ __forceinline__ void gpuq_jit_pre(const DevProgram& P, i64 pos, JitPre& q) {\n" + SP + "  (void)P; (void)q;\n}\n";
  R += "__device__ __forceinline__ void gpuq_jit_load(const DevProgram& P, i64 pos, const JitPre& q, JitRaw& w) {\n" + SLD + "  (void)q; (void)w;\n}\n";
  // The empty asm statements make the wave wait for the whole record at one point that every path crosses; without them a
  // field first used under a branch (predicate, inactive tail lanes) stays "pending" on the bypass path and the next loop
  // iteration has to drain all loads before it may reuse the registers.
  R += "__device__ __forceinline__ bool gpuq_jit_compute(const DevProgram& P, i64 pos, const JitRaw& w, GPUQ_REGS_PARAM) {\n" + touch + S + "}\n";
  R += "__device__ __forceinline__ bool gpuq_jit_eval(const DevProgram& P, i64 pos, GPUQ_REGS_PARAM) {\n"
       "  JitPre q; gpuq_jit_pre(P, pos, q);\n  JitRaw w; gpuq_jit_load(P, pos, q, w);\n  return gpuq_jit_compute(P, pos, w, GPUQ_REGS);\n}\n";
  return R;
}

}  // namespace gpuq
