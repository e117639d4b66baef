// Host-side expression compiler: PhysicalExpr tree (JSON mirror of
// ballista/core/proto/datafusion.proto:1142-1180 PhysicalExprNode) -> typed DAG -> register bytecode
// for the device interpreter (gpuq_dev.h).
//
// Type rules restate DataFusion v34 / arrow 49 behaviour [UPSTREAM-KNOWLEDGE, SURVEY.md §8c: that
// source is not in the container]:
//   Int64 in decimal context -> Decimal128(20,0); Int32 -> Decimal128(10,0)
//   add/sub: s = max(s1,s2), p = min(38, max(p1-s1,p2-s2) + s + 1)
//   mul:     s = s1+s2,      p = min(38, p1+p2+1)
//   SUM(decimal(p,s)) -> Decimal128(min(38,p+10), s);  AVG -> Decimal128(min(38,p+4), min(38,s+4)),
//   AVG value = sum * 10^(s_avg - s) / count with truncating i128 division
//   SUM/AVG over integers: SUM -> Int64 (wrapping), AVG -> Float64 (f64 sum / count)
#pragma once
#include "gpuq_dev.h"
#include "json.h"
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace gpuq {

enum TypeId : int32_t {
  T_NULL = 0, T_BOOL = 1, T_INT32 = 2, T_INT64 = 3, T_DATE32 = 4, T_FLOAT64 = 5,
  T_DECIMAL128 = 6, T_UTF8 = 7, T_UINT32 = 8, T_UINT64 = 9,
  // the second wave (datafusion.proto:1004-1040 ArrowType): narrow integers and Float32 stay narrow in memory (column classes of
  // their own) and are widened on load: registers hold them as the first-wave types do (64-bit integers, doubles)
  T_INT8 = 10, T_INT16 = 11, T_UINT8 = 12, T_UINT16 = 13, T_FLOAT32 = 14,
  T_TIMESTAMP = 15,   // p = TimeUnit (0 Second, 1 Millisecond, 2 Microsecond, 3 Nanosecond); 8-byte count since the epoch; tz is schema metadata
  T_DATE64 = 16,      // milliseconds since the epoch
};

struct DType {
  int32_t id = T_NULL;
  int32_t p = 0, s = 0;
  bool operator==(const DType& o) const { return id == o.id && (id != T_DECIMAL128 || (p == o.p && s == o.s)) && (id != T_TIMESTAMP || p == o.p); }
  bool operator!=(const DType& o) const { return !(*this == o); }
  bool is_int() const { return id == T_INT32 || id == T_INT64 || id == T_UINT32 || id == T_UINT64 || id == T_INT8 || id == T_INT16 || id == T_UINT8 || id == T_UINT16; }
  bool is_unsigned() const { return id == T_UINT8 || id == T_UINT16 || id == T_UINT32 || id == T_UINT64; }
  bool is_temporal() const { return id == T_DATE32 || id == T_DATE64 || id == T_TIMESTAMP; }
  bool is_decimal() const { return id == T_DECIMAL128; }
  bool is_float() const { return id == T_FLOAT64 || id == T_FLOAT32; }
  std::string to_string() const;
};
DType dtype_from_json(const Json& j);
Json dtype_to_json_text(const DType& t);
DType dtype_from_arrow_format(const char* f, bool* large_utf8);   // Arrow C data interface format string; large_utf8 = nullptr refuses "U"
std::string arrow_format_of(const DType& t);
int col_class_for(const DType& t);   // ColClass used to load / store this type
int type_width(const DType& t);      // bytes per value in the fixed-width device layout (Utf8: 16, packed)

struct Field {
  std::string name;
  DType type;
  bool nullable = true;
  int side = 0;        // 0: addressed by the driving position; k>0: through index vector k
  int raw128 = 0;      // 1: device column is a raw (lo,hi) pair array regardless of the logical type
  int dense = 0;       // side > 0 only: the index vector never holds NULL_ROW, so going through it adds no nulls
};
struct Schema {
  std::vector<Field> fields;
  int index_of(const std::string& name) const;
};
Schema schema_from_json(const Json& j);

struct Node;
typedef std::shared_ptr<Node> NodeP;
struct Node {
  enum Kind { COL, LIT, OPN } kind = LIT;
  int op = OP_NOP;           // OPN
  std::vector<NodeP> ch;     // operands (<= 3)
  uint32_t imm = 0;          // OP_SHL amount
  DType type;
  bool nullable = true;
  int bits = 127;            // bound on the magnitude: |value| < 2^bits (integers / decimals)
  int col = -1;              // COL: schema field index
  u64 lit_lo = 0, lit_hi = 0; bool lit_null = false;   // LIT
  bool lit_long = false;     // LIT: a Utf8 literal beyond 15 bytes -- lit_lo / lit_hi pack its first 15; only a comparison may consume it (binary())
  std::string key;           // canonical text, CSE key
};

struct CompiledProgram {
  DevCode code{};
  int n_insns = 0;
  std::vector<int> col_field;   // program column slot -> schema field index
  std::vector<bool> col_loose;  // Utf8 slot only compared for (in)equality with literals / tested for NULL: values beyond 15 bytes are fine (CC_STRQ)
  int pred_reg = -1;
  std::vector<int> out_reg;
  std::vector<DType> out_type;
  std::vector<bool> out_nullable;
  std::vector<std::string> out_key;
  std::vector<int> out_bits;    // |value| < 2^bits for every output (from the declared types; 127 = unknown)
  std::string jit_src;          // C++ source of gpuq_jit_eval() for this program (typed, straight-line)
};

class ExprCompiler {
 public:
  explicit ExprCompiler(const Schema& schema) : schema_(schema) {}
  // --- AST construction (typed, coercing)
  NodeP from_json(const Json& e);
  NodeP column(int field_index);
  NodeP lit_int(DType t, i128 v);
  NodeP lit_f64(double v);
  NodeP lit_f32(float v);
  NodeP floor_div(NodeP e, i64 d, DType rt);
  NodeP rescale_time(NodeP e, i64 from_per_s, i64 to_per_s, DType rt);
  NodeP lit_null(DType t);
  NodeP lit_str(const std::string& s);
  NodeP binary(const std::string& op, NodeP l, NodeP r);
  NodeP cast(NodeP e, DType to);
  NodeP not_(NodeP e);
  NodeP is_null(NodeP e, bool negate);
  NodeP negative(NodeP e);
  NodeP select(NodeP c, NodeP t, NodeP f);
  NodeP raw(int op, DType t, bool nullable, int bits, std::vector<NodeP> ch, uint32_t imm = 0);
  NodeP rescale(NodeP e, int new_scale);         // decimal scale change (exact when increasing)
  NodeP coalesce0(NodeP e);
  NodeP nullif0(NodeP e, NodeP guard);
  // --- program assembly
  void add_predicate(NodeP e);                    // AND-ed with any previous predicate
  int add_output(NodeP e);                        // value must be live at program end; returns output slot
  CompiledProgram finish();
  const Schema& schema() const { return schema_; }

 private:
  Schema schema_;
  NodeP pred_;
  std::vector<NodeP> outs_;
  std::map<std::string, NodeP> interned_;
  NodeP intern(NodeP n);
  std::string jit_source(const CompiledProgram& C, const std::vector<Node*>& order, const std::map<Node*, int>& reg);
};

i128 pow10_i128(int k);
int bits_for_precision(int p);
i128 parse_i128(const std::string& s);
std::string i128_to_string(i128 v);
bool pack_str15(const std::string& s, u64& lo, u64& hi);

}  // namespace gpuq
