// Minimal JSON value + recursive-descent parser for operator / plan descriptors.
// (No third-party JSON library is in the image; descriptors are small.)
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <cstdlib>
#include <cstring>

namespace gpuq {

struct Json {
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  bool b = false;
  std::string s;      // STR, and the raw text of NUM (so 128-bit integers survive)
  std::vector<Json> a;
  std::vector<std::pair<std::string, Json>> o;

  bool is_null() const { return kind == NUL; }
  bool is_obj() const { return kind == OBJ; }
  bool is_arr() const { return kind == ARR; }
  bool is_str() const { return kind == STR; }
  bool is_num() const { return kind == NUM; }
  const Json* find(const std::string& k) const {
    for (auto& kv : o) if (kv.first == k) return &kv.second;
    return nullptr;
  }
  bool has(const std::string& k) const { const Json* j = find(k); return j && !j->is_null(); }
  const Json& at(const std::string& k) const {
    const Json* j = find(k);
    if (!j) throw std::runtime_error("descriptor: missing key '" + k + "'");
    return *j;
  }
  const std::string& str() const { if (kind != STR) throw std::runtime_error("descriptor: expected string"); return s; }
  long long i64() const {
    if (kind == NUM) return std::strtoll(s.c_str(), nullptr, 10);
    if (kind == BOOL) return b;
    if (kind == STR) return std::strtoll(s.c_str(), nullptr, 10);
    throw std::runtime_error("descriptor: expected number");
  }
  double f64() const { if (kind == NUM || kind == STR) return std::strtod(s.c_str(), nullptr); throw std::runtime_error("descriptor: expected number"); }
  bool boolean() const { if (kind == BOOL) return b; if (kind == NUM) return i64() != 0; throw std::runtime_error("descriptor: expected bool"); }
  long long get_i64(const std::string& k, long long dflt) const { const Json* j = find(k); return (j && !j->is_null()) ? j->i64() : dflt; }
  bool get_bool(const std::string& k, bool dflt) const { const Json* j = find(k); return (j && !j->is_null()) ? j->boolean() : dflt; }
  std::string get_str(const std::string& k, const std::string& dflt) const { const Json* j = find(k); return (j && j->is_str()) ? j->s : dflt; }
  // canonical text (used as a CSE key)
  std::string dump() const {
    switch (kind) {
      case NUL: return "null";
      case BOOL: return b ? "true" : "false";
      case NUM: return s;
      case STR: { std::string r = "\""; for (char c : s) { if (c == '"' || c == '\\') { r += '\\'; r += c; } else if (c == '\n') r += "\\n"; else if (c == '\t') r += "\\t"; else r += c; } return r + "\""; }
      case ARR: { std::string r = "["; for (size_t i = 0; i < a.size(); ++i) { if (i) r += ","; r += a[i].dump(); } return r + "]"; }
      case OBJ: { std::string r = "{"; for (size_t i = 0; i < o.size(); ++i) { if (i) r += ","; r += "\"" + o[i].first + "\":" + o[i].second.dump(); } return r + "}"; }
    }
    return "";
  }
};

class JsonParser {
 public:
  explicit JsonParser(const char* text) : p_(text), e_(text + std::strlen(text)) {}
  Json parse() { Json v = value(); ws(); if (p_ != e_) fail("trailing characters"); return v; }

 private:
  const char* p_; const char* e_;
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("descriptor JSON: ") + m); }
  void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_; }
  Json value() {
    ws();
    if (p_ >= e_) fail("unexpected end");
    Json v;
    char c = *p_;
    if (c == '{') {
      v.kind = Json::OBJ; ++p_; ws();
      if (p_ < e_ && *p_ == '}') { ++p_; return v; }
      for (;;) {
        ws(); Json k = value(); if (k.kind != Json::STR) fail("object key must be a string");
        ws(); if (p_ >= e_ || *p_ != ':') fail("expected ':'"); ++p_;
        Json val = value(); v.o.emplace_back(k.s, std::move(val));
        ws(); if (p_ < e_ && *p_ == ',') { ++p_; continue; }
        if (p_ < e_ && *p_ == '}') { ++p_; break; }
        fail("expected ',' or '}'");
      }
      return v;
    }
    if (c == '[') {
      v.kind = Json::ARR; ++p_; ws();
      if (p_ < e_ && *p_ == ']') { ++p_; return v; }
      for (;;) {
        v.a.push_back(value());
        ws(); if (p_ < e_ && *p_ == ',') { ++p_; continue; }
        if (p_ < e_ && *p_ == ']') { ++p_; break; }
        fail("expected ',' or ']'");
      }
      return v;
    }
    if (c == '"') {
      v.kind = Json::STR; ++p_;
      while (p_ < e_ && *p_ != '"') {
        if (*p_ == '\\') {
          ++p_; if (p_ >= e_) fail("bad escape");
          switch (*p_) {
            case 'n': v.s += '\n'; break; case 't': v.s += '\t'; break; case 'r': v.s += '\r'; break;
            case 'b': v.s += '\b'; break; case 'f': v.s += '\f'; break;
            case 'u': {
              if (e_ - p_ < 5) fail("bad \\u escape");
              unsigned cp = (unsigned)std::strtoul(std::string(p_ + 1, p_ + 5).c_str(), nullptr, 16); p_ += 4;
              if (cp < 0x80) v.s += (char)cp;
              else if (cp < 0x800) { v.s += (char)(0xC0 | (cp >> 6)); v.s += (char)(0x80 | (cp & 0x3F)); }
              else { v.s += (char)(0xE0 | (cp >> 12)); v.s += (char)(0x80 | ((cp >> 6) & 0x3F)); v.s += (char)(0x80 | (cp & 0x3F)); }
              break;
            }
            default: v.s += *p_;
          }
          ++p_;
        } else v.s += *p_++;
      }
      if (p_ >= e_) fail("unterminated string");
      ++p_; return v;
    }
    if (!std::strncmp(p_, "true", 4)) { v.kind = Json::BOOL; v.b = true; p_ += 4; return v; }
    if (!std::strncmp(p_, "false", 5)) { v.kind = Json::BOOL; v.b = false; p_ += 5; return v; }
    if (!std::strncmp(p_, "null", 4)) { p_ += 4; return v; }
    if (c == '-' || (c >= '0' && c <= '9')) {
      v.kind = Json::NUM; const char* s = p_; ++p_;
      while (p_ < e_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) ++p_;
      v.s.assign(s, p_); return v;
    }
    fail("unexpected character");
  }
};

}  // namespace gpuq
