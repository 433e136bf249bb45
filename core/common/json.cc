#include "json.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace gpumon {
namespace {

struct Parser {
  const std::string& s;
  size_t p = 0;
  int depth = 0;
  explicit Parser(const std::string& t) : s(t) {}
  void ws() {
    while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++;
  }
  bool lit(const char* w) {
    size_t n = 0;
    while (w[n]) n++;
    if (s.compare(p, n, w) != 0) return false;
    p += n;
    return true;
  }
  bool string(std::string* out) {
    if (p >= s.size() || s[p] != '"') return false;
    p++;
    out->clear();
    while (p < s.size() && s[p] != '"') {
      char c = s[p++];
      if (c != '\\') {
        *out += c;
        continue;
      }
      if (p >= s.size()) return false;
      char e = s[p++];
      switch (e) {
        case 'n': *out += '\n'; break;
        case 't': *out += '\t'; break;
        case 'r': *out += '\r'; break;
        case 'b': *out += '\b'; break;
        case 'f': *out += '\f'; break;
        case 'u': {
          if (p + 4 > s.size()) return false;
          unsigned cp = (unsigned)strtoul(s.substr(p, 4).c_str(), nullptr, 16);
          p += 4;
          if (cp < 0x80) *out += (char)cp;
          else if (cp < 0x800) { *out += (char)(0xc0 | (cp >> 6)); *out += (char)(0x80 | (cp & 0x3f)); }
          else { *out += (char)(0xe0 | (cp >> 12)); *out += (char)(0x80 | ((cp >> 6) & 0x3f)); *out += (char)(0x80 | (cp & 0x3f)); }
          break;
        }
        default: *out += e;
      }
    }
    if (p >= s.size()) return false;
    p++;
    return true;
  }
  bool value(Json* out) {
    if (++depth > 64) return false;
    ws();
    if (p >= s.size()) return false;
    bool ok = false;
    char c = s[p];
    if (c == '{') {
      p++;
      out->type = Json::OBJ;
      ws();
      if (p < s.size() && s[p] == '}') { p++; ok = true; }
      else while (true) {
        ws();
        std::string k;
        if (!string(&k)) break;
        ws();
        if (p >= s.size() || s[p++] != ':') break;
        Json v;
        if (!value(&v)) break;
        out->obj.emplace_back(std::move(k), std::move(v));
        ws();
        if (p < s.size() && s[p] == ',') { p++; continue; }
        if (p < s.size() && s[p] == '}') { p++; ok = true; }
        break;
      }
    } else if (c == '[') {
      p++;
      out->type = Json::ARR;
      ws();
      if (p < s.size() && s[p] == ']') { p++; ok = true; }
      else while (true) {
        Json v;
        if (!value(&v)) break;
        out->arr.push_back(std::move(v));
        ws();
        if (p < s.size() && s[p] == ',') { p++; continue; }
        if (p < s.size() && s[p] == ']') { p++; ok = true; }
        break;
      }
    } else if (c == '"') {
      out->type = Json::STR;
      ok = string(&out->str);
    } else if (lit("true")) { out->type = Json::BOOL; out->b = true; ok = true; }
    else if (lit("false")) { out->type = Json::BOOL; out->b = false; ok = true; }
    else if (lit("null")) { out->type = Json::NUL; ok = true; }
    else {
      char* end = nullptr;
      out->num = strtod(s.c_str() + p, &end);
      if (end != s.c_str() + p) {
        out->type = Json::NUM;
        p = (size_t)(end - s.c_str());
        ok = true;
      }
    }
    depth--;
    return ok;
  }
};

}  // namespace

bool json_parse(const std::string& text, Json* out) {
  Parser ps(text);
  if (!ps.value(out)) return false;
  ps.ws();
  return ps.p == text.size();
}

std::string json_escape(const std::string& s) {
  std::string o;
  o.reserve(s.size() + 2);
  for (unsigned char c : s) {
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      default:
        if (c < 0x20) {
          char b[8];
          snprintf(b, sizeof b, "\\u%04x", c);
          o += b;
        } else {
          o += (char)c;
        }
    }
  }
  return o;
}

void JsonWriter::num(double v) {
  sep();
  if (std::isnan(v) || std::isinf(v)) { out_ += "null"; return; }
  char b[40];
  snprintf(b, sizeof b, "%.10g", v);
  out_ += b;
}

void JsonWriter::integer(long long v) {
  sep();
  char b[32];
  snprintf(b, sizeof b, "%lld", v);
  out_ += b;
}

}  // namespace gpumon
