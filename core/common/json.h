// Tiny JSON value + recursive-descent parser + string escaper.  Enough for the
// kubelet device-plugin checkpoint and for emitting REST responses.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace gpumon {

struct Json {
  enum Type { NUL, BOOL, NUM, STR, ARR, OBJ } type = NUL;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json* get(const std::string& key) const {
    if (type != OBJ) return nullptr;
    for (auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

// Returns false on malformed input.
bool json_parse(const std::string& text, Json* out);
std::string json_escape(const std::string& s);

// Streaming writer for the REST layer.
class JsonWriter {
 public:
  void begin_obj() { sep(); out_ += '{'; first_ = true; }
  void end_obj() { out_ += '}'; first_ = false; }
  void begin_arr() { sep(); out_ += '['; first_ = true; }
  void end_arr() { out_ += ']'; first_ = false; }
  void key(const char* k) { sep(); out_ += '"'; out_ += k; out_ += "\":"; first_ = true; }
  void str(const std::string& v) { sep(); out_ += '"'; out_ += json_escape(v); out_ += '"'; }
  void num(double v);
  void integer(long long v);
  void boolean(bool v) { sep(); out_ += v ? "true" : "false"; }
  void null() { sep(); out_ += "null"; }
  const std::string& text() const { return out_; }

 private:
  void sep() {
    if (!first_) out_ += ',';
    first_ = false;
  }
  std::string out_;
  bool first_ = true;
};

}  // namespace gpumon
