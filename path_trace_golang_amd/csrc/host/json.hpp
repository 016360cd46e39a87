// json.hpp -- minimal JSON value, parser and writer for the scene files.
// Stands where Go's encoding/json stands in /root/reference/internal/scene/io.go:10-38.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace pthost {
namespace json {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<ValuePtr> arr;
    std::vector<std::pair<std::string, ValuePtr>> obj;  // insertion order, duplicate keys kept (last wins on lookup)

    // encoding/json field matching: exact key first, else case-insensitive; null counts as absent
    const Value *get(const std::string &key) const;
    double number(const std::string &key) const;   // 0 when absent / not a number
    long long integer(const std::string &key) const;
    std::string string(const std::string &key) const;
    bool boolean(const std::string &key) const;
};

// Parses one JSON document (json.Decoder.Decode semantics: trailing data after the first value is ignored).
// Throws std::runtime_error with a position on malformed input.
ValuePtr parse(const std::string &text);

// Writer used by scene::Save: two-space indent like json.Encoder.SetIndent("", "  ").
class Writer {
  public:
    void begin_object();
    void end_object();
    void begin_array();
    void end_array();
    void key(const std::string &k);
    void value(double v);
    void value(long long v);
    void value(bool v);
    void value(const std::string &v);
    void null();
    std::string str() const { return out_; }

  private:
    void comma_and_indent();
    void raw_string(const std::string &v);
    std::string out_;
    std::vector<int> counts_;
    bool after_key_ = false;
};

std::string format_number(double v);  // shortest round-trip decimal, Go's float formatting for JSON ('g'-like, integers without ".0")

}  // namespace json
}  // namespace pthost
