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
    std::string lit;   // Number: the literal as written (integer fields take integer literals only, like encoding/json)
    std::string str;
    std::vector<ValuePtr> arr;
    std::vector<std::pair<std::string, ValuePtr>> obj;  // insertion order, duplicate keys kept (last wins on lookup)

    // encoding/json field matching: exact key first, else case-insensitive; null counts as absent
    const Value *get(const std::string &key) const;
    // Typed field reads with json.Unmarshal's rules: absent / null gives the zero value; a value of another JSON
    // type, a float literal that overflows float64 or a non-integer literal for an int field throws
    // std::runtime_error ("json: cannot unmarshal ..."), which fails the load like an UnmarshalTypeError.
    double number(const std::string &key) const;
    long long integer(const std::string &key) const;
    std::string string(const std::string &key) const;
    bool boolean(const std::string &key) const;
    const Value *object(const std::string &key) const;  // nullptr when absent / null
    const Value *array(const std::string &key) const;   // nullptr when absent / null
    static const char *kind_name(Kind k);
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
