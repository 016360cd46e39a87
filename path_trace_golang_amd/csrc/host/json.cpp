#include "json.hpp"

#include <cerrno>
#include <cstdint>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace pthost {
namespace json {

namespace {

bool iequals(const std::string &a, const std::string &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (std::tolower((unsigned char)a[i]) != std::tolower((unsigned char)b[i])) return false;
    return true;
}

struct Parser {
    const std::string &s;
    size_t i = 0;
    explicit Parser(const std::string &t) : s(t) {}

    [[noreturn]] void fail(const char *what) const {
        throw std::runtime_error(std::string("invalid JSON at offset ") + std::to_string(i) + ": " + what);
    }
    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++;
    }
    void append_utf8(std::string &o, unsigned cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (i + 4 > s.size()) fail("short \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; k++) {
            char c = s[i++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string_lit() {
        if (s[i] != '"') fail("expected string");
        i++;
        std::string o;
        while (true) {
            if (i >= s.size()) fail("unterminated string");
            char c = s[i++];
            if (c == '"') break;
            if ((unsigned char)c < 0x20) fail("control character in string");
            if (c != '\\') { o += c; continue; }
            if (i >= s.size()) fail("unterminated escape");
            char e = s[i++];
            switch (e) {
                case '"': o += '"'; break;
                case '\\': o += '\\'; break;
                case '/': o += '/'; break;
                case 'b': o += '\b'; break;
                case 'f': o += '\f'; break;
                case 'n': o += '\n'; break;
                case 'r': o += '\r'; break;
                case 't': o += '\t'; break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && i + 1 < s.size() && s[i] == '\\' && s[i + 1] == 'u') {
                        i += 2;
                        unsigned lo = hex4();
                        if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else cp = 0xFFFD;
                    }
                    append_utf8(o, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
        return o;
    }
    ValuePtr value() {
        ws();
        if (i >= s.size()) fail("unexpected end of input");
        auto v = std::make_shared<Value>();
        char c = s[i];
        if (c == '{') {
            v->kind = Value::Object;
            i++;
            ws();
            if (i < s.size() && s[i] == '}') { i++; return v; }
            while (true) {
                ws();
                if (i >= s.size() || s[i] != '"') fail("expected object key");
                std::string k = string_lit();
                ws();
                if (i >= s.size() || s[i] != ':') fail("expected ':'");
                i++;
                v->obj.emplace_back(k, value());
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v->kind = Value::Array;
            i++;
            ws();
            if (i < s.size() && s[i] == ']') { i++; return v; }
            while (true) {
                v->arr.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v->kind = Value::String;
            v->str = string_lit();
        } else if (s.compare(i, 4, "true") == 0) {
            v->kind = Value::Bool; v->b = true; i += 4;
        } else if (s.compare(i, 5, "false") == 0) {
            v->kind = Value::Bool; v->b = false; i += 5;
        } else if (s.compare(i, 4, "null") == 0) {
            v->kind = Value::Null; i += 4;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            size_t j = i;
            if (s[j] == '-') j++;
            if (j >= s.size() || !(s[j] >= '0' && s[j] <= '9')) fail("bad number");
            if (s[j] == '0') j++;
            else while (j < s.size() && std::isdigit((unsigned char)s[j])) j++;
            if (j < s.size() && s[j] == '.') {
                j++;
                if (j >= s.size() || !std::isdigit((unsigned char)s[j])) fail("bad fraction");
                while (j < s.size() && std::isdigit((unsigned char)s[j])) j++;
            }
            if (j < s.size() && (s[j] == 'e' || s[j] == 'E')) {
                j++;
                if (j < s.size() && (s[j] == '+' || s[j] == '-')) j++;
                if (j >= s.size() || !std::isdigit((unsigned char)s[j])) fail("bad exponent");
                while (j < s.size() && std::isdigit((unsigned char)s[j])) j++;
            }
            v->kind = Value::Number;
            // strtod is correctly rounded in glibc, like Go's strconv.ParseFloat
            v->lit = s.substr(i, j - i);
            v->num = std::strtod(v->lit.c_str(), nullptr);
            i = j;
        } else {
            fail("unexpected character");
        }
        return v;
    }
};

}  // namespace

const Value *Value::get(const std::string &key) const {
    if (kind != Object) return nullptr;
    const Value *found = nullptr;
    for (const auto &kv : obj)
        if (kv.first == key) found = kv.second.get();
    if (!found)
        for (const auto &kv : obj)
            if (iequals(kv.first, key)) found = kv.second.get();
    if (found && found->kind == Null) return nullptr;
    return found;
}
const char *Value::kind_name(Kind k) {
    switch (k) {
        case Null: return "null";
        case Bool: return "bool";
        case Number: return "number";
        case String: return "string";
        case Array: return "array";
        default: return "object";
    }
}

namespace {
[[noreturn]] void mismatch(const Value *v, const std::string &key, const char *want) {
    throw std::runtime_error(std::string("json: cannot unmarshal ") + Value::kind_name(v->kind) + " into field " + key +
                             " of type " + want);
}
}  // namespace

double Value::number(const std::string &key) const {
    const Value *v = get(key);
    if (!v) return 0.0;
    if (v->kind != Number) mismatch(v, key, "float64");
    if (std::isinf(v->num)) throw std::runtime_error("json: number " + v->lit + " out of range for field " + key + " of type float64");
    return v->num;
}
long long Value::integer(const std::string &key) const {
    const Value *v = get(key);
    if (!v) return 0;
    if (v->kind != Number) mismatch(v, key, "int");
    // strconv.ParseInt on the literal: digits with an optional minus sign, nothing else
    if (v->lit.find_first_of(".eE") != std::string::npos) mismatch(v, key, "int");
    errno = 0;
    const long long r = std::strtoll(v->lit.c_str(), nullptr, 10);
    if (errno == ERANGE) throw std::runtime_error("json: number " + v->lit + " out of range for field " + key + " of type int");
    return r;
}
std::string Value::string(const std::string &key) const {
    const Value *v = get(key);
    if (!v) return std::string();
    if (v->kind != String) mismatch(v, key, "string");
    return v->str;
}
bool Value::boolean(const std::string &key) const {
    const Value *v = get(key);
    if (!v) return false;
    if (v->kind != Bool) mismatch(v, key, "bool");
    return v->b;
}
const Value *Value::object(const std::string &key) const {
    const Value *v = get(key);
    if (v && v->kind != Object) mismatch(v, key, "struct");
    return v;
}
const Value *Value::array(const std::string &key) const {
    const Value *v = get(key);
    if (v && v->kind != Array) mismatch(v, key, "slice");
    return v;
}

ValuePtr parse(const std::string &text) {
    Parser p(text);
    return p.value();
}

std::string format_number(double v) {
    if (std::isnan(v) || std::isinf(v)) throw std::runtime_error("json: unsupported value (NaN/Inf)");
    if (v == 0) return std::signbit(v) ? "-0" : "0";
    // shortest decimal that round-trips
    char buf[40];
    for (int prec = 1; prec <= 17; prec++) {
        std::snprintf(buf, sizeof buf, "%.*g", prec, v);
        if (std::strtod(buf, nullptr) == v) break;
    }
    std::string s(buf);
    // Go switches to exponent form below 1e-6 and from 1e21; %g does so earlier: re-render plain decimals
    double a = std::fabs(v);
    if (a >= 1e-6 && a < 1e21 && s.find('e') != std::string::npos) {
        for (int prec = 0; prec <= 30; prec++) {
            std::snprintf(buf, sizeof buf, "%.*f", prec, v);
            if (std::strtod(buf, nullptr) == v) break;
        }
        s = buf;
    } else if (s.find('e') != std::string::npos) {
        // strconv's 'e' layout pads the exponent to two digits like C; encoding/json then cleans up e-09 to e-9
        // (encode.go floatEncoder) and leaves e+21 alone
        const size_t n = s.size();
        if (n >= 4 && s[n - 4] == 'e' && s[n - 3] == '-' && s[n - 2] == '0') s.erase(n - 2, 1);
    }
    return s;
}

void Writer::comma_and_indent() {
    if (after_key_) { after_key_ = false; return; }
    if (!counts_.empty()) {
        if (counts_.back() > 0) out_ += ",";
        counts_.back()++;
        out_ += "\n";
        out_.append(2 * counts_.size(), ' ');
    }
}
void Writer::begin_object() { comma_and_indent(); out_ += "{"; counts_.push_back(0); }
void Writer::end_object() {
    int n = counts_.back();
    counts_.pop_back();
    if (n > 0) { out_ += "\n"; out_.append(2 * counts_.size(), ' '); }
    out_ += "}";
}
void Writer::begin_array() { comma_and_indent(); out_ += "["; counts_.push_back(0); }
void Writer::end_array() {
    int n = counts_.back();
    counts_.pop_back();
    if (n > 0) { out_ += "\n"; out_.append(2 * counts_.size(), ' '); }
    out_ += "]";
}
// String escaping of encoding/json (encode.go, with the Encoder's default HTML escaping): ", \\, \b, \f, \n, \r, \t;
// other control characters, <, >, & and U+2028 / U+2029 as \u00xx / \u20xx; invalid UTF-8 becomes \ufffd.
void Writer::raw_string(const std::string &v) {
    out_ += '"';
    const size_t n = v.size();
    for (size_t i = 0; i < n;) {
        const unsigned char c = (unsigned char)v[i];
        if (c < 0x80) {
            switch (c) {
                case '"': out_ += "\\\""; break;
                case '\\': out_ += "\\\\"; break;
                case '\b': out_ += "\\b"; break;
                case '\f': out_ += "\\f"; break;
                case '\n': out_ += "\\n"; break;
                case '\r': out_ += "\\r"; break;
                case '\t': out_ += "\\t"; break;
                case '<': out_ += "\\u003c"; break;
                case '>': out_ += "\\u003e"; break;
                case '&': out_ += "\\u0026"; break;
                default:
                    if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); out_ += b; }
                    else out_ += (char)c;
            }
            i++;
            continue;
        }
        // multi-byte sequence: length, continuation bytes, no overlong forms, no surrogates, <= U+10FFFF
        int len = (c >= 0xf0) ? 4 : (c >= 0xe0) ? 3 : (c >= 0xc2) ? 2 : 0;
        uint32_t cp = 0;
        bool ok = len != 0 && c <= 0xf4 && i + (size_t)len <= n;
        if (ok) {
            cp = c & (0xffu >> (len + 1));
            for (int k = 1; k < len; k++) {
                const unsigned char d = (unsigned char)v[i + (size_t)k];
                if ((d & 0xc0) != 0x80) { ok = false; break; }
                cp = (cp << 6) | (d & 0x3fu);
            }
            if (ok && ((len == 3 && cp < 0x800) || (len == 4 && cp < 0x10000) || (cp >= 0xd800 && cp <= 0xdfff) || cp > 0x10ffff)) ok = false;
        }
        if (!ok) {
            out_ += "\\ufffd";
            i++;
        } else if (cp == 0x2028 || cp == 0x2029) {
            out_ += cp == 0x2028 ? "\\u2028" : "\\u2029";
            i += (size_t)len;
        } else {
            out_.append(v, i, (size_t)len);
            i += (size_t)len;
        }
    }
    out_ += '"';
}
void Writer::key(const std::string &k) {
    comma_and_indent();
    raw_string(k);
    out_ += ": ";
    after_key_ = true;
}
void Writer::value(double v) { comma_and_indent(); out_ += format_number(v); }
void Writer::value(long long v) { comma_and_indent(); out_ += std::to_string(v); }
void Writer::value(bool v) { comma_and_indent(); out_ += v ? "true" : "false"; }
void Writer::null() { comma_and_indent(); out_ += "null"; }
void Writer::value(const std::string &v) { comma_and_indent(); raw_string(v); }

}  // namespace json
}  // namespace pthost
