// Minimal JSON reader/writer for sourmash signature files (host only).
// Replaces serde_json as used by src/core/src/signature.rs:569-659,786-794 and
// src/core/src/sketch/minhash.rs:103-184.  Numbers keep their source text so
// 64-bit hashes round-trip exactly.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "smg_errors.hpp"

namespace smg {
namespace json {

struct Value;
using ValuePtr = std::unique_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    std::string text;                                   // Number: source token; String: decoded
    std::vector<ValuePtr> items;                        // Array
    std::vector<std::pair<std::string, ValuePtr>> members;  // Object, in file order

    const Value* get(const std::string& key) const {
        for (auto& m : members) if (m.first == key) return m.second.get();
        return nullptr;
    }
    uint64_t as_u64() const {
        if (kind != Number) throw Error(E_SERDE, "JSON: expected a number");
        // tolerate "123" and "123.0"
        const char* s = text.c_str();
        if (text.find_first_of(".eE") != std::string::npos) {
            const double d = strtod(s, nullptr);
            if (d < 0) throw Error(E_SERDE, "JSON: expected an unsigned integer");
            return (uint64_t)d;
        }
        if (*s == '-') throw Error(E_SERDE, "JSON: expected an unsigned integer");
        return strtoull(s, nullptr, 10);
    }
    double as_f64() const {
        if (kind != Number) throw Error(E_SERDE, "JSON: expected a number");
        return strtod(text.c_str(), nullptr);
    }
    const std::string& as_str() const {
        if (kind != String) throw Error(E_SERDE, "JSON: expected a string");
        return text;
    }
};

class Parser {
  public:
    Parser(const char* p, size_t n) : p_(p), end_(p + n) {}

    ValuePtr parse_document() {
        ValuePtr v = parse_value();
        skip_ws();
        if (p_ != end_) fail("trailing characters");
        return v;
    }

  protected:
    [[noreturn]] void fail(const char* what) { throw Error(E_SERDE, std::string("JSON parse error: ") + what); }
    void skip_ws() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_; }
    bool eat(char c) { skip_ws(); if (p_ < end_ && *p_ == c) { ++p_; return true; } return false; }
    void expect_word(const char* w) {
        for (; *w; ++w, ++p_) if (p_ >= end_ || *p_ != *w) fail("bad literal");
    }

    static void put_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    uint32_t hex4() {
        if (end_ - p_ < 4) fail("short \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i, ++p_) {
            const char c = *p_;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }

    std::string parse_string_body() {
        std::string out;
        for (;;) {
            if (p_ >= end_) fail("unterminated string");
            const char c = *p_++;
            if (c == '"') return out;
            if (c != '\\') { out += c; continue; }
            if (p_ >= end_) fail("unterminated escape");
            const char e = *p_++;
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                uint32_t cp = hex4();
                if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                    p_ += 2;
                    const uint32_t lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                put_utf8(out, cp);
                break;
            }
            default: fail("bad escape");
            }
        }
    }

    ValuePtr parse_value() {
        skip_ws();
        if (p_ >= end_) fail("unexpected end");
        ValuePtr v(new Value());
        const char c = *p_;
        if (c == '{') {
            ++p_;
            v->kind = Value::Object;
            if (eat('}')) return v;
            for (;;) {
                skip_ws();
                if (p_ >= end_ || *p_ != '"') fail("expected object key");
                ++p_;
                std::string key = parse_string_body();
                if (!eat(':')) fail("expected ':'");
                v->members.emplace_back(std::move(key), parse_value());
                if (eat(',')) continue;
                if (eat('}')) return v;
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p_;
            v->kind = Value::Array;
            if (eat(']')) return v;
            for (;;) {
                v->items.push_back(parse_value());
                if (eat(',')) continue;
                if (eat(']')) return v;
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') { ++p_; v->kind = Value::String; v->text = parse_string_body(); return v; }
        if (c == 't') { expect_word("true"); v->kind = Value::Bool; v->b = true; return v; }
        if (c == 'f') { expect_word("false"); v->kind = Value::Bool; v->b = false; return v; }
        if (c == 'n') { expect_word("null"); v->kind = Value::Null; return v; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* s = p_;
            ++p_;
            while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) ++p_;
            v->kind = Value::Number;
            v->text.assign(s, (size_t)(p_ - s));
            return v;
        }
        fail("unexpected character");
    }

    const char* p_;
    const char* end_;
};

inline void write_string(std::string& out, const std::string& s) {
    out += '"';
    for (unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        default:
            if (c < 0x20) { char buf[8]; snprintf(buf, sizeof buf, "\\u%04x", c); out += buf; }
            else out += (char)c;
        }
    }
    out += '"';
}

inline void write_u64(std::string& out, uint64_t v) {
    char tmp[24];
    int i = 24;
    do { tmp[--i] = (char)('0' + v % 10); v /= 10; } while (v);
    out.append(tmp + i, (size_t)(24 - i));
}

}  // namespace json
}  // namespace smg
