// json_min.h — a small recursive-descent JSON reader, enough for AIDA-X model
// files (objects, arrays, numbers, strings, true/false/null). The reference
// reads the same files with nlohmann::json bundled inside RTNeural
// (rt-neural-generic.cpp:970-974); that library is not in the reference tree,
// and nothing here derives from it.
#pragma once

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace aidax {

struct JsonError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Json {
public:
    enum Kind { Null, Bool, Number, String, Array, Object };

    Kind kind = Null;
    bool b = false;
    double num = 0.0;
    bool num_is_int = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;   // insertion order kept

    bool is_number() const { return kind == Number; }
    bool is_array() const { return kind == Array; }
    bool is_object() const { return kind == Object; }
    bool is_string() const { return kind == String; }
    bool is_null() const { return kind == Null; }

    // object access: missing key -> shared null (like operator[] on a const json)
    const Json& operator[](const char* key) const
    {
        static const Json null_json;
        if (kind != Object) return null_json;
        for (const auto& kv : obj)
            if (kv.first == key) return kv.second;
        return null_json;
    }
    // .at(): missing key is an error, like the predicates of model_variant.hpp:62-71
    const Json& at(const char* key) const
    {
        if (kind != Object) throw JsonError(std::string("json: not an object, key ") + key);
        for (const auto& kv : obj)
            if (kv.first == key) return kv.second;
        throw JsonError(std::string("json: key not found: ") + key);
    }
    const Json& at(size_t i) const
    {
        if (kind != Array || i >= arr.size()) throw JsonError("json: array index out of range");
        return arr[i];
    }
    const Json& at(int i) const { return at(static_cast<size_t>(i)); }
    const Json& back() const
    {
        if (kind != Array || arr.empty()) throw JsonError("json: back() on empty/non-array");
        return arr.back();
    }
    size_t size() const { return kind == Array ? arr.size() : kind == Object ? obj.size() : 0; }

    int as_int() const
    {
        if (kind != Number) throw JsonError("json: number expected");
        return static_cast<int>(num);
    }
    double as_double() const
    {
        if (kind != Number) throw JsonError("json: number expected");
        return num;
    }
    const std::string& as_string() const
    {
        if (kind != String) throw JsonError("json: string expected");
        return str;
    }

    static Json parse(const char* text, size_t len)
    {
        Parser p{text, text + len};
        p.skip_ws();
        Json v = p.value(0);
        p.skip_ws();
        if (p.cur != p.end) throw JsonError("json: trailing characters");
        return v;
    }

private:
    struct Parser {
        const char* cur;
        const char* end;

        void skip_ws()
        {
            while (cur < end && (*cur == ' ' || *cur == '\n' || *cur == '\t' || *cur == '\r')) ++cur;
        }
        [[noreturn]] void fail(const char* what) const { throw JsonError(std::string("json: ") + what); }

        Json value(int depth)
        {
            if (depth > 64) fail("nesting too deep");
            if (cur >= end) fail("unexpected end");
            switch (*cur) {
            case '{': return object(depth);
            case '[': return array(depth);
            case '"': { Json j; j.kind = String; j.str = string(); return j; }
            case 't': literal("true");  { Json j; j.kind = Bool; j.b = true; return j; }
            case 'f': literal("false"); { Json j; j.kind = Bool; j.b = false; return j; }
            case 'n': literal("null");  return Json();
            default:  return number();
            }
        }
        void literal(const char* w)
        {
            const size_t n = std::strlen(w);
            if (static_cast<size_t>(end - cur) < n || std::strncmp(cur, w, n) != 0) fail("bad literal");
            cur += n;
        }
        Json number()
        {
            const char* s = cur;
            if (cur < end && (*cur == '-' || *cur == '+')) ++cur;
            bool any = false, is_int = true;
            while (cur < end && ((*cur >= '0' && *cur <= '9') || *cur == '.' || *cur == 'e' || *cur == 'E' ||
                                 *cur == '-' || *cur == '+')) {
                if (*cur == '.' || *cur == 'e' || *cur == 'E') is_int = false;
                any = true;
                ++cur;
            }
            if (!any) fail("bad number");
            std::string tmp(s, cur);
            char* endp = nullptr;
            const double v = std::strtod(tmp.c_str(), &endp);
            if (endp == tmp.c_str() || *endp != '\0') fail("bad number");
            Json j; j.kind = Number; j.num = v; j.num_is_int = is_int;
            return j;
        }
        std::string string()
        {
            ++cur;  // opening quote
            std::string out;
            while (cur < end && *cur != '"') {
                if (*cur == '\\') {
                    if (++cur >= end) fail("bad escape");
                    switch (*cur) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (end - cur < 5) fail("bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; ++i) {
                            const char c = cur[i];
                            cp <<= 4;
                            if (c >= '0' && c <= '9') cp |= static_cast<unsigned>(c - '0');
                            else if (c >= 'a' && c <= 'f') cp |= static_cast<unsigned>(c - 'a' + 10);
                            else if (c >= 'A' && c <= 'F') cp |= static_cast<unsigned>(c - 'A' + 10);
                            else fail("bad \\u escape");
                        }
                        cur += 4;
                        if (cp < 0x80) out += static_cast<char>(cp);
                        else if (cp < 0x800) { out += static_cast<char>(0xC0 | (cp >> 6)); out += static_cast<char>(0x80 | (cp & 0x3F)); }
                        else { out += static_cast<char>(0xE0 | (cp >> 12)); out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F)); out += static_cast<char>(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *cur; break;   // \" \\ \/
                    }
                    ++cur;
                } else {
                    out += *cur++;
                }
            }
            if (cur >= end) fail("unterminated string");
            ++cur;  // closing quote
            return out;
        }
        Json array(int depth)
        {
            Json j; j.kind = Array;
            ++cur;
            skip_ws();
            if (cur < end && *cur == ']') { ++cur; return j; }
            for (;;) {
                skip_ws();
                j.arr.push_back(value(depth + 1));
                skip_ws();
                if (cur >= end) fail("unterminated array");
                if (*cur == ',') { ++cur; continue; }
                if (*cur == ']') { ++cur; break; }
                fail("expected , or ]");
            }
            return j;
        }
        Json object(int depth)
        {
            Json j; j.kind = Object;
            ++cur;
            skip_ws();
            if (cur < end && *cur == '}') { ++cur; return j; }
            for (;;) {
                skip_ws();
                if (cur >= end || *cur != '"') fail("expected key string");
                std::string key = string();
                skip_ws();
                if (cur >= end || *cur != ':') fail("expected :");
                ++cur;
                skip_ws();
                j.obj.emplace_back(std::move(key), value(depth + 1));
                skip_ws();
                if (cur >= end) fail("unterminated object");
                if (*cur == ',') { ++cur; continue; }
                if (*cur == '}') { ++cur; break; }
                fail("expected , or }");
            }
            return j;
        }
    };
};

}  // namespace aidax
