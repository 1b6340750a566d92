// toml_lite.h — the subset of TOML 1.0 the reference's config / scene / library files use (they are read with the `toml`
// crate + serde, src/parsing/mod.rs:110-143): comments, bare / quoted / dotted keys, [tables], [[arrays of tables]],
// basic and literal strings (single- and multi-line), integers (with _ and 0x/0o/0b), floats (inf/nan, exponents),
// booleans, arrays (nested, multi-line, trailing comma), inline tables.  Dates are not supported (the reference's
// structs have no date fields).  Tables keep insertion order.
#ifndef PT_TOML_LITE_H
#define PT_TOML_LITE_H
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace toml {

struct Value;
using Array = std::vector<Value>;
struct Table {
    std::vector<std::pair<std::string, Value>> items;
    const Value* find(const std::string& key) const;
    Value* find(const std::string& key);
    Value& insert(const std::string& key, Value v);
};

struct Value {
    enum Kind { String, Integer, Float, Boolean, ArrayKind, TableKind } kind = String;
    std::string s;
    int64_t i = 0;
    double f = 0.0;
    bool b = false;
    std::shared_ptr<Array> a;
    std::shared_ptr<Table> t;
    bool defined_inline = false;   // inline tables and static arrays may not be extended by later headers
    bool array_of_tables = false;
    int line = 0;

    static Value string(std::string v) { Value r; r.kind = String; r.s = std::move(v); return r; }
    static Value integer(int64_t v) { Value r; r.kind = Integer; r.i = v; return r; }
    static Value floating(double v) { Value r; r.kind = Float; r.f = v; return r; }
    static Value boolean(bool v) { Value r; r.kind = Boolean; r.b = v; return r; }
    static Value array() { Value r; r.kind = ArrayKind; r.a = std::make_shared<Array>(); return r; }
    static Value table() { Value r; r.kind = TableKind; r.t = std::make_shared<Table>(); return r; }
    bool is_number() const { return kind == Integer || kind == Float; }
    double number() const { return kind == Integer ? (double)i : f; }
};

inline const Value* Table::find(const std::string& key) const {
    for (auto& kv : items) if (kv.first == key) return &kv.second;
    return nullptr;
}
inline Value* Table::find(const std::string& key) {
    for (auto& kv : items) if (kv.first == key) return &kv.second;
    return nullptr;
}
inline Value& Table::insert(const std::string& key, Value v) { items.emplace_back(key, std::move(v)); return items.back().second; }

struct ParseError : std::runtime_error {
    explicit ParseError(const std::string& m) : std::runtime_error(m) {}
};

class Parser {
public:
    explicit Parser(const std::string& text) : s_(text) {}
    Value parse() {
        Value root = Value::table();
        Table* current = root.t.get();
        for (;;) {
            skip_ws_comments_newlines();
            if (eof()) break;
            if (peek() == '[') {
                bool aot = s_.compare(p_, 2, "[[") == 0;
                p_ += aot ? 2 : 1;
                skip_ws();
                std::vector<std::string> path = key_path();
                skip_ws();
                expect(']');
                if (aot) expect(']');
                end_of_line();
                current = aot ? open_array_table(root.t.get(), path) : open_table(root.t.get(), path);
            } else {
                std::vector<std::string> path = key_path();
                skip_ws();
                expect('=');
                skip_ws();
                Value v = value();
                end_of_line();
                Table* t = current;
                for (size_t k = 0; k + 1 < path.size(); ++k) t = descend(t, path[k], /*dotted=*/true);
                if (t->find(path.back())) fail("duplicate key '" + path.back() + "'");
                t->insert(path.back(), std::move(v));
            }
        }
        return root;
    }

private:
    const std::string& s_;
    size_t p_ = 0;
    int line_ = 1;
    std::vector<Table*> defined_;  // tables opened by a [header]

    bool eof() const { return p_ >= s_.size(); }
    char peek(size_t k = 0) const { return p_ + k < s_.size() ? s_[p_ + k] : '\0'; }
    [[noreturn]] void fail(const std::string& m) const { throw ParseError("TOML line " + std::to_string(line_) + ": " + m); }
    void expect(char c) { if (peek() != c) fail(std::string("expected '") + c + "'"); ++p_; }
    void skip_ws() { while (peek() == ' ' || peek() == '\t') ++p_; }
    void skip_comment() { if (peek() == '#') while (!eof() && peek() != '\n') ++p_; }
    bool newline() {
        if (peek() == '\n') { ++p_; ++line_; return true; }
        if (peek() == '\r' && peek(1) == '\n') { p_ += 2; ++line_; return true; }
        return false;
    }
    void skip_ws_comments_newlines() { for (;;) { skip_ws(); skip_comment(); if (!newline()) break; } }
    void end_of_line() { skip_ws(); skip_comment(); if (!eof() && !newline()) fail("unexpected characters after value"); }

    static bool bare(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-'; }
    std::vector<std::string> key_path() {
        std::vector<std::string> path;
        for (;;) {
            skip_ws();
            if (peek() == '"') path.push_back(basic_string());
            else if (peek() == '\'') path.push_back(literal_string());
            else {
                size_t b = p_;
                while (bare(peek())) ++p_;
                if (p_ == b) fail("expected a key");
                path.push_back(s_.substr(b, p_ - b));
            }
            skip_ws();
            if (peek() == '.') { ++p_; continue; }
            return path;
        }
    }
    Table* descend(Table* t, const std::string& key, bool dotted) {
        Value* v = t->find(key);
        if (!v) return t->insert(key, Value::table()).t.get();
        if (v->kind == Value::TableKind) { if (v->defined_inline) fail("cannot extend inline table '" + key + "'"); return v->t.get(); }
        if (v->kind == Value::ArrayKind && v->array_of_tables && !dotted) return v->a->back().t.get();
        fail("key '" + key + "' is not a table");
    }
    Table* open_table(Table* root, const std::vector<std::string>& path) {
        Table* t = root;
        for (auto& k : path) t = descend(t, k, false);
        for (Table* d : defined_) if (d == t) fail("table '" + path.back() + "' defined twice");
        defined_.push_back(t);
        return t;
    }
    Table* open_array_table(Table* root, const std::vector<std::string>& path) {
        Table* t = root;
        for (size_t k = 0; k + 1 < path.size(); ++k) t = descend(t, path[k], false);
        Value* v = t->find(path.back());
        if (!v) { Value arr = Value::array(); arr.array_of_tables = true; v = &t->insert(path.back(), std::move(arr)); }
        if (v->kind != Value::ArrayKind || !v->array_of_tables) fail("'" + path.back() + "' is not an array of tables");
        v->a->push_back(Value::table());
        return v->a->back().t.get();
    }

    void append_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    std::string basic_string() {
        bool multi = s_.compare(p_, 3, "\"\"\"") == 0;
        p_ += multi ? 3 : 1;
        if (multi) newline();
        std::string out;
        for (;;) {
            if (eof()) fail("unterminated string");
            char c = peek();
            if (multi && s_.compare(p_, 3, "\"\"\"") == 0) { p_ += 3; while (peek() == '"') { out += '"'; ++p_; } return out; }
            if (!multi && c == '"') { ++p_; return out; }
            if (c == '\n' || c == '\r') { if (!multi) fail("newline in string"); if (newline()) out += '\n'; else ++p_; continue; }
            if (c == '\\') {
                ++p_;
                char e = peek(); ++p_;
                switch (e) {
                    case 'b': out += '\b'; break; case 't': out += '\t'; break; case 'n': out += '\n'; break; case 'f': out += '\f'; break;
                    case 'r': out += '\r'; break; case '"': out += '"'; break; case '\\': out += '\\'; break;
                    case 'u': case 'U': {
                        int n = e == 'u' ? 4 : 8; uint32_t cp = 0;
                        for (int k = 0; k < n; ++k) { char h = peek(); ++p_; cp = cp * 16 + (uint32_t)(h <= '9' ? h - '0' : (h | 32) - 'a' + 10); }
                        append_utf8(out, cp); break;
                    }
                    case ' ': case '\t': case '\n': case '\r':
                        if (!multi) fail("bad escape");
                        --p_;
                        for (;;) { skip_ws(); if (!newline()) break; }
                        break;
                    default: fail("bad escape");
                }
                continue;
            }
            out += c; ++p_;
        }
    }
    std::string literal_string() {
        bool multi = s_.compare(p_, 3, "'''") == 0;
        p_ += multi ? 3 : 1;
        if (multi) newline();
        std::string out;
        for (;;) {
            if (eof()) fail("unterminated string");
            if (multi && s_.compare(p_, 3, "'''") == 0) { p_ += 3; return out; }
            if (!multi && peek() == '\'') { ++p_; return out; }
            if (peek() == '\n') { if (!multi) fail("newline in string"); ++line_; }
            out += peek(); ++p_;
        }
    }
    Value value() {
        Value v;
        int line = line_;
        char c = peek();
        if (c == '"') v = Value::string(basic_string());
        else if (c == '\'') v = Value::string(literal_string());
        else if (c == '[') v = array();
        else if (c == '{') v = inline_table();
        else if (s_.compare(p_, 4, "true") == 0 && !bare(peek(4))) { p_ += 4; v = Value::boolean(true); }
        else if (s_.compare(p_, 5, "false") == 0 && !bare(peek(5))) { p_ += 5; v = Value::boolean(false); }
        else v = number();
        v.line = line;
        return v;
    }
    Value array() {
        expect('[');
        Value v = Value::array();
        v.defined_inline = true;
        for (;;) {
            skip_ws_comments_newlines();
            if (peek() == ']') { ++p_; return v; }
            v.a->push_back(value());
            skip_ws_comments_newlines();
            if (peek() == ',') { ++p_; continue; }
            if (peek() == ']') { ++p_; return v; }
            fail("expected ',' or ']' in array");
        }
    }
    Value inline_table() {
        expect('{');
        Value v = Value::table();
        v.defined_inline = true;
        skip_ws();
        if (peek() == '}') { ++p_; return v; }
        for (;;) {
            skip_ws();
            std::vector<std::string> path = key_path();
            skip_ws(); expect('='); skip_ws();
            Value item = value();
            Table* t = v.t.get();
            for (size_t k = 0; k + 1 < path.size(); ++k) {
                Value* sub = t->find(path[k]);
                if (!sub) sub = &t->insert(path[k], Value::table());
                if (sub->kind != Value::TableKind) fail("key is not a table");
                t = sub->t.get();
            }
            if (t->find(path.back())) fail("duplicate key '" + path.back() + "'");
            t->insert(path.back(), std::move(item));
            skip_ws();
            if (peek() == ',') { ++p_; continue; }
            if (peek() == '}') { ++p_; return v; }
            fail("expected ',' or '}' in inline table");
        }
    }
    Value number() {
        size_t b = p_;
        while (!eof() && (bare(peek()) || peek() == '+' || peek() == '.' || peek() == ':')) ++p_;
        std::string tok = s_.substr(b, p_ - b);
        if (tok.empty()) fail("expected a value");
        std::string clean;
        for (size_t k = 0; k < tok.size(); ++k) {
            if (tok[k] == '_') { if (k == 0 || k + 1 == tok.size() || !isdigit_any(tok[k - 1]) || !isdigit_any(tok[k + 1])) fail("misplaced '_' in number"); continue; }
            clean += tok[k];
        }
        std::string body = clean; int sign = 1;
        if (body[0] == '+' || body[0] == '-') { sign = body[0] == '-' ? -1 : 1; body = body.substr(1); }
        if (body == "inf") return Value::floating(sign * __builtin_inf());
        if (body == "nan") return Value::floating(__builtin_nan(""));
        if (body.size() > 2 && body[0] == '0' && (body[1] == 'x' || body[1] == 'o' || body[1] == 'b')) {
            int base = body[1] == 'x' ? 16 : body[1] == 'o' ? 8 : 2;
            char* end = nullptr;
            long long v = std::strtoll(body.c_str() + 2, &end, base);
            if (*end) fail("bad integer '" + tok + "'");
            return Value::integer(sign * v);
        }
        bool is_float = body.find_first_of(".eE") != std::string::npos;
        for (char ch : body) if (!(ch >= '0' && ch <= '9') && ch != '.' && ch != 'e' && ch != 'E' && ch != '+' && ch != '-') fail("unsupported value '" + tok + "'");
        char* end = nullptr;
        if (is_float) {
            double v = std::strtod(clean.c_str(), &end);
            if (*end) fail("bad float '" + tok + "'");
            return Value::floating(v);
        }
        long long v = std::strtoll(clean.c_str(), &end, 10);
        if (*end) fail("bad integer '" + tok + "'");
        return Value::integer(v);
    }
    static bool isdigit_any(char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace toml
#endif
