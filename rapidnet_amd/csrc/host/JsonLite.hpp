// JsonLite.hpp -- minimal ordered JSON reader for the reference's input schema: one top-level object whose members are
// arrays of numbers, numbers or strings (the reference parses it with the vendored rapidjson, src/rapidjson/).
// Member ORDER is preserved because Forecaster addresses members by position (Forecaster.cu:94,108).
// Numbers are parsed as double (the reference truncates to float with GetFloat()).
#ifndef RAPIDNET_JSONLITE_HPP_
#define RAPIDNET_JSONLITE_HPP_

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace jsonlite {

struct Value {
    enum Kind { ARRAY, NUMBER, STRING } kind = ARRAY;
    std::vector<double> arr;
    std::string str;
    bool IsArray() const { return kind == ARRAY; }
    bool IsString() const { return kind == STRING; }
    size_t Size() const { return arr.size(); }
    double operator[](size_t i) const { return arr.at(i); }
};

class Document {
public:
    explicit Document(const std::string &path) {
        std::ifstream f(path.c_str());
        if (!f.good()) throw std::runtime_error("cannot open " + path);
        std::stringstream ss;
        ss << f.rdbuf();
        text_ = ss.str();
        parse();
    }
    bool HasMember(const std::string &k) const { return find(k) >= 0; }
    const Value &operator[](const std::string &k) const {
        int i = find(k);
        if (i < 0) throw std::runtime_error("missing JSON member \"" + k + "\"");
        return members_[i].second;
    }
    size_t MemberCount() const { return members_.size(); }
    const std::string &MemberName(size_t i) const { return members_.at(i).first; }
    const Value &MemberValue(size_t i) const { return members_.at(i).second; }

private:
    std::string text_;
    size_t pos_ = 0;
    std::vector<std::pair<std::string, Value> > members_;

    int find(const std::string &k) const {
        for (size_t i = 0; i < members_.size(); i++) if (members_[i].first == k) return (int)i;
        return -1;
    }
    void ws() { while (pos_ < text_.size() && std::isspace((unsigned char)text_[pos_])) pos_++; }
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("JSON parse error: ") + what + " at offset " + std::to_string(pos_)); }
    void expect(char c) { ws(); if (pos_ >= text_.size() || text_[pos_] != c) fail("unexpected character"); pos_++; }
    std::string parseString() {
        expect('"');
        std::string s;
        while (pos_ < text_.size() && text_[pos_] != '"') {
            if (text_[pos_] == '\\' && pos_ + 1 < text_.size()) { pos_++; char c = text_[pos_]; s += (c == 'n' ? '\n' : c == 't' ? '\t' : c); }
            else s += text_[pos_];
            pos_++;
        }
        if (pos_ >= text_.size()) fail("unterminated string");
        pos_++;
        return s;
    }
    double parseNumber() {
        ws();
        const char *b = text_.c_str() + pos_;
        char *e = nullptr;
        double v = std::strtod(b, &e);
        if (e == b) fail("number expected");
        pos_ += (size_t)(e - b);
        return v;
    }
    void parse() {
        expect('{');
        ws();
        if (pos_ < text_.size() && text_[pos_] == '}') return;
        for (;;) {
            ws();
            std::string key = parseString();
            expect(':');
            ws();
            Value v;
            if (pos_ >= text_.size()) fail("value expected");
            if (text_[pos_] == '[') {
                pos_++;
                v.kind = Value::ARRAY;
                ws();
                if (text_[pos_] == ']') pos_++;
                else for (;;) {
                    v.arr.push_back(parseNumber());
                    ws();
                    if (text_[pos_] == ',') { pos_++; continue; }
                    if (text_[pos_] == ']') { pos_++; break; }
                    fail("',' or ']' expected");
                }
            } else if (text_[pos_] == '"') { v.kind = Value::STRING; v.str = parseString(); }
            else { v.kind = Value::NUMBER; v.arr.push_back(parseNumber()); }
            members_.push_back(std::make_pair(key, v));
            ws();
            if (pos_ < text_.size() && text_[pos_] == ',') { pos_++; continue; }
            if (pos_ < text_.size() && text_[pos_] == '}') break;
            fail("',' or '}' expected");
        }
    }
};

}  // namespace jsonlite
#endif
