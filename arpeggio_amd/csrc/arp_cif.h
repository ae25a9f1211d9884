// arp_cif.h — host-side reader of one category of an mmCIF file (no GPU involved).
//
// The reference goes through gemmi (protein_reader.py:258-289, 415-441: gemmi.read_structure / gemmi.cif.read, then
// cif_block.get_mmcif_category('_atom_site.') — a dict of columns in which '?' is None, '.' is False and quoted values
// are unquoted).  gemmi is not part of the reference tree; this restates the CIF 1.1 syntax it implements:
//   * tokens are separated by white space; '#' at the start of a token begins a comment that runs to the end of the line;
//   * data_<name> opens a block, loop_ a table (tags, then values row by row), _tag value a single item;
//   * a value is a bare word, a '...' or "..." string (the closing quote counts only when white space or the end of the
//     file follows it), or a text field: a line that STARTS with ';' up to the next line that starts with ';';
//   * bare ? and . mean unknown / inapplicable.
// A category is every item whose tag starts with the prefix (case-insensitive, as CIF tags are); a pair item becomes a
// table of one row.  Cells are slices of the file's text (kept in the object), so opening a category copies nothing but
// the text itself.
#pragma once
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace arpcif {

enum Kind : uint8_t { VALUE = 0, UNKNOWN = 1, INAPPLICABLE = 2 };   // '?' and '.' when they are bare words
struct Cell { uint64_t begin; uint32_t len; uint8_t kind; };

struct Table {
    std::string text;
    std::vector<std::string> tags;      // item names after the category prefix, as spelt in the file
    std::vector<Cell> cells;            // row-major
    int64_t rows = 0;
    int n_blocks = 0;
    std::string error;
    int64_t ncols() const { return (int64_t)tags.size(); }
};

inline bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

struct Token { enum T { END, DATA, LOOP, SAVE, TAG, VAL } t; uint64_t begin; uint32_t len; uint8_t kind; };

class Lexer {
  public:
    Lexer(const char* s, uint64_t n) : s_(s), n_(n) {}
    bool fail = false;
    std::string why;
    Token next() {
        for (;;) {   // white space and comments
            while (p_ < n_ && is_ws(s_[p_])) { bol_ = (s_[p_] == '\n'); ++p_; }
            if (p_ < n_ && s_[p_] == '#') { while (p_ < n_ && s_[p_] != '\n') ++p_; continue; }
            break;
        }
        if (p_ >= n_) return Token{Token::END, p_, 0, 0};
        const char c = s_[p_];
        if (c == ';' && (bol_ || p_ == 0)) {   // text field
            const uint64_t b = p_ + 1;
            uint64_t q = b;
            for (;;) {
                while (q < n_ && s_[q] != '\n') ++q;
                if (q >= n_) { fail = true; why = "unterminated text field"; return Token{Token::END, p_, 0, 0}; }
                if (q + 1 < n_ && s_[q + 1] == ';') break;
                ++q;
            }
            uint64_t e = q;                      // the line break before the closing ';' is not part of the value
            if (e > b && s_[e - 1] == '\r') --e;
            p_ = q + 2;
            bol_ = false;
            return Token{Token::VAL, b, (uint32_t)(e - b), VALUE};
        }
        bol_ = false;
        if (c == '\'' || c == '"') {
            const uint64_t b = p_ + 1;
            uint64_t q = b;
            for (;;) {
                while (q < n_ && s_[q] != c && s_[q] != '\n') ++q;
                if (q >= n_ || s_[q] == '\n') { fail = true; why = "unterminated quoted string"; return Token{Token::END, p_, 0, 0}; }
                if (q + 1 >= n_ || is_ws(s_[q + 1])) break;
                ++q;
            }
            p_ = q + 1;
            return Token{Token::VAL, b, (uint32_t)(q - b), VALUE};
        }
        const uint64_t b = p_;
        while (p_ < n_ && !is_ws(s_[p_])) ++p_;
        const uint32_t len = (uint32_t)(p_ - b);
        if (c == '_') return Token{Token::TAG, b, len, 0};
        if (len >= 5 && s_[b + 4] == '_') {   // the reserved words end in '_' at the fifth character
            if (ieq(b, "data_", 5)) return Token{Token::DATA, b, len, 0};
            if (len == 5 && ieq(b, "loop_", 5)) return Token{Token::LOOP, b, len, 0};
            if (ieq(b, "save_", 5)) return Token{Token::SAVE, b, len, 0};
        }
        uint8_t kind = VALUE;
        if (len == 1 && c == '?') kind = UNKNOWN;
        else if (len == 1 && c == '.') kind = INAPPLICABLE;
        return Token{Token::VAL, b, len, kind};
    }

  private:
    bool ieq(uint64_t b, const char* w, size_t k) const {
        for (size_t i = 0; i < k; ++i)
            if (tolower((unsigned char)s_[b + i]) != w[i]) return false;
        return true;
    }
    const char* s_;
    uint64_t n_, p_ = 0;
    bool bol_ = true;
};

inline bool has_prefix(const std::string& text, const Token& t, const std::string& prefix_lower) {
    if (t.len < prefix_lower.size()) return false;
    for (size_t i = 0; i < prefix_lower.size(); ++i)
        if (tolower((unsigned char)text[t.begin + i]) != prefix_lower[i]) return false;
    return true;
}

// Every item of `category` in the FIRST data block (n_blocks tells how many there were: gemmi's sole_block() refuses
// files with more than one).  Returns false and sets error on malformed input.
inline bool read_category(Table& T, const char* s, uint64_t n, const char* category) {
    T.text.assign(s, n);
    std::string pre(category);
    for (auto& ch : pre) ch = (char)tolower((unsigned char)ch);
    Lexer lx(T.text.data(), T.text.size());
    Token t = lx.next();
    bool in_first = false;
    std::vector<Cell> single_cells;           // pair items of the category (one row)
    std::vector<std::string> single_tags;
    bool loop_found = false;
    while (t.t != Token::END) {
        if (t.t == Token::DATA) {
            ++T.n_blocks;
            in_first = T.n_blocks == 1;
            t = lx.next();
        } else if (t.t == Token::SAVE) {
            t = lx.next();                    // save frames (dictionaries): their items are read like any other
        } else if (t.t == Token::LOOP) {
            std::vector<Token> tags;
            t = lx.next();
            while (t.t == Token::TAG) { tags.push_back(t); t = lx.next(); }
            if (tags.empty()) { T.error = "loop_ without tags"; return false; }
            const bool mine = in_first && has_prefix(T.text, tags[0], pre);
            if (mine && (loop_found || !single_tags.empty())) { T.error = "category occurs more than once"; return false; }
            size_t count = 0;
            std::vector<Cell> cells;
            if (mine) cells.reserve((size_t)(n / 6 + 16));
            while (t.t == Token::VAL) {
                if (mine) cells.push_back(Cell{t.begin, t.len, t.kind});
                ++count;
                t = lx.next();
            }
            if (count % tags.size() != 0) { T.error = "loop_ with an incomplete last row"; return false; }
            if (mine) {
                loop_found = true;
                for (const Token& g : tags) {
                    if (!has_prefix(T.text, g, pre)) { T.error = "loop_ mixes categories"; return false; }
                    T.tags.emplace_back(T.text.substr(g.begin + pre.size(), g.len - pre.size()));
                }
                T.cells.swap(cells);
                T.rows = (int64_t)(count / tags.size());
            }
        } else if (t.t == Token::TAG) {
            const Token tag = t;
            t = lx.next();
            if (t.t != Token::VAL) { T.error = "tag without a value"; return false; }
            if (in_first && has_prefix(T.text, tag, pre)) {
                if (loop_found) { T.error = "category occurs more than once"; return false; }
                single_tags.emplace_back(T.text.substr(tag.begin + pre.size(), tag.len - pre.size()));
                single_cells.push_back(Cell{t.begin, t.len, t.kind});
            }
            t = lx.next();
        } else {   // a value where none belongs
            T.error = "value outside a loop_ or a tag";
            return false;
        }
        if (lx.fail) break;
    }
    if (lx.fail) { T.error = lx.why; return false; }
    if (T.n_blocks == 0) { T.error = "no data_ block"; return false; }
    if (!loop_found && !single_tags.empty()) {
        T.tags.swap(single_tags);
        T.cells.swap(single_cells);
        T.rows = 1;
    }
    return true;
}

}  // namespace arpcif
