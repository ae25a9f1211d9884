// arp_cif_api.h — the C entry points of the mmCIF category reader (include/arpeggio_hip.h: arp_cif_*), host code only.
// Included by arp_api.hip (libarpeggio_hip.so) and by arp_host.cpp (libarpeggio_host.so: the same entry points built with g++,
// for machines without hipcc — the fixture generators under tests/golden/ and the reader of a CPU-only front end).  Include it
// after "arp_cif.h" and the public header, whose prototypes give the functions their C linkage.
#pragma once
#include <cstdio>
#include <new>
#include <string>

struct arp_cif { arpcif::Table t; };

int arp_cif_open(const char* text, uint64_t len, const char* category, arp_cif** out, char* err, uint64_t err_cap) {
    auto fail = [&](const std::string& m) {
        if (err && err_cap) { snprintf(err, (size_t)err_cap, "%s", m.c_str()); }
        return ARP_E_ARG;
    };
    if (!text || !category || !out || category[0] != '_') return fail("arp_cif_open: bad argument");
    arp_cif* c = new (std::nothrow) arp_cif();
    if (!c) return ARP_E_NOMEM;
    if (!arpcif::read_category(c->t, text, len, category)) {
        const std::string m = "arp_cif_open: " + c->t.error;
        delete c;
        return fail(m);
    }
    *out = c;
    return ARP_OK;
}
void arp_cif_close(arp_cif* t) { delete t; }
int64_t arp_cif_rows(const arp_cif* t) { return t ? t->t.rows : 0; }
int arp_cif_cols(const arp_cif* t) { return t ? (int)t->t.ncols() : 0; }
int arp_cif_blocks(const arp_cif* t) { return t ? t->t.n_blocks : 0; }
const char* arp_cif_tag(const arp_cif* t, int col) { return (t && col >= 0 && col < (int)t->t.ncols()) ? t->t.tags[(size_t)col].c_str() : nullptr; }
const char* arp_cif_text(const arp_cif* t) { return t ? t->t.text.c_str() : nullptr; }
int arp_cif_column(const arp_cif* t, int col, uint64_t* begin, uint32_t* len, uint8_t* kind) {
    if (!t || col < 0 || col >= (int)t->t.ncols() || !begin || !len || !kind) return ARP_E_ARG;
    const int64_t nc = t->t.ncols();
    for (int64_t r = 0; r < t->t.rows; ++r) {
        const arpcif::Cell& c = t->t.cells[(size_t)(r * nc + col)];
        begin[r] = c.begin; len[r] = c.len; kind[r] = c.kind;
    }
    return ARP_OK;
}
namespace {
template <class T, class F>
int cif_numbers(const arp_cif* t, int col, T missing, T* out, int64_t* bad_row, F parse) {
    if (!t || col < 0 || col >= (int)t->t.ncols() || !out) return ARP_E_ARG;
    const int64_t nc = t->t.ncols();
    std::string tmp;
    for (int64_t r = 0; r < t->t.rows; ++r) {
        const arpcif::Cell& c = t->t.cells[(size_t)(r * nc + col)];
        if (c.kind != arpcif::VALUE) { out[r] = missing; continue; }
        tmp.assign(t->t.text, (size_t)c.begin, (size_t)c.len);
        char* end = nullptr;
        out[r] = parse(tmp.c_str(), &end);
        if (tmp.empty() || end != tmp.c_str() + tmp.size()) {
            if (bad_row) *bad_row = r;
            return -1;
        }
    }
    return ARP_OK;
}
}  // namespace
int arp_cif_column_f64(const arp_cif* t, int col, double missing, double* out, int64_t* bad_row) {
    return cif_numbers<double>(t, col, missing, out, bad_row, [](const char* s, char** e) { return strtod(s, e); });
}
int arp_cif_column_i64(const arp_cif* t, int col, int64_t missing, int64_t* out, int64_t* bad_row) {
    return cif_numbers<int64_t>(t, col, missing, out, bad_row, [](const char* s, char** e) { return (int64_t)strtoll(s, e, 10); });
}
