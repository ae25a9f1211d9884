// arp_json.h — host-side writer of the atom-atom part of the JSON output (no GPU involved).
//
// The reference's CLI ends with json.dump(i_complex.get_contacts(), fh, indent=4, sort_keys=True)
// (scripts/process_protein_cli.py:184-188; records built at core/interactions.py:172-212).  A whole-structure run has
// millions of atom-atom records, and building a Python dict per record costs seconds; this writes the same bytes from the
// result arrays directly: keys in sorted order, indent levels, float formatting (shortest round-trip repr, as
// float.__repr__) and string escaping (ensure_ascii) of the json module.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace arpjson {

inline void escape(std::string& o, const char* s) {   // json.dumps(str) with ensure_ascii=True
    o.push_back('"');
    const unsigned char* p = (const unsigned char*)s;
    char buf[16];
    while (*p) {
        unsigned c = *p;
        if (c == '"') { o += "\\\""; ++p; }
        else if (c == '\\') { o += "\\\\"; ++p; }
        else if (c == '\n') { o += "\\n"; ++p; }
        else if (c == '\r') { o += "\\r"; ++p; }
        else if (c == '\t') { o += "\\t"; ++p; }
        else if (c == '\b') { o += "\\b"; ++p; }
        else if (c == '\f') { o += "\\f"; ++p; }
        else if (c < 0x20) { snprintf(buf, sizeof buf, "\\u%04x", c); o += buf; ++p; }
        else if (c < 0x80) { o.push_back((char)c); ++p; }
        else {   // UTF-8 sequence -> \uXXXX (surrogate pair beyond the BMP)
            unsigned cp = 0;
            int extra = 0;
            if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; extra = 1; }
            else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; extra = 2; }
            else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; extra = 3; }
            else { cp = 0xFFFD; extra = 0; }
            ++p;
            for (int k = 0; k < extra && (*p & 0xC0) == 0x80; ++k, ++p) cp = (cp << 6) | (*p & 0x3F);
            if (cp >= 0x10000) {
                cp -= 0x10000;
                snprintf(buf, sizeof buf, "\\u%04x\\u%04x", 0xD800 + (cp >> 10), 0xDC00 + (cp & 0x3FF));
            } else {
                snprintf(buf, sizeof buf, "\\u%04x", cp);
            }
            o += buf;
        }
    }
    o.push_back('"');
}

inline void number(std::string& o, double v) {   // float.__repr__ for the magnitudes a distance can have
    if (std::isnan(v)) { o += "NaN"; return; }
    if (std::isinf(v)) { o += v > 0 ? "Infinity" : "-Infinity"; return; }
    char buf[40];
    auto r = std::to_chars(buf, buf + sizeof buf, v);   // shortest representation that round-trips
    std::string s(buf, r.ptr);
    // to_chars may pick scientific notation where repr() does not (and vice versa) only outside [1e-4, 1e16): format those
    // the way repr() does; inside, make sure there is a fractional part
    const double a = std::fabs(v);
    if (a != 0.0 && (a < 1e-4 || a >= 1e16)) {
        r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
        s.assign(buf, r.ptr);   // d.ddde-05 -> Python prints the exponent with at least two digits, as to_chars does
        const size_t e = s.find('e');
        if (e != std::string::npos && s.find('.') == std::string::npos) {}   // "1e-05" is what repr() prints too
    } else {
        if (s.find('e') != std::string::npos) {
            r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
            s.assign(buf, r.ptr);
        }
        if (s.find('.') == std::string::npos) s += ".0";
    }
    o += s;
}

inline void pad(std::string& o, int n) { o.append((size_t)n, ' '); }

}  // namespace arpjson

extern "C" int arp_write_contacts_json(const char* path, int indent, int append_mode, int64_t n, const int32_t* ci, const int32_t* cj,
                                       const double* dist_rounded, const uint16_t* sift, const uint8_t* ctype, int64_t n_atoms,
                                       const int32_t* atom_res, const char* const* atom_name, int64_t n_res,
                                       const char* const* res_name, const int32_t* res_seq, const char* const* res_chain,
                                       const char* const* res_icode, const char* const* res_comp_type,
                                       const char* const* sift_names, const char* const* ctype_names, const char* tail_records,
                                       int64_t n_tail) {
    // Writes "[" + the n atom-atom records + tail_records (already rendered records of the other bags, comma-separated at the
    // same indentation, n_tail of them) + "]" exactly as json.dump(list, indent=indent, sort_keys=True) would.
    (void)append_mode;
    if (!path || indent < 0 || n < 0 || n_atoms < 0 || n_res < 0) return -1;
    FILE* fh = fopen(path, "wb");
    if (!fh) return -2;
    using namespace arpjson;
    const int i1 = indent, i2 = 2 * indent, i3 = 3 * indent;
    // per-atom inner object text (keys sorted: auth_asym_id, auth_atom_id, auth_seq_id, label_comp_id, label_comp_type,
    // pdbx_PDB_ins_code), rendered lazily
    std::vector<std::string> atom_text((size_t)n_atoms);
    auto atom_obj = [&](int a) -> const std::string& {
        std::string& t = atom_text[(size_t)a];
        if (!t.empty()) return t;
        const int r = atom_res[a];
        t += "{\n";
        pad(t, i3); t += "\"auth_asym_id\": "; escape(t, res_chain[r]); t += ",\n";
        pad(t, i3); t += "\"auth_atom_id\": "; escape(t, atom_name[a]); t += ",\n";
        pad(t, i3); t += "\"auth_seq_id\": " + std::to_string(res_seq[r]) + ",\n";
        pad(t, i3); t += "\"label_comp_id\": "; escape(t, res_name[r]); t += ",\n";
        pad(t, i3); t += "\"label_comp_type\": "; escape(t, res_comp_type[r]); t += ",\n";
        pad(t, i3); t += "\"pdbx_PDB_ins_code\": "; escape(t, res_icode[r]); t += "\n";
        pad(t, i2); t += "}";
        return t;
    };
    std::unordered_map<unsigned, std::string> contact_text;
    auto contact_list = [&](unsigned s) -> const std::string& {
        auto it = contact_text.find(s);
        if (it != contact_text.end()) return it->second;
        std::string t;
        bool any = false;
        for (int k = 0; k < 15; ++k)
            if ((s >> k) & 1u) {
                t += any ? ",\n" : "[\n";
                pad(t, i3); escape(t, sift_names[k]);
                any = true;
            }
        if (any) { t += "\n"; pad(t, i2); t += "]"; }
        else t = "[]";
        return contact_text.emplace(s, std::move(t)).first->second;
    };
    std::string buf;
    buf.reserve(1 << 22);
    const int64_t total = n + n_tail;
    if (total == 0) { fputs("[]", fh); fclose(fh); return 0; }
    buf += "[\n";
    for (int64_t k = 0; k < n; ++k) {
        const int a = ci[k], b = cj[k];
        if (a < 0 || a >= n_atoms || b < 0 || b >= n_atoms || ctype[k] > 6) { fclose(fh); return -3; }
        pad(buf, i1); buf += "{\n";
        pad(buf, i2); buf += "\"bgn\": "; buf += atom_obj(a); buf += ",\n";
        pad(buf, i2); buf += "\"contact\": "; buf += contact_list(sift[k]); buf += ",\n";
        pad(buf, i2); buf += "\"distance\": "; number(buf, dist_rounded[k]); buf += ",\n";
        pad(buf, i2); buf += "\"end\": "; buf += atom_obj(b); buf += ",\n";
        pad(buf, i2); buf += "\"interacting_entities\": "; escape(buf, ctype_names[ctype[k]]); buf += ",\n";
        pad(buf, i2); buf += "\"type\": \"atom-atom\"\n";
        pad(buf, i1); buf += (k + 1 < total) ? "},\n" : "}\n";
        if (buf.size() > (1u << 22) - 4096) { fwrite(buf.data(), 1, buf.size(), fh); buf.clear(); }
    }
    if (n_tail > 0 && tail_records) { buf += tail_records; buf += "\n"; }
    buf += "]";
    fwrite(buf.data(), 1, buf.size(), fh);
    const int rc = ferror(fh) ? -4 : 0;
    fclose(fh);
    return rc;
}
