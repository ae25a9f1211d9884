// arp_json.h — host-side writer of the atom-atom part of the JSON output (no GPU involved).
//
// The reference's CLI ends with json.dump(i_complex.get_contacts(), fh, indent=4, sort_keys=True)
// (scripts/process_protein_cli.py:184-188; records built at core/interactions.py:172-212).  A whole-structure run has
// millions of atom-atom records, and building a Python dict per record costs seconds; this writes the same bytes from the
// result arrays directly: keys in sorted order, indent levels, float formatting (shortest round-trip repr, as
// float.__repr__) and string escaping (ensure_ascii) of the json module.
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <thread>
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
    //
    // A whole-structure run of 100 k atoms is 1.25 M records = 814 MB of text: rendered and written by one thread that took
    // 0.41 s, three orders of magnitude above the GPU pass that produced the records.  The records are independent, so the
    // file is produced by ARP_EXPORT_THREADS host threads (default: the CPUs this process may run on, at most 32): the
    // per-atom and per-fingerprint text fragments are rendered once, pass 1 sizes every block of 8192 records (each record's
    // length is the sum of its fragments' lengths), a prefix sum gives every block its file offset, pass 2 renders the blocks
    // and writes them with pwrite() at their offsets.
    // flags (the parameter once called append_mode; 0 keeps the old meaning): bit 0 = `dist_rounded` holds the distances as
    // computed (float32 values widened to double) and they are rounded here the way np.round(x, 2) rounds — rint(x * 100) / 100 —
    const bool round_here = (append_mode & 1) != 0;
    if (!path || indent < 0 || n < 0 || n_atoms < 0 || n_res < 0) return -1;
    using namespace arpjson;
    auto dist_of = [&](int64_t k) -> double { return round_here ? std::nearbyint(dist_rounded[k] * 100.0) / 100.0 : dist_rounded[k]; };
    const int i1 = indent, i2 = 2 * indent, i3 = 3 * indent;
    const int64_t total = n + n_tail;
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return -2;
    auto write_all = [&](const char* p, size_t len, off_t off) -> bool {
        while (len > 0) {
            const ssize_t k = pwrite(fd, p, len, off);
            if (k <= 0) return false;
            p += k; len -= (size_t)k; off += k;
        }
        return true;
    };
    if (total == 0) { const bool ok = write_all("[]", 2, 0); close(fd); return ok ? 0 : -4; }
    int nthreads = 0;
    if (const char* e = getenv("ARP_EXPORT_THREADS")) nthreads = atoi(e);
    if (nthreads <= 0) {
        cpu_set_t set;
        nthreads = (sched_getaffinity(0, sizeof set, &set) == 0) ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        if (FILE* q = fopen("/sys/fs/cgroup/cpu.max", "r")) {      // a container's CPU quota counts, not the cores the host shows
            long long quota = 0, period = 0;
            if (fscanf(q, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                nthreads = std::min<long long>(nthreads, std::max<long long>(1, quota / period));
            fclose(q);
        }
        nthreads = std::min(nthreads, 32);
    }
    nthreads = std::max(1, nthreads);
    auto parallel = [&](int64_t items, auto&& body) {   // body(first, last, thread) over [0, items) in contiguous slices
        const int T = (int)std::min<int64_t>(nthreads, std::max<int64_t>(items, 1));
        if (T <= 1) { body((int64_t)0, items, 0); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t]() { body(items * t / T, items * (t + 1) / T, t); });
        for (auto& x : th) x.join();
    };
    // ---- fragments: the inner object of every atom that occurs (keys sorted: auth_asym_id, auth_atom_id, auth_seq_id,
    // label_comp_id, label_comp_type, pdbx_PDB_ins_code) and the "contact" list of every fingerprint that occurs
    std::vector<uint8_t> atom_used((size_t)n_atoms, 0);
    std::vector<uint8_t> sift_used(65536, 0);
    std::atomic<int> bad{0};
    parallel(n, [&](int64_t k0, int64_t k1, int) {
        for (int64_t k = k0; k < k1; ++k) {
            const int a = ci[k], b = cj[k];
            if (a < 0 || a >= n_atoms || b < 0 || b >= n_atoms || ctype[k] > 6) { bad = 1; return; }
            atom_used[(size_t)a] = 1; atom_used[(size_t)b] = 1; sift_used[sift[k]] = 1;     // (racing stores of the same value)
        }
    });
    if (bad) { close(fd); return -3; }
    std::vector<std::string> atom_text((size_t)n_atoms);
    parallel(n_atoms, [&](int64_t a0, int64_t a1, int) {
        for (int64_t a = a0; a < a1; ++a) {
            if (!atom_used[(size_t)a]) continue;
            std::string& t = atom_text[(size_t)a];
            const int r = atom_res[a];
            t += "{\n";
            pad(t, i3); t += "\"auth_asym_id\": "; escape(t, res_chain[r]); t += ",\n";
            pad(t, i3); t += "\"auth_atom_id\": "; escape(t, atom_name[a]); t += ",\n";
            pad(t, i3); t += "\"auth_seq_id\": " + std::to_string(res_seq[r]) + ",\n";
            pad(t, i3); t += "\"label_comp_id\": "; escape(t, res_name[r]); t += ",\n";
            pad(t, i3); t += "\"label_comp_type\": "; escape(t, res_comp_type[r]); t += ",\n";
            pad(t, i3); t += "\"pdbx_PDB_ins_code\": "; escape(t, res_icode[r]); t += "\n";
            pad(t, i2); t += "}";
        }
    });
    std::vector<std::string> contact_text(65536);
    for (unsigned s_ = 0; s_ < 65536; ++s_) {
        if (!sift_used[s_]) continue;
        std::string& t = contact_text[s_];
        bool any = false;
        for (int k = 0; k < 15; ++k)
            if ((s_ >> k) & 1u) {
                t += any ? ",\n" : "[\n";
                pad(t, i3); escape(t, sift_names[k]);
                any = true;
            }
        if (any) { t += "\n"; pad(t, i2); t += "]"; }
        else t = "[]";
    }
    std::string ctype_text[7];
    for (int k = 0; k < 7; ++k) escape(ctype_text[k], ctype_names[k]);
    // fixed text of a record around its five fragments
    std::string f0, f1, f2, f3, f4, f5;
    pad(f0, i1); f0 += "{\n"; pad(f0, i2); f0 += "\"bgn\": ";
    f1 = ",\n"; pad(f1, i2); f1 += "\"contact\": ";
    f2 = ",\n"; pad(f2, i2); f2 += "\"distance\": ";
    f3 = ",\n"; pad(f3, i2); f3 += "\"end\": ";
    f4 = ",\n"; pad(f4, i2); f4 += "\"interacting_entities\": ";
    f5 = ",\n"; pad(f5, i2); f5 += "\"type\": \"atom-atom\"\n"; pad(f5, i1); f5 += "}";
    const size_t fixed = f0.size() + f1.size() + f2.size() + f3.size() + f4.size() + f5.size();
    auto number_len = [](double v) -> size_t { std::string t; number(t, v); return t.size(); };
    // ---- pass 1: bytes of every block of records (a record ends with ",\n", the last one of the file with "\n")
    constexpr int64_t BLOCK = 8192;
    const int64_t nblocks = (n + BLOCK - 1) / BLOCK;
    std::vector<uint64_t> off((size_t)nblocks + 1, 0);
    parallel(nblocks, [&](int64_t b0, int64_t b1, int) {
        for (int64_t b = b0; b < b1; ++b) {
            uint64_t bytes = 0;
            for (int64_t k = b * BLOCK, k1 = std::min(n, (b + 1) * BLOCK); k < k1; ++k)
                bytes += fixed + atom_text[(size_t)ci[k]].size() + atom_text[(size_t)cj[k]].size() + contact_text[sift[k]].size() +
                         number_len(dist_of(k)) + ctype_text[ctype[k]].size() + ((k + 1 < total) ? 2 : 1);
            off[(size_t)b + 1] = bytes;
        }
    });
    off[0] = 2;   // "[\n"
    for (int64_t b = 0; b < nblocks; ++b) off[(size_t)b + 1] += off[(size_t)b];
    // ---- pass 2: render in parallel, write each block at its offset.  Buffered pwrite()s to ONE file take the inode lock one
    // after the other (the copies into the page cache serialise at ~2-3 GB/s; the rendering overlaps them);
    // ARP_EXPORT_MMAP=1 writes through a shared mapping instead (no inode lock, one page fault per 4 KiB: faster on tmpfs,
    // slower on the overlay / ext4 file systems of the test boxes).
    std::string tail;
    if (n_tail > 0 && tail_records) { tail += tail_records; tail += "\n"; }
    tail += "]";
    const uint64_t file_bytes = off[(size_t)nblocks] + tail.size();
    char* map = nullptr;
    const char* use_map = getenv("ARP_EXPORT_MMAP");
    if (use_map && atoi(use_map) > 0 && ftruncate(fd, (off_t)file_bytes) == 0) {
        // (O_WRONLY files cannot be mapped shared: reopen read-write for the mapping)
        const int fd2 = open(path, O_RDWR);
        if (fd2 >= 0) {
            void* m = mmap(nullptr, (size_t)file_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd2, 0);
            if (m != MAP_FAILED) map = (char*)m;
            close(fd2);
        }
    }
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    parallel((int64_t)nthreads, [&](int64_t, int64_t, int) {
        std::string buf;
        for (;;) {
            const int64_t b = next.fetch_add(1);
            if (b >= nblocks || failed) break;
            buf.clear();
            buf.reserve((size_t)(off[(size_t)b + 1] - off[(size_t)b]));
            for (int64_t k = b * BLOCK, k1 = std::min(n, (b + 1) * BLOCK); k < k1; ++k) {
                buf += f0; buf += atom_text[(size_t)ci[k]];
                buf += f1; buf += contact_text[sift[k]];
                buf += f2; number(buf, dist_of(k));
                buf += f3; buf += atom_text[(size_t)cj[k]];
                buf += f4; buf += ctype_text[ctype[k]];
                buf += f5; buf += (k + 1 < total) ? ",\n" : "\n";
            }
            if (buf.size() != off[(size_t)b + 1] - off[(size_t)b]) { failed = 1; break; }
            if (map) memcpy(map + off[(size_t)b], buf.data(), buf.size());
            else if (!write_all(buf.data(), buf.size(), (off_t)off[(size_t)b])) failed = 1;
        }
    });
    bool ok = !failed;
    if (map) {
        memcpy(map, "[\n", 2);
        memcpy(map + off[(size_t)nblocks], tail.data(), tail.size());
        if (munmap(map, (size_t)file_bytes) != 0) ok = false;
    } else {
        ok = ok && write_all("[\n", 2, 0) && write_all(tail.data(), tail.size(), (off_t)off[(size_t)nblocks]);
    }
    if (close(fd) != 0) ok = false;
    return ok ? 0 : -4;
}
