// gravity_io.cpp — host-side loaders of spherical-harmonics coefficient files, part of the
// propagation path's boundary ("parsers run on host, tables go to GPU").
//
// Behaviour follows GravityFieldData::from_cof / ::load of the reference
// (nyx-core/src/io/gravity.rs:150-367, 370-501) including its quirks:
//   * only lines starting with 'R' are data in a .cof; fields are whitespace separated;
//   * C and S are glued together when S is negative ("1.0e-06-2.0e-07"): detected by counting '-';
//   * reading stops at the first line whose degree exceeds the request;
//   * coefficients with order > request are skipped but still counted for the reported max order;
//   * the reported degree/order are the maxima SEEN, not the request (io/gravity.rs:330-366).
// Output layout differs from the reference's dense DMatrix: packed lower-triangular,
// idx(n, m) = n (n + 1) / 2 + m, which is what the device tables are built from.

#include "../../include/nyx_hip.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

void nyx_set_error(const char *fmt, ...);  // abi.cpp

namespace {

bool read_all(const char *path, bool gunzipped, std::string &out) {
    if (gunzipped) {
        gzFile f = gzopen(path, "rb");
        if (!f) return false;
        char buf[1 << 16];
        int n;
        while ((n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
        gzclose(f);
        return n == 0;
    }
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    std::fclose(f);
    return true;
}

void split_ws(const std::string &line, std::vector<std::string> &items) {
    items.clear();
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && std::isspace((unsigned char)line[i])) ++i;
        size_t j = i;
        while (j < line.size() && !std::isspace((unsigned char)line[j])) ++j;
        if (j > i) items.emplace_back(line, i, j - i);
        i = j;
    }
}

bool parse_usize(const std::string &s, long &v) {
    if (s.empty()) return false;
    for (char ch : s)
        if (ch < '0' || ch > '9') return false;  // usize::from_str: digits only (a leading '+' is accepted by Rust, not seen in files)
    v = std::strtol(s.c_str(), nullptr, 10);
    return true;
}

bool parse_f64(const std::string &s, double &v) {
    if (s.empty()) return false;
    char *end = nullptr;
    v = std::strtod(s.c_str(), &end);
    return end && *end == '\0';
}

std::vector<std::string> split_char(const std::string &s, char ch) {
    std::vector<std::string> parts;
    size_t start = 0;
    for (;;) {
        size_t p = s.find(ch, start);
        if (p == std::string::npos) {
            parts.emplace_back(s, start);
            break;
        }
        parts.emplace_back(s, start, p - start);
        start = p + 1;
    }
    return parts;
}

struct Packed {
    int degree;
    std::vector<double> c, s;
    explicit Packed(int d) : degree(d), c((size_t)(d + 1) * (d + 2) / 2, 0.0), s(c.size(), 0.0) {}
    void set(long n, long m, double cv, double sv) {
        if (m > n) return;  // never present in a well-formed file; the dense reference matrix would hold it unused
        size_t i = (size_t)n * (n + 1) / 2 + (size_t)m;
        c[i] = cv;
        s[i] = sv;
    }
};

int finish(Packed &pk, long max_degree, long max_order, int32_t *out_degree, int32_t *out_order, double **c_nm, double **s_nm) {
    size_t bytes = pk.c.size() * sizeof(double);
    double *pc = (double *)std::malloc(bytes), *ps = (double *)std::malloc(bytes);
    if (!pc || !ps) {
        std::free(pc);
        std::free(ps);
        nyx_set_error("out of memory");
        return NYX_HIP_RC_BAD_ARG;
    }
    std::memcpy(pc, pk.c.data(), bytes);
    std::memcpy(ps, pk.s.data(), bytes);
    *c_nm = pc;
    *s_nm = ps;
    *out_degree = (int32_t)max_degree;
    *out_order = (int32_t)max_order;
    return NYX_HIP_RC_OK;
}

}  // namespace

extern "C" int32_t nyx_hip_load_cof(const char *path, int32_t degree, int32_t order, int32_t gunzipped, int32_t *out_degree,
                                    int32_t *out_order, double **c_nm, double **s_nm) {
    if (!path || degree < 0 || order < 0 || !out_degree || !out_order || !c_nm || !s_nm) {
        nyx_set_error("nyx_hip_load_cof: bad argument");
        return NYX_HIP_RC_BAD_ARG;
    }
    std::string data;
    if (!read_all(path, gunzipped != 0, data)) {
        nyx_set_error("File not found or unreadable: %s", path);
        return NYX_HIP_RC_BAD_ARG;
    }
    Packed pk(degree);
    long max_order = 0, max_degree = 0;
    std::vector<std::string> items;
    size_t pos = 0;
    long lno = 0;
    while (pos <= data.size()) {
        size_t nl = data.find('\n', pos);
        std::string line = data.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = (nl == std::string::npos) ? data.size() + 1 : nl + 1;
        const long this_lno = lno++;
        if (line.empty() || line[0] != 'R') continue;  // comment, header or "END"
        long cur_degree = 0, cur_order = 0;
        double cv = 0.0, sv = 0.0;
        split_ws(line, items);
        for (size_t ino = 0; ino < items.size(); ++ino) {
            const std::string &item = items[ino];
            if (ino == 0) continue;
            if (ino == 1) {
                if (!parse_usize(item, cur_degree)) {
                    nyx_set_error("Harmonics file: could not parse degree `%s` on line %ld", item.c_str(), this_lno);
                    return NYX_HIP_RC_BAD_ARG;
                }
            } else if (ino == 2) {
                if (!parse_usize(item, cur_order)) {
                    nyx_set_error("Harmonics file: could not parse order `%s` on line %ld", item.c_str(), this_lno);
                    return NYX_HIP_RC_BAD_ARG;
                }
            } else if (ino == 3) {
                bool ok = true;
                if (degree == 0) {
                    sv = 0.0;
                    ok = parse_f64(item, cv);
                } else {
                    long minus = 0;
                    for (char ch : item) minus += (ch == '-');
                    if ((minus == 3 && item[0] != '-') || minus == 4) {
                        std::vector<std::string> parts = split_char(item, '-');
                        if (parts.size() == 5) {  // both negative
                            ok = parse_f64("-" + parts[1] + "-" + parts[2], cv) && parse_f64("-" + parts[3] + "-" + parts[4], sv);
                        } else if (parts.size() >= 4) {  // C positive, S negative
                            ok = parse_f64(parts[0] + "-" + parts[1], cv) && parse_f64("-" + parts[2] + "-" + parts[3], sv);
                        } else {
                            ok = false;
                        }
                    } else {
                        ok = parse_f64(item, cv);
                    }
                }
                if (!ok) {
                    nyx_set_error("Harmonics file: could not parse C_nm/S_nm `%s` on line %ld", item.c_str(), this_lno);
                    return NYX_HIP_RC_BAD_ARG;
                }
            } else if (ino == 4) {
                if (!parse_f64(item, sv)) {
                    nyx_set_error("Harmonics file: could not parse S_nm `%s` on line %ld", item.c_str(), this_lno);
                    return NYX_HIP_RC_BAD_ARG;
                }
            } else {
                break;  // covariances are not stored
            }
        }
        if (cur_degree > degree) break;  // file is ordered by degree
        if (cur_order <= order) pk.set(cur_degree, cur_order, cv, sv);
        if (cur_order > max_order) max_order = cur_order;
        if (cur_degree > max_degree) max_degree = cur_degree;
    }
    return finish(pk, max_degree, max_order, out_degree, out_order, c_nm, s_nm);
}

extern "C" int32_t nyx_hip_load_shadr(const char *path, int32_t degree, int32_t order, int32_t gunzipped, int32_t *out_degree,
                                      int32_t *out_order, double **c_nm, double **s_nm) {
    if (!path || degree < 0 || order < 0 || !out_degree || !out_order || !c_nm || !s_nm) {
        nyx_set_error("nyx_hip_load_shadr: bad argument");
        return NYX_HIP_RC_BAD_ARG;
    }
    std::string data;
    if (!read_all(path, gunzipped != 0, data)) {
        nyx_set_error("File not found or unreadable: %s", path);
        return NYX_HIP_RC_BAD_ARG;
    }
    Packed pk(degree);
    long max_order = 0, max_degree = 0;
    std::vector<std::string> items;
    size_t pos = 0;
    long lno = 0;
    while (pos <= data.size()) {
        size_t nl = data.find('\n', pos);
        std::string line = data.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = (nl == std::string::npos) ? data.size() + 1 : nl + 1;
        const long this_lno = lno++;
        if (this_lno == 0) continue;  // SHADR header line
        for (char &ch : line)
            if (ch == ',') ch = ' ';
        long cur_degree = 0, cur_order = 0;
        double cv = 0.0, sv = 0.0;
        split_ws(line, items);
        for (size_t ino = 0; ino < items.size() && ino < 4; ++ino) {
            std::string item = items[ino];
            bool ok = true;
            if (ino == 0) ok = parse_usize(item, cur_degree);
            else if (ino == 1) ok = parse_usize(item, cur_order);
            else {
                for (char &ch : item)
                    if (ch == 'D') ch = 'E';
                ok = parse_f64(item, ino == 2 ? cv : sv);
            }
            if (!ok) {
                nyx_set_error("Harmonics file: could not parse field %zu `%s` on line %ld", ino, item.c_str(), this_lno);
                return NYX_HIP_RC_BAD_ARG;
            }
        }
        if (cur_degree > degree) break;
        if (cur_order <= order) pk.set(cur_degree, cur_order, cv, sv);
        if (cur_order > max_order) max_order = cur_order;
        if (cur_degree > max_degree) max_degree = cur_degree;
    }
    return finish(pk, max_degree, max_order, out_degree, out_order, c_nm, s_nm);
}

extern "C" void nyx_hip_free(void *p) { std::free(p); }
