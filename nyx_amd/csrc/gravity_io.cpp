// gravity_io.cpp — host-side loaders of spherical-harmonics coefficient files, part of the propagation path's boundary
// ("parsers run on host, tables go to GPU").
//
// Two text formats, one table.  Each format is a RECORD READER that turns one line into (degree, order, C, S) or says
// "not a data line"; a common driver applies the selection rules that GravityFieldData::from_cof / ::load observe
// (nyx-core/src/io/gravity.rs:150-367, 370-501) and that callers of the reference rely on:
//   * the file is ordered by degree: reading stops at the first record beyond the requested degree;
//   * records beyond the requested ORDER are not stored but still count for the reported maximum order;
//   * the reported degree / order are the maxima SEEN, not the request (io/gravity.rs:330-366).
// GMAT .cof: data lines start with 'R' (RECOEF n m C [S] ...); when S is negative the two numbers are written without a
// blank ("2.43e-06-1.40e-06") - split here at the end of the first NUMBER, which a float scanner finds by itself.
// SHADR: first line is a header; fields separated by blanks and / or commas; Fortran 'D' exponents.
// Output: packed lower-triangular arrays, idx(n, m) = n (n + 1) / 2 + m, which is what the device tables are built from
// (the reference keeps dense (N+1) x (N+1) matrices).

#include "../../include/nyx_hip.h"

#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

void nyx_set_error(const char *fmt, ...);  // abi.cpp

namespace {

struct Record {
    long degree = 0, order = 0;
    double c = 0.0, s = 0.0;
};
enum class Line { Data, Skip, Malformed };

// A cursor over the blank- (and, for SHADR, comma-) separated fields of one line; no copies of the line are made.
class Fields {
  public:
    Fields(const char *begin, const char *end, bool commas) : p_(begin), end_(end), commas_(commas) {}
    bool next(std::string &field) {
        while (p_ < end_ && sep(*p_)) ++p_;
        const char *q = p_;
        while (q < end_ && !sep(*q)) ++q;
        if (q == p_) return false;
        field.assign(p_, q);
        p_ = q;
        return true;
    }

  private:
    bool sep(char ch) const { return std::isspace((unsigned char)ch) || (commas_ && ch == ','); }
    const char *p_, *end_;
    bool commas_;
};

bool to_index(const std::string &s, long &v) {  // digits only, as usize::from_str accepts them in these files
    if (s.empty()) return false;
    for (char ch : s)
        if (!std::isdigit((unsigned char)ch)) return false;
    v = std::strtol(s.c_str(), nullptr, 10);
    return true;
}

// One or two numbers out of a field: "1.0e-06" or the glued "1.0e-06-2.0e-07".  Returns how many were read (0 = malformed).
int scan_numbers(const std::string &s, double &first, double &second) {
    const char *p = s.c_str();
    char *end = nullptr;
    first = std::strtod(p, &end);
    if (end == p) return 0;
    if (*end == '\0') return 1;
    const char *q = end;
    second = std::strtod(q, &end);
    return (end != q && *end == '\0') ? 2 : 0;
}

struct Problem {
    const char *what = nullptr;  // "degree", "order", "C_nm/S_nm", "S_nm", or a field number for SHADR
    std::string text;
};

// RECOEF n m C [S] [sigmas ...]
Line read_cof(const char *b, const char *e, bool want_s, Record &r, Problem &pb) {
    if (b == e || *b != 'R') return Line::Skip;  // comment, header, POTFIELD, END
    Fields f(b, e, false);
    std::string tag, fn, fm, fc, fs;
    f.next(tag);
    if (!f.next(fn)) return Line::Data;  // (a bare tag: all zeros, like the reference's item loop that never reaches a field)
    if (!to_index(fn, r.degree)) { pb = {"degree", fn}; return Line::Malformed; }
    if (!f.next(fm)) return Line::Data;
    if (!to_index(fm, r.order)) { pb = {"order", fm}; return Line::Malformed; }
    if (!f.next(fc)) return Line::Data;
    double c = 0.0, s = 0.0;
    const int got = scan_numbers(fc, c, s);
    if (got == 0 || (!want_s && got == 2)) { pb = {"C_nm/S_nm", fc}; return Line::Malformed; }
    r.c = c;
    if (got == 2) {
        r.s = s;       // glued pair: the next field, if any, is a sigma
        return Line::Data;
    }
    if (f.next(fs)) {
        double extra;
        if (scan_numbers(fs, s, extra) != 1) { pb = {"S_nm", fs}; return Line::Malformed; }
        r.s = want_s ? s : 0.0;
    }
    return Line::Data;
}

// n, m, C, S [, sigmas]  with blanks and / or commas, 'D' exponents
Line read_shadr(const char *b, const char *e, bool, Record &r, Problem &pb) {
    Fields f(b, e, true);
    std::string fld;
    static const char *const names[4] = {"field 0", "field 1", "field 2", "field 3"};
    if (!f.next(fld)) return Line::Skip;  // blank line
    for (int k = 0; k < 4; ++k) {
        if (k > 0 && !f.next(fld)) break;
        bool ok;
        if (k < 2) {
            ok = to_index(fld, k == 0 ? r.degree : r.order);
        } else {
            for (char &ch : fld)
                if (ch == 'D' || ch == 'd') ch = 'E';
            double extra;
            ok = scan_numbers(fld, k == 2 ? r.c : r.s, extra) == 1;
        }
        if (!ok) { pb = {names[k], fld}; return Line::Malformed; }
    }
    return Line::Data;
}

bool slurp(const char *path, bool gunzipped, std::string &out) {
    char buf[1 << 16];
    if (gunzipped) {
        gzFile f = gzopen(path, "rb");
        if (!f) return false;
        int n;
        while ((n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
        gzclose(f);
        return n == 0;
    }
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    std::fclose(f);
    return true;
}

typedef Line (*Reader)(const char *, const char *, bool, Record &, Problem &);

int load_table(const char *who, const char *path, int32_t degree, int32_t order, int32_t gunzipped, int skip_lines, Reader reader,
               int32_t *out_degree, int32_t *out_order, double **c_nm, double **s_nm) {
    if (!path || degree < 0 || order < 0 || !out_degree || !out_order || !c_nm || !s_nm) {
        nyx_set_error("%s: bad argument", who);
        return NYX_HIP_RC_BAD_ARG;
    }
    std::string text;
    if (!slurp(path, gunzipped != 0, text)) {
        nyx_set_error("File not found or unreadable: %s", path);
        return NYX_HIP_RC_BAD_ARG;
    }
    const size_t n_coef = (size_t)(degree + 1) * (size_t)(degree + 2) / 2;
    std::vector<double> c(n_coef, 0.0), s(n_coef, 0.0);
    long seen_degree = 0, seen_order = 0, lno = 0;
    const char *p = text.data(), *const end = p + text.size();
    for (; p <= end; ++lno) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *b = p;
        p = nl ? nl + 1 : end + 1;
        if (lno < skip_lines) continue;
        Record r;
        Problem pb;
        const Line kind = reader(b, le, degree != 0, r, pb);
        if (kind == Line::Skip) continue;
        if (kind == Line::Malformed) {
            nyx_set_error("Harmonics file: could not parse %s `%s` on line %ld", pb.what, pb.text.c_str(), lno);
            return NYX_HIP_RC_BAD_ARG;
        }
        if (r.degree > degree) break;
        if (r.order <= order && r.order <= r.degree) {
            const size_t i = (size_t)r.degree * (size_t)(r.degree + 1) / 2 + (size_t)r.order;
            c[i] = r.c;
            s[i] = r.s;
        }
        if (r.order > seen_order) seen_order = r.order;
        if (r.degree > seen_degree) seen_degree = r.degree;
    }
    double *pc = (double *)std::malloc(n_coef * sizeof(double)), *ps = (double *)std::malloc(n_coef * sizeof(double));
    if (!pc || !ps) {
        std::free(pc);
        std::free(ps);
        nyx_set_error("out of memory");
        return NYX_HIP_RC_BAD_ARG;
    }
    std::memcpy(pc, c.data(), n_coef * sizeof(double));
    std::memcpy(ps, s.data(), n_coef * sizeof(double));
    *c_nm = pc;
    *s_nm = ps;
    *out_degree = (int32_t)seen_degree;
    *out_order = (int32_t)seen_order;
    return NYX_HIP_RC_OK;
}

}  // namespace

extern "C" int32_t nyx_hip_load_cof(const char *path, int32_t degree, int32_t order, int32_t gunzipped, int32_t *out_degree,
                                    int32_t *out_order, double **c_nm, double **s_nm) {
    return load_table("nyx_hip_load_cof", path, degree, order, gunzipped, 0, read_cof, out_degree, out_order, c_nm, s_nm);
}

extern "C" int32_t nyx_hip_load_shadr(const char *path, int32_t degree, int32_t order, int32_t gunzipped, int32_t *out_degree,
                                      int32_t *out_order, double **c_nm, double **s_nm) {
    return load_table("nyx_hip_load_shadr", path, degree, order, gunzipped, 1, read_shadr, out_degree, out_order, c_nm, s_nm);
}

extern "C" void nyx_hip_free(void *p) { std::free(p); }
