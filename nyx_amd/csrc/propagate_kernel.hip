// propagate_kernel.hip — the MI355X (gfx950) ensemble integrator.
//
// One workgroup integrates 64 trajectories from t0 to tf in a single launch:
//
//   * lane <-> trajectory.  Every lane of every wave of the workgroup is bound to the same
//     trajectory slot, so all shared tables (Stokes coefficients and Legendre recursion constants,
//     column schedule) are WAVE-UNIFORM and are fetched with scalar loads (4 x s_load_dwordx16 = four
//     64-byte harmonics entries per batch) straight into SGPRs: the f64 VALU ops take them as scalar
//     operands, no LDS/VGPR traffic for the big table at all.
//   * the waves of the workgroup are ROLE-SPECIALISED (all roles also carry harmonics columns):
//       wave 0  "integrator"    RK state machine of the 64 trajectories: per-lane adaptive step,
//                               accept/reject, integer-ns epoch bookkeeping (reference instance.rs:87-493),
//                               stage combination, two-body term, final accumulation of k_i.
//       wave 1  "almanac"       everything that depends only on the stage EPOCH — body-fixed DCM
//                               (3 sincos), Sun/Moon Chebyshev chains — one stage AHEAD, into LDS.
//       wave 2  "perturbations" position-dependent third-body and SRP/eclipse terms of the current stage.
//       wave 3+ "columns"       spherical-harmonics column workers.
//     Roles only meet in LDS; two workgroup barriers per force evaluation.  The split keeps every code
//     path under 128 VGPRs so that 16 waves (4 per SIMD) fit and hide the scalar-load latency.
//   * the spherical-harmonics double sum (reference gravity_field.rs:148-268), ~97 % of the work,
//     is split BY COLUMN (order m) over the waves.  Columns of the normalised derived-Legendre table
//     are independent given u = z/r, so each wave runs a rolling 2-term recursion down its columns
//     with O(1) registers instead of the reference's (N+3)^2 matrix; rho^n is folded into the
//     recursion and (s+it)^m into a per-column complex power.
//   * the 16 stage derivatives k_i, the Butcher tableau and the ephemeris records live in LDS.
//
// FP64 VALU bound by design (no MFMA: there is no dense contraction; HBM traffic is ~270 B per
// trajectory per launch).  Compiled with -ffp-contract=off: the RK / two-body part reproduces the
// reference's operation order (bit-exact golden vectors); FMAs in the harmonics are explicit.

#include <hip/hip_runtime.h>

#include "../../include/nyx_hip.h"
#include "devcfg.h"

#define CAS __attribute__((address_space(4)))
typedef const CAS DevCfg *CfgPtr;
typedef const CAS HarmEntry *HarmPtr;
typedef const CAS ColHdr *ColPtr;

#define DEVFN static __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// hifitime conversions (restated; see oracle/nyx_oracle.c for the reference call sites)
// ---------------------------------------------------------------------------------------------

DEVFN int64_t seconds_to_ns(double s) {
    double total = s * 1e9;
    if (total != total) return 0;
    if (total >= 9.2233720368547758e18) return INT64_MAX;
    if (total <= -9.2233720368547758e18) return INT64_MIN;
    return (int64_t)total;  // `as i64`: truncation toward zero
}

// floor-div / mod by 1e9 without the 64-bit integer divide (one f64 estimate + fix-up)
DEVFN void divmod_1e9(int64_t v, int64_t &q, int64_t &r) {
    int64_t e = (int64_t)((double)v * 1e-9);
    int64_t rem = v - e * 1000000000LL;
    if (rem < 0) { e -= 1; rem += 1000000000LL; }
    if (rem < 0) { e -= 1; rem += 1000000000LL; }
    if (rem >= 1000000000LL) { e += 1; rem -= 1000000000LL; }
    if (rem >= 1000000000LL) { e += 1; rem -= 1000000000LL; }
    q = e;
    r = rem;
}

DEVFN double ns_to_seconds(int64_t ns) {
    const int64_t NS_PER_CENTURY = 3155760000000000000LL;
    if (ns >= 0 && ns < NS_PER_CENTURY) {
        int64_t q, r;
        divmod_1e9(ns, q, r);
        return (double)q + (double)r * 1e-9;
    }
    int64_t cent;
    if (ns < 0) {
        cent = (ns >= -NS_PER_CENTURY) ? -1 : -2;
    } else {
        cent = (ns < 2 * NS_PER_CENTURY) ? 1 : 2;
    }
    int64_t rem = ns - cent * NS_PER_CENTURY;
    int64_t q, r;
    divmod_1e9(rem, q, r);
    return (double)cent * 3155760000.0 + (double)q + (double)r * 1e-9;
}

DEVFN double norm3(double x, double y, double z) { return sqrt(x * x + y * y + z * z); }
DEVFN double cube(double x) { return x * (x * x); }  // f64::powi(3)
DEVFN double clamp02(double x) { return x < 0.0 ? 0.0 : (x > 2.0 ? 2.0 : x); }

// ---------------------------------------------------------------------------------------------
// Epoch-only data of one stage: body-fixed DCM and body positions
// ---------------------------------------------------------------------------------------------

// LDS slot of one stage's epoch data, per lane: m[9] (DCM inertial -> body-fixed, row-major) then
// bp[DEV_MAX_SLOTS][3] (slot positions w.r.t. the integration centre).  Field-major: slot[f * 64 + lane].
#define ED_FIELDS (9 + 3 * DEV_MAX_SLOTS)

DEVFN void rotation_dcm(const CAS DevRot &rot, double et_s, double *m) {
    const double DEG = 3.14159265358979323846 / 180.0;
    const double HALF_PI = 1.57079632679489661923;
    const double d = et_s / 86400.0;
    const double T = et_s / (86400.0 * 36525.0);
    const double ra = (rot.ra[0] + rot.ra[1] * T + rot.ra[2] * T * T) * DEG;
    const double dec = (rot.dec[0] + rot.dec[1] * T + rot.dec[2] * T * T) * DEG;
    const double w = (rot.w[0] + rot.w[1] * d + rot.w[2] * d * d) * DEG;
    const double a1 = HALF_PI + ra, a2 = HALF_PI - dec, a3 = w;
    double s1, c1, s2, c2, s3, c3;
    sincos(a1, &s1, &c1);
    sincos(a2, &s2, &c2);
    sincos(a3, &s3, &c3);
    m[0] = c3 * c1 - s3 * c2 * s1;
    m[1] = c3 * s1 + s3 * c2 * c1;
    m[2] = s3 * s2;
    m[3] = -s3 * c1 - c3 * c2 * s1;
    m[4] = -s3 * s1 + c3 * c2 * c1;
    m[5] = c3 * s2;
    m[6] = s2 * s1;
    m[7] = -s2 * c1;
    m[8] = c2;
}

// SPK type 2 evaluation (Clenshaw); record index is per lane, metadata is uniform.  `records` is the
// LDS copy of the segment table when it fits (cfg->rec_in_lds), else the global array.  The 16-wide
// coefficient window is loaded before the recurrence starts (the table is padded by 16 doubles), so the
// loads are independent of the serial w0/w1/w2 chain.
#define CHEB_MAXC 16
template <typename P>
DEVFN int cheby_eval(const CAS DevSeg &sg, P records, double et_s, double *r3) {
    const double rel = (et_s - sg.init_et) / sg.interval;
    int idx = (int)floor(rel);
    int st = NYX_HIP_OK;
    if (idx < 0 || idx > sg.n_rec || (idx == sg.n_rec && et_s > sg.end_et)) st = NYX_HIP_ERR_EPHEM_RANGE;
    idx = idx < 0 ? 0 : (idx >= sg.n_rec ? sg.n_rec - 1 : idx);
    const int nc = sg.n_coef;
    P rec = records + sg.offset + idx * sg.stride;
    const double t = (et_s - rec[0]) / rec[1];
    const double two_t = 2.0 * t;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        P cf = rec + 2 + c * nc;
        double cv[CHEB_MAXC];
#pragma unroll
        for (int j = 0; j < CHEB_MAXC; ++j) cv[j] = cf[j];
        double w0 = 0.0, w1 = 0.0, w2;
#pragma unroll
        for (int j = CHEB_MAXC - 1; j >= 1; --j) {
            if (j < nc) {  // uniform
                w2 = w1;
                w1 = w0;
                w0 = cv[j] + (two_t * w1 - w2);
            }
        }
        r3[c] = cv[0] + (t * w0 - w1);
    }
    return st;
}

template <typename P>
DEVFN int epoch_data(CfgPtr cfg, P records, int64_t epoch_ns, double *slot, int lane) {
    const double et = ns_to_seconds(epoch_ns);
    int status = NYX_HIP_OK;
    if (cfg->has_grav) {
        double m[9];
        rotation_dcm(cfg->g_rot, et, m);
#pragma unroll
        for (int q = 0; q < 9; ++q) slot[q * DEV_LANES + lane] = m[q];
    }
    const int ns = cfg->n_slots;
#pragma unroll
    for (int s = 0; s < DEV_MAX_SLOTS; ++s) {
        if (s < ns) {
            double b0 = 0.0, b1 = 0.0, b2 = 0.0;
            const int nch = cfg->slot[s].n_chain;
            for (int k = 0; k < nch; ++k) {
                double p[3];
                const int sgi = cfg->slot[s].seg[k];
                int st = cheby_eval(cfg->seg[sgi], records, et, p);
                if (st) status = st;
                const double sg = cfg->slot[s].sign[k];
                b0 = b0 + sg * p[0];
                b1 = b1 + sg * p[1];
                b2 = b2 + sg * p[2];
            }
            slot[(9 + 3 * s + 0) * DEV_LANES + lane] = b0;
            slot[(9 + 3 * s + 1) * DEV_LANES + lane] = b1;
            slot[(9 + 3 * s + 2) * DEV_LANES + lane] = b2;
        }
    }
    return status;
}

#define ED_BP(slot, s, c) (slot)[(9 + 3 * (s) + (c)) * DEV_LANES + lane]

// ---------------------------------------------------------------------------------------------
// Position-dependent non-harmonic terms (master, inside the harmonics window)
// ---------------------------------------------------------------------------------------------

// PointMasses::eom, reference dynamics/orbital.rs:214-247
DEVFN void point_masses_accel(CfgPtr cfg, const double *ed, int lane, const double *r, double *acc) {
    acc[0] = acc[1] = acc[2] = 0.0;
    const int npm = cfg->n_pm;
#pragma unroll
    for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
        if (k < npm) {
            const int s = cfg->pm_slot[k];
            const double pij[3] = {ED_BP(ed, s, 0), ED_BP(ed, s, 1), ED_BP(ed, s, 2)};
            const double r_ij3 = cube(norm3(pij[0], pij[1], pij[2]));
            const double rj0 = r[0] - pij[0], rj1 = r[1] - pij[1], rj2 = r[2] - pij[2];
            const double r_j3 = cube(norm3(rj0, rj1, rj2));
            const double nmu = -cfg->slot[s].mu;
            acc[0] += nmu * (rj0 / r_j3 + pij[0] / r_ij3);
            acc[1] += nmu * (rj1 / r_j3 + pij[1] / r_ij3);
            acc[2] += nmu * (rj2 / r_j3 + pij[2] / r_ij3);
        }
    }
}

DEVFN double circ_seg_area(double r, double d) { return r * r * acos(d / r) - d * sqrt(r * r - d * d); }

// anise Occultation.percentage restated (apparent-disk overlap); see oracle for the definition.
DEVFN double occultation_pct(double r_back, double r_front, const double *r_eb, const double *r_ls) {
    const double n_ls = norm3(r_ls[0], r_ls[1], r_ls[2]), n_eb = norm3(r_eb[0], r_eb[1], r_eb[2]);
    const double ls_p = (r_back >= n_ls) ? r_back : asin(r_back / n_ls);
    const double fo_p = (r_front >= n_eb) ? r_front : asin(r_front / n_eb);
    const double dot = r_ls[0] * r_eb[0] + r_ls[1] * r_eb[1] + r_ls[2] * r_eb[2];
    const double d_p = acos(-dot / (n_eb * n_ls));
    double pct;
    if (d_p - ls_p > fo_p) {
        pct = 0.0;
    } else if (fo_p > d_p + ls_p) {
        pct = 100.0;
    } else if (fabs(ls_p - fo_p) < d_p && d_p < ls_p + fo_p) {
        const double d1 = (d_p * d_p - ls_p * ls_p + fo_p * fo_p) / (2.0 * d_p);
        const double d2 = (d_p * d_p + ls_p * ls_p - fo_p * fo_p) / (2.0 * d_p);
        const double shadow = circ_seg_area(fo_p, d1) + circ_seg_area(ls_p, d2);
        if (shadow != shadow) {
            pct = 100.0;
        } else {
            const double nominal = 3.14159265358979323846 * (ls_p * ls_p);
            pct = 100.0 * shadow / nominal;
        }
    } else {
        pct = 100.0 * (fo_p * fo_p) / (ls_p * ls_p);
    }
    return pct;
}

// SolarPressure::eom (reference dynamics/solarpressure.rs:135-165) + ShadowModel::compute (cosmic/eclipse.rs:69-83)
DEVFN void srp_force(CfgPtr cfg, const double *ed, int lane, const double *r, double cr, double area, double *force) {
    const int ss = cfg->sun_slot;
    const double ps[3] = {ED_BP(ed, ss, 0), ED_BP(ed, ss, 1), ED_BP(ed, ss, 2)};
    const double rs0 = r[0] - ps[0], rs1 = r[1] - ps[1], rs2 = r[2] - ps[2];
    const double n = norm3(rs0, rs1, rs2);
    const double u0 = rs0 / n, u1 = rs1 / n, u2 = rs2 / n;
    const double sun_radius = cfg->slot[ss].radius;
    double best = 0.0;
    const int nsh = cfg->n_shadow;
#pragma unroll
    for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
        if (k < nsh) {
            const int sb = cfg->shadow_slot[k];
            double pb[3] = {0.0, 0.0, 0.0};
            double rad = cfg->central_radius;
            if (sb >= 0) {  // uniform
                pb[0] = ED_BP(ed, sb, 0); pb[1] = ED_BP(ed, sb, 1); pb[2] = ED_BP(ed, sb, 2);
                rad = cfg->slot[sb].radius;
            }
            const double r_eb[3] = {r[0] - pb[0], r[1] - pb[1], r[2] - pb[2]};
            const double r_ls[3] = {ps[0] - r[0], ps[1] - r[1], ps[2] - r[2]};
            const double pct = occultation_pct(sun_radius, rad, r_eb, r_ls);
            if (pct > best) best = pct;
        }
    }
    const double occult = best / 100.0;
    const double k = fabs(occult - 1.0);
    const double r_au = n / 149597870.700;
    const double inv = 1.0 / r_au;
    const double flux = (k * cfg->phi / cfg->c_m_s) * (inv * inv);
    const double scal = 1e-3 * cr * area * flux;
    force[0] = scal * u0;
    force[1] = scal * u1;
    force[2] = scal * u2;
}

// ---------------------------------------------------------------------------------------------
// Spherical harmonics, column-split.  Inputs are per lane (trajectory); every table operand is
// wave-uniform (scalar loads).  Scaled recursion for column c, rows n' = c..N+1:
//   At_c = rho * diag[c];  At_n' = (rho u) b[n'][c] At_{n'-1} - rho^2 c[n'][c] At_{n'-2}
// (At_n' = rho^(n'-c+1) A[n'][c]); per-column complex power (Rc, Ic) = (rho (s + i t))^(c-1).
// ---------------------------------------------------------------------------------------------

DEVFN void cpow_uniform(double zr, double zi, int e, double &pr, double &pi) {
    pr = 1.0;
    pi = 0.0;
    double br = zr, bi = zi;
    while (e) {  // e is wave-uniform
        if (e & 1) {
            const double t = pr * br - pi * bi;
            pi = pr * bi + pi * br;
            pr = t;
        }
        const double t = br * br - bi * bi;
        bi = 2.0 * (br * bi);
        br = t;
        e >>= 1;
    }
}

#define HARM_TERM(h)                                                              \
    {                                                                             \
        const double an = __builtin_fma((h).bb * rho_u, a1, -(((h).cc * rho2) * a2)); \
        s1 = __builtin_fma(an, (h).t1, s1);                                       \
        s2 = __builtin_fma(an, (h).t2, s2);                                       \
        s3 = __builtin_fma(an, (h).t3, s3);                                       \
        s4 = __builtin_fma(an, (h).t4, s4);                                       \
        s5 = __builtin_fma(an, (h).t5, s5);                                       \
        s6 = __builtin_fma(an, (h).t6, s6);                                       \
        a2 = a1;                                                                  \
        a1 = an;                                                                  \
    }

DEVFN uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

DEVFN ColHdr load_hdr(ColPtr cols, int c) {
    const ColHdr CAS &r = cols[c];
    ColHdr h;
    h.start = r.start; h.nb = r.nb; h.scale = r.scale; h.diag = r.diag; h._pad = 0.0;
    return h;
}

struct Partial4 {
    double x, y, z, w;
};

// Not inlined on purpose: the batch loop wants 64 SGPRs for its four in-flight table entries, which it only
// gets when it is register-allocated on its own, away from the role code that calls it.  Arguments arrive in
// VGPRs under the device-function ABI, so the wave-uniform ones are re-scalarised with v_readfirstlane.
static __device__ __attribute__((noinline)) Partial4 harmonics_partial(uint64_t cfg_u, uint64_t htab_u, uint64_t cols_u, int wave_v,
                                                                     double zr, double zi, double rho_u, double rho,
                                                                     double inv_rho) {
    CfgPtr cfg = (CfgPtr)uniform_u64(cfg_u);
    HarmPtr htab = (HarmPtr)uniform_u64(htab_u);
    ColPtr cols = (ColPtr)uniform_u64(cols_u);
    const int wave = __builtin_amdgcn_readfirstlane(wave_v);
    double px = 0.0, py = 0.0, pz = 0.0, pw = 0.0;
    const double rho2 = rho * rho;
    const int nr = cfg->n_ranges[wave];
    for (int q = 0; q < nr; ++q) {
        const int c0 = cfg->range_c0[wave][q];
        const int cnt = cfg->range_cnt[wave][q];
        double rc, ic;
        cpow_uniform(zr, zi, c0 - 1, rc, ic);
        ColHdr hd = load_hdr(cols, c0);  // the next column's header is fetched under this column's batches
        for (int c = c0; c < c0 + cnt; ++c) {
            const ColHdr hn = load_hdr(cols, c + 1);  // (the header array has a spare tail entry)
            HarmPtr e = htab + hd.start;
            const int nb = hd.nb;
            double a1 = 0.0, a2 = hd.diag * inv_rho;
            double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0;
            for (int b = 0; b < nb; ++b, e += 4) {
                // four 64-byte entries per batch: 4 x s_load_dwordx16 in flight, then 40 f64 VALU ops
                const HarmEntry CAS &h0 = e[0];
                const HarmEntry CAS &h1 = e[1];
                const HarmEntry CAS &h2 = e[2];
                const HarmEntry CAS &h3 = e[3];
                HARM_TERM(h0)
                HARM_TERM(h1)
                HARM_TERM(h2)
                HARM_TERM(h3)
            }
            const double sc = rho * hd.scale;  // rho * c * sqrt(2)
            px = __builtin_fma(sc, __builtin_fma(rc, s1, ic * s2), px);
            py = __builtin_fma(sc, __builtin_fma(rc, s2, -(ic * s1)), py);
            pz = __builtin_fma(rho, __builtin_fma(rc, s3, ic * s4), pz);
            pw = pw - __builtin_fma(rc, s5, ic * s6);
            const double t = rc * zr - ic * zi;
            ic = rc * zi + ic * zr;
            rc = t;
            hd = hn;
        }
    }
    Partial4 r = {px, py, pz, pw};
    return r;
}

// ---------------------------------------------------------------------------------------------
// ErrorControl::estimate on the 9-vector (reference propagators/error_ctrl.rs:79-229).
// Elements 9..89 of the reference's 90-vector are zero without an STM and do not contribute.
// ---------------------------------------------------------------------------------------------

DEVFN double rss_step3(const double *e, const double *cand, const double *cur) {
    const double mag = norm3(cand[0] - cur[0], cand[1] - cur[1], cand[2] - cur[2]);
    const double err = norm3(e[0], e[1], e[2]);
    return (mag > sqrt(0.1)) ? err / mag : err;
}
DEVFN double rss_state3(const double *e, const double *cand, const double *cur) {
    const double mag = 0.5 * norm3(cand[0] + cur[0], cand[1] + cur[1], cand[2] + cur[2]);
    const double err = norm3(e[0], e[1], e[2]);
    return (mag > 0.1) ? err / mag : err;
}

// nalgebra's 8-accumulator dot over the 9 leading entries of the 90-vector: entries 0..7 land
// in acc0..acc7, entry 8 in acc0 of the second block; the remaining blocks add zeros.
DEVFN double nalgebra_norm9(const double *x) {
    const double a0 = x[0] * x[0] + x[8] * x[8];
    double res = 0.0;
    res += a0 + x[4] * x[4];
    res += x[1] * x[1] + x[5] * x[5];
    res += x[2] * x[2] + x[6] * x[6];
    res += x[3] * x[3] + x[7] * x[7];
    return sqrt(res);
}

DEVFN double error_estimate(int ec, const double *e, const double *cand, const double *cur) {
    double tmp[9];
    switch (ec) {
    case NYX_HIP_RSS_CARTESIAN_STATE: return fmax(rss_state3(e, cand, cur), rss_state3(e + 3, cand + 3, cur + 3));
    case NYX_HIP_RSS_CARTESIAN_STEP: return fmax(rss_step3(e, cand, cur), rss_step3(e + 3, cand + 3, cur + 3));
    case NYX_HIP_RSS_STATE: {
        for (int i = 0; i < 9; ++i) tmp[i] = cand[i] + cur[i];
        const double mag = 0.5 * nalgebra_norm9(tmp), err = nalgebra_norm9(e);
        return (mag > 0.1) ? err / mag : err;
    }
    case NYX_HIP_RSS_STEP: {
        for (int i = 0; i < 9; ++i) tmp[i] = cand[i] - cur[i];
        const double mag = nalgebra_norm9(tmp), err = nalgebra_norm9(e);
        return (mag > sqrt(0.1)) ? err / mag : err;
    }
    case NYX_HIP_LARGEST_ERROR: {
        double mx = 0.0;
        for (int i = 0; i < 9; ++i) {
            const double dl = cand[i] - cur[i];
            const double er = (dl > 0.1) ? fabs(e[i] / dl) : fabs(e[i]);
            if (er > mx) mx = er;
        }
        return mx;
    }
    case NYX_HIP_LARGEST_STATE: {
        double mag = 0.0, err = 0.0;
        for (int i = 0; i < 9; ++i) { mag += 0.5 * fabs(cand[i] + cur[i]); err += fabs(e[i]); }
        return (mag > 0.1) ? err / mag : err;
    }
    default: {
        double mag = 0.0, err = 0.0;
        for (int i = 0; i < 9; ++i) { mag += fabs(cand[i] - cur[i]); err += fabs(e[i]); }
        return (mag > 0.1) ? err / mag : err;
    }
    }
}

// ---------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------

#define NIN 5
// timing-only debug switches (NYX_HIP_DEBUG env, never set in production): results are physically wrong
#define DBG_SKIP_SERIAL 0x100
#define DBG_SKIP_HARMONICS 0x200
#define KB(stage, comp) kbuf[((stage)*6 + (comp)) * DEV_LANES + lane]

// Integrator state that is only touched between attempts lives in LDS (per lane, field-major), not in
// registers: the stage loop then keeps ~30 VGPRs of integrator state live instead of ~90 (no scratch spills).
#define CS_FIELDS 21
struct ColdState {
    int64_t epoch, stop, step_size, prev_step, det_step, n_acc, n_rej, n_evals;
    double y[9];
    double h, det_error;
    int det_attempts, attempts, status;
    bool done, fresh, is_final, fixed, prev_kind, backprop, massless;
};
#define CS_I64(f) __double_as_longlong(cs[(f)*DEV_LANES + lane])
DEVFN void cold_load(const double *cs, int lane, ColdState &c) {
    c.epoch = CS_I64(0); c.stop = CS_I64(1); c.step_size = CS_I64(2); c.prev_step = CS_I64(3);
    c.det_step = CS_I64(4); c.n_acc = CS_I64(5); c.n_rej = CS_I64(6); c.n_evals = CS_I64(7);
#pragma unroll
    for (int e = 0; e < 9; ++e) c.y[e] = cs[(8 + e) * DEV_LANES + lane];
    c.h = cs[17 * DEV_LANES + lane];
    c.det_error = cs[18 * DEV_LANES + lane];
    const int64_t a = CS_I64(19), b = CS_I64(20);
    c.det_attempts = (int)(a & 0xffff); c.attempts = (int)((a >> 16) & 0xffff); c.status = (int)((a >> 32) & 0xffff);
    c.done = b & 1; c.fresh = b & 2; c.is_final = b & 4; c.fixed = b & 8; c.prev_kind = b & 16; c.backprop = b & 32; c.massless = b & 64;
}
#define CS_SET_I64(f, v) cs[(f)*DEV_LANES + lane] = __longlong_as_double(v)
DEVFN void cold_store(double *cs, int lane, const ColdState &c) {
    CS_SET_I64(0, c.epoch); CS_SET_I64(1, c.stop); CS_SET_I64(2, c.step_size); CS_SET_I64(3, c.prev_step);
    CS_SET_I64(4, c.det_step); CS_SET_I64(5, c.n_acc); CS_SET_I64(6, c.n_rej); CS_SET_I64(7, c.n_evals);
#pragma unroll
    for (int e = 0; e < 9; ++e) cs[(8 + e) * DEV_LANES + lane] = c.y[e];
    cs[17 * DEV_LANES + lane] = c.h;
    cs[18 * DEV_LANES + lane] = c.det_error;
    const int64_t a = (int64_t)(c.det_attempts & 0xffff) | ((int64_t)(c.attempts & 0xffff) << 16) | ((int64_t)(c.status & 0xffff) << 32);
    const int64_t b = (c.done ? 1 : 0) | (c.fresh ? 2 : 0) | (c.is_final ? 4 : 0) | (c.fixed ? 8 : 0) | (c.prev_kind ? 16 : 0) |
                      (c.backprop ? 32 : 0) | (c.massless ? 64 : 0);
    CS_SET_I64(19, a); CS_SET_I64(20, b);
}
#define CS_Y(e) L.cs[(8 + (e)) * DEV_LANES + lane]

// LDS carve (doubles unless noted), see nyx_kernel_lds_bytes()
struct LdsMap {
    double *kbuf;   // [16][6][64]    stage derivatives k_i
    double *tabl;   // [16*16 + 3*16] Butcher tableau: rows of A (padded to 16), b, b - b*, c
    double *ys;     // [6][64]        stage state published by the integrator
    double *inb;    // [NIN][64]      zr, zi, rho_u, rho, 1/rho
    double *ed;     // [2][ED_FIELDS][64]  epoch data, double-buffered by stage parity
    double *pert;   // [6][64]        point-mass accel (3) and SRP force / mass (3)
    double *step;   // [2][64]        epoch (as i64 bits) and h of the current attempt
    double *cs;     // [CS_FIELDS][64] integrator cold state
    double *part;   // [P][4][64]     harmonics partials (wave 0's slot unused)
    int *edst;      // [2][64]        almanac status per buffer
    int *pertst;    // [64]
    int *ctl;       // [16]
    double *rec;    // [rec_doubles]
};

DEVFN LdsMap carve_lds(char *smem, int n_waves) {
    LdsMap m;
    double *p = (double *)smem;
    m.kbuf = p; p += DEV_MAX_STAGES * 6 * DEV_LANES;
    m.tabl = p; p += DEV_MAX_STAGES * DEV_MAX_STAGES + 3 * DEV_MAX_STAGES;
    m.ys = p; p += 6 * DEV_LANES;
    m.inb = p; p += NIN * DEV_LANES;
    m.ed = p; p += 2 * ED_FIELDS * DEV_LANES;
    m.pert = p; p += 6 * DEV_LANES;
    m.step = p; p += 2 * DEV_LANES;
    m.cs = p; p += CS_FIELDS * DEV_LANES;
    m.part = p; p += DEV_MAX_WAVES * 4 * DEV_LANES;
    m.edst = (int *)p; p += DEV_LANES;       // 2*64 ints
    m.pertst = (int *)p; p += DEV_LANES / 2; // 64 ints
    m.ctl = (int *)p; p += 8;
    m.rec = p;
    return m;
}

extern "C" size_t nyx_kernel_lds_bytes(int n_waves, int rec_doubles) {
    size_t d = (size_t)DEV_MAX_STAGES * 6 * DEV_LANES + DEV_MAX_STAGES * DEV_MAX_STAGES + 3 * DEV_MAX_STAGES + 6 * DEV_LANES +
               NIN * DEV_LANES + 2 * ED_FIELDS * DEV_LANES + 6 * DEV_LANES + 2 * DEV_LANES + CS_FIELDS * DEV_LANES + (size_t)DEV_MAX_WAVES * 4 * DEV_LANES +
               DEV_LANES + DEV_LANES / 2 + 8 + (size_t)rec_doubles;
    return d * sizeof(double) + 64;
}

#define PROF_T0() const int64_t pt0_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0
#define PROF_ADD(slot) if (prof_on) prof_acc[slot] += (int64_t)__builtin_readcyclecounter() - pt0_
#define A_ROW(i, j) tabl[(i)*DEV_MAX_STAGES + (j)]
#define B_COEF(i) tabl[DEV_MAX_STAGES * DEV_MAX_STAGES + (i)]
#define BD_COEF(i) tabl[DEV_MAX_STAGES * DEV_MAX_STAGES + DEV_MAX_STAGES + (i)]
#define C_COEF(i) tabl[DEV_MAX_STAGES * DEV_MAX_STAGES + 2 * DEV_MAX_STAGES + (i)]

// One role (or a merged set of roles) of the workgroup.  Every instantiation executes the SAME sequence of
// workgroup barriers; only the work between them differs, so that each role keeps just its own state live.
template <bool INTEG, bool ALMANAC, bool PERT>
DEVFN void role_loop(const DevBatch &bt, CfgPtr cfg, const DevCfg *cfg_g, HarmPtr htab, ColPtr cols,
                     const double *__restrict__ records, const LdsMap &L, const int lane, const int wave, const int nw) {
    double *const kbuf = L.kbuf;
    double *const tabl = L.tabl;
    const int stages = cfg->stages;
    const bool has_grav = cfg->has_grav != 0;
    const bool has_srp = cfg->has_srp != 0;
    const bool has_pm = cfg->n_pm > 0;
    const bool need_almanac = has_grav || cfg->n_slots > 0;
    const bool rec_in_lds = cfg->rec_in_lds != 0;
    const bool dbg_skip_serial = (cfg->flags & DBG_SKIP_SERIAL) != 0;
    const bool dbg_skip_harm = (cfg->flags & DBG_SKIP_HARMONICS) != 0;
    // optional cycle accounting (workgroup 0 only): [0] phase A, [1] window duty (almanac / pert), [2] harmonics,
    // [3] phase C, [4] step control, [5] total, [6] barrier waits, [7] realtime (100 MHz)
    const bool prof_on = bt.prof != nullptr && blockIdx.x == 0;
    int64_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t prof_start = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
    const int64_t prof_rt0 = prof_on ? (int64_t)__builtin_amdgcn_s_memrealtime() : 0;

    // ---- per-lane trajectory binding (every wave maps lane -> the same trajectory)
    const int64_t gid = (int64_t)blockIdx.x * DEV_LANES + lane;
    const bool valid = gid < bt.n;
    const int64_t idx = valid ? gid : bt.n - 1;

    // perturbation-wave constants
    double p_cr = 0.0, p_area = 0.0, p_mass = 1.0;

    if (INTEG) {
        ColdState c;
        c.epoch = bt.epoch_ns[idx];
        c.y[0] = bt.x[idx]; c.y[1] = bt.y[idx]; c.y[2] = bt.z[idx];
        c.y[3] = bt.vx[idx]; c.y[4] = bt.vy[idx]; c.y[5] = bt.vz[idx];
        c.y[6] = bt.cr ? bt.cr[idx] : 0.0;
        c.y[7] = bt.cd ? bt.cd[idx] : 0.0;
        c.y[8] = bt.mprop ? bt.mprop[idx] : 0.0;
        const int64_t duration = bt.use_end_epoch ? (bt.end_epoch_ns - c.epoch) : bt.duration_ns;
        c.stop = c.epoch + duration;
        c.backprop = duration < 0;
        c.step_size = (bt.step_in && bt.step_in[idx] != 0) ? bt.step_in[idx] : cfg->init_step_ns;
        c.prev_step = 0; c.prev_kind = false;
        c.fixed = cfg->fixed_step != 0;
        c.det_step = cfg->init_step_ns; c.det_error = 0.0; c.det_attempts = 1; c.attempts = 1;
        c.n_acc = c.n_rej = c.n_evals = 0;
        c.h = 0.0; c.status = NYX_HIP_OK; c.fresh = true; c.is_final = false;
        c.done = !valid || duration == 0;
        if (!c.done && c.y[8] < 0.0) { c.status = NYX_HIP_ERR_FUEL_EXHAUSTED; c.done = true; }  // dynamics.finally
        if (c.backprop) c.step_size = -c.step_size;
        const double mass = (bt.mdry ? bt.mdry[idx] : 0.0) + c.y[8] + (bt.mextra ? bt.mextra[idx] : 0.0);
        c.massless = has_srp && !(mass > 0.0);  // MasslessSpacecraft (spacecraft.rs:201-203)
        cold_store(L.cs, lane, c);
    }
    if (PERT) {
        // constant along the trajectory: no guidance law on this path => d(Cr, mass)/dt = 0
        p_cr = clamp02(bt.cr ? bt.cr[idx] : 0.0);
        p_area = bt.asrp ? bt.asrp[idx] : 0.0;
        p_mass = (bt.mdry ? bt.mdry[idx] : 0.0) + (bt.mprop ? bt.mprop[idx] : 0.0) + (bt.mextra ? bt.mextra[idx] : 0.0);
    }
    __syncthreads();

    for (;;) {  // one iteration = one RK attempt for every live lane (derive(), instance.rs:368-414)
        double h = 0.0;
        if (INTEG) {
            // start of a step: final-step test on integer epochs (instance.rs:149-186)
            ColdState c;
            cold_load(L.cs, lane, c);
            if (!c.done && c.fresh) {
                if ((!c.backprop && c.epoch + c.step_size > c.stop) || (c.backprop && c.epoch + c.step_size <= c.stop)) {
                    if (c.stop == c.epoch) {
                        c.done = true;
                    } else {
                        c.prev_step = c.step_size;
                        c.prev_kind = c.fixed;
                        c.step_size = c.stop - c.epoch;
                        c.fixed = true;
                        c.is_final = true;
                    }
                }
                c.attempts = 1;
                c.h = ns_to_seconds(c.step_size);
                c.fresh = false;
            }
            if (!c.done && c.massless) { c.status = NYX_HIP_ERR_MASSLESS; c.done = true; }
            cold_store(L.cs, lane, c);
            h = c.h;
            L.step[lane] = __longlong_as_double(c.epoch);
            L.step[DEV_LANES + lane] = h;
            if (!__any(!c.done)) {
                if (lane == 0) L.ctl[0] = 1;
            }
        }
        __syncthreads();  // B0: attempt published (or exit requested)
        if (((volatile int *)L.ctl)[0]) break;

        // prologue: epoch data of stage 0
        if (ALMANAC && need_almanac) {
            const int64_t ep = __double_as_longlong(L.step[lane]);
            int st = rec_in_lds ? epoch_data(cfg, (const double *)L.rec, ep, L.ed, lane) : epoch_data(cfg, records, ep, L.ed, lane);
            L.edst[lane] = st;
        }
        __syncthreads();  // Bp

        int st_att = NYX_HIP_OK;
        double wpre[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < stages; ++i) {
            double *const edc = L.ed + (i & 1) * ED_FIELDS * DEV_LANES;
            double s_ = 0.0, t_ = 0.0, u_ = 0.0, kfac = 0.0;
            double ys[6];
            PROF_T0();
            if (INTEG) {
                // ---- Phase A: stage state  y + h * sum_j a_ij k_j   (instance.rs:376-394)
                if (i == 0) {
#pragma unroll
                    for (int e = 0; e < 6; ++e) ys[e] = CS_Y(e);
                } else {
                    // wpre = sum_{j < i-1} a_ij k_j was accumulated in the previous window (same j order as the
                    // reference, zero coefficients add an exact 0); only the newest k enters on the critical path
                    const double a_last = A_ROW(i, i - 1);
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        const double wi = wpre[e] + a_last * KB(i - 1, e);
                        ys[e] = CS_Y(e) + h * wi;
                    }
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) L.ys[e * DEV_LANES + lane] = ys[e];
                if (need_almanac && L.edst[(i & 1) * DEV_LANES + lane]) st_att = L.edst[(i & 1) * DEV_LANES + lane];
                if (has_grav) {
                    // body-fixed position and the scaled inputs of the column recursion
                    const double rb0 = edc[0 * DEV_LANES + lane] * ys[0] + edc[1 * DEV_LANES + lane] * ys[1] + edc[2 * DEV_LANES + lane] * ys[2];
                    const double rb1 = edc[3 * DEV_LANES + lane] * ys[0] + edc[4 * DEV_LANES + lane] * ys[1] + edc[5 * DEV_LANES + lane] * ys[2];
                    const double rb2 = edc[6 * DEV_LANES + lane] * ys[0] + edc[7 * DEV_LANES + lane] * ys[1] + edc[8 * DEV_LANES + lane] * ys[2];
                    // one sqrt and one divide on the critical path; the rest are multiplies
                    const double r_ = norm3(rb0, rb1, rb2);
                    const double inv_r = 1.0 / r_;
                    s_ = rb0 * inv_r; t_ = rb1 * inv_r; u_ = rb2 * inv_r;
                    const double rho = cfg->g_re * inv_r;
                    kfac = (cfg->g_mu * inv_r) * cfg->g_inv_re;  // (mu / r) / R_eq
                    L.inb[0 * DEV_LANES + lane] = rho * s_;
                    L.inb[1 * DEV_LANES + lane] = rho * t_;
                    L.inb[2 * DEV_LANES + lane] = rho * u_;
                    L.inb[3 * DEV_LANES + lane] = rho;
                    L.inb[4 * DEV_LANES + lane] = r_ * cfg->g_inv_re;
                }
            }
            PROF_ADD(0);
            {
                PROF_T0();
                __syncthreads();  // B1: stage state and harmonics inputs published
                PROF_ADD(6);
            }
            const int64_t ptw_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;

            // ---- window --------------------------------------------------------------------------
            if (ALMANAC && need_almanac && i + 1 < stages) {
                // epoch-only data of the NEXT stage
                const int64_t ep = __double_as_longlong(L.step[lane]) + seconds_to_ns(C_COEF(i + 1) * L.step[DEV_LANES + lane]);
                double *edn = L.ed + ((i + 1) & 1) * ED_FIELDS * DEV_LANES;
                int st = NYX_HIP_OK;
                if (!dbg_skip_serial || i == 0)  // (timing switch: reuse the data of stages 0/1)
                    st = rec_in_lds ? epoch_data(cfg, (const double *)L.rec, ep, edn, lane) : epoch_data(cfg, records, ep, edn, lane);
                L.edst[((i + 1) & 1) * DEV_LANES + lane] = st;
            }
            if (PERT && (has_pm || has_srp)) {
                // position-dependent third-body and SRP terms of THIS stage
                double r[3] = {L.ys[0 * DEV_LANES + lane], L.ys[1 * DEV_LANES + lane], L.ys[2 * DEV_LANES + lane]};
                double a3[3] = {0.0, 0.0, 0.0}, f3[3] = {0.0, 0.0, 0.0};
                if (has_pm && !dbg_skip_serial) point_masses_accel(cfg, edc, lane, r, a3);
                if (has_srp && !dbg_skip_serial) {
                    srp_force(cfg, edc, lane, r, p_cr, p_area, f3);
                    f3[0] = f3[0] / p_mass; f3[1] = f3[1] / p_mass; f3[2] = f3[2] / p_mass;
                }
#pragma unroll
                for (int e = 0; e < 3; ++e) { L.pert[e * DEV_LANES + lane] = a3[e]; L.pert[(3 + e) * DEV_LANES + lane] = f3[e]; }
            }
            double acc[3] = {0.0, 0.0, 0.0};
            if (INTEG) {
                // two-body term of this stage (orbital.rs:86-92) and sum_{j<i} a_{i+1,j} k_j of the next one
                const double rmag = norm3(ys[0], ys[1], ys[2]);
                const double f = -cfg->mu_central / cube(rmag);
                acc[0] = f * ys[0]; acc[1] = f * ys[1]; acc[2] = f * ys[2];
#pragma unroll
                for (int e = 0; e < 6; ++e) wpre[e] = 0.0;
                if (i + 1 < stages) {
#pragma unroll
                    for (int j = 0; j < DEV_MAX_STAGES - 2; ++j) {
                        if (j < i) {  // uniform
                            const double a_nj = A_ROW(i + 1, j);
#pragma unroll
                            for (int e = 0; e < 6; ++e) wpre[e] += a_nj * KB(j, e);
                        }
                    }
                }
            }
            if (prof_on) prof_acc[1] += (int64_t)__builtin_readcyclecounter() - ptw_;
            const int64_t pth_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
            double px = 0.0, py = 0.0, pz = 0.0, pw = 0.0;
            if (has_grav && dbg_skip_harm && !INTEG) {
                double *pp = L.part + wave * 4 * DEV_LANES;
                pp[0 * DEV_LANES + lane] = 0.0; pp[1 * DEV_LANES + lane] = 0.0;
                pp[2 * DEV_LANES + lane] = 0.0; pp[3 * DEV_LANES + lane] = 0.0;
            }
            if (has_grav && !dbg_skip_harm) {
                const Partial4 pr = harmonics_partial((uint64_t)cfg, (uint64_t)htab, (uint64_t)cols, wave, L.inb[0 * DEV_LANES + lane],
                                                      L.inb[1 * DEV_LANES + lane], L.inb[2 * DEV_LANES + lane],
                                                      L.inb[3 * DEV_LANES + lane], L.inb[4 * DEV_LANES + lane]);
                px = pr.x; py = pr.y; pz = pr.z; pw = pr.w;
                if (!INTEG) {
                    double *pp = L.part + wave * 4 * DEV_LANES;
                    pp[0 * DEV_LANES + lane] = px; pp[1 * DEV_LANES + lane] = py;
                    pp[2 * DEV_LANES + lane] = pz; pp[3 * DEV_LANES + lane] = pw;
                }
            }
            if (prof_on) prof_acc[2] += (int64_t)__builtin_readcyclecounter() - pth_;
            {
                PROF_T0();
                __syncthreads();  // B2: partials / perturbations / next epoch data published
                PROF_ADD(6);
            }
            const int64_t ptc_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;

            if (INTEG) {
                // ---- Phase C: assemble the derivative in the reference's order (orbital.rs:80-114, spacecraft.rs:227-243)
                if (has_pm) {
                    acc[0] += L.pert[0 * DEV_LANES + lane]; acc[1] += L.pert[1 * DEV_LANES + lane]; acc[2] += L.pert[2 * DEV_LANES + lane];
                }
                if (has_grav) {
                    // fixed wave order; all 15 slots are read unconditionally (slots of absent waves hold an exact
                    // 0.0) so that the LDS reads carry no control dependence and pipeline
#pragma unroll
                    for (int w = 1; w < DEV_MAX_WAVES; ++w) {
                        const double *pp = L.part + w * 4 * DEV_LANES;
                        px += pp[0 * DEV_LANES + lane]; py += pp[1 * DEV_LANES + lane];
                        pz += pp[2 * DEV_LANES + lane]; pw += pp[3 * DEV_LANES + lane];
                    }
                    px *= kfac; py *= kfac; pz *= kfac; pw *= kfac;
                    const double al0 = px + pw * s_, al1 = py + pw * t_, al2 = pz + pw * u_;
                    acc[0] += edc[0 * DEV_LANES + lane] * al0 + edc[3 * DEV_LANES + lane] * al1 + edc[6 * DEV_LANES + lane] * al2;
                    acc[1] += edc[1 * DEV_LANES + lane] * al0 + edc[4 * DEV_LANES + lane] * al1 + edc[7 * DEV_LANES + lane] * al2;
                    acc[2] += edc[2 * DEV_LANES + lane] * al0 + edc[5 * DEV_LANES + lane] * al1 + edc[8 * DEV_LANES + lane] * al2;
                }
                if (has_srp) {
                    acc[0] += L.pert[3 * DEV_LANES + lane]; acc[1] += L.pert[4 * DEV_LANES + lane]; acc[2] += L.pert[5 * DEV_LANES + lane];
                }
                KB(i, 0) = ys[3]; KB(i, 1) = ys[4]; KB(i, 2) = ys[5];
                KB(i, 3) = acc[0]; KB(i, 4) = acc[1]; KB(i, 5) = acc[2];
            }
            if (prof_on) prof_acc[3] += (int64_t)__builtin_readcyclecounter() - ptc_;
        }

        const int64_t pts_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
        if (INTEG) {
            ColdState c;
            cold_load(L.cs, lane, c);
            double *const y = c.y;
            if (!c.done) c.n_evals += stages;
            // ---- next state and error estimate (instance.rs:401-414).  d(Cr, Cd, prop mass)/dt = 0.
            double next[9], err[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) { next[e] = y[e]; err[e] = 0.0; }
            for (int i = 0; i < stages; ++i) {
                const double ce = h * BD_COEF(i);
                const double cb = h * B_COEF(i);
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const double kv = KB(i, e);
                    err[e] += ce * kv;
                    next[e] += cb * kv;
                }
            }
            if (!c.done) {
                bool accept = false;
                if (st_att != NYX_HIP_OK) {
                    c.status = st_att;
                    c.done = true;
                } else if (c.fixed) {
                    c.det_step = c.step_size;
                    accept = true;
                } else {
                    c.det_error = error_estimate(cfg->error_ctrl, err, next, y);
                    if (c.det_error <= cfg->tol || h <= cfg->min_step_s || c.attempts >= cfg->attempts) {
                        bool nan = false;
#pragma unroll
                        for (int e = 0; e < 9; ++e) nan = nan || (next[e] != next[e]);
                        if (nan) {
                            c.status = NYX_HIP_ERR_NAN;
                            c.done = true;
                        } else {
                            c.det_step = seconds_to_ns(h);
                            if (c.det_error < cfg->tol) {
                                const double prop = 0.9 * h * pow(cfg->tol / c.det_error, cfg->inv_order);
                                h = (fabs(prop) > fabs(cfg->max_step_s)) ? cfg->max_step_s * copysign(1.0, prop) : prop;
                            }
                            c.step_size = seconds_to_ns(h);
                            const int64_t ab = c.step_size < 0 ? -c.step_size : c.step_size;
                            if (ab < cfg->min_step_ns) c.step_size = (c.step_size < 0) ? -cfg->min_step_ns : cfg->min_step_ns;
                            accept = true;
                        }
                    } else {
                        c.attempts += 1;
                        c.n_rej += 1;
                        const double prop = 0.9 * h * pow(cfg->tol / c.det_error, cfg->inv_order_m1);
                        h = (prop < cfg->min_step_s) ? cfg->min_step_s : prop;
                    }
                }
                if (accept) {
                    // single_step(): state.set(c.epoch + t, vec) with the Cr clamp, then finally()
                    c.epoch += c.det_step;
#pragma unroll
                    for (int e = 0; e < 9; ++e) y[e] = next[e];
                    y[6] = clamp02(y[6]);
                    c.n_acc += 1;
                    c.det_attempts = c.attempts;
                    if (y[8] < 0.0) { c.status = NYX_HIP_ERR_FUEL_EXHAUSTED; c.done = true; }
                    if (c.is_final) {
                        c.step_size = c.prev_step;
                        c.fixed = c.prev_kind;
                        if (c.backprop) c.step_size = -c.step_size;
                        c.is_final = false;
                        c.done = true;
                    }
                    c.fresh = true;
                }
            }
            c.h = h;
            cold_store(L.cs, lane, c);
        }
        if (prof_on) prof_acc[4] += (int64_t)__builtin_readcyclecounter() - pts_;
    }
    if (prof_on && lane == 0) {
        prof_acc[5] = (int64_t)__builtin_readcyclecounter() - prof_start;
        prof_acc[7] = (int64_t)__builtin_amdgcn_s_memrealtime() - prof_rt0;
        for (int q = 0; q < 8; ++q) bt.prof[wave * 8 + q] = prof_acc[q];
    }

    if (INTEG && valid) {
        ColdState c;
        cold_load(L.cs, lane, c);
        bt.o_epoch_ns[gid] = c.epoch;
        bt.o_x[gid] = c.y[0]; bt.o_y[gid] = c.y[1]; bt.o_z[gid] = c.y[2];
        bt.o_vx[gid] = c.y[3]; bt.o_vy[gid] = c.y[4]; bt.o_vz[gid] = c.y[5];
        if (bt.o_cr) bt.o_cr[gid] = c.y[6];
        if (bt.o_cd) bt.o_cd[gid] = c.y[7];
        if (bt.o_mprop) bt.o_mprop[gid] = c.y[8];
        if (bt.o_mdry) bt.o_mdry[gid] = bt.mdry ? bt.mdry[idx] : 0.0;
        if (bt.o_mextra) bt.o_mextra[gid] = bt.mextra ? bt.mextra[idx] : 0.0;
        if (bt.o_asrp) bt.o_asrp[gid] = bt.asrp ? bt.asrp[idx] : 0.0;
        if (bt.o_adrag) bt.o_adrag[gid] = bt.adrag ? bt.adrag[idx] : 0.0;
        if (bt.o_step) bt.o_step[gid] = c.step_size;
        if (bt.status) bt.status[gid] = c.status;
        if (bt.last_step_ns) bt.last_step_ns[gid] = c.det_step;
        if (bt.last_error) bt.last_error[gid] = c.det_error;
        if (bt.last_attempts) bt.last_attempts[gid] = c.det_attempts;
        if (bt.n_acc) bt.n_acc[gid] = c.n_acc;
        if (bt.n_rej) bt.n_rej[gid] = c.n_rej;
        if (bt.n_evals) bt.n_evals[gid] = c.n_evals;
    }
}

extern "C" __global__ void __launch_bounds__(DEV_MAX_WAVES *DEV_LANES)
    nyx_propagate_kernel(DevBatch bt, const DevCfg *cfg_g, const HarmEntry *htab_g, const ColHdr *cols_g,
                         const double *__restrict__ records) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & (DEV_LANES - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    const LdsMap L = carve_lds(smem, nw);
    double *const kbuf = L.kbuf;
    double *const tabl = L.tabl;
    CfgPtr cfg = (CfgPtr)cfg_g;
    HarmPtr htab = (HarmPtr)htab_g;
    ColPtr cols = (ColPtr)cols_g;

    const int stages = cfg->stages;
    const bool rec_in_lds = cfg->rec_in_lds != 0;

    // ---- one-time staging: ephemeris records and the Butcher tableau -> LDS (all waves cooperate)
    if (rec_in_lds) {
        const int nd = cfg->rec_doubles;
        for (int q = (int)threadIdx.x; q < nd; q += (int)blockDim.x) L.rec[q] = records[q];
    }
    for (int q = (int)threadIdx.x; q < DEV_MAX_STAGES * DEV_MAX_STAGES + 3 * DEV_MAX_STAGES; q += (int)blockDim.x) {
        double v;
        if (q < DEV_MAX_STAGES * DEV_MAX_STAGES) {
            const int i = q / DEV_MAX_STAGES, j = q % DEV_MAX_STAGES;
            v = (j < i && i < stages) ? cfg_g->a[i * (i - 1) / 2 + j] : 0.0;
        } else {
            const int r = q - DEV_MAX_STAGES * DEV_MAX_STAGES;
            const int which = r / DEV_MAX_STAGES, i = r % DEV_MAX_STAGES;
            v = (i < stages) ? (which == 0 ? cfg_g->b[i] : (which == 1 ? cfg_g->bdiff[i] : cfg_g->c[i])) : 0.0;
        }
        tabl[q] = v;
    }
    if (threadIdx.x == 0) L.ctl[0] = 0;
    for (int q = (int)threadIdx.x; q < DEV_MAX_WAVES * 4 * DEV_LANES; q += (int)blockDim.x) L.part[q] = 0.0;

    // ---- role dispatch (wave-uniform): merged roles when the workgroup has fewer than three waves
    if (nw == 1) {
        role_loop<true, true, true>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw);
    } else if (nw == 2) {
        if (wave == 0) role_loop<true, false, false>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw);
        else role_loop<false, true, true>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw);
    } else {
        if (wave == 0) role_loop<true, false, false>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw);
        else if (wave == 1) role_loop<false, true, false>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw);
        else if (wave == 2) role_loop<false, false, true>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw);
        else role_loop<false, false, false>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw);
    }
}

extern "C" hipError_t nyx_launch_propagate(const DevBatch &bt, const DevCfg *cfg, const HarmEntry *htab,
                                           const ColHdr *cols, const double *records, int n_waves, int rec_lds_doubles,
                                           hipStream_t stream) {
    const int64_t blocks = (bt.n + DEV_LANES - 1) / DEV_LANES;
    if (blocks == 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)nyx_propagate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(nyx_propagate_kernel, dim3((unsigned)blocks), dim3((unsigned)(n_waves * DEV_LANES)),
                       nyx_kernel_lds_bytes(n_waves, rec_lds_doubles), stream, bt, cfg, htab, cols, records);
    return hipGetLastError();
}
