// propagate_kernel.hip — the MI355X (gfx950) ensemble integrator.
//
// One workgroup integrates 64 trajectories from t0 to tf in a single launch:
//
//   * lane <-> trajectory.  Every lane of every wave of the workgroup is bound to the same
//     trajectory slot, so all shared tables (Butcher tableau, Stokes coefficients and Legendre
//     recursion constants, column schedule) are WAVE-UNIFORM and are fetched with scalar loads
//     (s_load_dwordx16 = one 64-byte harmonics entry) straight into SGPRs: the f64 VALU ops take
//     them as scalar operands, no LDS/VGPR traffic for tables at all.
//   * wave 0 ("master") owns the RK state machine of the 64 trajectories: per-lane adaptive step,
//     accept/reject, integer-nanosecond epoch bookkeeping (reference instance.rs:87-493).  The 16
//     stage derivatives k_i live in LDS (48 KiB per workgroup), not in registers.
//   * the spherical-harmonics double sum (reference gravity_field.rs:148-268), ~97 % of the work,
//     is split BY COLUMN (order m) over the P waves of the workgroup.  Columns of the normalised
//     derived-Legendre table are independent given u = z/r, so each wave runs a rolling 2-term
//     recursion down its columns with O(1) registers instead of the reference's (N+3)^2 matrix;
//     rho^n is folded into the recursion and (s+it)^m into a per-column complex power.  Partial
//     accelerations meet in LDS (2 barriers per force evaluation).
//   * everything that depends only on the stage EPOCH (body-fixed DCM: 3 sincos; Sun/Moon Chebyshev
//     chains) is computed by the master one stage ahead, inside the window in which the other waves
//     are busy with harmonics; the position-dependent third-body / SRP / eclipse terms run in the
//     same window.
//
// FP64 VALU bound by design (no MFMA: there is no dense contraction; HBM traffic is ~250 B per
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

struct EpochData {
    double m[9];                  // DCM inertial -> body-fixed, row-major
    double bp[DEV_MAX_SLOTS][3];  // slot positions w.r.t. the integration centre
    int32_t status;
};

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
DEVFN void epoch_data(CfgPtr cfg, P records, int64_t epoch_ns, EpochData &ed) {
    const double et = ns_to_seconds(epoch_ns);
    ed.status = NYX_HIP_OK;
    if (cfg->has_grav) rotation_dcm(cfg->g_rot, et, ed.m);
    const int ns = cfg->n_slots;
#pragma unroll
    for (int s = 0; s < DEV_MAX_SLOTS; ++s) {
        ed.bp[s][0] = ed.bp[s][1] = ed.bp[s][2] = 0.0;
        if (s < ns) {
            const int nch = cfg->slot[s].n_chain;
            for (int k = 0; k < nch; ++k) {
                double p[3];
                const int sgi = cfg->slot[s].seg[k];
                int st = cheby_eval(cfg->seg[sgi], records, et, p);
                if (st) ed.status = st;
                const double sg = cfg->slot[s].sign[k];
                ed.bp[s][0] = ed.bp[s][0] + sg * p[0];
                ed.bp[s][1] = ed.bp[s][1] + sg * p[1];
                ed.bp[s][2] = ed.bp[s][2] + sg * p[2];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Position-dependent non-harmonic terms (master, inside the harmonics window)
// ---------------------------------------------------------------------------------------------

// PointMasses::eom, reference dynamics/orbital.rs:214-247
DEVFN void point_masses_accel(CfgPtr cfg, const EpochData &ed, const double *r, double *acc) {
    acc[0] = acc[1] = acc[2] = 0.0;
    const int npm = cfg->n_pm;
#pragma unroll
    for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
        if (k < npm) {
            const int s = cfg->pm_slot[k];
            double pij[3];
#pragma unroll
            for (int q = 0; q < DEV_MAX_SLOTS; ++q)
                if (q == s) { pij[0] = ed.bp[q][0]; pij[1] = ed.bp[q][1]; pij[2] = ed.bp[q][2]; }
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
DEVFN void srp_force(CfgPtr cfg, const EpochData &ed, const double *r, double cr, double area, double *force) {
    const int ss = cfg->sun_slot;
    double ps[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < DEV_MAX_SLOTS; ++q)
        if (q == ss) { ps[0] = ed.bp[q][0]; ps[1] = ed.bp[q][1]; ps[2] = ed.bp[q][2]; }
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
            if (sb >= 0) {
#pragma unroll
                for (int q = 0; q < DEV_MAX_SLOTS; ++q)
                    if (q == sb) { pb[0] = ed.bp[q][0]; pb[1] = ed.bp[q][1]; pb[2] = ed.bp[q][2]; }
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

DEVFN void harmonics_partial(CfgPtr cfg, HarmPtr htab, ColPtr cols, int wave, double zr, double zi, double rho_u,
                             double rho, double inv_rho, double &px, double &py, double &pz, double &pw) {
    px = py = pz = pw = 0.0;
    const double rho2 = rho * rho;
    const int nr = cfg->n_ranges[wave];
    for (int q = 0; q < nr; ++q) {
        const int c0 = cfg->range_c0[wave][q];
        const int cnt = cfg->range_cnt[wave][q];
        double rc, ic;
        cpow_uniform(zr, zi, c0 - 1, rc, ic);
        for (int c = c0; c < c0 + cnt; ++c) {
            const ColHdr CAS &hd = cols[c];
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
        }
    }
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

extern "C" __global__ void __launch_bounds__(DEV_MAX_WAVES *DEV_LANES)
    nyx_propagate_kernel(DevBatch bt, const DevCfg *cfg_g, const HarmEntry *htab_g, const ColHdr *cols_g,
                         const double *__restrict__ records) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *kbuf = (double *)smem;                          // [16][6][64]
    double *inb = kbuf + DEV_MAX_STAGES * 6 * DEV_LANES;    // [5][64]: zr, zi, rho_u, rho, 1/rho
    double *part = inb + NIN * DEV_LANES;                     // [P-1][4][64]
    volatile int *ctl = (volatile int *)(part + (DEV_MAX_WAVES - 1) * 4 * DEV_LANES);
    double *rec_lds = (double *)(ctl + 16);  // [cfg->rec_doubles] when cfg->rec_in_lds

    const int lane = threadIdx.x & (DEV_LANES - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    CfgPtr cfg = (CfgPtr)cfg_g;
    HarmPtr htab = (HarmPtr)htab_g;
    ColPtr cols = (ColPtr)cols_g;

    // ephemeris records -> LDS (all waves cooperate), so the per-lane Chebyshev windows are LDS reads
    const bool rec_in_lds = cfg->rec_in_lds != 0;
    if (rec_in_lds) {
        const int nd = cfg->rec_doubles;
        for (int q = (int)threadIdx.x; q < nd; q += (int)blockDim.x) rec_lds[q] = records[q];
    }
    if (lane == 0 && wave == 0) ctl[0] = 0;
    __syncthreads();

    // ------------------------------------------------------------------ workers
    if (wave != 0) {
        for (;;) {
            __syncthreads();  // B1: inputs published (or exit requested)
            if (ctl[0]) break;
            const double zr = inb[0 * DEV_LANES + lane], zi = inb[1 * DEV_LANES + lane];
            const double rho_u = inb[2 * DEV_LANES + lane], rho = inb[3 * DEV_LANES + lane];
            const double inv_rho = inb[4 * DEV_LANES + lane];
            double px = 0.0, py = 0.0, pz = 0.0, pw = 0.0;
            if (!(cfg->flags & DBG_SKIP_HARMONICS))
                harmonics_partial(cfg, htab, cols, wave, zr, zi, rho_u, rho, inv_rho, px, py, pz, pw);
            double *pp = part + (wave - 1) * 4 * DEV_LANES;
            pp[0 * DEV_LANES + lane] = px;
            pp[1 * DEV_LANES + lane] = py;
            pp[2 * DEV_LANES + lane] = pz;
            pp[3 * DEV_LANES + lane] = pw;
            __syncthreads();  // B2: partials published
        }
        return;
    }

    // ------------------------------------------------------------------ master
    const int64_t gid = (int64_t)blockIdx.x * DEV_LANES + lane;
    const bool valid = gid < bt.n;
    const int64_t idx = valid ? gid : bt.n - 1;

    int64_t epoch = bt.epoch_ns[idx];
    double y[9];
    y[0] = bt.x[idx]; y[1] = bt.y[idx]; y[2] = bt.z[idx];
    y[3] = bt.vx[idx]; y[4] = bt.vy[idx]; y[5] = bt.vz[idx];
    y[6] = bt.cr ? bt.cr[idx] : 0.0;
    y[7] = bt.cd ? bt.cd[idx] : 0.0;
    y[8] = bt.mprop ? bt.mprop[idx] : 0.0;
    const double m_dry = bt.mdry ? bt.mdry[idx] : 0.0;
    const double m_extra = bt.mextra ? bt.mextra[idx] : 0.0;
    const double a_srp = bt.asrp ? bt.asrp[idx] : 0.0;

    const int stages = cfg->stages;
    const bool has_grav = cfg->has_grav != 0;
    const bool has_srp = cfg->has_srp != 0;
    const bool any_force = has_srp;  // drag is not on the device path yet (ctx_create refuses it)
    const int64_t min_step_ns = cfg->min_step_ns;
    const bool dbg_skip_serial = (cfg->flags & DBG_SKIP_SERIAL) != 0;

    const int64_t duration = bt.use_end_epoch ? (bt.end_epoch_ns - epoch) : bt.duration_ns;
    const int64_t stop = epoch + duration;
    const bool backprop = duration < 0;

    int64_t step_size = (bt.step_in && bt.step_in[idx] != 0) ? bt.step_in[idx] : cfg->init_step_ns;
    bool fixed = cfg->fixed_step != 0;
    int64_t det_step = cfg->init_step_ns;
    double det_error = 0.0;
    int det_attempts = 1;
    int64_t n_acc = 0, n_rej = 0, n_evals = 0;
    int status = NYX_HIP_OK;

    bool done = !valid || duration == 0;
    if (!done && y[8] < 0.0) { status = NYX_HIP_ERR_FUEL_EXHAUSTED; done = true; }  // dynamics.finally
    if (backprop) step_size = -step_size;

    bool fresh = true, is_final = false;
    int64_t prev_step = 0;
    bool prev_kind = false;
    double h = 0.0;
    int attempts = 1;

    EpochData cur, nxt;

    while (__any(!done)) {
        // ---- start of a step (per lane): final-step test on integer epochs (instance.rs:149-186)
        if (!done && fresh) {
            if ((!backprop && epoch + step_size > stop) || (backprop && epoch + step_size <= stop)) {
                if (stop == epoch) {
                    done = true;
                } else {
                    prev_step = step_size;
                    prev_kind = fixed;
                    step_size = stop - epoch;
                    fixed = true;
                    is_final = true;
                }
            }
            attempts = 1;
            h = ns_to_seconds(step_size);
            fresh = false;
        }
        if (!__any(!done)) break;

        // ---- one RK attempt for every live lane (derive(), instance.rs:368-414)
        const double cr = clamp02(y[6]);
        const double mass = m_dry + y[8] + m_extra;
        int st_att = NYX_HIP_OK;
        if (any_force && !(mass > 0.0)) st_att = NYX_HIP_ERR_MASSLESS;

        int a_idx = 0;
        for (int i = -1; i < stages; ++i) {
            double ys[6];
            double px = 0.0, py = 0.0, pz = 0.0, pw = 0.0;
            double s_ = 0.0, t_ = 0.0, u_ = 0.0, kfac = 0.0;
            if (i >= 0) {
                // ---- Phase A: stage state  y + h * sum_j a_ij k_j   (instance.rs:376-394)
                if (i == 0) {
#pragma unroll
                    for (int e = 0; e < 6; ++e) ys[e] = y[e];
                } else {
                    double wi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                    for (int j = 0; j < i; ++j) {
                        const double a_ij = cfg->a[a_idx + j];
                        if (a_ij != 0.0) {
#pragma unroll
                            for (int e = 0; e < 6; ++e) wi[e] += a_ij * KB(j, e);
                        }
                    }
                    a_idx += i;
#pragma unroll
                    for (int e = 0; e < 6; ++e) ys[e] = y[e] + h * wi[e];
                }
                if (cur.status) st_att = cur.status;
                if (has_grav) {
                    // body-fixed position and the scaled inputs of the column recursion
                    const double rb0 = cur.m[0] * ys[0] + cur.m[1] * ys[1] + cur.m[2] * ys[2];
                    const double rb1 = cur.m[3] * ys[0] + cur.m[4] * ys[1] + cur.m[5] * ys[2];
                    const double rb2 = cur.m[6] * ys[0] + cur.m[7] * ys[1] + cur.m[8] * ys[2];
                    const double r_ = norm3(rb0, rb1, rb2);
                    s_ = rb0 / r_; t_ = rb1 / r_; u_ = rb2 / r_;
                    const double rho = cfg->g_re / r_;
                    kfac = cfg->g_mu / r_ / cfg->g_re;  // (mu / r) / R_eq
                    inb[0 * DEV_LANES + lane] = rho * s_;
                    inb[1 * DEV_LANES + lane] = rho * t_;
                    inb[2 * DEV_LANES + lane] = rho * u_;
                    inb[3 * DEV_LANES + lane] = rho;
                    inb[4 * DEV_LANES + lane] = r_ / cfg->g_re;
                    if (nw > 1) __syncthreads();  // B1
                }
            }

            // ---- window: position-dependent non-harmonic terms for stage i, epoch data for stage i+1,
            //      and this wave's own share of the harmonics columns
            double acc[3] = {0.0, 0.0, 0.0};
            if (i >= 0) {
                const double rmag = norm3(ys[0], ys[1], ys[2]);
                const double f = -cfg->mu_central / cube(rmag);
                acc[0] = f * ys[0]; acc[1] = f * ys[1]; acc[2] = f * ys[2];
                if (cfg->n_pm > 0 && !dbg_skip_serial) {
                    double a3[3];
                    point_masses_accel(cfg, cur, ys, a3);
                    acc[0] += a3[0]; acc[1] += a3[1]; acc[2] += a3[2];
                }
            }
            double fsrp[3] = {0.0, 0.0, 0.0};
            if (i >= 0 && has_srp && !dbg_skip_serial) srp_force(cfg, cur, ys, cr, a_srp, fsrp);
            if (i + 1 < stages && !(dbg_skip_serial && i >= 0)) {
                const double dt = (i + 1 == 0) ? 0.0 : cfg->c[i + 1] * h;
                if (rec_in_lds) epoch_data(cfg, (const double *)rec_lds, epoch + seconds_to_ns(dt), nxt);
                else epoch_data(cfg, records, epoch + seconds_to_ns(dt), nxt);
            }
            if (i >= 0 && has_grav) {
                harmonics_partial(cfg, htab, cols, 0, inb[0 * DEV_LANES + lane], inb[1 * DEV_LANES + lane],
                                  inb[2 * DEV_LANES + lane], inb[3 * DEV_LANES + lane], inb[4 * DEV_LANES + lane], px, py, pz, pw);
                if (nw > 1) __syncthreads();  // B2
                // ---- Phase C: fold the partials (fixed wave order), rotate back
                for (int w = 1; w < nw; ++w) {
                    const double *pp = part + (w - 1) * 4 * DEV_LANES;
                    px += pp[0 * DEV_LANES + lane];
                    py += pp[1 * DEV_LANES + lane];
                    pz += pp[2 * DEV_LANES + lane];
                    pw += pp[3 * DEV_LANES + lane];
                }
                px *= kfac; py *= kfac; pz *= kfac; pw *= kfac;
                const double al0 = px + pw * s_, al1 = py + pw * t_, al2 = pz + pw * u_;
                acc[0] += cur.m[0] * al0 + cur.m[3] * al1 + cur.m[6] * al2;
                acc[1] += cur.m[1] * al0 + cur.m[4] * al1 + cur.m[7] * al2;
                acc[2] += cur.m[2] * al0 + cur.m[5] * al1 + cur.m[8] * al2;
            }
            if (i >= 0) {
                if (has_srp) {
                    acc[0] += fsrp[0] / mass; acc[1] += fsrp[1] / mass; acc[2] += fsrp[2] / mass;
                }
                KB(i, 0) = ys[3]; KB(i, 1) = ys[4]; KB(i, 2) = ys[5];
                KB(i, 3) = acc[0]; KB(i, 4) = acc[1]; KB(i, 5) = acc[2];
            }
            cur = nxt;
        }
        if (!done) n_evals += stages;

        // ---- next state and error estimate (instance.rs:401-414).  d(Cr, Cd, prop mass)/dt = 0.
        double next[9], err[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) { next[e] = y[e]; err[e] = 0.0; }
        for (int i = 0; i < stages; ++i) {
            const double b_i = cfg->b[i];
            const double bd = cfg->bdiff[i];
            if (bd != 0.0 && !fixed) {
                const double ce = h * bd;
#pragma unroll
                for (int e = 0; e < 6; ++e) err[e] += ce * KB(i, e);
            }
            if (b_i != 0.0) {
                const double cb = h * b_i;
#pragma unroll
                for (int e = 0; e < 6; ++e) next[e] += cb * KB(i, e);
            }
        }

        if (!done) {
            bool accept = false;
            if (st_att != NYX_HIP_OK) {
                status = st_att;
                done = true;
            } else if (fixed) {
                det_step = step_size;
                accept = true;
            } else {
                det_error = error_estimate(cfg->error_ctrl, err, next, y);
                if (det_error <= cfg->tol || h <= cfg->min_step_s || attempts >= cfg->attempts) {
                    bool nan = false;
#pragma unroll
                    for (int e = 0; e < 9; ++e) nan = nan || (next[e] != next[e]);
                    if (nan) {
                        status = NYX_HIP_ERR_NAN;
                        done = true;
                    } else {
                        det_step = seconds_to_ns(h);
                        if (det_error < cfg->tol) {
                            const double prop = 0.9 * h * pow(cfg->tol / det_error, cfg->inv_order);
                            h = (fabs(prop) > fabs(cfg->max_step_s)) ? cfg->max_step_s * copysign(1.0, prop) : prop;
                        }
                        step_size = seconds_to_ns(h);
                        const int64_t ab = step_size < 0 ? -step_size : step_size;
                        if (ab < min_step_ns) step_size = (step_size < 0) ? -min_step_ns : min_step_ns;
                        accept = true;
                    }
                } else {
                    attempts += 1;
                    n_rej += 1;
                    const double prop = 0.9 * h * pow(cfg->tol / det_error, cfg->inv_order_m1);
                    h = (prop < cfg->min_step_s) ? cfg->min_step_s : prop;
                }
            }
            if (accept) {
                // single_step(): state.set(epoch + t, vec) with the Cr clamp, then finally()
                epoch += det_step;
#pragma unroll
                for (int e = 0; e < 9; ++e) y[e] = next[e];
                y[6] = clamp02(y[6]);
                n_acc += 1;
                det_attempts = attempts;
                if (y[8] < 0.0) { status = NYX_HIP_ERR_FUEL_EXHAUSTED; done = true; }
                if (is_final) {
                    step_size = prev_step;
                    fixed = prev_kind;
                    if (backprop) step_size = -step_size;
                    is_final = false;
                    done = true;
                }
                fresh = true;
            }
        }
    }

    // release the workers
    if (nw > 1) {
        if (lane == 0) ctl[0] = 1;
        __syncthreads();
    }

    if (valid) {
        bt.o_epoch_ns[gid] = epoch;
        bt.o_x[gid] = y[0]; bt.o_y[gid] = y[1]; bt.o_z[gid] = y[2];
        bt.o_vx[gid] = y[3]; bt.o_vy[gid] = y[4]; bt.o_vz[gid] = y[5];
        if (bt.o_cr) bt.o_cr[gid] = y[6];
        if (bt.o_cd) bt.o_cd[gid] = y[7];
        if (bt.o_mprop) bt.o_mprop[gid] = y[8];
        if (bt.o_mdry) bt.o_mdry[gid] = m_dry;
        if (bt.o_mextra) bt.o_mextra[gid] = m_extra;
        if (bt.o_asrp) bt.o_asrp[gid] = a_srp;
        if (bt.o_adrag) bt.o_adrag[gid] = bt.adrag ? bt.adrag[idx] : 0.0;
        if (bt.o_step) bt.o_step[gid] = step_size;
        if (bt.status) bt.status[gid] = status;
        if (bt.last_step_ns) bt.last_step_ns[gid] = det_step;
        if (bt.last_error) bt.last_error[gid] = det_error;
        if (bt.last_attempts) bt.last_attempts[gid] = det_attempts;
        if (bt.n_acc) bt.n_acc[gid] = n_acc;
        if (bt.n_rej) bt.n_rej[gid] = n_rej;
        if (bt.n_evals) bt.n_evals[gid] = n_evals;
    }
}

extern "C" size_t nyx_kernel_lds_bytes(int rec_doubles) {
    return (size_t)(DEV_MAX_STAGES * 6 * DEV_LANES + NIN * DEV_LANES + (DEV_MAX_WAVES - 1) * 4 * DEV_LANES + rec_doubles) * sizeof(double) + 64;
}

extern "C" hipError_t nyx_launch_propagate(const DevBatch &bt, const DevCfg *cfg, const HarmEntry *htab,
                                           const ColHdr *cols, const double *records, int n_waves, int rec_lds_doubles,
                                           hipStream_t stream) {
    const int64_t blocks = (bt.n + DEV_LANES - 1) / DEV_LANES;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(nyx_propagate_kernel, dim3((unsigned)blocks), dim3((unsigned)(n_waves * DEV_LANES)),
                       nyx_kernel_lds_bytes(rec_lds_doubles), stream, bt, cfg, htab, cols, records);
    return hipGetLastError();
}
