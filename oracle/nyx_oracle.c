/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see nyx_oracle.h).  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math
 * so that no FMA contraction changes the operation order of the reference.
 *
 * Every function cites the reference lines (under /root/reference/nyx-core/src)
 * it restates.  Nothing here is copied: the reference is Rust over nalgebra /
 * anise / hifitime; this is scalar C over plain arrays.
 */
#define _GNU_SOURCE
#include "nyx_oracle.h"
#include "rk_tableaux.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#define NV_MAX 90
#define MAX_STAGES 16

/* ------------------------------------------------------------------------- */
/* hifitime 4.3 (absent crate) — restated                                     */
/* ------------------------------------------------------------------------- */

static int g_ns_rounding = 0;
void nyx_oracle_set_ns_rounding(int32_t mode) { g_ns_rounding = mode; }

/* `f64 * Unit::Second -> Duration`: total_ns = q * 1e9, then `as i64`
 * (saturating truncation toward zero).  Call sites: instance.rs:447,464;
 * cosmic/mod.rs:102 (Epoch + f64). */
int64_t nyx_oracle_seconds_to_ns(double s) {
    double total = s * 1e9;
    if (total != total) return 0; /* NaN as i64 == 0 in Rust */
    if (g_ns_rounding == 1) total = round(total);
    if (total >= 9.2233720368547758e18) return INT64_MAX;
    if (total <= -9.2233720368547758e18) return INT64_MIN;
    return (int64_t)total;
}

/* `Duration::to_seconds()`: hifitime keeps (centuries: i16, nanoseconds: u64) with
 * nanoseconds always positive; seconds = whole + sub*1e-9, plus centuries*SPC. */
double nyx_oracle_ns_to_seconds(int64_t ns) {
    const int64_t NS_PER_CENTURY = 3155760000000000000LL;
    const double S_PER_CENTURY = 3155760000.0;
    int64_t centuries = 0;
    int64_t rem = ns;
    if (ns < 0) {
        /* euclidean split so that rem >= 0 */
        centuries = -((-ns + NS_PER_CENTURY - 1) / NS_PER_CENTURY);
        rem = ns - centuries * NS_PER_CENTURY;
    } else if (ns >= NS_PER_CENTURY) {
        centuries = ns / NS_PER_CENTURY;
        rem = ns - centuries * NS_PER_CENTURY;
    }
    int64_t whole = rem / 1000000000LL;
    int64_t sub = rem % 1000000000LL;
    if (centuries == 0) return (double)whole + (double)sub * 1e-9;
    return (double)centuries * S_PER_CENTURY + (double)whole + (double)sub * 1e-9;
}

/* f64::powi as LLVM expands it (binary method, LSB first): x^3 = x*(x*x), x^6 = x^2*x^4. */
static double powi_(double x, int n) {
    double res = 1.0;
    int have = 0;
    double sq = x;
    int v = n < 0 ? -n : n;
    while (v) {
        if (v & 1) {
            res = have ? res * sq : sq;
            have = 1;
        }
        sq = sq * sq;
        v >>= 1;
    }
    if (!have) res = 1.0;
    return n < 0 ? 1.0 / res : res;
}

static double norm3(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* ------------------------------------------------------------------------- */
/* anise 0.10 (absent crate) — restated from published definitions             */
/* ------------------------------------------------------------------------- */

/* SPK type 2 (Chebyshev position) evaluation with the Clenshaw recurrence, as in
 * NAIF CHBINT.  Replaces what almanac.transform() reads (orbital.rs:230-234). */
static int cheby_eval(const nyx_hip_cheby_segment_t *seg, double et_s, double *r3) {
    double rel = (et_s - seg->init_et_s) / seg->interval_s;
    long idx = (long)floor(rel);
    if (idx < 0 || idx > seg->n_records) return NYX_HIP_ERR_EPHEM_RANGE;
    if (idx == seg->n_records) {
        if (et_s > seg->init_et_s + seg->interval_s * (double)seg->n_records) return NYX_HIP_ERR_EPHEM_RANGE;
        idx = seg->n_records - 1; /* exactly the end of coverage */
    }
    const int nc = seg->n_coeffs;
    const double *rec = seg->records + (size_t)idx * (size_t)(2 + 3 * nc);
    const double t = (et_s - rec[0]) / rec[1];
    const double two_t = 2.0 * t;
    for (int c = 0; c < 3; ++c) {
        const double *cf = rec + 2 + c * nc;
        double w0 = 0.0, w1 = 0.0, w2;
        for (int j = nc - 1; j >= 1; --j) {
            w2 = w1;
            w1 = w0;
            w0 = cf[j] + (two_t * w1 - w2);
        }
        r3[c] = cf[0] + (t * w0 - w1);
    }
    return NYX_HIP_OK;
}

/* SPK type 2 with the derivative: the value as cheby_eval, the derivative by the companion recurrence of SPICE's CHBINT
 * (dW_j = 2 W_{j+1} + 2t dW_{j+1} - dW_{j+2}), scaled by 1 / radius.  anise's `chebyshev_eval` (absent) is a port of CHBINT: unpinned. */
static int cheby_eval_pv(const nyx_hip_cheby_segment_t *seg, double et_s, double *r3, double *v3) {
    double rel = (et_s - seg->init_et_s) / seg->interval_s;
    long idx = (long)floor(rel);
    if (idx < 0 || idx > seg->n_records) return NYX_HIP_ERR_EPHEM_RANGE;
    if (idx == seg->n_records) {
        if (et_s > seg->init_et_s + seg->interval_s * (double)seg->n_records) return NYX_HIP_ERR_EPHEM_RANGE;
        idx = seg->n_records - 1;
    }
    const int nc = seg->n_coeffs;
    const double *rec = seg->records + (size_t)idx * (size_t)(2 + 3 * nc);
    const double t = (et_s - rec[0]) / rec[1];
    const double two_t = 2.0 * t;
    for (int c = 0; c < 3; ++c) {
        const double *cf = rec + 2 + c * nc;
        double w0 = 0.0, w1 = 0.0, w2, d0 = 0.0, d1 = 0.0, d2;
        for (int j = nc - 1; j >= 1; --j) {
            w2 = w1; w1 = w0;
            w0 = cf[j] + (two_t * w1 - w2);
            d2 = d1; d1 = d0;
            d0 = (2.0 * w1 + two_t * d1) - d2;
        }
        r3[c] = cf[0] + (t * w0 - w1);
        v3[c] = ((w0 + t * d0) - d1) / rec[1];
    }
    return NYX_HIP_OK;
}

/* opts.integration_frame (instance.rs:117-142, 211-220): almanac.transform_to(orbit, frame, None) between two frames of one
 * orientation = the state of bodies[b] w.r.t. the integration centre added (dir = +1, into the integration frame) or
 * subtracted (dir = -1, back). */
static int frame_shift(const nyx_hip_config_t *cfg, int b, int64_t epoch_ns, double *y6, double dir) {
    const nyx_hip_body_t *body = &cfg->bodies[b];
    const double et = nyx_oracle_ns_to_seconds(epoch_ns);
    double r[3] = {0.0, 0.0, 0.0}, v[3] = {0.0, 0.0, 0.0};
    int st = NYX_HIP_OK;
    for (int k = 0; k < body->n_chain; ++k) {
        double p[3] = {0.0, 0.0, 0.0}, pv[3] = {0.0, 0.0, 0.0};
        const int s1 = cheby_eval_pv(&cfg->segments[body->chain_segment[k]], et, p, pv);
        if (s1) st = s1;
        const double sg = (double)body->chain_sign[k];
        for (int c = 0; c < 3; ++c) { r[c] = r[c] + sg * p[c]; v[c] = v[c] + sg * pv[c]; }
    }
    for (int c = 0; c < 3; ++c) { y6[c] = y6[c] + dir * r[c]; y6[3 + c] = y6[3 + c] + dir * v[c]; }
    return st;
}

/* Position of bodies[b] w.r.t. the integration centre = signed sum over its chain. */
static int body_position(const nyx_hip_config_t *cfg, int b, double et_s, double *r3) {
    const nyx_hip_body_t *body = &cfg->bodies[b];
    r3[0] = r3[1] = r3[2] = 0.0;
    for (int k = 0; k < body->n_chain; ++k) {
        double p[3];
        int st = cheby_eval(&cfg->segments[body->chain_segment[k]], et_s, p);
        if (st) return st;
        double sg = (double)body->chain_sign[k];
        for (int c = 0; c < 3; ++c) r3[c] = r3[c] + sg * p[c];
    }
    return NYX_HIP_OK;
}

void nyx_oracle_body_position(const nyx_hip_config_t *cfg, int32_t body, int64_t epoch_ns, double *r3, int32_t *status) {
    int st = body_position(cfg, body, nyx_oracle_ns_to_seconds(epoch_ns), r3);
    if (status) *status = st;
}

/* Body-fixed orientation (nyx_hip_rotation_t).  Replaces almanac.transform_to / almanac.rotate at
 * gravity_field.rs:150-154,258-265.  IAU kind: DCM(inertial -> body-fixed) = R3(W) R1(pi/2 - dec) R3(pi/2 + ra), angles
 * from the PCK polynomials (T centuries / d days past J2000 TDB) plus the trigonometric nutation-precession series
 * (SPICE TISBOD: alpha += a_k sin theta_k, delta += d_k cos theta_k, W += w_k sin theta_k).  Euler/Chebyshev kind (binary
 * PCK type 2, SPICE PCKE02 + EUL2M(w, delta, phi, 3, 1, 3)): three Chebyshev angles of the segment, R3(A3) R1(A2) R3(A1),
 * times the constant base rotation.  `w_rate` (optional): dW/dt (resp. dA3/dt) in rad/s, what the drag model's
 * velocity transform uses. */
static void r3r1r3(double a1, double a2, double a3, double m[3][3]) {
    const double c1 = cos(a1), s1 = sin(a1);
    const double c2 = cos(a2), s2 = sin(a2);
    const double c3 = cos(a3), s3 = sin(a3);
    /* R3(a3) * R1(a2) * R3(a1), multiplied out */
    m[0][0] = c3 * c1 - s3 * c2 * s1;
    m[0][1] = c3 * s1 + s3 * c2 * c1;
    m[0][2] = s3 * s2;
    m[1][0] = -s3 * c1 - c3 * c2 * s1;
    m[1][1] = -s3 * s1 + c3 * c2 * c1;
    m[1][2] = c3 * s2;
    m[2][0] = s2 * s1;
    m[2][1] = -s2 * c1;
    m[2][2] = c2;
}

static int rotation_dcm_rate(const nyx_hip_rotation_t *rot, const nyx_hip_cheby_segment_t *segs, double et_s, double m[3][3], double *w_rate) {
    const double DEG = M_PI / 180.0;
    if (rot->kind == NYX_HIP_ROT_EULER_CHEBY) {
        const nyx_hip_cheby_segment_t *seg = &segs[rot->euler_segment];
        double ang[3];
        int st = cheby_eval(seg, et_s, ang);
        if (st) return st;
        double e[3][3];
        r3r1r3(ang[0], ang[1], ang[2], e);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                m[i][j] = e[i][0] * rot->base_dcm[0 + j] + e[i][1] * rot->base_dcm[3 + j] + e[i][2] * rot->base_dcm[6 + j];
        if (w_rate) { /* derivative of the third angle's Chebyshev series: sum c_j T_j'(t) / radius, T_j' by its recurrence */
            const int nc = seg->n_coeffs;
            long idx = (long)floor((et_s - seg->init_et_s) / seg->interval_s);
            if (idx >= seg->n_records) idx = seg->n_records - 1;
            const double *rec = seg->records + (size_t)idx * (size_t)(2 + 3 * nc);
            const double t = (et_s - rec[0]) / rec[1];
            const double *cf = rec + 2 + 2 * nc;
            double tjm1 = 1.0, tj = t, djm1 = 0.0, dj = 1.0, acc = 0.0; /* T_0, T_1, T_0', T_1' */
            for (int j = 1; j < nc; ++j) {
                acc = acc + cf[j] * dj;
                const double tn = 2.0 * t * tj - tjm1;
                const double dn = 2.0 * tj + 2.0 * t * dj - djm1;
                tjm1 = tj; tj = tn; djm1 = dj; dj = dn;
            }
            *w_rate = acc / rec[1];
        }
        return NYX_HIP_OK;
    }
    const double d = et_s / 86400.0;
    const double T = et_s / (86400.0 * 36525.0);
    double ra = rot->ra_deg[0] + rot->ra_deg[1] * T + rot->ra_deg[2] * T * T;
    double dec = rot->dec_deg[0] + rot->dec_deg[1] * T + rot->dec_deg[2] * T * T;
    double w = rot->w_deg[0] + rot->w_deg[1] * d + rot->w_deg[2] * d * d;
    double wd = rot->w_deg[1] + 2.0 * rot->w_deg[2] * d; /* deg / day */
    for (int k = 0; k < rot->n_nut_prec; ++k) {
        const double th = (rot->nut_prec_angle_deg[k][0] + rot->nut_prec_angle_deg[k][1] * T) * DEG;
        const double sn = sin(th), cs = cos(th);
        ra = ra + rot->nut_prec_ra[k] * sn;
        dec = dec + rot->nut_prec_dec[k] * cs;
        w = w + rot->nut_prec_w[k] * sn;
        wd = wd + rot->nut_prec_w[k] * cs * (rot->nut_prec_angle_deg[k][1] * DEG / 36525.0);
    }
    r3r1r3(M_PI_2 + ra * DEG, M_PI_2 - dec * DEG, w * DEG, m);
    if (w_rate) *w_rate = wd * DEG / 86400.0;
    return NYX_HIP_OK;
}

static int rotation_dcm(const nyx_hip_rotation_t *rot, const nyx_hip_cheby_segment_t *segs, double et_s, double m[3][3]) {
    return rotation_dcm_rate(rot, segs, et_s, m, NULL);
}

int32_t nyx_oracle_rotation_dcm(const nyx_hip_rotation_t *rot, const nyx_hip_cheby_segment_t *segments, int64_t epoch_ns, double *dcm9, double *w_rate) {
    double m[3][3];
    int st = rotation_dcm_rate(rot, segments, nyx_oracle_ns_to_seconds(epoch_ns), m, w_rate);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) dcm9[3 * i + j] = m[i][j];
    return st;
}

/* Area of a circular segment of radius r cut at distance d from the centre. */
static double circ_seg_area(double r, double d) { return r * r * acos(d / r) - d * sqrt(r * r - d * d); }

/* anise Almanac::solar_eclipsing -> Occultation.factor(): fraction (0..1) of the
 * solar disk hidden by `front` as seen from the spacecraft; apparent-disk overlap.
 * r_eb = observer w.r.t. eclipsing body, r_ls = Sun w.r.t. observer. */
static double occultation_pct(double r_back_km, double r_front_km, const double *r_eb, const double *r_ls) {
    const double n_ls = norm3(r_ls), n_eb = norm3(r_eb);
    const double ls_p = (r_back_km >= n_ls) ? r_back_km : asin(r_back_km / n_ls);
    const double fo_p = (r_front_km >= n_eb) ? r_front_km : asin(r_front_km / n_eb);
    const double dot = r_ls[0] * r_eb[0] + r_ls[1] * r_eb[1] + r_ls[2] * r_eb[2];
    const double d_p = acos(-dot / (n_eb * n_ls));
    double pct;
    if (d_p - ls_p > fo_p) {
        pct = 0.0; /* Sun fully visible */
    } else if (fo_p > d_p + ls_p) {
        pct = 100.0; /* umbra */
    } else if (fabs(ls_p - fo_p) < d_p && d_p < ls_p + fo_p) {
        /* penumbra: asymmetric lens of two overlapping disks */
        const double d1 = (d_p * d_p - ls_p * ls_p + fo_p * fo_p) / (2.0 * d_p);
        const double d2 = (d_p * d_p + ls_p * ls_p - fo_p * fo_p) / (2.0 * d_p);
        const double shadow = circ_seg_area(fo_p, d1) + circ_seg_area(ls_p, d2);
        if (shadow != shadow) {
            pct = 100.0;
        } else {
            const double nominal = M_PI * (ls_p * ls_p);
            pct = 100.0 * shadow / nominal;
        }
    } else {
        pct = 100.0 * (fo_p * fo_p) / (ls_p * ls_p); /* annular */
    }
    return pct; /* Occultation.percentage; .factor() = percentage / 100 */
}

/* ------------------------------------------------------------------------- */
/* Prepared (per-config) tables: GravityField::new, gravity_field.rs:52-132    */
/* ------------------------------------------------------------------------- */

typedef struct {
    int deg, ord, ld; /* ld = deg+3: leading dimension of a_nm (column-major like DMatrix) */
    int ld2;          /* deg+2: leading dimension of b,c,vr01,vr11 */
    double *a_diag_tmpl; /* a_nm template with diagonal set, ld*ld */
    double *b, *c, *vr01, *vr11;
    double *cfull, *sfull; /* (deg+1)^2 column-major copies of C,S */
} grav_tables_t;

static void grav_tables_free(grav_tables_t *t) {
    free(t->a_diag_tmpl); free(t->b); free(t->c); free(t->vr01); free(t->vr11); free(t->cfull); free(t->sfull);
    memset(t, 0, sizeof *t);
}

static void grav_tables_build(const nyx_hip_gravity_field_t *g, grav_tables_t *t) {
    const int np2 = g->degree + 2;
    t->deg = g->degree; t->ord = g->order; t->ld = np2 + 1; t->ld2 = np2;
    t->a_diag_tmpl = calloc((size_t)t->ld * t->ld, sizeof(double));
    t->b = calloc((size_t)np2 * np2, sizeof(double));
    t->c = calloc((size_t)np2 * np2, sizeof(double));
    t->vr01 = calloc((size_t)np2 * np2, sizeof(double));
    t->vr11 = calloc((size_t)np2 * np2, sizeof(double));
#define A_(n, m) t->a_diag_tmpl[(n) + (size_t)(m) * t->ld]
    A_(0, 0) = 1.0;
    for (int n = 1; n <= np2; ++n) { /* :61-66 */
        double nf = (double)n;
        A_(n, n) = sqrt(1.0 + 1.0 / (2.0 * nf)) * A_(n - 1, n - 1);
    }
#undef A_
    for (int n = 0; n < np2; ++n) { /* :69-92 */
        for (int m = 0; m < np2; ++m) {
            double nf = (double)n, mf = (double)m;
            size_t i = (size_t)n + (size_t)m * np2;
            t->c[i] = sqrt(((2.0 * nf + 1.0) * (nf + mf - 1.0) * (nf - mf - 1.0)) / ((nf - mf) * (nf + mf) * (2.0 * nf - 3.0)));
            t->b[i] = sqrt(((2.0 * nf + 1.0) * (2.0 * nf - 1.0)) / ((nf + mf) * (nf - mf)));
            t->vr01[i] = sqrt((nf - mf) * (nf + mf + 1.0));
            t->vr11[i] = sqrt(((2.0 * nf + 1.0) * (nf + mf + 2.0) * (nf + mf + 1.0)) / (2.0 * nf + 3.0));
            if (m == 0) {
                t->vr01[i] /= sqrt(2.0);
                t->vr11[i] /= sqrt(2.0);
            }
        }
    }
    const int n1 = g->degree + 1;
    t->cfull = calloc((size_t)n1 * n1, sizeof(double));
    t->sfull = calloc((size_t)n1 * n1, sizeof(double));
    for (int n = 0; n <= g->degree; ++n)
        for (int m = 0; m <= n; ++m) {
            t->cfull[n + (size_t)m * n1] = g->c_nm[(size_t)n * (n + 1) / 2 + m];
            t->sfull[n + (size_t)m * n1] = g->s_nm[(size_t)n * (n + 1) / 2 + m];
        }
}

typedef struct {
    const nyx_hip_config_t *cfg;
    grav_tables_t gt, gt2; /* gt2: a second GravityField of the same accel_models list (config.gravity2) */
    int has_grav, has_grav2;
} prepared_t;

static void prepared_init(prepared_t *p, const nyx_hip_config_t *cfg) {
    memset(p, 0, sizeof *p);
    p->cfg = cfg;
    if (cfg->gravity) {
        grav_tables_build(cfg->gravity, &p->gt);
        p->has_grav = 1;
    }
    if (cfg->gravity2) {
        grav_tables_build(cfg->gravity2, &p->gt2);
        p->has_grav2 = 1;
    }
}
static void prepared_free(prepared_t *p) {
    if (p->has_grav) grav_tables_free(&p->gt);
    if (p->has_grav2) grav_tables_free(&p->gt2);
}

/* ------------------------------------------------------------------------- */
/* GravityField::eom, gravity_field.rs:148-268                                 */
/* ------------------------------------------------------------------------- */

static int gravity_eom(const nyx_hip_gravity_field_t *g, const nyx_hip_cheby_segment_t *segs, const grav_tables_t *t, double et_s, const double *r_in,
                       double *acc, double *a_work, double *rm_work, double *im_work) {
    double dcm[3][3];
    { int st = rotation_dcm(&g->rotation, segs, et_s, dcm); if (st) return st; }
    /* almanac.transform_to(osc, frame): same centre, rotate the position (:150-154) */
    double rb[3];
    for (int i = 0; i < 3; ++i) rb[i] = dcm[i][0] * r_in[0] + dcm[i][1] * r_in[1] + dcm[i][2] * r_in[2];

    const double r_ = norm3(rb);
    const double s_ = rb[0] / r_, t_ = rb[1] / r_, u_ = rb[2] / r_;
    const int N = t->deg, M = t->ord, ld = t->ld, ld2 = t->ld2;
    double *a = a_work;
    memcpy(a, t->a_diag_tmpl, sizeof(double) * (size_t)ld * ld); /* `self.a_nm.clone()` :165 */
#define A_(n, m) a[(n) + (size_t)(m) * ld]
#define T2(tab, n, m) t->tab[(n) + (size_t)(m) * ld2]
    A_(1, 0) = u_ * sqrt(3.0);
    for (int n = 1; n <= N + 1; ++n) {
        double nf = (double)n;
        A_(n + 1, n) = sqrt(2.0 * nf + 3.0) * u_ * A_(n, n);
    }
    for (int m = 0; m <= M + 1; ++m)
        for (int n = m + 2; n <= N + 1; ++n) A_(n, m) = u_ * T2(b, n, m) * A_(n - 1, m) - T2(c, n, m) * A_(n - 2, m);

    const int mm = N < M ? N : M;
    rm_work[0] = 1.0;
    im_work[0] = 0.0;
    for (int m = 1; m <= mm; ++m) {
        rm_work[m] = s_ * rm_work[m - 1] - t_ * im_work[m - 1];
        im_work[m] = s_ * im_work[m - 1] + t_ * rm_work[m - 1];
    }
    const double re = g->eq_radius_km, mu = g->mu_km3_s2;
    const double rho = re / r_;
    double rho_np1 = mu / r_ * rho;
    double ax = 0.0, ay = 0.0, az = 0.0, aw = 0.0;
    const double SQ2 = sqrt(2.0);
    const int n1 = N + 1;
    for (int n = 1; n <= N; ++n) {
        double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
        rho_np1 *= rho;
        const int mtop = n < M ? n : M;
        for (int m = 0; m <= mtop; ++m) {
            const double cv = t->cfull[n + (size_t)m * n1], sv = t->sfull[n + (size_t)m * n1];
            const double d_ = (cv * rm_work[m] + sv * im_work[m]) * SQ2;
            const double e_ = (m == 0) ? 0.0 : (cv * rm_work[m - 1] + sv * im_work[m - 1]) * SQ2;
            const double f_ = (m == 0) ? 0.0 : (sv * rm_work[m - 1] - cv * im_work[m - 1]) * SQ2;
            sx += (double)m * A_(n, m) * e_;
            sy += (double)m * A_(n, m) * f_;
            sz += T2(vr01, n, m) * A_(n, m + 1) * d_;
            sw -= T2(vr11, n, m) * A_(n + 1, m + 1) * d_;
        }
        const double rr = rho_np1 / re;
        ax += rr * sx; ay += rr * sy; az += rr * sz; aw += rr * sw;
    }
#undef A_
#undef T2
    const double al[3] = {ax + aw * s_, ay + aw * t_, az + aw * u_};
    /* dcm.rot_mat (fixed -> inertial) = transpose (:258-267) */
    for (int i = 0; i < 3; ++i) acc[i] = dcm[0][i] * al[0] + dcm[1][i] * al[1] + dcm[2][i] * al[2];
    return NYX_HIP_OK;
}

void nyx_oracle_gravity_accel(const nyx_hip_gravity_field_t *g, int64_t epoch_ns, const double *r3, double *a3) {
    grav_tables_t t;
    grav_tables_build(g, &t);
    double *a = malloc(sizeof(double) * (size_t)t.ld * t.ld);
    double *rm = malloc(sizeof(double) * (size_t)(t.deg + 2)), *im = malloc(sizeof(double) * (size_t)(t.deg + 2));
    (void)gravity_eom(g, NULL, &t, nyx_oracle_ns_to_seconds(epoch_ns), r3, a3, a, rm, im);  /* (IAU-oriented fields only) */
    free(a); free(rm); free(im);
    grav_tables_free(&t);
}

/* ------------------------------------------------------------------------- */
/* Forward-mode duals (value + 3 position partials), stand-in for hyperdual    */
/* OHyperdual<f64, 7> whose slots 1..3 carry d/dx, d/dy, d/dz.                 */
/* ------------------------------------------------------------------------- */

typedef struct { double v, d[3]; } d3;
static d3 d3c(double v) { d3 r = {v, {0, 0, 0}}; return r; }
static d3 d3add(d3 a, d3 b) { d3 r = {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; return r; }
static d3 d3sub(d3 a, d3 b) { d3 r = {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; return r; }
static d3 d3mul(d3 a, d3 b) {
    d3 r = {a.v * b.v, {a.v * b.d[0] + a.d[0] * b.v, a.v * b.d[1] + a.d[1] * b.v, a.v * b.d[2] + a.d[2] * b.v}};
    return r;
}
static d3 d3scale(d3 a, double s) { d3 r = {a.v * s, {a.d[0] * s, a.d[1] * s, a.d[2] * s}}; return r; }
static d3 d3div(d3 a, d3 b) {
    /* hyperdual 1.5 Div: real = a/b, dual_i = (a_i*b - a*b_i) / (b*b) */
    double dd = b.v * b.v;
    d3 r = {a.v / b.v, {(a.d[0] * b.v - a.v * b.d[0]) / dd, (a.d[1] * b.v - a.v * b.d[1]) / dd,
                        (a.d[2] * b.v - a.v * b.d[2]) / dd}};
    return r;
}
static d3 d3divs(d3 a, double s) { d3 r = {a.v / s, {a.d[0] / s, a.d[1] / s, a.d[2] / s}}; return r; }
static d3 d3sqrt(d3 a) {
    double s = sqrt(a.v);
    double h = 0.5 / s;
    d3 r = {s, {a.d[0] * h, a.d[1] * h, a.d[2] * h}};
    return r;
}
static d3 d3powi(d3 a, int n) {
    double p = powi_(a.v, n - 1);
    double f = (double)n * p;
    d3 r = {p * a.v, {a.d[0] * f, a.d[1] * f, a.d[2] * f}};
    return r;
}
static d3 d3norm(const d3 *v) {
    return d3sqrt(d3add(d3add(d3mul(v[0], v[0]), d3mul(v[1], v[1])), d3mul(v[2], v[2])));
}

/* GravityField::gradient, gravity_field.rs:273-431: same recursion on duals seeded
 * in the body-fixed frame; returns accel (inertial) and dcm * grad_local * dcm^T. */
static int gravity_gradient(const nyx_hip_gravity_field_t *g, const nyx_hip_cheby_segment_t *segs, const grav_tables_t *t, double et_s,
                            const double *r_in, double *acc, double grad[3][3]) {
    double dcm[3][3];
    { int st = rotation_dcm(&g->rotation, segs, et_s, dcm); if (st) return st; }
    double rb[3];
    for (int i = 0; i < 3; ++i) rb[i] = dcm[i][0] * r_in[0] + dcm[i][1] * r_in[1] + dcm[i][2] * r_in[2];
    d3 rad[3];
    for (int i = 0; i < 3; ++i) { rad[i] = d3c(rb[i]); rad[i].d[i] = 1.0; }
    const d3 r_ = d3norm(rad);
    const d3 s_ = d3div(rad[0], r_), t_ = d3div(rad[1], r_), u_ = d3div(rad[2], r_);
    const int N = t->deg, M = t->ord, ld = t->ld, ld2 = t->ld2;
    d3 *a = calloc((size_t)ld * ld, sizeof(d3));
#define A_(n, m) a[(n) + (size_t)(m) * ld]
#define T2(tab, n, m) t->tab[(n) + (size_t)(m) * ld2]
    for (int i = 0; i <= N + 1; ++i) A_(i, i) = d3c(t->a_diag_tmpl[i + (size_t)i * ld]);
    A_(1, 0) = d3scale(u_, sqrt(3.0));
    for (int n = 1; n <= N + 1; ++n) A_(n + 1, n) = d3mul(d3mul(d3c(sqrt(2.0 * (double)n + 3.0)), u_), A_(n, n));
    for (int m = 0; m <= M + 1; ++m)
        for (int n = m + 2; n <= N + 1; ++n)
            A_(n, m) = d3sub(d3mul(d3mul(u_, d3c(T2(b, n, m))), A_(n - 1, m)), d3mul(d3c(T2(c, n, m)), A_(n - 2, m)));
    const int mm = N < M ? N : M;
    d3 *rm = calloc((size_t)mm + 2, sizeof(d3)), *im = calloc((size_t)mm + 2, sizeof(d3));
    rm[0] = d3c(1.0); im[0] = d3c(0.0);
    for (int m = 1; m <= mm; ++m) {
        rm[m] = d3sub(d3mul(s_, rm[m - 1]), d3mul(t_, im[m - 1]));
        im[m] = d3add(d3mul(s_, im[m - 1]), d3mul(t_, rm[m - 1]));
    }
    const d3 re = d3c(g->eq_radius_km);
    const d3 rho = d3div(re, r_);
    d3 rho_np1 = d3mul(d3div(d3c(g->mu_km3_s2), r_), rho);
    d3 a0 = d3c(0), a1 = d3c(0), a2 = d3c(0), a3 = d3c(0);
    const d3 sq2 = d3c(sqrt(2.0));
    const int n1 = N + 1;
    for (int n = 1; n <= N; ++n) {
        d3 s0 = d3c(0), s1 = d3c(0), s2 = d3c(0), s3 = d3c(0);
        rho_np1 = d3mul(rho_np1, rho);
        const int mtop = n < M ? n : M;
        for (int m = 0; m <= mtop; ++m) {
            const d3 cv = d3c(t->cfull[n + (size_t)m * n1]), sv = d3c(t->sfull[n + (size_t)m * n1]);
            const d3 dd = d3mul(d3add(d3mul(cv, rm[m]), d3mul(sv, im[m])), sq2);
            const d3 ee = (m == 0) ? d3c(0) : d3mul(d3add(d3mul(cv, rm[m - 1]), d3mul(sv, im[m - 1])), sq2);
            const d3 ff = (m == 0) ? d3c(0) : d3mul(d3sub(d3mul(sv, rm[m - 1]), d3mul(cv, im[m - 1])), sq2);
            s0 = d3add(s0, d3mul(d3mul(d3c((double)m), A_(n, m)), ee));
            s1 = d3add(s1, d3mul(d3mul(d3c((double)m), A_(n, m)), ff));
            s2 = d3add(s2, d3mul(d3mul(d3c(T2(vr01, n, m)), A_(n, m + 1)), dd));
            s3 = d3add(s3, d3mul(d3mul(d3c(T2(vr11, n, m)), A_(n + 1, m + 1)), dd));
        }
        const d3 rr = d3div(rho_np1, re);
        a0 = d3add(a0, d3mul(rr, s0));
        a1 = d3add(a1, d3mul(rr, s1));
        a2 = d3add(a2, d3mul(rr, s2));
        a3 = d3sub(a3, d3mul(rr, s3));
    }
#undef A_
#undef T2
    const d3 al[3] = {d3add(a0, d3mul(a3, s_)), d3add(a1, d3mul(a3, t_)), d3add(a2, d3mul(a3, u_))};
    for (int i = 0; i < 3; ++i) acc[i] = dcm[0][i] * al[0].v + dcm[1][i] * al[1].v + dcm[2][i] * al[2].v;
    /* grad = dcm^T * G_local * dcm  (with dcm = inertial->fixed) */
    double tmp[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) tmp[i][j] = dcm[0][i] * al[0].d[j] + dcm[1][i] * al[1].d[j] + dcm[2][i] * al[2].d[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) grad[i][j] = tmp[i][0] * dcm[0][j] + tmp[i][1] * dcm[1][j] + tmp[i][2] * dcm[2][j];
    free(a); free(rm); free(im);
    return NYX_HIP_OK;
}

/* ------------------------------------------------------------------------- */
/* SolidTides, dynamics/solid_tides.rs:65-559                                   */
/* ------------------------------------------------------------------------- */

/* TidalPerturber::compute_pert (:74-175) for every perturber (accumulate_deltas :221-235) */
static int tides_deltas(const nyx_hip_config_t *cfg, double et_s, const double dcm[3][3], double dc[4][4], double ds[4][4]) {
    const nyx_hip_solid_tides_t *td = cfg->tides;
    memset(dc, 0, 16 * sizeof(double));
    memset(ds, 0, 16 * sizeof(double));
    for (int j = 0; j < td->n_perturbers; ++j) {
        const int b = td->perturber_body[j];
        double pin[3], rb[3];
        int st = body_position(cfg, b, et_s, pin);
        if (st) return st;
        /* almanac.transform(perturber, tidal frame): origin of the perturber seen in the body-fixed frame */
        for (int i = 0; i < 3; ++i) rb[i] = dcm[i][0] * pin[0] + dcm[i][1] * pin[1] + dcm[i][2] * pin[2];
        const double r_body = norm3(rb);
        const double s_body = rb[0] / r_body, t_body = rb[1] / r_body, u_body = rb[2] / r_body;
        const double sin_phi = u_body;
        const double cos_phi = sqrt(fmax(1.0 - powi_(sin_phi, 2), 0.0));
        const double cos_lambda = cos_phi > 1e-12 ? s_body / cos_phi : 1.0;
        const double sin_lambda = cos_phi > 1e-12 ? t_body / cos_phi : 0.0;
        const double p2[3] = {0.5 * (3.0 * powi_(sin_phi, 2) - 1.0) * sqrt(5.0), 3.0 * sin_phi * cos_phi * sqrt(5.0 / 3.0),
                              3.0 * powi_(cos_phi, 2) * sqrt(5.0 / 12.0)};
        const double p3[4] = {0.5 * (5.0 * powi_(sin_phi, 3) - 3.0 * sin_phi) * sqrt(7.0),
                              1.5 * (5.0 * powi_(sin_phi, 2) - 1.0) * cos_phi * sqrt(7.0 / 6.0),
                              15.0 * sin_phi * powi_(cos_phi, 2) * sqrt(7.0 / 60.0), 15.0 * powi_(cos_phi, 3) * sqrt(7.0 / 360.0)};
        const double gm_ratio = cfg->bodies[b].mu_km3_s2 / td->mu_km3_s2;
        const double r_ratio = td->eq_radius_km / r_body;
        const int top = td->compute_degree_3[j] ? 3 : 2;
        for (int n = 2; n <= top; ++n) {
            const double kn = n == 2 ? td->k2 : td->k3;
            const double common = kn / (2.0 * (double)n + 1.0) * gm_ratio * powi_(r_ratio, n + 1);
            for (int m = 0; m <= n; ++m) {
                const double p_nm = n == 2 ? p2[m] : p3[m];
                double cos_ml, sin_ml;
                switch (m) {
                case 0: cos_ml = 1.0; sin_ml = 0.0; break;
                case 1: cos_ml = cos_lambda; sin_ml = sin_lambda; break;
                case 2: cos_ml = powi_(cos_lambda, 2) - powi_(sin_lambda, 2); sin_ml = 2.0 * sin_lambda * cos_lambda; break;
                default:
                    cos_ml = cos_lambda * (powi_(cos_lambda, 2) - 3.0 * powi_(sin_lambda, 2));
                    sin_ml = sin_lambda * (3.0 * powi_(cos_lambda, 2) - powi_(sin_lambda, 2));
                }
                dc[n][m] += common * p_nm * cos_ml;
                ds[n][m] += common * p_nm * sin_ml;
            }
        }
    }
    return NYX_HIP_OK;
}

static double tide_b(int n, int m) {
    const double nf = n, mf = m;
    return sqrt(((2.0 * nf + 1.0) * (2.0 * nf - 1.0)) / ((nf + mf) * (nf - mf)));
}
static double tide_c(int n, int m) {
    const double nf = n, mf = m;
    return sqrt(((2.0 * nf + 1.0) * (nf + mf - 1.0) * (nf - mf - 1.0)) / ((nf - mf) * (nf + mf) * (2.0 * nf - 3.0)));
}
static double tide_vr01(int n, int m) {
    double v = sqrt(((double)n - (double)m) * ((double)n + (double)m + 1.0));
    if (m == 0) v /= sqrt(2.0);
    return v;
}
static double tide_vr11(int n, int m) {
    const double nf = n, mf = m;
    double v = sqrt(((2.0 * nf + 1.0) * (nf + mf + 2.0) * (nf + mf + 1.0)) / (2.0 * nf + 3.0));
    if (m == 0) v /= sqrt(2.0);
    return v;
}

/* SolidTides::eom (:238-385) */
static int tides_eom(const nyx_hip_config_t *cfg, double et_s, const double *r_in, double *acc) {
    const nyx_hip_solid_tides_t *td = cfg->tides;
    double dcm[3][3], dc[4][4], ds[4][4];
    { int st0 = rotation_dcm(&td->rotation, cfg->segments, et_s, dcm); if (st0) return st0; }
    int st = tides_deltas(cfg, et_s, dcm, dc, ds);
    if (st) return st;
    double rb[3];
    for (int i = 0; i < 3; ++i) rb[i] = dcm[i][0] * r_in[0] + dcm[i][1] * r_in[1] + dcm[i][2] * r_in[2];
    const double r_ = norm3(rb);
    const double s_ = rb[0] / r_, t_ = rb[1] / r_, u_ = rb[2] / r_;
    double a[6][6];
    memset(a, 0, sizeof a);
    a[0][0] = 1.0;
    for (int n = 1; n <= 4; ++n) a[n][n] = sqrt(1.0 + 1.0 / (2.0 * (double)n)) * a[n - 1][n - 1];
    a[1][0] = u_ * sqrt(3.0);
    for (int n = 1; n <= 4; ++n) a[n + 1][n] = sqrt(2.0 * (double)n + 3.0) * u_ * a[n][n];
    for (int m = 0; m <= 3; ++m)
        for (int n = m + 2; n <= 4; ++n) a[n][m] = u_ * tide_b(n, m) * a[n - 1][m] - tide_c(n, m) * a[n - 2][m];
    double rm[4], im[4];
    rm[0] = 1.0; im[0] = 0.0;
    for (int m = 1; m <= 3; ++m) {
        rm[m] = s_ * rm[m - 1] - t_ * im[m - 1];
        im[m] = s_ * im[m - 1] + t_ * rm[m - 1];
    }
    const double rho = td->eq_radius_km / r_;
    double rho_np1 = td->mu_km3_s2 / r_ * rho;
    double a4[4] = {0, 0, 0, 0};
    const double sqrt2 = sqrt(2.0);
    for (int n = 1; n <= 3; ++n) {
        rho_np1 *= rho;
        if (n < 2) continue;
        double sum[4] = {0, 0, 0, 0};
        for (int m = 0; m <= n; ++m) {
            const double c_val = dc[n][m], s_val = ds[n][m];
            const double d_ = (c_val * rm[m] + s_val * im[m]) * sqrt2;
            const double e_ = m == 0 ? 0.0 : (c_val * rm[m - 1] + s_val * im[m - 1]) * sqrt2;
            const double f_ = m == 0 ? 0.0 : (s_val * rm[m - 1] - c_val * im[m - 1]) * sqrt2;
            sum[0] += (double)m * a[n][m] * e_;
            sum[1] += (double)m * a[n][m] * f_;
            sum[2] += tide_vr01(n, m) * a[n][m + 1] * d_;
            sum[3] -= tide_vr11(n, m) * a[n + 1][m + 1] * d_;
        }
        const double k = rho_np1 / td->eq_radius_km;
        for (int q = 0; q < 4; ++q) a4[q] += k * sum[q];
    }
    const double al[3] = {a4[0] + a4[3] * s_, a4[1] + a4[3] * t_, a4[2] + a4[3] * u_};
    /* almanac.rotate(tidal frame -> integration frame) = dcm^T */
    for (int i = 0; i < 3; ++i) acc[i] = dcm[0][i] * al[0] + dcm[1][i] * al[1] + dcm[2][i] * al[2];
    return NYX_HIP_OK;
}

/* SolidTides::gradient (:387-559): the same evaluation on position duals, deltas constant */
static int tides_gradient(const nyx_hip_config_t *cfg, double et_s, const double *r_in, double *acc, double grad[3][3]) {
    const nyx_hip_solid_tides_t *td = cfg->tides;
    double dcm[3][3], dc[4][4], ds[4][4];
    { int st0 = rotation_dcm(&td->rotation, cfg->segments, et_s, dcm); if (st0) return st0; }
    int st = tides_deltas(cfg, et_s, dcm, dc, ds);
    if (st) return st;
    double rb[3];
    for (int i = 0; i < 3; ++i) rb[i] = dcm[i][0] * r_in[0] + dcm[i][1] * r_in[1] + dcm[i][2] * r_in[2];
    d3 rad[3];
    for (int i = 0; i < 3; ++i) { rad[i] = d3c(rb[i]); rad[i].d[i] = 1.0; }
    const d3 r_ = d3norm(rad);
    const d3 s_ = d3div(rad[0], r_), t_ = d3div(rad[1], r_), u_ = d3div(rad[2], r_);
    d3 a[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) a[i][j] = d3c(0.0);
    a[0][0] = d3c(1.0);
    for (int n = 1; n <= 4; ++n) a[n][n] = d3mul(d3c(sqrt(1.0 + 1.0 / (2.0 * (double)n))), a[n - 1][n - 1]);
    a[1][0] = d3mul(u_, d3c(sqrt(3.0)));
    for (int n = 1; n <= 4; ++n) a[n + 1][n] = d3mul(d3mul(d3c(sqrt(2.0 * (double)n + 3.0)), u_), a[n][n]);
    for (int m = 0; m <= 3; ++m)
        for (int n = m + 2; n <= 4; ++n)
            a[n][m] = d3sub(d3mul(d3mul(u_, d3c(tide_b(n, m))), a[n - 1][m]), d3mul(d3c(tide_c(n, m)), a[n - 2][m]));
    d3 rm[4], im[4];
    rm[0] = d3c(1.0); im[0] = d3c(0.0);
    for (int m = 1; m <= 3; ++m) {
        rm[m] = d3sub(d3mul(s_, rm[m - 1]), d3mul(t_, im[m - 1]));
        im[m] = d3add(d3mul(s_, im[m - 1]), d3mul(t_, rm[m - 1]));
    }
    const d3 eq = d3c(td->eq_radius_km);
    const d3 rho = d3div(eq, r_);
    d3 rho_np1 = d3mul(d3div(d3c(td->mu_km3_s2), r_), rho);
    d3 a0 = d3c(0.0), a1 = d3c(0.0), a2 = d3c(0.0), a3 = d3c(0.0);
    const d3 sqrt2 = d3c(sqrt(2.0));
    for (int n = 1; n <= 3; ++n) {
        rho_np1 = d3mul(rho_np1, rho);
        if (n < 2) continue;
        d3 s0 = d3c(0.0), s1 = d3c(0.0), s2 = d3c(0.0), s3 = d3c(0.0);
        for (int m = 0; m <= n; ++m) {
            const d3 c_val = d3c(dc[n][m]), s_val = d3c(ds[n][m]);
            const d3 d_ = d3mul(d3add(d3mul(c_val, rm[m]), d3mul(s_val, im[m])), sqrt2);
            const d3 e_ = m == 0 ? d3c(0.0) : d3mul(d3add(d3mul(c_val, rm[m - 1]), d3mul(s_val, im[m - 1])), sqrt2);
            const d3 f_ = m == 0 ? d3c(0.0) : d3mul(d3sub(d3mul(s_val, rm[m - 1]), d3mul(c_val, im[m - 1])), sqrt2);
            s0 = d3add(s0, d3mul(d3mul(d3c((double)m), a[n][m]), e_));
            s1 = d3add(s1, d3mul(d3mul(d3c((double)m), a[n][m]), f_));
            s2 = d3add(s2, d3mul(d3mul(d3c(tide_vr01(n, m)), a[n][m + 1]), d_));
            s3 = d3add(s3, d3mul(d3mul(d3c(tide_vr11(n, m)), a[n + 1][m + 1]), d_));
        }
        const d3 rr = d3div(rho_np1, eq);
        a0 = d3add(a0, d3mul(rr, s0));
        a1 = d3add(a1, d3mul(rr, s1));
        a2 = d3add(a2, d3mul(rr, s2));
        a3 = d3sub(a3, d3mul(rr, s3));
    }
    const d3 al[3] = {d3add(a0, d3mul(a3, s_)), d3add(a1, d3mul(a3, t_)), d3add(a2, d3mul(a3, u_))};
    for (int i = 0; i < 3; ++i) acc[i] = dcm[0][i] * al[0].v + dcm[1][i] * al[1].v + dcm[2][i] * al[2].v;
    double tmp[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) tmp[i][j] = dcm[0][i] * al[0].d[j] + dcm[1][i] * al[1].d[j] + dcm[2][i] * al[2].d[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) grad[i][j] = tmp[i][0] * dcm[0][j] + tmp[i][1] * dcm[1][j] + tmp[i][2] * dcm[2][j];
    return NYX_HIP_OK;
}

int32_t nyx_oracle_tides_accel(const nyx_hip_config_t *cfg, int64_t epoch_ns, const double *r3, double *a3, double *grad9_rowmajor,
                               double *dc16, double *ds16) {
    if (!cfg || !cfg->tides) return NYX_HIP_RC_BAD_ARG;
    const double et = nyx_oracle_ns_to_seconds(epoch_ns);
    int st = tides_eom(cfg, et, r3, a3);
    if (st) return st;
    if (grad9_rowmajor) {
        double acc[3], g[3][3];
        st = tides_gradient(cfg, et, r3, acc, g);
        if (st) return st;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) grad9_rowmajor[3 * i + j] = g[i][j];
    }
    if (dc16 && ds16) {
        double dcm[3][3], dc[4][4], ds[4][4];
        (void)rotation_dcm(&cfg->tides->rotation, cfg->segments, et, dcm);
        st = tides_deltas(cfg, et, dcm, dc, ds);
        memcpy(dc16, dc, sizeof dc);
        memcpy(ds16, ds, sizeof ds);
    }
    return st;
}

/* ------------------------------------------------------------------------- */
/* Osculating spacecraft rebuilt from the state vector                         */
/* (Spacecraft::set, cosmic/spacecraft.rs:477-497)                             */
/* ------------------------------------------------------------------------- */

typedef struct {
    double dry, extra, srp_area, drag_area;
} sc_const_t;

typedef struct {
    double *a_work, *rm, *im; /* gravity scratch */
} scratch_t;

static double clamp02(double x) { return x < 0.0 ? 0.0 : (x > 2.0 ? 2.0 : x); } /* f64::clamp */

/* PointMasses::eom, orbital.rs:214-247 */
static int point_masses_eom(const nyx_hip_config_t *cfg, double et_s, const double *r, double *acc) {
    acc[0] = acc[1] = acc[2] = 0.0;
    for (int k = 0; k < cfg->n_point_masses; ++k) {
        const int b = cfg->point_mass_body[k];
        const nyx_hip_body_t *body = &cfg->bodies[b];
        if (body->n_chain == 0) continue; /* the central body itself (:219-222) */
        double r_ij[3];
        int st = body_position(cfg, b, et_s, r_ij);
        if (st) return st;
        const double r_ij3 = powi_(norm3(r_ij), 3);
        double r_j[3] = {r[0] - r_ij[0], r[1] - r_ij[1], r[2] - r_ij[2]};
        const double r_j3 = powi_(norm3(r_j), 3);
        const double nmu = -body->mu_km3_s2;
        for (int c = 0; c < 3; ++c) acc[c] += nmu * (r_j[c] / r_j3 + r_ij[c] / r_ij3);
    }
    return NYX_HIP_OK;
}

/* ShadowModel::compute (cosmic/eclipse.rs:69-83): max occultation over shadow bodies. */
static int shadow_factor(const nyx_hip_config_t *cfg, double et_s, const double *r, const double *r_sun_wrt_center,
                         double *factor) {
    const nyx_hip_srp_t *srp = cfg->srp;
    const double sun_radius = cfg->bodies[srp->sun_body].mean_radius_km;
    double best_pct = 0.0;
    for (int k = 0; k < srp->n_shadow_bodies; ++k) {
        const int b = srp->shadow_body[k];
        double pb[3] = {0, 0, 0};
        if (cfg->bodies[b].n_chain > 0) {
            int st = body_position(cfg, b, et_s, pb);
            if (st) return st;
        }
        double r_eb[3] = {r[0] - pb[0], r[1] - pb[1], r[2] - pb[2]};
        double r_ls[3] = {r_sun_wrt_center[0] - r[0], r_sun_wrt_center[1] - r[1], r_sun_wrt_center[2] - r[2]};
        double pct = occultation_pct(sun_radius, cfg->bodies[b].mean_radius_km, r_eb, r_ls);
        if (pct > best_pct) best_pct = pct;
    }
    *factor = best_pct / 100.0;
    return NYX_HIP_OK;
}

double nyx_oracle_occultation_factor(const nyx_hip_config_t *cfg, int32_t eclipsing_body, int32_t sun_body,
                                     int64_t epoch_ns, const double *r3, int32_t *status) {
    double et = nyx_oracle_ns_to_seconds(epoch_ns);
    double ps[3], pb[3] = {0, 0, 0};
    int st = body_position(cfg, sun_body, et, ps);
    if (!st && cfg->bodies[eclipsing_body].n_chain > 0) st = body_position(cfg, eclipsing_body, et, pb);
    if (status) *status = st;
    if (st) return 0.0;
    double r_eb[3] = {r3[0] - pb[0], r3[1] - pb[1], r3[2] - pb[2]};
    double r_ls[3] = {ps[0] - r3[0], ps[1] - r3[1], ps[2] - r3[2]};
    return occultation_pct(cfg->bodies[sun_body].mean_radius_km, cfg->bodies[eclipsing_body].mean_radius_km, r_eb, r_ls) / 100.0;
}

#define AU_KM 149597870.700 /* cosmic/mod.rs:183 */

/* SolarPressure::eom, solarpressure.rs:135-165 (force in kg km/s^2, caller divides by mass) */
static int srp_eom(const nyx_hip_config_t *cfg, double et_s, const double *r, double cr, double area, double *force) {
    const nyx_hip_srp_t *srp = cfg->srp;
    double ps[3];
    int st = body_position(cfg, srp->sun_body, et_s, ps);
    if (st) return st;
    double r_sun[3] = {r[0] - ps[0], r[1] - ps[1], r[2] - ps[2]}; /* s/c as seen from the Sun */
    const double n = norm3(r_sun);
    double unit[3] = {r_sun[0] / n, r_sun[1] / n, r_sun[2] / n};
    double occult;
    st = shadow_factor(cfg, et_s, r, ps, &occult);
    if (st) return st;
    const double k = fabs(occult - 1.0);
    const double r_au = n / AU_KM;
    const double c_m_s = cfg->speed_of_light_km_s * 1e3; /* cosmic/mod.rs:179 */
    const double flux = (k * srp->phi_w_m2 / c_m_s) * powi_(1.0 / r_au, 2);
    const double scal = 1e-3 * cr * area * flux;
    for (int c = 0; c < 3; ++c) force[c] = scal * unit[c];
    return NYX_HIP_OK;
}

/* SolarPressure::gradient, solarpressure.rs:167-232: position Jacobian with k frozen;
 * row 3 = d force / d Cr. */
static int srp_gradient(const nyx_hip_config_t *cfg, double et_s, const double *r, double cr, double area,
                        double *force, double grad[4][3]) {
    const nyx_hip_srp_t *srp = cfg->srp;
    double ps[3];
    int st = body_position(cfg, srp->sun_body, et_s, ps);
    if (st) return st;
    d3 rs[3];
    for (int i = 0; i < 3; ++i) { rs[i] = d3c(r[i] - ps[i]); rs[i].d[i] = 1.0; }
    const d3 n = d3norm(rs);
    d3 unit[3] = {d3div(rs[0], n), d3div(rs[1], n), d3div(rs[2], n)};
    double occult;
    st = shadow_factor(cfg, et_s, r, ps, &occult);
    if (st) return st;
    const double k = fabs(occult - 1.0);
    const d3 r_au = d3divs(n, AU_KM);
    const d3 inv = d3div(d3c(1.0), r_au);
    const d3 inv2 = d3powi(inv, 2);
    const double c_m_s = cfg->speed_of_light_km_s * 1e3;
    const d3 flux = d3mul(d3c(k * srp->phi_w_m2 / c_m_s), inv2);
    const d3 scal = d3c(1e-3 * cr * area);
    for (int i = 0; i < 3; ++i) {
        d3 f = d3mul(d3mul(scal, flux), unit[i]);
        force[i] = f.v;
        for (int j = 0; j < 3; ++j) grad[i][j] = f.d[j];
    }
    double fr[3];
    st = srp_eom(cfg, et_s, r, cr, area, fr);
    if (st) return st;
    for (int j = 0; j < 3; ++j) grad[3][j] = fr[j] / cr;
    return NYX_HIP_OK;
}

/* Drag::eom, drag.rs:181-284 — including the reference's unit/frame quirks. */
static int drag_eom(const nyx_hip_config_t *cfg, double et_s, const double *r, const double *v, double cd, double area,
                    double *force) {
    const nyx_hip_drag_t *dg = cfg->drag;
    double m[3][3], wdot;
    { int st = rotation_dcm_rate(&dg->rotation, cfg->segments, et_s, m, &wdot); if (st) return st; }
    /* transform_to(orbit, drag frame): r' = R r, v' = R v + dR/dt r.  dR/dt from the
     * twist rate only would be an approximation; ANISE differentiates all three
     * angles.  We use a symmetric finite difference-free analytic form: dR/dt = -[w]x R
     * with w = W_dot * z_body (pole drift neglected: |ra_dot|,|dec_dot| ~ 1e-13 rad/s). */
    double rb[3], vb[3];
    for (int i = 0; i < 3; ++i) {
        rb[i] = m[i][0] * r[0] + m[i][1] * r[1] + m[i][2] * r[2];
        vb[i] = m[i][0] * v[0] + m[i][1] * v[1] + m[i][2] * v[2];
    }
    /* v_fixed = R v - w x (R r), w = (0,0,wdot) in the fixed frame */
    vb[0] = vb[0] + wdot * rb[1];
    vb[1] = vb[1] - wdot * rb[0];
    double rho;
    const double rmag = norm3(rb);
    if (dg->density == NYX_HIP_RHO_CONSTANT) {
        rho = dg->rho0;
        const double vn = norm3(vb);
        const double s = -0.5 * 1e3 * rho * cd * area * vn;
        for (int c = 0; c < 3; ++c) force[c] = s * vb[c];
        return NYX_HIP_OK;
    } else if (dg->density == NYX_HIP_RHO_EXPONENTIAL) {
        rho = dg->rho0 * exp(-(rmag - (dg->r0 + dg->eq_radius_km)) / dg->ref_alt_m);
    } else {
        const double alt = rmag - dg->eq_radius_km;
        if (alt > dg->max_alt_m / 1000.0) {
            rho = pow(10.0, (-7e-5) * alt - 14.464);
        } else {
            const double sc = (alt - 526.8000) / 292.8563;
            const double lg = 0.34047 * powi_(sc, 6) - 0.5889 * powi_(sc, 5) - 0.5269 * powi_(sc, 4) +
                              1.0036 * powi_(sc, 3) + 0.60713 * powi_(sc, 2) - 2.3024 * sc - 12.575;
            rho = pow(10.0, lg);
        }
    }
    /* velocity_integr_frame - osc_drag_frame.velocity (drag.rs:223-230): the
     * round-trip back to the integration frame returns the inertial velocity. */
    double vel[3] = {v[0] - vb[0], v[1] - vb[1], v[2] - vb[2]};
    const double vn = norm3(vel);
    const double s = -0.5 * 1e3 * rho * cd * area * vn;
    for (int c = 0; c < 3; ++c) force[c] = s * vel[c];
    return NYX_HIP_OK;
}

/* ------------------------------------------------------------------------- */
/* SpacecraftDynamics::eom / dual_eom                                          */
/* ------------------------------------------------------------------------- */

static int any_force_model(const nyx_hip_config_t *cfg) { return cfg->srp != NULL || cfg->drag != NULL; }

/* OrbitalDynamics::dual_eom (orbital.rs:116-172) + SpacecraftDynamics::dual_eom (spacecraft.rs:312-363).
 * grad is 9x9 column-major (nalgebra storage). */
static int dual_eom(const prepared_t *p, double et_s, const double *y9, const sc_const_t *sc, double *fx, double *grad) {
    const nyx_hip_config_t *cfg = p->cfg;
    memset(fx, 0, 9 * sizeof(double));
    memset(grad, 0, 81 * sizeof(double));
#define G(i, j) grad[(i) + 9 * (j)]
    const double *r = y9, *v = y9 + 3;
    /* two-body via duals: radius * (-mu / rmag^3) */
    d3 rad[3];
    for (int i = 0; i < 3; ++i) { rad[i] = d3c(r[i]); rad[i].d[i] = 1.0; }
    const d3 rmag = d3norm(rad);
    const d3 fac = d3div(d3c(-cfg->central_mu_km3_s2), d3powi(rmag, 3));
    for (int i = 0; i < 3; ++i) {
        fx[i] = v[i];
        G(i, i + 3) = 1.0;
        d3 a = d3mul(rad[i], fac);
        fx[i + 3] = a.v;
        for (int j = 0; j < 3; ++j) G(i + 3, j) = a.d[j];
    }
    /* accel models: PointMasses::gradient orbital.rs:249-308 */
    if (cfg->n_point_masses > 0) {
        double acc[3] = {0, 0, 0}, g3[3][3] = {{0}};
        for (int k = 0; k < cfg->n_point_masses; ++k) {
            const int b = cfg->point_mass_body[k];
            const nyx_hip_body_t *body = &cfg->bodies[b];
            if (body->n_chain == 0) continue;
            double pij[3];
            int st = body_position(cfg, b, et_s, pij);
            if (st) return st;
            d3 r_ij[3] = {d3c(pij[0]), d3c(pij[1]), d3c(pij[2])};
            const d3 r_ij3 = d3powi(d3norm(r_ij), 3);
            d3 r_j[3];
            for (int i = 0; i < 3; ++i) { r_j[i] = d3c(r[i] - pij[i]); r_j[i].d[i] = 1.0; }
            const d3 r_j3 = d3powi(d3norm(r_j), 3);
            const d3 gm = d3c(-body->mu_km3_s2);
            for (int i = 0; i < 3; ++i) {
                d3 t = d3mul(d3add(d3div(r_j[i], r_j3), d3div(r_ij[i], r_ij3)), gm);
                acc[i] += t.v;
                for (int j = 0; j < 3; ++j) g3[i][j] += t.d[j];
            }
        }
        for (int i = 0; i < 3; ++i) {
            fx[i + 3] += acc[i];
            for (int j = 0; j < 3; ++j) G(i + 3, j) += g3[i][j];
        }
    }
    for (int which = 0; which < 2; ++which) { /* every GravityField of accel_models in turn (orbital.rs:116-172 dual_eom -> gradient of each model) */
        const nyx_hip_gravity_field_t *g = which == 0 ? cfg->gravity : cfg->gravity2;
        if (!g) continue;
        double acc[3], g3[3][3], rg[3] = {r[0], r[1], r[2]};
        /* GravityField::gradient, gravity_field.rs:279-283: almanac.transform_to(osc, grav_data.frame) translates to the field's body and
         * rotates; the duals are seeded on THAT radius (:285), so the partials are with respect to r - r_body(t), i.e. to r: the
         * translation carries none.  The gradient is rotated back, dcm * grad_local * dcm^T (:429), like the acceleration. */
        const int gb = g->offset_body - 1;
        if (gb >= 0 && gb < cfg->n_bodies && cfg->bodies[gb].n_chain > 0) {
            double pb[3];
            int st = body_position(cfg, gb, et_s, pb);
            if (st) return st;
            for (int c = 0; c < 3; ++c) rg[c] = r[c] - pb[c];
        }
        { int st = gravity_gradient(g, cfg->segments, which == 0 ? &p->gt : &p->gt2, et_s, rg, acc, g3); if (st) return st; }
        for (int i = 0; i < 3; ++i) {
            fx[i + 3] += acc[i];
            for (int j = 0; j < 3; ++j) G(i + 3, j) += g3[i][j];
        }
    }
    if (cfg->tides) {
        double acc[3], g3[3][3];
        int st = tides_gradient(cfg, et_s, r, acc, g3);
        if (st) return st;
        for (int i = 0; i < 3; ++i) {
            fx[i + 3] += acc[i];
            for (int j = 0; j < 3; ++j) G(i + 3, j) += g3[i][j];
        }
    }
    /* force models (spacecraft.rs:339-360) */
    const double cr = clamp02(y9[6]);
    const double total_mass = sc->dry + y9[8] + sc->extra;
    if (cfg->srp) {
        double f[3], g4[4][3];
        int st = srp_gradient(cfg, et_s, r, cr, sc->srp_area, f, g4);
        if (st) return st;
        for (int i = 0; i < 3; ++i) {
            fx[i + 3] += f[i] / total_mass;
            for (int j = 0; j < 3; ++j) G(i + 3, j) += g4[i][j] / total_mass;
        }
        if (cfg->srp->estimate)
            for (int j = 0; j < 3; ++j) G(j + 3, 6) += g4[3][j] / total_mass;
    }
    if (cfg->drag) return NYX_HIP_ERR_UNSUPPORTED; /* PartialsUndefined, drag.rs:286-294 */
#undef G
    return NYX_HIP_OK;
}

/* SpacecraftDynamics::eom, spacecraft.rs:191-310 (no guidance law on this path). */
static int sc_eom(const prepared_t *p, int64_t ctx_epoch_ns, double dt_s, const double *y, int nv, const double *ctx_stm,
                  const sc_const_t *sc, scratch_t *w, double *dy) {
    const nyx_hip_config_t *cfg = p->cfg;
    /* ctx.set_with_delta_seconds: epoch + dt rounded to the ns grain (cosmic/mod.rs:94-104) */
    const int64_t epoch_ns = ctx_epoch_ns + nyx_oracle_seconds_to_ns(dt_s);
    const double et_s = nyx_oracle_ns_to_seconds(epoch_ns);
    const double cr = clamp02(y[6]);
    const double cd = y[7];
    const double mass = sc->dry + y[8] + sc->extra;
    if (any_force_model(cfg) && !(mass > 0.0)) return NYX_HIP_ERR_MASSLESS;
    for (int i = 0; i < nv; ++i) dy[i] = 0.0;

    if (ctx_stm) {
        double fx[9], grad[81];
        int st = dual_eom(p, et_s, y, sc, fx, grad);
        if (st) return st;
        for (int i = 0; i < 9; ++i) dy[i] = fx[i];
        if (cfg->flags & NYX_HIP_FLAG_STM_TEXTBOOK) {
            /* The textbook form SURVEY 8a-11 asks to expose beside the reference's: d(Phi)/dt = A(t) Phi, with Phi the STM of the STAGE
             * vector (y[9..90), column-major) - the variational equations integrated by the same tableau as the state.  No reference
             * line computes this (the reference right-multiplies the step-start STM, below); the definition here is the test's:
             * entry (i, j) = sum_k grad[i][k] * Phi_stage[k][j], k ascending from 0.0. */
            const double *ph = y + 9;
            for (int j = 0; j < 9; ++j)
                for (int i = 0; i < 9; ++i) {
                    double s = 0.0;
                    for (int k = 0; k < 9; ++k) s += grad[i + 9 * k] * ph[k + 9 * j];
                    dy[9 + i + 9 * j] = s;
                }
            return NYX_HIP_OK;
        }
        /* stm_dt = ctx.stm * grad  (spacecraft.rs:214), both column-major */
        for (int j = 0; j < 9; ++j)
            for (int i = 0; i < 9; ++i) {
                double s = 0.0;
                for (int k = 0; k < 9; ++k) s += ctx_stm[i + 9 * k] * grad[k + 9 * j];
                dy[9 + i + 9 * j] = s;
            }
        return NYX_HIP_OK;
    }

    const double *r = y, *v = y + 3;
    /* OrbitalDynamics::eom orbital.rs:80-114 */
    const double rmag = norm3(r);
    const double f = -cfg->central_mu_km3_s2 / powi_(rmag, 3);
    dy[0] = v[0]; dy[1] = v[1]; dy[2] = v[2];
    dy[3] = f * r[0]; dy[4] = f * r[1]; dy[5] = f * r[2];
    if (cfg->n_point_masses > 0) {
        double a[3];
        int st = point_masses_eom(cfg, et_s, r, a);
        if (st) return st;
        for (int c = 0; c < 3; ++c) dy[3 + c] += a[c];
    }
    for (int which = 0; which < 2; ++which) { /* accel_models is a list: each GravityField in turn (orbital.rs:100-108) */
        const nyx_hip_gravity_field_t *g = which == 0 ? cfg->gravity : cfg->gravity2;
        if (!g) continue;
        double a[3], rg[3] = {r[0], r[1], r[2]};
        /* almanac.transform_to(osc, grav_data.frame) (gravity_field.rs:150-154) translates to the field's body before it rotates: the
         * field of another body than the integration centre is evaluated at r - r_body(t); the acceleration is only rotated back
         * (:258-265, "no center change needed, it's just a vector") */
        const int gb = g->offset_body - 1;
        if (gb >= 0 && gb < cfg->n_bodies && cfg->bodies[gb].n_chain > 0) {
            double pb[3];
            int st = body_position(cfg, gb, et_s, pb);
            if (st) return st;
            for (int c = 0; c < 3; ++c) rg[c] = r[c] - pb[c];
        }
        { int st = gravity_eom(g, cfg->segments, which == 0 ? &p->gt : &p->gt2, et_s, rg, a, w->a_work, w->rm, w->im); if (st) return st; }
        for (int c = 0; c < 3; ++c) dy[3 + c] += a[c];
    }
    if (cfg->tides) { /* third accel model of Dynamics::build (dynamics/sequence/config.rs:105-118) */
        double a[3];
        int st = tides_eom(cfg, et_s, r, a);
        if (st) return st;
        for (int c = 0; c < 3; ++c) dy[3 + c] += a[c];
    }
    if (cfg->srp) {
        double fo[3];
        int st = srp_eom(cfg, et_s, r, cr, sc->srp_area, fo);
        if (st) return st;
        for (int c = 0; c < 3; ++c) dy[3 + c] += fo[c] / mass;
    }
    if (cfg->drag) {
        double fo[3];
        int st = drag_eom(cfg, et_s, r, v, cd, sc->drag_area, fo);
        if (st) return st;
        for (int c = 0; c < 3; ++c) dy[3 + c] += fo[c] / mass;
    }
    return NYX_HIP_OK;
}

static void scratch_init(scratch_t *w, const prepared_t *p) {
    memset(w, 0, sizeof *w);
    if (p->has_grav) {
        const size_t ld = (size_t)(p->has_grav2 && p->gt2.ld > p->gt.ld ? p->gt2.ld : p->gt.ld);
        const size_t dg = (size_t)(p->has_grav2 && p->gt2.deg > p->gt.deg ? p->gt2.deg : p->gt.deg);
        w->a_work = malloc(sizeof(double) * ld * ld);
        w->rm = malloc(sizeof(double) * (dg + 2));
        w->im = malloc(sizeof(double) * (dg + 2));
    }
}
static void scratch_free(scratch_t *w) { free(w->a_work); free(w->rm); free(w->im); }

int32_t nyx_oracle_eom(const nyx_hip_config_t *cfg, int64_t ctx_epoch_ns, double delta_t_s, const double *y,
                       const double *ctx_stm, double dry, double extra, double srp_area, double drag_area, double *dydt) {
    prepared_t p;
    prepared_init(&p, cfg);
    scratch_t w;
    scratch_init(&w, &p);
    sc_const_t sc = {dry, extra, srp_area, drag_area};
    int st = sc_eom(&p, ctx_epoch_ns, delta_t_s, y, ctx_stm ? 90 : 9, ctx_stm, &sc, &w, dydt);
    scratch_free(&w);
    prepared_free(&p);
    return st;
}

int32_t nyx_oracle_dual_eom(const nyx_hip_config_t *cfg, int64_t epoch_ns, const double *y9, double dry, double extra,
                            double srp_area, double *fx9, double *grad81) {
    prepared_t p;
    prepared_init(&p, cfg);
    sc_const_t sc = {dry, extra, srp_area, 0.0};
    int st = dual_eom(&p, nyx_oracle_ns_to_seconds(epoch_ns), y9, &sc, fx9, grad81);
    prepared_free(&p);
    return st;
}

/* ------------------------------------------------------------------------- */
/* ErrorControl::estimate, error_ctrl.rs:79-229                                */
/* ------------------------------------------------------------------------- */

/* nalgebra's dot for long vectors: 8 independent accumulators, folded as
 * (0+4),(1+5),(2+6),(3+7), then the tail. */
static double nalgebra_norm(const double *x, int n) {
    double res = 0.0, acc[8] = {0};
    int i = 0;
    while (n - i >= 8) {
        for (int l = 0; l < 8; ++l) acc[l] += x[i + l] * x[i + l];
        i += 8;
    }
    res += acc[0] + acc[4];
    res += acc[1] + acc[5];
    res += acc[2] + acc[6];
    res += acc[3] + acc[7];
    for (; i < n; ++i) res += x[i] * x[i];
    return sqrt(res);
}

static double rss_step3(const double *e, const double *cand, const double *cur) {
    double dlt[3] = {cand[0] - cur[0], cand[1] - cur[1], cand[2] - cur[2]};
    double mag = norm3(dlt), err = norm3(e);
    return (mag > sqrt(0.1)) ? err / mag : err;
}
static double rss_state3(const double *e, const double *cand, const double *cur) {
    double sm[3] = {cand[0] + cur[0], cand[1] + cur[1], cand[2] + cur[2]};
    double mag = 0.5 * norm3(sm), err = norm3(e);
    return (mag > 0.1) ? err / mag : err;
}

double nyx_oracle_error_estimate(int32_t ec, int32_t nv, const double *e, const double *cand, const double *cur) {
    double tmp[NV_MAX];
    switch (ec) {
    case NYX_HIP_RSS_CARTESIAN_STATE: {
        double a = rss_state3(e, cand, cur), b = rss_state3(e + 3, cand + 3, cur + 3);
        return fmax(a, b);
    }
    case NYX_HIP_RSS_CARTESIAN_STEP: {
        double a = rss_step3(e, cand, cur), b = rss_step3(e + 3, cand + 3, cur + 3);
        return fmax(a, b);
    }
    case NYX_HIP_RSS_STATE: {
        for (int i = 0; i < nv; ++i) tmp[i] = cand[i] + cur[i];
        double mag = 0.5 * nalgebra_norm(tmp, nv), err = nalgebra_norm(e, nv);
        return (mag > 0.1) ? err / mag : err;
    }
    case NYX_HIP_RSS_STEP: {
        for (int i = 0; i < nv; ++i) tmp[i] = cand[i] - cur[i];
        double mag = nalgebra_norm(tmp, nv), err = nalgebra_norm(e, nv);
        return (mag > sqrt(0.1)) ? err / mag : err;
    }
    case NYX_HIP_LARGEST_ERROR: {
        double mx = 0.0;
        for (int i = 0; i < nv; ++i) {
            double dl = cand[i] - cur[i];
            double er = (dl > 0.1) ? fabs(e[i] / dl) : fabs(e[i]);
            if (er > mx) mx = er;
        }
        return mx;
    }
    case NYX_HIP_LARGEST_STATE: {
        double mag = 0.0, err = 0.0;
        for (int i = 0; i < nv; ++i) {
            mag += 0.5 * fabs(cand[i] + cur[i]);
            err += fabs(e[i]);
        }
        return (mag > 0.1) ? err / mag : err;
    }
    case NYX_HIP_LARGEST_STEP: {
        double mag = 0.0, err = 0.0;
        for (int i = 0; i < nv; ++i) {
            mag += fabs(cand[i] - cur[i]);
            err += fabs(e[i]);
        }
        return (mag > 0.1) ? err / mag : err;
    }
    }
    return NAN;
}

/* ------------------------------------------------------------------------- */
/* PropInstance: propagate / single_step / derive, instance.rs:87-493          */
/* ------------------------------------------------------------------------- */

typedef struct {
    const prepared_t *p;
    const oracle_tableau_t *tab;
    nyx_hip_integ_opts_t opts;
    int nv;       /* 9 or 90 */
    int has_stm;
    /* Spacecraft */
    int64_t epoch_ns;
    double y[NV_MAX];
    sc_const_t sc;
    /* PropInstance */
    int64_t step_size_ns;
    int fixed_step;
    int64_t det_step_ns;
    double det_error;
    int det_attempts;
    int64_t n_acc, n_rej, n_evals;
    double k[MAX_STAGES][NV_MAX];
    scratch_t w;
    /* dense output (for_duration_with_traj, instance.rs:297-326) */
    const nyx_hip_traj_t *traj;
    int64_t traj_n, traj_i;
    /* stop condition (propagators/event.rs:108-146) */
    const nyx_hip_event_t *ev;
    double ev_prev;
    int ev_count, ev_found;
} inst_t;

/* ---------------------------------------------------------------------------------------------
 * Event evaluation.  `Event`, `ScalarExpr::evaluate` and `brent_solver` belong to anise 0.10.2 (feature
 * "analysis"), absent from /root/reference: PARITY UNPINNED.  Restated from the classical definitions
 * (Vallado RV2COE for the elements, the `roots` crate's Brent as used by earlier Nyx releases) and anchored
 * on what nyx-core itself fixes: the crossing rule (propagators/event.rs:124-141) and the assertions of
 * tests/propagation/stopcond.rs (third apoapsis inside [2P, 3P], |180 - TA| < 1e-6 deg, ...).
 * --------------------------------------------------------------------------------------------- */
static int ev_is_angle(int scalar) { return scalar == NYX_HIP_EV_TRUE_ANOMALY_DEG || scalar == NYX_HIP_EV_LONGITUDE_DEG; }

/* Geodetic latitude (deg) and height (km) on the frame's ellipsoid: the classical iteration (Vallado, Algorithm 12);
 * anise's Orbit::latitude_deg / height_km (absent crate) - PARITY UNPINNED. */
static void ev_geodetic(double a, double f, const double *y, double *lat_deg, double *height_km) {
    const double e2 = f * (2.0 - f);
    const double r_delta = sqrt(y[0] * y[0] + y[1] * y[1]);
    double lat = atan2(y[2], r_delta), c = a;
    for (int it = 0; it < 20; ++it) {
        const double sl = sin(lat);
        c = a / sqrt(1.0 - e2 * sl * sl);
        const double nl = atan2(y[2] + c * e2 * sl, r_delta);
        const int done = fabs(nl - lat) < 1e-12;
        lat = nl;
        if (done) break;
    }
    *lat_deg = lat * (180.0 / M_PI);
    const double sl = sin(lat), cl = cos(lat);
    c = a / sqrt(1.0 - e2 * sl * sl);
    *height_km = (fabs(cl) > 1e-6) ? r_delta / cl - c : fabs(y[2]) / fabs(sl) - c * (1.0 - e2);
}

static double ev_scalar(const nyx_hip_event_t *ev, double mu, const double *y) {
    const int scalar = ev->scalar;
    const double *r = y, *v = y + 3;
    const double rmag = norm3(r), vmag = norm3(v);
    switch (scalar) {
    case NYX_HIP_EV_LONGITUDE_DEG: {
        const double deg = atan2(y[1], y[0]) * (180.0 / M_PI);
        return deg < 0.0 ? deg + 360.0 : deg;
    }
    case NYX_HIP_EV_DECLINATION_DEG: return asin(y[2] / rmag) * (180.0 / M_PI);
    case NYX_HIP_EV_LATITUDE_DEG: case NYX_HIP_EV_HEIGHT_KM: {
        double lat, h;
        ev_geodetic(ev->frame_eq_radius_km, ev->frame_flattening, y, &lat, &h);
        return scalar == NYX_HIP_EV_LATITUDE_DEG ? lat : h;
    }
    case NYX_HIP_EV_RMAG_KM: return rmag;
    case NYX_HIP_EV_VMAG_KM_S: return vmag;
    case NYX_HIP_EV_X_KM: case NYX_HIP_EV_Y_KM: case NYX_HIP_EV_Z_KM: return y[scalar - NYX_HIP_EV_X_KM];
    case NYX_HIP_EV_VX_KM_S: case NYX_HIP_EV_VY_KM_S: case NYX_HIP_EV_VZ_KM_S: return y[3 + scalar - NYX_HIP_EV_VX_KM_S];
    case NYX_HIP_EV_SMA_KM: {
        const double energy = vmag * vmag / 2.0 - mu / rmag;
        return -mu / (2.0 * energy);
    }
    default: break;
    }
    /* eccentricity vector: ((v^2 - mu/r) r - (r.v) v) / mu */
    const double rv = r[0] * v[0] + r[1] * v[1] + r[2] * v[2];
    const double k = vmag * vmag - mu / rmag;
    double e[3];
    for (int i = 0; i < 3; ++i) e[i] = (k * r[i] - rv * v[i]) / mu;
    const double ecc = norm3(e);
    if (scalar == NYX_HIP_EV_ECC) return ecc;
    /* true anomaly in [0, 360): atan2 of (sin, cos) projected on the orbit plane.  The textbook acos(e.r / (|e||r|)) loses
     * half the digits at the apsides (1e-6 deg), exactly where Event::apoapsis / periapsis need it. */
    const double h[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
    const double exr[3] = {e[1] * r[2] - e[2] * r[1], e[2] * r[0] - e[0] * r[2], e[0] * r[1] - e[1] * r[0]};
    const double sin_part = (exr[0] * h[0] + exr[1] * h[1] + exr[2] * h[2]) / norm3(h);
    const double cos_part = e[0] * r[0] + e[1] * r[1] + e[2] * r[2];
    const double deg = atan2(sin_part, cos_part) * (180.0 / 3.14159265358979323846);
    return deg < 0.0 ? deg + 360.0 : deg;
}

/* Event::eval for Condition::Equals: value - desired, wrapped to [-180, 180) for angles.  With an observer frame
 * (until_nth_event's event_frame, event.rs:104-117) the state is first expressed in that body-fixed frame:
 * almanac.transform_to(orbit, frame) for a frame of the same centre = rotation of position and velocity. */
static double ev_eval(const nyx_hip_event_t *ev, double mu, int64_t epoch_ns, const double *y_in) {
    double yf[6];
    const double *y = y_in;
    if (ev->has_frame) {
        double m[3][3], wdot;
        (void)rotation_dcm_rate(&ev->frame, NULL, nyx_oracle_ns_to_seconds(epoch_ns), m, &wdot);
        for (int i = 0; i < 3; ++i) {
            yf[i] = m[i][0] * y_in[0] + m[i][1] * y_in[1] + m[i][2] * y_in[2];
            yf[3 + i] = m[i][0] * y_in[3] + m[i][1] * y_in[4] + m[i][2] * y_in[5];
        }
        yf[3] = yf[3] + wdot * yf[1];
        yf[4] = yf[4] - wdot * yf[0];
        y = yf;
    }
    const double d = ev_scalar(ev, mu, y) - ev->desired;
    if (!ev_is_angle(ev->scalar)) return d;
    double w = fmod(d + 180.0, 360.0);
    if (w < 0.0) w += 360.0;
    return w - 180.0;
}

static double signum_(double x) { return x != x ? x : (signbit(x) ? -1.0 : 1.0); } /* f64::signum */

/* the `enough_crossings` closure (event.rs:108-146) */
static int ev_step(inst_t *s) {
    const double y_next = ev_eval(s->ev, s->p->cfg->central_mu_km3_s2, s->epoch_ns, s->y);
    const double delta = fabs(y_next - s->ev_prev);
    if (ev_is_angle(s->ev->scalar)) {
        if (signum_(s->ev_prev) != signum_(y_next) && delta < 180.0) s->ev_count += 1;
    } else if (s->ev_prev * y_next < 0.0) {
        s->ev_count += 1;
    }
    s->ev_prev = y_next;
    return s->ev_count >= s->ev->trigger;
}

static void traj_push(inst_t *s) {
    const nyx_hip_traj_t *t = s->traj;
    if (!t) return;
    const int64_t k = t->len[s->traj_i]; /* 0 = start state (len is reset by the caller) */
    if (k < t->capacity) {
        const int64_t at = k * s->traj_n + s->traj_i;
        t->epoch_ns[at] = s->epoch_ns;
        t->x_km[at] = s->y[0]; t->y_km[at] = s->y[1]; t->z_km[at] = s->y[2];
        t->vx_km_s[at] = s->y[3]; t->vy_km_s[at] = s->y[4]; t->vz_km_s[at] = s->y[5];
    }
    t->len[s->traj_i] = (int32_t)(k + 1);
}

static int64_t i64abs(int64_t x) { return x < 0 ? -x : x; }

/* derive(): one adaptive RK step (instance.rs:358-493).  Returns status; on success
 * writes the step actually taken (ns) and the new state vector. */
static int derive(inst_t *s, int64_t *step_taken_ns, double *next) {
    const oracle_tableau_t *tb = s->tab;
    const int nv = s->nv, stages = tb->stages;
    const double *y = s->y;
    const double *ctx_stm = s->has_stm ? s->y + 9 : NULL;
    const double min_step_s = nyx_oracle_ns_to_seconds(s->opts.min_step_ns);
    const double max_step_s = nyx_oracle_ns_to_seconds(s->opts.max_step_ns);
    s->det_attempts = 1;
    double h = nyx_oracle_ns_to_seconds(s->step_size_ns);
    double wi[NV_MAX], ys[NV_MAX], err[NV_MAX];
    for (;;) {
        int st = sc_eom(s->p, s->epoch_ns, 0.0, y, nv, ctx_stm, &s->sc, &s->w, s->k[0]);
        if (st) return st;
        int a_idx = 0;
        for (int i = 0; i < stages - 1; ++i) {
            double ci = 0.0;
            for (int e = 0; e < nv; ++e) wi[e] = 0.0;
            for (int j = 0; j <= i; ++j) {
                const double a_ij = tb->a[a_idx++];
                ci += a_ij;
                for (int e = 0; e < nv; ++e) wi[e] += a_ij * s->k[j][e];
            }
            for (int e = 0; e < nv; ++e) ys[e] = y[e] + h * wi[e];
            st = sc_eom(s->p, s->epoch_ns, ci * h, ys, nv, ctx_stm, &s->sc, &s->w, s->k[i + 1]);
            if (st) return st;
        }
        s->n_evals += stages;
        for (int e = 0; e < nv; ++e) { next[e] = y[e]; err[e] = 0.0; }
        for (int i = 0; i < stages; ++i) {
            const double b_i = tb->b[i];
            if (!s->fixed_step) {
                const double b_s = tb->b[i + stages];
                const double ce = h * (b_i - b_s);
                for (int e = 0; e < nv; ++e) err[e] += ce * s->k[i][e];
            }
            const double cb = h * b_i;
            for (int e = 0; e < nv; ++e) next[e] += cb * s->k[i][e];
        }
        if (s->fixed_step) {
            s->det_step_ns = s->step_size_ns;
            *step_taken_ns = s->det_step_ns;
            return NYX_HIP_OK;
        }
        s->det_error = nyx_oracle_error_estimate(s->opts.error_ctrl, nv, err, next, y);
        if (s->det_error <= s->opts.tolerance || h <= min_step_s || s->det_attempts >= s->opts.attempts) {
            for (int e = 0; e < nv; ++e)
                if (next[e] != next[e]) return NYX_HIP_ERR_NAN;
            s->det_step_ns = nyx_oracle_seconds_to_ns(h);
            if (s->det_error < s->opts.tolerance) {
                const double prop = 0.9 * h * pow(s->opts.tolerance / s->det_error, 1.0 / (double)tb->order);
                h = (fabs(prop) > fabs(max_step_s)) ? max_step_s * copysign(1.0, prop) : prop;
            }
            s->step_size_ns = nyx_oracle_seconds_to_ns(h);
            if (i64abs(s->step_size_ns) < s->opts.min_step_ns)
                s->step_size_ns = (s->step_size_ns < 0) ? -s->opts.min_step_ns : s->opts.min_step_ns;
            *step_taken_ns = s->det_step_ns;
            return NYX_HIP_OK;
        }
        s->det_attempts += 1;
        s->n_rej += 1;
        const double prop = 0.9 * h * pow(s->opts.tolerance / s->det_error, 1.0 / (double)(tb->order - 1));
        h = (prop < min_step_s) ? min_step_s : prop;
    }
}

/* dynamics.finally (spacecraft.rs:158-189) without guidance: prop mass check only. */
static int finally_(const inst_t *s) { return (s->y[8] < 0.0) ? NYX_HIP_ERR_FUEL_EXHAUSTED : NYX_HIP_OK; }

static int single_step(inst_t *s) {
    double next[NV_MAX];
    int64_t t;
    int st = derive(s, &t, next);
    if (st) return st;
    /* state.set(epoch + t, vec): Cr clamp (cosmic/spacecraft.rs:494) */
    s->epoch_ns += t;
    memcpy(s->y, next, sizeof(double) * (size_t)s->nv);
    s->y[6] = clamp02(s->y[6]);
    s->n_acc += 1;
    return finally_(s);
}

static int propagate(inst_t *s, int64_t duration_ns) {
    if (duration_ns == 0) return NYX_HIP_OK;
    const int64_t stop = s->epoch_ns + duration_ns;
    int st = finally_(s);
    if (st) return st;
    const int backprop = duration_ns < 0;
    if (backprop) s->step_size_ns = -s->step_size_ns;
    for (;;) {
        const int64_t epoch = s->epoch_ns;
        if ((!backprop && epoch + s->step_size_ns > stop) || (backprop && epoch + s->step_size_ns <= stop)) {
            if (stop == epoch) return NYX_HIP_OK;
            const int64_t prev_step = s->step_size_ns;
            const int prev_kind = s->fixed_step;
            s->step_size_ns = stop - epoch;
            s->fixed_step = 1;
            st = single_step(s);
            if (st) return st;
            traj_push(s); /* chan.send(self.state) (instance.rs:188-193) */
            s->step_size_ns = prev_step;
            s->fixed_step = prev_kind;
            if (backprop) s->step_size_ns = -s->step_size_ns;
            return NYX_HIP_OK;
        }
        st = single_step(s);
        if (st) return st;
        /* stop condition: the triggering state is NOT published (instance.rs:243-252) */
        if (s->ev && ev_step(s)) { s->ev_found = 1; return NYX_HIP_OK; }
        traj_push(s); /* chan.send(self.state) (instance.rs:254-259) */
    }
}

/* ------------------------------------------------------------------------- */
/* Batch driver (rayon par_iter analogue)                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    const prepared_t *p;
    const nyx_hip_states_t *in;
    nyx_hip_states_t *out;
    nyx_hip_step_stats_t *stats;
    int64_t duration_ns;
    atomic_long next;
    const nyx_hip_traj_t *traj;
} job_t;

static void inst_init(inst_t *s, const prepared_t *p, const nyx_hip_states_t *in, int64_t i) {
    const nyx_hip_config_t *cfg = p->cfg;
    s->p = p;
    s->tab = &ORC_TABLEAUX[cfg->opts.method];
    s->opts = cfg->opts;
    s->has_stm = (cfg->flags & NYX_HIP_FLAG_STM) && in->stm;
    s->nv = s->has_stm ? 90 : 9;
    s->epoch_ns = in->epoch_ns[i];
    memset(s->y, 0, sizeof s->y);
    s->y[0] = in->x_km[i]; s->y[1] = in->y_km[i]; s->y[2] = in->z_km[i];
    s->y[3] = in->vx_km_s[i]; s->y[4] = in->vy_km_s[i]; s->y[5] = in->vz_km_s[i];
    s->y[6] = in->cr ? in->cr[i] : 0.0;
    s->y[7] = in->cd ? in->cd[i] : 0.0;
    s->y[8] = in->prop_mass_kg ? in->prop_mass_kg[i] : 0.0;
    if (s->has_stm) memcpy(s->y + 9, in->stm + 81 * i, 81 * sizeof(double));
    s->sc.dry = in->dry_mass_kg ? in->dry_mass_kg[i] : 0.0;
    s->sc.extra = in->extra_mass_kg ? in->extra_mass_kg[i] : 0.0;
    s->sc.srp_area = in->srp_area_m2 ? in->srp_area_m2[i] : 0.0;
    s->sc.drag_area = in->drag_area_m2 ? in->drag_area_m2[i] : 0.0;
    s->step_size_ns = (in->step_ns && in->step_ns[i] != 0) ? in->step_ns[i] : cfg->opts.init_step_ns;
    s->fixed_step = cfg->opts.fixed_step;
    s->det_step_ns = cfg->opts.init_step_ns;
    s->det_error = 0.0;
    s->det_attempts = 1;
    s->n_acc = s->n_rej = s->n_evals = 0;
    s->traj = NULL; s->traj_n = in->n; s->traj_i = i;
    s->ev = NULL; s->ev_prev = 0.0; s->ev_count = 0; s->ev_found = 0;
}

static void inst_store(const inst_t *s, nyx_hip_states_t *o, nyx_hip_step_stats_t *t, int64_t i, int st) {
    o->epoch_ns[i] = s->epoch_ns;
    o->x_km[i] = s->y[0]; o->y_km[i] = s->y[1]; o->z_km[i] = s->y[2];
    o->vx_km_s[i] = s->y[3]; o->vy_km_s[i] = s->y[4]; o->vz_km_s[i] = s->y[5];
    if (o->cr) o->cr[i] = s->y[6];
    if (o->cd) o->cd[i] = s->y[7];
    if (o->prop_mass_kg) o->prop_mass_kg[i] = s->y[8];
    if (o->dry_mass_kg) o->dry_mass_kg[i] = s->sc.dry;
    if (o->extra_mass_kg) o->extra_mass_kg[i] = s->sc.extra;
    if (o->srp_area_m2) o->srp_area_m2[i] = s->sc.srp_area;
    if (o->drag_area_m2) o->drag_area_m2[i] = s->sc.drag_area;
    if (o->stm && s->has_stm) memcpy(o->stm + 81 * i, s->y + 9, 81 * sizeof(double));
    if (o->step_ns) o->step_ns[i] = s->step_size_ns;
    if (t) {
        if (t->status) t->status[i] = st;
        if (t->last_step_ns) t->last_step_ns[i] = s->det_step_ns;
        if (t->last_error) t->last_error[i] = s->det_error;
        if (t->last_attempts) t->last_attempts[i] = s->det_attempts;
        if (t->n_accepted) t->n_accepted[i] = s->n_acc;
        if (t->n_rejected) t->n_rejected[i] = s->n_rej;
        if (t->n_evals) t->n_evals[i] = s->n_evals;
    }
}

static void run_one(const job_t *jb, inst_t *s, int64_t i) {
    inst_init(s, jb->p, jb->in, i);
    s->traj = jb->traj;
    if (jb->traj) jb->traj->len[i] = 0;
    const nyx_hip_config_t *cfg = jb->p->cfg;
    const int swap = cfg->state_frame_body > 0 && cfg->state_frame_body < cfg->n_bodies && cfg->bodies[cfg->state_frame_body].n_chain > 0;
    /* with a trajectory (instance.rs:297-326): `start_state = self.state` is taken BEFORE propagate() translates the state, the channel is
     * fed inside the loop (integration frame, :188-193, :254-259) and only the RETURNED state is translated back (:211-220): the Traj
     * holds its first state in the caller's frame and every other one in the integration frame.  Restated as is. */
    traj_push(s);
    if (swap) (void)frame_shift(cfg, cfg->state_frame_body, s->epoch_ns, s->y, +1.0);  /* instance.rs:117-142 */
    int st = propagate(s, jb->duration_ns);
    if (swap) {  /* instance.rs:211-220; an epoch outside the ephemeris is reported by this translation */
        const int s2 = frame_shift(cfg, cfg->state_frame_body, s->epoch_ns, s->y, -1.0);
        if (s2 && st == NYX_HIP_OK) st = s2;
    }
    inst_store(s, jb->out, jb->stats, i, st);
}

static void *worker(void *arg) {
    job_t *jb = arg;
    inst_t *s = malloc(sizeof *s);
    scratch_init(&s->w, jb->p);
    for (;;) {
        long i = atomic_fetch_add(&jb->next, 1);
        if (i >= jb->in->n) break;
        scratch_t keep = s->w;
        run_one(jb, s, i);
        s->w = keep;
    }
    scratch_free(&s->w);
    free(s);
    return NULL;
}

/* ---------------------------------------------------------------------------------------------
 * Traj evaluation (md/trajectory/traj.rs:82-162, interpolatable.rs:52-108).
 *
 * `hermite_eval` lives in the third-party crate anise (Cargo.toml:34, anise = "0.10.2",
 * anise/src/math/interpolation/hermite.rs), which is NOT vendored under /root/reference.  It is a
 * port of NAIF SPICE's HRMINT; what follows restates HRMINT's published algorithm (divided-difference
 * table with every abscissa doubled; NAIF toolkit, hrmint.f "Particulars") in the operation order
 * of the SPICE routine.  Pinned on HRMINT's documented example (a degree-7 polynomial through four
 * points: f(2) = 141, f'(2) = 456) and on the reference's own test properties
 * (tests/propagation/trajectory.rs:103-135, 358-420); the exact rounding of anise's port is
 * "parity unpinned".
 * --------------------------------------------------------------------------------------------- */
#define INTERPOLATION_SAMPLES 13 /* interpolatable.rs:22 */

int32_t nyx_oracle_hermite_eval(const double *xs, const double *ys, const double *ydots, int32_t n, double x_eval,
                                double *f, double *df) {
    double work[4 * INTERPOLATION_SAMPLES];
    if (n < 1 || n > INTERPOLATION_SAMPLES) return NYX_HIP_INTERP_MATH;
    /* first column of the table: function values and derivatives interleaved */
    for (int i = 0; i < n; ++i) {
        work[2 * i] = ys[i];
        work[2 * i + 1] = ydots[i];
    }
    /* second column: first-degree interpolants (Fortran indices, as the routine is written) */
    for (int i = 1; i <= n - 1; ++i) {
        const double c1 = xs[i] - x_eval;
        const double c2 = x_eval - xs[i - 1];
        const double denom = xs[i] - xs[i - 1];
        if (fabs(denom) < DBL_EPSILON) return NYX_HIP_INTERP_MATH;
        const int prev = 2 * i - 1, cur = prev + 1, next = cur + 1;
        work[prev + 2 * n - 1] = work[cur - 1];
        work[cur + 2 * n - 1] = (work[next - 1] - work[prev - 1]) / denom;
        const double temp = work[cur - 1] * (x_eval - xs[i - 1]) + work[prev - 1];
        work[cur - 1] = (c1 * work[prev - 1] + c2 * work[next - 1]) / denom;
        work[prev - 1] = temp;
    }
    work[4 * n - 2] = work[2 * n - 1];
    work[2 * n - 2] = work[2 * n - 1] * (x_eval - xs[n - 1]) + work[2 * n - 2];
    /* columns 3 .. 2n */
    for (int j = 2; j <= 2 * n - 1; ++j) {
        for (int i = 1; i <= 2 * n - j; ++i) {
            const int xi = (i + 1) / 2;
            const int xij = (i + j + 1) / 2;
            const double c1 = xs[xij - 1] - x_eval;
            const double c2 = x_eval - xs[xi - 1];
            const double denom = xs[xij - 1] - xs[xi - 1];
            if (fabs(denom) < DBL_EPSILON) return NYX_HIP_INTERP_MATH;
            work[i + 2 * n - 1] = (c1 * work[i + 2 * n - 1] + c2 * work[i + 2 * n] + (work[i] - work[i - 1])) / denom;
            work[i - 1] = (c1 * work[i - 1] + c2 * work[i]) / denom;
        }
    }
    *f = work[0];
    *df = work[2 * n];
    return NYX_HIP_INTERP_OK;
}

/* stored states of trajectory i read as the finalize()d (epoch-sorted) sequence */
typedef struct { const nyx_hip_traj_t *t; int64_t n, i, len; int desc; } traj_view_t;

static traj_view_t traj_view(const nyx_hip_traj_t *t, int64_t n, int64_t i) {
    traj_view_t v = {t, n, i, t->len[i] < t->capacity ? t->len[i] : t->capacity, 0};
    if (v.len > 1) v.desc = t->epoch_ns[(v.len - 1) * n + i] < t->epoch_ns[i];
    return v;
}
static int64_t view_at(const traj_view_t *v, int64_t k) { return (v->desc ? v->len - 1 - k : k) * v->n + v->i; }

/* Traj::at (traj.rs:82-127) + Spacecraft::interpolate (interpolatable.rs:52-108) */
int32_t nyx_oracle_traj_at(const nyx_hip_traj_t *traj, int64_t n, int64_t i, int64_t epoch_ns, double *state6) {
    const traj_view_t v = traj_view(traj, n, i);
    const double *comp[6] = {traj->x_km, traj->y_km, traj->z_km, traj->vx_km_s, traj->vy_km_s, traj->vz_km_s};
    for (int c = 0; c < 6; ++c) state6[c] = NAN;
    if (v.len == 0 || traj->epoch_ns[view_at(&v, 0)] > epoch_ns || traj->epoch_ns[view_at(&v, v.len - 1)] < epoch_ns)
        return NYX_HIP_INTERP_NO_DATA;
    /* binary search: first index whose epoch is > query, or the exact hit */
    int64_t lo = 0, hi = v.len;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        const int64_t e = traj->epoch_ns[view_at(&v, mid)];
        if (e == epoch_ns) {
            for (int c = 0; c < 6; ++c) state6[c] = comp[c][view_at(&v, mid)];
            return NYX_HIP_INTERP_OK;
        }
        if (e < epoch_ns) lo = mid + 1; else hi = mid;
    }
    const int64_t idx = lo;
    if (idx == 0 || idx >= v.len) return NYX_HIP_INTERP_NO_DATA;
    const int64_t num_left = INTERPOLATION_SAMPLES / 2;
    int64_t first_idx = idx > num_left ? idx - num_left : 0;
    const int64_t last_idx = v.len < first_idx + INTERPOLATION_SAMPLES ? v.len : first_idx + INTERPOLATION_SAMPLES;
    if (last_idx == v.len) first_idx = last_idx > 2 * num_left ? last_idx - 2 * num_left : 0; /* 12, sic (traj.rs:112-114) */
    const int32_t ns = (int32_t)(last_idx - first_idx);
    double xs[INTERPOLATION_SAMPLES], ys[INTERPOLATION_SAMPLES], yd[INTERPOLATION_SAMPLES];
    for (int k = 0; k < ns; ++k) xs[k] = nyx_oracle_ns_to_seconds(traj->epoch_ns[view_at(&v, first_idx + k)]);
    const double x_eval = nyx_oracle_ns_to_seconds(epoch_ns);
    for (int c = 0; c < 3; ++c) {
        for (int k = 0; k < ns; ++k) {
            ys[k] = comp[c][view_at(&v, first_idx + k)];
            yd[k] = comp[c + 3][view_at(&v, first_idx + k)];
        }
        const int32_t st = nyx_oracle_hermite_eval(xs, ys, yd, ns, x_eval, &state6[c], &state6[c + 3]);
        if (st != NYX_HIP_INTERP_OK) {
            for (int q = 0; q < 6; ++q) state6[q] = NAN;
            return st;
        }
    }
    return NYX_HIP_INTERP_OK;
}

/* The conditioning warning of nyx_hip_traj_at (NYX_HIP_INTERP_ILL_CONDITIONED, include/nyx_hip.h): 1 when `epoch_ns` is interpolated
 * from a window that holds two states closer than 1e-4 of the window's mean spacing.  Not part of the reference (which returns the
 * fitted state without comment); the window is the one nyx_oracle_traj_at uses (traj.rs:100-115). */
int32_t nyx_oracle_traj_window_ill(const nyx_hip_traj_t *traj, int64_t n, int64_t i, int64_t epoch_ns) {
    const traj_view_t v = traj_view(traj, n, i);
    if (v.len == 0 || traj->epoch_ns[view_at(&v, 0)] > epoch_ns || traj->epoch_ns[view_at(&v, v.len - 1)] < epoch_ns) return 0;
    int64_t lo = 0, hi = v.len;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        const int64_t e = traj->epoch_ns[view_at(&v, mid)];
        if (e == epoch_ns) return 0;
        if (e < epoch_ns) lo = mid + 1; else hi = mid;
    }
    const int64_t idx = lo;
    if (idx == 0 || idx >= v.len) return 0;
    const int64_t num_left = INTERPOLATION_SAMPLES / 2;
    int64_t first_idx = idx > num_left ? idx - num_left : 0;
    const int64_t last_idx = v.len < first_idx + INTERPOLATION_SAMPLES ? v.len : first_idx + INTERPOLATION_SAMPLES;
    if (last_idx == v.len) first_idx = last_idx > 2 * num_left ? last_idx - 2 * num_left : 0;
    const int32_t ns = (int32_t)(last_idx - first_idx);
    if (ns < 2) return 0;
    double xs[INTERPOLATION_SAMPLES];
    for (int k = 0; k < ns; ++k) xs[k] = nyx_oracle_ns_to_seconds(traj->epoch_ns[view_at(&v, first_idx + k)]);
    double dmin = fabs(xs[1] - xs[0]);
    for (int k = 2; k < ns; ++k) dmin = fmin(dmin, fabs(xs[k] - xs[k - 1]));
    return dmin < 1e-4 * (fabs(xs[ns - 1] - xs[0]) / (double)(ns - 1)) ? 1 : 0;
}

/* Traj::every (traj.rs:148-162) through TrajIterator (traj_it.rs:33-62); TimeSeries::inclusive yields first + k*step
 * while k*step <= last - first. */
int32_t nyx_oracle_traj_every(const nyx_hip_traj_t *traj, int64_t n, int64_t step_ns, nyx_hip_traj_t *out) {
    if (!traj || !out || step_ns <= 0) return NYX_HIP_RC_BAD_ARG;
    double *oc[6] = {out->x_km, out->y_km, out->z_km, out->vx_km_s, out->vy_km_s, out->vz_km_s};
    for (int64_t i = 0; i < n; ++i) {
        const traj_view_t v = traj_view(traj, n, i);
        out->len[i] = 0;
        if (v.len == 0) continue;
        const int64_t first = traj->epoch_ns[view_at(&v, 0)], last = traj->epoch_ns[view_at(&v, v.len - 1)];
        for (int64_t k = 0; k * step_ns <= last - first; ++k) {
            double s6[6];
            if (nyx_oracle_traj_at(traj, n, i, first + k * step_ns, s6) != NYX_HIP_INTERP_OK) break;
            if (k < out->capacity) {
                out->epoch_ns[k * n + i] = first + k * step_ns;
                for (int c = 0; c < 6; ++c) oc[c][k * n + i] = s6[c];
            }
            out->len[i] = (int32_t)(k + 1);
        }
    }
    return NYX_HIP_RC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * until_nth_event (propagators/event.rs:88-211): propagate with the stop condition, then Brent on the
 * interpolant between the last published state and the end state.  brent_solver is anise's (absent):
 * restated from the `roots` crate's Brent as earlier Nyx releases embedded it — PARITY UNPINNED.
 * --------------------------------------------------------------------------------------------- */
static int ev_at(const nyx_hip_traj_t *traj, int64_t n, int64_t i, const nyx_hip_event_t *ev, double mu, int64_t epoch_ns,
                 double *value) {
    double s6[6];
    const int st = nyx_oracle_traj_at(traj, n, i, epoch_ns, s6);
    if (st != NYX_HIP_INTERP_OK) return st;
    *value = ev_eval(ev, mu, epoch_ns, s6);
    return NYX_HIP_INTERP_OK;
}

/* returns 0 and the event epoch, or 1 = not found in the bracket / 2 = evaluation failed / 3 = max iterations */
static int brent_event(const nyx_hip_traj_t *traj, int64_t n, int64_t i, const nyx_hip_event_t *ev, double mu, int64_t start_ns,
                       int64_t end_ns, int64_t *event_ns) {
    const double eps_t = nyx_oracle_ns_to_seconds(ev->epoch_precision_ns);
    const double eps_v = fabs(ev->value_precision);
    double xa = 0.0, xb = nyx_oracle_ns_to_seconds(end_ns - start_ns);
    double ya, yb;
    if (ev_at(traj, n, i, ev, mu, start_ns, &ya) || ev_at(traj, n, i, ev, mu, end_ns, &yb)) return 2;
    if (fabs(ya) <= eps_v) { *event_ns = start_ns; return 0; }
    if (fabs(yb) <= eps_v) { *event_ns = end_ns; return 0; }
    double xc = xa, yc = ya, xd = xa;
    int flag = 1;
    for (int it = 0; it < 50; ++it) {
        if (fabs(ya) < eps_v) { *event_ns = start_ns + nyx_oracle_seconds_to_ns(xa); return 0; }
        if (fabs(yb) < eps_v) { *event_ns = start_ns + nyx_oracle_seconds_to_ns(xb); return 0; }
        if (fabs(xa - xb) <= eps_t) return 1;
        double sx;
        if (fabs(ya - yc) > DBL_EPSILON && fabs(yb - yc) > DBL_EPSILON)
            sx = xa * yb * yc / ((ya - yb) * (ya - yc)) + xb * ya * yc / ((yb - ya) * (yb - yc)) + xc * ya * yb / ((yc - ya) * (yc - yb));
        else
            sx = xb - yb * (xb - xa) / (yb - ya);
        const int cond1 = (sx - xb) * (sx - (3.0 * xa + xb) / 4.0) > 0.0;
        const int cond2 = flag && fabs(sx - xb) >= fabs(xb - xc) / 2.0;
        const int cond3 = !flag && fabs(sx - xb) >= fabs(xc - xd) / 2.0;
        const int cond4 = flag && fabs(xb - xc) <= eps_t;
        const int cond5 = !flag && fabs(xc - xd) <= eps_t;
        if (cond1 || cond2 || cond3 || cond4 || cond5) { sx = (xa + xb) / 2.0; flag = 1; } else { flag = 0; }
        double ys;
        if (ev_at(traj, n, i, ev, mu, start_ns + nyx_oracle_seconds_to_ns(sx), &ys)) return 2;
        xd = xc; xc = xb; yc = yb;
        if (ya * ys < 0.0) { /* root between a and s */
            if (fabs(ya) > fabs(ys)) { xb = sx; yb = ys; } else { xb = xa; yb = ya; xa = sx; ya = ys; }
        } else {             /* root between s and b */
            if (fabs(ys) > fabs(yb)) { xa = sx; ya = ys; } else { xa = xb; ya = yb; xb = sx; yb = ys; }
        }
    }
    return 3;
}

int32_t nyx_oracle_until_event(const nyx_hip_config_t *cfg, const nyx_hip_states_t *in, int64_t max_duration_ns,
                               const nyx_hip_event_t *ev, nyx_hip_states_t *out, nyx_hip_step_stats_t *stats,
                               nyx_hip_traj_t *traj, int32_t *crossings) {
    if (!cfg || !in || !out || !ev || !traj || traj->capacity < 2 || ev->trigger < 1) return NYX_HIP_RC_BAD_ARG;
    prepared_t p;
    prepared_init(&p, cfg);
    inst_t *s = malloc(sizeof *s);
    scratch_init(&s->w, &p);
    double *oc[6] = {out->x_km, out->y_km, out->z_km, out->vx_km_s, out->vy_km_s, out->vz_km_s};
    for (int64_t i = 0; i < in->n; ++i) {
        scratch_t keep = s->w;
        inst_init(s, &p, in, i);
        s->traj = traj;
        traj->len[i] = 0;
        traj_push(s);
        s->ev = ev;
        s->ev_prev = ev_eval(ev, cfg->central_mu_km3_s2, s->epoch_ns, s->y); /* y_prev of the start state (event.rs:104-106) */
        int st = propagate(s, max_duration_ns);
        inst_store(s, out, stats, i, st);
        if (crossings) crossings[i] = s->ev_count;
        if (st == NYX_HIP_OK) {
            if (!s->ev_found) {
                st = NYX_HIP_ERR_EVENT_NOT_FOUND; /* end_state == last_traj_state (event.rs:170-176) */
            } else {
                const int64_t last = traj->len[i] - 1;
                if (traj->len[i] >= traj->capacity) {
                    st = NYX_HIP_ERR_EVENT_SEARCH; /* the bracket does not fit the caller's buffer */
                } else {
                    /* traj.states.last() after finalize() sorted by epoch (event.rs:165-168): for a back-propagation
                     * that is the START state, and the bracket is the whole arc */
                    const int64_t start_ns = max_duration_ns < 0 ? traj->epoch_ns[i] : traj->epoch_ns[last * in->n + i];
                    traj_push(s); /* traj.states.push(end_state) (event.rs:179) */
                    int64_t ev_ns = 0;
                    double s6[6];
                    if (brent_event(traj, in->n, i, ev, cfg->central_mu_km3_s2, start_ns, s->epoch_ns, &ev_ns) != 0 ||
                        nyx_oracle_traj_at(traj, in->n, i, ev_ns, s6) != NYX_HIP_INTERP_OK) {
                        st = NYX_HIP_ERR_EVENT_SEARCH;
                    } else {
                        out->epoch_ns[i] = ev_ns;
                        for (int c = 0; c < 6; ++c) oc[c][i] = s6[c];
                    }
                }
            }
            if (stats && stats->status) stats->status[i] = st;
        }
        s->w = keep;
    }
    scratch_free(&s->w);
    free(s);
    prepared_free(&p);
    return NYX_HIP_RC_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Covariance mapping: KalmanODProcess::predict_until (od/process/mod.rs:440-486) with
 * KalmanFilter::time_update (od/kalman/filtering.rs:59-99) and ProcessNoise::propagate (od/snc.rs:165-283).
 * 9x9 matrices are column-major (nalgebra storage); products accumulate k ascending, multiply then add, as
 * nalgebra's gemv/axcpy loop does for statically sized matrices.
 * --------------------------------------------------------------------------------------------- */
static void mat9_mul(const double *a, const double *b, int b_transposed, double *out) {
    for (int c = 0; c < 9; ++c)
        for (int r = 0; r < 9; ++r) {
            double acc = a[r] * (b_transposed ? b[c] : b[c * 9]);
            for (int k = 1; k < 9; ++k) acc = acc + a[k * 9 + r] * (b_transposed ? b[k * 9 + c] : b[c * 9 + k]);
            out[c * 9 + r] = acc;
        }
}

int32_t nyx_oracle_predict_until(const nyx_hip_config_t *cfg, const nyx_hip_states_t *in, const nyx_hip_predict_t *pc,
                                 nyx_hip_estimates_t *est, nyx_hip_states_t *out, nyx_hip_step_stats_t *stats,
                                 nyx_hip_predict_history_t *hist) {
    if (!cfg || !in || !out || !pc || !est || !est->covar || !(cfg->flags & NYX_HIP_FLAG_STM) || pc->max_step_ns <= 0)
        return NYX_HIP_RC_BAD_ARG;
    prepared_t p;
    prepared_init(&p, cfg);
    inst_t *s = malloc(sizeof *s);
    scratch_init(&s->w, &p);
    double *ident = calloc((size_t)in->n * 81, sizeof(double));
    for (int64_t i = 0; i < in->n; ++i)
        for (int k = 0; k < 9; ++k) ident[i * 81 + k * 10] = 1.0;
    nyx_hip_states_t in_stm = *in;
    in_stm.stm = ident; /* nominal_state().with_stm() (mod.rs:452) */
    for (int64_t i = 0; i < in->n; ++i) {
        scratch_t keep = s->w;
        inst_init(s, &p, &in_stm, i);
        double *covar = est->covar + 81 * i;
        double dev[9];
        for (int k = 0; k < 9; ++k) dev[k] = est->state_dev ? est->state_dev[9 * i + k] : 0.0;
        int64_t prev_epoch = s->epoch_ns;
        const int64_t init_epoch = s->epoch_ns; /* ProcessNoise::init_epoch = the initial estimate's epoch (kalman/initializers.rs:75) */
        int32_t n_up = 0;
        int st = NYX_HIP_OK;
        /* opts.integration_frame: every segment is one `prop.until_epoch` (mod.rs:453-468), i.e. one translation in and one back
         * (instance.rs:117-142, 211-220); the STM does not see a translation */
        const int swap = cfg->state_frame_body > 0 && cfg->state_frame_body < cfg->n_bodies && cfg->bodies[cfg->state_frame_body].n_chain > 0;
        for (;;) {
            if (swap) (void)frame_shift(cfg, cfg->state_frame_body, s->epoch_ns, s->y, +1.0);
            st = propagate(s, pc->max_step_ns);
            if (swap) {
                const int s2 = frame_shift(cfg, cfg->state_frame_body, s->epoch_ns, s->y, -1.0);
                if (s2 && st == NYX_HIP_OK) st = s2;
            }
            if (st) break;
            const double *stm = s->y + 9;
            double m[81], cb[81];
            mat9_mul(stm, covar, 0, m);
            mat9_mul(m, stm, 1, cb);
            const int64_t delta_ns = s->epoch_ns - prev_epoch;
            for (int q = pc->n_process_noise - 1; q >= 0; --q) {
                const nyx_hip_process_noise_t *pn = &pc->process_noise[q];
                if (pn->has_start_time && pn->start_time_ns > s->epoch_ns) continue;
                if (delta_ns > pn->disable_time_ns) continue;
                /* ProcessNoise::to_matrix (snc.rs:165-205): the diagonal at this epoch; ::propagate (:219-239): expressed in the
                 * state frame through dcm_to_inertial(local_frame) at the nominal orbit (anise, absent: RIC = [r^, c^ x r^, c^],
                 * VNC = [v^, n^, v^ x n^]), only the diagonal of dcm * snc * dcm^T kept */
                double d[3] = {pn->diag[0], pn->diag[1], pn->diag[2]};
                if (pn->has_decay) {
                    const int64_t init = pn->init_epoch_ns != INT64_MIN ? pn->init_epoch_ns : init_epoch;
                    const double total = nyx_oracle_ns_to_seconds(s->epoch_ns - init);
                    for (int k = 0; k < 3; ++k) d[k] = d[k] * exp(-pn->decay_s[k] * total);
                }
                if (pn->local_frame != NYX_HIP_FRAME_INERTIAL) {
                    const double *r = s->y, *v = s->y + 3;
                    double h[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
                    const double hn = norm3(h);
                    for (int k = 0; k < 3; ++k) h[k] = h[k] / hn;
                    double e0[3], e1[3], e2[3];
                    if (pn->local_frame == NYX_HIP_FRAME_RIC) {
                        const double rn = norm3(r);
                        for (int k = 0; k < 3; ++k) { e0[k] = r[k] / rn; e2[k] = h[k]; }
                        e1[0] = e2[1] * e0[2] - e2[2] * e0[1]; e1[1] = e2[2] * e0[0] - e2[0] * e0[2]; e1[2] = e2[0] * e0[1] - e2[1] * e0[0];
                    } else {
                        const double vn = norm3(v);
                        for (int k = 0; k < 3; ++k) { e0[k] = v[k] / vn; e1[k] = h[k]; }
                        e2[0] = e0[1] * e1[2] - e0[2] * e1[1]; e2[1] = e0[2] * e1[0] - e0[0] * e1[2]; e2[2] = e0[0] * e1[1] - e0[1] * e1[0];
                    }
                    double nd[3];
                    for (int k = 0; k < 3; ++k) nd[k] = ((e0[k] * d[0]) * e0[k] + (e1[k] * d[1]) * e1[k]) + (e2[k] * d[2]) * e2[k];
                    for (int k = 0; k < 3; ++k) d[k] = nd[k];
                }
                const double dt = nyx_oracle_ns_to_seconds(delta_ns);
                const double half_dt2 = (dt * dt) / 2.0;
                for (int c = 0; c < 6; ++c)
                    for (int r = 0; r < 6; ++r)
                        if (r % 3 == c % 3) {
                            const double g_r = r < 3 ? half_dt2 : dt, g_c = c < 3 ? half_dt2 : dt;
                            cb[c * 9 + r] = cb[c * 9 + r] + (g_r * d[r % 3]) * g_c;
                        }
                break;
            }
            double sb[9];
            for (int r = 0; r < 9; ++r) {
                sb[r] = 0.0;
                if (pc->deviation_tracking) {
                    sb[r] = stm[r] * dev[0];
                    for (int k = 1; k < 9; ++k) sb[r] = sb[r] + stm[k * 9 + r] * dev[k];
                }
            }
            if (hist && n_up < hist->capacity) {
                const int64_t slot = (int64_t)n_up * in->n + i;
                if (hist->epoch_ns) hist->epoch_ns[slot] = s->epoch_ns;
                if (hist->state) memcpy(hist->state + slot * 9, s->y, 9 * sizeof(double));
                if (hist->stm) memcpy(hist->stm + slot * 81, stm, 81 * sizeof(double));
                if (hist->covar) memcpy(hist->covar + slot * 81, cb, 81 * sizeof(double));
                if (hist->state_dev) memcpy(hist->state_dev + slot * 9, sb, 9 * sizeof(double));
            }
            memcpy(covar, cb, sizeof cb);
            memcpy(dev, sb, sizeof sb);
            n_up += 1;
            prev_epoch = s->epoch_ns;
            memset(s->y + 9, 0, 81 * sizeof(double)); /* reset_stm() (mod.rs:479) */
            for (int k = 0; k < 9; ++k) s->y[9 + k * 10] = 1.0;
            if (s->epoch_ns >= pc->end_epoch_ns) break;
        }
        if (est->state_dev) memcpy(est->state_dev + 9 * i, dev, sizeof dev);
        if (hist && hist->n_updates) hist->n_updates[i] = n_up;
        inst_store(s, out, stats, i, st);
        s->w = keep;
    }
    free(ident);
    scratch_free(&s->w);
    free(s);
    prepared_free(&p);
    return NYX_HIP_RC_OK;
}

int32_t nyx_oracle_propagate_batch(const nyx_hip_config_t *cfg, const nyx_hip_states_t *in, int64_t duration_ns,
                                   nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, int32_t n_threads) {
    return nyx_oracle_propagate_batch_traj(cfg, in, duration_ns, out, stats, NULL, n_threads);
}

int32_t nyx_oracle_propagate_batch_traj(const nyx_hip_config_t *cfg, const nyx_hip_states_t *in, int64_t duration_ns,
                                        nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, const nyx_hip_traj_t *traj,
                                        int32_t n_threads) {
    if (!cfg || !in || !out || cfg->opts.method < 0 || cfg->opts.method > 5) return NYX_HIP_RC_BAD_ARG;
    prepared_t p;
    prepared_init(&p, cfg);
    job_t jb = {&p, in, out, stats, duration_ns, 0, traj};
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (n_threads == 1 || in->n <= 1) {
        worker(&jb);
    } else {
        pthread_t th[256];
        for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, worker, &jb);
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    }
    prepared_free(&p);
    return NYX_HIP_RC_OK;
}
