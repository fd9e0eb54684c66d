// pk_harmonics.h - part of propagate_kernel.hip (included there, in this order; not a stand-alone header): spherical harmonics, column-split: the scalar feed, the streamed feed (harm_stream_asm.h), second field, dual variants.
// ---------------------------------------------------------------------------------------------
// Spherical harmonics, column-split.  Inputs are per lane (trajectory); every table operand is
// wave-uniform (scalar loads).  Scaled recursion for column c, rows n' = c..N+1:
//   At_c = rho * diag[c];  At_n' = (rho u) b[n'][c] At_{n'-1} - rho^2 c[n'][c] At_{n'-2}
// (At_n' = rho^(n'-c+1) A[n'][c]); per-column complex power (Rc, Ic) = (rho (s + i t))^(c-1).
// ---------------------------------------------------------------------------------------------

// Forward-mode dual number: value + partials w.r.t. the three position components (stand-in for the
// reference's OHyperdual<f64, 7> whose slots 1..3 carry d/dx, d/dy, d/dz; gravity_field.rs:273-431).
struct D3 {
    double v, x, y, z;
};
DEVFN D3 d3c(double v) { D3 r = {v, 0.0, 0.0, 0.0}; return r; }
DEVFN D3 operator+(D3 a, D3 b) { D3 r = {a.v + b.v, a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
DEVFN D3 operator-(D3 a, D3 b) { D3 r = {a.v - b.v, a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
DEVFN D3 operator-(D3 a) { D3 r = {-a.v, -a.x, -a.y, -a.z}; return r; }
DEVFN D3 operator*(D3 a, D3 b) {
    D3 r = {a.v * b.v, __builtin_fma(a.v, b.x, a.x * b.v), __builtin_fma(a.v, b.y, a.y * b.v), __builtin_fma(a.v, b.z, a.z * b.v)};
    return r;
}
DEVFN D3 operator*(D3 a, double s) { D3 r = {a.v * s, a.x * s, a.y * s, a.z * s}; return r; }
DEVFN D3 operator*(double s, D3 a) { return a * s; }
DEVFN D3 d3div(D3 a, D3 b) {  // hyperdual Div: real = a/b, dual_i = (a_i b - a b_i) / b^2
    const double dd = b.v * b.v;
    D3 r = {a.v / b.v, (a.x * b.v - a.v * b.x) / dd, (a.y * b.v - a.v * b.y) / dd, (a.z * b.v - a.v * b.z) / dd};
    return r;
}
DEVFN D3 d3sqrt(D3 a) {
    const double s = sqrt(a.v);
    const double hh = 0.5 / s;
    D3 r = {s, a.x * hh, a.y * hh, a.z * hh};
    return r;
}
DEVFN D3 d3norm(D3 a, D3 b, D3 c) { return d3sqrt(a * a + b * b + c * c); }
DEVFN D3 d3cube(D3 a) {  // powi(3): real = (a*a)*a, dual = 3 a^2 da
    const double p = a.v * a.v;
    const double f = 3.0 * p;
    D3 r = {p * a.v, a.x * f, a.y * f, a.z * f};
    return r;
}

// One-partial dual: value + ONE position partial.  QUAD LAYOUT of the STM kernel (small ensembles): the four lanes of a
// quad belong to ONE trajectory; each runs the same dual program as D3 but carries a single partial - lane 1: d/dx,
// lane 2: d/dy, lane 3: d/dz (lane 0: the value only, d = 0).  Every operation below is D3's own expression for `v` and
// for one of its three partial slots, so value and partials are bit-identical to the 64-lane D3 layout; what changes is
// 3 f64 operations per product instead of 7 and a quarter of the registers, i.e. a kernel that fits 16 waves per
// workgroup where the D3 variant fits 4.
struct D1 {
    double v, d;
};
DEVFN D1 d1c(double v) { D1 r = {v, 0.0}; return r; }
DEVFN D1 operator+(D1 a, D1 b) { D1 r = {a.v + b.v, a.d + b.d}; return r; }
DEVFN D1 operator-(D1 a, D1 b) { D1 r = {a.v - b.v, a.d - b.d}; return r; }
DEVFN D1 operator-(D1 a) { D1 r = {-a.v, -a.d}; return r; }
DEVFN D1 operator*(D1 a, D1 b) { D1 r = {a.v * b.v, __builtin_fma(a.v, b.d, a.d * b.v)}; return r; }
DEVFN D1 operator*(D1 a, double s) { D1 r = {a.v * s, a.d * s}; return r; }
DEVFN D1 operator*(double s, D1 a) { return a * s; }
DEVFN D1 d1div(D1 a, D1 b) {
    const double dd = b.v * b.v;
    D1 r = {a.v / b.v, (a.d * b.v - a.v * b.d) / dd};
    return r;
}
DEVFN D1 d1sqrt(D1 a) {
    const double s = sqrt(a.v);
    const double hh = 0.5 / s;
    D1 r = {s, a.d * hh};
    return r;
}
DEVFN D1 d1norm(D1 a, D1 b, D1 c) { return d1sqrt(a * a + b * b + c * c); }
DEVFN D1 d1cube(D1 a) {
    const double p = a.v * a.v;
    const double f = 3.0 * p;
    D1 r = {p * a.v, a.d * f};
    return r;
}
// the seed of position component `comp` (0..2) in quad lane `ql`: d(r_comp)/d(r_{ql-1})
DEVFN D1 d1seed(double v, int comp, int ql) { D1 r = {v, (ql == comp + 1) ? 1.0 : 0.0}; return r; }

// scalar-generic helpers so that the column recursion is written once for double and D3
DEVFN double sfma(double a, double s, double c) { return __builtin_fma(a, s, c); }              // a * s + c, s uniform
DEVFN D3 sfma(D3 a, double s, D3 c) {
    D3 r = {__builtin_fma(a.v, s, c.v), __builtin_fma(a.x, s, c.x), __builtin_fma(a.y, s, c.y), __builtin_fma(a.z, s, c.z)};
    return r;
}
DEVFN D1 sfma(D1 a, double s, D1 c) { D1 r = {__builtin_fma(a.v, s, c.v), __builtin_fma(a.d, s, c.d)}; return r; }
DEVFN double gmul(double a, double b) { return a * b; }
DEVFN D3 gmul(D3 a, D3 b) { return a * b; }
DEVFN D1 gmul(D1 a, D1 b) { return a * b; }
DEVFN D1 gfma(D1 a, D1 b, D1 c) { return a * b + c; }
DEVFN D1 gzero(D1) { return d1c(0.0); }
DEVFN D1 gone(D1) { return d1c(1.0); }
DEVFN D1 gdiv(D1 a, D1 b) { return d1div(a, b); }
DEVFN D1 gnorm3(D1 a, D1 b, D1 c) { return d1norm(a, b, c); }
DEVFN D1 glift(double v, D1) { return d1c(v); }
DEVFN double gfma(double a, double b, double c) { return __builtin_fma(a, b, c); }               // a * b + c
DEVFN D3 gfma(D3 a, D3 b, D3 c) { return a * b + c; }
DEVFN double gzero(double) { return 0.0; }
DEVFN D3 gzero(D3) { return d3c(0.0); }
DEVFN double gone(double) { return 1.0; }
DEVFN D3 gone(D3) { return d3c(1.0); }
DEVFN double gdiv(double a, double b) { return a / b; }
DEVFN D3 gdiv(D3 a, D3 b) { return d3div(a, b); }
DEVFN double gnorm3(double a, double b, double c) { return norm3(a, b, c); }
DEVFN D3 gnorm3(D3 a, D3 b, D3 c) { return d3norm(a, b, c); }
DEVFN double glift(double v, double) { return v; }
DEVFN D3 glift(double v, D3) { return d3c(v); }

// SolidTides (reference dynamics/solid_tides.rs): delta-C/S of degrees 2-3 raised by the perturbers
// (TidalPerturber::compute_pert, :74-175) and the degree-3 evaluation at the spacecraft (eom :238-385; gradient
// :387-559 when T = D3, the deltas being functions of the epoch only).  `ed` holds this stage's DCM inertial ->
// body-fixed and the perturber positions.  r and acc are inertial; with T = D3 the partials are w.r.t. inertial r.
// The derived-Legendre table is walked column by column (only the 11 entries the two degrees touch are formed).
template <typename T>
DEVFN void tides_accel(CfgPtr cfg, const double *ed, int lane, const T (&r)[3], T (&acc)[3]) {
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) m[q] = ed[q * DEV_LANES + lane];
    double c2[3] = {0.0, 0.0, 0.0}, s2[3] = {0.0, 0.0, 0.0}, c3[4] = {0.0, 0.0, 0.0, 0.0}, s3[4] = {0.0, 0.0, 0.0, 0.0};
    const int np = cfg->t_n;
#pragma unroll
    for (int j = 0; j < DEV_MAX_SLOTS; ++j) {
        if (j < np) {
            const int sl = cfg->t_slot[j];
            double psl[3];
            ed_body(cfg, ed, lane, sl, psl);
            const double p0 = psl[0], p1 = psl[1], p2 = psl[2];
            const double b0 = m[0] * p0 + m[1] * p1 + m[2] * p2;
            const double b1 = m[3] * p0 + m[4] * p1 + m[5] * p2;
            const double b2 = m[6] * p0 + m[7] * p1 + m[8] * p2;
            const double r_body = norm3(b0, b1, b2);
            const double s_body = b0 / r_body, t_body = b1 / r_body, sin_phi = b2 / r_body;
            const double cos_phi = sqrt(fmax(1.0 - sin_phi * sin_phi, 0.0));
            const double cl = cos_phi > 1e-12 ? s_body / cos_phi : 1.0;
            const double sn = cos_phi > 1e-12 ? t_body / cos_phi : 0.0;
            const double gm = cfg->t_gm_ratio[j];
            const double rr = cfg->t_re / r_body;
            const double cl2 = cl * cl, sn2 = sn * sn;
            const double cos2 = cl2 - sn2, sin2 = 2.0 * sn * cl;
            {
                const double common = cfg->t_k2_5 * gm * powi_dev(rr, 3);
                const double p20 = 0.5 * (3.0 * (sin_phi * sin_phi) - 1.0) * sqrt(5.0);
                const double p21 = 3.0 * sin_phi * cos_phi * sqrt(5.0 / 3.0);
                const double p22 = 3.0 * (cos_phi * cos_phi) * sqrt(5.0 / 12.0);
                c2[0] += common * p20;
                c2[1] += common * p21 * cl;  s2[1] += common * p21 * sn;
                c2[2] += common * p22 * cos2; s2[2] += common * p22 * sin2;
            }
            if (cfg->t_deg3[j]) {
                const double common = cfg->t_k3_7 * gm * powi_dev(rr, 4);
                const double p30 = 0.5 * (5.0 * powi_dev(sin_phi, 3) - 3.0 * sin_phi) * sqrt(7.0);
                const double p31 = 1.5 * (5.0 * (sin_phi * sin_phi) - 1.0) * cos_phi * sqrt(7.0 / 6.0);
                const double p32 = 15.0 * sin_phi * (cos_phi * cos_phi) * sqrt(7.0 / 60.0);
                const double p33 = 15.0 * powi_dev(cos_phi, 3) * sqrt(7.0 / 360.0);
                const double cos3 = cl * (cl2 - 3.0 * sn2), sin3 = sn * (3.0 * cl2 - sn2);
                c3[0] += common * p30;
                c3[1] += common * p31 * cl;   s3[1] += common * p31 * sn;
                c3[2] += common * p32 * cos2; s3[2] += common * p32 * sin2;
                c3[3] += common * p33 * cos3; s3[3] += common * p33 * sin3;
            }
        }
    }
    // ---- spacecraft side
    const T rb0 = r[0] * m[0] + r[1] * m[1] + r[2] * m[2];
    const T rb1 = r[0] * m[3] + r[1] * m[4] + r[2] * m[5];
    const T rb2 = r[0] * m[6] + r[1] * m[7] + r[2] * m[8];
    const T rmag = gnorm3(rb0, rb1, rb2);
    const T s_ = gdiv(rb0, rmag), t_ = gdiv(rb1, rmag), u_ = gdiv(rb2, rmag);
    // diagonal a[n][n] = sqrt(1 + 1/(2n)) a[n-1][n-1]: position-independent
    const double d1 = sqrt(1.5), d2 = sqrt(1.25) * d1, d3 = sqrt(1.0 + 1.0 / 6.0) * d2, d4 = sqrt(1.125) * d3;
    // b(n, m), c(n, m) of solid_tides.rs:266-276
#define TB(n, m) sqrt(((2.0 * (n) + 1.0) * (2.0 * (n) - 1.0)) / (((n) + (m)) * (double)((n) - (m))))
#define TC(n, m) sqrt(((2.0 * (n) + 1.0) * ((n) + (m) - 1.0) * ((n) - (m) - 1.0)) / (((n) - (m)) * (double)((n) + (m)) * (2.0 * (n) - 3.0)))
    // (column 0 never enters: m * a[n][0] = 0, and the z / w sums read columns m + 1)
    const T a21 = u_ * (sqrt(5.0) * d1);
    const T a31 = (u_ * TB(3, 1)) * a21 - glift(TC(3, 1) * d1, u_);
    const T a41 = (u_ * TB(4, 1)) * a31 - a21 * TC(4, 1);
    const T a32 = u_ * (sqrt(7.0) * d2);
    const T a42 = (u_ * TB(4, 2)) * a32 - glift(TC(4, 2) * d2, u_);
    const T a43 = u_ * (3.0 * d3);
#undef TB
#undef TC
    const T r2 = s_ * s_ - t_ * t_, i2 = s_ * t_ + t_ * s_;
    const T r3 = s_ * r2 - t_ * i2, i3 = s_ * i2 + t_ * r2;
    const double SQ2 = 1.41421356237309504880;
    // vr01(n, m) = sqrt((n-m)(n+m+1)) [/ sqrt2 for m = 0], vr11(n, m) = sqrt((2n+1)(n+m+2)(n+m+1)/(2n+3)) [/ sqrt2]
#define VR01(n, m) (sqrt(((n) - (m)) * ((n) + (m) + 1.0)) / ((m) == 0 ? SQ2 : 1.0))
#define VR11(n, m) (sqrt(((2.0 * (n) + 1.0) * ((n) + (m) + 2.0) * ((n) + (m) + 1.0)) / (2.0 * (n) + 3.0)) / ((m) == 0 ? SQ2 : 1.0))
    // degree 2
    T x2, y2, z2, w2;
    {
        const T dd0 = glift(c2[0] * SQ2, u_);                                   // (C r_0 + S i_0) sqrt2, r_0 = 1, i_0 = 0
        const T dd1 = (s_ * c2[1] + t_ * s2[1]) * SQ2;
        const T dd2 = (r2 * c2[2] + i2 * s2[2]) * SQ2;
        const double e1 = c2[1] * SQ2, f1 = s2[1] * SQ2;                        // m = 1: r_0, i_0
        const T e2 = (s_ * c2[2] + t_ * s2[2]) * SQ2, f2 = (s_ * s2[2] - t_ * c2[2]) * SQ2;
        x2 = a21 * e1 + e2 * (2.0 * d2);                                        // sum m a[2][m] e_m, a22 = d2
        y2 = a21 * f1 + f2 * (2.0 * d2);
        z2 = a21 * dd0 * VR01(2, 0) + dd1 * (VR01(2, 1) * d2);                  // a[2][3] = 0
        w2 = -(a31 * dd0 * VR11(2, 0) + a32 * dd1 * VR11(2, 1) + dd2 * (VR11(2, 2) * d3));
    }
    // degree 3
    T x3, y3, z3, w3;
    {
        const T dd0 = glift(c3[0] * SQ2, u_);
        const T dd1 = (s_ * c3[1] + t_ * s3[1]) * SQ2;
        const T dd2 = (r2 * c3[2] + i2 * s3[2]) * SQ2;
        const T dd3 = (r3 * c3[3] + i3 * s3[3]) * SQ2;
        const double e1 = c3[1] * SQ2, f1 = s3[1] * SQ2;
        const T e2 = (s_ * c3[2] + t_ * s3[2]) * SQ2, f2 = (s_ * s3[2] - t_ * c3[2]) * SQ2;
        const T e3 = (r2 * c3[3] + i2 * s3[3]) * SQ2, f3 = (r2 * s3[3] - i2 * c3[3]) * SQ2;
        x3 = a31 * e1 + a32 * e2 * 2.0 + e3 * (3.0 * d3);
        y3 = a31 * f1 + a32 * f2 * 2.0 + f3 * (3.0 * d3);
        z3 = a31 * dd0 * VR01(3, 0) + a32 * dd1 * VR01(3, 1) + dd2 * (VR01(3, 2) * d3);   // a[3][4] = 0
        w3 = -(a41 * dd0 * VR11(3, 0) + a42 * dd1 * VR11(3, 1) + a43 * dd2 * VR11(3, 2) + dd3 * (VR11(3, 3) * d4));
    }
#undef VR01
#undef VR11
    const T rho = gdiv(glift(cfg->t_re, u_), rmag);
    const T rho3 = gdiv(glift(cfg->t_mu, u_), rmag) * rho * rho * rho;  // rho_np1 at n = 2
    const T rho4 = rho3 * rho;
    const double inv_re = 1.0 / cfg->t_re;
    const T k2 = rho3 * inv_re, k3 = rho4 * inv_re;
    const T ax = k2 * x2 + k3 * x3, ay = k2 * y2 + k3 * y3, az = k2 * z2 + k3 * z3, aw = k2 * w2 + k3 * w3;
    const T l0 = ax + aw * s_, l1 = ay + aw * t_, l2 = az + aw * u_;
    acc[0] = l0 * m[0] + l1 * m[3] + l2 * m[6];
    acc[1] = l0 * m[1] + l1 * m[4] + l2 * m[7];
    acc[2] = l0 * m[2] + l1 * m[5] + l2 * m[8];
}

// Out of line on purpose (like harmonics_partial): inlined, the model's ~60 live doubles perturb the register
// allocation of the whole perturbation role and cost 3 % of the north-star run even when no tides are configured.
static __device__ __attribute__((noinline)) void tides_into_pert(CfgPtr cfg, const double *ed, int lane, const double *ys, double *pert) {
    const double r[3] = {ys[0 * DEV_LANES + lane], ys[1 * DEV_LANES + lane], ys[2 * DEV_LANES + lane]};
    double a[3];
    tides_accel<double>(cfg, ed, lane, r, a);
#pragma unroll
    for (int e = 0; e < 3; ++e) pert[e * DEV_LANES + lane] = pert[e * DEV_LANES + lane] + a[e];
}

// (zr + i zi)^e by binary exponentiation, e wave-uniform.  Real branches on the bits of e (the optimiser's if-converted
// form multiplies in every round and selects): the first set bit copies the base instead of multiplying by one, the
// last round does not square.  Every product that is formed is the one the plain loop forms: same value bit for bit.
template <typename T>
DEVFN void cpow_uniform(T zr, T zi, int e, T &pr, T &pi) {
    pr = gone(zr);
    pi = gzero(zr);
    T br = zr, bi = zi;
    bool first = true;
    while (e) {
        if (e & 1) {
            if (first) {
                pr = br; pi = bi;
                first = false;
                asm volatile("" ::: "memory");
            } else {
                const T t = gmul(pr, br) - gmul(pi, bi);
                pi = gmul(pr, bi) + gmul(pi, br);
                pr = t;
            }
            asm volatile("" ::: "memory");  // keep this a branch
        }
        e >>= 1;
        if (e) {
            const T t = gmul(br, br) - gmul(bi, bi);
            bi = (gmul(br, bi)) * 2.0;
            br = t;
            asm volatile("" ::: "memory");
        }
    }
}

#define HARM_TERM(h)                                                                       \
    {                                                                                      \
        const T an = gfma(rho_u, a1, -(gmul(rho2 * (h).g, a2)));                           \
        s1 = sfma(an, (h).t1, s1);                                                         \
        s2 = sfma(an, (h).t2, s2);                                                         \
        s3 = sfma(an, (h).t3, s3);                                                         \
        s4 = sfma(an, (h).t4, s4);                                                         \
        s5 = sfma(an, (h).t5, s5);                                                         \
        s6 = sfma(an, (h).t6, s6);                                                         \
        a2 = a1;                                                                           \
        a1 = an;                                                                           \
    }

#ifndef TOUCH_AHEAD
#define TOUCH_AHEAD 1
#endif
// One batch of the table = 280 contiguous bytes = 70 SGPRs, fetched by six scalar loads behind a single wait.
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
struct HarmBatch {
    v16i q0, q1, q2, q3;
    v4i q4;
    v2i q5;
};
static_assert(HARM_BATCH == 5 && sizeof(HarmEntry) == 56, "load_batch spells out five 56-byte entries");
DEVFN void load_batch(HarmPtr e, HarmBatch &b) {
    asm volatile(
        "s_load_dwordx16 %0, %6, 0x0\n\t"
        "s_load_dwordx16 %1, %6, 0x40\n\t"
        "s_load_dwordx16 %2, %6, 0x80\n\t"
        "s_load_dwordx16 %3, %6, 0xc0\n\t"
        "s_load_dwordx4 %4, %6, 0x100\n\t"
        "s_load_dwordx2 %5, %6, 0x110\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(b.q0), "=&s"(b.q1), "=&s"(b.q2), "=&s"(b.q3), "=&s"(b.q4), "=&s"(b.q5)
        : "s"(e)
        : "memory");
}
// Touch the (up to six) 64-byte lines of the batch TOUCH_AHEAD batches further on (results discarded): by the time its
// loads are issued the lines are in the scalar cache or on their way.  (The table is padded accordingly.)
// `sink` is read and written so that the register stays allocated for as long as a touch can be in flight: until the
// wait inside the next load_batch(), or touch_done() after the last batch of a column.
DEVFN void touch_batch(HarmPtr e, int &sink) {
    asm volatile(
        "s_load_dword %0, %1, %2\n\t"
        "s_load_dword %0, %1, %3\n\t"
        "s_load_dword %0, %1, %4\n\t"
        "s_load_dword %0, %1, %5\n\t"
        "s_load_dword %0, %1, %6\n\t"
        "s_load_dword %0, %1, %7"
        : "+&s"(sink)
        : "s"(e), "n"(TOUCH_AHEAD * 0x118), "n"(TOUCH_AHEAD * 0x118 + 0x40), "n"(TOUCH_AHEAD * 0x118 + 0x80), "n"(TOUCH_AHEAD * 0x118 + 0xc0),
          "n"(TOUCH_AHEAD * 0x118 + 0x100), "n"(TOUCH_AHEAD * 0x118 + 0x114)
        : "memory");
}
DEVFN void touch_done(int &sink) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sink) : : "memory"); }
#define HB_D(v, i) __builtin_bit_cast(double, (v2i){(v)[(i)], (v)[(i) + 1]})
#define HB_ENTRY(v0, v1, v2, v3, v4, v5, v6, i0, i1, i2, i3, i4, i5, i6) \
    { HB_D(v0, i0), HB_D(v1, i1), HB_D(v2, i2), HB_D(v3, i3), HB_D(v4, i4), HB_D(v5, i5), HB_D(v6, i6) }

DEVFN uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

DEVFN ColHdr load_hdr(ColPtr cols, int c) {
    const ColHdr CAS &r = cols[c];
    ColHdr h;
    h.start = r.start; h.nb = r.nb; h.scale = r.scale; h.diag = r.diag; h.rows = r.rows; h._pad = 0;
    return h;
}

template <typename T>
struct Partial4T {
    T x, y, z, w;
};
typedef Partial4T<double> Partial4;

// Not inlined on purpose: the batch loop wants 64 SGPRs for its four in-flight table entries, which it only
// gets when it is register-allocated on its own, away from the role code that calls it.  Arguments arrive in
// VGPRs under the device-function ABI, so the wave-uniform ones are re-scalarised with v_readfirstlane.
// T = double: accelerations only; T = D3: accelerations and their body-fixed position partials (STM path).
template <typename T>
DEVFN Partial4T<T> harmonics_core(CfgPtr cfg, HarmPtr htab, ColPtr cols, const int wave, const int sched, T zr, T zi, T rho_u, T rho,
                                  T inv_rho) {
    T px = gzero(zr), py = gzero(zr), pz = gzero(zr), pw = gzero(zr);
    const T rho2 = gmul(rho, rho);
    const CAS DevSched &sd = cfg->sched[sched];
    const int nr = sd.n_ranges[wave];
    for (int q = 0; q < nr; ++q) {
        const int c0 = sd.range_c0[wave][q];
        const int cnt = sd.range_cnt[wave][q];
        T rc, ic;
        cpow_uniform(zr, zi, c0 - 1, rc, ic);
        ColHdr hd = load_hdr(cols, c0);  // the next column's header is fetched under this column's batches
        for (int c = c0; c < c0 + cnt; ++c) {
            const ColHdr hn = load_hdr(cols, c + 1);  // (the header array has a spare tail entry)
            HarmPtr e = htab + hd.start;
            const int nb = hd.nb & 0xffff, rem = hd.nb >> 16;
            T a1 = gzero(zr), a2 = inv_rho * hd.diag;
            T s1 = gzero(zr), s2 = gzero(zr), s3 = gzero(zr), s4 = gzero(zr), s5 = gzero(zr), s6 = gzero(zr);
            int sink = 0;
            for (int b = 0; b < nb; ++b, e += HARM_BATCH) {
                // five 56-byte entries per batch: 70 SGPRs of scalar loads in flight behind ONE wait, then 45 f64 VALU ops per
                // lane.  The loads are spelled out: left to the scheduler, instantiations under register pressure wait after every load.
                HarmBatch hb;
                load_batch(e, hb);
                if (TOUCH_AHEAD) touch_batch(e, sink);
                const HarmEntry h0 = HB_ENTRY(hb.q0, hb.q0, hb.q0, hb.q0, hb.q0, hb.q0, hb.q0, 0, 2, 4, 6, 8, 10, 12);
                const HarmEntry h1 = HB_ENTRY(hb.q0, hb.q1, hb.q1, hb.q1, hb.q1, hb.q1, hb.q1, 14, 0, 2, 4, 6, 8, 10);
                const HarmEntry h2 = HB_ENTRY(hb.q1, hb.q1, hb.q2, hb.q2, hb.q2, hb.q2, hb.q2, 12, 14, 0, 2, 4, 6, 8);
                const HarmEntry h3 = HB_ENTRY(hb.q2, hb.q2, hb.q2, hb.q3, hb.q3, hb.q3, hb.q3, 10, 12, 14, 0, 2, 4, 6);
                const HarmEntry h4 = HB_ENTRY(hb.q3, hb.q3, hb.q3, hb.q3, hb.q4, hb.q4, hb.q5, 8, 10, 12, 14, 0, 2, 0);
                HARM_TERM(h0)
                HARM_TERM(h1)
                HARM_TERM(h2)
                HARM_TERM(h3)
                HARM_TERM(h4)
            }
            if (rem) {
                // the last 1..4 rows of the column: ONE more batch load (it runs into the next column's rows, or into the
                // table's padding) and only the first `rem` terms - one scalar-load latency instead of `rem` of them
                HarmBatch hb;
                load_batch(e, hb);
                const HarmEntry h0 = HB_ENTRY(hb.q0, hb.q0, hb.q0, hb.q0, hb.q0, hb.q0, hb.q0, 0, 2, 4, 6, 8, 10, 12);
                const HarmEntry h1 = HB_ENTRY(hb.q0, hb.q1, hb.q1, hb.q1, hb.q1, hb.q1, hb.q1, 14, 0, 2, 4, 6, 8, 10);
                const HarmEntry h2 = HB_ENTRY(hb.q1, hb.q1, hb.q2, hb.q2, hb.q2, hb.q2, hb.q2, 12, 14, 0, 2, 4, 6, 8);
                const HarmEntry h3 = HB_ENTRY(hb.q2, hb.q2, hb.q2, hb.q3, hb.q3, hb.q3, hb.q3, 10, 12, 14, 0, 2, 4, 6);
                HARM_TERM(h0)
                if (rem > 1) {
                    HARM_TERM(h1)
                    if (rem > 2) {
                        HARM_TERM(h2)
                        if (rem > 3) HARM_TERM(h3)
                    }
                }
            }
            if (TOUCH_AHEAD) touch_done(sink);
            const T sc = rho * hd.scale;  // rho * c * sqrt(2)
            px = gfma(sc, gfma(rc, s1, gmul(ic, s2)), px);
            py = gfma(sc, gfma(rc, s2, -(gmul(ic, s1))), py);
            pz = gfma(rho, gfma(rc, s3, gmul(ic, s4)), pz);
            pw = pw - gfma(rc, s5, gmul(ic, s6));
            const T t = gmul(rc, zr) - gmul(ic, zi);
            ic = gmul(rc, zi) + gmul(ic, zr);
            rc = t;
            hd = hn;
        }
    }
    Partial4T<T> r = {px, py, pz, pw};
    return r;
}


// ---------------------------------------------------------------------------------------------
// Hybrid feed of the column recursion (devcfg.h HYB_*): {g, t1, t2} of eight rows through three scalar loads behind one
// wait, t3..t6 of sixteen rows in four VGPR pairs (one coalesced 128-byte load each: lane e of every 16-lane row holds row e)
// and picked by the DPP row_newbcast of v_fmac_f64.  Per row: 24 scalar bytes instead of 56, the same nine f64 operations on
// the same operands in the same order (v_fmac_f64 IS fma(src0, src1, dst)): bit-identical to the scalar stream.
// Hazards (GCNHazardRecognizer does not look inside inline asm): a VALU write of a VGPR needs two wait states before a DPP
// read of it, a VALU write of EXEC five - the DPP operand registers are written by VMEM loads only, EXEC is never written;
// tools/check_dpp_hazards.py scans the code object.
#include "harm_stream_asm.h"

// The column shares of one wave (schedule `sched`) over the hybrid stream: per range of consecutive columns ONE pass of the
// generated loop (tools/gen_harm_stream.py -> harm_stream_asm.h).  Out of line like harmonics_partial, and on its own (the
// scalar loop must not carry this one's registers: with both in one function the callers' save / restore cost 6 % of the run).
static __device__ __attribute__((noinline)) Partial4 harmonics_stream(uint64_t cfg_u, uint64_t cols_u, int wave_v, int sched_v, double zr,
                                                                    double zi, double rho_u, double rho, double inv_rho) {
    CfgPtr cfg = (CfgPtr)uniform_u64(cfg_u);
    ColPtr cols = (ColPtr)uniform_u64(cols_u);
    const int wave = __builtin_amdgcn_readfirstlane(wave_v);
    const int sched = __builtin_amdgcn_readfirstlane(sched_v);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    double px = 0.0, py = 0.0, pz = 0.0, pw = 0.0;
    const double rho2 = rho * rho;
    const CAS DevSched &sd = cfg->sched[sched];
    const int nr = sd.n_ranges[wave];
    uint64_t hs0 = cfg->hyb, hv0 = cfg->hyb_v;
    {   // (uniform) the run stream of this schedule, every range at the head of a group of its own (DevCfg.rs_*)
        const int rs = sched == DEV_SCHED_SOLO ? 0 : (sched == DEV_SCHED_PRIMARY ? 1 : ((sched == DEV_SCHED_HELPER || sched == DEV_SCHED_HELPER2) ? 2 : -1));
        if (rs >= 0 && cfg->rs_hyb[rs >= 0 ? rs : 0] != 0) {
            hs0 = cfg->rs_hyb[rs]; hv0 = cfg->rs_hyb_v[rs];
            cols = (ColPtr)cfg->rs_cols[rs];
        }
    }
    const int voff = (lane & 15) * 8;
    for (int q = 0; q < nr; ++q) {
        const int c0 = sd.range_c0[wave][q];
        int cols_left = sd.range_cnt[wave][q];
        const int srow = cols[c0].start;  // stream row of the range's first row (= its index in the entry table)
        // start at the batch that holds it; the rows in front of it (the previous column's last ones) run through the recursion
        // with a zero state - every sum stays an exact zero - and are dropped when the first column is started
        const uint64_t e = hs0 + (uint64_t)(srow & ~7) * (HYB_KS * 8);
        const uint64_t vp = hv0 + (uint64_t)(srow >> 4) * (HYB_GROUP * 8);
        const uint64_t hp = (uint64_t)cols + (uint64_t)c0 * sizeof(ColHdr);
        double rc, ic;
        cpow_uniform(zr, zi, c0 - 1, rc, ic);
        int left = srow & 7, first = 1, sink;
        const int low_half = (srow & 8) == 0 ? 1 : 0;
        HARM_STREAM_ASM(e, vp, hp, voff, left, cols_left, first, low_half, sink, rho_u, rho2, rho, inv_rho, zr, zi, px, py, pz, pw, rc, ic);
        (void)sink;
    }
    Partial4 r = {px, py, pz, pw};
    return r;
}

static __device__ __attribute__((noinline)) Partial4 harmonics_partial(uint64_t cfg_u, uint64_t htab_u, uint64_t cols_u, int wave_v,
                                                                     int sched_v, double zr, double zi, double rho_u, double rho,
                                                                     double inv_rho) {
    CfgPtr cfg = (CfgPtr)uniform_u64(cfg_u);
    HarmPtr htab = (HarmPtr)uniform_u64(htab_u);
    ColPtr cols = (ColPtr)uniform_u64(cols_u);
    const int wave = __builtin_amdgcn_readfirstlane(wave_v);
    const int sched = __builtin_amdgcn_readfirstlane(sched_v);
    return harmonics_core<double>(cfg, htab, cols, wave, sched, zr, zi, rho_u, rho, inv_rho);
}

// The second gravity field of a configuration (nyx_hip_config_t.gravity2; GravityField::eom, gravity_field.rs:148-268, a second time):
// walked in one piece by the perturbation wave that has the point-mass share, beside the column waves of the first field - the same
// recursion (harmonics_partial over the schedule DEV_SCHED_SECOND = every column of the second table), the epilogue of phase C
// ((mu / r) / R_eq, the s, t, u terms, rotation back), its own DCM evaluated here (epoch-only, but this wave is not the critical path),
// the position translated to the field's body when that is not the integration centre.  Added to the point-mass rows.
// Returns the status of the field's own orientation (a binary PCK whose coverage the epoch has left: the record is clamped, the
// DCM is wrong, and neither the first field nor the bodies need share that segment): the caller leaves it in the stage's status
// row for the integrator wave, as the almanac waves do with theirs.
static __device__ __attribute__((noinline)) int second_field_into_pert(CfgPtr cfg, const double *records, const double *ed, int lane, int wave,
                                                                      double et_s, const double *ys, double *pert) {
    double r[3] = {ys[0 * DEV_LANES + lane], ys[1 * DEV_LANES + lane], ys[2 * DEV_LANES + lane]};
    if (cfg->g2_slot >= 0) {  // (uniform)
        double pg[3];
        ed_body(cfg, ed, lane, cfg->g2_slot, pg);
        r[0] = r[0] - pg[0]; r[1] = r[1] - pg[1]; r[2] = r[2] - pg[2];
    }
    double m[9];
    const int st = rotation_dcm(cfg, cfg->g2_rot, records, et_s, m);
    const double rb0 = m[0] * r[0] + m[1] * r[1] + m[2] * r[2];
    const double rb1 = m[3] * r[0] + m[4] * r[1] + m[5] * r[2];
    const double rb2 = m[6] * r[0] + m[7] * r[1] + m[8] * r[2];
    const double r_ = norm3(rb0, rb1, rb2);
    const double inv_r = 1.0 / r_;
    const double s_ = rb0 * inv_r, t_ = rb1 * inv_r, u_ = rb2 * inv_r;
    const double rho = cfg->g2_re * inv_r;
    const double kfac = (cfg->g2_mu * inv_r) * cfg->g2_inv_re;
    const Partial4 pr = harmonics_partial((uint64_t)cfg, cfg->htab2, cfg->cols2, wave, DEV_SCHED_SECOND, rho * s_, rho * t_, rho * u_, rho,
                                          r_ * cfg->g2_inv_re);
    const double px = pr.x * kfac, py = pr.y * kfac, pz = pr.z * kfac, pw = pr.w * kfac;
    const double al0 = px + pw * s_, al1 = py + pw * t_, al2 = pz + pw * u_;
    pert[0 * DEV_LANES + lane] = pert[0 * DEV_LANES + lane] + (m[0] * al0 + m[3] * al1 + m[6] * al2);
    pert[1 * DEV_LANES + lane] = pert[1 * DEV_LANES + lane] + (m[1] * al0 + m[4] * al1 + m[7] * al2);
    pert[2 * DEV_LANES + lane] = pert[2 * DEV_LANES + lane] + (m[2] * al0 + m[5] * al1 + m[8] * al2);
    return st;
}

// GravityField::gradient (gravity_field.rs:273-431) of the SECOND field, for the 64-lane dual (D3) layout of the STM kernel: the same
// frame handling as eom (:279-283: translate to the field's body, rotate; the translation carries no partials), duals seeded on the
// body-fixed position (hyperspace_from_vector, :285), the column recursion on value + three partials over every column of the second
// table, the epilogue of phase C in duals, a = R^T a_bf and G = R^T G_bf R (:403-430).  Added to the point-mass rows of the dual
// perturbation block (a_pm, G_pm: the integrator adds them after the two-body term, like the first field's - the order of two terms
// of a sum).  Returns the status of the field's own orientation.
static __device__ __attribute__((noinline)) int second_field_into_pertD(CfgPtr cfg, const double *records, const double *ed, int lane, int wave,
                                                                       double et_s, const double *ys, double *pertD) {
    double r[3] = {ys[0 * DEV_LANES + lane], ys[1 * DEV_LANES + lane], ys[2 * DEV_LANES + lane]};
    if (cfg->g2_slot >= 0) {  // (uniform)
        double pg[3];
        ed_body(cfg, ed, lane, cfg->g2_slot, pg);
        r[0] = r[0] - pg[0]; r[1] = r[1] - pg[1]; r[2] = r[2] - pg[2];
    }
    double m[9];
    const int st = rotation_dcm(cfg, cfg->g2_rot, records, et_s, m);
    const D3 x0 = {m[0] * r[0] + m[1] * r[1] + m[2] * r[2], 1.0, 0.0, 0.0};
    const D3 x1 = {m[3] * r[0] + m[4] * r[1] + m[5] * r[2], 0.0, 1.0, 0.0};
    const D3 x2 = {m[6] * r[0] + m[7] * r[1] + m[8] * r[2], 0.0, 0.0, 1.0};
    const D3 rD = d3norm(x0, x1, x2);
    const D3 sD = d3div(x0, rD), tD = d3div(x1, rD), uD = d3div(x2, rD);
    const D3 rhoD = d3div(d3c(cfg->g2_re), rD);
    const D3 kD = d3div(d3div(d3c(cfg->g2_mu), rD), d3c(cfg->g2_re));
    const D3 invD = rD * cfg->g2_inv_re;
    // (arguments arrive in VGPRs under the device-function ABI: the wave-uniform ones are re-scalarised, as in harmonics_partial)
    CfgPtr cfg_s = (CfgPtr)uniform_u64((uint64_t)cfg);
    Partial4T<D3> pd = harmonics_core<D3>(cfg_s, (HarmPtr)uniform_u64(cfg_s->htab2), (ColPtr)uniform_u64(cfg_s->cols2), __builtin_amdgcn_readfirstlane(wave),
                                          DEV_SCHED_SECOND, rhoD * sD, rhoD * tD, rhoD * uD, rhoD, invD);
    const D3 p0 = pd.x * kD, p1 = pd.y * kD, p2 = pd.z * kD, p3 = pd.w * kD;
    const D3 al[3] = {p0 + p3 * sD, p1 + p3 * tD, p2 + p3 * uD};
    double tmp[9];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pertD[a * DEV_LANES + lane] = pertD[a * DEV_LANES + lane] + (m[0 + a] * al[0].v + m[3 + a] * al[1].v + m[6 + a] * al[2].v);
        tmp[3 * a + 0] = m[0 + a] * al[0].x + m[3 + a] * al[1].x + m[6 + a] * al[2].x;
        tmp[3 * a + 1] = m[0 + a] * al[0].y + m[3 + a] * al[1].y + m[6 + a] * al[2].y;
        tmp[3 * a + 2] = m[0 + a] * al[0].z + m[3 + a] * al[1].z + m[6 + a] * al[2].z;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            pertD[(3 + 3 * a + b) * DEV_LANES + lane] =
                pertD[(3 + 3 * a + b) * DEV_LANES + lane] + (tmp[3 * a + 0] * m[0 + b] + tmp[3 * a + 1] * m[3 + b] + tmp[3 * a + 2] * m[6 + b]);
    return st;
}

// Dual variant: inputs and outputs go through LDS (20 + 16 doubles per lane) instead of the register ABI.
static __device__ __attribute__((noinline)) void harmonics_partial_dual(uint64_t cfg_u, uint64_t htab_u, uint64_t cols_u, int wave_v,
                                                                      const double *inbD, double *outD, int lane) {
    CfgPtr cfg = (CfgPtr)uniform_u64(cfg_u);
    HarmPtr htab = (HarmPtr)uniform_u64(htab_u);
    ColPtr cols = (ColPtr)uniform_u64(cols_u);
    const int wave = __builtin_amdgcn_readfirstlane(wave_v);
    D3 in[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        in[q].v = inbD[(4 * q + 0) * DEV_LANES + lane]; in[q].x = inbD[(4 * q + 1) * DEV_LANES + lane];
        in[q].y = inbD[(4 * q + 2) * DEV_LANES + lane]; in[q].z = inbD[(4 * q + 3) * DEV_LANES + lane];
    }
    const Partial4T<D3> pd = harmonics_core<D3>(cfg, htab, cols, wave, DEV_SCHED_SOLO, in[0], in[1], in[2], in[3], in[4]);
    const D3 o4[4] = {pd.x, pd.y, pd.z, pd.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        outD[(4 * q + 0) * DEV_LANES + lane] = o4[q].v; outD[(4 * q + 1) * DEV_LANES + lane] = o4[q].x;
        outD[(4 * q + 2) * DEV_LANES + lane] = o4[q].y; outD[(4 * q + 3) * DEV_LANES + lane] = o4[q].z;
    }
}

// Quad layout (D1): 5 inputs of (value, this lane's partial) in, 4 partial sums out, through LDS.  The slot of a wave is
// QSLOT doubles: [4 sums][64] partials, then [4 sums][16] values (the value is the same in the four lanes of a quad).
#define QSLOT (4 * DEV_LANES + 4 * (DEV_LANES / 4))
typedef __attribute__((address_space(3))) double *LdsPtr;
static __device__ __attribute__((noinline)) void harmonics_partial_d1(uint64_t cfg_u, uint64_t htab_u, uint64_t cols_u, int wave_v,
                                                                    LdsCPtr inbQ, LdsPtr outQ, int lane, LdsFlagPtr gate, int need_v) {
    CfgPtr cfg = (CfgPtr)uniform_u64(cfg_u);
    HarmPtr htab = (HarmPtr)uniform_u64(htab_u);
    ColPtr cols = (ColPtr)uniform_u64(cols_u);
    const int wave = __builtin_amdgcn_readfirstlane(wave_v);
    D1 in[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) { in[q].v = inbQ[(2 * q + 0) * DEV_LANES + lane]; in[q].d = inbQ[(2 * q + 1) * DEV_LANES + lane]; }
    const Partial4T<D1> pd = harmonics_core<D1>(cfg, htab, cols, wave, DEV_SCHED_SOLO, in[0], in[1], in[2], in[3], in[4]);
    const D1 o4[4] = {pd.x, pd.y, pd.z, pd.w};
    // pipelined stage loop: the slot still holds the previous stage's sums until the integrator wave has folded them
    const int need = __builtin_amdgcn_readfirstlane(need_v);
    if (need > 0) {
        int spin = 0;
        while (*gate < need && ++spin < 4000000) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { outQ[q * DEV_LANES + lane] = o4[q].d; outQ[4 * DEV_LANES + q * (DEV_LANES / 4) + (lane >> 2)] = o4[q].v; }
}

// The five inputs of the column recursion as one-partial duals of the body-fixed position (quad layout), into `dst` [10][64].
DEVFN void publish_d1_inputs(CfgPtr cfg, double rb0, double rb1, double rb2, int ql, double *dst, int lane) {
    const D1 x0 = d1seed(rb0, 0, ql), x1 = d1seed(rb1, 1, ql), x2 = d1seed(rb2, 2, ql);
    const D1 rD = d1norm(x0, x1, x2);
    const D1 sD = d1div(x0, rD), tD = d1div(x1, rD), uD = d1div(x2, rD);
    const D1 rhoD = d1div(d1c(cfg->g_re), rD);
    const D1 invD = rD * cfg->g_inv_re;
    const D1 pub[5] = {rhoD * sD, rhoD * tD, rhoD * uD, rhoD, invD};
#pragma unroll
    for (int q = 0; q < 5; ++q) { dst[(2 * q + 0) * DEV_LANES + lane] = pub[q].v; dst[(2 * q + 1) * DEV_LANES + lane] = pub[q].d; }
}

// Quad-lane exchange (DPP quad_perm broadcast of lane SEL of every quad; two 32-bit moves per double).
template <int SEL>
DEVFN double quad_bcast(double x) {
    constexpr int ctrl = SEL | (SEL << 2) | (SEL << 4) | (SEL << 6);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
DEVFN int quad_or(int x) {
    x |= __builtin_amdgcn_mov_dpp(x, 0xb1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]
    x |= __builtin_amdgcn_mov_dpp(x, 0x4e, 0xf, 0xf, true);  // quad_perm [2, 3, 0, 1]
    return x;
}

// second_field_into_pertD for the QUAD layout (round 5: the quad layout used to be refused with a second field): the four lanes of a quad
// walk every column of the second table on ONE-partial duals (lane k carries d/dx_k of the body-fixed position; lane 0 the value
// alone), the epilogue is second_field_into_pertD's in D1 - every expression is that function's for the value and for one partial
// slot, so value and gradient are bit-identical to the 64-lane layout -, and R^T G_bf R is formed with the quad exchange of phase_c_quad:
// this lane's column (ql - 1) of G from the three partial lanes' rows.  Added to rows 0..2 (a) and 3..5 (this lane's column of G) of the
// quad layout's perturbation block, i.e. to the point-mass share.  Returns the status of the field's own orientation.
static __device__ __attribute__((noinline)) int second_field_into_pert_q(CfgPtr cfg, const double *records, const double *ed, int lane, int ql, int wave,
                                                                        double et_s, const double *ys, double *pertq) {
    double r[3] = {ys[0 * DEV_LANES + lane], ys[1 * DEV_LANES + lane], ys[2 * DEV_LANES + lane]};
    if (cfg->g2_slot >= 0) {  // (uniform)
        double pg[3];
        ed_body(cfg, ed, lane, cfg->g2_slot, pg);
        r[0] = r[0] - pg[0]; r[1] = r[1] - pg[1]; r[2] = r[2] - pg[2];
    }
    double m[9];
    const int st = rotation_dcm(cfg, cfg->g2_rot, records, et_s, m);
    const D1 x0 = d1seed(m[0] * r[0] + m[1] * r[1] + m[2] * r[2], 0, ql);
    const D1 x1 = d1seed(m[3] * r[0] + m[4] * r[1] + m[5] * r[2], 1, ql);
    const D1 x2 = d1seed(m[6] * r[0] + m[7] * r[1] + m[8] * r[2], 2, ql);
    const D1 rD = d1norm(x0, x1, x2);
    const D1 sD = d1div(x0, rD), tD = d1div(x1, rD), uD = d1div(x2, rD);
    const D1 rhoD = d1div(d1c(cfg->g2_re), rD);
    const D1 kD = d1div(d1div(d1c(cfg->g2_mu), rD), d1c(cfg->g2_re));
    const D1 invD = rD * cfg->g2_inv_re;
    CfgPtr cfg_s = (CfgPtr)uniform_u64((uint64_t)cfg);
    Partial4T<D1> pd = harmonics_core<D1>(cfg_s, (HarmPtr)uniform_u64(cfg_s->htab2), (ColPtr)uniform_u64(cfg_s->cols2), __builtin_amdgcn_readfirstlane(wave),
                                          DEV_SCHED_SECOND, rhoD * sD, rhoD * tD, rhoD * uD, rhoD, invD);
    const D1 p0 = pd.x * kD, p1 = pd.y * kD, p2 = pd.z * kD, p3 = pd.w * kD;
    const D1 al[3] = {p0 + p3 * sD, p1 + p3 * tD, p2 + p3 * uD};
    double tmpc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pertq[a * DEV_LANES + lane] = pertq[a * DEV_LANES + lane] + (m[0 + a] * al[0].v + m[3 + a] * al[1].v + m[6 + a] * al[2].v);
        tmpc[a] = m[0 + a] * al[0].d + m[3 + a] * al[1].d + m[6 + a] * al[2].d;
    }
    const int b = ql > 0 ? ql - 1 : 0;
    const double mb0 = b == 0 ? m[0] : (b == 1 ? m[1] : m[2]), mb1 = b == 0 ? m[3] : (b == 1 ? m[4] : m[5]), mb2 = b == 0 ? m[6] : (b == 1 ? m[7] : m[8]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double t0 = quad_bcast<1>(tmpc[a]), t1 = quad_bcast<2>(tmpc[a]), t2 = quad_bcast<3>(tmpc[a]);
        pertq[(3 + a) * DEV_LANES + lane] = pertq[(3 + a) * DEV_LANES + lane] + (t0 * mb0 + t1 * mb1 + t2 * mb2);
    }
    return st;
}

