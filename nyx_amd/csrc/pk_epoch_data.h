// pk_epoch_data.h - part of propagate_kernel.hip (included there, in this order; not a stand-alone header): epoch-only data of a stage: ephemeris chains (SPK type 2), body-fixed orientations (IAU / binary PCK), the incremental polynomial DCM.
// ---------------------------------------------------------------------------------------------
// Epoch-only data of one stage: body-fixed DCM and body positions
// ---------------------------------------------------------------------------------------------

// LDS slot of one stage's epoch data, per lane: m[9] (DCM inertial -> body-fixed, row-major) then
// bp[DEV_MAX_SLOTS][3] (slot positions w.r.t. the integration centre).  Field-major: slot[f * 64 + lane].
#define ED_FIELDS (9 + 3 * DEV_MAX_SLOTS)

// SPK type 2 evaluation (Clenshaw); record index is per lane, metadata is uniform.  `records` is the
// LDS copy of the segment table when it fits (cfg->rec_in_lds), else the global array.  The 16-wide
// coefficient window is loaded before the recurrence starts (the table is padded by 16 doubles), so the
// loads are independent of the serial w0/w1/w2 chain.  Segments with more than CHEB_MAXC coefficients take a rolled loop.
#define CHEB_MAXC 16
template <typename P>
DEVFN int cheby_eval(const CAS DevSeg &sg, P records, double et_s, double *r3) {
    const double rel = (et_s - sg.init_et) / sg.interval;
    int idx = (int)floor(rel);
    int st = NYX_HIP_OK;
    if (idx < 0 || idx > sg.n_rec || (idx == sg.n_rec && et_s > sg.end_et)) st = NYX_HIP_ERR_EPHEM_RANGE;
    idx = idx < 0 ? 0 : (idx >= sg.n_rec ? sg.n_rec - 1 : idx);
    const int nc = sg.n_coef;
    const int cs = (sg.stride - 2) / 3;  // doubles per component: n_coef, or CHEB_MAXC when the host padded the record with zeros (below)
    P rec = records + sg.offset + idx * sg.stride;
    const double t = (et_s - rec[0]) / rec[1];
    const double two_t = 2.0 * t;
    if (nc > CHEB_MAXC) {  // (uniform) up to NYX_HIP_MAX_CHEBY_COEFFS: the tail beyond the 16-wide register window is walked first,
        // coefficient by coefficient from the table - the same recurrence in the same order (DE440's Mercury / Sun segments, binary PCKs)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            P cf = rec + 2 + c * cs;
            double w0 = 0.0, w1 = 0.0, w2;
            for (int j = nc - 1; j >= 1; --j) {
                w2 = w1;
                w1 = w0;
                w0 = cf[j] + (two_t * w1 - w2);
            }
            r3[c] = cf[0] + (t * w0 - w1);
        }
        return st;
    }
    if (cs == CHEB_MAXC) {
        // (uniform) the host laid the record out sixteen-wide, the coefficients past the segment's count being +0.0 IN THE TABLE: the
        // selects below (two v_cndmask per coefficient, a quarter of this function's instructions) are not needed - same values, same bits
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            P cf = rec + 2 + c * CHEB_MAXC;
            double cv[CHEB_MAXC];
#pragma unroll
            for (int j = 0; j < CHEB_MAXC; ++j) cv[j] = cf[j];
            double w0 = 0.0, w1 = 0.0, w2;
#pragma unroll
            for (int j = CHEB_MAXC - 1; j >= 1; --j) {
                w2 = w1;
                w1 = w0;
                w0 = cv[j] + (two_t * w1 - w2);
            }
            r3[c] = cv[0] + (t * w0 - w1);
        }
        return st;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        P cf = rec + 2 + c * nc;
        double cv[CHEB_MAXC];
#pragma unroll
        for (int j = 0; j < CHEB_MAXC; ++j) cv[j] = cf[j];
        // Coefficients past the segment's own count are taken as +0.0 and every one of the fifteen steps runs: a step with a zero
        // coefficient and w0 = w1 = +0 leaves +0 (0 + (2t * 0 - 0) = +0 for either sign of t), so the chain reaches j = nc - 1 in the
        // state it would start from - the same bits as skipping those steps - while the (uniform) `j < nc` selects sit on the loads,
        // not on the serial w0 / w1 / w2 chain (guarding the steps cost six v_cndmask per step there: two thirds of this function).
#pragma unroll
        for (int j = 1; j < CHEB_MAXC; ++j) cv[j] = (j < nc) ? cv[j] : 0.0;
        double w0 = 0.0, w1 = 0.0, w2;
#pragma unroll
        for (int j = CHEB_MAXC - 1; j >= 1; --j) {
            w2 = w1;
            w1 = w0;
            w0 = cv[j] + (two_t * w1 - w2);
        }
        r3[c] = cv[0] + (t * w0 - w1);
    }
    return st;
}

// SPK type 2 with the derivative (the integration-frame swap needs the velocity of a chain): value as cheby_eval, derivative by
// the companion recurrence of SPICE's CHBINT, dW_j = 2 W_{j+1} + 2t dW_{j+1} - dW_{j+2}, scaled by 1 / radius.
template <typename P>
DEVFN int cheby_eval_pv(const CAS DevSeg &sg, P records, double et_s, double *r3, double *v3) {
    const double rel = (et_s - sg.init_et) / sg.interval;
    int idx = (int)floor(rel);
    int st = NYX_HIP_OK;
    if (idx < 0 || idx > sg.n_rec || (idx == sg.n_rec && et_s > sg.end_et)) st = NYX_HIP_ERR_EPHEM_RANGE;
    idx = idx < 0 ? 0 : (idx >= sg.n_rec ? sg.n_rec - 1 : idx);
    const int nc = sg.n_coef;
    P rec = records + sg.offset + idx * sg.stride;
    const double t = (et_s - rec[0]) / rec[1];
    const double two_t = 2.0 * t;
    const int cs = (sg.stride - 2) / 3;  // (component stride: see cheby_eval)
    for (int c = 0; c < 3; ++c) {
        P cf = rec + 2 + c * cs;
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
    return st;
}

struct FrameChain {
    int32_t n_chain, seg[4];
    double sign[4];
};
#if NYX_HOST_TU
// opts.integration_frame (instance.rs:117-142, 211-220): x += dir * (state of the chain's body w.r.t. the integration centre at the
// trajectory's epoch); dir = +1 into the integration frame, -1 back.  One thread per trajectory.
__global__ __launch_bounds__(256) void nyx_frame_shift_kernel(const DevCfg *cfg_g, const double *records, FrameChain ch, int64_t n,
                                                              const int64_t *epoch_ns, double *x, double *y, double *z, double *vx,
                                                              double *vy, double *vz, double dir, int32_t *status, const int32_t *prior,
                                                              const int64_t *dur_ns) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // per-trajectory durations (the covariance-mapping loop): a run that has reached its end is not propagated by the reference any
    // more - it is not translated either ((x + b) - b is not x)
    if (dur_ns && dur_ns[i] == 0) return;
    CfgPtr cfg = (CfgPtr)cfg_g;
    const double et = ns_to_seconds(epoch_ns[i]);
    double b[3] = {0.0, 0.0, 0.0}, bv[3] = {0.0, 0.0, 0.0};
    int st = NYX_HIP_OK;
    for (int k = 0; k < ch.n_chain; ++k) {
        double p[3], v[3];
        const int s1 = cheby_eval_pv(cfg->seg[ch.seg[k]], records, et, p, v);
        if (s1) st = s1;
        for (int c = 0; c < 3; ++c) { b[c] = b[c] + ch.sign[k] * p[c]; bv[c] = bv[c] + ch.sign[k] * v[c]; }
    }
    x[i] = x[i] + dir * b[0]; y[i] = y[i] + dir * b[1]; z[i] = z[i] + dir * b[2];
    vx[i] = vx[i] + dir * bv[0]; vy[i] = vy[i] + dir * bv[1]; vz[i] = vz[i] + dir * bv[2];
    if (prior && prior[i] != NYX_HIP_OK) st = prior[i];  // (the translation INTO the integration frame had failed already)
    if (st && status && status[i] == NYX_HIP_OK) status[i] = st;
}
extern "C" hipError_t nyx_launch_frame_shift(const DevCfg *cfg, const double *records, const int32_t *chain_seg, const double *chain_sign,
                                             int n_chain, int64_t n, const int64_t *epoch_ns, double *x, double *y, double *z, double *vx,
                                             double *vy, double *vz, double dir, int32_t *status, const int32_t *prior, const int64_t *dur_ns,
                                             hipStream_t stream) {
    FrameChain ch;
    ch.n_chain = n_chain;
    for (int k = 0; k < 4; ++k) { ch.seg[k] = k < n_chain ? chain_seg[k] : 0; ch.sign[k] = k < n_chain ? chain_sign[k] : 0.0; }
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(nyx_frame_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cfg, records, ch, n, epoch_ns, x, y, z,
                       vx, vy, vz, dir, status, prior, dur_ns);
    return hipGetLastError();
}
#endif

// Body-fixed orientation (nyx_hip_rotation_t, see include/nyx_hip.h): the IAU phase angles with their trigonometric series, or
// the Chebyshev Euler angles of a binary PCK.  `w_rate` (optional): dW/dt in rad/s (the drag model's velocity transform).
DEVFN void dcm_from_sincos(double s1, double c1, double s2, double c2, double s3, double c3, double *m);
DEVFN void r3r1r3(double a1, double a2, double a3, double *m) {
    double s1, c1, s2, c2, s3, c3;
    sincos(a1, &s1, &c1);
    sincos(a2, &s2, &c2);
    sincos(a3, &s3, &c3);
    dcm_from_sincos(s1, c1, s2, c2, s3, c3, m);
}
DEVFN void dcm_from_sincos(double s1, double c1, double s2, double c2, double s3, double c3, double *m) {
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
template <typename P>
DEVFN int rotation_dcm(CfgPtr cfg, const CAS DevRot &rot, P records, double et_s, double *m, double *w_rate = nullptr) {
    const double DEG = 3.14159265358979323846 / 180.0;
    const double HALF_PI = 1.57079632679489661923;
    if (rot.kind == NYX_HIP_ROT_EULER_CHEBY) {  // (uniform)
        const CAS DevSeg &sg = cfg->seg[rot.euler_seg];
        double ang[3];
        const int st = cheby_eval(sg, records, et_s, ang);
        double e[9];
        r3r1r3(ang[0], ang[1], ang[2], e);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) m[3 * i + j] = e[3 * i + 0] * rot.base[0 + j] + e[3 * i + 1] * rot.base[3 + j] + e[3 * i + 2] * rot.base[6 + j];
        if (w_rate) {  // derivative of the third angle's series: sum c_j T_j'(t) / radius
            int idx = (int)floor((et_s - sg.init_et) / sg.interval);
            idx = idx < 0 ? 0 : (idx >= sg.n_rec ? sg.n_rec - 1 : idx);
            P rec = records + sg.offset + idx * sg.stride;
            const double t = (et_s - rec[0]) / rec[1];
            P cf = rec + 2 + 2 * ((sg.stride - 2) / 3);
            double tjm1 = 1.0, tj = t, djm1 = 0.0, dj = 1.0, acc = 0.0;
            for (int j = 1; j < sg.n_coef; ++j) {
                acc = acc + cf[j] * dj;
                const double tn = 2.0 * t * tj - tjm1;
                const double dn = 2.0 * tj + 2.0 * t * dj - djm1;
                tjm1 = tj; tj = tn; djm1 = dj; dj = dn;
            }
            *w_rate = acc / rec[1];
        }
        return st;
    }
    const double d = et_s / 86400.0;
    const double T = et_s / (86400.0 * 36525.0);
    double ra = rot.ra[0] + rot.ra[1] * T + rot.ra[2] * T * T;
    double dec = rot.dec[0] + rot.dec[1] * T + rot.dec[2] * T * T;
    double w = rot.w[0] + rot.w[1] * d + rot.w[2] * d * d;
    double wd = rot.w[1] + 2.0 * rot.w[2] * d;
    const int np = rot.n_np;
    for (int k = 0; k < np; ++k) {
        const double th = (rot.np_ang[k][0] + rot.np_ang[k][1] * T) * DEG;
        double sn, cs;
        sincos(th, &sn, &cs);
        ra = ra + rot.np_ra[k] * sn;
        dec = dec + rot.np_dec[k] * cs;
        w = w + rot.np_w[k] * sn;
        wd = wd + rot.np_w[k] * cs * (rot.np_ang[k][1] * DEG / 36525.0);
    }
    r3r1r3(HALF_PI + ra * DEG, HALF_PI - dec * DEG, w * DEG, m);
    if (w_rate) *w_rate = wd * DEG / 86400.0;
    return NYX_HIP_OK;
}

// ---- IAU orientation advanced from a base epoch (almanac wave, per lane) -------------------------------------------------------
// A body whose pole and prime meridian are POLYNOMIALS of time (no trigonometric terms: the Earth of the IAU reports) is rotated
// by three angles that move by less than 0.1 rad within a quarter of an hour.  rotation_dcm() pays three full-range sincos per
// stage for that (arguments of ~5e4 rad: ~600 instructions on the almanac wave, a quarter of its duty).  Here the sines and cosines
// are computed at the nearest point of a fixed 2 048 s grid of epochs (the same expressions, the same bits as rotation_dcm there)
// and advanced to the stage epoch by the angle-addition formulas with the increment's own short series:
//     delta = p(t0 + tau) - p(t0) = (p1 + p2 (2 t0 + tau)) tau          (tau = the integer-ns epoch difference: exact)
//     sin(a0 + delta) = s0 cos(delta) + c0 sin(delta),  |delta| < 0.25:  sin to delta^13, cos to delta^14  (< 3e-18)
// The base is a function of the lane's own epoch alone (its grid point), renewed per lane when the epoch moves to another grid
// point: a trajectory's bits do not depend on which lanes share its wave (tuning.deterministic, the quad / 64-lane STM layouts).
// Against rotation_dcm() the angles differ by the rounding of the LARGE argument there (ulp(3e6 deg) = 8e-12 rad), not by anything
// this formulation adds: a change of summation-order size (0.06 mm on the Earth's surface), inside every parity bar.  Plain
// kernels only (the STM tests compare step sequences with the oracle bit for bit); tuning.debug_flags 0x4000 switches it off.
#define ROT_GRID_NS (2048LL * 1000000000LL)
struct RotBase {
    int64_t ep;  // the grid epoch the sines and cosines belong to (INT64_MIN: none yet)
    double sn[3], cs[3];
};
DEVFN void small_sincos(double d, double &sn, double &cs) {  // |d| < 0.25
    const double z = d * d;
    double p = __builtin_fma(z, 1.0 / 6227020800.0, -1.0 / 39916800.0);
    p = __builtin_fma(z, p, 1.0 / 362880.0);
    p = __builtin_fma(z, p, -1.0 / 5040.0);
    p = __builtin_fma(z, p, 1.0 / 120.0);
    p = __builtin_fma(z, p, -1.0 / 6.0);
    sn = __builtin_fma(d * z, p, d);
    double q = __builtin_fma(z, -1.0 / 87178291200.0, 1.0 / 479001600.0);
    q = __builtin_fma(z, q, -1.0 / 3628800.0);
    q = __builtin_fma(z, q, 1.0 / 40320.0);
    q = __builtin_fma(z, q, -1.0 / 720.0);
    q = __builtin_fma(z, q, 1.0 / 24.0);
    q = __builtin_fma(z, q, -0.5);
    cs = __builtin_fma(z, q, 1.0);
}
DEVFN void iau_poly_angles(const CAS DevRot &rot, double et_s, double *a) {  // rotation_dcm's expressions (n_np == 0), radians
    const double DEG = 3.14159265358979323846 / 180.0;
    const double HALF_PI = 1.57079632679489661923;
    const double d = et_s / 86400.0;
    const double T = et_s / (86400.0 * 36525.0);
    const double ra = rot.ra[0] + rot.ra[1] * T + rot.ra[2] * T * T;
    const double dec = rot.dec[0] + rot.dec[1] * T + rot.dec[2] * T * T;
    const double w = rot.w[0] + rot.w[1] * d + rot.w[2] * d * d;
    a[0] = HALF_PI + ra * DEG; a[1] = HALF_PI - dec * DEG; a[2] = w * DEG;
}
DEVFN void rotation_dcm_iau_poly(const CAS DevRot &rot, int64_t epoch_ns, RotBase &rb, double *m) {
    const double DEG = 3.14159265358979323846 / 180.0;
    // nearest grid point (floor division: epochs before J2000 are negative)
    const int64_t sh = epoch_ns + ROT_GRID_NS / 2;
    const int64_t grid = (sh >= 0 ? sh / ROT_GRID_NS : -((-sh + ROT_GRID_NS - 1) / ROT_GRID_NS)) * ROT_GRID_NS;
    if (grid != rb.ep) {  // (per lane: usually the whole wave crosses a grid boundary within a few stages of each other)
        double a[3];
        iau_poly_angles(rot, ns_to_seconds(grid), a);
        sincos(a[0], &rb.sn[0], &rb.cs[0]);
        sincos(a[1], &rb.sn[1], &rb.cs[1]);
        sincos(a[2], &rb.sn[2], &rb.cs[2]);
        rb.ep = grid;
    }
    const double et0 = ns_to_seconds(grid);
    const double tau = ns_to_seconds(epoch_ns - grid);
    const double dd = tau / 86400.0, dT = tau / (86400.0 * 36525.0);
    const double day0 = et0 / 86400.0, T0 = et0 / (86400.0 * 36525.0);
    const double dl[3] = {((rot.ra[1] + rot.ra[2] * (2.0 * T0 + dT)) * dT) * DEG, -(((rot.dec[1] + rot.dec[2] * (2.0 * T0 + dT)) * dT) * DEG),
                          ((rot.w[1] + rot.w[2] * (2.0 * day0 + dd)) * dd) * DEG};
    double s[3], c[3];
    if (fabs(dl[0]) < 0.25 && fabs(dl[1]) < 0.25 && fabs(dl[2]) < 0.25) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double sd, cd;
            small_sincos(dl[k], sd, cd);
            s[k] = __builtin_fma(rb.sn[k], cd, rb.cs[k] * sd);
            c[k] = __builtin_fma(rb.cs[k], cd, -(rb.sn[k] * sd));
        }
    } else {  // (a rotator too fast for the grid: the full-range evaluation, per lane)
        double a[3];
        iau_poly_angles(rot, ns_to_seconds(epoch_ns), a);
#pragma unroll
        for (int k = 0; k < 3; ++k) sincos(a[k], &s[k], &c[k]);
    }
    dcm_from_sincos(s[0], c[0], s[1], c[1], s[2], c[2], m);
}

template <typename P>
// `dcm_flag` (pipelined stage loop): LDS word that is set to `dcm_val` as soon as the DCM is written - the integrator wave
// needs only that to form the next stage's recursion inputs, the body positions are for the next window.
// `amask`: the share of this almanac wave when the duty is dealt over several (role fan-out, DEV_ROLE_DCM = the DCM, bit s =
// body slot s); every wave writes only its own rows of `slot`.
// `gate` / `gate_val` (INTEG_OOL): the DCM rows of `slot` still hold the orientation of two stages ago, which the integrator's phase C
// reads late (behind the stage barrier, see integ_back) - they are not overwritten before *gate >= gate_val (the fold counter).
DEVFN int epoch_data(CfgPtr cfg, P records, int64_t epoch_ns, double *slot, int lane, int amask, LdsFlagPtr dcm_flag = nullptr, int dcm_val = 0,
                     RotBase *rbase = nullptr, LdsFlagPtr gate = nullptr, int gate_val = 0) {
    const double et = ns_to_seconds(epoch_ns);
    int status = NYX_HIP_OK;
    if ((amask & DEV_ROLE_DCM) && (cfg->has_grav || cfg->has_drag || cfg->has_tides)) {  // (ctx_create requires these body-fixed frames to coincide)
        double m[9];
        int st;
        if (rbase && cfg->dcm_incr) {  // (uniform; the host sets dcm_incr for a polynomial IAU orientation of the frame this wave rotates into)
            rotation_dcm_iau_poly(cfg->has_grav ? cfg->g_rot : (cfg->has_drag ? cfg->d_rot : cfg->t_rot), epoch_ns, *rbase, m);
            st = NYX_HIP_OK;
        } else
        if (cfg->has_grav) st = rotation_dcm(cfg, cfg->g_rot, records, et, m);
        else if (cfg->has_drag) st = rotation_dcm(cfg, cfg->d_rot, records, et, m);
        else st = rotation_dcm(cfg, cfg->t_rot, records, et, m);
        if (st) status = st;
        if (gate) {  // (bounded: a protocol error must end as a failed run, never as a hung GPU)
            int spin = 0;
            while (*gate < gate_val && ++spin < 4000000) __builtin_amdgcn_s_sleep(1);
            if (spin >= 4000000) status = NYX_HIP_ERR_NAN;
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) slot[q * DEV_LANES + lane] = m[q];
    }
    if (dcm_flag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *dcm_flag = dcm_val;
    }
    if (cfg->seg_mode) {  // (uniform) one distinct segment per unit: the chains are summed by the readers, ed_bp()
        const int nu = cfg->n_useg, base = cfg->ed_seg_base;
        // (a ROLLED loop: the five almanac waves of a fan-out workgroup walk the same Chebyshev code instead of five unrolled copies of
        //  it, a single almanac wave one copy four times; round 5, same bits: config 3 44.8 -> 43.7 ms, config 4 9.19 -> 9.08, configs[1] -0.8 %)
#pragma unroll 1
        for (int u = 0; u < DEV_MAX_SEG; ++u) {
            if (u < nu && ((amask >> u) & 1)) {
                double p[3];
                const int st = cheby_eval(cfg->seg[cfg->useg_seg[u]], records, et, p);
                if (st) status = st;
                slot[(base + 3 * u + 0) * DEV_LANES + lane] = p[0];
                slot[(base + 3 * u + 1) * DEV_LANES + lane] = p[1];
                slot[(base + 3 * u + 2) * DEV_LANES + lane] = p[2];
            }
        }
        return status;
    }
    const int ns = cfg->n_slots;
#pragma unroll
    for (int s = 0; s < DEV_MAX_SLOTS; ++s) {
        if (s < ns && ((amask >> s) & 1)) {
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

// Position of body slot s.  Slot mode: the rows epoch_data() wrote.  Segment mode: the chain summed here, in chain order (sign = +-1:
// every product is exact, the additions are those of epoch_data()).
DEVFN void ed_body(CfgPtr cfg, const double *ed, int lane, int s, double *p) {
    if (!cfg->seg_mode) {  // (uniform)
        p[0] = ed[(9 + 3 * s + 0) * DEV_LANES + lane];
        p[1] = ed[(9 + 3 * s + 1) * DEV_LANES + lane];
        p[2] = ed[(9 + 3 * s + 2) * DEV_LANES + lane];
        return;
    }
    const int nch = cfg->slot[s].n_chain, base = cfg->ed_seg_base;
    double b0 = 0.0, b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < nch) {  // (uniform)
            const double sg = cfg->slot[s].sign[k];
            const double *row = ed + (base + 3 * cfg->slot[s].useg[k]) * DEV_LANES + lane;
            b0 = b0 + sg * row[0];
            b1 = b1 + sg * row[DEV_LANES];
            b2 = b2 + sg * row[2 * DEV_LANES];
        }
    }
    p[0] = b0; p[1] = b1; p[2] = b2;
}

