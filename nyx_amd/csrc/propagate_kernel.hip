// propagate_kernel.hip — the MI355X (gfx950) ensemble integrator.
//
// One workgroup integrates 64 trajectories from t0 to tf in a single launch:
//
//   * lane <-> trajectory.  Every lane of every wave of the workgroup is bound to the same
//     trajectory slot, so all shared tables (Stokes coefficients and Legendre recursion constants,
//     column schedule) are WAVE-UNIFORM and are fetched with scalar loads (a batch of five 56-byte
//     harmonics entries = 70 SGPRs behind one wait) straight into SGPRs: the f64 VALU ops take them as
//     scalar operands, no LDS/VGPR traffic for the big table at all.
//   * the waves of the workgroup are ROLE-SPECIALISED (all roles also carry harmonics columns):
//       wave 0  "integrator"    RK state machine of the 64 trajectories: per-lane adaptive step,
//                               accept/reject, integer-ns epoch bookkeeping (reference instance.rs:87-493),
//                               stage combination, two-body term, final accumulation of k_i.
//       wave 1  "almanac"       everything that depends only on the stage EPOCH — body-fixed DCM
//                               (3 sincos), Sun/Moon Chebyshev chains — one stage AHEAD, into LDS.
//       wave 2  "perturbations" position-dependent third-body and SRP/eclipse terms of the current stage.
//       wave 3+ "columns"       spherical-harmonics column workers.
//     Roles only meet in LDS: one workgroup barrier per force evaluation in the pipelined stage loop (the
//     integrator publishes the next stage's position inside the current window; see role_loop), two in the
//     plain one.  The split keeps every code path under 128 VGPRs so that 16 waves (4 per SIMD) fit and
//     hide the scalar-load latency.
//   * when the launch has fewer workgroups than the chip has CUs, HELPER workgroups on the idle CUs take over
//     a share of the harmonics columns through mailboxes in uncached memory ("cooperative mode" below).
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

#include <mutex>

#include "../../include/nyx_hip.h"
#include "devcfg.h"
#include "hifitime_dev.h"
#include "event_dev.h"
#include "predict_args.h"

// which kernels this translation unit emits (see NYX_KERNEL at the end of the file; propagate_*.hip include this file)
#ifndef NYX_EMIT
#define NYX_EMIT 1
#endif
#define NYX_EMIT_PLAIN16 1  /* sixteen waves (the north-star shape), helpers */
#define NYX_EMIT_PLAIN8 2   /* eight waves or fewer */
#define NYX_EMIT_STM 4      /* D3 duals: at most DEV_MAX_WAVES_STM waves (dual-number harmonics need ~4x the registers) */
#define NYX_EMIT_STMQ16 8   /* quad layout (16 trajectories per workgroup, four lanes per trajectory, D1 duals), sixteen waves */
#define NYX_EMIT_STMQ8 16   /* quad layout, eight waves or fewer */
#define NYX_EMIT_PLAIN16_P2 32 /* sixteen waves, cooperative launches whose hand-off has TWO parts (two helper workgroups per owner and evaluation) */
#define NYX_EMIT_PLAIN8N 64   /* eight waves or fewer, dynamics WITHOUT a body-fixed model (no gravity field, drag, tides): NYX_ASSUME_SMALL */
#define NYX_EMIT_PLAIN16_FAN 128 /* sixteen waves, cooperative launches in the FAN-OUT mode (small shards: several dedicated helper workgroups per owner, NYX_COOP_FAN) */
// The in-kernel accounting (tuning.profile / tuning.calibrate: cycle counters per wave and phase, the mailbox counts, the first
// helper's rows) is loop-carried state and s_memtime reads in every role; switched off at run time it still costs the launch
// (measured round 5: 1.3 % of the headline launch, 4.5 % of config 4, 7 % of config 3). Every propagation kernel is therefore
// compiled TWICE from this source: the product kernel without the accounting (NYX_PROF 0) and a twin `<name>_prof` with it
// (__graft_entry__.build compiles each propagate_*.hip a second time with -DNYX_PROF=1); nyx_launch_propagate picks the twin
// when the batch carries a profile buffer. Same arithmetic, same bits.
#ifndef NYX_PROF
#define NYX_PROF 0
#endif
#define NYX_HOST_TU ((NYX_EMIT & NYX_EMIT_PLAIN16) && !NYX_PROF)  /* the one object that carries the host side: launch, LDS sizing, frame shift */
// The two-part hand-off is a property of the TRANSLATION UNIT (NYX_COOP_TWO_PARTS, set by propagate_p2.hip), not a run-time branch:
// the role code of the integrator is register-allocated around the mailbox calls, and the mere presence of the two-part calls in
// the default kernel cost 3-5 % of the north-star run (8 h of propagation: 246.7 against 235.7 ms), whichever way they were folded.
#ifdef NYX_COOP_TWO_PARTS
#define COOP_PARTS_HERE 2
#else
#define COOP_PARTS_HERE 1
#endif

#define CAS __attribute__((address_space(4)))
typedef const CAS DevCfg *CfgPtr;
typedef const CAS HarmEntry *HarmPtr;
typedef const CAS ColHdr *ColPtr;

#define DEVFN static __device__ __forceinline__
typedef const __attribute__((address_space(3))) double *LdsCPtr;
// The control words of a workgroup (LdsMap.ctl) through an LDS-qualified pointer: a volatile access through a GENERIC pointer
// is left alone by the address-space inference and becomes a flat_load ... sc0 sc1 behind s_waitcnt vmcnt(0) - which also waits
// for every scratch reload in flight - where ds_read_b32 is meant.
typedef volatile __attribute__((address_space(3))) int *LdsFlagPtr;
#define LCTL ((LdsFlagPtr)L.ctl)

DEVFN double norm3(double x, double y, double z) { return sqrt(x * x + y * y + z * z); }
DEVFN double cube(double x) { return x * (x * x); }  // f64::powi(3)
DEVFN double clamp02(double x) { return x < 0.0 ? 0.0 : (x > 2.0 ? 2.0 : x); }

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
            double pij[3];
            ed_body(cfg, ed, lane, s, pij);
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
    {
        // Full sunlight and full umbra decided on COSINES, for the whole wave at once.  The exact path below compares angles -
        // d_p - ls_p > fo_p  (no occultation: 0.0 exactly)  and  fo_p > d_p + ls_p  (total: 100.0 exactly) - which costs two asin and one
        // acos per shadow body per stage, almost always to return one of those two constants.  With all three angles in [0, pi] and
        // the apparent radii below pi / 2 the same inequalities read  cos d_p < cos(ls_p + fo_p)  and  cos d_p > cos(fo_p - ls_p);
        // they are taken here only with a margin of 1e-9 in the cosine (>= 1e-9 rad in the angles, seven orders above the rounding of
        // either formulation), and only when EVERY lane of the wave is decided - then the exact path would return the same constant,
        // bit for bit; in the penumbra band, or when any lane is near a boundary, the exact path runs as before.
        const double dotq = r_ls[0] * r_eb[0] + r_ls[1] * r_eb[1] + r_ls[2] * r_eb[2];
        const double sl = r_back / n_ls, sf = r_front / n_eb;  // sines of the apparent radii
        const double cd = -dotq / (n_eb * n_ls);               // the argument of the exact path's acos, same expression
        const double cl = sqrt(1.0 - sl * sl), cf = sqrt(1.0 - sf * sf);
        const bool angles = r_back < n_ls && r_front < n_eb && cd >= -1.0 && cd <= 1.0;
        const bool lit = angles && cd < (cl * cf - sl * sf) - 1e-9;
        const bool dark = angles && sf > sl && cd > (cf * cl + sf * sl) + 1e-9;
        if (__all(lit || dark)) return lit ? 0.0 : 100.0;
    }
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
DEVFN double srp_force(CfgPtr cfg, const double *ed, int lane, const double *r, double cr, double area, double *force) {
    const int ss = cfg->sun_slot;
    double ps[3];
    ed_body(cfg, ed, lane, ss, ps);
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
                ed_body(cfg, ed, lane, sb, pb);
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
    return k;  // illumination factor |occultation - 1|, frozen in the partials (solarpressure.rs:194-203)
}

// f64::powi as LLVM expands it (binary method, LSB first)
DEVFN double powi_dev(double x, int n) {
    double res = 1.0, sq = x;
    bool have = false;
    while (n) {
        if (n & 1) { res = have ? res * sq : sq; have = true; }
        sq = sq * sq;
        n >>= 1;
    }
    return res;
}

// Drag::eom (reference dynamics/drag.rs:181-284) with its unit / frame quirks, as restated in the oracle (drag_eom):
// velocity in the drag frame = R v - w x (R r) with w = W_dot z_body; Exponential mixes metres and km; the relative
// velocity is (inertial velocity) - (drag-frame velocity components).  `m` = DCM inertial -> drag frame of this stage.
// dW/dt of an orientation (rad/s) without its DCM: the polynomial rate, plus the series / Chebyshev terms when there are any
DEVFN double rotation_w_rate(CfgPtr cfg, const CAS DevRot &rot, const double *records, double et_s) {
    const double DEG = 3.14159265358979323846 / 180.0;
    if (rot.kind == NYX_HIP_ROT_IAU && rot.n_np == 0) return (rot.w[1] + 2.0 * rot.w[2] * (et_s / 86400.0)) * DEG / 86400.0;
    double m[9], wr = 0.0;
    (void)rotation_dcm(cfg, rot, records, et_s, m, &wr);
    return wr;
}

DEVFN void drag_force(CfgPtr cfg, const double *records, const double *ed, int lane, double et_s, const double *r, const double *v, double cd, double area,
                      double *force) {
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) m[q] = ed[q * DEV_LANES + lane];
    const double DEG = 3.14159265358979323846 / 180.0;
    const double d = et_s / 86400.0;
    const double wdot = rotation_w_rate(cfg, cfg->d_rot, records, et_s);
    double rb[3], vb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rb[i] = m[3 * i + 0] * r[0] + m[3 * i + 1] * r[1] + m[3 * i + 2] * r[2];
        vb[i] = m[3 * i + 0] * v[0] + m[3 * i + 1] * v[1] + m[3 * i + 2] * v[2];
    }
    vb[0] = vb[0] + wdot * rb[1];
    vb[1] = vb[1] - wdot * rb[0];
    const double rmag = norm3(rb[0], rb[1], rb[2]);
    double rho;
    if (cfg->drag_density == NYX_HIP_RHO_CONSTANT) {
        rho = cfg->drag_rho0;
        const double vn = norm3(vb[0], vb[1], vb[2]);
        const double s = -0.5 * 1e3 * rho * cd * area * vn;
        force[0] = s * vb[0]; force[1] = s * vb[1]; force[2] = s * vb[2];
        return;
    } else if (cfg->drag_density == NYX_HIP_RHO_EXPONENTIAL) {
        rho = cfg->drag_rho0 * exp(-(rmag - (cfg->drag_r0 + cfg->drag_re)) / cfg->drag_ref_alt_m);
    } else {
        const double alt = rmag - cfg->drag_re;
        if (alt > cfg->drag_max_alt_m / 1000.0) {
            rho = pow(10.0, (-7e-5) * alt - 14.464);
        } else {
            const double sc = (alt - 526.8000) / 292.8563;
            const double lg = 0.34047 * powi_dev(sc, 6) - 0.5889 * powi_dev(sc, 5) - 0.5269 * powi_dev(sc, 4) + 1.0036 * powi_dev(sc, 3) +
                              0.60713 * powi_dev(sc, 2) - 2.3024 * sc - 12.575;
            rho = pow(10.0, lg);
        }
    }
    const double vel[3] = {v[0] - vb[0], v[1] - vb[1], v[2] - vb[2]};
    const double vn = norm3(vel[0], vel[1], vel[2]);
    const double s = -0.5 * 1e3 * rho * cd * area * vn;
    force[0] = s * vel[0]; force[1] = s * vel[1]; force[2] = s * vel[2];
}

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

// ---------------------------------------------------------------------------------------------
// Cooperative mode: idle CUs lend a hand.
//
// A workgroup holds 64 trajectories and fills one CU; an ensemble of 10 000 therefore occupies 157 of the 256 CUs.
// When workgroups are fewer than CUs, the launch adds HELPER workgroups on the idle CUs.  For every force evaluation
// the trajectory-owning workgroup posts the five per-lane inputs of the column recursion (2.5 KB) in a mailbox in
// global memory, keeps the columns of DEV_SCHED_PRIMARY for itself, and its helper evaluates the columns of
// DEV_SCHED_HELPER for the same 64 lanes and answers with four partial sums per lane (2 KB).  The exchange overlaps
// the owner's own window; in the pipelined stage loop the job of stage i+1 is posted inside the window of stage i
// (mailbox halves by the parity of the sequence number: an owner has up to two jobs outstanding, claimed in order).
// Deadlock-free without any residency assumption: the owner waits a bounded time for an answer, and if none comes it
// evaluates the helper's columns itself (walking DEV_SCHED_HELPER) and goes back to DEV_SCHED_SOLO for the rest of the
// launch; helpers leave when every workgroup they serve has finished.  The owner adds the helper's partial after its
// own sixteen, in a fixed order: results are deterministic for a given split.
// ---------------------------------------------------------------------------------------------
// The mailboxes live in UNCACHED device memory and are only touched with device-scope relaxed atomics (loads and stores
// that go past the L1 / the XCD's L2), ordered by workgroup-scope fences, i.e. s_waitcnt on the wave's own accesses: no
// cache write-back or invalidate anywhere (a device-scope fence per evaluation also throws the harmonics table out of
// L2 and doubled the run time).
// (through GLOBAL-qualified pointers: a generic pointer makes these flat_load / flat_store, which count on lgkmcnt as well as on vmcnt -
//  every LDS wait behind a post then also waited for the stores' round trip to uncached memory)
#define GAS __attribute__((address_space(1)))
DEVFN uint32_t coop_load(const uint32_t *p) { return __hip_atomic_load((const GAS uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN void coop_store(uint32_t *p, uint32_t v) { __hip_atomic_store((GAS uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN void coop_release() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }  // (inline asm: never elided by the compiler)
DEVFN void coop_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#define COOP_TIMEOUT_TICKS 200000LL  /* 2 ms of the 100 MHz realtime counter */
#define COOP_SET 16                  /* owners per set */

DEVFN uint64_t coop_loadu(const uint64_t *p) { return __hip_atomic_load((const GAS uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN void coop_storeu(uint64_t *p, uint64_t v) { __hip_atomic_store((GAS uint64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a double as two tagged granules (CoopBox): g[0] = {low half | seq << 32}, g[DEV_LANES] = {high half | seq << 32}
DEVFN void coop_put(uint64_t *g, double v, uint32_t seq) {
    const uint64_t b = (uint64_t)__double_as_longlong(v), t = (uint64_t)seq << 32;
    coop_storeu(g, (b & 0xffffffffull) | t);
    coop_storeu(g + DEV_LANES, (b >> 32) | t);
}
DEVFN bool coop_get(const uint64_t *g, uint32_t seq, double &v) {
    const uint64_t lo = coop_loadu(g), hi = coop_loadu(g + DEV_LANES);
    v = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
    return (uint32_t)(lo >> 32) == seq && (uint32_t)(hi >> 32) == seq;
}

// Posting happens from the LDS copy of the inputs, when the integrator wave has nothing else to do (start of the
// window in the plain loop, right after the next stage's inputs are formed in the pipelined one).  The inputs are tagged
// granules: the sequence number is written right behind them, with no wait in between - a helper that sees it before the
// data simply polls the granules until their tags agree.
// `parts` sub-jobs per evaluation (1, or 2: the helpers' columns in two halves, claimed by two helper workgroups): the words the
// helpers scan count SUB-JOBS - posted = parts * seq; sub-job c (1, 2, ...) is part (c - 1) % parts of evaluation (c + parts - 1) / parts.
// The single-part functions are kept exactly as small as they were before the two-part hand-off existed, and the two-part ones are
// their own functions behind a uniform branch at the call site: measured on the north-star run (8 h of propagation), folding both into
// one function with a run-time part count cost 2.8 % - the integrator's role code is register-allocated around these calls.
// (round 5: the five rows are read from LDS through an LDS-qualified pointer and all at once, THEN stored.  Through the generic pointer
//  of rounds 1-4 every row was a flat_load behind `s_waitcnt vmcnt(0) lgkmcnt(0)`, i.e. behind the previous row's stores to uncached
//  memory: five serial round trips, 4.7 k cycles of the integrator's window per evaluation.)
#ifndef COOP_INLINE
#define COOP_INLINE 0
#endif
#if COOP_INLINE
#define COOP_FN static __device__ __forceinline__
#else
#define COOP_FN static __device__ __attribute__((noinline))
#endif
// (`mult`: what the scan words count - sub-jobs: 1 per evaluation, or 2 with the two-part hand-off)
DEVFN void coop_post_inl(CoopBox *box, uint32_t *posted, int lane, uint32_t seq, LdsCPtr inb, uint32_t mult) {
    double v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = inb[q * DEV_LANES + lane];
#pragma unroll
    for (int q = 0; q < 5; ++q) coop_put(&box->in[seq & 1u][q][0][lane], v[q], seq);
    if (lane == 0) coop_store(posted, mult * seq);
}
COOP_FN void coop_post(CoopBox *box, uint32_t *posted, int lane, uint32_t seq, LdsCPtr inb) {
    double v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = inb[q * DEV_LANES + lane];
#pragma unroll
    for (int q = 0; q < 5; ++q) coop_put(&box->in[seq & 1u][q][0][lane], v[q], seq);
    if (lane == 0) coop_store(posted, seq);
}
static __device__ __attribute__((noinline)) void coop_post2(CoopBox *box, uint32_t *posted, int lane, uint32_t seq, LdsCPtr inb) {
    double v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = inb[q * DEV_LANES + lane];
#pragma unroll
    for (int q = 0; q < 5; ++q) coop_put(&box->in[seq & 1u][q][0][lane], v[q], seq);
    if (lane == 0) coop_store(posted, 2u * seq);  // (the scan words count SUB-JOBS)
}

struct CoopAnswer {
    double x, y, z, w;
    int ok;
};
// The answer needs no flag: every lane polls the LAST granule the helper writes for it, and when all of them carry this
// evaluation's tag the other seven are read and checked the same way (they were stored earlier, but nothing orders them).
DEVFN CoopAnswer coop_wait_inl(CoopBox *box, int lane, uint32_t seq) {
    CoopAnswer a = {0.0, 0.0, 0.0, 0.0, 0};
    const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
    const unsigned par = seq & 1u;
    // Round 6: the answer is read OPTIMISTICALLY first - all eight granules of every lane in one batch of loads, one round trip.  With the
    // late collection (phase C, behind the stage barrier) the answer is almost always in the mailbox by the time the integrator asks
    // (in-kernel accounting, round 5: "wait for the answer" 3.6 k cycles = exactly the three SERIAL uncached loads this function used to
    // make - poll lane 0's last granule, every lane's last granule, then the eight - with nothing to wait for), and this wave's
    // chain answer -> next post is what bounds a cooperative owner's period.  Only when the optimistic read misses does it fall back to
    // the light poll (ONE granule, one request: the traffic of 157 polling owners is not free) and then reads again.
    {
        const bool ok = coop_get(&box->out[par][0][0][lane], seq, a.x) & coop_get(&box->out[par][1][0][lane], seq, a.y) &
                        coop_get(&box->out[par][2][0][lane], seq, a.z) & coop_get(&box->out[par][3][0][lane], seq, a.w);
        if (__all(ok)) { a.ok = 1; return a; }
    }
    while ((uint32_t)(coop_loadu(&box->out[par][3][1][0]) >> 32) != seq) {
        if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > COOP_TIMEOUT_TICKS) { a.x = a.y = a.z = a.w = 0.0; return a; }
        __builtin_amdgcn_s_sleep(1);
    }
    for (;;) {
        const bool ok = coop_get(&box->out[par][0][0][lane], seq, a.x) & coop_get(&box->out[par][1][0][lane], seq, a.y) &
                        coop_get(&box->out[par][2][0][lane], seq, a.z) & coop_get(&box->out[par][3][0][lane], seq, a.w);
        if (__all(ok)) break;
        if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > COOP_TIMEOUT_TICKS) { a.x = a.y = a.z = a.w = 0.0; return a; }
        __builtin_amdgcn_s_sleep(1);
    }
    a.ok = 1;
    return a;
}
COOP_FN CoopAnswer coop_wait(CoopBox *box, int lane, uint32_t seq) { return coop_wait_inl(box, lane, seq); }
// two parts: part 0 from the mailbox, part 1 from the array of second answers, added in that order whichever helper answered first
DEVFN CoopAnswer coop_wait2_inl(CoopBox *box, CoopOut *out2, int lane, uint32_t seq) {
    CoopAnswer a = {0.0, 0.0, 0.0, 0.0, 0};
    const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
    const unsigned par = seq & 1u;
    for (int part = 0; part < 2; ++part) {
        uint64_t *o = part ? &out2->out[par][0][0][0] : &box->out[par][0][0][0];  // [4][2][64] granules of this part
        double x, y, z, w;
        while ((uint32_t)(coop_loadu(o + (3 * 2 + 1) * DEV_LANES) >> 32) != seq) {  // (one granule, one request: see coop_wait)
            if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > COOP_TIMEOUT_TICKS) return a;
            __builtin_amdgcn_s_sleep(1);
        }
        for (;;) {
            const bool there = (uint32_t)(coop_loadu(o + (3 * 2 + 1) * DEV_LANES + lane) >> 32) == seq;
            if (__all(there)) {
                const bool ok = coop_get(o + 0 * 2 * DEV_LANES + lane, seq, x) & coop_get(o + 1 * 2 * DEV_LANES + lane, seq, y) &
                                coop_get(o + 2 * 2 * DEV_LANES + lane, seq, z) & coop_get(o + 3 * 2 * DEV_LANES + lane, seq, w);
                if (__all(ok)) break;
            }
            if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > COOP_TIMEOUT_TICKS) { a.x = a.y = a.z = a.w = 0.0; return a; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (part == 0) { a.x = x; a.y = y; a.z = z; a.w = w; }
        else { a.x += x; a.y += y; a.z += z; a.w += w; }
    }
    a.ok = 1;
    return a;
}
static __device__ __attribute__((noinline)) CoopAnswer coop_wait2(CoopBox *box, CoopOut *out2, int lane, uint32_t seq) { return coop_wait2_inl(box, out2, lane, seq); }

// What the owner does when no helper answers: the helper's sixteen wave slots one after the other, summed in the
// helper's fold order, i.e. bit for bit the answer it did not get.  Out of line: a rare path must not cost the
// integrator role registers.
static __device__ __attribute__((noinline)) Partial4 coop_fallback(uint64_t cfg_u, uint64_t htab_u, uint64_t cols_u, const double *inb, int lane_p) {
#ifdef NYX_COOP_FAN
    const int lane = lane_p & 0xff, parts = (lane_p >> 8) & 0xf;   // (bits 8-11 of the lane argument: the parts of the fan-out)
#else
    const int lane = lane_p & 0xff, parts = (lane_p & 0x100) ? 2 : 1;  // (bit 8 of the lane argument: two parts)
#endif
    const double v0 = inb[0 * DEV_LANES + lane], v1 = inb[1 * DEV_LANES + lane], v2 = inb[2 * DEV_LANES + lane],
                 v3 = inb[3 * DEV_LANES + lane], v4 = inb[4 * DEV_LANES + lane];
    Partial4 tot = {0.0, 0.0, 0.0, 0.0};
    for (int part = 0; part < parts; ++part) {  // (every part summed on its own, then added in part order: what coop_wait does with the answers)
#ifdef NYX_COOP_FAN
        const int sched = DEV_SCHED_FAN0 + part;
#else
        const int sched = part ? DEV_SCHED_HELPER2 : DEV_SCHED_HELPER;
#endif
        Partial4 o = {0.0, 0.0, 0.0, 0.0};
        for (int hw = 0; hw < DEV_MAX_WAVES; ++hw) {
            const Partial4 p = (((CfgPtr)uniform_u64(cfg_u))->harm_feed & 2) ? harmonics_stream(cfg_u, cols_u, hw, sched, v0, v1, v2, v3, v4)
                                                                             : harmonics_partial(cfg_u, htab_u, cols_u, hw, sched, v0, v1, v2, v3, v4);
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        if (part == 0) tot = o;
        else { tot.x += o.x; tot.y += o.y; tot.z += o.z; tot.w += o.w; }
    }
    return tot;
}

// Helper workgroup.  Jobs are CLAIMED, not assigned: the owners are dealt into sets of at most 16, a helper watches
// one set (lane l < 16 of its wave 0 <-> one owner: two 64-byte loads scan the set), and whichever helper of the set is
// free takes the next posted job with a compare-and-swap on claimed[owner].  The load evens out by itself whatever the
// ratio of helpers to owners.
//
// Inside the workgroup the job is a two-slot software pipeline with no workgroup barrier: wave 0 is the PRODUCER (it
// claims job j+1 and fetches its five input rows from the mailbox into LDS while the others work on job j), waves
// 1..14 are the column waves (one column each), and wave 15 ANSWERS: it waits for the fourteen partial sums, folds
// them in the fixed wave order and writes the answer.  So the two memory round trips of a job (inputs in, answer out,
// ~2 us each on uncached memory) overlap the arithmetic of its neighbours: a helper's job period is its longest
// column, not column + latencies.  LDS words: ready[s] / answered[s] = 1 + number of the job last published /
// answered in slot s, cnt[s] = column waves that have delivered.
#ifndef NYX_SEG_PROF
#define NYX_SEG_PROF 0  /* 1 adds the integrator's per-piece timers (rows 34-35); off in the product build, they cost registers */
#endif
#ifndef STEP_ONE_POW
/* step control: one pow in front of the accept / reject branches (round 6).  The sixteen-wave plain kernels only - measured same box,
 * three interleaved pairs each: 24 h of configs[1] 595.1 -> 592.1 ms (the decision 13.0 k -> 10.7 k cycles per attempt); the eight-wave
 * kernel of config 3, whose integrator shares its SIMD with one almanac wave, 46.8 ms with two pows against 47.2 with one */
#define STEP_ONE_POW ((NYX_EMIT & (NYX_EMIT_PLAIN16 | NYX_EMIT_PLAIN16_P2 | NYX_EMIT_PLAIN16_FAN)) ? 1 : 0)
#endif
#ifndef FAN_SUMS
#ifdef NYX_FAN_SUMS
#define FAN_SUMS 1
#else
#define FAN_SUMS 0  /* fan-out mode: the integrator's two stage sums formed by a column wave of their own (fan_sums, DevCfg.sums_wave1).  Built, bit-identical, and measured in round 6 (1 250 x 24 h, same box): 412.7 ms with it against 419.4 / 416.6 without - the integrator's window shrinks from 12.7 k to 7.7 k cycles per evaluation, but the almanac wave (18.6 k busy) then bounds the period; with the almanac duty fanned out as well (role_fanout + chained attempts) the integrator's phase C and the helpers' turnaround do (20.1 k).  Off: its six LDS rows (3 KB) pushed config 3's padded ephemeris records out of LDS (43.9 -> 47.1 ms) */
#endif
#endif
#ifndef STEP_OOL
#ifdef NYX_COOP_FAN
#define STEP_OOL 1          /* step control out of line (integ_step, round 6): the fan-out kernel, whose period IS the integrator's chain (1 250 x 24 h: 391 -> 382.5 ms) */
#else
#define STEP_OOL 0          /* the other INTEG_OOL kernels keep it inline: measured same box, 24 h of configs[1]: 601.9 ms out of line against 597.9 inline (three interleaved pairs; step control 19 k -> 11.9 k cycles per attempt either way, but the period there is the column waves') */
#endif
#endif
#ifndef STEP_SUMS_UNROLL
#define STEP_SUMS_UNROLL 0  /* step control: unroll factor of the loop over the stages of its two sums (0: as the compiler leaves it) */
#endif
#ifndef COOP_AFFINITY
#define COOP_AFFINITY 1  /* helpers take a job of their own first (see helper_body) */
#endif
#ifndef HELPER_SLOTS
#define HELPER_SLOTS 2  /* jobs in flight inside a helper (see helper_body: three and four were measured, slower) */
#endif
#define HELPER_LDS_BYTES ((HELPER_SLOTS * DEV_MAX_WAVES * 4 * DEV_LANES + HELPER_SLOTS * 5 * DEV_LANES) * 8 + 64 * 4)
DEVFN void helper_body(const DevBatch &bt, CfgPtr cfg, HarmPtr htab, ColPtr cols, char *smem, int lane, int wave) {
    // HELPER_SLOTS jobs in flight.  Round 5 measured three and four (in-kernel accounting of a helper, tools/sweep.py "profile"): with
    // two slots the producer waits ~9 k cycles per job for a slot and only then scans, claims and fetches (~10 k cycles of uncached
    // round trips); more slots do move the claim under the arithmetic - and lose, 84.9 -> 92.3 -> 103.3 ms per 3 h of configs[1]:
    // a job claimed early queues INSIDE this helper behind two or three others while another helper would have been free sooner
    // (lost claims per job 2.4 -> 2.6 -> 4.0): the rate of jobs is the owners', what counts is each job's turnaround.
    constexpr int NS = HELPER_SLOTS;
    double *part = (double *)smem;                                  // [NS][16][4][64]
    double *inl = part + NS * DEV_MAX_WAVES * 4 * DEV_LANES;        // [NS][5][64]
    int *ctl = (int *)(inl + NS * 5 * DEV_LANES);
    const LdsFlagPtr ready = (LdsFlagPtr)ctl, answered = (LdsFlagPtr)ctl + 4, jown = (LdsFlagPtr)ctl + 8, jseq = (LdsFlagPtr)ctl + 12, jpart = (LdsFlagPtr)ctl + 20;
    int *cnt = ctl + 16;
    constexpr int parts = COOP_PARTS_HERE;
    const int answer_wave = (int)(blockDim.x / DEV_LANES) - 1;
    const int n_col_waves = answer_wave - 1;
    if (wave == 0 || wave == answer_wave) {
        if (wave == 0 && lane < 32) ctl[lane] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) part[((sl * DEV_MAX_WAVES + wave) * 4 + q) * DEV_LANES + lane] = 0.0;
        }
    }
    __syncthreads();
#ifdef NYX_COOP_FAN
    // FAN-OUT mode (small shards: the idle CUs outnumber the owners at least two to one).  Helper h is DEDICATED to owner h % owners and
    // evaluates part h / owners of that owner's hand-off - no scan words, no claim, no lost race: its producer polls the tag of the
    // owner's input rows (the poll is half of the fetch) and the columns of an evaluation are dealt over coop_parts helper workgroups,
    // so a job is a fraction of a column set (two waves per SIMD or fewer finish in ~10 k cycles where fourteen need ~17 k) and the
    // owner keeps next to nothing.  The owner's side is the single-part protocol unchanged - one post, one answer in its mailbox -:
    // the helpers of the parts 1.. write their sums to coop_out2[owner * parts + part], the part-0 helper (the LEAD) waits for them,
    // adds them to its own in part order and answers.  That hop is on no critical path: the owner asks for the answer ~1.5 periods
    // after the post.  Nothing assumes residency: a part that never answers makes the lead give up, the owner time out after 2 ms and
    // walk every part's columns itself (coop_fallback: the same sums in the same order).
    const int fan_h = (int)blockIdx.x - bt.coop_base;
    const int fan_own_n = (int)((bt.n + DEV_LANES - 1) / DEV_LANES);
    const int fan_owner = fan_h % fan_own_n, fan_part = fan_h / fan_own_n;
    const int fan_parts = bt.coop_parts;
    if (fan_part >= fan_parts) return;
    if (wave == 0) {
        const int fan_widx = bt.coop_sets > 0 ? (fan_owner % bt.coop_sets) * COOP_SET + fan_owner / bt.coop_sets : 0;  // (the owner's coop_widx)
        const CoopBox *b = bt.coop_box + fan_owner;
        for (int j = 0;; ++j) {
            const int s = j % NS;
            const uint32_t seq = (uint32_t)j + 1u;   // the owner's evaluations, in order: every one of them is this helper's job
            const unsigned par = seq & 1u;
            int owner = fan_owner;
            const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
            bool slot_free = j < NS;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
            for (int it = 0;; ++it) {
                if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > 5000 * COOP_TIMEOUT_TICKS) { owner = -1; break; }  // 10 s: never spin forever
                if (!slot_free) {  // the job that used this slot NS rounds ago has been answered
                    slot_free = answered[s] == j - (NS - 1);
                    if (!slot_free) { __builtin_amdgcn_s_sleep(4); continue; }
                }
                // one request: the last granule the owner writes for lane 0 (the owner stores its rows in order, nothing orders them in
                // memory: the fetch below checks every tag)
                if ((uint32_t)(coop_loadu(&b->in[par][4][1][0]) >> 32) == seq) {
                    const bool got = coop_get(&b->in[par][0][0][lane], seq, v0) & coop_get(&b->in[par][1][0][lane], seq, v1) &
                                     coop_get(&b->in[par][2][0][lane], seq, v2) & coop_get(&b->in[par][3][0][lane], seq, v3) &
                                     coop_get(&b->in[par][4][0][lane], seq, v4);
                    if (__all(got)) break;
                    continue;
                }
                if ((it & 7) == 7 && coop_load(bt.coop_finished + fan_widx) != 0u) { owner = -1; break; }  // the owner is done (or carries on alone)
                __builtin_amdgcn_s_sleep(2);
            }
            if (owner >= 0) {
                double *il = inl + s * 5 * DEV_LANES;
                il[0 * DEV_LANES + lane] = v0; il[1 * DEV_LANES + lane] = v1; il[2 * DEV_LANES + lane] = v2;
                il[3 * DEV_LANES + lane] = v3; il[4 * DEV_LANES + lane] = v4;
            }
            if (lane == 0) { jown[s] = owner; jseq[s] = (int)seq; jpart[s] = fan_part; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) ready[s] = j + 1;
            if (owner < 0) break;
        }
        return;
    }
#else
    if (wave == 0) {
        const int h = (int)blockIdx.x - bt.coop_base;
        const int64_t n_own = (bt.n + DEV_LANES - 1) / DEV_LANES;
        const int n_sets = bt.coop_sets;
        const int set = h % n_sets;
        const int64_t mine = (int64_t)set + (int64_t)lane * n_sets;  // the owner this lane watches (lanes 0..15)
        const bool has = lane < COOP_SET && mine < n_own;
        const int widx = set * COOP_SET + lane;                       // its scan words
        unsigned turn = (unsigned)h;
        const bool pprof = NYX_PROF && bt.prof != nullptr && (int)blockIdx.x == bt.coop_base;
        int64_t pp_slot = 0, pp_scan = 0, pp_jobs = 0, pp_lost = 0;
        const int64_t pp_start = pprof ? (int64_t)__builtin_readcyclecounter() : 0;
        for (int j = 0;; ++j) {
            const int s = j % NS;
            int owner = -1, sub = 0;
            uint32_t seq = 0;
            const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
            const int64_t pc0 = pprof ? (int64_t)__builtin_readcyclecounter() : 0;
            int64_t pc1 = pc0;
            bool slot_free = j < NS;
            for (;;) {
                if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > 5000 * COOP_TIMEOUT_TICKS) { owner = -1; break; }  // 10 s: never spin forever
                if (!slot_free) {  // the job that used this slot NS rounds ago has been answered
                    slot_free = answered[s] == j - (NS - 1);
                    if (!slot_free) { __builtin_amdgcn_s_sleep(4); continue; }
                    if (pprof) pc1 = (int64_t)__builtin_readcyclecounter();
                }
                // the two words are read by independent loads: a pair (old posted, new claimed) is possible and must not look
                // like a job, hence "posted is AHEAD of claimed", not "differs from"
                const uint32_t posted = has ? coop_load(bt.coop_posted + widx) : 0u;
                const uint32_t claimed = has ? coop_load(bt.coop_claimed + widx) : 0u;
                const uint64_t cand = __ballot(has && (int32_t)(posted - claimed) > 0);
                if (cand) {
                    // first candidate at or after a rotating start lane, so that the helpers of a set spread over the jobs
                    const unsigned rot = turn++ & 63u;
                    const uint64_t hi = cand >> rot;
                    int pick = hi ? (int)rot + __builtin_ctzll(hi) : __builtin_ctzll(cand);
#if COOP_AFFINITY
                    // ... but a job has a PREFERRED helper - (owner slot + job number) mod the set's helpers, so that an owner's consecutive
                    // jobs go round the set - and a helper takes one of its own first: two idle helpers of a set that see the same jobs no
                    // longer go for the same one.  Round 5, 3 h of configs[1], same box, alternating: 82.6 / 83.3 ms without, 79.3 / 79.2 with
                    // (lost claims per job 1.75 -> 1.3; the results are the same bits).  Measured and dropped: a static owner -> helper
                    // preference (80.7-81.2), waiting one more scan for a job of its own (82.8-83.3: lost claims 0.55, but the wait is on the
                    // job's path), every helper taking the waiting job NEAREST to its rank (83.1-84.3: it takes its neighbour's).
                    {
                        const int hs = (bt.coop_helpers - set + n_sets - 1) / n_sets;   // helpers watching this set
                        const int rank = h / n_sets;
                        const uint64_t pref = __ballot(has && (int32_t)(posted - claimed) > 0 && hs > 0 && (int)(((unsigned)lane + claimed) % (unsigned)hs) == rank);
                        if (pref) pick = __builtin_ctzll(pref);
                    }
#endif
                    // jobs are taken in order, one at a time: an owner may have two outstanding (the pipelined loop posts
                    // stage i+1 before it has read the answer of stage i).  The five input rows of the job are fetched in the
                    // shadow of the compare-and-swap (they were complete before `posted` moved): one memory round trip, not two.
                    const int owner_c = (int)__shfl((int)mine, pick);
                    const uint32_t sub_c = (uint32_t)__shfl((int)claimed, pick) + 1u;           // the sub-job being claimed (1, 2, ...)
                    const uint32_t seq_c = parts == 2 ? (sub_c + 1u) >> 1 : sub_c;             // its evaluation ...
                    const int part_c = parts == 2 ? (int)((sub_c - 1u) & 1u) : 0;              // ... and which part of the hand-off
                    int won = 0;
                    if (lane == pick) {
                        uint32_t expect = claimed;
                        won = __hip_atomic_compare_exchange_strong(bt.coop_claimed + widx, &expect, claimed + 1u, __ATOMIC_RELAXED,
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                    }
                    const CoopBox *b = bt.coop_box + owner_c;
                    const unsigned par = seq_c & 1u;
                    // (measured: fetching only after the claim has succeeded costs 7 % of the north-star run - the helper's job
                    //  latency is what bounds its share)
                    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
                    bool got = false;
                    // Fetch the inputs only AFTER the claim has succeeded.  (Rounds 1-3 fetched them in the shadow of the compare-and-swap -
                    // measured then as 7 % faster; with the tagged-granule transport the opposite holds: every lost race was 5 KB of
                    // uncached reads, and the north-star run is 5.5 % FASTER without them - 719.5 -> 679.8 ms, same box.  coop_mute bit 1
                    // = debug_flags 0x200000 restores the speculative fetch.)
                    const bool lazy = (bt.coop_mute & 2) == 0;
                    if (!lazy)
                        got = coop_get(&b->in[par][0][0][lane], seq_c, v0) & coop_get(&b->in[par][1][0][lane], seq_c, v1) &
                              coop_get(&b->in[par][2][0][lane], seq_c, v2) & coop_get(&b->in[par][3][0][lane], seq_c, v3) &
                              coop_get(&b->in[par][4][0][lane], seq_c, v4);
                    if (__shfl(won, pick)) {
                        // the job is ours; its inputs were stored before the sequence number, but nothing orders the two: poll until
                        // every granule carries the tag (normally the first look already does)
                        const int64_t tw = (int64_t)__builtin_amdgcn_s_memrealtime();
                        bool first = lazy;
                        while (!__all(got)) {
                            if ((int64_t)__builtin_amdgcn_s_memrealtime() - tw > 100 * COOP_TIMEOUT_TICKS) break;  // (0.2 s: the owner has long given up on us)
                            if (!first) __builtin_amdgcn_s_sleep(1);
                            first = false;
                            got = coop_get(&b->in[par][0][0][lane], seq_c, v0) & coop_get(&b->in[par][1][0][lane], seq_c, v1) &
                                  coop_get(&b->in[par][2][0][lane], seq_c, v2) & coop_get(&b->in[par][3][0][lane], seq_c, v3) &
                                  coop_get(&b->in[par][4][0][lane], seq_c, v4);
                        }
                        // the poll timed out: the inputs were never seen whole.  The job is NOT worked on - a tagged answer vouches for
                        // the data it was computed from, and this one would be computed from torn or zero inputs; the owner gave up
                        // waiting 2 ms in, walks these columns itself and never looks at the mailbox again
                        if (!__all(got)) continue;
                        owner = owner_c;
                        seq = seq_c;
                        sub = part_c;
                        double *il = inl + s * 5 * DEV_LANES;
                        il[0 * DEV_LANES + lane] = v0; il[1 * DEV_LANES + lane] = v1; il[2 * DEV_LANES + lane] = v2;
                        il[3 * DEV_LANES + lane] = v3; il[4 * DEV_LANES + lane] = v4;
                        break;
                    }
                    if (pprof) ++pp_lost;
                    continue;  // another helper was faster: look again
                }
                const uint32_t fin = has ? coop_load(bt.coop_finished + widx) : 1u;
                if (__all(fin != 0u)) { owner = -1; break; }
                __builtin_amdgcn_s_sleep(8);  // ~0.2 us between scans: the set's words are one memory line shared by ~10 helpers (scanning 2-5x less often: no change)
            }
            if (lane == 0) { jown[s] = owner; jseq[s] = (int)seq; jpart[s] = sub; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) ready[s] = j + 1;
            if (pprof) { const int64_t now = (int64_t)__builtin_readcyclecounter(); pp_slot += pc1 - pc0; pp_scan += now - pc1; ++pp_jobs; }
            if (owner < 0) break;
        }
        if (pprof && lane == 0) {  // [0] cycles waiting for a free slot (the column waves are behind), [1] cycles from a free slot to a won and fetched job, [2] jobs, [3] lost claims
            int64_t *row = bt.prof + 17 * 8;
            row[0] = pp_slot; row[1] = pp_scan; row[2] = pp_jobs; row[3] = pp_lost; row[5] = (int64_t)__builtin_readcyclecounter() - pp_start;
        }
        return;
    }
#endif  // NYX_COOP_FAN
    // optional accounting of the FIRST helper workgroup (NYX_HIP_PROFILE; rows 17.. of the profile, one per wave): [0] cycles in the
    // column walk, [1] cycles waiting for a job, [2] jobs, [3] cycles from a job's publication in LDS to this wave's delivery, [5] total
#ifdef HELPER_PRIO
    // issue priority against the arbiter's oldest-first rule: the four waves of a SIMD start a job together, and served oldest first the
    // oldest is done after half the job's time and runs ahead into the next job while the youngest - whose column the answer waits
    // for - gets what is left
    if (wave != answer_wave) {
        const int pr = HELPER_PRIO == 1 ? (wave >> 2) : (3 - (wave >> 2));
        if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
#endif
    const bool hprof = NYX_PROF && bt.prof != nullptr && (int)blockIdx.x == bt.coop_base;
    int64_t hp_busy = 0, hp_wait = 0, hp_jobs = 0;
    const int64_t hp_start = hprof ? (int64_t)__builtin_readcyclecounter() : 0;
    for (int j = 0;; ++j) {
        const int s = j % NS;
        {
            const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
            const int64_t c0 = hprof ? (int64_t)__builtin_readcyclecounter() : 0;
            while (ready[s] != j + 1) {
                if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > 6000 * COOP_TIMEOUT_TICKS) return;  // (the producer gives up after 10 s)
                __builtin_amdgcn_s_sleep(4);
            }
            if (hprof) hp_wait += (int64_t)__builtin_readcyclecounter() - c0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int owner = jown[s];
        const uint32_t seq = (uint32_t)jseq[s];
        const int sub = jpart[s];
        if (owner < 0) break;
        double *ps = part + s * DEV_MAX_WAVES * 4 * DEV_LANES;
        const int64_t hp_c0 = hprof ? (int64_t)__builtin_readcyclecounter() : 0;
        if (wave != answer_wave) {
            const double *il = inl + s * 5 * DEV_LANES;
            const double v0 = il[0 * DEV_LANES + lane], v1 = il[1 * DEV_LANES + lane], v2 = il[2 * DEV_LANES + lane],
                         v3 = il[3 * DEV_LANES + lane], v4 = il[4 * DEV_LANES + lane];
#ifdef NYX_COOP_FAN
            const int hsched = DEV_SCHED_FAN0 + sub;
#else
            const int hsched = sub ? DEV_SCHED_HELPER2 : DEV_SCHED_HELPER;
#endif
            const Partial4 pr = (cfg->harm_feed & 2) ? harmonics_stream((uint64_t)cfg, (uint64_t)cols, wave, hsched, v0, v1, v2, v3, v4)
                                               : harmonics_partial((uint64_t)cfg, (uint64_t)htab, (uint64_t)cols, wave, hsched, v0, v1, v2, v3, v4);
            double *pp = ps + wave * 4 * DEV_LANES;
            pp[0 * DEV_LANES + lane] = pr.x; pp[1 * DEV_LANES + lane] = pr.y; pp[2 * DEV_LANES + lane] = pr.z; pp[3 * DEV_LANES + lane] = pr.w;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) (void)__hip_atomic_fetch_add(cnt + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (hprof) { hp_busy += (int64_t)__builtin_readcyclecounter() - hp_c0; ++hp_jobs; }
            continue;
        }
        // ---- the answering wave: wait for the column waves, fold in the fixed wave order (the slots of the producer and of
        // this wave hold zeros), answer.  None of this is on a column wave's path.
        {
            const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(cnt + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != n_col_waves) {
                if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > 6000 * COOP_TIMEOUT_TICKS) return;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        CoopBox *b = bt.coop_box + owner;
        const unsigned par = seq & 1u;
        double o[4] = {0.0, 0.0, 0.0, 0.0};
        for (int w = 0; w < DEV_MAX_WAVES; ++w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] += ps[(w * 4 + q) * DEV_LANES + lane];
        }
#ifdef NYX_COOP_FAN
        bool fan_ok = true;
        if (sub == 0) {
            // the lead: the sums of the parts 1.., in part order (what coop_fallback adds up when the owner walks the parts itself)
            const int64_t tl = (int64_t)__builtin_amdgcn_s_memrealtime();
            for (int pq = 1; pq < fan_parts && fan_ok; ++pq) {
                const uint64_t *o2 = &bt.coop_out2[owner * fan_parts + pq].out[par][0][0][0];
                double x = 0.0, y = 0.0, z = 0.0, w = 0.0;
                for (;;) {
                    const bool got = coop_get(o2 + 0 * 2 * DEV_LANES + lane, seq, x) & coop_get(o2 + 1 * 2 * DEV_LANES + lane, seq, y) &
                                     coop_get(o2 + 2 * 2 * DEV_LANES + lane, seq, z) & coop_get(o2 + 3 * 2 * DEV_LANES + lane, seq, w);
                    if (__all(got)) break;
                    if ((int64_t)__builtin_amdgcn_s_memrealtime() - tl > COOP_TIMEOUT_TICKS) { fan_ok = false; break; }  // (the owner gives up at the same age)
                    __builtin_amdgcn_s_sleep(2);
                }
                o[0] += x; o[1] += y; o[2] += z; o[3] += w;
            }
        }
        if (fan_ok) {
            uint64_t *og = sub ? &bt.coop_out2[owner * fan_parts + sub].out[par][0][0][0] : &b->out[par][0][0][0];
#pragma unroll
            for (int q = 0; q < 4; ++q) coop_put(og + q * 2 * DEV_LANES + lane, o[q], seq);
        }
#else
        {
            uint64_t *og = (sub && bt.coop_out2) ? &bt.coop_out2[owner].out[par][0][0][0] : &b->out[par][0][0][0];
#pragma unroll
            for (int q = 0; q < 4; ++q) coop_put(og + q * 2 * DEV_LANES + lane, o[q], seq);  // tagged granules: no drain, no flag (the owner polls the last one)
        }
#endif
        if (lane == 0) __hip_atomic_store(cnt + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) answered[s] = j + 1;  // the slot may be refilled: its partial sums are in registers
        if (lane == 0 && bt.prof != nullptr) atomicAdd((unsigned long long *)bt.prof + 16 * 8 + 4, 1ull);
        if (hprof) { hp_busy += (int64_t)__builtin_readcyclecounter() - hp_c0; ++hp_jobs; }
    }
    if (hprof && lane == 0) {
        int64_t *row = bt.prof + (17 + wave) * 8;
        row[0] = hp_busy; row[1] = hp_wait; row[2] = hp_jobs; row[5] = (int64_t)__builtin_readcyclecounter() - hp_start;
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

// Fold of the 15 workers' partial accelerations (fixed wave order => deterministic).  Kept out of line on purpose:
// inside the integrator role (at its 128-VGPR cap) the scheduler serialised the 60 LDS reads at one LDS latency each
// (5 k cycles on the critical path of every force evaluation); on its own the function batches them.
#ifndef FOLD_INLINE
#define FOLD_INLINE 0
#endif
#if FOLD_INLINE
// (inlined variant: the sixty reads in four batches of fifteen - one component at a time - so that they need 30 registers, not 120)
static __device__ __forceinline__ Partial4 fold_partials(LdsCPtr part, int lane, double px, double py, double pz, double pw) {
    double o[4] = {px, py, pz, pw};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v[DEV_MAX_WAVES - 1];
#pragma unroll
        for (int w = 1; w < DEV_MAX_WAVES; ++w) v[w - 1] = part[(w * 4 + q) * DEV_LANES + lane];
#pragma unroll
        for (int w = 1; w < DEV_MAX_WAVES; ++w) o[q] += v[w - 1];
    }
    Partial4 r = {o[0], o[1], o[2], o[3]};
    return r;
}
#else
static __device__ __attribute__((noinline)) Partial4 fold_partials(LdsCPtr part, int lane, double px, double py, double pz, double pw) {
    double v[4][DEV_MAX_WAVES - 1];
#pragma unroll
    for (int w = 1; w < DEV_MAX_WAVES; ++w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q][w - 1] = part[(w * 4 + q) * DEV_LANES + lane];
    }
#pragma unroll
    for (int w = 1; w < DEV_MAX_WAVES; ++w) {
        px += v[0][w - 1]; py += v[1][w - 1]; pz += v[2][w - 1]; pw += v[3][w - 1];
    }
    Partial4 r = {px, py, pz, pw};
    return r;
}
#endif

// ---------------------------------------------------------------------------------------------
// STM variant: position partials of the perturbations (perturbation wave) and the per-step update
// ---------------------------------------------------------------------------------------------

// PointMasses::gradient (orbital.rs:249-308) and SolarPressure::gradient (solarpressure.rs:167-232, k frozen).
// out[27][64]: a_pm(3), G_pm(9 row-major), f_srp/m(3), G_srp/m(9), c = (F/Cr)/m (3, zero unless `estimate`).
DEVFN void pert_gradients(CfgPtr cfg, const double *ed, int lane, const double *r, double cr, double area, double mass,
                          bool has_pm, bool has_srp, bool has_tides, double *out) {
    double o[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) o[q] = 0.0;
    if (has_pm) {
        const int npm = cfg->n_pm;
#pragma unroll
        for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
            if (k < npm) {
                const int s = cfg->pm_slot[k];
                double pb3[3];
                ed_body(cfg, ed, lane, s, pb3);
                const D3 rij[3] = {d3c(pb3[0]), d3c(pb3[1]), d3c(pb3[2])};
                const D3 rij3 = d3cube(d3norm(rij[0], rij[1], rij[2]));
                const D3 rj[3] = {{r[0] - rij[0].v, 1.0, 0.0, 0.0}, {r[1] - rij[1].v, 0.0, 1.0, 0.0}, {r[2] - rij[2].v, 0.0, 0.0, 1.0}};
                const D3 rj3 = d3cube(d3norm(rj[0], rj[1], rj[2]));
                const D3 gm = d3c(-cfg->slot[s].mu);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const D3 t = (d3div(rj[i], rj3) + d3div(rij[i], rij3)) * gm;
                    o[i] += t.v;
                    o[3 + 3 * i + 0] += t.x; o[3 + 3 * i + 1] += t.y; o[3 + 3 * i + 2] += t.z;
                }
            }
        }
    }
    if (has_tides) {  // SolidTides::gradient: added to the orbital (point-mass) block
        const D3 rd[3] = {{r[0], 1.0, 0.0, 0.0}, {r[1], 0.0, 1.0, 0.0}, {r[2], 0.0, 0.0, 1.0}};
        D3 at[3];
        tides_accel<D3>(cfg, ed, lane, rd, at);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            o[i] += at[i].v;
            o[3 + 3 * i + 0] += at[i].x; o[3 + 3 * i + 1] += at[i].y; o[3 + 3 * i + 2] += at[i].z;
        }
    }
    if (has_srp) {
        const int ss = cfg->sun_slot;
        double ps[3];
    ed_body(cfg, ed, lane, ss, ps);
        const D3 rs[3] = {{r[0] - ps[0], 1.0, 0.0, 0.0}, {r[1] - ps[1], 0.0, 1.0, 0.0}, {r[2] - ps[2], 0.0, 0.0, 1.0}};
        const D3 n = d3norm(rs[0], rs[1], rs[2]);
        // illumination factor exactly as the real path computes it (frozen in the partials)
        double f3[3];
        const double kfro = srp_force(cfg, ed, lane, r, cr, area, f3);  // real path: force and the frozen illumination factor
        const D3 r_au = n * (1.0 / 149597870.700);
        const D3 inv = d3div(d3c(1.0), r_au);
        const D3 flux = (inv * inv) * (kfro * cfg->phi / cfg->c_m_s);
        const double scal = 1e-3 * cr * area;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const D3 f = (flux * scal) * d3div(rs[i], n);
            o[12 + i] = f3[i] / mass;  // real part from the real path (spacecraft.rs:349)
            o[15 + 3 * i + 0] = f.x / mass; o[15 + 3 * i + 1] = f.y / mass; o[15 + 3 * i + 2] = f.z / mass;
            if (cfg->srp_estimate) o[24 + i] = (f3[i] / cr) / mass;  // solarpressure.rs:225-229, spacecraft.rs:355-359
        }
    }
#pragma unroll
    for (int q = 0; q < 27; ++q) out[q * DEV_LANES + lane] = o[q];
}

// Phi_next = Phi + h * Phi * A_sum with A_sum = [[0, (sum b) I, 0], [Gs, 0, cs], [0, 0, 0]]  — the reference integrates
// Phi_dot = Phi_ctx * A with the STEP-START Phi (dynamics/spacecraft.rs:214), so the RK sum factorises exactly.
// phi: this trajectory's 81 entries, column-major (cosmic/spacecraft.rs:467-471).
DEVFN bool stm_update(double *phi, double h, const double *sacc, int lane, double sumb) {
    double gs[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) gs[q] = sacc[q * DEV_LANES + lane];
    bool nan = false;
    for (int r = 0; r < 9; ++r) {
        double row[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) row[c] = phi[r + 9 * c];
        double nw[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            nw[j] = row[j] + h * (row[3] * gs[0 * 3 + j] + row[4] * gs[1 * 3 + j] + row[5] * gs[2 * 3 + j]);
            nw[3 + j] = row[3 + j] + h * (sumb * row[j]);
        }
        nw[6] = row[6] + h * (row[3] * gs[9] + row[4] * gs[10] + row[5] * gs[11]);
        nw[7] = row[7];
        nw[8] = row[8];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            nan = nan || (nw[c] != nw[c]);
            phi[r + 9 * c] = nw[c];
        }
    }
    return nan;
}

// Quad layout of the two functions above.  out[15][64]: a_pm(3), column (ql - 1) of G_pm (3), f_srp/m(3), column of
// G_srp/m (3), c(3); every expression is pert_gradients' own for the value and for ONE partial slot.
DEVFN void pert_gradients_q(CfgPtr cfg, const double *ed, int lane, int ql, const double *r, double cr, double area, double mass,
                            bool has_pm, bool has_srp, bool has_tides, int pmask, double *out) {
    double o[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) o[q] = 0.0;
    if (has_pm) {
        const int npm = cfg->n_pm;
#pragma unroll
        for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
            if (k < npm) {
                const int s = cfg->pm_slot[k];
                double pb3[3];
                ed_body(cfg, ed, lane, s, pb3);
                const D1 rij[3] = {d1c(pb3[0]), d1c(pb3[1]), d1c(pb3[2])};
                const D1 rij3 = d1cube(d1norm(rij[0], rij[1], rij[2]));
                const D1 rj[3] = {d1seed(r[0] - rij[0].v, 0, ql), d1seed(r[1] - rij[1].v, 1, ql), d1seed(r[2] - rij[2].v, 2, ql)};
                const D1 rj3 = d1cube(d1norm(rj[0], rj[1], rj[2]));
                const D1 gm = d1c(-cfg->slot[s].mu);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const D1 t = (d1div(rj[i], rj3) + d1div(rij[i], rij3)) * gm;
                    o[i] += t.v;
                    o[3 + i] += t.d;
                }
            }
        }
    }
    if (has_tides) {
        const D1 rd[3] = {d1seed(r[0], 0, ql), d1seed(r[1], 1, ql), d1seed(r[2], 2, ql)};
        D1 at[3];
        tides_accel<D1>(cfg, ed, lane, rd, at);
#pragma unroll
        for (int i = 0; i < 3; ++i) { o[i] += at[i].v; o[3 + i] += at[i].d; }
    }
    if (has_srp) {
        const int ss = cfg->sun_slot;
        double ps[3];
    ed_body(cfg, ed, lane, ss, ps);
        const D1 rs[3] = {d1seed(r[0] - ps[0], 0, ql), d1seed(r[1] - ps[1], 1, ql), d1seed(r[2] - ps[2], 2, ql)};
        const D1 n = d1norm(rs[0], rs[1], rs[2]);
        double f3[3];
        const double kfro = srp_force(cfg, ed, lane, r, cr, area, f3);
        const D1 r_au = n * (1.0 / 149597870.700);
        const D1 inv = d1div(d1c(1.0), r_au);
        const D1 flux = (inv * inv) * (kfro * cfg->phi / cfg->c_m_s);
        const double scal = 1e-3 * cr * area;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const D1 f = (flux * scal) * d1div(rs[i], n);
            o[6 + i] = f3[i] / mass;
            o[9 + i] = f.d / mass;
            if (cfg->srp_estimate) o[12 + i] = (f3[i] / cr) / mass;
        }
    }
    // (role fan-out: rows 0..5 belong to the point-mass share, 6..14 to the SRP share)
#pragma unroll
    for (int q = 0; q < 15; ++q)
        if (pmask & (q < 6 ? DEV_PERT_PM : DEV_PERT_SRP)) out[q * DEV_LANES + lane] = o[q];
}

// stm_update for the quad layout: sacc rows 0..2 hold, per lane, column (ql - 1) of sum b_i G_i and rows 3..5 sum b_i c_i
// (the same in the four lanes); the nine rows of Phi are dealt over the quad's lanes.
DEVFN bool stm_update_q(double *phi, double h, const double *sacc, int lane, int ql, double sumb) {
    double gs[12];
    const int base = lane & ~3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) gs[3 * i + j] = sacc[i * DEV_LANES + base + 1 + j];
        gs[9 + i] = sacc[(3 + i) * DEV_LANES + lane];
    }
    bool nan = false;
    for (int r = ql; r < 9; r += 4) {
        double row[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) row[c] = phi[r + 9 * c];
        double nw[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            nw[j] = row[j] + h * (row[3] * gs[0 * 3 + j] + row[4] * gs[1 * 3 + j] + row[5] * gs[2 * 3 + j]);
            nw[3 + j] = row[3 + j] + h * (sumb * row[j]);
        }
        nw[6] = row[6] + h * (row[3] * gs[9] + row[4] * gs[10] + row[5] * gs[11]);
        nw[7] = row[7];
        nw[8] = row[8];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            nan = nan || (nw[c] != nw[c]);
            phi[r + 9 * c] = nw[c];
        }
    }
    return quad_or(nan ? 1 : 0) != 0;
}

// Phase C of the quad layout (assembly of f(x) and of this lane's column of A = df/dx from the partial sums, the
// perturbation rows and the position-only pieces formed in the window; accumulation of sum b_i A_i; k_i), OUT OF LINE: inside
// the integrator role (128 VGPRs = 64 doubles for everything it keeps live) it ran through scratch, 10 k cycles per
// evaluation; on its own it has the whole register file.  Everything goes through LDS: `qpre` rows 0..2 two-body
// acceleration, 3..5 this lane's column of its gradient, 6..13 the duals of s, t, u and (mu / r) / R_eq.
#define QPRE_ROWS 23  /* + rows 14..22: the DCM of the stage (the almanac wave recycles its LDS buffer in the pipelined loop) */
#define PC_HAS_PM 1
#define PC_HAS_GRAV 2
#define PC_HAS_SRP 4
// (LDS pointers are passed as such: through generic pointers every access pays an address-space test)
static __device__ __attribute__((noinline)) void phase_c_quad(LdsCPtr pertD, LdsCPtr partD, LdsCPtr qpre,
                                                            LdsCPtr ysl, LdsPtr sacc, LdsPtr kb, int kb_str, double b_i, int nw_v,
                                                            int flags_v, int lane, int ql, LdsFlagPtr gate, int gate_val_v,
                                                            int64_t *pslot = nullptr) {
    const int64_t pc0 = pslot ? (int64_t)__builtin_readcyclecounter() : 0;
    const int nw = __builtin_amdgcn_readfirstlane(nw_v);
    const int flags = __builtin_amdgcn_readfirstlane(flags_v);
    double acc[3], Gc[3], cv[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 3; ++q) { acc[q] = qpre[q * DEV_LANES + lane]; Gc[q] = qpre[(3 + q) * DEV_LANES + lane]; }
    if (flags & PC_HAS_PM) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { acc[q] += pertD[q * DEV_LANES + lane]; Gc[q] += pertD[(3 + q) * DEV_LANES + lane]; }
    }
    if (flags & PC_HAS_GRAV) {
        D1 pD[4] = {d1c(0.0), d1c(0.0), d1c(0.0), d1c(0.0)};
        for (int w0 = 0; w0 < nw; w0 += 4) {  // fixed wave order; four waves' worth of loads in flight
            double v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                LdsCPtr pp = partD + (w0 + k < nw ? w0 + k : 0) * QSLOT;
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[k][2 * q] = pp[4 * DEV_LANES + q * (DEV_LANES / 4) + (lane >> 2)]; v[k][2 * q + 1] = pp[q * DEV_LANES + lane]; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (w0 + k < nw) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { pD[q].v += v[k][2 * q]; pD[q].d += v[k][2 * q + 1]; }
                }
            }
        }
        {   // the sums are in registers: the column waves may write the next stage's into their slots
            const int gate_val = __builtin_amdgcn_readfirstlane(gate_val_v);
            if (gate_val > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) *gate = gate_val;
            }
        }
        if (pslot && lane == 0) pslot[5] += (int64_t)__builtin_readcyclecounter() - pc0;
        double m[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) m[q] = qpre[(14 + q) * DEV_LANES + lane];
        D1 aux[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { aux[q].v = qpre[(6 + 2 * q) * DEV_LANES + lane]; aux[q].d = qpre[(7 + 2 * q) * DEV_LANES + lane]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) pD[q] = pD[q] * aux[3];
        const D1 al[3] = {pD[0] + pD[3] * aux[0], pD[1] + pD[3] * aux[1], pD[2] + pD[3] * aux[2]};
        // a = R^T a_bf ; G_h = R^T G_bf R: the first product is linear in the partial slot (this lane's), the second
        // mixes the three slots: fetched from the quad's lanes 1..3; this lane forms column b = ql - 1
        double tmpc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc[a] += m[0 + a] * al[0].v + m[3 + a] * al[1].v + m[6 + a] * al[2].v;
            tmpc[a] = m[0 + a] * al[0].d + m[3 + a] * al[1].d + m[6 + a] * al[2].d;
        }
        const int b = ql > 0 ? ql - 1 : 0;
        const double mb0 = qpre[(14 + b) * DEV_LANES + lane], mb1 = qpre[(17 + b) * DEV_LANES + lane], mb2 = qpre[(20 + b) * DEV_LANES + lane];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double t0 = quad_bcast<1>(tmpc[a]), t1 = quad_bcast<2>(tmpc[a]), t2 = quad_bcast<3>(tmpc[a]);
            Gc[a] += t0 * mb0 + t1 * mb1 + t2 * mb2;
        }
    } else {
        const int gate_val = __builtin_amdgcn_readfirstlane(gate_val_v);
        if (gate_val > 0 && lane == 0) *gate = gate_val;
    }
    if (flags & PC_HAS_SRP) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            acc[q] += pertD[(6 + q) * DEV_LANES + lane]; cv[q] = pertD[(12 + q) * DEV_LANES + lane];
            Gc[q] += pertD[(9 + q) * DEV_LANES + lane];
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        sacc[q * DEV_LANES + lane] += b_i * Gc[q];
        sacc[(3 + q) * DEV_LANES + lane] += b_i * cv[q];
    }
    // k_i = [velocity of the stage state, f(x)]
#pragma unroll
    for (int e = 0; e < 3; ++e) { kb[e * kb_str] = ysl[(3 + e) * DEV_LANES + lane]; kb[(3 + e) * kb_str] = acc[e]; }
    if (pslot && lane == 0) pslot[6] += (int64_t)__builtin_readcyclecounter() - pc0;
}

// Quad layout: the position-only pieces of phase C - the two-body dual, the duals of s, t, u and (mu / r) / R_eq, the stage's DCM -
// formed inside the window and left in L.qpre for phase C.  By the integrator wave, or (DevCfg.qpre_off, round 5) by the almanac wave
// that holds DEV_ROLE_QPRE: the integrator's chain - phase C, phase A, window - is what bounds a quad workgroup's period, and this is
// 4-5 k cycles of its window that need nothing but the published position and the stage's epoch data.  Same operations on the same
// operands in the same lanes: same bits.
DEVFN void quad_pre(CfgPtr cfg, const double *edc, double y0, double y1, double y2, int ql, int lane, double *qpre, bool has_grav) {
    double q_acc[3], q_gc[3];
    D1 q_aux[4] = {d1c(0.0), d1c(0.0), d1c(0.0), d1c(0.0)};
    const D1 rad[3] = {d1seed(y0, 0, ql), d1seed(y1, 1, ql), d1seed(y2, 2, ql)};
    const D1 fac = d1div(d1c(-cfg->mu_central), d1cube(d1norm(rad[0], rad[1], rad[2])));
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const D1 a = rad[q] * fac;
        q_acc[q] = a.v; q_gc[q] = a.d;
    }
    if (has_grav) {
        double m[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) m[q] = edc[q * DEV_LANES + lane];
        double rq[3] = {y0, y1, y2};
        if (cfg->g_slot >= 0) {  // (uniform; plain stage loop then: edc is this stage's data)
            double pg[3];
            ed_body(cfg, edc, lane, cfg->g_slot, pg);
            rq[0] = y0 - pg[0]; rq[1] = y1 - pg[1]; rq[2] = y2 - pg[2];
        }
        const D1 x0 = d1seed(m[0] * rq[0] + m[1] * rq[1] + m[2] * rq[2], 0, ql);
        const D1 x1 = d1seed(m[3] * rq[0] + m[4] * rq[1] + m[5] * rq[2], 1, ql);
        const D1 x2 = d1seed(m[6] * rq[0] + m[7] * rq[1] + m[8] * rq[2], 2, ql);
        const D1 rD = d1norm(x0, x1, x2);
        q_aux[0] = d1div(x0, rD); q_aux[1] = d1div(x1, rD); q_aux[2] = d1div(x2, rD);
        q_aux[3] = d1div(d1div(d1c(cfg->g_mu), rD), d1c(cfg->g_re));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { qpre[q * DEV_LANES + lane] = q_acc[q]; qpre[(3 + q) * DEV_LANES + lane] = q_gc[q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { qpre[(6 + 2 * q) * DEV_LANES + lane] = q_aux[q].v; qpre[(7 + 2 * q) * DEV_LANES + lane] = q_aux[q].d; }
    if (has_grav) {  // the DCM of this stage, for phase C (its LDS buffer is recycled by the almanac wave in the pipelined loop)
#pragma unroll
        for (int q = 0; q < 9; ++q) qpre[(14 + q) * DEV_LANES + lane] = edc[q * DEV_LANES + lane];
    }
}

// ---------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------

#define NIN 5
// Pipelined plain loop: what the integrator forms in window i for stage i + 1 - its position, s, t, u, (mu / r) / R_eq, its DCM - is
// left in LDS and read back in phase A of stage i + 1 instead of being carried in registers across the window's and phase C's calls
// (coop_post, fold_partials, coop_wait: the ABI keeps 48 VGPRs across a call, the role had ~90 live and spilled the rest to scratch
// around each of them, every evaluation).  Same values, same bits.
#ifndef NX_IN_LDS
#define NX_IN_LDS 1
#endif
// timing-only debug switches (NYX_HIP_DEBUG env, never set in production): results are physically wrong
#define DBG_SKIP_SERIAL 0x100
#define DBG_SKIP_HARMONICS 0x200
// (quad layout: the four lanes of a quad share ONE k-buffer column, KB_STR = 16 trajectories per workgroup)
#define KB(stage, comp) kbuf[((stage)*6 + (comp)) * KB_STR + kb_li]

// Integrator state that is only touched between attempts lives in LDS (per lane, field-major), not in
// registers: the stage loop then keeps ~30 VGPRs of integrator state live instead of ~90 (no scratch spills).
#define CS_FIELDS 21
// The `enough_crossings` closure of until_nth_event (propagators/event.rs:108-146) for one accepted state: the event
// state (previous value, crossings) lives in global memory, touched once per accepted step and only when a stop
// condition is set; out of line so that the integrator's register allocation does not see it.
static __device__ __attribute__((noinline)) bool event_step(const nyx_hip_event_t *ev, double mu, int64_t epoch_ns, double *prev, int32_t *count,
                                                            double y0, double y1, double y2, double y3, double y4, double y5) {
    const double y[6] = {y0, y1, y2, y3, y4, y5};
    const double y_next = ev_eval(*ev, mu, epoch_ns, y);
    int n = *count;
    if (ev_crossing(ev->scalar, *prev, y_next)) n += 1;
    *prev = y_next;
    *count = n;
    return n >= ev->trigger;
}

struct ColdState {
    int64_t epoch, stop, step_size, prev_step, det_step, n_acc, n_rej, n_evals;
    double y[9];
    double h, det_error;
    int det_attempts, attempts, status;
    bool done, fresh, is_final, fixed, prev_kind, backprop, massless;
};
#define CS_I64(f) __double_as_longlong(cs[(f)*DEV_LANES + lane])
template <typename P>
DEVFN void cold_load(P cs, int lane, ColdState &c) {
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
template <typename P>
DEVFN void cold_store(P cs, int lane, const ColdState &c) {
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
    double *pert;   // [9][64]        point-mass accel (3), SRP force / mass (3), drag force / mass (3)
    double *step;   // [2][64]        epoch (as i64 bits) and h of the current attempt
    double *cs;     // [CS_FIELDS][64] integrator cold state
    double *part;   // [P][4][64]     harmonics partials (wave 0's slot unused)
    int *edst;      // [DEV_MAX_ALM][2][64] almanac status per almanac wave and buffer
    int *pertst;    // [2][64]        status of the perturbation wave's own epoch-dependent work (the second field's orientation), by stage parity
    int *ctl;       // [16]
    double *rec;    // [rec_doubles]
    // pipelined stage loop (non-STM): buffers of odd stages
    double *ys2, *inb2, *pert2;
    double *ixs;    // [4][64]  s, t, u, (mu / r) / R_eq of the ODD stages (the even ones: wave 0's slot of `part`), see INTEG_OOL
    double *sums;   // [6][64]  fan-out mode: the velocity part of the next stage's sum and the position part of the one after, formed by the sums wave (fan_sums)
    // epoch data carried between attempts (cfg->ed_reuse fields per lane), behind the ephemeris records
    double *ed0;         // [ed_reuse][64]  stage-0 data of the current attempt (what a rejected attempt starts from again)
    long long *ed0_ep;   // [64]            its epoch
    long long *spec_ep;  // [64]            epoch of the data the almanac wave left in buffer 0 during the last window
    int *ed0st;          // [64]
    // STM variant only
    double *inbD;   // [20][64]       5 dual inputs (zr, zi, rho_u, rho, 1/rho)
    double *pertD;  // [27][64]       a_pm(3) G_pm(9) f_srp/m(3) G_srp/m(9) c_srp(3)
    double *sacc;   // [12][64]       sum_i b_i * (G_i (9, row-major), c_i (3)) of the current attempt
    double *qpre;   // [QPRE_ROWS][64] quad layout: position-only pieces of phase C, formed in the window
    double *partD;  // [P][16][64]    dual harmonics partials
};

DEVFN LdsMap carve_lds(char *smem, int n_waves, bool stm, int rec_lds_doubles, int reuse_fields, bool quad = false) {
    LdsMap m;
    double *p = (double *)smem;
    m.kbuf = p; p += DEV_MAX_STAGES * 6 * (quad ? DEV_LANES / 4 : DEV_LANES);
    m.tabl = p; p += DEV_MAX_STAGES * DEV_MAX_STAGES + 3 * DEV_MAX_STAGES;
    m.ys = p; p += 6 * DEV_LANES;
    m.ed = p; p += 2 * ED_FIELDS * DEV_LANES;
    m.step = p; p += 2 * DEV_LANES;
    m.cs = p; p += CS_FIELDS * DEV_LANES;
    m.part = p; p += quad ? DEV_MAX_WAVES * QSLOT : DEV_MAX_WAVES * 4 * DEV_LANES;  // = DEV_MAX_WAVES_STM * 16 * DEV_LANES: reused for the dual partials
    m.edst = (int *)p; p += DEV_MAX_ALM * DEV_LANES;   // DEV_MAX_ALM * 2 * 64 ints
    m.pertst = (int *)p; p += DEV_LANES;     // 2 x 64 ints
    m.ctl = (int *)p; p += 8;
    m.inbD = m.pertD = m.sacc = m.partD = m.qpre = nullptr;
    if (stm) {
        // the plain inb / pert slots alias the head of their dual counterparts (written first, overwritten after)
        m.inbD = p; m.inb = p; p += (quad ? 10 : 20) * DEV_LANES;
        m.pertD = p; m.pert = p; p += (quad ? 15 : 27) * DEV_LANES;
        m.sacc = p; p += (quad ? 6 : 12) * DEV_LANES;
        m.qpre = p; p += (quad ? QPRE_ROWS : 0) * DEV_LANES;
        m.partD = m.part;
    } else {
        m.inb = p; p += NIN * DEV_LANES;
        m.pert = p; p += 9 * DEV_LANES;
    }
    m.ys2 = m.ys; m.inb2 = m.inb; m.pert2 = m.pert;
    m.ixs = m.part;
    m.sums = m.part;
    if (!stm) {
        m.ys2 = p; p += 6 * DEV_LANES;
        m.inb2 = p; p += NIN * DEV_LANES;
        m.pert2 = p; p += 9 * DEV_LANES;
        m.ixs = p; p += 4 * DEV_LANES;
#if FAN_SUMS
        m.sums = p; p += 6 * DEV_LANES;
#endif
    } else if (quad) {  // pipelined stage loop of the quad layout: second set of the dual buffers
        m.ys2 = p; p += 6 * DEV_LANES;
        m.inb2 = p; p += 10 * DEV_LANES;
        m.pert2 = p; p += 15 * DEV_LANES;
    }
    m.rec = p; p += rec_lds_doubles;
    m.ed0 = p; p += reuse_fields * DEV_LANES;
    m.ed0_ep = (long long *)p; p += DEV_LANES;
    m.spec_ep = (long long *)p; p += DEV_LANES;
    m.ed0st = (int *)p;
    return m;
}

#if !NYX_HOST_TU
static
#else
extern "C"
#endif
size_t nyx_kernel_lds_bytes(int n_waves, int rec_doubles, int stm, int reuse_fields) {  // stm: 0 = plain, 1 = D3, 2 = quad layout
    const bool quad = stm == 2;
    size_t d = (size_t)DEV_MAX_STAGES * 6 * (quad ? DEV_LANES / 4 : DEV_LANES) + DEV_MAX_STAGES * DEV_MAX_STAGES + 3 * DEV_MAX_STAGES + 6 * DEV_LANES +
               2 * ED_FIELDS * DEV_LANES + 2 * DEV_LANES + CS_FIELDS * DEV_LANES + (size_t)(quad ? DEV_MAX_WAVES * QSLOT : DEV_MAX_WAVES * 4 * DEV_LANES) + DEV_MAX_ALM * DEV_LANES +
               DEV_LANES + 8 + (size_t)rec_doubles;
    d += quad ? (size_t)(10 + 15 + 6 + QPRE_ROWS + 6 + 10 + 15) * DEV_LANES : (stm ? (size_t)(20 + 27 + 12) * DEV_LANES : (size_t)(NIN + 9 + 6 + NIN + 9 + 4 + (FAN_SUMS ? 6 : 0)) * DEV_LANES);
    (void)n_waves;
    if (reuse_fields > 0) d += (size_t)reuse_fields * DEV_LANES + 2 * DEV_LANES + DEV_LANES / 2;
    return d * sizeof(double) + 64;
}

// start of a step: final-step test on integer epochs (instance.rs:149-186), then epoch and step published to the other waves
DEVFN void begin_attempt_fn(const LdsMap &L, int lane, ColdState &c) {
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
    L.step[lane] = __longlong_as_double(c.epoch);
    L.step[DEV_LANES + lane] = c.h;
    if (!__any(!c.done)) {
        if (lane == 0) L.ctl[0] = 1;
    }
}

#define PROF_T0() const int64_t pt0_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0
#define PROF_ADD(slot) if (prof_on) prof_acc[slot] += (int64_t)__builtin_readcyclecounter() - pt0_
#define A_ROW(i, j) tabl[(i)*DEV_MAX_STAGES + (j)]
#define B_COEF(i) tabl[DEV_MAX_STAGES * DEV_MAX_STAGES + (i)]
#define BD_COEF(i) tabl[DEV_MAX_STAGES * DEV_MAX_STAGES + DEV_MAX_STAGES + (i)]
#define C_COEF(i) tabl[DEV_MAX_STAGES * DEV_MAX_STAGES + 2 * DEV_MAX_STAGES + (i)]

// NYX_HIP_FLAG_STM_TEXTBOOK: the variational equations d(Phi)/dt = A(t) Phi integrated by the step's own tableau (the form SURVEY 8a-11
// asks to expose beside the reference's Phi_ctx * A).  A(t) does not depend on Phi and the error control does not look at Phi, so
// integrating Phi "in the stage vector" is the same arithmetic as replaying the tableau over the stage matrices A_i of the ACCEPTED
// attempt - which phase C left in `hist` ([stage][12][stride]: G_i row-major, c_i) - once the step is accepted: per column of Phi,
//     Phi_s = Phi + h sum_{j<i} a_ij K_j,   K_i = A_i Phi_s,   Phi_next = Phi + sum_i (h b_i) K_i
// with the oracle's operation order (oracle/nyx_oracle.c, sc_eom / derive: sums from 0.0 with ascending index, products unfused).
// A = [[0 I 0], [G 0 c], [0 0 0]]: rows 0..2 of K are rows 3..5 of Phi_s, rows 3..5 are G Phi_s[0..2] + c Phi_s[6], rows 6..8 of Phi
// never move.  The K_i of a column (16 x 6 per lane) live in the k-buffer, which the attempt no longer needs once it is accepted.
// Out of line: the 64-lane dual kernel has no registers to spare at its call site.
static __device__ __attribute__((noinline)) bool stm_update_textbook(double *phi, double h, const double *hist, int64_t stride, int64_t gid, double *kb,
                                                                  const double *tabl, int stages_v, int lane) {
    const int stages = __builtin_amdgcn_readfirstlane(stages_v);
    bool nan = false;
    for (int col = 0; col < 9; ++col) {
        double p[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) p[e] = phi[e + 9 * col];
        const double gam = phi[6 + 9 * col];
        for (int i = 0; i < stages; ++i) {
            double wi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int j = 0; j < i; ++j) {
                const double a_ij = A_ROW(i, j);
#pragma unroll
                for (int e = 0; e < 6; ++e) wi[e] += a_ij * kb[(j * 6 + e) * DEV_LANES + lane];
            }
            double ps[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) ps[e] = p[e] + h * wi[e];
            double g[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) g[q] = hist[(int64_t)(i * 12 + q) * stride + gid];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double s = g[3 * a + 0] * ps[0];
                s += g[3 * a + 1] * ps[1];
                s += g[3 * a + 2] * ps[2];
                s += g[9 + a] * gam;
                kb[(i * 6 + a) * DEV_LANES + lane] = ps[3 + a];
                kb[(i * 6 + 3 + a) * DEV_LANES + lane] = s;
            }
        }
        double nx[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) nx[e] = p[e];
        for (int i = 0; i < stages; ++i) {
            const double cb = h * B_COEF(i);
#pragma unroll
            for (int e = 0; e < 6; ++e) nx[e] += cb * kb[(i * 6 + e) * DEV_LANES + lane];
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            nan = nan || (nx[e] != nx[e]);
            phi[e + 9 * col] = nx[e];
        }
    }
    return nan;
}


// ---------------------------------------------------------------------------------------------
// INTEG_OOL (round 6): the integrator wave of the sixteen-wave plain kernels, out of line.
//
// In the pipelined stage loop the integrator wave is a serial, latency-bound chain - phase C of stage i - 1 (fold, the helper's answer,
// assembly of k), phase A of stage i, position and recursion inputs of stage i + 1, the mailbox post - and in a cooperative launch the
// two ends of that chain (answer in, post out) close the loop that bounds the owner's period.  Inlined into role_loop at the 128-VGPR
// budget of sixteen waves it kept ~30 doubles live across its three calls per stage (coop_post, fold_partials, coop_wait; the ABI
// preserves 24): 115 scratch loads, 122 stores and 475 SGPR-spill lane moves per stage loop (tests/golden/code_budget.json, round 5),
// every reload a trip to L2 on the critical path, and every scratch reload behind a post also waits for the post's uncached stores
// (loads and stores share vmcnt on gfx9).  Here the chain is TWO functions with register files of their own that talk through LDS -
// the treatment phase_c_quad got in round 4 -:
//   integ_front(i): phase A of stage i (velocity of the stage state; the position was published a window earlier), then position,
//                   DCM rotation, recursion inputs of stage i + 1 into LDS and the mailbox post;
//   integ_back(i):  phase C of stage i behind the stage barrier: fold of the fifteen partial sums, the helper's answer, s / t / u /
//                   (mu / r) / R_eq and the stage's DCM read HERE (not carried from phase A), assembly of the acceleration, k_i.
// What role_loop keeps across the two calls is the velocity part of the next stage sum and the position part of the one after (six
// doubles) - inside the callee-saved set.  Two protocol consequences: (1) s, t, u, (mu / r) / R_eq of stage i + 1 are written in window
// i and read in phase C(i + 1), AFTER window i + 1 has written those of stage i + 2: two row sets by stage parity (wave 0's slot of the
// partial sums and LdsMap.ixs); (2) phase C(i) reads the DCM of stage i from the epoch data behind B2(i), when the almanac wave is
// about to write the DCM of stage i + 2 over it: the almanac wave holds that write until the fold counter (ctl[3]) says phase C(i) has
// its operands (epoch_data `gate`; the column waves wait on the same word before they overwrite their partial sums).
// Same operations on the same operands in the same order as the inline code: bit-identical results (digests in tests/).
// ---------------------------------------------------------------------------------------------
#ifndef INTEG_OOL
#define INTEG_OOL ((NYX_EMIT & (NYX_EMIT_PLAIN16 | NYX_EMIT_PLAIN16_P2 | NYX_EMIT_PLAIN16_FAN)) ? 1 : 0)
#endif
#ifndef IX_SUMS_OOL
#define IX_SUMS_OOL 0   /* 1: the window's two stage sums out of line too (integ_sums) - built and measured in round 6, same box, 24 h of configs[1]: 610 ms against 598.5 inline (fan-out shard of 1 250: 399 against 392): branch-free, it issues five times the VALU instructions of the branchy inline loops on the SIMD that also hosts three column waves */
#endif
#if INTEG_OOL
#define IX_HOT 1       /* phase A from the position the previous window published (else: the caller did phase A, v3..5 are the stage velocity) */
#define IX_SPEC_NOW 2  /* stage 0 of this attempt was published speculatively */
#define IX_COOP 4      /* this workgroup shares its columns with the helpers */
#define IX_PROF 8
#define IX_SHARED 16   /* integ_back: the column waves of THIS stage left columns to a helper */
#define IXR_ANSWER 0x10000
#define IXR_FALLBACK 0x20000
DEVFN char *lds_from_u32(uint32_t a) { return (char *)(__attribute__((address_space(3))) char *)(uintptr_t)a; }
// An LDS array's row base for this lane as ONE address register the optimiser cannot take apart: the carve's offsets are constants
// beyond the 16-bit offset field of the ds instructions, and folded into every access they cost an address VGPR per row (the first
// cut of integ_back: sixty of them, all 48 callee-saved VGPRs saved and restored per call).  Rows are then base[row * DEV_LANES].
DEVFN LdsPtr ix_rows(const double *arr, int lane) {
    uint32_t a = (uint32_t)(uintptr_t)(LdsCPtr)arr + (uint32_t)lane * 8u;
    asm volatile("" : "+v"(a));
    return (LdsPtr)(uintptr_t)a;
}
DEVFN void ix_stamp(LdsFlagPtr ctl, int k) {  // (accounting twin only) a 64-bit cycle stamp in two control words
    const int64_t t = (int64_t)__builtin_readcyclecounter();
    ctl[8 + 2 * k] = (int)(uint32_t)t; ctl[9 + 2 * k] = (int)(uint32_t)(t >> 32);
}
// kbuf / tabl / L in scope: the KB / A_ROW / B_COEF / CS_Y macros of role_loop
#define IX_PROLOGUE                                                                                                        \
    CfgPtr cfg = (CfgPtr)uniform_u64(cfg_u);                                                                               \
    const LdsMap L = carve_lds(lds_from_u32(__builtin_amdgcn_readfirstlane(lds_v)), 0, false, cfg->rec_in_lds ? cfg->rec_doubles : 0, cfg->ed_reuse, false); \
    double *const kbuf = L.kbuf;                                                                                           \
    double *const tabl = L.tabl;                                                                                           \
    constexpr int KB_STR = DEV_LANES;                                                                                      \
    const int kb_li = lane;                                                                                                \
    const int i = __builtin_amdgcn_readfirstlane(i_v);                                                                     \
    const int flags = __builtin_amdgcn_readfirstlane(flags_v);                                                             \
    const int stages = cfg->stages;

static __device__ __attribute__((noinline)) int integ_front(uint32_t lds_v, uint64_t cfg_u, int i_v, int flags_v, int lane, double h,
                                                           double v3, double v4, double v5, double p0, double p1, double p2,
                                                           uint64_t cbox_u, uint64_t posted_u, uint32_t seq_nx_v, int keep_k0) {
    IX_PROLOGUE
    const bool has_grav = cfg->has_grav != 0;
    const bool need_almanac = has_grav || cfg->has_drag != 0 || cfg->has_tides != 0 || cfg->n_slots > 0;
    const bool spec = cfg->spec != 0;
    int st = NYX_HIP_OK;
    double vel[3] = {v3, v4, v5};
    if (flags & IX_HOT) {
        // ---- Phase A: the velocity of the stage state (instance.rs:376-394); its position was published in the previous window
        double *const ysb = (i & 1) ? L.ys2 : L.ys;
        if (i == 0) {
            // speculative stage 0: the state step control has just stored (accepted lanes: its position IS the published one, bit for
            // bit; rejected lanes: the result of this stage is dropped, k_0 stands)
#pragma unroll
            for (int e = 0; e < 3; ++e) vel[e] = CS_Y(3 + e);
        } else {
            const double a_last = A_ROW(i, i - 1);
            const double w[3] = {v3, v4, v5};   // (the velocity part of sum_{j < i-1} a_ij k_j, accumulated in the previous window)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const double wi = w[e] + a_last * KB(i - 1, 3 + e);
                vel[e] = CS_Y(3 + e) + h * wi;
            }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) ysb[(3 + e) * DEV_LANES + lane] = vel[e];
#if defined(NYX_COOP_FAN) && FAN_SUMS
        if (cfg->has_drag || cfg->sums_wave1 != 0) {  // (... and the sums wave, which adds this stage's velocity term last: fan_sums)
#else
        if (cfg->has_drag) {  // the perturbation wave is already in this stage's window; drag is the one term that wants the velocity
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) LCTL[4] = i + 1;
        }
        // (the almanac wave finished this stage's data before the barrier this wave has just passed)
        if (need_almanac && !(i == 0 && keep_k0)) {  // (a rejected lane's stage 0 is not evaluated: its epoch data at t + h does not count)
            const int n_alm = cfg->n_alm;
            for (int a = 0; a < n_alm; ++a) {
                const int es = L.edst[(2 * a + (i & 1)) * DEV_LANES + lane];
                if (es) st = es;
            }
        }
    }
    if (i + 1 < stages || spec) {
        // ---- position and recursion inputs of stage i+1, published inside the window of stage i.
        // k_i[0..2] is this stage's velocity, so  y + h (pre + a_{i+1,i} k_i)  is complete for the position
        double nx_pos[3];
        const double pre[3] = {p0, p1, p2};
        if (i + 1 < stages) {
            const double a_nl = A_ROW(i + 1, i);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const double wi = pre[e] + a_nl * vel[e];
                nx_pos[e] = CS_Y(e) + h * wi;
            }
        } else {
            // last window: stage 0 of the next attempt, should this one be accepted - the position step control will form
            // (next[e] = y[e]; next[e] += (h b_j) k_j[e], j ascending: y + the terms j < i were added up in the previous window)
            const double cb = h * B_COEF(i);
#pragma unroll
            for (int e = 0; e < 3; ++e) nx_pos[e] = pre[e] + cb * vel[e];
        }
        double *const ysn = ((i + 1) & 1) ? L.ys2 : L.ys;
        double *const inbn = ((i + 1) & 1) ? L.inb2 : L.inb;
#pragma unroll
        for (int e = 0; e < 3; ++e) ysn[e * DEV_LANES + lane] = nx_pos[e];
        if (has_grav) {  // (without a gravity field the position is all the next window needs)
            if (need_almanac) {  // the almanac wave writes the DCM of stage i+1 first thing in this window
                int spin = 0;
                while (LCTL[2] != i + 1 && ++spin < 4000000) __builtin_amdgcn_s_sleep(4);
                if (spin >= 4000000) st = NYX_HIP_ERR_NAN;  // (bounded: a protocol error must end as a failed run, never as a hung GPU)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            const double *const edn = L.ed + ((i + 1) & 1) * ED_FIELDS * DEV_LANES;  // (its DCM: the flag is raised before the body positions are evaluated)
            double m_nx[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) m_nx[q] = edn[q * DEV_LANES + lane];
            const double rb0 = m_nx[0] * nx_pos[0] + m_nx[1] * nx_pos[1] + m_nx[2] * nx_pos[2];
            const double rb1 = m_nx[3] * nx_pos[0] + m_nx[4] * nx_pos[1] + m_nx[5] * nx_pos[2];
            const double rb2 = m_nx[6] * nx_pos[0] + m_nx[7] * nx_pos[1] + m_nx[8] * nx_pos[2];
            const double r_ = norm3(rb0, rb1, rb2);
            const double inv_r = 1.0 / r_;
            const double nx_s = rb0 * inv_r, nx_t = rb1 * inv_r, nx_u = rb2 * inv_r;
            const double rho = cfg->g_re * inv_r;
            const double nx_kfac = (cfg->g_mu * inv_r) * cfg->g_inv_re;
            inbn[0 * DEV_LANES + lane] = rho * nx_s;
            inbn[1 * DEV_LANES + lane] = rho * nx_t;
            inbn[2 * DEV_LANES + lane] = rho * nx_u;
            inbn[3 * DEV_LANES + lane] = rho;
            inbn[4 * DEV_LANES + lane] = r_ * cfg->g_inv_re;
            double *const sx = ((i + 1) & 1) ? L.ixs : L.part;  // (read back in phase C of stage i + 1: integ_back)
            sx[0 * DEV_LANES + lane] = nx_s; sx[1 * DEV_LANES + lane] = nx_t; sx[2 * DEV_LANES + lane] = nx_u; sx[3 * DEV_LANES + lane] = nx_kfac;
        }
        if (lane == 0) L.ctl[1] = (flags & IX_COOP) ? 1 : 0;  // the workers read it after B2(i), for stage i+1
        if ((flags & IX_COOP) && has_grav) {
            CoopBox *const cbox = (CoopBox *)uniform_u64(cbox_u);
            uint32_t *const posted = (uint32_t *)uniform_u64(posted_u);
            const uint32_t seq_nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)seq_nx_v);
            if (flags & IX_PROF) ix_stamp(LCTL, 1);
            coop_post_inl(cbox, posted, lane, seq_nx, (LdsCPtr)inbn, COOP_PARTS_HERE);  // (inline: this function stays a leaf)
            if (flags & IX_PROF) ix_stamp(LCTL, 2);
        }
    }
    return st;
}

// The two stage sums the integrator's window forms beside the column walk, out of line as well (round 6): the velocity part of
// sum_{j<i} a_{i+1,j} k_j (phase A of the next stage adds the newest term) and the position part of the sum the NEXT window publishes
// from (stage i + 2: j < i, then this stage's velocity; or, when the next window is the last of a chained attempt, y + sum (h b_j) k_j).
// Inline in role_loop these were two loops of up to fourteen iterations with a uniform branch and an LDS round trip each - ~7 k cycles
// of the integrator's ~19 k busy per evaluation, which is the owner's whole period once dedicated helpers carry its columns (fan-out
// mode).  Here: the tableau rows as scalar loads from DevCfg (the same doubles propagate_body staged into LDS), the k rows in two
// branch-free batches of seven stages (absent stages select +0.0 operands: +0.0 * +0.0 added to a sum that started from +0.0 leaves
// its bits alone), the additions in the same ascending order: bit-identical sums.  A leaf inside the caller-saved registers.
struct IxSums {
    double w3, w4, w5, p0, p1, p2;
};
template <int COMP0>
DEVFN void ix_sum_rows(const LdsPtr kb0, const CAS double *coef, double scale, bool scaled, int i, double (&acc)[3]) {
    // acc[e] += c_j * k_j[COMP0 + e], j = 0 .. i - 1 ascending; c_j = coef[j], or scale * coef[j] (the h b_j of step control's sum)
#pragma unroll
    for (int j0 = 0; j0 < DEV_MAX_STAGES - 2; j0 += 7) {
        if (j0 < i) {  // (uniform)
            double c[7], k[7][3];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const int j = j0 + q;
                const bool on = j < i;  // (uniform)
                const double cj = coef[on ? j : 0];
                c[q] = on ? (scaled ? scale * cj : cj) : 0.0;
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const double kv = kb0[(j * 6 + COMP0 + e) * DEV_LANES];
                    k[q][e] = on ? kv : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < 7; ++q) {
#pragma unroll
                for (int e = 0; e < 3; ++e) acc[e] += c[q] * k[q][e];
            }
        }
    }
}
static __device__ __attribute__((noinline)) IxSums integ_sums(uint32_t lds_v, uint64_t cfg_u, int i_v, int lane, double h, double v3, double v4, double v5) {
    const int flags_v = 0;
    IX_PROLOGUE
    (void)flags; (void)kbuf; (void)kb_li; (void)KB_STR; (void)tabl;
    const bool spec = cfg->spec != 0;
    const LdsPtr kb0 = ix_rows(L.kbuf, lane);
    const double vel[3] = {v3, v4, v5};
    double w[3] = {0.0, 0.0, 0.0}, p[3] = {0.0, 0.0, 0.0};
    if (i + 1 < stages) ix_sum_rows<3>(kb0, cfg->a + (i + 1) * i / 2, 0.0, false, i, w);
    if (i + 2 < stages) {
        const CAS double *row = cfg->a + (i + 2) * (i + 1) / 2;
        ix_sum_rows<0>(kb0, row, 0.0, false, i, p);
        const double a_ni = row[i];
#pragma unroll
        for (int e = 0; e < 3; ++e) p[e] += a_ni * vel[e];
    } else if (i + 2 == stages && spec) {
        // the next window is the last: it publishes stage 0 of the next attempt, y + sum_j (h b_j) k_j
#pragma unroll
        for (int e = 0; e < 3; ++e) p[e] = CS_Y(e);
        ix_sum_rows<0>(kb0, cfg->b, h, true, i, p);
        const double cbi = h * cfg->b[i];
#pragma unroll
        for (int e = 0; e < 3; ++e) p[e] += cbi * vel[e];
    }
    IxSums r = {w[0], w[1], w[2], p[0], p[1], p[2]};
    return r;
}

// Phase C of stage i (orbital.rs:80-114, spacecraft.rs:227-243), behind the stage barrier.  (a0, a1, a2): the two-body term formed in the
// window.  A LEAF like integ_front (a function that keeps values live across calls of its own has to save the callee-saved registers it
// uses in its prologue - fifty scratch stores and loads per call, measured on the first cut of this function): the wait for the helper's
// answer is inlined, and the one thing that needs a call - walking the helper's columns here when no answer comes, coop_fallback - is
// left to the caller: the function then returns IXR_NEED_FB with its own fifteen-slot fold in (px..pw) and the caller finishes the
// stage through integ_back_slow.  `ret`: status of the second field's orientation (low 16 bits) | IXR_ANSWER (a helper answered) |
// IXR_NEED_FB.  skip_k (per lane): a rejected lane's speculative stage 0 (nothing of it is kept).
struct IxBack {
    double px, py, pz, pw;
    int ret;
};
#define IXR_NEED_FB 0x40000
DEVFN void ix_assemble(CfgPtr cfg, const LdsMap &L, int i, int lane, double (&acc)[3], double px, double py, double pz, double pw,
                       const double (&m_cur)[9], double s_, double t_, double u_, double kfac, int skip_k) {
    const LdsPtr pertc = ix_rows((i & 1) ? L.pert2 : L.pert, lane);
    const LdsPtr ysb = ix_rows((i & 1) ? L.ys2 : L.ys, lane);
    const LdsPtr kb = ix_rows(L.kbuf + i * 6 * DEV_LANES, lane);
    if (cfg->has_grav) {
        px *= kfac; py *= kfac; pz *= kfac; pw *= kfac;
        const double al0 = px + pw * s_, al1 = py + pw * t_, al2 = pz + pw * u_;
        acc[0] += m_cur[0] * al0 + m_cur[3] * al1 + m_cur[6] * al2;
        acc[1] += m_cur[1] * al0 + m_cur[4] * al1 + m_cur[7] * al2;
        acc[2] += m_cur[2] * al0 + m_cur[5] * al1 + m_cur[8] * al2;
    }
    if (cfg->has_srp) {
        acc[0] += pertc[3 * DEV_LANES]; acc[1] += pertc[4 * DEV_LANES]; acc[2] += pertc[5 * DEV_LANES];
    }
    if (cfg->has_drag) {
        acc[0] += pertc[6 * DEV_LANES]; acc[1] += pertc[7 * DEV_LANES]; acc[2] += pertc[8 * DEV_LANES];
    }
    if (!skip_k) {
        // k_i = [velocity of the stage state, f(x)]
        kb[0 * DEV_LANES] = ysb[3 * DEV_LANES]; kb[1 * DEV_LANES] = ysb[4 * DEV_LANES]; kb[2 * DEV_LANES] = ysb[5 * DEV_LANES];
        kb[3 * DEV_LANES] = acc[0]; kb[4 * DEV_LANES] = acc[1]; kb[5 * DEV_LANES] = acc[2];
    }
}
static __device__ __attribute__((noinline)) IxBack integ_back(uint32_t lds_v, uint64_t cfg_u, int i_v, int flags_v, int lane, double a0, double a1,
                                                             double a2, double px, double py, double pz, double pw, uint32_t seq_cur_v, int fold_val_v,
                                                             uint64_t cbox_u, uint64_t out2_u, int skip_k) {
    IX_PROLOGUE
    (void)stages; (void)tabl; (void)kbuf; (void)kb_li; (void)KB_STR;
    const bool has_grav = cfg->has_grav != 0, has_grav2 = cfg->has_grav2 != 0;
#ifdef NYX_NO_TIDES
    const bool has_tides = false;
#else
    const bool has_tides = cfg->has_tides != 0;
#endif
    const bool has_pm = cfg->n_pm > 0;
    IxBack out = {0.0, 0.0, 0.0, 0.0, 0};
    double acc[3] = {a0, a1, a2};
    {
        const LdsPtr pertc = ix_rows((i & 1) ? L.pert2 : L.pert, lane);
        if (has_pm || has_tides || has_grav2) {
            acc[0] += pertc[0 * DEV_LANES]; acc[1] += pertc[1 * DEV_LANES]; acc[2] += pertc[2 * DEV_LANES];
        }
    }
    if (has_grav2 && !skip_k) {  // the second field's orientation status of THIS stage (a rejected lane's speculative stage 0 does not count)
        const int es = L.pertst[(i & 1) * DEV_LANES + lane];
        if (es) out.ret = es & 0xffff;
    }
    // (px..pw: the fold of the fifteen partial sums, made by the caller through fold_partials - sixty reads that want a register file of
    //  their own: inlined here they pushed this function into the callee-saved registers)
    double m_cur[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double s_ = 0.0, t_ = 0.0, u_ = 0.0, kfac = 0.0;
    if (has_grav) {
        // the operands phase C keeps from the stage's own data: its DCM (the almanac wave overwrites those rows once ctl[3] moves) and
        // s, t, u, (mu / r) / R_eq from the rows the publishing window left them in
        const LdsPtr edc = ix_rows(L.ed + (i & 1) * ED_FIELDS * DEV_LANES, lane);
        const LdsPtr sx = ix_rows((i & 1) ? L.ixs : L.part, lane);
#pragma unroll
        for (int q = 0; q < 9; ++q) m_cur[q] = edc[q * DEV_LANES];
        s_ = sx[0 * DEV_LANES]; t_ = sx[1 * DEV_LANES]; u_ = sx[2 * DEV_LANES]; kfac = sx[3 * DEV_LANES];
        {   // the partial sums of stage i and the DCM are in registers: the workers may overwrite their slots, the almanac wave its rows
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) LCTL[3] = __builtin_amdgcn_readfirstlane(fold_val_v);
        }
        if (flags & IX_SHARED) {
            if (flags & IX_PROF) ix_stamp(LCTL, 3);
            CoopAnswer ans = {0.0, 0.0, 0.0, 0.0, 0};
            if (flags & IX_COOP) {
                CoopBox *const cbox = (CoopBox *)uniform_u64(cbox_u);
                const uint32_t seq_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)seq_cur_v);
#if COOP_PARTS_HERE == 2
                ans = coop_wait2_inl(cbox, (CoopOut *)uniform_u64(out2_u), lane, seq_cur);
#else
                ans = coop_wait_inl(cbox, lane, seq_cur);
#endif
            }
            if (flags & IX_PROF) ix_stamp(LCTL, 0);
            if (!ans.ok) {  // (uniform) no answer in time: the caller walks the helper's columns and finishes the stage (integ_back_slow)
                out.px = px; out.py = py; out.pz = pz; out.pw = pw;
                out.ret |= IXR_NEED_FB;
                return out;
            }
            px += ans.x; py += ans.y; pz += ans.z; pw += ans.w;  // + the helper's columns
            out.ret |= IXR_ANSWER;
        } else {
            px += 0.0; py += 0.0; pz += 0.0; pw += 0.0;  // (the inline code adds the helper's share unconditionally: 0.0 when working alone)
        }
    }
    ix_assemble(cfg, L, i, lane, acc, px, py, pz, pw, m_cur, s_, t_, u_, kfac, skip_k);
#if defined(NYX_COOP_FAN) && FAN_SUMS
    if (cfg->sums_wave1 != 0) {  // k_i is written: the sums wave may add its term (fan_sums; ctl[6] counts like the fold counter)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) LCTL[6] = __builtin_amdgcn_readfirstlane(fold_val_v);
    }
#endif
    return out;
}
// The rare other half of integ_back: the helper did not answer, the caller has walked its columns (fx..fw) on top of the fold (px..pw).
// The stage's DCM is no longer in LDS (the almanac wave was told it may overwrite those rows) and is evaluated again - the same
// function of the stage epoch the almanac wave evaluates, bit for bit (rotation_dcm_iau_poly's base depends on the lane's epoch alone).
static __device__ __attribute__((noinline)) void integ_back_slow(uint32_t lds_v, uint64_t cfg_u, uint64_t rec_u, int i_v, int lane, double a0, double a1, double a2,
                                                                double px, double py, double pz, double pw, double fx, double fy, double fz, double fw, int skip_k) {
    const int flags_v = 0;
    IX_PROLOGUE
    (void)stages; (void)flags; (void)kbuf; (void)kb_li; (void)KB_STR;
    double acc[3] = {a0, a1, a2};
    const double *const pertc = (i & 1) ? L.pert2 : L.pert;
#ifdef NYX_NO_TIDES
    const bool has_tides = false;
#else
    const bool has_tides = cfg->has_tides != 0;
#endif
    if (cfg->n_pm > 0 || has_tides || cfg->has_grav2 != 0) {
        acc[0] += pertc[0 * DEV_LANES + lane]; acc[1] += pertc[1 * DEV_LANES + lane]; acc[2] += pertc[2 * DEV_LANES + lane];
    }
    const int64_t ep = __double_as_longlong(L.step[lane]) + seconds_to_ns(C_COEF(i) * L.step[DEV_LANES + lane]);
    double m_cur[9];
    if (cfg->dcm_incr) {
        RotBase rb;
        rb.ep = INT64_MIN;
#pragma unroll
        for (int q = 0; q < 3; ++q) { rb.sn[q] = 0.0; rb.cs[q] = 1.0; }
        rotation_dcm_iau_poly(cfg->g_rot, ep, rb, m_cur);
    } else {
        const double *records = cfg->rec_in_lds ? (const double *)L.rec : (const double *)uniform_u64(rec_u);
        (void)rotation_dcm(cfg, cfg->g_rot, records, ns_to_seconds(ep), m_cur);
    }
    const double *const sx = (i & 1) ? L.ixs : L.part;
    const double s_ = sx[0 * DEV_LANES + lane], t_ = sx[1 * DEV_LANES + lane], u_ = sx[2 * DEV_LANES + lane], kfac = sx[3 * DEV_LANES + lane];
    px += fx; py += fy; pz += fz; pw += fw;
    ix_assemble(cfg, L, i, lane, acc, px, py, pz, pw, m_cur, s_, t_, u_, kfac, skip_k);
}

// Step control out of line (round 6): error estimate, accept / reject, the next step size, the accepted state and - chained attempts -
// the next attempt opened (derive(), instance.rs:401-493).  Inline in role_loop it ran on what the stage loop's carried values left of
// the 128 VGPRs (59 scratch loads in the integrator's tail) and took ~20 k cycles per attempt, all of them between the last stage's
// phase C and the first window of the next attempt - the one place where the column waves wait for the integrator (they walk the
// speculative stage 0 in ~26 k cycles; phase C + step control + the first window's post took ~31 k).  Here: a leaf with a register
// file of its own, the cold state and the k-buffer through one address register each, the tableau's b / b - b* as scalar loads from
// DevCfg (the doubles propagate_body staged into LDS), the k rows of four stages loaded together.  Same operations on the same
// operands in the same order: bit-identical results.  Not here: stop conditions (a call: role_loop keeps its inline step control for
// launches with an event) and the dense output (the caller writes it from the cold state this function stored).
#define IXS_ACCEPT 1
#define IXS_KEEP_K0 2
#define IXS_CHAIN 1   /* flags: chained attempts - open the next attempt and publish it (ctl[5]) */
struct IxStep {
    double h_next;
    int ret;
};
static __device__ __attribute__((noinline)) IxStep integ_step(uint32_t lds_v, uint64_t cfg_u, int lane, double h, int st_att, int att_v, int flags_v) {
    const int i_v = 0;
    IX_PROLOGUE
    (void)i; (void)kbuf; (void)tabl; (void)kb_li; (void)KB_STR;
    const LdsPtr cs = ix_rows(L.cs, lane);
    const LdsPtr kb0 = ix_rows(L.kbuf, lane);
    ColdState c;
    cold_load(cs, 0, c);
    double *const y = c.y;
    if (!c.done) c.n_evals += stages;
    // ---- next state and error estimate (instance.rs:401-414).  d(Cr, Cd, prop mass)/dt = 0.
    double next[9], err[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) { next[e] = y[e]; err[e] = 0.0; }
    int j0 = 0;
    for (; j0 + 4 <= stages; j0 += 4) {  // (uniform) four stages per batch: the loads first, the additions in ascending stage order
        double ce[4], cb[4], kv[4][6];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ce[q] = h * cfg->bdiff[j0 + q];
            cb[q] = h * cfg->b[j0 + q];
#pragma unroll
            for (int e = 0; e < 6; ++e) kv[q][e] = kb0[((j0 + q) * 6 + e) * DEV_LANES];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                err[e] += ce[q] * kv[q][e];
                next[e] += cb[q] * kv[q][e];
            }
        }
    }
    for (; j0 < stages; ++j0) {
        const double ce = h * cfg->bdiff[j0];
        const double cb = h * cfg->b[j0];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const double kv = kb0[(j0 * 6 + e) * DEV_LANES];
            err[e] += ce * kv;
            next[e] += cb * kv;
        }
    }
    bool accept = false, keep = false;
    // the error estimate, the accept test and the controller's power for every lane at once, in front of the branches (STEP_ONE_POW)
    double de = c.det_error, pw = 0.0;
    bool take = false;
    if (__any(!c.done && st_att == NYX_HIP_OK && !c.fixed)) {  // (uniform)
        de = error_estimate(cfg->error_ctrl, err, next, y);
        take = de <= cfg->tol || h <= cfg->min_step_s || c.attempts >= cfg->attempts;
        pw = pow(cfg->tol / de, take ? cfg->inv_order : cfg->inv_order_m1);
    }
    if (!c.done) {
        if (st_att != NYX_HIP_OK) {
            c.status = st_att;
            c.done = true;
        } else if (c.fixed) {
            c.det_step = c.step_size;
            accept = true;
        } else {
            c.det_error = de;
            if (take) {
                bool nan = false;
#pragma unroll
                for (int e = 0; e < 9; ++e) nan = nan || (next[e] != next[e]);
                if (nan) {
                    c.status = NYX_HIP_ERR_NAN;
                    c.done = true;
                } else {
                    c.det_step = seconds_to_ns(h);
                    if (c.det_error < cfg->tol) {
                        const double prop = 0.9 * h * pw;
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
                const double prop = 0.9 * h * pw;
                h = (prop < cfg->min_step_s) ? cfg->min_step_s : prop;
                keep = true;
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
    IxStep out = {0.0, (accept ? IXS_ACCEPT : 0) | (keep ? IXS_KEEP_K0 : 0)};
    if (flags & IXS_CHAIN) {
        // with chained attempts the next one is opened first (all lanes together: the exit test is a wave vote), the other waves
        // are waiting for its epoch and step
        begin_attempt_fn(L, lane, c);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) LCTL[5] = __builtin_amdgcn_readfirstlane(att_v) + 1;  // the almanac wave waits for this word before it reads the new epoch and step
        out.h_next = c.h;
    }
    cold_store(cs, 0, c);
    return out;
}

#if defined(NYX_COOP_FAN) && FAN_SUMS
// FAN-OUT mode: the integrator's two stage sums on a wave of their own (round 6).  With dedicated helpers an owner's period IS its
// integrator's chain (~19 k cycles per evaluation: integ_front 4.4 k, read-back + two-body + the two sums 8.3 k, fold + integ_back 5.3 k,
// step control 0.7 k), while thirteen column waves of the workgroup hold three rows between them.  One of them (DevCfg.sums_wave1)
// forms, in the window of stage i, what the integrator's window formed behind its post:
//     W = sum_{j<i} a_{i+1,j} k_j[3..5]                          (phase A of stage i + 1 adds the newest term)
//     P = sum_{j<i} a_{i+2,j} k_j[0..2] + a_{i+2,i} v_i          (the position part the NEXT window publishes from;
//         or, when that window is the last of a chained attempt,  y + sum_{j<i} (h b_j) k_j[0..2] + (h b_i) v_i)
// - the terms j <= i - 2 at once (their k rows were complete before the barrier this window starts behind), the term j = i - 1 when
// the integrator's phase C of stage i - 1 has written k_{i-1} (ctl[6], raised by integ_back), the velocity term when integ_front has
// stored v_i (ctl[4]) - and leaves the six values in LdsMap.sums, which the integrator reads behind the stage barrier, in front of the
// next integ_front.  The same additions in the same order as the inline sums: bit-identical results.  Every spin is bounded; a wait
// that expires leaves NaNs, which end the step as NYX_HIP_ERR_NAN.
static __device__ __attribute__((noinline)) void fan_sums(uint32_t lds_v, uint64_t cfg_u, int i_v, int lane, int flags_v, int kdone_v) {
    IX_PROLOGUE
    (void)kbuf; (void)kb_li; (void)KB_STR;
    const bool spec = cfg->spec != 0;
    const LdsPtr kb0 = ix_rows(L.kbuf, lane);
    const LdsPtr ysb = ix_rows((i & 1) ? L.ys2 : L.ys, lane);
    const LdsPtr out = ix_rows(L.sums, lane);
    const bool need_w = i + 1 < stages, need_p = i + 2 < stages, need_b = !need_p && i + 2 == stages && spec;  // (uniform)
    // the tableau from its LDS copy (uniform addresses: broadcast reads that queue with the k rows; scalar loads would drain the LDS queue
    // at every wait): rows i + 1 and i + 2 of A, or h b for the last window of a chained attempt
    const LdsCPtr row_w = (LdsCPtr)tabl + (need_w ? (i + 1) * DEV_MAX_STAGES : 0);
    const LdsCPtr row_p = (LdsCPtr)tabl + (need_p ? (i + 2) * DEV_MAX_STAGES : DEV_MAX_STAGES * DEV_MAX_STAGES);
    double w[3] = {0.0, 0.0, 0.0}, p[3] = {0.0, 0.0, 0.0};
    double hh = 1.0;
    bool bad = false;
    if (need_b) {
        hh = L.step[DEV_LANES + lane];
#pragma unroll
        for (int e = 0; e < 3; ++e) p[e] = CS_Y(e);
    }
    const bool any_p = need_p || need_b;
    // one term: w += a_{i+1,j} k_j[3..5];  p += a_{i+2,j} k_j[0..2]  (or (h b_j) k_j[0..2])
    auto term = [&](const int j) __attribute__((always_inline)) {
        if (need_w) {
            const double a_nj = row_w[j];
#pragma unroll
            for (int e = 0; e < 3; ++e) w[e] += a_nj * kb0[(j * 6 + 3 + e) * DEV_LANES];
        }
        if (any_p) {
            const double c_nj = need_b ? hh * row_p[j] : row_p[j];
#pragma unroll
            for (int e = 0; e < 3; ++e) p[e] += c_nj * kb0[(j * 6 + e) * DEV_LANES];
        }
    };
    if (need_w || any_p) {
        const int nh = i - 1;  // the terms j < i - 1: their k rows were complete before the barrier this window starts behind
        int j = 0;
        for (; j + 4 <= nh; j += 4) {  // four terms per batch: the loads together, the additions in ascending j
            double cw[4], cp[4], kv[4][6];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cw[q] = row_w[j + q];
                cp[q] = row_p[j + q];
#pragma unroll
                for (int e = 0; e < 6; ++e) kv[q][e] = kb0[((j + q) * 6 + e) * DEV_LANES];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (need_w) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) w[e] += cw[q] * kv[q][3 + e];
                }
                if (any_p) {
                    const double c_nj = need_b ? hh * cp[q] : cp[q];
#pragma unroll
                    for (int e = 0; e < 3; ++e) p[e] += c_nj * kv[q][e];
                }
            }
        }
        for (; j < nh; ++j) term(j);
        if (i >= 1) {
            const int want = __builtin_amdgcn_readfirstlane(kdone_v);
            int spin = 0;
            while (LCTL[6] < want && ++spin < 4000000) __builtin_amdgcn_s_sleep(1);
            if (spin >= 4000000) bad = true;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            term(i - 1);
        }
    }
    // (always behind the velocity flag of this window: the integrator reads the previous window's six values in front of integ_front,
    //  which raises it - the rows are free then)
    if (flags & 1) {  // (a stage whose velocity integ_front forms in this window; else: stage 0 of an attempt opened behind barriers)
        int spin = 0;
        while (LCTL[4] != i + 1 && ++spin < 4000000) __builtin_amdgcn_s_sleep(1);
        if (spin >= 4000000) bad = true;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (any_p) {
        const double cv = need_b ? hh * row_p[i] : row_p[i];
#pragma unroll
        for (int e = 0; e < 3; ++e) p[e] += cv * ysb[(3 + e) * DEV_LANES];
    }
    if (bad) {
        const double qn = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
        for (int e = 0; e < 3; ++e) { w[e] = qn; p[e] = qn; }
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) { out[e * DEV_LANES] = w[e]; out[(3 + e) * DEV_LANES] = p[e]; }
}
#endif
#endif  // INTEG_OOL

// ---------------------------------------------------------------------------------------------
// The covariance-mapping loop in ONE launch (round 6; STM kernels, DevBatch.pred).
//
// KalmanODProcess::predict_until (od/process/mod.rs:440-486) is, per trajectory, `for_duration(max_step)` - then
// KalmanFilter::time_update (od/kalman/filtering.rs:59-99) - then reset_stm(), until the end epoch.  Rounds 2-5 enqueued one segment
// launch of this kernel and one nyx_time_update_kernel (predict_kernel.hip) per segment: 120 launches for the sixty one-minute updates of
// BASELINE config 4, and a segment launch costs ~30 us beyond its sixteen force evaluations (round 6, tools/seg_cost.py: 121.6 us per
// RK89 step + 29.8 us per launch - LDS zeroing, table staging, a cold instruction cache, the first attempt's barriers, the launch itself),
// a fifth of the loop.  Here the workgroup stays resident: when every one of its trajectories has finished its segment, the integrator
// wave performs their time updates - this function, the arithmetic of nyx_time_update_kernel operation for operation (nalgebra's order:
// k ascending, multiply then add) with the wave's 64 lanes over the 81 elements -, resets Phi, and re-arms the trajectories that go on;
// the other waves wait at the attempt barrier as they do between any two attempts and see nothing but the next attempt's epoch and step.
// Same states, same Phi, same covariances as the launch-per-segment loop (tests/test_gpu_predict.py).
// `scr`: 256 doubles of LDS scratch per wave (its slot of the partial sums: idle between attempts, re-zeroed before returning).
// cs: the cold state rows (the trajectories' epochs, states and status words).  go[lane of the trajectory] = 1 if it goes on.
// ---------------------------------------------------------------------------------------------
// (what this wave - other lanes of it, or this function a segment earlier - stored in the same launch is read past the L1: the separate
//  kernels of the launch-per-segment loop had a kernel boundary between a store and its reader)
DEVFN double ld_l2(const double *p) { return __longlong_as_double(__hip_atomic_load((const long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
DEVFN int64_t ld_l2(const int64_t *p) { return (int64_t)__hip_atomic_load((const long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN int32_t ld_l2(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __attribute__((noinline)) void segment_update(const PredictArgs *pa_g, double *o_stm, const double *cs_g, double *scr_g, int *go_g, int lane, int quad_v,
                                                               int64_t gid0, int64_t n, int wave_v, int nw_v) {
    const PredictArgs &a = *pa_g;
    const int quad = __builtin_amdgcn_readfirstlane(quad_v);
    const int wave = __builtin_amdgcn_readfirstlane(wave_v), nw = __builtin_amdgcn_readfirstlane(nw_v);
    __attribute__((address_space(3))) int *const gof = (__attribute__((address_space(3))) int *)go_g;
    const LdsPtr scr = (LdsPtr)scr_g;
    const LdsCPtr cs = (LdsCPtr)cs_g;
    const LdsPtr phi = scr, p = scr + 81, m = scr + 162, dev = scr + 243, snc = scr + 252;   // 81 + 81 + 81 + 9 + 3 = 255 doubles
    const int per_wg = quad ? DEV_LANES / 4 : DEV_LANES;
    // (the trajectories of the workgroup dealt over its waves: every wave is at the attempt barrier anyway, and one trajectory's update
    //  is a chain of memory round trips - sixteen of them one after the other on the integrator wave cost more than the launches they replace)
    for (int tj = wave; tj < per_wg; tj += nw) {
        const int64_t i = gid0 + tj;
        if (i >= n) break;  // (uniform)
        // every load of this trajectory that does not depend on another, at once (one round trip past the L1 instead of three one
        // behind the other: the whole workgroup waits for the slowest wave of this function); a finished trajectory's are dropped
        const int64_t dur_i = ld_l2(&a.dur[i]);
        const int64_t prev_ep = ld_l2(&a.prev_epoch[i]);
        const int32_t u = ld_l2(&a.hist.n_updates[i]);
        // 81 elements over 64 lanes: t = lane, and t = lane + 64 for the first 17
        const double phi_a = ld_l2(&o_stm[i * 81 + lane]), p_a = ld_l2(&a.covar[i * 81 + lane]);
        double phi_b = 0.0, p_b = 0.0, dev_v = 0.0;
        if (lane < 17) { phi_b = ld_l2(&o_stm[i * 81 + 64 + lane]); p_b = ld_l2(&a.covar[i * 81 + 64 + lane]); }
        if (lane < 9 && a.state_dev) dev_v = ld_l2(&a.state_dev[i * 9 + lane]);
        if (dur_i == 0) continue;  // (uniform) finished or failed earlier
        const int ln = quad ? 4 * tj : tj;  // the lane that owns trajectory tj's cold state
        const int64_t epoch = __double_as_longlong(cs[0 * DEV_LANES + ln]);
        const int st = (int)((__double_as_longlong(cs[19 * DEV_LANES + ln]) >> 32) & 0xffff);
        if (st != 0) {  // the reference returns the propagation error: no estimate for this segment, the run ends
            if (lane == 0) { a.status[i] = st; a.dur[i] = 0; }
            continue;
        }
        const int64_t delta_ns = epoch - prev_ep;
        phi[lane] = phi_a; p[lane] = p_a;
        if (lane < 17) { phi[64 + lane] = phi_b; p[64 + lane] = p_b; }
        if (lane < 9) dev[lane] = dev_v;
        int snc_q = -1;
        {
            // the process noise that applies: last applicable entry (filtering.rs:64-80), its diagonal at this epoch
            // (ProcessNoise::to_matrix, snc.rs:165-205) expressed in the state frame (ProcessNoise::propagate, snc.rs:219-239); every lane
            // computes the same values (uniform operands)
            for (int q = a.cfg.n_process_noise - 1; q >= 0; --q) {
                const nyx_hip_process_noise_t &pn = a.cfg.process_noise[q];
                if (pn.has_start_time && pn.start_time_ns > epoch) continue;  // snc.rs:168-175
                if (delta_ns > pn.disable_time_ns) continue;                  // snc.rs:178-186, 248-250
                snc_q = q;
                break;
            }
            if (snc_q >= 0 && lane == 0) {
                const nyx_hip_process_noise_t &pn = a.cfg.process_noise[snc_q];
                double d[3] = {pn.diag[0], pn.diag[1], pn.diag[2]};
                if (pn.has_decay) {
                    const int64_t init = pn.init_epoch_ns != INT64_MIN ? pn.init_epoch_ns : a.init_epoch[i];
                    const double total = ns_to_seconds(epoch - init);
                    for (int k = 0; k < 3; ++k) d[k] = d[k] * exp(-pn.decay_s[k] * total);
                }
                if (pn.local_frame != NYX_HIP_FRAME_INERTIAL) {
                    const double r[3] = {cs[8 * DEV_LANES + ln], cs[9 * DEV_LANES + ln], cs[10 * DEV_LANES + ln]};
                    const double v[3] = {cs[11 * DEV_LANES + ln], cs[12 * DEV_LANES + ln], cs[13 * DEV_LANES + ln]};
                    double h[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
                    const double hn = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
                    for (int k = 0; k < 3; ++k) h[k] = h[k] / hn;
                    double e0[3], e1[3], e2[3];
                    if (pn.local_frame == NYX_HIP_FRAME_RIC) {
                        const double rn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                        for (int k = 0; k < 3; ++k) { e0[k] = r[k] / rn; e2[k] = h[k]; }
                        e1[0] = e2[1] * e0[2] - e2[2] * e0[1]; e1[1] = e2[2] * e0[0] - e2[0] * e0[2]; e1[2] = e2[0] * e0[1] - e2[1] * e0[0];
                    } else {
                        const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                        for (int k = 0; k < 3; ++k) { e0[k] = v[k] / vn; e1[k] = h[k]; }
                        e2[0] = e0[1] * e1[2] - e0[2] * e1[1]; e2[1] = e0[2] * e1[0] - e0[0] * e1[2]; e2[2] = e0[0] * e1[1] - e0[1] * e1[0];
                    }
                    double nd[3];
                    for (int k = 0; k < 3; ++k) {  // (dcm * snc) * dcm^T, entry (k, k); dcm[k][j] = e_j[k]
                        const double c0 = e0[k], c1 = e1[k], c2 = e2[k];
                        nd[k] = ((c0 * d[0]) * c0 + (c1 * d[1]) * c1) + (c2 * d[2]) * c2;
                    }
                    for (int k = 0; k < 3; ++k) d[k] = nd[k];
                }
                for (int k = 0; k < 3; ++k) snc[k] = d[k];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (one wave: its LDS writes are ordered; the loads above have landed)
        __builtin_amdgcn_wave_barrier();
        // M = Phi * P, element (r, c) at c * 9 + r
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t = lane + 64 * half;
            if (t < 81) {
                const int r = t % 9, c = t / 9;
                double acc = phi[r] * p[c * 9];  // k = 0
                for (int k = 1; k < 9; ++k) acc = acc + phi[k * 9 + r] * p[c * 9 + k];
                m[t] = acc;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const bool keep = u < a.hist.capacity;
        const int64_t slot = (int64_t)u * a.n + i;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t = lane + 64 * half;
            if (t < 81) {
                const int r = t % 9, c = t / 9;
                double acc = m[r] * phi[c];  // (stm * covar) * stm^T: sum_k M[r,k] * Phi[c,k]
                for (int k = 1; k < 9; ++k) acc = acc + m[k * 9 + r] * phi[k * 9 + c];
                if (snc_q >= 0 && r < 6 && c < 6 && r % 3 == c % 3) {
                    const double dt = ns_to_seconds(delta_ns);
                    const double half_dt2 = (dt * dt) / 2.0;  // delta_t.powi(2) / 2.0
                    const double g_r = r < 3 ? half_dt2 : dt, g_c = c < 3 ? half_dt2 : dt;
                    acc = acc + (g_r * snc[r % 3]) * g_c;  // (Gamma * Q) * Gamma^T, single non-zero term
                }
                a.covar[i * 81 + t] = acc;
                if (keep && a.hist.covar) a.hist.covar[slot * 81 + t] = acc;
                if (keep && a.hist.stm) a.hist.stm[slot * 81 + t] = phi[t];
                o_stm[i * 81 + t] = (r == c) ? 1.0 : 0.0;  // reset_stm() (mod.rs:479)
            }
        }
        if (lane < 9) {
            double sb = 0.0;
            if (a.cfg.deviation_tracking) {
                sb = phi[lane] * dev[0];
                for (int k = 1; k < 9; ++k) sb = sb + phi[k * 9 + lane] * dev[k];
            }
            if (a.state_dev) a.state_dev[i * 9 + lane] = sb;
            if (keep && a.hist.state_dev) a.hist.state_dev[slot * 9 + lane] = sb;
            if (keep && a.hist.state) a.hist.state[slot * 9 + lane] = cs[(8 + lane) * DEV_LANES + ln];
        }
        const bool go = epoch < a.cfg.end_epoch_ns;  // mod.rs:480-482
        if (lane == 0) {
            if (keep && a.hist.epoch_ns) a.hist.epoch_ns[slot] = epoch;
            a.hist.n_updates[i] = u + 1;
            a.prev_epoch[i] = epoch;
            a.dur[i] = go ? a.cfg.max_step_ns : 0;
        }
        if (lane == 0 && go) gof[ln] = 1;   // (read by the integrator wave behind the next barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the scratch rows are rewritten by the next trajectory)
        __builtin_amdgcn_wave_barrier();
    }
    // the scratch is this wave's slot of the partial sums: a wave without columns leaves the fold an exact zero there
    for (int q = lane; q < 256; q += DEV_LANES) scr[q] = 0.0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// One role (or a merged set of roles) of the workgroup.  Every instantiation executes the SAME sequence of
// workgroup barriers; only the work between them differs, so that each role keeps just its own state live.
template <bool INTEG, bool ALMANAC, bool PERT, bool STM, bool QUAD = false, bool PIPE = false>
DEVFN void role_loop(const DevBatch &bt, CfgPtr cfg, const DevCfg *cfg_g, HarmPtr htab, ColPtr cols,
                     const double *__restrict__ records, const LdsMap &L, const int lane, const int wave, const int nw) {
    double *const kbuf = L.kbuf;
    double *const tabl = L.tabl;
    const int stages = cfg->stages;
    // Markers for tools/kernel_roles.py: every instantiation of this function is inlined into one kernel, and which ROLE owns the
    // scratch traffic of a code object cannot be told from its metadata.  `s_nop 13; s_nop <id>` opens a role's code, `s_nop 13; s_nop 15`
    // closes it (two scalar no-ops per wave and launch); id = INTEG | ALMANAC << 1 | PERT << 2 | PIPE << 3 (the STM / quad layouts live in
    // kernels of their own).
    asm volatile("s_nop 13\n\ts_nop %0" ::"n"((INTEG ? 1 : 0) | (ALMANAC ? 2 : 0) | (PERT ? 4 : 0) | (PIPE ? 8 : 0)) : "memory");
#ifdef NYX_ASSUME_SMALL
    // propagate_w8n.hip: the kernel of workgroups whose dynamics have no body-fixed model at all (point masses and SRP around a
    // two-body term: BASELINE config 3) - the four switches are compile-time constants there.  The general eight-wave kernel is 2.8 MB
    // of code with every force model behind uniform branches and misses its instruction cache three to five times per wave and
    // stage; the same source without those models runs config 3 in 51.5 ms instead of 56.2, bit for bit the same results.
    const bool has_grav = false, has_drag = false, has_tides = false, has_grav2 = false;
    const bool has_srp = cfg->has_srp != 0;
    const bool has_pm = cfg->n_pm > 0;
#else
    const bool has_grav = cfg->has_grav != 0;
    const bool has_srp = cfg->has_srp != 0;
    const bool has_pm = cfg->n_pm > 0;
    const bool has_drag = cfg->has_drag != 0;
#ifdef NYX_NO_TIDES
    const bool has_tides = false;
#else
    const bool has_tides = cfg->has_tides != 0;
#endif
    const bool has_grav2 = cfg->has_grav2 != 0;  // (plain kernel: value; STM kernels: value and gradient, in either layout)
#endif
    const bool need_almanac = has_grav || has_drag || has_tides || cfg->n_slots > 0;
    // role fan-out: this wave's share of the almanac / perturbation duties, and its status slot
    const int amask = ALMANAC ? cfg->role_mask[wave] : 0;
    const int pmask = PERT ? cfg->role_mask[wave] >> 16 : 0;
    const int n_alm = cfg->n_alm;
    int *const my_edst = L.edst + (ALMANAC ? cfg->role_slot[wave] : 0) * 2 * DEV_LANES;
    const bool rec_in_lds = cfg->rec_in_lds != 0;
    const bool dbg_skip_serial = (cfg->flags & DBG_SKIP_SERIAL) != 0;
    const bool dbg_skip_harm = (cfg->flags & DBG_SKIP_HARMONICS) != 0;
    // optional cycle accounting (workgroup 0 only): [0] phase A, [1] window duty (almanac / pert), [2] harmonics,
    // [3] phase C, [4] step control, [5] total, [6] barrier waits, [7] realtime (100 MHz)
    const bool prof_on = NYX_PROF && bt.prof != nullptr && blockIdx.x == 0;
    int64_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t prof_start = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
    const int64_t prof_rt0 = prof_on ? (int64_t)__builtin_amdgcn_s_memrealtime() : 0;

    // ---- per-lane trajectory binding (every wave maps lane -> the same trajectory).  Quad layout: 16 trajectories per
    // workgroup, the four lanes of a quad are bound to the same one (everything that is per trajectory is computed four
    // times over, identically; `wr` picks the lane that writes it out) and differ in the partial their duals carry.
    static_assert(!QUAD || STM, "the quad layout is the STM kernel's");
    const int ql = QUAD ? (lane & 3) : 0;
    constexpr int KB_STR = QUAD ? DEV_LANES / 4 : DEV_LANES;
    const int kb_li = QUAD ? (lane >> 2) : lane;
    const int64_t gid = QUAD ? (int64_t)blockIdx.x * (DEV_LANES / 4) + (lane >> 2) : (int64_t)blockIdx.x * DEV_LANES + lane;
    const bool valid = gid < bt.n;
    const bool wr = valid && ql == 0;
    const int64_t idx = valid ? gid : bt.n - 1;

    // perturbation-wave constants
    double p_cr = 0.0, p_area = 0.0, p_mass = 1.0, p_cd = 0.0, p_darea = 0.0;

    if (INTEG) {
        ColdState c;
        c.epoch = bt.epoch_ns[idx];
        c.y[0] = bt.x[idx]; c.y[1] = bt.y[idx]; c.y[2] = bt.z[idx];
        c.y[3] = bt.vx[idx]; c.y[4] = bt.vy[idx]; c.y[5] = bt.vz[idx];
        c.y[6] = bt.cr ? bt.cr[idx] : 0.0;
        c.y[7] = bt.cd ? bt.cd[idx] : 0.0;
        c.y[8] = bt.mprop ? bt.mprop[idx] : 0.0;
        const int64_t duration = bt.dur_ns ? bt.dur_ns[idx] : (bt.use_end_epoch ? (bt.end_epoch_ns - c.epoch) : bt.duration_ns);
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
        c.massless = (has_srp || has_drag) && !(mass > 0.0);  // MasslessSpacecraft (spacecraft.rs:201-203)
        cold_store(L.cs, lane, c);
        if (bt.ev_on && wr) {  // y_prev of the start state (event.rs:104-106)
            const double y0[6] = {c.y[0], c.y[1], c.y[2], c.y[3], c.y[4], c.y[5]};
            bt.ev_prev[gid] = ev_eval(*bt.ev, bt.ev_mu, c.epoch, y0);
            bt.ev_count[gid] = 0;
            bt.ev_found[gid] = 0;
        }
        if (bt.traj_cap > 0 && wr) {  // dense output: the start state is entry 0 (instance.rs:319-321)
            bt.t_epoch[gid] = c.epoch;
#pragma unroll
            for (int e = 0; e < 6; ++e) bt.t_state[e][gid] = c.y[e];
            bt.t_len[gid] = (duration == 0 || c.done) ? 1 : 1;
        }
        if (STM && valid && bt.o_stm != bt.stm) {
            for (int q = ql; q < 81; q += (QUAD ? 4 : 1)) bt.o_stm[gid * 81 + q] = bt.stm[gid * 81 + q];
        }
    }
    if (PERT) {
        // constant along the trajectory: no guidance law on this path => d(Cr, mass)/dt = 0
        p_cr = clamp02(bt.cr ? bt.cr[idx] : 0.0);
        p_area = bt.asrp ? bt.asrp[idx] : 0.0;
        p_cd = bt.cd ? bt.cd[idx] : 0.0;
        p_darea = bt.adrag ? bt.adrag[idx] : 0.0;
        p_mass = (bt.mdry ? bt.mdry[idx] : 0.0) + (bt.mprop ? bt.mprop[idx] : 0.0) + (bt.mextra ? bt.mextra[idx] : 0.0);
    }
    __syncthreads();
    // cooperative mode (see above): the integrator owns the conversation with the helper
    CoopBox *const cbox = bt.coop_box + blockIdx.x;
    const int coop_widx = bt.coop_sets > 0 ? ((int)blockIdx.x % bt.coop_sets) * COOP_SET + (int)blockIdx.x / bt.coop_sets : 0;
#define coop_two (COOP_PARTS_HERE == 2) /* (compile-time: see NYX_COOP_TWO_PARTS) */
#ifdef NYX_COOP_FAN
#define COOP_FB_PARTS ((bt.coop_parts & 0xf) << 8) /* coop_fallback: the parts of the fan-out */
#else
#define COOP_FB_PARTS (coop_two ? 0x100 : 0)
#endif
    bool coop_on = !STM && LCTL[1] != 0;
    const bool coop_started = coop_on;
    uint32_t coop_seq = 0;
    bool coop_drop = false;  // fallback taken: the workers go back to DEV_SCHED_SOLO from the next evaluation on
    // Pipelined stage loop.  What the column workers need for stage i+1 is its POSITION (the five recursion inputs), and
    // that depends on the velocities of the stages up to i, i.e. on the accelerations up to stage i-1 only.  So the
    // integrator wave, idle in window i, publishes position and inputs of stage i+1 there (second set of LDS buffers, by
    // stage parity), the workers go from barrier B2(i) straight into the harmonics of stage i+1, and the integrator's
    // phase C(i) + the velocity of stage i+1 run beside them instead of in front of them: one barrier per stage, nobody
    // waits for the serial phases.  Three LDS words order the rest: ctl[2] = last stage whose epoch data the almanac wave
    // has written, ctl[3] = number of stages whose partial sums the integrator has folded (a worker does not overwrite
    // its slot before that), ctl[4] = last stage whose velocity is published (drag is the one position-AND-velocity term
    // of the perturbation wave).  Same arithmetic in the same order as the plain loop: bit-identical results.
    // (a template parameter: the two stage loops share this function, and compiled together each carries the other's live ranges;
    //  propagate_body() picks the instantiation from cfg->pipe)
    static_assert(!PIPE || ((!STM || QUAD) && !(INTEG && (ALMANAC || PERT))), "the pipelined stage loop needs the integrator in a wave of its own");
    constexpr bool pipe = PIPE;
    // Epoch data carried between attempts (almanac wave, host-enabled when the stage count is even and LDS has room).
    // Stage 0 of the next attempt sits at t + h if this attempt is accepted and at t again if it is rejected: the first
    // is computed by the almanac wave in the LAST window (where it has no next stage to prepare; buffer 0 is free by
    // then), the second is this attempt's own stage-0 data, kept aside.  Both are keyed by their integer epoch, so the
    // prologue only has to compare epochs - whatever the step logic did - and falls back to computing.
    RotBase rot_base;  // (almanac wave with the DCM share: the orientation's base epoch, see rotation_dcm_iau_poly)
    rot_base.ep = INT64_MIN;
#pragma unroll
    for (int q = 0; q < 3; ++q) { rot_base.sn[q] = 0.0; rot_base.cs[q] = 1.0; }
    const int reuse_nf = (!STM && ALMANAC && !INTEG && need_almanac && cfg->n_alm == 1) ? cfg->ed_reuse : 0;  // (one almanac wave only: the epoch tags have one writer)
    if (reuse_nf > 0) { L.ed0_ep[lane] = INT64_MIN; L.spec_ep[lane] = INT64_MIN; }
    // Speculative stage 0.  The attempt boundary is the one place where the pipelined loop still drains: phase C of the last stage,
    // error estimate, step control and phase A of stage 0 run with fifteen waves idle (~25 k cycles of ~510 k per RK89 attempt).
    // But stage 0 of the NEXT attempt sits at y + h sum b_i k_i if this attempt is accepted - a POSITION that needs the stage
    // velocities only, i.e. is complete inside the last window like the position of any next stage - and if the attempt is
    // rejected its k_0 is the one this attempt already holds (same epoch, same state: the reference recomputes it, to the same
    // bits).  So the integrator publishes that position in the last window (the almanac wave already evaluates the epoch data of
    // t + h there), the column waves go from the last stage straight into it, step control runs beside them, and per lane the
    // result is kept (accepted) or dropped in favour of the old k_0 (rejected).  No barrier at the attempt boundary any more:
    // ctl[5] = attempts published (the almanac wave waits for the new epoch and step), ctl[3] counts folds over the whole launch.
    // Same arithmetic in the same order: bit-identical to NYX_HIP_SPEC=0.  Host-enabled (cfg->spec): plain kernel, pipelined,
    // one almanac wave, even stage count (the last window then leaves the parity-0 buffers free).  It replaces the epoch data
    // carried between attempts (no copy of the stage-0 data is needed: nothing is evaluated at a rejected attempt's start) and
    // its LDS.  The exit is seen one (wasted) window late.
    const bool spec = pipe && !STM && cfg->spec != 0;
    const bool offl = PIPE && !STM && !INTEG_OOL && cfg->offload != 0;  // (uniform) see DevCfg.offload (shapes without a gravity field: never the sixteen-wave kernels)
    const bool qoff = STM && QUAD && cfg->qpre_off != 0;   // (uniform) quad layout: the position-only pieces of phase C formed by an almanac wave (quad_pre)
    // (uniform) the integrator's chain out of line (integ_front / integ_back, see INTEG_OOL): the sixteen-wave plain kernels, pipelined loop, a
    // central gravity field.  The almanac wave with the DCM share holds its write of the next-but-one DCM for the fold counter then.
#if INTEG_OOL
    constexpr bool ool = PIPE && !STM;   // (a compile-time property of these kernels: the host pipelines a sixteen-wave workgroup only with a gravity field, build_schedule)
#if defined(NYX_COOP_FAN) && FAN_SUMS
    // (uniform) fan-out mode: the integrator's two stage sums are formed by a column wave of their own (fan_sums)
    const bool sums_on = ool && cfg->sums_wave1 != 0;
    const bool sums_me = sums_on && !INTEG && !ALMANAC && !PERT && cfg->sums_wave1 == wave + 1;
#else
    constexpr bool sums_on = false;
#endif
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LdsPtr)L.kbuf;   // (the carve starts at the k-buffer)
#define IX_STAMP(k) (int64_t)(((uint64_t)(uint32_t)LCTL[9 + 2 * (k)] << 32) | (uint64_t)(uint32_t)LCTL[8 + 2 * (k)])
#else
    constexpr bool ool = false;
    constexpr bool sums_on = false;
#endif
    bool spec_now = false;  // stage 0 of the attempt being started was published in the previous attempt's last window
    bool keep_k0 = false;   // (integrator, per lane) the previous attempt was rejected: k_0 stands
    int att = 0;            // attempts started by this workgroup
    double nx_pos[3] = {0.0, 0.0, 0.0}, nx_s = 0.0, nx_t = 0.0, nx_u = 0.0, nx_kfac = 0.0;
    double m_cur[9], m_nx[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) { m_cur[q] = 0.0; m_nx[q] = 0.0; }
    uint32_t seq_cur = 0, seq_nx = 0;   // mailbox sequence numbers of this stage / the next one
    unsigned long long dbg_answers = 0, dbg_fallbacks = 0, dbg_fb_seq = 0;  // (NYX_HIP_PROFILE: row 16 of the profile)
    int64_t pl_tc = 0, pl_chain = 0, pl_wait = 0, pl_n = 0, pl_post = 0;
#if NYX_SEG_PROF
    int64_t sg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sg_t = 0;  // (NYX_HIP_PROFILE, rows 34-35: the integrator's stage in eleven pieces, step control in five: 12 cold state, 13 the two sums, 14 error estimate and decision, 15 next attempt opened, 11 the rest)
#define SEG(k) if (INTEG && prof_on) { const int64_t n_ = (int64_t)__builtin_readcyclecounter(); sg[k] += n_ - sg_t; sg_t = n_; }
    // (STM kernels, row 18 of the profile: the attempt / segment boundary of the integrator wave in pieces - 0 step control, 1 the next attempt opened,
    //  2 barrier B0, 3 the time updates of a segment boundary, 4 re-arming, 5 stage-0 epoch data + Bp, 6 phase A of stage 0 + B1)
    int64_t sb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb_t = 0;
#define SBD(k) if (INTEG && STM && prof_on) { const int64_t n_ = (int64_t)__builtin_readcyclecounter(); if (sb_t) sb[k] += n_ - sb_t; sb_t = n_; }
#else
#define SEG(k)
#define SBD(k)
#endif
    // (NYX_HIP_PROFILE, row 33: the latency loop of a cooperative owner - answer in hand -> next post)
    bool shared_cur = false, shared_nx = false;  // did the workers of this / the next stage leave columns to a helper?

    auto begin_attempt = [&](ColdState &c) __attribute__((always_inline)) { begin_attempt_fn(L, lane, c); };
    double h_next = 0.0;  // (chained attempts: the step of the attempt step control has just opened)

    for (;;) {  // one iteration = one RK attempt for every live lane (derive(), instance.rs:368-414)
        double h = h_next;
        if (INTEG && !spec_now) {
            ColdState c;
            cold_load(L.cs, lane, c);
            begin_attempt(c);
            if (STM && bt.pred != nullptr && !__any(!c.done)) {
                // every trajectory of the workgroup has finished its segment of the covariance-mapping loop (the final fixed step, or an
                // adaptive step that landed on the segment's end - which begin_attempt has just noticed): not the exit yet, a segment
                // boundary (below, behind the barrier: every wave takes a share of the time updates)
                if (lane == 0) { L.ctl[0] = 0; L.ctl[7] = 1; }
                L.pertst[lane] = 0;   // (the go flags of this boundary; the row is the second field's status of a stage otherwise, rewritten every stage)
            }
            cold_store(L.cs, lane, c);
            h = c.h;
            if (STM) {
#pragma unroll
                for (int q = 0; q < (QUAD ? 6 : 12); ++q) L.sacc[q * DEV_LANES + lane] = 0.0;
            }
        }
        SBD(1)
        if (!spec_now) {
        __syncthreads();  // B0: attempt published (or exit requested)
        SBD(2)
        if (STM && bt.pred != nullptr && LCTL[7] != 0) {  // (uniform) a segment boundary of the covariance-mapping loop, see segment_update
            segment_update(bt.pred, bt.o_stm, L.cs, L.part + wave * (QUAD ? QSLOT : (STM ? 16 * DEV_LANES : 4 * DEV_LANES)), L.pertst, lane, QUAD ? 1 : 0,
                           (int64_t)blockIdx.x * (QUAD ? DEV_LANES / 4 : DEV_LANES), bt.n, wave, nw);
            __syncthreads();
            SBD(3)
            if (INTEG) {
                // the next segment for the trajectories that have not reached the end epoch: the state, the step size and the counters carry
                // over as the reference's propagator instance carries them (od/process/mod.rs:466-483)
                ColdState c;
                cold_load(L.cs, lane, c);
                if (valid && L.pertst[lane & (QUAD ? ~3 : ~0)] != 0) {
                    c.stop = c.epoch + bt.pred->cfg.max_step_ns;
                    c.done = false;
                    c.fresh = true;
                    c.is_final = false;
                }
                if (lane == 0) L.ctl[7] = 0;
                begin_attempt(c);   // (raises the exit flag when no trajectory goes on)
                cold_store(L.cs, lane, c);
                h = c.h;
            }
            __syncthreads();
            SBD(4)
        }
        if (LCTL[0]) break;
        }

        // prologue: epoch data of stage 0
        if (!spec_now && ALMANAC && need_almanac) {
            const int64_t ep = __double_as_longlong(L.step[lane]);
            bool compute = true;
            if (reuse_nf > 0) {
                const bool hit_spec = ep == (int64_t)L.spec_ep[lane];  // accepted: buffer 0 already holds this epoch
                const bool hit_prev = ep == (int64_t)L.ed0_ep[lane];   // rejected (or finished): same epoch as last time
                compute = __any(!(hit_spec || hit_prev)) != 0;
                if (!compute && !hit_spec) {
                    for (int f = 0; f < reuse_nf; ++f) L.ed[f * DEV_LANES + lane] = L.ed0[f * DEV_LANES + lane];
                    my_edst[lane] = L.ed0st[lane];
                }
            }
            if (compute) {
                int st = rec_in_lds ? epoch_data(cfg, (const double *)L.rec, ep, L.ed, lane, amask, nullptr, 0, &rot_base)
                                    : epoch_data(cfg, records, ep, L.ed, lane, amask, nullptr, 0, &rot_base);
                my_edst[lane] = st;
            }
            if (reuse_nf > 0) {
                for (int f = 0; f < reuse_nf; ++f) L.ed0[f * DEV_LANES + lane] = L.ed[f * DEV_LANES + lane];
                L.ed0st[lane] = my_edst[lane];
                L.ed0_ep[lane] = ep;
            }
        }
        if (!spec_now) {
        if (INTEG && lane == 0) { L.ctl[2] = 0; L.ctl[3] = 0; L.ctl[4] = 0; L.ctl[6] = 0; }
        __syncthreads();  // Bp
        SBD(5)
        }
        const int fold_base = spec ? att * stages : 0;  // ctl[3] counts the folds of the whole launch when the attempts are chained
        bool leave = false;

        int st_att = NYX_HIP_OK;
        double wpre[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double pre_wr[3] = {0.0, 0.0, 0.0};  // position part of the stage sum the current window publishes from (pipelined plain loop; stage 0: empty)
        for (int i = 0; i < stages; ++i) {
            double *const edc = L.ed + (i & 1) * ED_FIELDS * DEV_LANES;
            double s_ = 0.0, t_ = 0.0, u_ = 0.0, kfac = 0.0;
            double ys[6];
            PROF_T0();
            SEG(0)   /* loop back-edge */
            if (INTEG) {
                // ---- Phase A: stage state  y + h * sum_j a_ij k_j   (instance.rs:376-394)
                double *const ysb = (pipe && (i & 1)) ? L.ys2 : L.ys;
                double *const inbb = (pipe && (i & 1)) ? L.inb2 : L.inb;
                if (ool && pipe && (i > 0 || spec_now)) {
                    // (phase A of this stage is the head of integ_front, called from the window below)
                    seq_cur = seq_nx; shared_cur = shared_nx;
                } else
                if (!ool && pipe && (i > 0 || spec_now)) {
                    // position and inputs of this stage were published in the previous window: only the velocity is left
                    if (i == 0) {
                        // speculative stage 0: the state step control has just stored (accepted lanes: its position IS the published
                        // one, bit for bit; rejected lanes: the result of this stage is dropped, k_0 stands)
#pragma unroll
                        for (int e = 0; e < 6; ++e) ys[e] = CS_Y(e);
#pragma unroll
                        for (int e = 3; e < 6; ++e) ysb[e * DEV_LANES + lane] = ys[e];
                    } else {
                    const double a_last = A_ROW(i, i - 1);
                    if (!STM && NX_IN_LDS) {  // (the position the previous window published: read back, not carried - see NX_IN_LDS)
#pragma unroll
                        for (int e = 0; e < 3; ++e) ys[e] = ysb[e * DEV_LANES + lane];
                    } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e) ys[e] = nx_pos[e];
                    }
#pragma unroll
                    for (int e = 3; e < 6; ++e) {
                        const double wi = wpre[e] + a_last * KB(i - 1, e);
                        ys[e] = CS_Y(e) + h * wi;
                        ysb[e * DEV_LANES + lane] = ys[e];
                    }
                    }
                    if (has_drag) {  // the perturbation wave is already in this stage's window; drag is the one term that wants the velocity
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) LCTL[4] = i + 1;
                    }
                    // (the almanac wave finished this stage's data before the barrier this wave has just passed)
                    if (need_almanac && !(i == 0 && keep_k0)) {  // (a rejected lane's stage 0 is not evaluated: its epoch data at t + h does not count)
                        for (int a = 0; a < n_alm; ++a) {
                            const int es = L.edst[(2 * a + (i & 1)) * DEV_LANES + lane];
                            if (es) st_att = es;
                        }
                    }
                    if (!STM && NX_IN_LDS) {
                        if (has_grav) {
                            // s, t, u, (mu / r) / R_eq of this stage from the rows the publishing window left them in (wave 0's slot of the partial
                            // sums: the integrator of a pipelined workgroup carries no columns), its DCM from the epoch data (this parity's
                            // buffer is rewritten in the NEXT window, behind the barrier this stage ends with; phase C needs it after that
                            // barrier: registers from here on)
                            s_ = L.part[0 * DEV_LANES + lane]; t_ = L.part[1 * DEV_LANES + lane]; u_ = L.part[2 * DEV_LANES + lane]; kfac = L.part[3 * DEV_LANES + lane];
#pragma unroll
                            for (int q = 0; q < 9; ++q) m_cur[q] = edc[q * DEV_LANES + lane];
                        }
                    } else {
                    s_ = nx_s; t_ = nx_t; u_ = nx_u; kfac = nx_kfac;
#pragma unroll
                    for (int q = 0; q < 9; ++q) m_cur[q] = m_nx[q];
                    }
                    seq_cur = seq_nx; shared_cur = shared_nx;
                } else {
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
                for (int e = 0; e < 6; ++e) ysb[e * DEV_LANES + lane] = ys[e];
                if (need_almanac) {
                    for (int a = 0; a < n_alm; ++a) {
                        const int es = L.edst[(2 * a + (i & 1)) * DEV_LANES + lane];
                        if (es) st_att = es;
                    }
                }
                if (has_grav) {
                    // body-fixed position and the scaled inputs of the column recursion
#pragma unroll
                    for (int q = 0; q < 9; ++q) m_cur[q] = edc[q * DEV_LANES + lane];
                    // the field of another body than the integration centre (gravity_field.rs:150-154: transform_to translates to the
                    // field's body before it rotates): evaluated at r - r_body(t).  Plain stage loop only (the host clears cfg->pipe)
                    double rg[3] = {ys[0], ys[1], ys[2]};
                    if (cfg->g_slot >= 0) {  // (uniform; the translation carries no partials: d(r - r_body(t)) / dr = 1)
                        double pg[3];
                        ed_body(cfg, edc, lane, cfg->g_slot, pg);
                        rg[0] = ys[0] - pg[0]; rg[1] = ys[1] - pg[1]; rg[2] = ys[2] - pg[2];
                    }
                    const double rb0 = m_cur[0] * rg[0] + m_cur[1] * rg[1] + m_cur[2] * rg[2];
                    const double rb1 = m_cur[3] * rg[0] + m_cur[4] * rg[1] + m_cur[5] * rg[2];
                    const double rb2 = m_cur[6] * rg[0] + m_cur[7] * rg[1] + m_cur[8] * rg[2];
                    // one sqrt and one divide on the critical path; the rest are multiplies
                    const double r_ = norm3(rb0, rb1, rb2);
                    const double inv_r = 1.0 / r_;
                    s_ = rb0 * inv_r; t_ = rb1 * inv_r; u_ = rb2 * inv_r;
                    const double rho = cfg->g_re * inv_r;
                    kfac = (cfg->g_mu * inv_r) * cfg->g_inv_re;  // (mu / r) / R_eq
                    inbb[0 * DEV_LANES + lane] = rho * s_;
                    inbb[1 * DEV_LANES + lane] = rho * t_;
                    inbb[2 * DEV_LANES + lane] = rho * u_;
                    inbb[3 * DEV_LANES + lane] = rho;
                    inbb[4 * DEV_LANES + lane] = r_ * cfg->g_inv_re;
                    if (STM && QUAD) {
                        publish_d1_inputs(cfg, rb0, rb1, rb2, ql, inbb, lane);  // (inb aliases the head of inbD: written after the plain rows)
                    } else if (STM) {
                        // the same quantities as duals seeded in the BODY-FIXED frame (gravity_field.rs:285-291)
                        const D3 x0 = {rb0, 1.0, 0.0, 0.0}, x1 = {rb1, 0.0, 1.0, 0.0}, x2 = {rb2, 0.0, 0.0, 1.0};
                        const D3 rD = d3norm(x0, x1, x2);
                        const D3 sD = d3div(x0, rD), tD = d3div(x1, rD), uD = d3div(x2, rD);
                        const D3 rhoD = d3div(d3c(cfg->g_re), rD);
                        const D3 kD = d3div(d3div(d3c(cfg->g_mu), rD), d3c(cfg->g_re));
                        const D3 invD = rD * cfg->g_inv_re;
                        (void)kD;
                        const D3 pub[5] = {rhoD * sD, rhoD * tD, rhoD * uD, rhoD, invD};
#pragma unroll
                        for (int q = 0; q < 5; ++q) {
                            L.inbD[(4 * q + 0) * DEV_LANES + lane] = pub[q].v; L.inbD[(4 * q + 1) * DEV_LANES + lane] = pub[q].x;
                            L.inbD[(4 * q + 2) * DEV_LANES + lane] = pub[q].y; L.inbD[(4 * q + 3) * DEV_LANES + lane] = pub[q].z;
                        }
                    }
                }
                if (pipe) {  // stage 0 of an attempt: the workers' schedule for it is decided here, before B1
                    shared_cur = coop_on;
                    if (lane == 0) L.ctl[1] = coop_on ? 1 : 0;
                }
                if (ool) {  // (integ_back reads s, t, u, (mu / r) / R_eq of a stage from its parity's rows: stage 0 here)
                    L.part[0 * DEV_LANES + lane] = s_; L.part[1 * DEV_LANES + lane] = t_; L.part[2 * DEV_LANES + lane] = u_; L.part[3 * DEV_LANES + lane] = kfac;
                }
                }
            }
            PROF_ADD(0);
            SEG(1)   /* phase A */
            if (!pipe || (i == 0 && !spec_now)) {
                PROF_T0();
                __syncthreads();  // B1: stage state and harmonics inputs published (pipelined: stage 0 only, and not when it was published speculatively)
                PROF_ADD(6);
            }
#if NYX_SEG_PROF
            if (i == 0) { SBD(6) }
            if (i == 1) sb_t = 0;  /* (the pieces are measured from the end of the stage loop to B1 of stage 0) */
#endif
            const int64_t ptw_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;

            // ---- window --------------------------------------------------------------------------
            if (INTEG && !STM && coop_on && has_grav && (!pipe || (i == 0 && !spec_now))) {
                seq_cur = ++coop_seq;
                if (coop_two) coop_post2(cbox, bt.coop_posted + coop_widx, lane, seq_cur, (LdsCPtr)L.inb); else coop_post(cbox, bt.coop_posted + coop_widx, lane, seq_cur, (LdsCPtr)L.inb);
            }
            const bool last_stage = i + 1 == stages;
            if (ALMANAC && need_almanac && (!last_stage || reuse_nf > 0 || spec)) {
                // epoch-only data of the NEXT stage; in the last window (carried epoch data, even stage count: parity 0 again)
                // that is stage 0 of the next attempt should this one be accepted: epoch + seconds_to_ns(h), instance.rs:401
                if (spec_now && i == 0) {  // epoch and step of this attempt: step control ran beside the start of this window
                    int spin = 0;
                    while (LCTL[5] != att && ++spin < 4000000) __builtin_amdgcn_s_sleep(4);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                const double c_next = last_stage ? 1.0 : C_COEF(i + 1);
                const int64_t ep = __double_as_longlong(L.step[lane]) + seconds_to_ns(c_next * L.step[DEV_LANES + lane]);
                double *edn = L.ed + ((i + 1) & 1) * ED_FIELDS * DEV_LANES;
                if (last_stage && reuse_nf > 0) L.spec_ep[lane] = ep;
                int st = NYX_HIP_OK;
                if (!dbg_skip_serial || i == 0)  // (timing switch: reuse the data of stages 0/1)
                {
                    const LdsFlagPtr fl = (pipe && (!last_stage || spec) && (amask & DEV_ROLE_DCM)) ? LCTL + 2 : nullptr;
                    // (INTEG_OOL: the DCM rows of `edn` are those of stage i - 1 until its phase C has read them - fold counter >= fold_base + i)
                    const LdsFlagPtr gt = ool ? LCTL + 3 : nullptr;
                    st = rec_in_lds ? epoch_data(cfg, (const double *)L.rec, ep, edn, lane, amask, fl, i + 1, &rot_base, gt, fold_base + i)
                                    : epoch_data(cfg, records, ep, edn, lane, amask, fl, i + 1, &rot_base, gt, fold_base + i);
                }
                my_edst[((i + 1) & 1) * DEV_LANES + lane] = st;
                if (pipe && (!last_stage || spec) && (amask & DEV_ROLE_DCM)) {  // tell the integrator wave (which publishes the inputs of stage i+1 inside this window)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) LCTL[2] = i + 1;
                }
            }
            if (PIPE && !STM && ALMANAC && offl) {
                // work taken off the integrator wave (the critical path of a pipelined workgroup without column waves), see DevCfg.offload.
                // L.part is free without a gravity field: rows 0-11 = two parities of the six partial stage sums, 12-17 = of the two-body term
                if (amask & DEV_ROLE_TWOBODY) {
                    const double *const ysp = (i & 1) ? L.ys2 : L.ys;
                    const double r0 = ysp[0 * DEV_LANES + lane], r1 = ysp[1 * DEV_LANES + lane], r2 = ysp[2 * DEV_LANES + lane];
                    const double rmag = norm3(r0, r1, r2);
                    const double f = -cfg->mu_central / cube(rmag);
                    double *const tb = L.part + (12 + 3 * (i & 1)) * DEV_LANES;
                    tb[0 * DEV_LANES + lane] = f * r0; tb[1 * DEV_LANES + lane] = f * r1; tb[2 * DEV_LANES + lane] = f * r2;
                }
                if ((amask & DEV_ROLE_SUMS) && i >= 2 && i + 2 < stages) {
                    // stage T = i + 2: sum_{j <= i-2} a_Tj k_j (k_{i-2} was written before the barrier this window started from)
                    double q6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
                    for (int j = 0; j <= i - 2; ++j) {
                        const double a_nj = A_ROW(i + 2, j);
#pragma unroll
                        for (int e = 0; e < 6; ++e) q6[e] += a_nj * KB(j, e);
                    }
                    double *const qb = L.part + (i & 1) * 6 * DEV_LANES;
#pragma unroll
                    for (int e = 0; e < 6; ++e) qb[e * DEV_LANES + lane] = q6[e];
                }
            }
            if (ALMANAC && STM && QUAD && qoff && (amask & DEV_ROLE_QPRE)) {
                // (after this wave's epoch data of the NEXT stage - the integrator's window is waiting for that DCM.  The rows are read
                //  by phase C of THIS stage, behind B2; phase C of the previous stage may still be reading the previous contents: ctl[6]
                //  = stages whose phase C is done, bounded spin)
                bool expired = false;
                if (i > 0) {
                    int spin = 0;
                    while (LCTL[6] < i && ++spin < 4000000) __builtin_amdgcn_s_sleep(2);
                    expired = spin >= 4000000;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                // (a protocol error ends as a FAILED run, like the other bounded spins: the rows are left alone - phase C of the previous stage
                //  may still be reading them - and every lane's status row of the next stage carries the error, which the integrator's phase A
                //  turns into the attempt's status)
                if (expired) my_edst[((i + 1) & 1) * DEV_LANES + lane] = NYX_HIP_ERR_NAN;
                const double *const ysq = (pipe && (i & 1)) ? L.ys2 : L.ys;
                if (!expired) quad_pre(cfg, edc, ysq[0 * DEV_LANES + lane], ysq[1 * DEV_LANES + lane], ysq[2 * DEV_LANES + lane], ql, lane, L.qpre, has_grav);
            }
            if (PERT && (has_pm || has_srp || has_drag || has_tides || has_grav2)) {
                // position-dependent third-body and SRP terms of THIS stage
                double *const ysp = (pipe && (i & 1)) ? L.ys2 : L.ys;
                double *const pertp = (pipe && (i & 1)) ? L.pert2 : L.pert;
                double r[3] = {ysp[0 * DEV_LANES + lane], ysp[1 * DEV_LANES + lane], ysp[2 * DEV_LANES + lane]};
                // role fan-out: share DEV_PERT_PM = point masses (+ tides), share DEV_PERT_SRP = SRP (+ drag); every share
                // writes only its own rows.  (STM: the rows of the plain path alias the dual ones and are not written.)
                const bool do_pm = (pmask & DEV_PERT_PM) != 0, do_srp = (pmask & DEV_PERT_SRP) != 0;
                if (!STM) {
                    double a3[3] = {0.0, 0.0, 0.0}, f3[3] = {0.0, 0.0, 0.0};
                    if (has_pm && do_pm && !dbg_skip_serial) point_masses_accel(cfg, edc, lane, r, a3);
                    if (has_srp && do_srp && !dbg_skip_serial) {
                        srp_force(cfg, edc, lane, r, p_cr, p_area, f3);
                        f3[0] = f3[0] / p_mass; f3[1] = f3[1] / p_mass; f3[2] = f3[2] / p_mass;
                    }
                    if (do_pm) {
#pragma unroll
                        for (int e = 0; e < 3; ++e) pertp[e * DEV_LANES + lane] = a3[e];
                    }
                    if (do_srp) {
#pragma unroll
                        for (int e = 0; e < 3; ++e) pertp[(3 + e) * DEV_LANES + lane] = f3[e];
                    }
                }
                if (has_drag && do_srp) {
                    if (pipe && (i > 0 || spec_now)) {  // velocity of this stage: written by phase A, which runs beside this window
                        int spin = 0;
                        while (LCTL[4] != i + 1 && ++spin < 4000000) __builtin_amdgcn_s_sleep(4);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    const double vv[3] = {ysp[3 * DEV_LANES + lane], ysp[4 * DEV_LANES + lane], ysp[5 * DEV_LANES + lane]};
                    const int64_t ep = __double_as_longlong(L.step[lane]) + seconds_to_ns(C_COEF(i) * L.step[DEV_LANES + lane]);
                    double d3f[3];
                    drag_force(cfg, rec_in_lds ? (const double *)L.rec : records, edc, lane, ns_to_seconds(ep), r, vv, p_cd, p_darea, d3f);
#pragma unroll
                    for (int e = 0; e < 3; ++e) pertp[(6 + e) * DEV_LANES + lane] = d3f[e] / p_mass;
                }
                if (STM && QUAD) pert_gradients_q(cfg, edc, lane, ql, r, p_cr, p_area, p_mass, has_pm && do_pm, has_srp && do_srp, has_tides && do_pm, pmask, pertp);
                else if (STM) pert_gradients(cfg, edc, lane, r, p_cr, p_area, p_mass, has_pm, has_srp, has_tides, L.pertD);
                // third accel model (dynamics/sequence/config.rs:116-118): added to the point-mass slot, last, so that no
                // live value of this role crosses the call
                if (has_tides && !STM && do_pm) tides_into_pert(cfg, edc, lane, ysp, pertp);
                if (has_grav2 && (do_pm || STM)) {  // a second gravity field (after the tides: the reference's model order does not reach the bits the parity bar looks at)
                    const int64_t ep2 = __double_as_longlong(L.step[lane]) + seconds_to_ns(C_COEF(i) * L.step[DEV_LANES + lane]);
                    if (STM && QUAD) {
                        if (do_pm)   // (role fan-out: the wave with the point-mass share)
                            L.pertst[(i & 1) * DEV_LANES + lane] =
                                second_field_into_pert_q(cfg, rec_in_lds ? (const double *)L.rec : records, edc, lane, ql, wave, ns_to_seconds(ep2), ysp, pertp);
                    } else if (STM)
                        L.pertst[(i & 1) * DEV_LANES + lane] =
                            second_field_into_pertD(cfg, rec_in_lds ? (const double *)L.rec : records, edc, lane, wave, ns_to_seconds(ep2), ysp, L.pertD);
                    else
                        L.pertst[(i & 1) * DEV_LANES + lane] =
                            second_field_into_pert(cfg, rec_in_lds ? (const double *)L.rec : records, edc, lane, wave, ns_to_seconds(ep2), ysp, pertp);
                }
            }
#if defined(NYX_COOP_FAN) && FAN_SUMS
            if (sums_me) fan_sums(lds_base, (uint64_t)cfg, i, lane, (i > 0 || spec_now) ? 1 : 0, fold_base + i);
#endif
            double acc[3] = {0.0, 0.0, 0.0};
            // The integrator's window.  Round 5: in the pipelined plain loop the position and the recursion inputs of the next stage are
            // formed and POSTED first, everything else (two-body term, the velocity part of the next stage sum, the position part of the
            // one after) behind them.  The helper's answer of stage i - 1 and the post of stage i + 1 are the two ends of the loop
            // that bounds a cooperative workgroup's period, (chain + helper latency) / 2: the chain was phase C, phase A, two-body,
            // the whole 6 x i stage sum out of LDS, THEN the position.  The position part of the stage sum needs the stage VELOCITIES
            // only - known one window earlier - and is carried in registers (pre_wr); same terms, same order, same bits.
            const bool fastp = PIPE && !STM && !offl;
            auto publish_next = [&](const bool from_pre) __attribute__((always_inline)) {
                if (pipe && (i + 1 < stages || spec)) {
                    // ---- position and recursion inputs of stage i+1, published inside the window of stage i.
                    // k_i[0..2] is this stage's velocity, so  y + h (wpre + a_{i+1,i} k_i)  is complete for the position
                    if (i + 1 < stages) {
                    const double a_nl = A_ROW(i + 1, i);
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const double wi = (from_pre ? pre_wr[e] : wpre[e]) + a_nl * ys[3 + e];
                        nx_pos[e] = CS_Y(e) + h * wi;
                    }
                    } else {
                        // last window: stage 0 of the next attempt, should this one be accepted - the position step control will form
                        // (same operations in the same order: next[e] = y[e]; next[e] += (h b_j) k_j[e], j ascending)
                        if (from_pre) {  // (y + the terms j < i: added up in the previous window)
#pragma unroll
                            for (int e = 0; e < 3; ++e) nx_pos[e] = pre_wr[e];
                        } else {
#pragma unroll
                        for (int e = 0; e < 3; ++e) nx_pos[e] = CS_Y(e);
                        for (int j = 0; j < i; ++j) {
                            const double cb = h * B_COEF(j);
#pragma unroll
                            for (int e = 0; e < 3; ++e) nx_pos[e] += cb * KB(j, e);
                        }
                        }
                        {
                            const double cb = h * B_COEF(i);
#pragma unroll
                            for (int e = 0; e < 3; ++e) nx_pos[e] += cb * ys[3 + e];
                        }
                    }
                    double *const ysn = ((i + 1) & 1) ? L.ys2 : L.ys;
                    double *const inbn = ((i + 1) & 1) ? L.inb2 : L.inb;
#pragma unroll
                    for (int e = 0; e < 3; ++e) ysn[e * DEV_LANES + lane] = nx_pos[e];
                    SEG(2)   /* window: next position formed and stored */
                    if (has_grav) {  // (without a gravity field the position is all the next window needs: the perturbation waves read it after B2)
                    if (need_almanac) {  // the almanac wave writes the DCM of stage i+1 first thing in this window
                        // (bounded: a protocol error must end as a failed run, never as a hung GPU)
                        int spin = 0;
                        while (LCTL[2] != i + 1 && ++spin < 4000000) __builtin_amdgcn_s_sleep(4);
                        if (spin >= 4000000) st_att = NYX_HIP_ERR_NAN;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    const double *const edn = L.ed + ((i + 1) & 1) * ED_FIELDS * DEV_LANES;  // (its DCM: the flag is raised before the body positions are evaluated)
#pragma unroll
                    for (int q = 0; q < 9; ++q) m_nx[q] = edn[q * DEV_LANES + lane];
                    SEG(3)   /* DCM flag wait */
                    const double rb0 = m_nx[0] * nx_pos[0] + m_nx[1] * nx_pos[1] + m_nx[2] * nx_pos[2];
                    const double rb1 = m_nx[3] * nx_pos[0] + m_nx[4] * nx_pos[1] + m_nx[5] * nx_pos[2];
                    const double rb2 = m_nx[6] * nx_pos[0] + m_nx[7] * nx_pos[1] + m_nx[8] * nx_pos[2];
                    const double r_ = norm3(rb0, rb1, rb2);
                    const double inv_r = 1.0 / r_;
                    nx_s = rb0 * inv_r; nx_t = rb1 * inv_r; nx_u = rb2 * inv_r;
                    const double rho = cfg->g_re * inv_r;
                    nx_kfac = (cfg->g_mu * inv_r) * cfg->g_inv_re;
                    inbn[0 * DEV_LANES + lane] = rho * nx_s;
                    inbn[1 * DEV_LANES + lane] = rho * nx_t;
                    inbn[2 * DEV_LANES + lane] = rho * nx_u;
                    inbn[3 * DEV_LANES + lane] = rho;
                    inbn[4 * DEV_LANES + lane] = r_ * cfg->g_inv_re;
                    if (STM && QUAD) publish_d1_inputs(cfg, rb0, rb1, rb2, ql, inbn, lane);
                    if (!STM && NX_IN_LDS) {
                        L.part[0 * DEV_LANES + lane] = nx_s; L.part[1 * DEV_LANES + lane] = nx_t; L.part[2 * DEV_LANES + lane] = nx_u; L.part[3 * DEV_LANES + lane] = nx_kfac;
                    }
                    }
                    SEG(4)   /* rotate, norm, inputs to LDS */
                    shared_nx = coop_on;
                    if (lane == 0) L.ctl[1] = coop_on ? 1 : 0;  // the workers read it after B2(i), for stage i+1
                    if (coop_on) {
                        seq_nx = ++coop_seq;
                        const int64_t p0_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
                        if (coop_two) coop_post2(cbox, bt.coop_posted + coop_widx, lane, seq_nx, (LdsCPtr)inbn); else coop_post(cbox, bt.coop_posted + coop_widx, lane, seq_nx, (LdsCPtr)inbn);
                        if (prof_on && pl_tc != 0) { const int64_t now_ = (int64_t)__builtin_readcyclecounter(); pl_chain += p0_ - pl_tc; pl_post += now_ - p0_; ++pl_n; pl_tc = 0; }
                    }
                }
            };
#if INTEG_OOL
            if (INTEG && fastp && ool) {
                const bool hot = i > 0 || spec_now;
                const bool pub = i + 1 < stages || spec;
                if (sums_on && i > 0) {  // the two sums the sums wave formed in the previous window (behind the stage barrier: complete)
#pragma unroll
                    for (int e = 0; e < 3; ++e) { wpre[3 + e] = L.sums[e * DEV_LANES + lane]; pre_wr[e] = L.sums[(3 + e) * DEV_LANES + lane]; }
                }
                uint32_t sq = 0;
                if (pub && coop_on) sq = ++coop_seq;
                const int64_t pf0_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;   // (accounting twin: slot 0 = integ_front, slot 1 = the whole window)
                const int st1 = integ_front(lds_base, (uint64_t)cfg, i, (hot ? IX_HOT : 0) | (spec_now ? IX_SPEC_NOW : 0) | (coop_on ? IX_COOP : 0) | (prof_on ? IX_PROF : 0),
                                            lane, h, hot ? wpre[3] : ys[3], hot ? wpre[4] : ys[4], hot ? wpre[5] : ys[5], pre_wr[0], pre_wr[1], pre_wr[2],
                                            (uint64_t)cbox, (uint64_t)(bt.coop_posted + coop_widx), sq, keep_k0 ? 1 : 0);
                if (prof_on) prof_acc[0] += (int64_t)__builtin_readcyclecounter() - pf0_;
                SEG(5)   /* integ_front: phase A, next position, DCM wait, rotate, inputs, post */
                if (st1) st_att = st1;
                if (pub) {
                    shared_nx = coop_on;
                    if (coop_on) {
                        seq_nx = sq;
                        if (prof_on && pl_tc != 0) { const int64_t p0_ = IX_STAMP(1), now_ = IX_STAMP(2); pl_chain += p0_ - pl_tc; pl_post += now_ - p0_; ++pl_n; pl_tc = 0; }
                    }
                }
                // the stage state, back from the rows phase A completed (position: published a window ago; a speculative stage 0 starts from
                // the state step control stored, as the inline code does - rejected lanes drop this stage anyway)
                {
                    const double *const ysb = (i & 1) ? L.ys2 : L.ys;
#pragma unroll
                    for (int e = 0; e < 6; ++e) ys[e] = ysb[e * DEV_LANES + lane];
                    if (hot && i == 0) {
#pragma unroll
                        for (int e = 0; e < 3; ++e) ys[e] = CS_Y(e);
                    }
                }
                // two-body term of this stage (orbital.rs:86-92)
                {
                    const double rmag = norm3(ys[0], ys[1], ys[2]);
                    const double f = -cfg->mu_central / cube(rmag);
                    acc[0] = f * ys[0]; acc[1] = f * ys[1]; acc[2] = f * ys[2];
                }
                // the two stage sums (integ_sums): velocity part of the next stage's, position part of the one the NEXT window publishes from
#if IX_SUMS_OOL
                {
                    const IxSums sm = integ_sums(lds_base, (uint64_t)cfg, i, lane, h, ys[3], ys[4], ys[5]);
                    wpre[0] = wpre[1] = wpre[2] = 0.0;
                    wpre[3] = sm.w3; wpre[4] = sm.w4; wpre[5] = sm.w5;
                    if (i + 2 < stages || (i + 2 == stages && spec)) { pre_wr[0] = sm.p0; pre_wr[1] = sm.p1; pre_wr[2] = sm.p2; }
                }
#else
                // (A/B switch: the sums inline, as the first cut of the out-of-line integrator had them)
                if (!sums_on) {
#pragma unroll
                for (int e = 0; e < 6; ++e) wpre[e] = 0.0;
                if (i + 1 < stages) {
#pragma unroll
                    for (int j = 0; j < DEV_MAX_STAGES - 2; ++j) {
                        if (j < i) {  // uniform
                            const double a_nj = A_ROW(i + 1, j);
#pragma unroll
                            for (int e = 3; e < 6; ++e) wpre[e] += a_nj * KB(j, e);
                        }
                    }
                }
                if (i + 2 < stages) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) pre_wr[e] = 0.0;
#pragma unroll
                    for (int j = 0; j < DEV_MAX_STAGES - 2; ++j) {
                        if (j < i) {  // uniform
                            const double a_nj = A_ROW(i + 2, j);
#pragma unroll
                            for (int e = 0; e < 3; ++e) pre_wr[e] += a_nj * KB(j, e);
                        }
                    }
                    const double a_ni = A_ROW(i + 2, i);
#pragma unroll
                    for (int e = 0; e < 3; ++e) pre_wr[e] += a_ni * ys[3 + e];
                } else if (i + 2 == stages && spec) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) pre_wr[e] = CS_Y(e);
                    for (int j = 0; j < i; ++j) {
                        const double cb = h * B_COEF(j);
#pragma unroll
                        for (int e = 0; e < 3; ++e) pre_wr[e] += cb * KB(j, e);
                    }
                    const double cbi = h * B_COEF(i);
#pragma unroll
                    for (int e = 0; e < 3; ++e) pre_wr[e] += cbi * ys[3 + e];
                }
                }
#endif
            } else
#endif
            if (INTEG && fastp && !ool) {
                publish_next(true);
                SEG(5)   /* post */
                // two-body term of this stage (orbital.rs:86-92)
                {
                    const double rmag = norm3(ys[0], ys[1], ys[2]);
                    const double f = -cfg->mu_central / cube(rmag);
                    acc[0] = f * ys[0]; acc[1] = f * ys[1]; acc[2] = f * ys[2];
                }
                // velocity part of sum_{j<i} a_{i+1,j} k_j (phase A of the next stage adds the newest term)
#pragma unroll
                for (int e = 0; e < 6; ++e) wpre[e] = 0.0;
                if (i + 1 < stages) {
#pragma unroll
                    for (int j = 0; j < DEV_MAX_STAGES - 2; ++j) {
                        if (j < i) {  // uniform
                            const double a_nj = A_ROW(i + 1, j);
#pragma unroll
                            for (int e = 3; e < 6; ++e) wpre[e] += a_nj * KB(j, e);
                        }
                    }
                }
                // position part of the stage sum the NEXT window publishes from: k_j[0..2] are the stage velocities, this stage's (ys[3..5],
                // written to k_i in phase C) included - j ascending from 0.0, the newest term last, as the plain loop adds them
                if (i + 2 < stages) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) pre_wr[e] = 0.0;
#pragma unroll
                    for (int j = 0; j < DEV_MAX_STAGES - 2; ++j) {
                        if (j < i) {  // uniform
                            const double a_nj = A_ROW(i + 2, j);
#pragma unroll
                            for (int e = 0; e < 3; ++e) pre_wr[e] += a_nj * KB(j, e);
                        }
                    }
                    const double a_ni = A_ROW(i + 2, i);
#pragma unroll
                    for (int e = 0; e < 3; ++e) pre_wr[e] += a_ni * ys[3 + e];
                } else if (i + 2 == stages && spec) {
                    // the next window is the last: it publishes stage 0 of the next attempt, y + sum_j (h b_j) k_j
#pragma unroll
                    for (int e = 0; e < 3; ++e) pre_wr[e] = CS_Y(e);
                    for (int j = 0; j < i; ++j) {
                        const double cb = h * B_COEF(j);
#pragma unroll
                        for (int e = 0; e < 3; ++e) pre_wr[e] += cb * KB(j, e);
                    }
                    const double cbi = h * B_COEF(i);
#pragma unroll
                    for (int e = 0; e < 3; ++e) pre_wr[e] += cbi * ys[3 + e];
                }
            } else
            if (INTEG) {
                // two-body term of this stage (orbital.rs:86-92) and sum_{j<i} a_{i+1,j} k_j of the next one
                if (!offl) {
                    const double rmag = norm3(ys[0], ys[1], ys[2]);
                    const double f = -cfg->mu_central / cube(rmag);
                    acc[0] = f * ys[0]; acc[1] = f * ys[1]; acc[2] = f * ys[2];
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) wpre[e] = 0.0;
                if (offl) {
                    // the terms j <= i - 3 of the sum were added up - from 0.0, j ascending: the same additions - by an almanac wave
                    // in the previous window (every k_j it read was behind a barrier by then); the two newest terms are added here
                    if (i + 1 < stages) {
                        int j0 = 0;
                        if (i >= 3) {
                            const double *const qb = L.part + ((i + 1) & 1) * 6 * DEV_LANES;
#pragma unroll
                            for (int e = 0; e < 6; ++e) wpre[e] = qb[e * DEV_LANES + lane];
                            j0 = i - 2;
                        }
#pragma unroll 2
                        for (int j = j0; j < i; ++j) {
                            const double a_nj = A_ROW(i + 1, j);
#pragma unroll
                            for (int e = 0; e < 6; ++e) wpre[e] += a_nj * KB(j, e);
                        }
                    }
                } else
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
                publish_next(false);
            }
            // quad layout: the position-only parts of phase C (two-body dual, the duals of s, t, u and (mu / r) / R_eq) are formed
            // HERE, inside the window, where the integrator wave has nothing else to do (after the next stage's inputs: those gate the column waves)
            if (INTEG && STM && QUAD && !qoff) quad_pre(cfg, edc, ys[0], ys[1], ys[2], ql, lane, L.qpre, has_grav);
            SEG(6)   /* two-body + stage sums */
            if (prof_on) prof_acc[1] += (int64_t)__builtin_readcyclecounter() - ptw_;
            const int64_t pth_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
            double px = 0.0, py = 0.0, pz = 0.0, pw = 0.0;
            double coop_x = 0.0, coop_y = 0.0, coop_z = 0.0, coop_w = 0.0;
            if (has_grav && dbg_skip_harm && !INTEG) {
                double *pp = L.part + wave * 4 * DEV_LANES;
                pp[0 * DEV_LANES + lane] = 0.0; pp[1 * DEV_LANES + lane] = 0.0;
                pp[2 * DEV_LANES + lane] = 0.0; pp[3 * DEV_LANES + lane] = 0.0;
            }
            if (STM && QUAD) {
                if (has_grav && cfg->sched[DEV_SCHED_SOLO].n_ranges[wave] > 0)  // (a wave without columns keeps the zeros of its slot)
                    harmonics_partial_d1((uint64_t)cfg, (uint64_t)htab, (uint64_t)cols, wave, (LdsCPtr)((pipe && (i & 1)) ? L.inb2 : L.inbD),
                                         (LdsPtr)(L.partD + wave * QSLOT), lane, LCTL + 3, pipe ? i : 0);
            } else if (STM && has_grav)
                harmonics_partial_dual((uint64_t)cfg, (uint64_t)htab, (uint64_t)cols, wave, L.inbD, L.partD + wave * 16 * DEV_LANES, lane);
            if (!STM && has_grav && !dbg_skip_harm) {
                // (ctl[1] is written before the barrier that precedes this read: B1 for stage 0, B2 of the previous stage otherwise;
                //  the integrator wave itself carries no columns in a pipelined workgroup and uses whatever it just wrote)
                const int sched = LCTL[1] ? DEV_SCHED_PRIMARY : DEV_SCHED_SOLO;
                const double *const inbw = (pipe && (i & 1)) ? L.inb2 : L.inb;
                Partial4 pr = {0.0, 0.0, 0.0, 0.0};
                if (!(pipe && INTEG)) {
                    const double v0 = inbw[0 * DEV_LANES + lane], v1 = inbw[1 * DEV_LANES + lane], v2 = inbw[2 * DEV_LANES + lane],
                                 v3 = inbw[3 * DEV_LANES + lane], v4 = inbw[4 * DEV_LANES + lane];
                    pr = (cfg->harm_feed & 1) ? harmonics_stream((uint64_t)cfg, (uint64_t)cols, wave, sched, v0, v1, v2, v3, v4)
                                        : harmonics_partial((uint64_t)cfg, (uint64_t)htab, (uint64_t)cols, wave, sched, v0, v1, v2, v3, v4);
                }
                px = pr.x; py = pr.y; pz = pr.z; pw = pr.w;
                if (!INTEG) {
                    if (pipe && (i > 0 || spec_now)) {  // the integrator folds the partials of stage i-1 at the start of this window
                        int spin = 0;
                        while (LCTL[3] < fold_base + i && ++spin < 4000000) __builtin_amdgcn_s_sleep(1);
                    }
                    double *pp = L.part + wave * 4 * DEV_LANES;
                    pp[0 * DEV_LANES + lane] = px; pp[1 * DEV_LANES + lane] = py;
                    pp[2 * DEV_LANES + lane] = pz; pp[3 * DEV_LANES + lane] = pw;
                }
            }
#define COOP_COLLECT()                                                                                                                          \
            if (INTEG && !STM && has_grav && (pipe ? shared_cur : coop_on)) {                                                                     \
                const CoopAnswer ans = !coop_on ? CoopAnswer{0.0, 0.0, 0.0, 0.0, 0} : (coop_two ? coop_wait2(cbox, bt.coop_out2 + blockIdx.x, lane, seq_cur) : coop_wait(cbox, lane, seq_cur)); \
                if (ans.ok) {                                                                                                                     \
                    coop_x = ans.x; coop_y = ans.y; coop_z = ans.z; coop_w = ans.w;                                                               \
                    ++dbg_answers;                                                                                                                \
                } else {                                                                                                                          \
                    if (coop_on) { ++dbg_fallbacks; dbg_fb_seq = seq_cur; }  /* no answer in time: do the helper's columns here, then carry on alone */ \
                    const Partial4 fb = coop_fallback((uint64_t)cfg, (uint64_t)htab, (uint64_t)cols, (pipe && (i & 1)) ? L.inb2 : L.inb, lane | COOP_FB_PARTS); \
                    coop_x = fb.x; coop_y = fb.y; coop_z = fb.z; coop_w = fb.w;                                                                   \
                    coop_on = false;                                                                                                              \
                    coop_drop = !pipe;  /* (pipelined: ctl[1] is rewritten for every stage, nothing to undo) */                                   \
                    if (lane == 0) coop_store(bt.coop_finished + coop_widx, 1u);                                                                  \
                }                                                                                                                                 \
            }
            // The helper's answer.  Rounds 1-4 collected it INSIDE the window, in front of B2 - a late answer then held the whole
            // workgroup at the barrier.  Pipelined loop (round 5): it is collected in phase C, BEHIND B2: the column waves are already
            // walking stage i + 1 (its inputs were published in this window), and what a late answer delays is this wave's serial chain
            // C(i) -> A(i + 1) -> the inputs of stage i + 2, which has ~12 k cycles to spare before the column waves ask for them.  The
            // deadline of a job moves out by that much: the helpers can be loaded further.  (The inputs of this stage in LDS - the
            // fallback's operands - are not overwritten before window i + 1 publishes stage i + 2 into the same parity: behind this point.)
            if ((!pipe || cfg->coop_late == 0) && !ool) {  // (INTEG_OOL: always collected in phase C)
                const int64_t w0_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
                COOP_COLLECT()
                if (prof_on && INTEG) { pl_tc = (int64_t)__builtin_readcyclecounter(); pl_wait += pl_tc - w0_; }
            }
            if (prof_on) prof_acc[2] += (int64_t)__builtin_readcyclecounter() - pth_;
            {
                PROF_T0();
                __syncthreads();  // B2: partials / perturbations / next epoch data published
                PROF_ADD(6);
            }
            SEG(7)   /* barrier */
            if (spec_now && i == 0 && LCTL[0]) {  // every lane had finished: the exit, one window late
                leave = true;
                break;
            }
            const int64_t ptc_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;

#if INTEG_OOL
            if (INTEG && ool) {
                // ---- Phase C, out of line (integ_back)
                const int skip_k = (spec_now && i == 0 && keep_k0) ? 1 : 0;
                // fixed wave order; all 15 slots are read unconditionally (slots of absent waves hold an exact 0.0); this wave walks no columns: 0.0
                const Partial4 f4 = fold_partials((LdsCPtr)L.part, lane, 0.0, 0.0, 0.0, 0.0);
                const IxBack rb_ = integ_back(lds_base, (uint64_t)cfg, i, (shared_cur ? IX_SHARED : 0) | (coop_on ? IX_COOP : 0) | (prof_on ? IX_PROF : 0), lane,
                                              acc[0], acc[1], acc[2], f4.x, f4.y, f4.z, f4.w, seq_cur, fold_base + i + 1, (uint64_t)cbox,
                                              (uint64_t)(bt.coop_out2 + blockIdx.x), skip_k);
                if (rb_.ret & 0xffff) st_att = rb_.ret & 0xffff;
                if (rb_.ret & IXR_ANSWER) ++dbg_answers;
                if (rb_.ret & IXR_NEED_FB) {  // (uniform) no answer in time: do the helper's columns here, then carry on alone
                    if (coop_on) { ++dbg_fallbacks; dbg_fb_seq = seq_cur; }
                    const Partial4 fb = coop_fallback((uint64_t)cfg, (uint64_t)htab, (uint64_t)cols, (i & 1) ? L.inb2 : L.inb, lane | COOP_FB_PARTS);
                    integ_back_slow(lds_base, (uint64_t)cfg, (uint64_t)records, i, lane, acc[0], acc[1], acc[2], rb_.px, rb_.py, rb_.pz, rb_.pw, fb.x, fb.y, fb.z, fb.w, skip_k);
                    if (sums_on) {  // (k_i is written: see integ_back)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) LCTL[6] = fold_base + i + 1;
                    }
                    coop_on = false;  // (pipelined: ctl[1] is rewritten for every stage, nothing to undo)
                    if (lane == 0) coop_store(bt.coop_finished + coop_widx, 1u);
                }
                if (prof_on && shared_cur) { pl_tc = IX_STAMP(0); pl_wait += pl_tc - IX_STAMP(3); }
            } else
#endif
            if (INTEG) {
                // ---- Phase C: assemble the derivative in the reference's order (orbital.rs:80-114, spacecraft.rs:227-243)
                const double *const pertc = (pipe && (i & 1)) ? L.pert2 : L.pert;
                if (offl) {  // the two-body term, formed beside the window by an almanac wave from the published position
                    const double *const tb = L.part + (12 + 3 * (i & 1)) * DEV_LANES;
                    acc[0] = tb[0 * DEV_LANES + lane]; acc[1] = tb[1 * DEV_LANES + lane]; acc[2] = tb[2 * DEV_LANES + lane];
                }
                if (!STM && (has_pm || has_tides || has_grav2)) {
                    acc[0] += pertc[0 * DEV_LANES + lane]; acc[1] += pertc[1 * DEV_LANES + lane]; acc[2] += pertc[2 * DEV_LANES + lane];
                }
                if (has_grav2 && !(spec_now && i == 0 && keep_k0)) {  // the second field's orientation status of THIS stage (written in the window B2 has just closed; a rejected lane's speculative stage 0 does not count)
                    const int es = L.pertst[(i & 1) * DEV_LANES + lane];
                    if (es) st_att = es;
                }
                if (!STM && has_grav) {
                    // fixed wave order; all 15 slots are read unconditionally (slots of absent waves hold an exact
                    // 0.0) so that the LDS reads carry no control dependence and pipeline
                    {
                        const Partial4 f4 = fold_partials((LdsCPtr)L.part, lane, px, py, pz, pw);
                        px = f4.x; py = f4.y; pz = f4.z; pw = f4.w;
                    }
                    if (pipe) {  // the partial sums of stage i are in registers: the workers may overwrite their slots
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) LCTL[3] = fold_base + i + 1;
                    }
                    SEG(8)   /* phase C up to the fold */
                    if (pipe && cfg->coop_late != 0) {
                        const int64_t w0_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
                        COOP_COLLECT()
                        SEG(9)   /* the answer */
                        if (prof_on) { pl_tc = (int64_t)__builtin_readcyclecounter(); pl_wait += pl_tc - w0_; }
                    }
                    px += coop_x; py += coop_y; pz += coop_z; pw += coop_w;  // + the helper's columns (0 when working alone)
                    if (coop_drop) {  // (between B2 and the next B1: no worker is reading ctl[1])
                        if (lane == 0) L.ctl[1] = 0;
                        coop_drop = false;
                    }
                    px *= kfac; py *= kfac; pz *= kfac; pw *= kfac;
                    const double al0 = px + pw * s_, al1 = py + pw * t_, al2 = pz + pw * u_;
                    // DCM of THIS stage from registers: in the pipelined loop the almanac wave is already overwriting that LDS buffer
                    acc[0] += m_cur[0] * al0 + m_cur[3] * al1 + m_cur[6] * al2;
                    acc[1] += m_cur[1] * al0 + m_cur[4] * al1 + m_cur[7] * al2;
                    acc[2] += m_cur[2] * al0 + m_cur[5] * al1 + m_cur[8] * al2;
                }
                if (!STM && has_srp) {
                    acc[0] += pertc[3 * DEV_LANES + lane]; acc[1] += pertc[4 * DEV_LANES + lane]; acc[2] += pertc[5 * DEV_LANES + lane];
                }
                if (!STM && has_drag) {
                    acc[0] += pertc[6 * DEV_LANES + lane]; acc[1] += pertc[7 * DEV_LANES + lane]; acc[2] += pertc[8 * DEV_LANES + lane];
                }
                if (STM && QUAD) {
                    // (out of line, see phase_c_quad: it also writes k_i)
                    phase_c_quad((LdsCPtr)((pipe && (i & 1)) ? L.pert2 : L.pertD), (LdsCPtr)L.partD, (LdsCPtr)L.qpre,
                                 (LdsCPtr)((pipe && (i & 1)) ? L.ys2 : L.ys), (LdsPtr)L.sacc,
                                 (LdsPtr)(kbuf + (i * 6) * KB_STR + kb_li), KB_STR, B_COEF(i), nw,
                                 ((has_pm || has_tides || has_grav2) ? PC_HAS_PM : 0) | (has_grav ? PC_HAS_GRAV : 0) | (has_srp ? PC_HAS_SRP : 0), lane, ql,
                                 LCTL + 3, pipe ? i + 1 : 0, prof_on ? bt.prof + 16 * 8 : nullptr);
                    if (qoff) {  // L.qpre of this stage has been read: the wave that forms it may write the next stage's
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) LCTL[6] = i + 1;
                    }
                } else if (STM) {
                    // dual path (dual_eom, spacecraft.rs:312-363): f(x) and A = df/dx; the derivative written to k_i is
                    // the dual path's real part, as in the reference's STM branch (spacecraft.rs:208-224)
                    double G[9], cv[3] = {0.0, 0.0, 0.0};
                    const D3 rad[3] = {{ys[0], 1.0, 0.0, 0.0}, {ys[1], 0.0, 1.0, 0.0}, {ys[2], 0.0, 0.0, 1.0}};
                    const D3 fac = d3div(d3c(-cfg->mu_central), d3cube(d3norm(rad[0], rad[1], rad[2])));
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const D3 a = rad[q] * fac;
                        acc[q] = a.v; G[3 * q + 0] = a.x; G[3 * q + 1] = a.y; G[3 * q + 2] = a.z;
                    }
                    if (has_pm || has_tides || has_grav2) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc[q] += L.pertD[q * DEV_LANES + lane];
#pragma unroll
                        for (int q = 0; q < 9; ++q) G[q] += L.pertD[(3 + q) * DEV_LANES + lane];
                    }
                    if (has_grav) {
                        D3 pD[4] = {d3c(0.0), d3c(0.0), d3c(0.0), d3c(0.0)};
                        for (int w = 0; w < nw; ++w) {  // fixed wave order
                            const double *pp = L.partD + w * 16 * DEV_LANES;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                pD[q].v += pp[(4 * q + 0) * DEV_LANES + lane]; pD[q].x += pp[(4 * q + 1) * DEV_LANES + lane];
                                pD[q].y += pp[(4 * q + 2) * DEV_LANES + lane]; pD[q].z += pp[(4 * q + 3) * DEV_LANES + lane];
                            }
                        }
                        double m[9];
#pragma unroll
                        for (int q = 0; q < 9; ++q) m[q] = edc[q * DEV_LANES + lane];
                        // s, t, u, (mu / r) / R_eq as duals of the body-fixed position (recomputed: cheaper than 16 LDS slots)
                        double rg[3] = {ys[0], ys[1], ys[2]};
                        if (cfg->g_slot >= 0) {  // (uniform) the field of another body: at r - r_body(t), as phase A formed the inputs
                            double pg[3];
                            ed_body(cfg, edc, lane, cfg->g_slot, pg);
                            rg[0] = ys[0] - pg[0]; rg[1] = ys[1] - pg[1]; rg[2] = ys[2] - pg[2];
                        }
                        const D3 x0 = {m[0] * rg[0] + m[1] * rg[1] + m[2] * rg[2], 1.0, 0.0, 0.0};
                        const D3 x1 = {m[3] * rg[0] + m[4] * rg[1] + m[5] * rg[2], 0.0, 1.0, 0.0};
                        const D3 x2 = {m[6] * rg[0] + m[7] * rg[1] + m[8] * rg[2], 0.0, 0.0, 1.0};
                        const D3 rD = d3norm(x0, x1, x2);
                        const D3 aux[4] = {d3div(x0, rD), d3div(x1, rD), d3div(x2, rD), d3div(d3div(d3c(cfg->g_mu), rD), d3c(cfg->g_re))};
#pragma unroll
                        for (int q = 0; q < 4; ++q) pD[q] = pD[q] * aux[3];
                        const D3 al[3] = {pD[0] + pD[3] * aux[0], pD[1] + pD[3] * aux[1], pD[2] + pD[3] * aux[2]};
                        // a = R^T a_bf ; G_h = R^T G_bf R   (gravity_field.rs:403-430)
                        double tmp[9];
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            acc[a] += m[0 + a] * al[0].v + m[3 + a] * al[1].v + m[6 + a] * al[2].v;
                            tmp[3 * a + 0] = m[0 + a] * al[0].x + m[3 + a] * al[1].x + m[6 + a] * al[2].x;
                            tmp[3 * a + 1] = m[0 + a] * al[0].y + m[3 + a] * al[1].y + m[6 + a] * al[2].y;
                            tmp[3 * a + 2] = m[0 + a] * al[0].z + m[3 + a] * al[1].z + m[6 + a] * al[2].z;
                        }
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int b = 0; b < 3; ++b)
                                G[3 * a + b] += tmp[3 * a + 0] * m[0 + b] + tmp[3 * a + 1] * m[3 + b] + tmp[3 * a + 2] * m[6 + b];
                    }
                    if (has_srp) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) { acc[q] += L.pertD[(12 + q) * DEV_LANES + lane]; cv[q] = L.pertD[(24 + q) * DEV_LANES + lane]; }
#pragma unroll
                        for (int q = 0; q < 9; ++q) G[q] += L.pertD[(15 + q) * DEV_LANES + lane];
                    }
                    const double b_i = B_COEF(i);
#pragma unroll
                    for (int q = 0; q < 9; ++q) L.sacc[q * DEV_LANES + lane] += b_i * G[q];
#pragma unroll
                    for (int q = 0; q < 3; ++q) L.sacc[(9 + q) * DEV_LANES + lane] += b_i * cv[q];
                    if (bt.stm_hist != nullptr && valid) {  // (uniform) the textbook form replays the tableau over the stage matrices at the accepted step
#pragma unroll
                        for (int q = 0; q < 9; ++q) bt.stm_hist[(int64_t)(i * 12 + q) * bt.stm_hist_stride + gid] = G[q];
#pragma unroll
                        for (int q = 0; q < 3; ++q) bt.stm_hist[(int64_t)(i * 12 + 9 + q) * bt.stm_hist_stride + gid] = cv[q];
                    }
                }
                if (!(STM && QUAD) && !(spec_now && i == 0 && keep_k0)) {
                    KB(i, 0) = ys[3]; KB(i, 1) = ys[4]; KB(i, 2) = ys[5];
                    KB(i, 3) = acc[0]; KB(i, 4) = acc[1]; KB(i, 5) = acc[2];
                }
            }
            SEG(10)  /* rest of phase C */
            if (prof_on) prof_acc[3] += (int64_t)__builtin_readcyclecounter() - ptc_;
        }
        if (leave) break;

        const int64_t pts_ = prof_on ? (int64_t)__builtin_readcyclecounter() : 0;
#if NYX_SEG_PROF
        if (INTEG && STM && prof_on) sb_t = (int64_t)__builtin_readcyclecounter();
#endif
        keep_k0 = false;
#if INTEG_OOL && STEP_OOL
        if (INTEG && ool && !bt.ev_on) {
            // ---- step control, out of line (integ_step)
            const IxStep sr = integ_step(lds_base, (uint64_t)cfg, lane, h, st_att, att, spec ? IXS_CHAIN : 0);
            keep_k0 = (sr.ret & IXS_KEEP_K0) != 0;
            if (spec) h_next = sr.h_next;
            if ((sr.ret & IXS_ACCEPT) && bt.traj_cap > 0 && wr) {  // chan.send(self.state) after every accepted step, final one included
                const int64_t acc_n = __double_as_longlong(L.cs[5 * DEV_LANES + lane]);
                if (acc_n < bt.traj_cap) {
                    const int64_t at = acc_n * bt.n + gid;
                    bt.t_epoch[at] = __double_as_longlong(L.cs[0 * DEV_LANES + lane]);
#pragma unroll
                    for (int e = 0; e < 6; ++e) bt.t_state[e][at] = CS_Y(e);
                }
                bt.t_len[gid] = (int32_t)(acc_n + 1);
            }
        } else
#endif
        if (INTEG) {
            ColdState c;
            cold_load(L.cs, lane, c);
            double *const y = c.y;
            if (!c.done) c.n_evals += stages;
            SEG(12)  /* cold state */
            // ---- next state and error estimate (instance.rs:401-414).  d(Cr, Cd, prop mass)/dt = 0.
            double next[9], err[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) { next[e] = y[e]; err[e] = 0.0; }
#if STEP_SUMS_UNROLL
#pragma unroll STEP_SUMS_UNROLL
#endif
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
            SEG(13)  /* the two sums */
            const double h_used = h;
            bool accept = false, ev_hit = false;
#if STEP_ONE_POW
            // The error estimate, the accept test and the controller's power for every lane at once, in front of the branches: an attempt
            // of configs[1] is rejected on 18 % of the lanes, so a wave nearly always walked BOTH branches below, each with its own inlined
            // pow (the two differ in the exponent only).  The same function of the same arguments: the same bits.
            double de = c.det_error, pw = 0.0;
            bool take = false;
            if (__any(!c.done && st_att == NYX_HIP_OK && !c.fixed)) {  // (uniform)
                de = error_estimate(cfg->error_ctrl, err, next, y);
                take = de <= cfg->tol || h <= cfg->min_step_s || c.attempts >= cfg->attempts;
                pw = pow(cfg->tol / de, take ? cfg->inv_order : cfg->inv_order_m1);
            }
#endif
            if (!c.done) {
                if (st_att != NYX_HIP_OK) {
                    c.status = st_att;
                    c.done = true;
                } else if (c.fixed) {
                    c.det_step = c.step_size;
                    accept = true;
                } else {
#if STEP_ONE_POW
                    c.det_error = de;
                    if (take) {
#else
                    c.det_error = error_estimate(cfg->error_ctrl, err, next, y);
                    if (c.det_error <= cfg->tol || h <= cfg->min_step_s || c.attempts >= cfg->attempts) {
#endif
                        bool nan = false;
#pragma unroll
                        for (int e = 0; e < 9; ++e) nan = nan || (next[e] != next[e]);
                        if (nan) {
                            c.status = NYX_HIP_ERR_NAN;
                            c.done = true;
                        } else {
                            c.det_step = seconds_to_ns(h);
                            if (c.det_error < cfg->tol) {
#if STEP_ONE_POW
                                const double prop = 0.9 * h * pw;
#else
                                const double prop = 0.9 * h * pow(cfg->tol / c.det_error, cfg->inv_order);
#endif
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
#if STEP_ONE_POW
                        const double prop = 0.9 * h * pw;
#else
                        const double prop = 0.9 * h * pow(cfg->tol / c.det_error, cfg->inv_order_m1);
#endif
                        h = (prop < cfg->min_step_s) ? cfg->min_step_s : prop;
                        keep_k0 = true;
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
                    // stop condition: checked after every step but the final fixed one; the triggering state is returned,
                    // not published (instance.rs:243-252)
                    if (bt.ev_on && valid && !c.is_final)  // (quad layout: the four lanes read and write the same words with the same values)
                        ev_hit = event_step(bt.ev, bt.ev_mu, c.epoch, bt.ev_prev + gid, bt.ev_count + gid, y[0], y[1],
                                            y[2], y[3], y[4], y[5]);
                    if (ev_hit) {
                        if (wr) bt.ev_found[gid] = 1;
                        c.done = true;
                    }
                }
                bool stm_bad = false;
                if (accept && STM && valid) {
                    double sumb = 0.0;
                    for (int q = 0; q < stages; ++q) sumb += B_COEF(q);
                    if (!QUAD && bt.stm_hist != nullptr)
                        stm_bad = stm_update_textbook(bt.o_stm + gid * 81, h_used, bt.stm_hist, bt.stm_hist_stride, gid, kbuf, tabl, stages, lane);
                    else
                    stm_bad = QUAD ? stm_update_q(bt.o_stm + gid * 81, h_used, L.sacc, lane, ql, sumb)
                                   : stm_update(bt.o_stm + gid * 81, h_used, L.sacc, lane, sumb);
                }
                if (accept) {
                    if (stm_bad) { c.status = NYX_HIP_ERR_NAN; c.done = true; }
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
            SEG(14)  /* error estimate, decision, state update */
            // what is left of an accepted step only writes results: with chained attempts the next one is opened first (all lanes
            // together: the exit test is a wave vote), the other waves are waiting for its epoch and step
            const int64_t acc_n = c.n_acc, acc_epoch = c.epoch;
            if (spec) {
                begin_attempt(c);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) LCTL[5] = att + 1;  // the almanac wave waits for this word before it reads the new epoch and step
                h_next = c.h;
            }
            SEG(15)  /* next attempt opened */
            if (accept && bt.traj_cap > 0 && wr && !ev_hit) {  // chan.send(self.state) after every accepted step, final one included
                if (acc_n < bt.traj_cap) {
                    const int64_t at = acc_n * bt.n + gid;
                    bt.t_epoch[at] = acc_epoch;
#pragma unroll
                    for (int e = 0; e < 6; ++e) bt.t_state[e][at] = y[e];
                }
                bt.t_len[gid] = (int32_t)(acc_n + 1);
            }
            cold_store(L.cs, lane, c);
        }
        SEG(11)  /* step control */
        SBD(0)
        if (prof_on) prof_acc[4] += (int64_t)__builtin_readcyclecounter() - pts_;
        spec_now = spec;
        ++att;
    }
    if (prof_on && lane == 0 && INTEG) {
        int64_t *row = bt.prof + 33 * 8;
        row[0] = pl_wait; row[1] = pl_chain; row[2] = pl_post; row[3] = pl_n;
#if NYX_SEG_PROF
        for (int q = 0; q < 16; ++q) bt.prof[34 * 8 + q] = sg[q];
        if (STM) { for (int q = 0; q < 8; ++q) bt.prof[18 * 8 + q] = sb[q]; }
#endif
    }
    if (prof_on && lane == 0) {
        prof_acc[5] = (int64_t)__builtin_readcyclecounter() - prof_start;
        prof_acc[7] = (int64_t)__builtin_amdgcn_s_memrealtime() - prof_rt0;
        for (int q = 0; q < 8; ++q) bt.prof[wave * 8 + q] = prof_acc[q];
    }

    if (INTEG && coop_started && lane == 0) coop_store(bt.coop_finished + coop_widx, 1u);
    if (INTEG && bt.prof != nullptr && lane == 0) {
        atomicAdd((unsigned long long *)bt.prof + 16 * 8 + 0, dbg_answers);
        atomicAdd((unsigned long long *)bt.prof + 16 * 8 + 1, dbg_fallbacks);
        atomicAdd((unsigned long long *)bt.prof + 16 * 8 + 2, dbg_fb_seq);
        atomicAdd((unsigned long long *)bt.prof + 16 * 8 + 3, (unsigned long long)coop_seq);
    }
    if (INTEG && wr) {
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
    asm volatile("s_nop 13\n\ts_nop 15" ::: "memory");
}

#undef coop_two
#undef COOP_FB_PARTS

template <bool STM, bool QUAD = false, bool W16 = true>  // W16: sixteen-wave workgroups (the shape whose pipelined stage loop serves the column waves)
DEVFN void propagate_body(const DevBatch &bt, const DevCfg *cfg_g, const HarmEntry *htab_g, const ColHdr *cols_g,
                          const double *__restrict__ records, char *smem) {
    const int lane = threadIdx.x & (DEV_LANES - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    // LDS comes as the previous workgroup on this CU left it.  Everything below is written before it is read by design, but a run
    // whose outcome could depend on what ran on the CU before (round 3: a cooperative launch that never finished, only after ~90
    // other tests in the same process) is not something to leave to design: every workgroup starts from zeroed LDS (20 k stores).
    for (int q = (int)threadIdx.x; q < bt.lds_bytes / 8; q += (int)blockDim.x) ((double *)smem)[q] = 0.0;
    __syncthreads();
    const LdsMap L = carve_lds(smem, nw, STM, cfg_g->rec_in_lds ? cfg_g->rec_doubles : 0, STM ? 0 : cfg_g->ed_reuse, QUAD);
    double *const kbuf = L.kbuf;
    double *const tabl = L.tabl;
    CfgPtr cfg = (CfgPtr)cfg_g;
    HarmPtr htab = (HarmPtr)htab_g;
    ColPtr cols = (ColPtr)cols_g;

    // cooperative mode: blocks past the trajectory-owning ones are padding (up to coop_base) or helpers
    if (!STM && bt.coop_helpers > 0) {
        const int64_t n_own = (bt.n + DEV_LANES - 1) / DEV_LANES;
        if ((int64_t)blockIdx.x >= n_own) {
            if ((int)blockIdx.x >= bt.coop_base && !(bt.coop_mute & 1)) helper_body(bt, cfg, htab, cols, smem, lane, wave);
            return;
        }
    }

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
    if (threadIdx.x == 0) {
        L.ctl[0] = 0;
        L.ctl[5] = 0;  // chained attempts: attempts published (LDS comes as the previous workgroup left it)
        // ctl[1]: 1 while this workgroup shares its columns with the helpers
        L.ctl[1] = (!STM && bt.coop_helpers > 0 && cfg->has_grav) ? 1 : 0;
    }
    for (int q = (int)threadIdx.x; q < (QUAD ? DEV_MAX_WAVES * QSLOT : DEV_MAX_WAVES * 4 * DEV_LANES); q += (int)blockDim.x) L.part[q] = 0.0;


    // ---- role dispatch (wave-uniform): the host deals the duties (cfg->role_kind / role_mask, see build_schedule)
    if constexpr (W16 ? (!STM || QUAD) : !STM) {  // (small shapes: the plain kernel only, for dynamics without a gravity field)
        if (cfg->pipe != 0 && cfg->role_kind[wave] != DEV_ROLE_ALL) {  // pipelined stage loop (uniform; the host sets cfg->pipe per shape)
            switch (cfg->role_kind[wave]) {
            case DEV_ROLE_INTEG: role_loop<true, false, false, STM, QUAD, true>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
            case DEV_ROLE_ALMANAC: role_loop<false, true, false, STM, QUAD, true>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
            case DEV_ROLE_PERT: role_loop<false, false, true, STM, QUAD, true>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
            case DEV_ROLE_ALMANAC_PERT: role_loop<false, true, true, STM, QUAD, true>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
            default: role_loop<false, false, false, STM, QUAD, true>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
            }
            return;
        }
    }
    switch (cfg->role_kind[wave]) {
    case DEV_ROLE_ALL: role_loop<true, true, true, STM, QUAD>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
    case DEV_ROLE_INTEG: role_loop<true, false, false, STM, QUAD>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
    case DEV_ROLE_ALMANAC: role_loop<false, true, false, STM, QUAD>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
    case DEV_ROLE_PERT: role_loop<false, false, true, STM, QUAD>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
    case DEV_ROLE_ALMANAC_PERT: role_loop<false, true, true, STM, QUAD>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
    default: role_loop<false, false, false, STM, QUAD>(bt, cfg, cfg_g, htab, cols, records, L, lane, wave, nw); break;
    }
}

// One kernel per workgroup SHAPE, each in its own translation unit (NYX_EMIT selects; propagate_*.hip include this file): the
// register budget follows __launch_bounds__ - 128 VGPRs for sixteen waves, 256 for eight or fewer - so the role code of the
// small shapes (fan-out workgroups of dynamics without a gravity field, the quad STM layout on eight waves) is compiled without
// the 128-VGPR cap that sixteen waves per workgroup impose, instead of one instantiation serving every shape.
#if NYX_PROF
#define NYX_KN(NAME) NAME##_prof
#else
#define NYX_KN(NAME) NAME
#endif
#define NYX_KERNEL(NAME, THREADS, ...) NYX_KERNEL_(NYX_KN(NAME), THREADS, __VA_ARGS__)
#define NYX_KERNEL_(NAME, THREADS, ...)                                                                                       \
    extern "C" __global__ void __launch_bounds__(THREADS)                                                                     \
        NAME(DevBatch bt, const DevCfg *cfg_g, const HarmEntry *htab_g, const ColHdr *cols_g, const double *__restrict__ records) { \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                                           \
        propagate_body<__VA_ARGS__>(bt, cfg_g, htab_g, cols_g, records, smem);                                                \
    }
#define NYX_KERNEL_DECL(NAME) \
    extern "C" __global__ void NAME(DevBatch bt, const DevCfg *cfg_g, const HarmEntry *htab_g, const ColHdr *cols_g, const double *__restrict__ records); \
    extern "C" __global__ void NAME##_prof(DevBatch bt, const DevCfg *cfg_g, const HarmEntry *htab_g, const ColHdr *cols_g, const double *__restrict__ records);
#if NYX_EMIT & NYX_EMIT_PLAIN16
NYX_KERNEL(nyx_propagate_kernel, DEV_MAX_WAVES *DEV_LANES, false)
#endif
#if NYX_EMIT & NYX_EMIT_PLAIN8
NYX_KERNEL(nyx_propagate_kernel_w8, 8 * DEV_LANES, false, false, false)
#endif
#if NYX_EMIT & NYX_EMIT_STM
NYX_KERNEL(nyx_propagate_kernel_stm, DEV_MAX_WAVES_STM *DEV_LANES, true)
#endif
#if NYX_EMIT & NYX_EMIT_STMQ16
NYX_KERNEL(nyx_propagate_kernel_stmq, DEV_MAX_WAVES *DEV_LANES, true, true)
#endif
#if NYX_EMIT & NYX_EMIT_STMQ8
NYX_KERNEL(nyx_propagate_kernel_stmq_w8, 8 * DEV_LANES, true, true, false)
#endif
#if NYX_EMIT & NYX_EMIT_PLAIN16_P2
NYX_KERNEL(nyx_propagate_kernel_p2, DEV_MAX_WAVES *DEV_LANES, false)
#endif
#if NYX_EMIT & NYX_EMIT_PLAIN8N
NYX_KERNEL(nyx_propagate_kernel_w8n, 8 * DEV_LANES, false, false, false)
#endif
#if NYX_EMIT & NYX_EMIT_PLAIN16_FAN
NYX_KERNEL(nyx_propagate_kernel_fan, DEV_MAX_WAVES *DEV_LANES, false)
#endif

#if NYX_HOST_TU
NYX_KERNEL_DECL(nyx_propagate_kernel)
NYX_KERNEL_DECL(nyx_propagate_kernel_w8)
NYX_KERNEL_DECL(nyx_propagate_kernel_stm)
NYX_KERNEL_DECL(nyx_propagate_kernel_stmq)
NYX_KERNEL_DECL(nyx_propagate_kernel_stmq_w8)
NYX_KERNEL_DECL(nyx_propagate_kernel_p2)
NYX_KERNEL_DECL(nyx_propagate_kernel_w8n)
NYX_KERNEL_DECL(nyx_propagate_kernel_fan)
extern "C" hipError_t nyx_launch_propagate(const DevBatch &bt, const DevCfg *cfg, const HarmEntry *htab,
                                           const ColHdr *cols, const double *records, int n_waves, int rec_lds_doubles,
                                           int reuse_fields, hipStream_t stream, int quad, int no_body_fixed) {
    const int64_t per_wg = quad ? DEV_LANES / 4 : DEV_LANES;
    const int64_t blocks = (bt.n + per_wg - 1) / per_wg;
    if (blocks == 0) return hipSuccess;
    {
        // the dynamic-LDS limit is a property of the function ON A DEVICE: once per device, and the shard threads of
        // nyx_hip_propagate_batch_sharded arrive here concurrently
        static std::mutex attr_mu;
        static bool attr_set[64] = {false};
        int devid = 0;
        (void)hipGetDevice(&devid);
        std::lock_guard<std::mutex> lk(attr_mu);
        if (devid < 0 || devid >= 64 || !attr_set[devid]) {
#define NYX_LDS_ATTR(K)                                                                                             \
    (void)hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
    (void)hipFuncSetAttribute((const void *)K##_prof, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            NYX_LDS_ATTR(nyx_propagate_kernel)
            NYX_LDS_ATTR(nyx_propagate_kernel_stm)
            NYX_LDS_ATTR(nyx_propagate_kernel_stmq)
            NYX_LDS_ATTR(nyx_propagate_kernel_w8)
            NYX_LDS_ATTR(nyx_propagate_kernel_stmq_w8)
            NYX_LDS_ATTR(nyx_propagate_kernel_p2)
            NYX_LDS_ATTR(nyx_propagate_kernel_w8n)
            NYX_LDS_ATTR(nyx_propagate_kernel_fan)
#undef NYX_LDS_ATTR
            if (devid >= 0 && devid < 64) attr_set[devid] = true;
        }
    }
    const bool stm = bt.o_stm != nullptr;
    size_t lds = nyx_kernel_lds_bytes(n_waves, rec_lds_doubles, stm ? (quad ? 2 : 1) : 0, stm ? 0 : reuse_fields);
    if (!stm && bt.coop_helpers > 0 && lds < (size_t)HELPER_LDS_BYTES) lds = HELPER_LDS_BYTES;
    DevBatch btl = bt;
    btl.lds_bytes = (int32_t)lds;
    const bool small = n_waves <= 8;  // (helpers are sixteen-wave workgroups: cooperative launches never are)
    // the kernel of the shape; with a profile buffer attached, its twin that carries the accounting (NYX_PROF)
#define NYX_PICK(K) (bt.prof != nullptr ? K##_prof : K)
    auto kern = NYX_PICK(nyx_propagate_kernel);
    int64_t grid = blocks;
    if (stm && quad)
        kern = small ? NYX_PICK(nyx_propagate_kernel_stmq_w8) : NYX_PICK(nyx_propagate_kernel_stmq);
    else if (stm)
        kern = NYX_PICK(nyx_propagate_kernel_stm);
    else {
        if (bt.coop_helpers > 0) grid = (int64_t)bt.coop_base + bt.coop_helpers;
        const bool two_parts = bt.coop_helpers > 0 && bt.coop_parts == 2 && bt.coop_out2 != nullptr && bt.coop_fan == 0;  // (its own kernel: NYX_COOP_TWO_PARTS)
        // (no_body_fixed: the host's statement that the configuration has no gravity field, drag or tides - propagate_w8n.hip)
        if (small && bt.coop_helpers == 0)
            kern = no_body_fixed ? NYX_PICK(nyx_propagate_kernel_w8n) : NYX_PICK(nyx_propagate_kernel_w8);
        else if (bt.coop_helpers > 0 && bt.coop_fan != 0)   // (its own kernel: NYX_COOP_FAN)
            kern = NYX_PICK(nyx_propagate_kernel_fan);
        else if (two_parts)
            kern = NYX_PICK(nyx_propagate_kernel_p2);
    }
#undef NYX_PICK
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)(n_waves * DEV_LANES)), lds, stream, btl, cfg, htab, cols, records);
    return hipGetLastError();
}
#endif  // NYX_HOST_TU
