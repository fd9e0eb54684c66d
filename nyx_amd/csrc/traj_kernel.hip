// traj_kernel.hip — batched `Traj::at` / `Traj::every` on the MI355X (gfx950).
//
// Reference: md/trajectory/traj.rs:82-162 (window selection, exact hits, bounds), interpolatable.rs:52-108 (one
// Hermite interpolation per axis over at most 13 stored states, abscissas in f64 seconds) and anise's
// `hermite_eval` = SPICE HRMINT (divided-difference table with doubled abscissas).
//
// Mapping: lane <-> trajectory, as in the propagation kernel, so that the step-major dense output
// ([k * n + i]) is read and the sample-major result ([q * n + i]) is written fully coalesced.  A workgroup is ONE
// wave that owns 64 trajectories x a chunk of consecutive samples (grid.y walks the chunks).  HRMINT's
// divided-difference table (2 x 26 entries per axis) is held in registers: rows unrolled, columns rolled (see
// hrmint_axis); x, y and z are built one after the other by the same code.  No LDS, <= 256 VGPRs, 2 waves per SIMD.
//
// Bound: FP64 VALU.  One sample costs 3 axes x n(2n-1) table updates x 2 IEEE divisions ~ 1 800 divisions
// (n = 13) plus ~12 kFLOP of multiply/add, against 7 x 13 x 8 B = 728 B of (cached, overlapping) reads and 56 B
// written: far to the right of the HBM ridge.  The divisions are the reference's (each table entry is divided by its
// own abscissa difference); they are kept so that the device result equals the CPU restatement bit for bit.
// Compiled with -ffp-contract=off for the same reason.

#include <hip/hip_runtime.h>

#include "../../include/nyx_hip.h"
#include "hifitime_dev.h"
#include "traj_args.h"
#include "event_dev.h"

#define DEVFN static __device__ __forceinline__

namespace {

constexpr int LANES = 64;
constexpr int SAMPLES = 13;                 // INTERPOLATION_SAMPLES, interpolatable.rs:22

// The stored states of one trajectory read as the finalize()d (epoch-sorted) sequence (traj.rs:75-80).
struct View {
    const int64_t *epoch;
    int64_t n, i, len;
    bool desc;
    __device__ __forceinline__ int64_t at(int64_t k) const { return (desc ? len - 1 - k : k) * n + i; }
};

DEVFN View make_view(const nyx_hip_traj_t &t, int64_t n, int64_t i) {
    View v;
    v.epoch = t.epoch_ns;
    v.n = n;
    v.i = i;
    const int64_t produced = t.len[i];
    v.len = produced < t.capacity ? produced : t.capacity;
    v.desc = v.len > 1 && t.epoch_ns[(v.len - 1) * n + i] < t.epoch_ns[i];
    return v;
}

// HRMINT for one axis, table in REGISTERS.  The published routine indexes its work array with loop counters; here the
// row loops are unrolled (static register indices) and only the column loop is rolled, so F (function column),
// D (derivative column) and the abscissas never leave the VGPRs.  The updates of one column are independent of each
// other (each reads only entries the column has not overwritten yet) and are issued WITHOUT per-entry branches, so
// the compiler interleaves their division sequences: that instruction-level parallelism is what hides the FP64
// latency at one or two waves per SIMD.  Entry (i, j) exists for i <= 2n - j; the columns are walked in four groups
// of six with static extents 24/18/12/6, the few entries computed beyond the triangle are never read by a valid
// one (their values, possibly inf/NaN, are dead).  The upper abscissa of entry (i, j), xs[(i+j+1)/2 - 1], moves one
// entry to the left per column: XB is shifted, not indexed.  Lanes with fewer than 13 states (short trajectories,
// the 12-state end windows, ns = 0 for lanes without a window) run the same code under selects.
// Returns false on |denominator| < f64::EPSILON (InterpMath, DivisionByZero) in a VALID entry.
template <int EXTENT>
DEVFN void hrmint_columns(int j0, int n2, double x_eval, const double (&XS)[SAMPLES], double (&F)[2 * SAMPLES],
                          double (&D)[2 * SAMPLES], double (&XB)[2 * SAMPLES], bool &bad, double &f, double &df) {
    const double EPS = 2.220446049250313e-16;
#pragma unroll 1
    for (int j = j0; j < j0 + 6; ++j) {
#pragma unroll
        for (int i = 1; i <= EXTENT; ++i) {
            const double xa = XS[(i + 1) / 2 - 1], xb = XB[i];
            const double c1 = xb - x_eval;
            const double c2 = x_eval - xa;
            const double denom = xb - xa;
            bad = bad || (i <= n2 - j && fabs(denom) < EPS);
            D[i - 1] = (c1 * D[i - 1] + c2 * D[i] + (F[i] - F[i - 1])) / denom;
            F[i - 1] = (c1 * F[i - 1] + c2 * F[i]) / denom;
        }
        // a lane whose table ends with this column keeps its result; later columns only touch dead entries
        f = j == n2 - 1 ? F[0] : f;
        df = j == n2 - 1 ? D[0] : df;
#pragma unroll
        for (int i = 1; i <= EXTENT; ++i) XB[i] = XB[i + 1];
    }
}

DEVFN bool hrmint_axis(const double (&XS)[SAMPLES], int ns, double x_eval, const double *py, const double *pv, const View &v,
                       int64_t first_idx, double &f, double &df) {
    const double EPS = 2.220446049250313e-16;
    double F[2 * SAMPLES], D[2 * SAMPLES], XB[2 * SAMPLES];
    bool bad = false;
    const int n2 = 2 * ns;
    // first column: values and derivatives interleaved (rows beyond ns repeat the last state: dead entries)
#pragma unroll
    for (int k = 0; k < SAMPLES; ++k) { F[2 * k] = 0.0; F[2 * k + 1] = 0.0; }
    if (ns > 0) {
#pragma unroll
        for (int k = 0; k < SAMPLES; ++k) {
            const int64_t at = v.at(first_idx + (k < ns ? k : ns - 1));
            F[2 * k] = py[at];
            F[2 * k + 1] = pv[at];
        }
    }
    // second column: first-degree interpolants
#pragma unroll
    for (int i = 1; i <= SAMPLES - 1; ++i) {
        const bool valid = i <= ns - 1;
        const double xa = XS[i - 1], xb = XS[i];
        const double c1 = xb - x_eval;
        const double c2 = x_eval - xa;
        const double denom = xb - xa;
        bad = bad || (valid && fabs(denom) < EPS);
        const double wp = F[2 * i - 2], wc = F[2 * i - 1], wn = F[2 * i];
        D[2 * i - 2] = wc;
        D[2 * i - 1] = (wn - wp) / denom;
        const double temp = wc * (x_eval - xa) + wp;
        F[2 * i - 1] = valid ? (c1 * wp + c2 * wn) / denom : wc;
        F[2 * i - 2] = valid ? temp : wp;
    }
    D[2 * SAMPLES - 2] = 0.0; D[2 * SAMPLES - 1] = 0.0;
#pragma unroll
    for (int k = 0; k < SAMPLES; ++k) {
        const bool last = k == ns - 1;
        D[2 * k] = last ? F[2 * k + 1] : D[2 * k];
        F[2 * k] = last ? F[2 * k + 1] * (x_eval - XS[k]) + F[2 * k] : F[2 * k];
    }
    // columns 3 .. 2n
    XB[0] = 0.0; XB[2 * SAMPLES - 1] = 0.0;
#pragma unroll
    for (int i = 1; i <= 2 * SAMPLES - 2; ++i) XB[i] = XS[(i + 3) / 2 - 1];
    f = F[0];   // n = 1: the table is complete already
    df = D[0];
    hrmint_columns<24>(2, n2, x_eval, XS, F, D, XB, bad, f, df);
    hrmint_columns<18>(8, n2, x_eval, XS, F, D, XB, bad, f, df);
    hrmint_columns<12>(14, n2, x_eval, XS, F, D, XB, bad, f, df);
    hrmint_columns<6>(20, n2, x_eval, XS, F, D, XB, bad, f, df);
    return !bad;
}

// `Traj::at` for the trajectory of this lane.
// `ill` (optional): set when the window of an interpolated sample holds two states closer than 1e-4 of its mean spacing (see
// NYX_HIP_INTERP_ILL_CONDITIONED); the sample itself is the reference's either way.
DEVFN int traj_at(const nyx_hip_traj_t &src, const View &v, int64_t epoch_ns, double s6[6], bool *ill = nullptr) {
    if (ill) *ill = false;
    const double qnan = __builtin_nan("");
    for (int c = 0; c < 6; ++c) s6[c] = qnan;
    int st = NYX_HIP_INTERP_OK;
    int64_t hit = -1, first_idx = 0;
    int ns = 0;
    if (v.len == 0 || v.epoch[v.at(0)] > epoch_ns || v.epoch[v.at(v.len - 1)] < epoch_ns) {
        st = NYX_HIP_INTERP_NO_DATA;
    } else {
        // binary search (traj.rs:88-91): exact hit, or the insertion index
        int64_t lo = 0, hi = v.len;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            const int64_t e = v.epoch[v.at(mid)];
            if (e == epoch_ns) { hit = mid; break; }
            if (e < epoch_ns) lo = mid + 1; else hi = mid;
        }
        if (hit < 0) {
            const int64_t idx = lo;
            if (idx == 0 || idx >= v.len) {
                st = NYX_HIP_INTERP_NO_DATA;
            } else {
                const int64_t num_left = SAMPLES / 2;
                first_idx = idx > num_left ? idx - num_left : 0;
                const int64_t last_idx = v.len < first_idx + SAMPLES ? v.len : first_idx + SAMPLES;
                if (last_idx == v.len) first_idx = last_idx > 2 * num_left ? last_idx - 2 * num_left : 0;  // 12 states, sic
                ns = (int)(last_idx - first_idx);
            }
        }
    }
    if (hit >= 0) {
        const int64_t at = v.at(hit);
        s6[0] = src.x_km[at]; s6[1] = src.y_km[at]; s6[2] = src.z_km[at];
        s6[3] = src.vx_km_s[at]; s6[4] = src.vy_km_s[at]; s6[5] = src.vz_km_s[at];
    }
    // the interpolation proper: lanes without a window carry ns = 0 and fall through every predicate
    if (__any(ns > 0)) {
        double XS[SAMPLES];
#pragma unroll
        for (int k = 0; k < SAMPLES; ++k) XS[k] = k < ns ? ns_to_seconds(v.epoch[v.at(first_idx + k)]) : 0.0;
        const double x_eval = ns_to_seconds(epoch_ns);
        if (ill && ns > 1) {
            double dmin = fabs(XS[1] - XS[0]);
#pragma unroll
            for (int k = 2; k < SAMPLES; ++k)
                if (k < ns) dmin = fmin(dmin, fabs(XS[k] - XS[k - 1]));
            *ill = dmin < 1e-4 * (fabs(XS[ns - 1] - XS[0]) / (double)(ns - 1));
        }
        bool ok = true;
        double fx = qnan, fy = qnan, fz = qnan, dx = qnan, dy = qnan, dz = qnan;
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            const double *py = c == 0 ? src.x_km : (c == 1 ? src.y_km : src.z_km);
            const double *pv = c == 0 ? src.vx_km_s : (c == 1 ? src.vy_km_s : src.vz_km_s);
            double f, df;
            ok = hrmint_axis(XS, ns, x_eval, py, pv, v, first_idx, f, df) && ok;
            if (c == 0) { fx = f; dx = df; } else if (c == 1) { fy = f; dy = df; } else { fz = f; dz = df; }
        }
        if (ns > 0) {
            if (ok) { s6[0] = fx; s6[1] = fy; s6[2] = fz; s6[3] = dx; s6[4] = dy; s6[5] = dz; }
            else st = NYX_HIP_INTERP_MATH;
        }
    }
    return st;
}

DEVFN void store_sample(const nyx_hip_traj_t &dst, int64_t at, int64_t epoch_ns, const double s6[6]) {
    dst.epoch_ns[at] = epoch_ns;
    dst.x_km[at] = s6[0]; dst.y_km[at] = s6[1]; dst.z_km[at] = s6[2];
    dst.vx_km_s[at] = s6[3]; dst.vy_km_s[at] = s6[4]; dst.vz_km_s[at] = s6[5];
}

}  // namespace

// dst.len[i]: AT -> 0 (the evaluation kernel counts the OK samples); EVERY -> the length of the inclusive time series
// (the evaluation kernel lowers it to the first failing sample, traj_it.rs:41-60).
__global__ __launch_bounds__(256) void nyx_traj_init_kernel(TrajEvalArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    int32_t len = 0;
    if (a.mode == TRAJ_MODE_EVERY) {
        const View v = make_view(a.src, a.n, i);
        if (v.len > 0) {
            const int64_t span = v.epoch[v.at(v.len - 1)] - v.epoch[v.at(0)];
            const int64_t count = span / a.step_ns + 1;
            len = count > INT32_MAX ? INT32_MAX : (int32_t)count;
        }
    }
    a.dst.len[i] = len;
}

__global__ __launch_bounds__(LANES) void nyx_traj_eval_kernel(TrajEvalArgs a) {
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * LANES + lane;
    const bool live = i < a.n;
    const int64_t ii = live ? i : a.n - 1;  // idle lanes shadow a valid trajectory and store nothing
    const View v = make_view(a.src, a.n, ii);
    const int64_t q0 = (int64_t)blockIdx.y * a.samples_per_block;
    int64_t q_end;
    int64_t first = 0;
    if (a.mode == TRAJ_MODE_EVERY) {
        int64_t count = 0;
        if (v.len > 0) {
            first = v.epoch[v.at(0)];
            count = (v.epoch[v.at(v.len - 1)] - first) / a.step_ns + 1;
        }
        q_end = count < a.dst.capacity ? count : a.dst.capacity;
    } else {
        q_end = a.m;
    }
    if (q_end > q0 + a.samples_per_block) q_end = q0 + a.samples_per_block;
    int32_t n_ok = 0;
    for (int64_t q = q0; __any(q < q_end); ++q) {
        const bool mine = live && q < q_end;
        const int64_t epoch = a.mode == TRAJ_MODE_EVERY ? first + q * a.step_ns : a.query[q < a.m ? q : a.m - 1];
        double s6[6];
        bool ill = false;
        int st = traj_at(a.src, v, epoch, s6, a.mode == TRAJ_MODE_AT ? &ill : nullptr);
        if (st == NYX_HIP_INTERP_OK && ill) st = NYX_HIP_INTERP_ILL_CONDITIONED;
        if (!mine) continue;
        const int64_t at = q * a.n + i;
        if (a.mode == TRAJ_MODE_EVERY) {
            if (st == NYX_HIP_INTERP_OK) store_sample(a.dst, at, epoch, s6);
            else atomicMin(&a.dst.len[i], (int32_t)q);
        } else {
            store_sample(a.dst, at, epoch, s6);
            a.status[at] = st;
            n_ok += st == NYX_HIP_INTERP_OK || st == NYX_HIP_INTERP_ILL_CONDITIONED;
        }
    }
    if (a.mode == TRAJ_MODE_AT && live && n_ok) atomicAdd(&a.dst.len[i], n_ok);
}

// ---------------------------------------------------------------------------------------------
// Event search (propagators/event.rs:178-197).  brent_solver is anise's (absent crate): restated from the `roots`
// crate's Brent as earlier Nyx releases embedded it; the oracle carries an independent copy of the same restatement.
// ---------------------------------------------------------------------------------------------
namespace {

DEVFN int ev_at(const nyx_hip_traj_t &traj, const View &v, const nyx_hip_event_t &ev, double mu, int64_t epoch_ns, double &value) {
    double s6[6];
    const int st = traj_at(traj, v, epoch_ns, s6);
    value = ev_eval(ev, mu, epoch_ns, s6);
    return st;
}

// 0 = found, 1 = not in the bracket, 2 = evaluation failed, 3 = iteration cap
DEVFN int brent_event(const nyx_hip_traj_t &traj, const View &v, const nyx_hip_event_t &ev, double mu, int64_t start_ns, int64_t end_ns,
                      int64_t &event_ns) {
    const double EPS = 2.220446049250313e-16;
    const double eps_t = ns_to_seconds(ev.epoch_precision_ns);
    const double eps_v = fabs(ev.value_precision);
    double xa = 0.0, xb = ns_to_seconds(end_ns - start_ns);
    double ya, yb;
    if (ev_at(traj, v, ev, mu, start_ns, ya) | ev_at(traj, v, ev, mu, end_ns, yb)) return 2;
    if (fabs(ya) <= eps_v) { event_ns = start_ns; return 0; }
    if (fabs(yb) <= eps_v) { event_ns = end_ns; return 0; }
    double xc = xa, yc = ya, xd = xa;
    bool flag = true;
    for (int it = 0; it < 50; ++it) {
        if (fabs(ya) < eps_v) { event_ns = start_ns + seconds_to_ns(xa); return 0; }
        if (fabs(yb) < eps_v) { event_ns = start_ns + seconds_to_ns(xb); return 0; }
        if (fabs(xa - xb) <= eps_t) return 1;
        double sx;
        if (fabs(ya - yc) > EPS && fabs(yb - yc) > EPS)
            sx = xa * yb * yc / ((ya - yb) * (ya - yc)) + xb * ya * yc / ((yb - ya) * (yb - yc)) + xc * ya * yb / ((yc - ya) * (yc - yb));
        else
            sx = xb - yb * (xb - xa) / (yb - ya);
        const bool cond1 = (sx - xb) * (sx - (3.0 * xa + xb) / 4.0) > 0.0;
        const bool cond2 = flag && fabs(sx - xb) >= fabs(xb - xc) / 2.0;
        const bool cond3 = !flag && fabs(sx - xb) >= fabs(xc - xd) / 2.0;
        const bool cond4 = flag && fabs(xb - xc) <= eps_t;
        const bool cond5 = !flag && fabs(xc - xd) <= eps_t;
        if (cond1 || cond2 || cond3 || cond4 || cond5) { sx = (xa + xb) / 2.0; flag = true; } else { flag = false; }
        double ys;
        if (ev_at(traj, v, ev, mu, start_ns + seconds_to_ns(sx), ys)) return 2;
        xd = xc; xc = xb; yc = yb;
        if (ya * ys < 0.0) {  // root between a and s
            if (fabs(ya) > fabs(ys)) { xb = sx; yb = ys; } else { xb = xa; yb = ya; xa = sx; ya = ys; }
        } else {              // root between s and b
            if (fabs(ys) > fabs(yb)) { xa = sx; ya = ys; } else { xa = xb; ya = yb; xb = sx; yb = ys; }
        }
    }
    return 3;
}

}  // namespace

__global__ __launch_bounds__(LANES) void nyx_event_search_kernel(EventSearchArgs a) {
    const int64_t i = (int64_t)blockIdx.x * LANES + threadIdx.x;
    if (i >= a.n) return;
    if (a.status[i] != NYX_HIP_OK) return;  // propagation error: reported as it is
    if (!a.found[i]) {                      // end_state == last published state (event.rs:170-176)
        a.status[i] = NYX_HIP_ERR_EVENT_NOT_FOUND;
        return;
    }
    const int64_t len = a.traj.len[i];
    if (len >= a.traj.capacity || len < 1) {  // the bracket does not fit the caller's buffer
        a.status[i] = NYX_HIP_ERR_EVENT_SEARCH;
        return;
    }
    // traj.states.push(end_state) (event.rs:179)
    const int64_t at = len * a.n + i;
    const int64_t end_ns = a.epoch_ns[i];
    a.traj.epoch_ns[at] = end_ns;
    a.traj.x_km[at] = a.state[0][i]; a.traj.y_km[at] = a.state[1][i]; a.traj.z_km[at] = a.state[2][i];
    a.traj.vx_km_s[at] = a.state[3][i]; a.traj.vy_km_s[at] = a.state[4][i]; a.traj.vz_km_s[at] = a.state[5][i];
    a.traj.len[i] = (int32_t)(len + 1);
    __threadfence();
    View v;
    v.epoch = a.traj.epoch_ns; v.n = a.n; v.i = i; v.len = len + 1;
    v.desc = end_ns < a.traj.epoch_ns[i];
    // `traj.states.last()` AFTER finalize() sorted the states by epoch (event.rs:165-168): the last published state when
    // propagating forward, but the START state of a back-propagation (the bracket is then the whole arc)
    const int64_t start_ns = v.desc ? a.traj.epoch_ns[i] : a.traj.epoch_ns[(len - 1) * a.n + i];
    int64_t ev_ns = 0;
    double s6[6];
    if (brent_event(a.traj, v, a.ev, a.mu, start_ns, end_ns, ev_ns) != 0 || traj_at(a.traj, v, ev_ns, s6) != NYX_HIP_INTERP_OK) {
        a.status[i] = NYX_HIP_ERR_EVENT_SEARCH;
        return;
    }
    a.epoch_ns[i] = ev_ns;
#pragma unroll
    for (int c = 0; c < 6; ++c) a.state[c][i] = s6[c];
}

extern "C" hipError_t nyx_launch_event_search(const EventSearchArgs *args, hipStream_t stream) {
    if (args->n <= 0) return hipSuccess;
    hipLaunchKernelGGL(nyx_event_search_kernel, dim3((unsigned)((args->n + LANES - 1) / LANES)), dim3(LANES), 0, stream, *args);
    return hipGetLastError();
}

extern "C" hipError_t nyx_launch_traj_eval(const TrajEvalArgs *args, hipStream_t stream) {
    TrajEvalArgs a = *args;
    if (a.n <= 0) return hipSuccess;
    const int64_t span = a.mode == TRAJ_MODE_EVERY ? a.dst.capacity : a.m;
    hipLaunchKernelGGL(nyx_traj_init_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, stream, a);
    if (span <= 0) return hipGetLastError();
    // grid.y <= 32768 chunks of consecutive samples
    int64_t spb = 16;
    if ((span + spb - 1) / spb > 32768) spb = (span + 32767) / 32768;
    a.samples_per_block = spb;
    const dim3 grid((unsigned)((a.n + LANES - 1) / LANES), (unsigned)((span + spb - 1) / spb));
    hipLaunchKernelGGL(nyx_traj_eval_kernel, grid, dim3(LANES), 0, stream, a);
    return hipGetLastError();
}
