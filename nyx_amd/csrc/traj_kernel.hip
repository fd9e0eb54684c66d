// traj_kernel.hip — batched `Traj::at` / `Traj::every` on the MI355X (gfx950).
//
// Reference: md/trajectory/traj.rs:82-162 (window selection, exact hits, bounds), interpolatable.rs:52-108 (one
// Hermite interpolation per axis over at most 13 stored states, abscissas in f64 seconds) and anise's
// `hermite_eval` = SPICE HRMINT (divided-difference table with doubled abscissas).
//
// Mapping: lane <-> trajectory, as in the propagation kernel, so that the step-major dense output
// ([k * n + i]) is read and the sample-major result ([q * n + i]) is written fully coalesced.  A workgroup is ONE
// wave that owns 64 trajectories x a chunk of consecutive samples (grid.y walks the chunks).  HRMINT's table is
// indexed by loop counters, so it lives in LDS, field-major ([slot * 64 + lane]: conflict-free ds_read_b64):
// 13 abscissas + 52 table entries = 33 KB per wave, 4 waves per CU.  The x, y and z tables are built one after
// the other in the same LDS slots.
//
// Bound: FP64 VALU.  One sample costs 3 axes x (n(2n-1) table updates) x 2 IEEE divisions ~ 1 800 divisions
// (n = 13), against 7 x 13 x 8 B = 728 B of (cached, overlapping) reads and 56 B written: ~50 FLOP/B, far
// to the right of the HBM ridge.  The divisions are the reference's (each table entry is divided by its own
// abscissa difference); they are kept so that the device result equals the CPU restatement bit for bit.
// Compiled with -ffp-contract=off for the same reason.

#include <hip/hip_runtime.h>

#include "../../include/nyx_hip.h"
#include "hifitime_dev.h"
#include "traj_args.h"

#define DEVFN static __device__ __forceinline__

namespace {

constexpr int LANES = 64;
constexpr int SAMPLES = 13;                 // INTERPOLATION_SAMPLES, interpolatable.rs:22
constexpr int LDS_SLOTS = SAMPLES + 4 * SAMPLES;  // xs + HRMINT work array (2 columns of 2n)

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

// HRMINT on the LDS-resident table of this lane.  xs = lds[0..13), work = lds[13..65); returns false on
// |denominator| < f64::EPSILON (InterpMath, DivisionByZero).  Fortran (1-based) indices, as the routine is published.
DEVFN bool hrmint(double *lds, int lane, int ns, double x_eval, double &f, double &df) {
#define XS(k) lds[(k) * LANES + lane]
#define WK(k) lds[(SAMPLES + (k)) * LANES + lane]
    const double EPS = 2.220446049250313e-16;
    bool ok = true;
    const int n2 = 2 * ns;
    for (int i = 1; i <= ns - 1; ++i) {
        const double xa = XS(i - 1), xb = XS(i);
        const double c1 = xb - x_eval;
        const double c2 = x_eval - xa;
        const double denom = xb - xa;
        ok = ok && !(fabs(denom) < EPS);
        const int prev = 2 * i - 1, cur = prev + 1, next = cur + 1;
        const double wp = WK(prev - 1), wc = WK(cur - 1), wn = WK(next - 1);
        WK(prev + n2 - 1) = wc;
        WK(cur + n2 - 1) = (wn - wp) / denom;
        const double temp = wc * (x_eval - xa) + wp;
        WK(cur - 1) = (c1 * wp + c2 * wn) / denom;
        WK(prev - 1) = temp;
    }
    {
        const double wl = WK(n2 - 1);
        WK(2 * n2 - 2) = wl;
        WK(n2 - 2) = wl * (x_eval - XS(ns - 1)) + WK(n2 - 2);
    }
    for (int j = 2; j <= n2 - 1; ++j) {
        double w_lo = WK(0);  // work[i-1] of the first entry; afterwards carried from the previous iteration's work[i]
        for (int i = 1; i <= n2 - j; ++i) {
            const int xi = (i + 1) >> 1;
            const int xij = (i + j + 1) >> 1;
            const double xa = XS(xi - 1), xb = XS(xij - 1);
            const double c1 = xb - x_eval;
            const double c2 = x_eval - xa;
            const double denom = xb - xa;
            ok = ok && !(fabs(denom) < EPS);
            const double w_hi = WK(i);
            WK(i + n2 - 1) = (c1 * WK(i + n2 - 1) + c2 * WK(i + n2) + (w_hi - w_lo)) / denom;
            WK(i - 1) = (c1 * w_lo + c2 * w_hi) / denom;
            w_lo = w_hi;  // the next entry reads the not-yet-overwritten work[i]
        }
    }
    f = WK(0);
    df = WK(n2);
    return ok;
#undef XS
#undef WK
}

// `Traj::at` for the trajectory of this lane.  All lanes of the wave call it together (it contains no barrier, but
// keeping the lanes converged keeps the LDS table accesses conflict-free).
DEVFN int traj_at(const nyx_hip_traj_t &src, const View &v, int64_t epoch_ns, double *lds, int lane, double s6[6]) {
    const double *comp[6] = {src.x_km, src.y_km, src.z_km, src.vx_km_s, src.vy_km_s, src.vz_km_s};
    const double qnan = __builtin_nan("");
    for (int c = 0; c < 6; ++c) s6[c] = qnan;
    if (v.len == 0) return NYX_HIP_INTERP_NO_DATA;
    if (v.epoch[v.at(0)] > epoch_ns || v.epoch[v.at(v.len - 1)] < epoch_ns) return NYX_HIP_INTERP_NO_DATA;
    // binary search (traj.rs:88-91): exact hit, or the insertion index
    int64_t lo = 0, hi = v.len, hit = -1;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        const int64_t e = v.epoch[v.at(mid)];
        if (e == epoch_ns) { hit = mid; break; }
        if (e < epoch_ns) lo = mid + 1; else hi = mid;
    }
    if (hit >= 0) {
        const int64_t at = v.at(hit);
        for (int c = 0; c < 6; ++c) s6[c] = comp[c][at];
        return NYX_HIP_INTERP_OK;
    }
    const int64_t idx = lo;
    if (idx == 0 || idx >= v.len) return NYX_HIP_INTERP_NO_DATA;
    const int64_t num_left = SAMPLES / 2;
    int64_t first_idx = idx > num_left ? idx - num_left : 0;
    const int64_t last_idx = v.len < first_idx + SAMPLES ? v.len : first_idx + SAMPLES;
    if (last_idx == v.len) first_idx = last_idx > 2 * num_left ? last_idx - 2 * num_left : 0;  // 12 states, sic
    const int ns = (int)(last_idx - first_idx);
    for (int k = 0; k < ns; ++k) lds[k * LANES + lane] = ns_to_seconds(v.epoch[v.at(first_idx + k)]);
    const double x_eval = ns_to_seconds(epoch_ns);
    bool ok = true;
    double f[3], df[3];
    for (int c = 0; c < 3; ++c) {
        for (int k = 0; k < ns; ++k) {
            const int64_t at = v.at(first_idx + k);
            lds[(SAMPLES + 2 * k) * LANES + lane] = comp[c][at];
            lds[(SAMPLES + 2 * k + 1) * LANES + lane] = comp[c + 3][at];
        }
        ok = hrmint(lds, lane, ns, x_eval, f[c], df[c]) && ok;
    }
    if (!ok) return NYX_HIP_INTERP_MATH;
    for (int c = 0; c < 3; ++c) { s6[c] = f[c]; s6[c + 3] = df[c]; }
    return NYX_HIP_INTERP_OK;
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
    __shared__ double lds[LDS_SLOTS * LANES];
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
        const int st = traj_at(a.src, v, epoch, lds, lane, s6);
        if (!mine) continue;
        const int64_t at = q * a.n + i;
        if (a.mode == TRAJ_MODE_EVERY) {
            if (st == NYX_HIP_INTERP_OK) store_sample(a.dst, at, epoch, s6);
            else atomicMin(&a.dst.len[i], (int32_t)q);
        } else {
            store_sample(a.dst, at, epoch, s6);
            a.status[at] = st;
            n_ok += st == NYX_HIP_INTERP_OK;
        }
    }
    if (a.mode == TRAJ_MODE_AT && live && n_ok) atomicAdd(&a.dst.len[i], n_ok);
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
