// predict_kernel.hip — the Kalman time update between two covariance-mapping segments, on the device.
//
// Reference: KalmanODProcess::predict_until (od/process/mod.rs:440-486) calls, per segment,
// `prop_instance.for_duration(max_step)` (the STM kernel variant of propagate_kernel.hip), then
// KalmanFilter::time_update (od/kalman/filtering.rs:59-99): covar_bar = stm * covar * stm^T, plus Gamma Q Gamma^T of
// the last applicable ProcessNoise (od/snc.rs:165-283), state_bar = stm * state_deviation, then reset_stm().
// Here the segment launches and these updates are enqueued back to back on one stream: the filter state (covariance,
// deviation, previous epoch, per-trajectory activity) never leaves HBM and the host is not in the loop.
//
// One workgroup per trajectory, one thread per matrix element (81 of 128 lanes busy); Phi, P and Phi*P sit in LDS.
// 2 x 729 multiply-adds per update: the cost is the launch, not the arithmetic.  Products are accumulated in
// nalgebra's order (k ascending, multiply then add; compiled with -ffp-contract=off) so that the oracle agrees bit for
// bit given the same Phi.

#include <hip/hip_runtime.h>

#include "hifitime_dev.h"
#include "predict_args.h"

__global__ __launch_bounds__(256) void nyx_predict_init_kernel(PredictArgs a, const int64_t *epoch0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    for (int k = 0; k < 81; ++k) a.stm[i * 81 + k] = (k % 10 == 0) ? 1.0 : 0.0;  // with_stm(): identity
    a.prev_epoch[i] = epoch0[i];
    a.init_epoch[i] = epoch0[i];
    a.dur[i] = a.cfg.max_step_ns;  // the loop body always runs once (mod.rs:465-484)
    a.status[i] = 0;
    a.acc_n_acc[i] = 0; a.acc_n_rej[i] = 0; a.acc_n_evals[i] = 0;
    a.hist.n_updates[i] = 0;
}

__global__ __launch_bounds__(128) void nyx_time_update_kernel(PredictArgs a) {
    __shared__ double phi[81], p[81], m[81], dev[9], snc[3];
    __shared__ int snc_q;
    const int64_t i = blockIdx.x;
    const int t = threadIdx.x;
    if (a.dur[i] == 0) return;  // finished or failed earlier: nothing was propagated for this trajectory
    const int st = a.seg_status[i];
    if (t == 0) {
        a.acc_n_acc[i] += a.seg_n_acc[i];
        a.acc_n_rej[i] += a.seg_n_rej[i];
        a.acc_n_evals[i] += a.seg_n_evals[i];
    }
    if (st != 0) {  // the reference returns the propagation error: no estimate for this segment, the run ends
        if (t == 0) { a.status[i] = st; a.dur[i] = 0; }
        return;
    }
    // everything thread 0 rewrites at the end is read before the first barrier
    const int64_t epoch = a.epoch[i];
    const int64_t delta_ns = epoch - a.prev_epoch[i];
    const int32_t u = a.hist.n_updates[i];
    if (t < 81) { phi[t] = a.stm[i * 81 + t]; p[t] = a.covar[i * 81 + t]; }
    if (t < 9) dev[t] = a.state_dev ? a.state_dev[i * 9 + t] : 0.0;
    if (t == 0) {
        // the process noise that applies: last applicable entry (filtering.rs:64-80), its diagonal at this epoch
        // (ProcessNoise::to_matrix, snc.rs:165-205: exponential decay since init_epoch) expressed in the state frame
        // (ProcessNoise::propagate, snc.rs:219-239: new = dcm * snc * dcm^T at the nominal orbit, DIAGONAL kept)
        int pick = -1;
        for (int q = a.cfg.n_process_noise - 1; q >= 0; --q) {
            const nyx_hip_process_noise_t &pn = a.cfg.process_noise[q];
            if (pn.has_start_time && pn.start_time_ns > epoch) continue;  // snc.rs:168-175
            if (delta_ns > pn.disable_time_ns) continue;                  // snc.rs:178-186, 248-250
            pick = q;
            break;
        }
        snc_q = pick;
        if (pick >= 0) {
            const nyx_hip_process_noise_t &pn = a.cfg.process_noise[pick];
            double d[3] = {pn.diag[0], pn.diag[1], pn.diag[2]};
            if (pn.has_decay) {
                const int64_t init = pn.init_epoch_ns != INT64_MIN ? pn.init_epoch_ns : a.init_epoch[i];
                const double total = ns_to_seconds(epoch - init);
                for (int k = 0; k < 3; ++k) d[k] = d[k] * exp(-pn.decay_s[k] * total);
            }
            if (pn.local_frame != NYX_HIP_FRAME_INERTIAL) {
                // dcm_to_inertial(local_frame) of the nominal orbit (anise, absent: classical definitions).  Columns:
                // RIC = [r^, c^ x r^, c^], VNC = [v^, n^, v^ x n^] with c^ = n^ = (r x v)^
                const double r[3] = {a.s9[0][i], a.s9[1][i], a.s9[2][i]}, v[3] = {a.s9[3][i], a.s9[4][i], a.s9[5][i]};
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
    __syncthreads();
    const int r = t % 9, c = t / 9;  // column-major: element (r, c) at c * 9 + r
    if (t < 81) {
        double acc = phi[r] * p[c * 9];  // k = 0
        for (int k = 1; k < 9; ++k) acc = acc + phi[k * 9 + r] * p[c * 9 + k];
        m[t] = acc;
    }
    __syncthreads();
    const bool keep = u < a.hist.capacity;
    const int64_t slot = (int64_t)u * a.n + i;
    if (t < 81) {
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
        a.stm[i * 81 + t] = (r == c) ? 1.0 : 0.0;  // reset_stm() (mod.rs:479)
    }
    if (t < 9) {
        double sb = 0.0;
        if (a.cfg.deviation_tracking) {
            sb = phi[t] * dev[0];
            for (int k = 1; k < 9; ++k) sb = sb + phi[k * 9 + t] * dev[k];
        }
        if (a.state_dev) a.state_dev[i * 9 + t] = sb;
        if (keep && a.hist.state_dev) a.hist.state_dev[slot * 9 + t] = sb;
        if (keep && a.hist.state) a.hist.state[slot * 9 + t] = a.s9[t][i];
    }
    if (t == 0) {
        if (keep && a.hist.epoch_ns) a.hist.epoch_ns[slot] = epoch;
        a.hist.n_updates[i] = u + 1;
        a.prev_epoch[i] = epoch;
        a.dur[i] = epoch >= a.cfg.end_epoch_ns ? 0 : a.cfg.max_step_ns;  // mod.rs:480-482
    }
}

extern "C" hipError_t nyx_launch_predict_init(const PredictArgs *a, const int64_t *epoch0, hipStream_t stream) {
    hipLaunchKernelGGL(nyx_predict_init_kernel, dim3((unsigned)((a->n + 255) / 256)), dim3(256), 0, stream, *a, epoch0);
    return hipGetLastError();
}

extern "C" hipError_t nyx_launch_time_update(const PredictArgs *a, hipStream_t stream) {
    hipLaunchKernelGGL(nyx_time_update_kernel, dim3((unsigned)a->n), dim3(128), 0, stream, *a);
    return hipGetLastError();
}
