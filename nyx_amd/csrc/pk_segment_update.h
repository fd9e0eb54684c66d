// pk_segment_update.h - part of propagate_kernel.hip (included there, in this order; not a stand-alone header): the covariance-mapping loop in one launch: the Kalman time updates of a segment boundary.
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

