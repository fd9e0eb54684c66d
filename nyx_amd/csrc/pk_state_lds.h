// pk_state_lds.h - part of propagate_kernel.hip (included there, in this order; not a stand-alone header): event step, the integrator's cold state, the LDS carve and its size, the textbook STM replay.
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
    if (!stm) {
        m.ys2 = p; p += 6 * DEV_LANES;
        m.inb2 = p; p += NIN * DEV_LANES;
        m.pert2 = p; p += 9 * DEV_LANES;
        m.ixs = p; p += 4 * DEV_LANES;
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
    d += quad ? (size_t)(10 + 15 + 6 + QPRE_ROWS + 6 + 10 + 15) * DEV_LANES : (stm ? (size_t)(20 + 27 + 12) * DEV_LANES : (size_t)(NIN + 9 + 6 + NIN + 9 + 4) * DEV_LANES);
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


