// pk_integrator_ool.h - part of propagate_kernel.hip (included there, in this order; not a stand-alone header): the integrator wave of the sixteen-wave plain kernels out of line: integ_front / integ_sums / integ_back / integ_step (and the fan-out mode's sums wave).
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
// stored v_i (ctl[4]) - and leaves the six values in the drag rows of the two perturbation buffers, which the integrator reads behind the stage barrier, in front of the
// next integ_front.  The same additions in the same order as the inline sums: bit-identical results.  Every spin is bounded; a wait
// that expires leaves NaNs, which end the step as NYX_HIP_ERR_NAN.
static __device__ __attribute__((noinline)) void fan_sums(uint32_t lds_v, uint64_t cfg_u, int i_v, int lane, int flags_v, int kdone_v) {
    IX_PROLOGUE
    (void)kbuf; (void)kb_li; (void)KB_STR;
    const bool spec = cfg->spec != 0;
    const LdsPtr kb0 = ix_rows(L.kbuf, lane);
    const LdsPtr ysb = ix_rows((i & 1) ? L.ys2 : L.ys, lane);
    // (no LDS of its own: W in rows 6..8 of the even stages' perturbation buffer, P in those of the odd stages' - the drag rows, which
    //  nothing touches in a configuration without drag; the host names a sums wave only then)
    const LdsPtr out_w = ix_rows(L.pert + 6 * DEV_LANES, lane), out_p = ix_rows(L.pert2 + 6 * DEV_LANES, lane);
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
    for (int e = 0; e < 3; ++e) { out_w[e * DEV_LANES] = w[e]; out_p[e * DEV_LANES] = p[e]; }
}
#endif
#endif  // INTEG_OOL

