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

#include "pk_epoch_data.h"
#include "pk_force_models.h"
#include "pk_harmonics.h"
#include "pk_cooperative.h"
#include "pk_error_stm.h"
#include "pk_state_lds.h"
#include "pk_integrator_ool.h"
#include "pk_segment_update.h"
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
#if defined(NYX_COOP_FAN) && FAN_SKIP
    const bool rows_primary = cfg->sched[DEV_SCHED_PRIMARY].n_ranges[wave] > 0, rows_solo = cfg->sched[DEV_SCHED_SOLO].n_ranges[wave] > 0;  // (uniform)
#endif

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
                    for (int e = 0; e < 3; ++e) { wpre[3 + e] = L.pert[(6 + e) * DEV_LANES + lane]; pre_wr[e] = L.pert2[(6 + e) * DEV_LANES + lane]; }
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
#if defined(NYX_COOP_FAN) && FAN_SKIP
                // (fan-out mode: the owner's waves hold next to no columns, and the walk of a wave WITHOUT columns - five LDS reads, a call,
                //  the schedule lookup through the scalar cache - is ~2 k cycles per stage on the almanac wave, which bounds the period there)
                if (!(pipe && INTEG) && (sched == DEV_SCHED_PRIMARY ? rows_primary : rows_solo)) {
#else
                if (!(pipe && INTEG)) {
#endif
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
