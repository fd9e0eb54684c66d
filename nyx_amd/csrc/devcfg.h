// devcfg.h — flattened, device-resident description of one propagation context.
// Built once by nyx_hip_ctx_create (abi.cpp) from nyx_hip_config_t; read by every wave through
// scalar loads (all fields are wave-uniform).
#ifndef NYX_AMD_DEVCFG_H
#define NYX_AMD_DEVCFG_H

#include <stdint.h>

#include "../../include/nyx_hip.h"

#define DEV_MAX_STAGES 16
#define DEV_MAX_SLOTS 4   /* non-central bodies whose position is evaluated per stage */
#define DEV_MAX_SEG 8
#define DEV_MAX_WAVES 16  /* waves per 64-trajectory workgroup (column split) */
#define DEV_MAX_WAVES_STM 4 /* STM variant: dual numbers need 256 VGPRs per wave and 16 partial slots per wave */
#define DEV_MAX_RANGES 6  /* contiguous column ranges per wave */
#define DEV_LANES 64
#define DEV_MAX_ALM 5     /* almanac waves of a workgroup (role fan-out) */
/* Roles of the waves of a workgroup, dealt by the host (DevCfg.role_kind / role_mask / role_slot).  role_mask: low 16 bits =
 * almanac share (bit s = body slot s - or distinct segment s, DevCfg.seg_mode -, DEV_ROLE_DCM = the body-fixed DCM), high 16 bits =
 * perturbation share. */
enum { DEV_ROLE_COLUMNS = 0, DEV_ROLE_ALL = 1, DEV_ROLE_INTEG = 2, DEV_ROLE_ALMANAC = 3, DEV_ROLE_PERT = 4, DEV_ROLE_ALMANAC_PERT = 5 };
#define DEV_ROLE_DCM 0x100
#define DEV_ROLE_SUMS 0x200    /* DevCfg.offload: the head of the next-but-one stage's sum_j a_ij k_j (see role_loop) */
#define DEV_ROLE_TWOBODY 0x400 /* DevCfg.offload: the two-body term of the current stage */
#define DEV_ROLE_QPRE 0x800    /* DevCfg.qpre_off (quad STM layout): the position-only pieces of phase C of the current stage (quad_pre) */
#define DEV_PERT_PM 1  /* point masses + solid tides */
#define DEV_PERT_SRP 2 /* solar radiation pressure + drag */

struct DevSeg {
    double init_et, interval, end_et;
    int32_t n_rec, n_coef, stride, offset; /* offset (in doubles) into the records array */
};

struct DevSlot { /* one evaluated body: position w.r.t. the integration centre = sum sign_k * seg_k */
    double mu, radius;
    int32_t n_chain;
    int32_t seg[4];
    double sign[4];
    int32_t useg[4]; /* seg[k] as an index into DevCfg.useg_seg (segment-level almanac units) */
};

/* One assignment of harmonics columns to the waves of a workgroup. */
struct DevSched {
    int32_t n_ranges[DEV_MAX_WAVES];
    int32_t range_c0[DEV_MAX_WAVES][DEV_MAX_RANGES];
    int32_t range_cnt[DEV_MAX_WAVES][DEV_MAX_RANGES];
};
/* SOLO: one workgroup does every column (all configurations).  Cooperative mode (idle CUs lend a hand, see
 * propagate_kernel.hip): PRIMARY = the columns the trajectory-owning workgroup keeps, HELPER = the columns a helper
 * workgroup on another CU evaluates (the owner walks that schedule itself, wave slot by wave slot, if no helper
 * answers). */
#ifndef NYX_FAN_SUMS
#define NYX_FAN_SUMS 0 /* 1: fan-out mode, a column wave of an owner forms the integrator's two stage sums beside it (host: DevCfg.sums_wave1; kernel: fan_sums; its six values live in the drag rows of the perturbation buffers: no LDS of its own).  Bit-identical; measured in round 6 on the final fan-out kernel, 1 250 / 2 500 / 5 000 trajectories x 24 h: 374.0-378.6 / 393.6 / 439.9 ms with it against 369.9-372.0 / 387.8 / 439.4 without - the integrator's window shrinks 13.5 k -> 8.0 k cycles per evaluation and it then waits 3 k for the helpers' answer: the turnaround of a job (a 71-row column on one wave + five uncached hops) bounds the period */
#endif
#define DEV_FAN_MAX 8 /* dedicated helper workgroups per owner in the fan-out mode (DevBatch.coop_fan) */
enum { DEV_SCHED_SOLO = 0, DEV_SCHED_PRIMARY = 1, DEV_SCHED_HELPER = 2, DEV_SCHED_SECOND = 3, DEV_SCHED_HELPER2 = 4, DEV_SCHED_FAN0 = 5, DEV_N_SCHED = 5 + DEV_FAN_MAX };
/* FAN0 + p: the columns of part p of the fan-out mode (small shards: every owner has coop_parts DEDICATED helper workgroups, see helper_body). */
/* SECOND: every column of the second field, for whichever wave walks it.  HELPER2: the second PART of an evaluation's hand-off when the
 * helpers' columns travel as two sub-jobs claimed by two different helper workgroups (DevBatch.coop_parts = 2, see propagate_kernel.hip). */
#define DEV_COOP_PARTS 2

#define DEV_MAX_NUT_PREC 16
struct DevRot { /* nyx_hip_rotation_t, flattened */
    double ra[3], dec[3], w[3];
    int32_t kind, n_np;                          /* NYX_HIP_ROT_*; number of trigonometric terms */
    double np_ang[DEV_MAX_NUT_PREC][2];          /* theta_k = [0] + [1] * T (deg) */
    double np_ra[DEV_MAX_NUT_PREC], np_dec[DEV_MAX_NUT_PREC], np_w[DEV_MAX_NUT_PREC];
    int32_t euler_seg, _pad;                     /* EULER_CHEBY: index into DevCfg.seg */
    double base[9];
};

struct DevCfg {
    /* --- integrator (IntegratorOptions + flattened tableau) --- */
    int32_t stages, order, fixed_step, error_ctrl, attempts, flags;
    double tol;
    int64_t init_step_ns, min_step_ns, max_step_ns;
    double min_step_s, max_step_s;
    double inv_order, inv_order_m1; /* 1/order, 1/(order-1) as the reference evaluates them */
    double a[DEV_MAX_STAGES * (DEV_MAX_STAGES - 1) / 2];
    double b[DEV_MAX_STAGES];
    double bdiff[DEV_MAX_STAGES]; /* b_i - b*_i */
    double c[DEV_MAX_STAGES];     /* running row sums of A, in the reference's order */

    /* --- dynamics --- */
    double mu_central;
    double central_radius;
    int32_t n_slots, n_seg;
    int32_t rec_in_lds, rec_doubles; /* segment records staged in LDS when they fit */
    DevSlot slot[DEV_MAX_SLOTS];
    DevSeg seg[DEV_MAX_SEG];
    int32_t n_pm;
    int32_t pm_slot[DEV_MAX_SLOTS];

    int32_t has_srp, srp_estimate, sun_slot, n_shadow;
    int32_t shadow_slot[DEV_MAX_SLOTS]; /* -1 => the central body */
    double phi, c_m_s;

    int32_t has_grav, deg, ord, n_cols; /* columns 1..n_cols (= deg+1) */
    int32_t g_slot, _pad_g;             /* >= 0: the field belongs to the body of that slot, not to the integration centre (evaluated at r - r_body) */
    double g_mu, g_re, g_inv_re;
    DevRot g_rot;
    /* second gravity field (nyx_hip_config_t.gravity2): walked in one piece by the perturbation wave with the point-mass share, its own
     * table (device addresses: the kernel's table arguments are the first field's), DCM evaluated there */
    int32_t has_grav2, n_cols2, g2_slot, _pad_g2;
    double g2_mu, g2_re, g2_inv_re;
    DevRot g2_rot;
    uint64_t htab2, cols2;

    int32_t has_drag, drag_density;
    double drag_rho0, drag_r0, drag_ref_alt_m, drag_max_alt_m, drag_re;
    DevRot d_rot;

    /* SolidTides (solid_tides.rs): perturber j = slot t_slot[j], GM ratio, degree-3 switch */
    int32_t has_tides, t_n;
    int32_t t_slot[DEV_MAX_SLOTS], t_deg3[DEV_MAX_SLOTS];
    double t_gm_ratio[DEV_MAX_SLOTS];
    double t_k2_5, t_k3_7; /* k2 / 5, k3 / 7 */
    double t_mu, t_re;
    DevRot t_rot;

    /* --- roles of the waves (see DEV_ROLE_*) --- */
    int32_t role_kind[DEV_MAX_WAVES], role_mask[DEV_MAX_WAVES], role_slot[DEV_MAX_WAVES]; /* role_slot: index of an almanac wave's status rows */
    int32_t n_alm;
    /* Segment-level almanac units (role fan-out): the almanac waves evaluate every DISTINCT ephemeris segment once (Earth -> EMB is
     * on the chain of every body of an Earth-centred run) and leave its vector in rows ed_seg_base + 3 u of the epoch data; whoever
     * needs body s sums its chain, sign_k * segment_k in chain order - the same additions epoch_data() makes in slot mode. */
    int32_t seg_mode, n_useg, ed_seg_base;
    int32_t offload; /* pipelined loop without a gravity field: almanac waves with time to spare take the two-body term and the head of
                      * the stage sums off the integrator wave, which is the critical path there (role_mask bits DEV_ROLE_SUMS / _TWOBODY) */
    int32_t useg_seg[DEV_MAX_SEG];
    int32_t spec; /* speculative stage 0 of the next attempt (pipelined loop, see role_loop) */
    int32_t dcm_incr, qpre_off; /* the body-fixed frame of the epoch data is a polynomial IAU orientation: its DCM is advanced from a base
                                * epoch by angle addition (rotation_dcm_iau_poly) instead of three full-range sincos per stage */

    /* --- column schedules: wave w walks n_ranges[w] contiguous column ranges --- */
    int32_t n_waves;
    int32_t merge_roles; /* almanac and perturbation duties share wave 1 */
    int32_t pipe;     /* pipelined stage loop (16-wave workgroups; any plain workgroup without a gravity field that has the integrator in a wave of its own): see role_loop */
    int32_t ed_reuse; /* > 0: stage-0 epoch data is carried between attempts; value = fields kept per lane (9 + 3 * n_slots) */
    DevSched sched[DEV_N_SCHED];
    double coop_frac; /* share of the harmonics terms a helper workgroup takes over (cooperative mode) */
    int32_t coop_ok;  /* PRIMARY / HELPER schedules are valid */
    int32_t harm_feed; /* hybrid feed (HYB_* below) instead of the scalar HarmEntry stream: bit 0 = in the trajectory-owning workgroups,
                        * bit 1 = in the helpers (and the owner's fallback for a helper that does not answer) */
    uint64_t hyb;     /* device address of the hybrid-feed stream: scalar side (24 bytes per stream row) */
    uint64_t hyb_v;   /* its vector side (groups of sixteen stream rows, [t3..t6][16]) */
    /* Run streams: the table once more per column schedule - [0] DEV_SCHED_SOLO, [1] DEV_SCHED_PRIMARY, [2] the helpers' schedules -,
     * every RANGE of a wave starting a sixteen-row group of its own: in the common stream a range begins inside a batch, up to seven rows
     * of the previous column in front of it (through the recursion with a zero state), per range and evaluation.  0 = not built: the
     * common stream serves.  Same rows, same operations: bit-identical sums.  rs_cols = the column headers with `start` pointing into
     * the stream. */
    uint64_t rs_hyb[3], rs_hyb_v[3], rs_cols[3];
    int32_t coop_late;  /* pipelined loop: the helper's answer of stage i is collected behind B2(i), in phase C, instead of inside the window */
    int32_t sums_wave1; /* fan-out mode (round 6): 1 + the column wave of an owner that forms the integrator's two stage sums beside it (fan_sums); 0 = the integrator forms them itself */
};

/* Column header (32 B = one s_load_dwordx8): rows of column c start at htab[start]: `nb & 0xffff` batches of HARM_BATCH
 * entries, then `nb >> 16` (< HARM_BATCH) single rows; the recursion is seeded with a2 = diag / rho, a1 = 0. */
#define HARM_BATCH 5
struct ColHdr {
    int32_t start, nb;
    double scale; /* c * sqrt(2) */
    double diag;  /* A[c][c] */
    int32_t rows, _pad; /* rows of the column (what `nb` encodes in batches of HARM_BATCH) */
};

/* Hybrid feed of the same table (plain f64 kernel).  The scalar data path delivers ~4 bytes per cycle and CU whatever the
 * occupancy, and a 56-byte entry feeds nine f64 instructions: 0.64 issue at best (tools/harm_microbench.hip).  So only the
 * first HYB_KS values of a row (g, t1, t2) travel as scalars - 24 bytes per row, batches of HYB_ROWS rows = three
 * s_load_dwordx16 - and t3..t6 travel through VECTOR registers: sixteen rows per register pair, lane e of every 16-lane row
 * holding row e's value, and v_fmac_f64_dpp row_newbcast:e multiplies by it - a wave-uniform operand without the scalar path.
 * The table is ONE stream of rows, the columns' rows back to back (stream row r = entry r of the HarmEntry table).  A wave walks a
 * range of consecutive columns as a contiguous piece of that stream - no tail batches, no per-column pipeline drain, a fixed
 * prefetch distance - and finds the column boundaries by counting rows (headers: ColHdr, one column ahead).
 *   scalar side : 24 bytes per stream row, batches of eight stream rows (64-byte aligned)
 *   vector side : groups of sixteen stream rows, [4 values t3..t6][16 rows] = 64 doubles
 * Same operations on the same operands in the same order as the scalar stream: bit-identical sums. */
#define HYB_KS 3
#define HYB_ROWS 8
#define HYB_GROUP 64

/* One (n', c) entry of the harmonics table: 56 B, seven scalar-register pairs; a batch of five is 70 SGPRs.
 * The column recursion of the reference, a_n = u b_n a_{n-1} - c_n a_{n-2} with c_n = b_n / b_{n-1}, is carried on
 * a~_n = a_n / B_n, B_n = prod_{k=c+1..n} b_k:  a~_n = (rho u) a~_{n-1} - rho^2 g_n a~_{n-2},  g_n = 1 / b_{n-1}^2 —
 * ONE coefficient per row instead of two — and B_n is folded into the Stokes coefficients:
 *   g      : g_n; row n' = c has g = -1 so that the generic step yields rho * diag from the seed, row c + 1 has g = 0
 *   t1, t2 : B * (C, S) of (n', c)                            -> x / y sums
 *   t3, t4 : B * sqrt2 * vr01[n'][c-1] * (C, S)[n'][c-1]      -> z sum
 *   t5, t6 : B * sqrt2 * vr11[n'-1][c-1] * (C, S)[n'-1][c-1]  -> w sum */
struct HarmEntry {
    double g, t1, t2, t3, t4, t5, t6;
};

/* Mailbox of a trajectory-owning workgroup (uncached global memory).  The owner writes the harmonics inputs of
 * evaluation `seq` (1, 2, ...) and then posted[owner] = seq; a helper claims the job by moving claimed[owner] from
 * seq - 1 to seq (compare-and-swap), evaluates its columns and answers with the partial sums and done = seq.  The
 * words the helpers scan (posted, claimed, finished) are packed per set of 16 owners (one 64-byte line per set and
 * kind: word index = set * 16 + slot, owner = set + slot * n_sets), so one load scans a set. */
struct CoopBox { /* double-buffered by the parity of `seq`: the next evaluation is posted before the previous answer is read */
    /* Every double travels as two TAGGED 8-byte granules, {low half | seq << 32} and {high half | seq << 32}, each written by one
     * naturally aligned 8-byte store - the unit the memory system never tears.  A reader accepts a value when both tags carry the
     * sequence number it expects; nothing has to be ordered against anything (no drain of the stores before a flag, no flag for the
     * answer, no second round trip for the data behind a flag): the memory was zeroed before the launch, sequence numbers start
     * at 1, and a slot is rewritten two evaluations later, with another tag. */
    uint64_t in[2][5][2][DEV_LANES];
    uint64_t out[2][4][2][DEV_LANES]; /* the answer (of part 0 when the hand-off has two parts) */
    uint32_t pad[16];
};
/* The answer of part 1, in an array of its own (DevBatch.coop_out2) that exists only when the hand-off has two parts.  Measured in
 * round 4: the mailboxes of configs[1] (157 owners) in ONE array of 26.6 KB boxes - 4.2 MB of uncached memory, whatever the padding -
 * cost 7 % of the run against the same code on 18.4 KB boxes (2.9 MB); the footprint of this memory is not free. */
struct CoopOut {
    uint64_t out[2][4][2][DEV_LANES];
};

struct DevBatch { /* device pointers of one launch */
    int64_t n;
    int64_t duration_ns;
    int64_t end_epoch_ns;
    int32_t use_end_epoch;
    int32_t _pad;
    const int64_t *epoch_ns;
    const double *x, *y, *z, *vx, *vy, *vz, *cr, *cd, *mprop, *mdry, *mextra, *asrp, *adrag;
    const int64_t *step_in;
    /* stop condition (propagators/event.rs:88-146); ev_on = 0 => none */
    int32_t ev_on, _pad2;
    double ev_mu;
    const nyx_hip_event_t *ev; /* DEVICE copy of the event (scalar, trigger, desired value, observer frame).  A pointer on purpose: by
                                  value its address would be taken (event_step) and the whole launch descriptor would move to scratch */
    double *ev_prev;   /* [n] event value of the previous accepted state */
    int32_t *ev_count; /* [n] crossings so far */
    int32_t *ev_found; /* [n] 1 when the propagation stopped on the event */
    /* cooperative mode: workgroups [0, ceil(n/64)) own trajectories, [coop_base, coop_base + coop_helpers) help */
    int32_t coop_helpers, coop_base;
    int32_t lds_bytes, coop_parts; /* dynamic LDS of the launch: zeroed by every workgroup before use (see propagate_body); coop_parts: sub-jobs per evaluation (1 or 2) */
    int32_t coop_mute, coop_sets; /* coop_sets: owners are dealt into this many sets of <= 16, each watched by its own helpers */ /* test switch (NYX_HIP_COOP_MUTE): helpers exit at once, as if they had never become resident */
    int32_t coop_fan, _pad_fan; /* 1: fan-out mode - helper h is DEDICATED to owner h % owners, part h / owners of coop_parts (no claims); answers of the parts
                                 * 1.. go to coop_out2[owner * coop_parts + part], the part-0 helper adds them to its own and answers the owner's mailbox */
    struct CoopBox *coop_box; /* one mailbox per trajectory-owning workgroup, zeroed before the launch */
    struct CoopOut *coop_out2; /* [owners] answers of part 1 (coop_parts == 2), else NULL */
    uint32_t *coop_posted, *coop_claimed, *coop_finished; /* [owners] packed scan words, zeroed before the launch */
    const int64_t *dur_ns; /* optional per-trajectory duration (covariance-mapping segments); overrides duration_ns */
    const double *stm; /* [n][81] column-major per trajectory, or NULL */
    double *o_stm;
    int64_t stm_hist_stride; /* elements between consecutive rows of stm_hist */
    double *stm_hist; /* NYX_HIP_FLAG_STM_TEXTBOOK: [16 stages][12][n] the stage matrices of the current attempt - G_i (9, row-major), c_i (3) -
                       * for the variational equations d(Phi)/dt = A Phi integrated at the accepted step (stm_update_textbook); else NULL */
    int64_t *o_epoch_ns;
    double *o_x, *o_y, *o_z, *o_vx, *o_vy, *o_vz, *o_cr, *o_cd, *o_mprop, *o_mdry, *o_mextra, *o_asrp, *o_adrag;
    int64_t *o_step;
    int32_t *status;
    int64_t *last_step_ns;
    double *last_error;
    int32_t *last_attempts;
    int64_t *n_acc, *n_rej, *n_evals;
    int64_t traj_cap;  /* 0 => no dense output */
    int64_t *t_epoch;  /* [cap][n] */
    double *t_state[6];
    int32_t *t_len;
    int64_t *prof; /* optional [16][8] cycle counters written by workgroup 0 (NYX_HIP_PROFILE) */
    const struct PredictArgs *pred; /* STM kernels, optional: the covariance-mapping loop of nyx_hip_predict_until in ONE launch - DEVICE copy of the
                                     * loop's arguments (predict_args.h); the integrator wave performs the Kalman time update of its trajectories at
                                     * every segment boundary and re-arms them (propagate_kernel.hip, segment_update) */
};

#endif
