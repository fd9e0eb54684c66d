/*
 * nyx_hip.h — C-ABI of the MI355X-native batched propagation path.
 *
 * This is the drop-in boundary for ONE hot path of nyx-space/nyx: the adaptive
 * Runge-Kutta integration of the SpacecraftDynamics force-model stack, batched
 * over many dispersed Spacecraft states.  The reference has no FFI for this path
 * (Propagator<D> is a monomorphised generic), so the cut is made at the level at
 * which the reference itself parallelises:
 *
 *   - `MonteCarlo::run_until_epoch` -> rayon `par_iter` over initial states
 *         nyx-core/src/mc/montecarlo.rs:233-253
 *   - Python `Propagator.many_for_duration(spacecraft, duration)`
 *         nyx-py/src/py_md.rs:275-321
 *   - one trajectory = `Propagator::with(state, almanac).for_duration(d)`
 *         nyx-core/src/propagators/propagator.rs:88-108, instance.rs:87-282
 *
 * Everything in here is plain data: no C++ types, no torch types, no ownership
 * transfer.  The caller allocates and frees every array; the library owns only
 * the device memory held inside a `nyx_hip_ctx`.  Functions return 0 on success
 * and never unwind across the ABI; per-trajectory failures are reported through
 * the `status` array (the analogue of `Run.result: Result<_, PropagationError>`,
 * nyx-core/src/mc/results.rs:48-59), index-stable.
 *
 * Units follow the reference: km, km/s, kg, m^2, seconds; epochs are integer
 * nanoseconds (hifitime's Duration grain) counted in TDB past J2000.
 */
#ifndef NYX_HIP_H
#define NYX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NYX_HIP_ABI_VERSION 4

/* ---- IntegratorMethod: nyx-core/src/propagators/rk_methods/mod.rs:65-79 ---- */
enum nyx_hip_method {
    NYX_HIP_RK89 = 0,     /* RungeKutta89 (default), rk.rs:91-251       */
    NYX_HIP_DP78 = 1,     /* DormandPrince78, dormand.rs:73-184         */
    NYX_HIP_DP45 = 2,     /* DormandPrince45, dormand.rs:24-66          */
    NYX_HIP_RK4 = 3,      /* RungeKutta4, rk.rs:66-81                   */
    NYX_HIP_CASHKARP45 = 4, /* CashKarp45, rk.rs:24-58                  */
    NYX_HIP_VERNER56 = 5  /* Verner56, verner.rs:26-79                  */
};

/* ---- ErrorControl: nyx-core/src/propagators/error_ctrl.rs:30-76 ---- */
enum nyx_hip_error_ctrl {
    NYX_HIP_RSS_CARTESIAN_STATE = 0,
    NYX_HIP_RSS_CARTESIAN_STEP = 1, /* default */
    NYX_HIP_RSS_STATE = 2,
    NYX_HIP_RSS_STEP = 3,
    NYX_HIP_LARGEST_ERROR = 4,
    NYX_HIP_LARGEST_STATE = 5,
    NYX_HIP_LARGEST_STEP = 6
};

/* ---- per-trajectory status (replaces Result<_, PropagationError>) ---- */
enum nyx_hip_status {
    NYX_HIP_OK = 0,
    NYX_HIP_ERR_NAN = 1,            /* PropMathError, instance.rs:432-439          */
    NYX_HIP_ERR_MASSLESS = 2,       /* MasslessSpacecraft, dynamics/spacecraft.rs:201-203 */
    NYX_HIP_ERR_FUEL_EXHAUSTED = 3, /* FuelExhausted, dynamics/spacecraft.rs:163-168 */
    NYX_HIP_ERR_EPHEM_RANGE = 4,    /* almanac lookup outside the loaded segments   */
    NYX_HIP_ERR_UNSUPPORTED = 5,    /* model combination the device path refuses   */
    NYX_HIP_ERR_EVENT_NOT_FOUND = 6, /* NthEventError: max_duration reached first (propagators/event.rs:170-176) */
    NYX_HIP_ERR_EVENT_SEARCH = 7    /* PropagationError::Analysis / TrajectoryEvent: the root search in the bracket failed */
};

/* ---- library-level return codes ---- */
enum nyx_hip_rc {
    NYX_HIP_RC_OK = 0,
    NYX_HIP_RC_BAD_ARG = 1,
    NYX_HIP_RC_NO_DEVICE = 2,
    NYX_HIP_RC_HIP_ERROR = 3,
    NYX_HIP_RC_UNSUPPORTED = 4
};

/* IntegratorOptions: nyx-core/src/propagators/options.rs:42-61 (defaults :172-186).
 * Durations are integer nanoseconds, exactly what hifitime stores. */
typedef struct nyx_hip_integ_opts {
    int64_t init_step_ns; /* 60 s   */
    int64_t min_step_ns;  /* 1 ms   */
    int64_t max_step_ns;  /* 2700 s */
    double tolerance;     /* 1e-12  */
    int32_t attempts;     /* u8 in the reference; 50 */
    int32_t fixed_step;   /* bool */
    int32_t error_ctrl;   /* enum nyx_hip_error_ctrl */
    int32_t method;       /* enum nyx_hip_method (PropagatorConfig.method, sequence/config.rs:138-143) */
} nyx_hip_integ_opts_t;

/* One SPK-type-2-style Chebyshev segment: position of `target` w.r.t. `center`
 * (replaces what `almanac.transform` reads from the BSP; call sites
 * dynamics/orbital.rs:230-234, dynamics/solarpressure.rs:138-143).
 * Record r covers [init_et_s + r*interval_s, +interval_s); each record is
 * [mid_et_s, radius_s, X[n_coeffs], Y[n_coeffs], Z[n_coeffs]]. */
typedef struct nyx_hip_cheby_segment {
    double init_et_s;
    double interval_s;
    int32_t n_records;
    int32_t n_coeffs;
    const double *records; /* n_records * (2 + 3*n_coeffs) doubles */
} nyx_hip_cheby_segment_t;
/* The device Clenshaw loop walks a fixed window: segments with more coefficients per component are refused by
 * nyx_hip_ctx_create (NYX_HIP_RC_UNSUPPORTED), never truncated.  DE440s uses 11-13. */
#define NYX_HIP_MAX_CHEBY_COEFFS 32

#define NYX_HIP_MAX_CHAIN 4
#define NYX_HIP_MAX_BODIES 8
#define NYX_HIP_MAX_SEGMENTS 16

/* A celestial body as the path needs it: GM, mean equatorial radius (eclipse),
 * and its position relative to the integration centre as a signed sum of
 * segments (the ephemeris-tree walk ANISE does, flattened by the host). */
typedef struct nyx_hip_body {
    int32_t naif_id;
    int32_t n_chain;                       /* 0 => the integration centre itself */
    int32_t chain_segment[NYX_HIP_MAX_CHAIN];
    int32_t chain_sign[NYX_HIP_MAX_CHAIN]; /* +1 / -1 */
    double mu_km3_s2;
    double mean_radius_km;
} nyx_hip_body_t;

/* Body-fixed orientation w.r.t. the inertial integration frame (replaces almanac.transform_to / almanac.rotate at
 * gravity_field.rs:150-154, 258-265, drag.rs:184-189, 223-228).  Two kinds, as ANISE has them:
 *
 * NYX_HIP_ROT_IAU (planetary constants, PCK text kernels): alpha = ra[0] + ra[1]*T + ra[2]*T^2 (deg, T in Julian
 *   centuries TDB), delta likewise, W = w[0] + w[1]*d + w[2]*d^2 (deg, d in days TDB), plus the trigonometric
 *   (nutation-precession) series of the IAU reports, BODYnnn_NUT_PREC_*: theta_k = angle[k][0] + angle[k][1]*T,
 *   alpha += ra_k sin theta_k, delta += dec_k cos theta_k, W += w_k sin theta_k (IAU_MOON has 13 terms, IAU_EARTH none);
 *   DCM(inertial->fixed) = R3(W) R1(90deg - delta) R3(90deg + alpha).
 * NYX_HIP_ROT_EULER_CHEBY (binary PCK, type 2: ITRF93 from the Earth high-precision BPCs, MOON_PA): three Euler
 *   angles (rad) as Chebyshev records [mid_et_s, radius_s, A1[n], A2[n], A3[n]] held as one of config.segments;
 *   DCM(base->fixed) = R3(A3) R1(A2) R3(A1) (SPICE EUL2M(w, delta, phi, 3, 1, 3)), times `base_dcm` = inertial
 *   integration frame -> the segment's base frame (identity for J2000-based files, the obliquity rotation for the
 *   ECLIPJ2000-based Earth BPCs), flattened by the host like the ephemeris chains. */
enum nyx_hip_rotation_kind { NYX_HIP_ROT_IAU = 0, NYX_HIP_ROT_EULER_CHEBY = 1 };
#define NYX_HIP_MAX_NUT_PREC 16
typedef struct nyx_hip_rotation {
    double ra_deg[3];
    double dec_deg[3];
    double w_deg[3];
    int32_t kind;       /* enum nyx_hip_rotation_kind */
    int32_t n_nut_prec; /* IAU: number of trigonometric terms (0 = polynomials only) */
    double nut_prec_angle_deg[NYX_HIP_MAX_NUT_PREC][2];
    double nut_prec_ra[NYX_HIP_MAX_NUT_PREC];
    double nut_prec_dec[NYX_HIP_MAX_NUT_PREC];
    double nut_prec_w[NYX_HIP_MAX_NUT_PREC];
    int32_t euler_segment; /* EULER_CHEBY: index into config.segments */
    int32_t _pad;
    double base_dcm[9];    /* EULER_CHEBY: row-major */
} nyx_hip_rotation_t;

/* GravityFieldData + frame constants: io/gravity.rs:90-96, gravity_field.rs:195-207.
 * c_nm/s_nm are fully-normalised, packed lower-triangular: idx(n,m)=n(n+1)/2+m,
 * n = 0..degree.  mu/R_eq are the FRAME's values (Appendix A.9 of SURVEY.md). */
typedef struct nyx_hip_gravity_field {
    int32_t degree;
    int32_t order;
    int32_t offset_body;   /* 0: the field of the integration centre.  k > 0: the field of config.bodies[k - 1], another body than the
                            * centre - GravityField::eom transforms the orbit into `grav_data.frame` whatever its centre
                            * (gravity_field.rs:150-154: translation to that body, then its body-fixed rotation) and rotates the
                            * acceleration back (:258-265); what the spacecraft feels is that body's harmonics at r - r_body(t) */
    int32_t _pad;
    double mu_km3_s2;
    double eq_radius_km;
    const double *c_nm;
    const double *s_nm;
    nyx_hip_rotation_t rotation;
} nyx_hip_gravity_field_t;

/* SolarPressure + ShadowModel: dynamics/solarpressure.rs:40-48, cosmic/eclipse.rs:34-37 */
typedef struct nyx_hip_srp {
    double phi_w_m2;         /* 1367.0 */
    int32_t estimate;        /* d a / d Cr into STM column 6 (solarpressure.rs:131-133) */
    int32_t sun_body;        /* index into bodies[] */
    int32_t n_shadow_bodies;
    int32_t shadow_body[NYX_HIP_MAX_BODIES]; /* indices into bodies[] */
} nyx_hip_srp_t;

/* AtmDensity / Drag: dynamics/drag.rs:39-45,115-123 */
enum nyx_hip_density { NYX_HIP_RHO_CONSTANT = 0, NYX_HIP_RHO_EXPONENTIAL = 1, NYX_HIP_RHO_STDATM = 2 };
typedef struct nyx_hip_drag {
    int32_t density;      /* enum nyx_hip_density */
    int32_t _pad;
    double rho0;          /* Constant(rho) or Exponential.rho0 */
    double r0;            /* Exponential.r0 (metres, sic) */
    double ref_alt_m;     /* Exponential.ref_alt_m */
    double max_alt_m;     /* StdAtm.max_alt_m */
    double eq_radius_km;  /* drag frame mean equatorial radius */
    nyx_hip_rotation_t rotation; /* drag frame (IAU Earth) */
} nyx_hip_drag_t;

/* SolidTides (dynamics/solid_tides.rs:43-72): IERS-2010 degree-2/3 deformation of the central body raised by the
 * perturbers, evaluated as time-varying delta-C/S on a body-fixed frame of the central body. */
typedef struct nyx_hip_solid_tides {
    double k2, k3;               /* Love numbers (earth_moon_system: 0.3019, 0.093; solid_tides.rs:189-218) */
    double mu_km3_s2;            /* frame.mu_km3_s2() of the tidal (body-fixed) frame */
    double eq_radius_km;         /* frame.mean_equatorial_radius_km() */
    nyx_hip_rotation_t rotation; /* orientation of the tidal frame */
    int32_t n_perturbers;
    int32_t perturber_body[NYX_HIP_MAX_BODIES];   /* TidalPerturber.frame as an index into bodies[] (its mu is used) */
    int32_t compute_degree_3[NYX_HIP_MAX_BODIES]; /* TidalPerturber.compute_degree_3 */
    int32_t _pad;
} nyx_hip_solid_tides_t;

/* ---- Execution tuning (ABI v4).  Everything that used to be a process environment variable: a host behind the C-ABI cannot
 * reason about the environment, and several of these change the ORDER in which the harmonics are summed, i.e. the last bits
 * of the result.  config.tuning = NULL selects the defaults (every field "auto").
 *
 * Reproducibility contract.  `schedule` fixes how the harmonics columns are dealt over the waves of a workgroup, hence the
 * summation order:
 *   NYX_HIP_SCHED_MODEL (default)  a cost model keyed on the workgroup shape and the force model only: the same config gives
 *                                  the same bits in every context, process, rank and on every box.  No hidden launches.
 *   NYX_HIP_SCHED_CALIBRATED       the first launch of every workgroup shape runs up to four short launches of the workload's
 *                                  own first steps (it SYNCHRONISES the stream and copies a few hundred bytes back: not
 *                                  graph-capturable) and deals the columns by the measured cycle counts: a few percent
 *                                  faster, and context-dependent in the last bits (~0.05 mm after 45 min).
 *                                  The calibration launches run the accounting twin of the kernel (its in-kernel counters
 *                                  are the measurement): the weights are that build's, 1-7 % off the product kernel's timing.
 *   NYX_HIP_SCHED_EXPLICIT         wave_weights[] / role_duties[] as given.
 * `deterministic` = 1 additionally makes every trajectory's bits independent of the BATCH it is launched in (batch size,
 * position in the batch, rank-sharding): the cooperative mode - whose column split follows the ratio of idle CUs to
 * workgroups - is switched off, and the workgroup SHAPE (waves per workgroup; the STM layout, which is then the quad layout
 * unless `stm_quad` says otherwise) is taken from the configuration alone instead of the batch size - both decide the column
 * split, i.e. the order of the sums.  The reference is reproducible per (seed, index) in exactly this sense
 * (mc/montecarlo.rs:208-224, 290-295).  Cost: the cooperative gain (~1.27x at 10 000 trajectories on 256 CUs) for ensembles
 * that leave CUs idle; the quad STM layout issues 1.6x the f64 slots of the 64-lane one above ~8 000 trajectories (set
 * stm_quad = 0 there).  (Since round 4 the plain kernel's shape - sixteen waves per workgroup from degree 24 on - does not depend
 * on the batch size in any mode.)  nyx_hip_ctx_set_column_waves / stm_quad pin the shape explicitly either way. */
enum nyx_hip_schedule { NYX_HIP_SCHED_MODEL = 0, NYX_HIP_SCHED_CALIBRATED = 1, NYX_HIP_SCHED_EXPLICIT = 2 };
typedef struct nyx_hip_tuning {
    int32_t schedule;          /* enum nyx_hip_schedule */
    int32_t deterministic;     /* 1: bits per trajectory independent of the batch (cooperative mode off) */
    int32_t cooperative;       /* -1 auto (on when the launch has fewer workgroups than CUs), 0 off, 1 on */
    int32_t pipelined;         /* -1 auto, 0 two-barrier stage loop, 1 pipelined stage loop (sixteen-wave workgroups; workgroups without a gravity field) */
    int32_t chained_attempts;  /* -1 auto, 0 off, 1 speculative stage 0 of the next attempt */
    int32_t epoch_data_reuse;  /* -1 auto, 0 off (only read when the attempts are not chained) */
    int32_t role_fanout;       /* -1 auto, 0 off, 1 almanac / perturbation duties dealt over several waves */
    int32_t merge_roles;       /* 0 / 1: almanac and perturbation duties share one wave */
    int32_t stm_quad;          /* -1 by ensemble size, 0 64 trajectories x 3-partial duals, 1 quad layout */
    int32_t harmonics_feed;    /* -1 auto, 0 scalar stream, 1 hybrid scalar + DPP stream, 2 hybrid in the trajectory-owning workgroups only, 3 in the helpers only */
    int32_t coop_max_columns;  /* 0 auto: columns a helper workgroup may take */
    int32_t coop_mute;         /* test switch: helpers never answer (exercises the owner's fallback) */
    int32_t profile;           /* 1: in-kernel cycle accounting of workgroup 0 (nyx_hip_debug_profile) */
    int32_t debug_flags;       /* timing-only switches (0x100 skip the serial role work, 0x200 skip the harmonics): WRONG RESULTS;
                                * 0x400 host trace; A/B switches with the SAME results: 0x800 role offload ON (off by default since round 5), 0x1000 no segment-level almanac
                                * units on a single almanac wave, 0x2000 packed Chebyshev records; schedule switches (another summation order):
                                * 0x8000 the two-ended column fill everywhere, 0x10000 one contiguous run of columns per wave whatever the feed,
                                * 0x20000 owners / helpers rounded to multiples of eight, 0x40000 / 0x80000 two-part / one-part hand-off,
                                * 0x100000 no mailbox pool (allocate and free per context), 0x200000 helpers fetch a job's inputs before they
                                * know they won its claim (rounds 1-3), 0x400000 the helper dealing of rounds 1-4, 0x800000 the common table stream for every
                                * schedule (no run streams), 0x1000000 the helper's answer collected inside the window (rounds 1-4; no effect in the sixteen-wave plain kernels since round 6, whose integrator always collects it in phase C), 0x8000000 no fan-out mode (small shards keep the claim mode of rounds 1-5: one helper workgroup per owner), 0x10000000 chained attempts also with the almanac duty fanned out over several waves of a gravity-field workgroup (tools), 0x40000000 fan-out mode: the integrator forms its two stage sums itself (A/B of the sums wave, which only exists in builds with -DNYX_FAN_SUMS: measured in round 6, no gain, compiled out), 0x20000000 nyx_hip_predict_until as one segment launch + one time-update launch per segment (rounds 2-5) instead of ONE launch for the whole loop, 0x4000000 quad STM layout: the position-only pieces of phase C formed by the integrator wave (rounds 3-4; since round 5 by an almanac wave), 0x2000000 the linear column partition of round 4 for the cooperative
                                * 70x70 shape too (default since round 5: its runs placed in a free wave order); 0x4000 full-range sincos for a polynomial
                                * IAU orientation every stage (results differ by the rounding of the large argument) */
    double coop_fraction;      /* 0 auto: share of the harmonics terms a helper takes */
    double coop_helper_ratio;  /* 0 auto: helper workgroups per trajectory-owning workgroup */
    double column_start_cost;  /* < 0 auto: rows a column's start is charged in the schedule */
    double role_duties[3];     /* all 0: model.  Integrator, almanac, perturbation duty in harmonics-term units */
    double age_weights[4];     /* NYX_HIP_SCHED_EXPLICIT, all 0: unused.  One speed weight per SIMD age class */
    double wave_weights[16];   /* NYX_HIP_SCHED_EXPLICIT: per-wave speed weights (all 0: age_weights) */
} nyx_hip_tuning_t;
/* all "auto" */
#define NYX_HIP_TUNING_DEFAULT {0, 0, -1, -1, -1, -1, -1, 0, -1, -1, 0, 0, 0, 0, 0.0, 0.0, -1.0, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0}}

/* PropagatorConfig{dynamics, method, options}: dynamics/sequence/config.rs:96-169.
 * Model order is the reference's `Dynamics::build` order: two-body, point
 * masses, gravity field, solid tides; then SRP, drag. */
typedef struct nyx_hip_config {
    uint32_t abi_version; /* NYX_HIP_ABI_VERSION */
    uint32_t flags;       /* NYX_HIP_FLAG_* */
    nyx_hip_integ_opts_t opts;

    double central_mu_km3_s2; /* osc.frame.mu_km3_s2(), orbital.rs:86-92 */

    int32_t n_segments;
    int32_t n_bodies;
    const nyx_hip_cheby_segment_t *segments;
    const nyx_hip_body_t *bodies;

    /* PointMasses.celestial_objects (orbital.rs:176-182), as indices into bodies[] */
    int32_t n_point_masses;
    int32_t point_mass_body[NYX_HIP_MAX_BODIES];

    const nyx_hip_gravity_field_t *gravity; /* NULL => none */
    const nyx_hip_srp_t *srp;               /* NULL => none */
    const nyx_hip_drag_t *drag;             /* NULL => none */

    double speed_of_light_km_s; /* anise::constants::SPEED_OF_LIGHT_KM_S (cosmic/mod.rs:179-180) */

    const nyx_hip_solid_tides_t *tides; /* NULL => none (accel_models.solid_tides, config.rs:116-118) */

    /* opts.integration_frame (propagators/options.rs:60; instance.rs:117-142, 211-220).  When the states of a batch are centred on
     * ANOTHER body than the integration centre (same orientation), name it here: bodies[state_frame_body], whose chain is that
     * body w.r.t. the integration centre.  nyx_hip_propagate_batch[_device] and nyx_hip_propagate_until_epoch then translate every
     * state into the integration frame at its start epoch (position AND velocity of the chain, no aberration: what
     * almanac.transform_to(orbit, frame, None) does between two frames of one orientation), propagate, and translate the final
     * state back at its final epoch.  0 (the integration centre itself) or a body without a chain: no swap.
     * With dense output (nyx_hip_traj_t) the reference's `Traj` is reproduced as it is built (instance.rs:297-326): entry 0 is the
     * start state in the CALLER's frame, every published state is in the INTEGRATION frame, only the returned final state is
     * translated back.  nyx_hip_predict_until translates in and out per segment, as the reference's loop of `until_epoch` calls does
     * (od/process/mod.rs:440-486).  The event search refuses a swap (NYX_HIP_RC_UNSUPPORTED): the reference returns from inside its
     * loop there without translating back (instance.rs:243-250). */
    int32_t state_frame_body;
    int32_t _pad_cfg;
    const nyx_hip_tuning_t *tuning; /* NULL => NYX_HIP_TUNING_DEFAULT (ABI v4) */
    /* A second GravityField of the same OrbitalDynamics (accel_models is a list: the reference's cislunar set-ups stack the Earth's
     * and the Moon's field; each transforms the orbit into ITS frame, gravity_field.rs:150-154).  `gravity` is the one that gets the
     * column waves - give it the larger field -, this one is evaluated in one piece by the perturbation wave beside them.  Either
     * may be the integration centre's or another body's (offset_body).  NULL => none.  With NYX_HIP_FLAG_STM the second field's
     * gradient (GravityField::gradient, gravity_field.rs:273-431: the same frame handling, duals on the translated and rotated
     * radius, dcm * grad * dcm^T) is formed by the perturbation wave, in either STM layout (round 5: the quad layout too, one partial
     * per lane, bit-identical to the 64-lane form); a non-central `gravity` takes either layout as well. */
    const nyx_hip_gravity_field_t *gravity2;
} nyx_hip_config_t;

/* flags */
#define NYX_HIP_FLAG_STM 0x1u          /* states carry a 9x9 STM (Spacecraft.stm = Some) */
#define NYX_HIP_FLAG_STM_TEXTBOOK 0x2u /* (with NYX_HIP_FLAG_STM) the variational equations dPhi/dt = A(t) Phi, integrated with the state by the step's
                                        * own tableau, instead of the reference's first-order Phi_{n+1} = Phi_n (I + h sum b_i A_i) (spacecraft.rs:208-224: the
                                        * step-start STM right-multiplied).  No reference vector exists for this form; it is pinned as the derivative of the
                                        * flow (tests/test_oracle_stm_textbook.py: 5e-7 of finite differences where the reference's form is 0.2 off) and device =
                                        * oracle to rounding (tests/test_gpu_stm_textbook.py).  The states are the same bits either way.  64-lane dual
                                        * layout only (tuning.stm_quad is ignored); through nyx_hip_predict_until as well. */

/* Spacecraft batch, structure-of-arrays (one array per field, length n).
 * Mirrors Spacecraft::to_vector / set (cosmic/spacecraft.rs:451-497) plus the
 * constants carried by the struct (Mass, SRPData, DragData). */
typedef struct nyx_hip_states {
    int64_t n;
    int64_t *epoch_ns;  /* TDB ns past J2000 */
    double *x_km, *y_km, *z_km;
    double *vx_km_s, *vy_km_s, *vz_km_s;
    double *cr;            /* srp.coeff_reflectivity  (vector[6]) */
    double *cd;            /* drag.coeff_drag         (vector[7]) */
    double *prop_mass_kg;  /* mass.prop_mass_kg       (vector[8]) */
    double *dry_mass_kg;
    double *extra_mass_kg;
    double *srp_area_m2;
    double *drag_area_m2;
    double *stm;           /* NULL or n*81, per trajectory column-major 9x9 (vector[9..90]) */
    int64_t *step_ns;      /* NULL, or in: PropInstance.step_size to resume with (0 => opts.init_step);
                              out: step_size after the call (instance.rs:464-473) */
} nyx_hip_states_t;

/* IntegrationDetails of the last step (propagators/mod.rs:49-56) + counters. All arrays length n, any may be NULL. */
typedef struct nyx_hip_step_stats {
    int32_t *status;        /* enum nyx_hip_status */
    int64_t *last_step_ns;  /* details.step */
    double *last_error;     /* details.error */
    int32_t *last_attempts; /* details.attempts */
    int64_t *n_accepted;    /* accepted steps (incl. the final fixed step) */
    int64_t *n_rejected;    /* rejected attempts */
    int64_t *n_evals;       /* eom calls = stages * attempts */
} nyx_hip_step_stats_t;

/* Dense output: every state the reference publishes on its channel (instance.rs:188-193, 254-259) preceded by the start
 * state — the content of `Traj` before `finalize()` sorts it (md/trajectory/traj.rs:75-80, instance.rs:297-326).
 * Step-major arrays: entry k of trajectory i is at [k * n + i]; k = 0 is the start state.  `len[i]` counts the states
 * PRODUCED; states beyond `capacity` are not stored. */
typedef struct nyx_hip_traj {
    int64_t capacity;
    int64_t *epoch_ns;
    double *x_km, *y_km, *z_km, *vx_km_s, *vy_km_s, *vz_km_s;
    int32_t *len;
} nyx_hip_traj_t;

typedef struct nyx_hip_ctx nyx_hip_ctx;

/* Number of visible HIP devices (0 if none / no runtime). */
int32_t nyx_hip_device_count(void);

/* Builds a propagation context on `device`: validates the config (counts, chain lengths, segment shapes: bad input is
 * NYX_HIP_RC_BAD_ARG / _UNSUPPORTED, never silently clipped), precomputes the recursion tables (GravityField::new,
 * gravity_field.rs:52-132) and uploads every shared table once.  `Propagator::new` + `Arc<Almanac>` analogue.
 * Concurrency: the physical model is immutable, and every entry point takes the context's lock, so host threads may
 * share a context (`Send + Sync`).  On the device the launches of ONE context run one after the other whatever their
 * streams (they share the launch descriptor, the cooperative-mode mailboxes and the staging blocks; each launch waits
 * on the previous one's completion event).  For kernels that overlap, create one context per stream. */
int32_t nyx_hip_ctx_create(const nyx_hip_config_t *cfg, int32_t device, nyx_hip_ctx **out);
void nyx_hip_ctx_destroy(nyx_hip_ctx *ctx);

/* Propagates every state of `in` for `duration_ns` (may be negative) — the batch
 * form of `prop.with(state, almanac).for_duration(duration)`.  `in` and `out`
 * are HOST SoA batches (out may alias in); H2D/D2H is done inside. */
int32_t nyx_hip_propagate_batch(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                nyx_hip_states_t *out, nyx_hip_step_stats_t *stats);

/* Same, but every pointer in `in`/`out`/`stats` is a DEVICE pointer and the
 * launch is enqueued on `hip_stream` (a hipStream_t, NULL = default stream);
 * returns without synchronising.  This is the entry bench.py times with inputs
 * resident in HBM. */
int32_t nyx_hip_propagate_batch_device(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                       nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, void *hip_stream);

/* `for_duration_with_traj` (instance.rs:297-326) for the batch: as nyx_hip_propagate_batch, plus the accepted states.
 * Host arrays; `traj` arrays must hold capacity * n elements. */
int32_t nyx_hip_propagate_batch_with_traj(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                          nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, nyx_hip_traj_t *traj);
/* Device-pointer flavour (every pointer in in/out/stats/traj is a device pointer), asynchronous on `hip_stream`.
 * Aliasing: the DEVICE entry points read `in` while they write `out`; `out` may alias `in` for a plain propagation (every workgroup
 * reads its own trajectories' start states before it writes anything), but NOT together with dense output AND an integration-frame
 * swap (config.state_frame_body): entry 0 of the dense output is then rewritten from `in` after the launch (instance.rs:319-321: the
 * start state in the caller's frame), and an aliased `in` would hold the final states by then.  The host flavours stage through
 * device blocks of their own and take any aliasing. */
int32_t nyx_hip_propagate_batch_with_traj_device(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                                 nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, nyx_hip_traj_t *traj,
                                                 void *hip_stream);

/* ---- `Traj` evaluation: md/trajectory/traj.rs:82-162, interpolatable.rs:52-108 ----
 * The stored states are read as the `finalize()`d trajectory (sorted by epoch; a back-propagated batch is simply read
 * in reverse).  One sample = `Traj::at(epoch)`: an epoch that is stored exactly returns the stored state (traj.rs:92-95);
 * otherwise a window of up to 13 stored states around the insertion index (traj.rs:104-115: 6 to the left, 13 in
 * total, and only the LAST 12 states when the window touches the end) feeds one Hermite interpolation per axis with
 * abscissas in seconds (anise `hermite_eval`, i.e. SPICE HRMINT).  Spacecraft fields other than the orbit are constant
 * on this path (no thruster), so only position and velocity are produced. */
enum nyx_hip_interp_status {
    NYX_HIP_INTERP_OK = 0,
    NYX_HIP_INTERP_NO_DATA = 1, /* TrajError::NoInterpolationData: empty trajectory or epoch outside [first, last] */
    NYX_HIP_INTERP_MATH = 2,    /* InterpolationError::InterpMath (two abscissas closer than f64::EPSILON seconds) */
    NYX_HIP_INTERP_ILL_CONDITIONED = 3 /* nyx_hip_traj_at only, a WARNING: the sample is what the reference's Hermite fit yields
                                          (it returns Ok), but its window holds two states closer than 1e-4 of the window's mean
                                          spacing - typically the exact-length final step of a propagation, a few ms after an
                                          accepted step of minutes - and the 13-point fit through such a pair is off by up to
                                          kilometres.  Counted in out->len like an OK sample. */
};

/* `Traj::at` for every trajectory at `m` shared epochs (host arrays).  Sample q of trajectory i lands at [q * n + i] of
 * `out` (out->capacity >= m; out->epoch_ns receives the query epoch; out->len[i] = number of samples with status OK) and
 * of `status` (enum nyx_hip_interp_status, m * n entries; failed samples are NaN). */
int32_t nyx_hip_traj_at(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, const int64_t *query_epoch_ns, int64_t m,
                        nyx_hip_traj_t *out, int32_t *status);
/* `Traj::every(step)` (traj.rs:148-162 + TrajIterator, traj_it.rs:33-62): per trajectory, the samples at
 * first + k*step <= last (hifitime TimeSeries::inclusive), k = 0, 1, ...  Sample k of trajectory i lands at [k * n + i];
 * out->len[i] = samples PRODUCED (those beyond out->capacity are not stored); the iteration of a trajectory ends at the
 * first failing sample, as the iterator does.  step_ns must be > 0. */
int32_t nyx_hip_traj_every(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, int64_t step_ns, nyx_hip_traj_t *out);
/* Device-pointer flavours (every pointer is a device pointer), asynchronous on `hip_stream`. */
int32_t nyx_hip_traj_at_device(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, const int64_t *query_epoch_ns,
                               int64_t m, nyx_hip_traj_t *out, int32_t *status, void *hip_stream);
int32_t nyx_hip_traj_every_device(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, int64_t step_ns,
                                  nyx_hip_traj_t *out, void *hip_stream);

/* Per-trajectory epochs variant of until_epoch (instance.rs:279-282): duration_i = end_epoch_ns - epoch_i. */
int32_t nyx_hip_propagate_until_epoch(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t end_epoch_ns,
                                      nyx_hip_states_t *out, nyx_hip_step_stats_t *stats);

/* ---- stop conditions: PropInstance::until_nth_event (propagators/event.rs:88-211) ----
 * The event is `Event{scalar, Condition::Equals(desired)}` evaluated on the integration-frame orbit (event_frame =
 * None).  After every accepted step except the final fixed one the scalar is evaluated (event.rs:108-146): a sign
 * change between consecutive states is a crossing (for angles: opposite signs AND |delta| < 180 deg, the event value
 * being the difference wrapped to [-180, 180)); the propagation stops at the end of the step in which crossing number
 * `trigger` happened, WITHOUT publishing that state; the root is then searched with Brent's method on the 13-state
 * Hermite interpolant between the last published state and the end state (event.rs:178-197), and the state returned
 * is `traj.at(event_epoch)`.  The scalars and the root search live in the absent anise crate (`analysis`): they are
 * restated (see oracle/nyx_oracle.c) and their precisions are explicit here. */
enum nyx_hip_event_scalar {
    NYX_HIP_EV_TRUE_ANOMALY_DEG = 0, /* OrbitalElement::TrueAnomaly, an angle: Event::apoapsis() = Equals(180), periapsis() = Equals(0) */
    NYX_HIP_EV_RMAG_KM = 1,
    NYX_HIP_EV_VMAG_KM_S = 2,
    NYX_HIP_EV_SMA_KM = 3,
    NYX_HIP_EV_ECC = 4,
    NYX_HIP_EV_X_KM = 5, NYX_HIP_EV_Y_KM = 6, NYX_HIP_EV_Z_KM = 7,
    NYX_HIP_EV_VX_KM_S = 8, NYX_HIP_EV_VY_KM_S = 9, NYX_HIP_EV_VZ_KM_S = 10,
    /* geometric scalars, normally evaluated in a body-fixed `frame` (tests/propagation/stopcond.rs:252-312) */
    NYX_HIP_EV_LONGITUDE_DEG = 11,   /* OrbitalElement::Longitude: atan2(y, x) in [0, 360), an angle */
    NYX_HIP_EV_DECLINATION_DEG = 12, /* OrbitalElement::Declination: asin(z / |r|) */
    NYX_HIP_EV_LATITUDE_DEG = 13,    /* OrbitalElement::Latitude: GEODETIC latitude on the frame's ellipsoid */
    NYX_HIP_EV_HEIGHT_KM = 14        /* OrbitalElement::Height: geodetic height above that ellipsoid */
};
typedef struct nyx_hip_event {
    int32_t scalar;             /* enum nyx_hip_event_scalar */
    int32_t trigger;            /* 1-based occurrence to stop at (until_event = 1) */
    double desired;             /* Condition::Equals(desired) */
    double value_precision;     /* the search ends when |event value| is below this */
    int64_t epoch_precision_ns; /* ... or when the bracket is narrower than this (then: not found in the bracket) */
    /* until_nth_event's `event_frame: Option<Frame>` (event.rs:104-117): when set, every state is expressed in this
     * body-fixed frame of the SAME centre before the scalar is evaluated (position R r, velocity R v - w x R r with
     * w = dW/dt about the frame's pole).  NYX_HIP_ROT_IAU orientations only. */
    int32_t has_frame;
    int32_t _pad;
    double frame_eq_radius_km; /* the frame's ellipsoid (geodetic scalars): mean equatorial radius ... */
    double frame_flattening;   /* ... and flattening (a - c) / a */
    nyx_hip_rotation_t frame;
} nyx_hip_event_t;

/* Batch form of `prop.with(state, almanac).until_nth_event(max_duration, &event, None, trigger)` (host arrays).
 * `out`: the interpolated state at the event (epoch_ns = event epoch); `traj` (mandatory, capacity >= 2): the states
 * recorded up to and including the end of the step where the event occurred, as the reference returns it; `crossings`
 * (n, may be NULL): sign changes counted.  stats->status: 0, NYX_HIP_ERR_EVENT_NOT_FOUND (out = state after
 * max_duration), NYX_HIP_ERR_EVENT_SEARCH, or a propagation error. */
int32_t nyx_hip_propagate_until_event(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t max_duration_ns,
                                      const nyx_hip_event_t *event, nyx_hip_states_t *out, nyx_hip_step_stats_t *stats,
                                      nyx_hip_traj_t *traj, int32_t *crossings);

/* ---- covariance mapping: KalmanODProcess::predict_until (od/process/mod.rs:440-486) for the batch ----
 * Per trajectory, repeated until epoch >= end_epoch_ns (at least once):
 *   nominal = prop_instance.for_duration(max_step)            with the STM, step size carried over (mod.rs:466)
 *   KalmanFilter::time_update(nominal)  (od/kalman/filtering.rs:59-99):
 *       covar_bar = stm * covar * stm^T  [+ Gamma * Q * Gamma^T of the LAST applicable process noise, od/snc.rs:210-283]
 *       state_bar = stm * state_deviation (DeviationTracking) or 0
 *   prop_instance.state.reset_stm()                           (mod.rs:479)
 * The context must have been created with NYX_HIP_FLAG_STM.  The propagation starts from an identity STM
 * (`nominal_state().with_stm()`, mod.rs:452); in->stm is not read. */
#define NYX_HIP_MAX_PROCESS_NOISE 4
enum nyx_hip_local_frame { NYX_HIP_FRAME_INERTIAL = 0, NYX_HIP_FRAME_RIC = 1, NYX_HIP_FRAME_VNC = 2 };
typedef struct nyx_hip_process_noise { /* ProcessNoise<U3> (od/snc.rs:40-59) */
    double diag[3];          /* ProcessNoise::diag (from_diagonal / from_velocity_km_s, snc.rs:108-135, 288-309) */
    int64_t disable_time_ns; /* no noise when the time update spans more than this (snc.rs:178-186, 248-250) */
    int64_t start_time_ns;   /* start_time, read when has_start_time != 0 (snc.rs:168-175) */
    int32_t has_start_time;
    int32_t local_frame;     /* enum nyx_hip_local_frame: the frame the diagonal is given in; rotated into the state frame at the
                                nominal orbit and only the DIAGONAL kept, as the reference does (snc.rs:219-239) */
    int32_t has_decay;       /* ProcessNoise::with_decay (snc.rs:145-160): diag_i * exp(-decay_i * (epoch - init_epoch)) (:193-197) */
    int32_t _pad;
    double decay_s[3];       /* decay constants, 1/s */
    int64_t init_epoch_ns;   /* ProcessNoise::init_epoch = the epoch of the initial estimate (kalman/initializers.rs:75-101);
                                INT64_MIN = every trajectory's own start epoch */
} nyx_hip_process_noise_t;

typedef struct nyx_hip_predict {
    int64_t max_step_ns;         /* KalmanODProcess::max_step, > 0 */
    int64_t end_epoch_ns;
    int32_t deviation_tracking;  /* KalmanVariant::DeviationTracking (filtering.rs:84-88) */
    int32_t n_process_noise;     /* KalmanFilter::process_noise, in order; scanned from the last (filtering.rs:64) */
    nyx_hip_process_noise_t process_noise[NYX_HIP_MAX_PROCESS_NOISE];
} nyx_hip_predict_t;

typedef struct nyx_hip_estimates { /* the KfEstimate fields that evolve (od/estimate/kfestimate.rs), trajectory-major */
    double *covar;     /* n*81, column-major 9x9 per trajectory: in = initial covar, out = covar_bar of the last update */
    double *state_dev; /* n*9 state_deviation in/out, or NULL (zeros in, not written) */
} nyx_hip_estimates_t;

/* ODSolution.estimates after the initial one (mod.rs:473-475): update u of trajectory i at [(u*n + i) * width]; every
 * array except n_updates may be NULL; updates beyond `capacity` are counted, not stored. */
typedef struct nyx_hip_predict_history {
    int64_t capacity;
    int64_t *epoch_ns; /* width 1  */
    double *state;     /* width 9 : x, y, z, vx, vy, vz, Cr, Cd, prop mass */
    double *stm;       /* width 81: column-major */
    double *covar;     /* width 81: covar_bar */
    double *state_dev; /* width 9  */
    int32_t *n_updates; /* [n] time updates performed */
} nyx_hip_predict_history_t;

/* Host arrays.  `out` receives the last nominal states (STM reset to identity, like the reference's instance), `stats`
 * the first failing status per trajectory (0 = Ok) with counters summed over the segments; `hist` may be NULL. */
int32_t nyx_hip_predict_until(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, const nyx_hip_predict_t *cfg, nyx_hip_estimates_t *est,
                              nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, nyx_hip_predict_history_t *hist);

/* ---- ensemble moments of the final states, on the device (the consumers of mc/results.rs:60-245 reduce a Monte Carlo result to
 * its mean and covariance; north_star: "RCCL ... only for the final trajectory/covariance reduction") ----
 *   out55[0]      number of runs whose status is 0 (status == NULL: every run)
 *   out55[1..9]   sum (x - x0),  x = [x y z vx vy vz Cr Cd prop_mass] (a NULL array of `states` counts as zeros)
 *   out55[10..54] upper triangle, row-major (i <= j), of sum (x - x0)(x - x0)^T
 * x0[9] (NULL = zeros): the caller's reference point, any state near the ensemble (the nominal final state, or the first
 * successful run's, which every rank can hold) - it keeps the sums well conditioned.  A rank-sharded host completes the
 * ensemble with ONE ncclAllReduce(sum) of these 55 doubles and no device-to-host copy of the states:
 *   mean = x0 + s / n,   cov = (S - n m m^T) / (n - 1)  with m = s / n.
 * Two small launches, a fixed grid and fixed summation order (no atomics): bit-reproducible for a given n.
 * _device: every pointer but x0 is a device pointer (out55 too), asynchronous on `hip_stream`.  Host flavour: host arrays. */
int32_t nyx_hip_ensemble_moments_device(nyx_hip_ctx *ctx, const nyx_hip_states_t *states, const int32_t *status, const double *x0,
                                        double *out55, void *hip_stream);
int32_t nyx_hip_ensemble_moments(nyx_hip_ctx *ctx, const nyx_hip_states_t *states, const int32_t *status, const double *x0, double *out55);

/* Tuning knob: number of waves that split the spherical-harmonics columns of one
 * 64-trajectory workgroup (0 = pick automatically from n). */
int32_t nyx_hip_ctx_set_column_waves(nyx_hip_ctx *ctx, int32_t waves);

/* One batch over SEVERAL contexts - the devices of one node driven by one process (what replaces the rayon par_iter of
 * mc/montecarlo.rs:233-253 when the host is not rank-sharded).  Host arrays.  Shard k of n_ctx is the contiguous index range
 * [k n / n_ctx, (k + 1) n / n_ctx); every context (created by the caller with the same config, one per device:
 * nyx_hip_device_count()) propagates its shard from its own host thread, so copies and kernels of the devices overlap; results
 * land in place (index-stable), a failing shard's return code and message are reported.  traj may be NULL.  With
 * tuning.deterministic = 1 the result is bit for bit the single-context one. */
int32_t nyx_hip_propagate_batch_sharded(nyx_hip_ctx *const *ctxs, int32_t n_ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                        nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, nyx_hip_traj_t *traj);

/* Changes the launch-time part of the tuning between launches (cooperative mode, determinism, schedule kind and explicit weights,
 * helper ratio / share, profiling); the create-time part (stage loop, role layout, table feed) must equal the context's. */
int32_t nyx_hip_ctx_set_tuning(nyx_hip_ctx *ctx, const nyx_hip_tuning_t *tuning);

/* Number of helper workgroups of the last propagate launch on this ctx (0 = every workgroup worked alone).  Cooperative
 * mode is chosen automatically when the 64-trajectory workgroups leave CUs idle (propagate_kernel.hip);
 * tuning.cooperative = 0 or tuning.deterministic = 1 disables it.
 *
 * Cooperative mode - what the library relies on, and what it does not:
 *   memory      the owner <-> helper mailboxes are one block of hipExtMallocWithFlags(hipDeviceMallocUncached) memory per context
 *               (pooled per process and device, zeroed by hipMemsetAsync on the launch stream before every cooperative launch); it
 *               is touched by device-scope RELAXED 8-byte atomic loads and stores and one compare-and-swap per job, nothing else;
 *   visibility  every value travels as naturally aligned 8-byte granules {half of a double | sequence number << 32}; a reader
 *               accepts a value when both tags carry the number it expects.  No ordering between two stores is assumed (no
 *               fence, no drain, no flag behind the data); single-copy atomicity of an aligned 8-byte access is;
 *   progress    NOT assumed.  Helpers may never become resident, or die: an owner waits at most 2 ms for an answer, then walks
 *               the helper's columns itself (bit-identical to the answer it did not get) and finishes alone; every spin in the
 *               kernel is bounded, a protocol error ends as a failed run (NYX_HIP_ERR_NAN) or a slow one, never as a hung device;
 *   placement   NOT assumed: no workgroup -> CU / XCD mapping, no dispatch order.
 * (tools/uncached_churn.hip exercises exactly this exchange stand-alone; tests/test_gpu_coop_contexts.py runs 40 cooperative contexts
 *  in one process with and without the pool.) */
int32_t nyx_hip_last_coop_helpers(nyx_hip_ctx *ctx);

/* Elapsed device time (ms) of the kernels launched by the last propagate call on
 * this ctx, measured with HIP events on the launch stream; <0 if unavailable. */
double nyx_hip_last_kernel_ms(nyx_hip_ctx *ctx);

/* Thread-local, NUL-terminated description of the last failure on this thread. */
const char *nyx_hip_last_error(void);

/* ---- host-side helpers that belong to the path (io/gravity.rs) ---- */

/* Parses a GMAT .cof (optionally gzipped) the way GravityFieldData::from_cof does
 * (io/gravity.rs:150-367).  On success c_nm and s_nm receive malloc'ed packed
 * lower-triangular arrays of (degree+1)(degree+2)/2 doubles that the caller frees
 * with nyx_hip_free; out_degree and out_order receive the max degree/order SEEN in the
 * file within the request (from_cof keeps those, io/gravity.rs:330-366). */
int32_t nyx_hip_load_cof(const char *path, int32_t degree, int32_t order, int32_t gunzipped,
                         int32_t *out_degree, int32_t *out_order, double **c_nm, double **s_nm);
/* SHADR flavour (io/gravity.rs:370-501). */
int32_t nyx_hip_load_shadr(const char *path, int32_t degree, int32_t order, int32_t gunzipped,
                           int32_t *out_degree, int32_t *out_order, double **c_nm, double **s_nm);
void nyx_hip_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* NYX_HIP_H */
