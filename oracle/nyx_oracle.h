/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, operation-for-operation) of the reference's
 * Propagator / SpacecraftDynamics hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (nyx_amd/, include/) never does.
 *
 * Parity status: the integrator, error control, two-body term and STM assembly are
 * PINNED against the reference's own golden vectors (the JSON files under tests/golden/, taken
 * from nyx-core/tests/propagation/propagators.rs and
 * tests/mission_design/orbitaldyn.rs).  Pieces whose algorithm lives in the absent
 * `anise` crate (0.10.2; SPK type-2 evaluation, IAU rotation, eclipse geometry) and
 * `hifitime` 4.3 (f64 s -> integer-ns conversion) are restated from their published
 * definitions and are "parity unpinned" against the real crates; see DESIGN.md.
 *
 * It consumes the SAME plain-data descriptors as the product C-ABI
 * (include/nyx_hip.h) so that a test feeds identical inputs to both.
 */
#ifndef NYX_ORACLE_H
#define NYX_ORACLE_H

#include "../include/nyx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Batch propagation on `n_threads` host threads (the analogue of rayon par_iter,
 * mc/montecarlo.rs:233).  Same contract as nyx_hip_propagate_batch with host arrays. */
int32_t nyx_oracle_propagate_batch(const nyx_hip_config_t *cfg, const nyx_hip_states_t *in, int64_t duration_ns,
                                   nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, int32_t n_threads);

/* Same, also recording the accepted states (for_duration_with_traj, instance.rs:297-326); traj may be NULL. */
int32_t nyx_oracle_propagate_batch_traj(const nyx_hip_config_t *cfg, const nyx_hip_states_t *in, int64_t duration_ns,
                                        nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, const nyx_hip_traj_t *traj,
                                        int32_t n_threads);

/* One call of SpacecraftDynamics::eom (dynamics/spacecraft.rs:191-310) for a single
 * state: y is the 9- (no STM) or 90-vector, dydt likewise.  ctx_stm is the
 * step-start STM (column-major 81) or NULL.  Returns a nyx_hip_status. */
int32_t nyx_oracle_eom(const nyx_hip_config_t *cfg, int64_t ctx_epoch_ns, double delta_t_s, const double *y,
                       const double *ctx_stm, double dry_mass_kg, double extra_mass_kg, double srp_area_m2,
                       double drag_area_m2, double *dydt);

/* SpacecraftDynamics::dual_eom (dynamics/spacecraft.rs:312-363): f(x) (9) and A = df/dx (9x9, column-major). */
int32_t nyx_oracle_dual_eom(const nyx_hip_config_t *cfg, int64_t epoch_ns, const double *y9, double dry_mass_kg,
                            double extra_mass_kg, double srp_area_m2, double *fx9, double *grad81);

/* Individual model terms, for unit tests. */
void nyx_oracle_body_position(const nyx_hip_config_t *cfg, int32_t body, int64_t epoch_ns, double *r3, int32_t *status);
/* `segments`: config.segments (needed by the Euler/Chebyshev kind; may be NULL for IAU); `w_rate` optional (rad/s). */
int32_t nyx_oracle_rotation_dcm(const nyx_hip_rotation_t *rot, const nyx_hip_cheby_segment_t *segments, int64_t epoch_ns, double *dcm9_rowmajor, double *w_rate);
void nyx_oracle_gravity_accel(const nyx_hip_gravity_field_t *g, int64_t epoch_ns, const double *r3, double *a3);
/* SolidTides::eom / gradient (dynamics/solid_tides.rs:238-559); grad and the delta tables ([n][m], 4x4) are optional. */
int32_t nyx_oracle_tides_accel(const nyx_hip_config_t *cfg, int64_t epoch_ns, const double *r3, double *a3, double *grad9_rowmajor,
                               double *dc16, double *ds16);
double nyx_oracle_occultation_factor(const nyx_hip_config_t *cfg, int32_t eclipsing_body, int32_t sun_body,
                                     int64_t epoch_ns, const double *r3, int32_t *status);
double nyx_oracle_error_estimate(int32_t error_ctrl, int32_t nv, const double *err, const double *cand, const double *cur);

/* Traj evaluation (md/trajectory/traj.rs:82-162, interpolatable.rs:52-108); statuses are nyx_hip_interp_status. */
int32_t nyx_oracle_hermite_eval(const double *xs, const double *ys, const double *ydots, int32_t n, double x_eval,
                                double *f, double *df);
int32_t nyx_oracle_traj_at(const nyx_hip_traj_t *traj, int64_t n, int64_t i, int64_t epoch_ns, double *state6);
int32_t nyx_oracle_traj_window_ill(const nyx_hip_traj_t *traj, int64_t n, int64_t i, int64_t epoch_ns);
int32_t nyx_oracle_traj_every(const nyx_hip_traj_t *traj, int64_t n, int64_t step_ns, nyx_hip_traj_t *out);

/* PropInstance::until_nth_event (propagators/event.rs:88-211): oracle twin of nyx_hip_propagate_until_event. */
int32_t nyx_oracle_until_event(const nyx_hip_config_t *cfg, const nyx_hip_states_t *in, int64_t max_duration_ns,
                               const nyx_hip_event_t *ev, nyx_hip_states_t *out, nyx_hip_step_stats_t *stats,
                               nyx_hip_traj_t *traj, int32_t *crossings);

/* KalmanODProcess::predict_until (od/process/mod.rs:440-486): oracle twin of nyx_hip_predict_until (one thread). */
int32_t nyx_oracle_predict_until(const nyx_hip_config_t *cfg, const nyx_hip_states_t *in, const nyx_hip_predict_t *pc,
                                 nyx_hip_estimates_t *est, nyx_hip_states_t *out, nyx_hip_step_stats_t *stats,
                                 nyx_hip_predict_history_t *hist);

/* hifitime conversions as restated (see nyx_oracle.c). */
int64_t nyx_oracle_seconds_to_ns(double s);
double nyx_oracle_ns_to_seconds(int64_t ns);
/* 0 = truncate toward zero (`as i64`, default), 1 = round half away from zero. */
void nyx_oracle_set_ns_rounding(int32_t mode);

#ifdef __cplusplus
}
#endif
#endif
