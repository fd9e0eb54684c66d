// nyx_hip.hpp — header-only C++ host mirror of the reference's operator interface for the batched
// propagation path, on top of the C-ABI (nyx_hip.h).  Rust is not available in this image; this is the
// compiled-language host side a Rust shim would otherwise provide (see INTEGRATION.md for that shim).
//
//   reference                                             here
//   Propagator::new(dynamics, method, opts)               nyx::GpuPropagator(config)
//   prop.with(state, almanac).for_duration(d)             prop.with(state).for_duration(d)
//   prop.with(state, almanac).until_epoch(e)              prop.with(state).until_epoch(e)
//   states.par_iter().map(|s| prop.with(s).for_duration)  prop.many_for_duration(batch, d)
//   prop.with(state, almanac).for_duration_with_traj(d)   prop.many_for_duration_with_traj(batch, d, ...) -> nyx::TrajBatch
//   traj.at(epoch) / traj.every(step)                     trajs.at(prop, epochs, ...) / trajs.every(prop, step, capacity)
//   prop.with(..).until_nth_event(max, &event, None, n)   prop.many_until_event(batch, max, event, ...)
//   odp.predict_until(estimate, end_epoch)                prop.predict_until(batch, cfg, covar, ...)
//   (propagators/propagator.rs:34-121, instance.rs:62-352, event.rs:88-211, md/trajectory/traj.rs:82-162,
//    mc/montecarlo.rs:93-253, od/process/mod.rs:440-486)
#ifndef NYX_HIP_HPP
#define NYX_HIP_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "nyx_hip.h"

namespace nyx {

struct PropagationError : std::runtime_error {
    int32_t status;
    int64_t index;
    PropagationError(int32_t st, int64_t idx)
        : std::runtime_error("run " + std::to_string(idx) + ": status " + std::to_string(st)), status(st), index(idx) {}
};

// IntegratorOptions (propagators/options.rs:42-61, defaults :172-186)
inline nyx_hip_integ_opts_t default_options(int32_t method = NYX_HIP_RK89) {
    return nyx_hip_integ_opts_t{60LL * 1000000000LL, 1000000LL, 2700LL * 1000000000LL, 1e-12, 50, 0, NYX_HIP_RSS_CARTESIAN_STEP, method};
}
inline nyx_hip_integ_opts_t with_fixed_step(int64_t step_ns, int32_t method = NYX_HIP_RK89) {
    return nyx_hip_integ_opts_t{step_ns, step_ns, step_ns, 0.0, 0, 1, NYX_HIP_RSS_CARTESIAN_STEP, method};
}
inline nyx_hip_integ_opts_t with_adaptive_step(int64_t min_ns, int64_t max_ns, double tol, int32_t error_ctrl, int32_t method = NYX_HIP_RK89) {
    return nyx_hip_integ_opts_t{max_ns, min_ns, max_ns, tol, 50, 0, error_ctrl, method};
}

// One Spacecraft (cosmic/spacecraft.rs:115-143), the fields this path reads.
struct Spacecraft {
    int64_t epoch_ns = 0;
    double rv[6] = {0, 0, 0, 0, 0, 0};
    double cr = 1.8, cd = 2.2, prop_mass_kg = 0, dry_mass_kg = 0, extra_mass_kg = 0, srp_area_m2 = 0, drag_area_m2 = 0;
};

// SoA batch owning its storage; `view()` is what crosses the ABI.
class StateBatch {
  public:
    explicit StateBatch(int64_t n) : n_(n), epoch_(n), step_(n, 0) {
        for (auto &f : f_) f.assign((size_t)n, 0.0);
    }
    int64_t size() const { return n_; }
    void set(int64_t i, const Spacecraft &s) {
        epoch_[i] = s.epoch_ns;
        for (int k = 0; k < 6; ++k) f_[k][i] = s.rv[k];
        f_[6][i] = s.cr; f_[7][i] = s.cd; f_[8][i] = s.prop_mass_kg; f_[9][i] = s.dry_mass_kg;
        f_[10][i] = s.extra_mass_kg; f_[11][i] = s.srp_area_m2; f_[12][i] = s.drag_area_m2;
    }
    Spacecraft get(int64_t i) const {
        Spacecraft s;
        s.epoch_ns = epoch_[i];
        for (int k = 0; k < 6; ++k) s.rv[k] = f_[k][i];
        s.cr = f_[6][i]; s.cd = f_[7][i]; s.prop_mass_kg = f_[8][i]; s.dry_mass_kg = f_[9][i];
        s.extra_mass_kg = f_[10][i]; s.srp_area_m2 = f_[11][i]; s.drag_area_m2 = f_[12][i];
        return s;
    }
    int64_t &step_ns(int64_t i) { return step_[i]; }
    nyx_hip_states_t view() {
        nyx_hip_states_t v{};
        v.n = n_; v.epoch_ns = epoch_.data();
        v.x_km = f_[0].data(); v.y_km = f_[1].data(); v.z_km = f_[2].data();
        v.vx_km_s = f_[3].data(); v.vy_km_s = f_[4].data(); v.vz_km_s = f_[5].data();
        v.cr = f_[6].data(); v.cd = f_[7].data(); v.prop_mass_kg = f_[8].data(); v.dry_mass_kg = f_[9].data();
        v.extra_mass_kg = f_[10].data(); v.srp_area_m2 = f_[11].data(); v.drag_area_m2 = f_[12].data();
        v.stm = nullptr; v.step_ns = step_.data();
        return v;
    }

  private:
    int64_t n_;
    std::vector<int64_t> epoch_, step_;
    std::vector<double> f_[13];
};

struct RunStats {
    std::vector<int32_t> status, last_attempts;
    std::vector<int64_t> last_step_ns, n_accepted, n_rejected, n_evals;
    std::vector<double> last_error;
    explicit RunStats(int64_t n) : status(n), last_attempts(n), last_step_ns(n), n_accepted(n), n_rejected(n), n_evals(n), last_error(n) {}
    nyx_hip_step_stats_t view() {
        return nyx_hip_step_stats_t{status.data(), last_step_ns.data(), last_error.data(), last_attempts.data(),
                                    n_accepted.data(), n_rejected.data(), n_evals.data()};
    }
};

class GpuPropagator;

// Traj<Spacecraft> of every run of a batch (md/trajectory/traj.rs:40-162): step-major storage, as the ABI.
class TrajBatch {
  public:
    TrajBatch(int64_t n, int64_t capacity) : n_(n), cap_(capacity), epoch_((size_t)(n * capacity)), len_((size_t)n, 0) {
        for (auto &f : f_) f.assign((size_t)(n * capacity), 0.0);
    }
    int64_t size() const { return n_; }
    int64_t capacity() const { return cap_; }
    int32_t len(int64_t i) const { return len_[i]; }
    int64_t epoch_ns(int64_t k, int64_t i) const { return epoch_[(size_t)(k * n_ + i)]; }
    double state(int64_t k, int64_t i, int c) const { return f_[c][(size_t)(k * n_ + i)]; }
    nyx_hip_traj_t view() {
        return nyx_hip_traj_t{cap_, epoch_.data(), f_[0].data(), f_[1].data(), f_[2].data(), f_[3].data(), f_[4].data(), f_[5].data(),
                              len_.data()};
    }
    // Traj::at for every run at shared epochs; status[q * n + i] = nyx_hip_interp_status
    TrajBatch at(GpuPropagator &prop, const std::vector<int64_t> &epochs_ns, std::vector<int32_t> &status);
    // Traj::every(step) for every run
    TrajBatch every(GpuPropagator &prop, int64_t step_ns, int64_t capacity);

  private:
    int64_t n_, cap_;
    std::vector<int64_t> epoch_;
    std::vector<double> f_[6];
    std::vector<int32_t> len_;
};

// PropInstance (propagators/instance.rs:62-352) for one state: a batch of one on the device.
class PropInstance {
  public:
    Spacecraft state;
    int64_t step_size_ns;
    nyx_hip_step_stats_t *unused_ = nullptr;
    PropInstance(GpuPropagator &p, const Spacecraft &s, int64_t init_step) : state(s), step_size_ns(init_step), prop_(p) {}
    Spacecraft for_duration(int64_t duration_ns);
    Spacecraft until_epoch(int64_t end_epoch_ns) { return for_duration(end_epoch_ns - state.epoch_ns); }

  private:
    GpuPropagator &prop_;
};

class GpuPropagator {
  public:
    explicit GpuPropagator(const nyx_hip_config_t &cfg, int device = 0) : init_step_(cfg.opts.init_step_ns) {
        if (nyx_hip_ctx_create(&cfg, device, &ctx_) != NYX_HIP_RC_OK) throw std::runtime_error(nyx_hip_last_error());
    }
    ~GpuPropagator() { nyx_hip_ctx_destroy(ctx_); }
    GpuPropagator(const GpuPropagator &) = delete;
    GpuPropagator &operator=(const GpuPropagator &) = delete;

    PropInstance with(const Spacecraft &s) { return PropInstance(*this, s, init_step_); }

    // the rayon par_iter of mc/montecarlo.rs:233-253 / nyx-py many_for_duration: failed runs stay in place with a status
    void many_for_duration(StateBatch &in, int64_t duration_ns, StateBatch &out, RunStats &stats) {
        nyx_hip_states_t vi = in.view(), vo = out.view();
        nyx_hip_step_stats_t vs = stats.view();
        if (nyx_hip_propagate_batch(ctx_, &vi, duration_ns, &vo, &vs) != NYX_HIP_RC_OK) throw std::runtime_error(nyx_hip_last_error());
    }
    void many_until_epoch(StateBatch &in, int64_t end_epoch_ns, StateBatch &out, RunStats &stats) {
        nyx_hip_states_t vi = in.view(), vo = out.view();
        nyx_hip_step_stats_t vs = stats.view();
        if (nyx_hip_propagate_until_epoch(ctx_, &vi, end_epoch_ns, &vo, &vs) != NYX_HIP_RC_OK) throw std::runtime_error(nyx_hip_last_error());
    }
    // for_duration_with_traj for the batch (instance.rs:297-326)
    void many_for_duration_with_traj(StateBatch &in, int64_t duration_ns, StateBatch &out, RunStats &stats, TrajBatch &traj) {
        nyx_hip_states_t vi = in.view(), vo = out.view();
        nyx_hip_step_stats_t vs = stats.view();
        nyx_hip_traj_t vt = traj.view();
        if (nyx_hip_propagate_batch_with_traj(ctx_, &vi, duration_ns, &vo, &vs, &vt) != NYX_HIP_RC_OK) throw std::runtime_error(nyx_hip_last_error());
    }
    // until_nth_event for the batch (event.rs:88-211; MonteCarlo::run_until_nth_event, montecarlo.rs:93-186): `out` holds the
    // states at the event, stats.status NYX_HIP_ERR_EVENT_NOT_FOUND where max_duration elapsed first
    void many_until_event(StateBatch &in, int64_t max_duration_ns, const nyx_hip_event_t &event, StateBatch &out, RunStats &stats,
                          TrajBatch &traj, std::vector<int32_t> *crossings = nullptr) {
        nyx_hip_states_t vi = in.view(), vo = out.view();
        nyx_hip_step_stats_t vs = stats.view();
        nyx_hip_traj_t vt = traj.view();
        if (crossings) crossings->assign((size_t)in.size(), 0);
        if (nyx_hip_propagate_until_event(ctx_, &vi, max_duration_ns, &event, &vo, &vs, &vt, crossings ? crossings->data() : nullptr) !=
            NYX_HIP_RC_OK)
            throw std::runtime_error(nyx_hip_last_error());
    }
    // KalmanODProcess::predict_until for the batch (od/process/mod.rs:440-486); covar: n * 81, column-major, in/out
    void predict_until(StateBatch &in, const nyx_hip_predict_t &cfg, std::vector<double> &covar, StateBatch &out, RunStats &stats,
                       nyx_hip_predict_history_t *history = nullptr, std::vector<double> *state_deviation = nullptr) {
        nyx_hip_states_t vi = in.view(), vo = out.view();
        nyx_hip_step_stats_t vs = stats.view();
        nyx_hip_estimates_t est{covar.data(), state_deviation ? state_deviation->data() : nullptr};
        if (nyx_hip_predict_until(ctx_, &vi, &cfg, &est, &vo, &vs, history) != NYX_HIP_RC_OK) throw std::runtime_error(nyx_hip_last_error());
    }
    nyx_hip_ctx *raw() { return ctx_; }

  private:
    nyx_hip_ctx *ctx_ = nullptr;
    int64_t init_step_;
};

// The same propagator on SEVERAL devices of one node, driven by one process: one context per device, a batch is cut into
// contiguous index shards, the devices work concurrently (nyx_hip_propagate_batch_sharded).  `devices` empty => every device
// nyx_hip_device_count() reports; a device may be named twice (two contexts on it: how the 1-GPU tests exercise this).
class MultiGpuPropagator {
  public:
    explicit MultiGpuPropagator(const nyx_hip_config_t &cfg, std::vector<int> devices = {}) {
        if (devices.empty())
            for (int d = 0; d < nyx_hip_device_count(); ++d) devices.push_back(d);
        if (devices.empty()) throw std::runtime_error("no HIP device");
        for (int d : devices) {
            nyx_hip_ctx *c = nullptr;
            if (nyx_hip_ctx_create(&cfg, d, &c) != NYX_HIP_RC_OK) {
                const std::string msg = nyx_hip_last_error();
                for (nyx_hip_ctx *q : ctxs_) nyx_hip_ctx_destroy(q);
                throw std::runtime_error(msg);
            }
            ctxs_.push_back(c);
        }
    }
    ~MultiGpuPropagator() { for (nyx_hip_ctx *c : ctxs_) nyx_hip_ctx_destroy(c); }
    MultiGpuPropagator(const MultiGpuPropagator &) = delete;
    MultiGpuPropagator &operator=(const MultiGpuPropagator &) = delete;
    int devices() const { return (int)ctxs_.size(); }

    void many_for_duration(StateBatch &in, int64_t duration_ns, StateBatch &out, RunStats &stats) {
        nyx_hip_states_t vi = in.view(), vo = out.view();
        nyx_hip_step_stats_t vs = stats.view();
        if (nyx_hip_propagate_batch_sharded(ctxs_.data(), (int32_t)ctxs_.size(), &vi, duration_ns, &vo, &vs, nullptr) != NYX_HIP_RC_OK)
            throw std::runtime_error(nyx_hip_last_error());
    }
    void many_for_duration_with_traj(StateBatch &in, int64_t duration_ns, StateBatch &out, RunStats &stats, TrajBatch &traj) {
        nyx_hip_states_t vi = in.view(), vo = out.view();
        nyx_hip_step_stats_t vs = stats.view();
        nyx_hip_traj_t vt = traj.view();
        if (nyx_hip_propagate_batch_sharded(ctxs_.data(), (int32_t)ctxs_.size(), &vi, duration_ns, &vo, &vs, &vt) != NYX_HIP_RC_OK)
            throw std::runtime_error(nyx_hip_last_error());
    }

  private:
    std::vector<nyx_hip_ctx *> ctxs_;
};

inline TrajBatch TrajBatch::at(GpuPropagator &prop, const std::vector<int64_t> &epochs_ns, std::vector<int32_t> &status) {
    const int64_t m = (int64_t)epochs_ns.size();
    TrajBatch out(n_, m > 0 ? m : 1);
    status.assign((size_t)((m > 0 ? m : 1) * n_), 0);
    nyx_hip_traj_t vi = view(), vo = out.view();
    if (nyx_hip_traj_at(prop.raw(), &vi, n_, epochs_ns.data(), m, &vo, status.data()) != NYX_HIP_RC_OK) throw std::runtime_error(nyx_hip_last_error());
    return out;
}

inline TrajBatch TrajBatch::every(GpuPropagator &prop, int64_t step_ns, int64_t capacity) {
    TrajBatch out(n_, capacity);
    nyx_hip_traj_t vi = view(), vo = out.view();
    if (nyx_hip_traj_every(prop.raw(), &vi, n_, step_ns, &vo) != NYX_HIP_RC_OK) throw std::runtime_error(nyx_hip_last_error());
    return out;
}

inline Spacecraft PropInstance::for_duration(int64_t duration_ns) {
    StateBatch b(1), o(1);
    b.set(0, state);
    b.step_ns(0) = step_size_ns;
    RunStats st(1);
    prop_.many_for_duration(b, duration_ns, o, st);
    if (st.status[0] != NYX_HIP_OK) throw PropagationError(st.status[0], 0);
    step_size_ns = o.step_ns(0);
    state = o.get(0);
    return state;
}

}  // namespace nyx
#endif
