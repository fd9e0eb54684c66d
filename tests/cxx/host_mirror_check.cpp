// Compile-and-link check of the C++ host mirror (include/nyx_hip.hpp) against libnyx_hip.so; with a GPU it
// also runs the reference's `val_two_body_dynamics` golden vector (tests/mission_design/orbitaldyn.rs:102-171).
#include <cmath>
#include <cstdio>
#include <vector>

#include "nyx_hip.hpp"
#include "nyx_hip_mc.hpp"

// The reference's seeded known-answer tests of the dispersions (nyx-core/src/mc/multivariate.rs:420-556), host side only:
// Pcg64Mcg::new(0), 1 000 samples, exact counts.
static bool rng_known_answers() {
    const double mu = 398600.4415;
    // Orbit::keplerian(8191.93, 1e-6, 12.85, 306.614, 314.19, 99.8877)
    const double sma = 8191.93, ecc = 1e-6, d2r = 3.14159265358979323846 / 180.0;
    const double inc = 12.85 * d2r, raan = 306.614 * d2r, aop = 314.19 * d2r, ta = 99.8877 * d2r;
    const double p = sma * (1 - ecc * ecc), r = p / (1 + ecc * std::cos(ta));
    const double rp[3] = {r * std::cos(ta), r * std::sin(ta), 0.0};
    const double vp[3] = {-std::sqrt(mu / p) * std::sin(ta), std::sqrt(mu / p) * (ecc + std::cos(ta)), 0.0};
    const double cO = std::cos(raan), sO = std::sin(raan), ci = std::cos(inc), si = std::sin(inc), cw = std::cos(aop), sw = std::sin(aop);
    const double R[3][3] = {{cO * cw - sO * sw * ci, -cO * sw - sO * cw * ci, sO * si},
                            {sO * cw + cO * sw * ci, -sO * sw + cO * cw * ci, -cO * si},
                            {sw * si, cw * si, ci}};
    nyx::Spacecraft t;
    for (int i = 0; i < 3; ++i) {
        t.rv[i] = R[i][0] * rp[0] + R[i][1] * rp[1] + R[i][2] * rp[2];
        t.rv[3 + i] = R[i][0] * vp[0] + R[i][1] * vp[1] + R[i][2] * vp[2];
    }
    // disperse_full_cartesian: components beyond one sigma, over 6: exactly 312
    const std::array<double, 9> sig = {10.0, 10.0, 10.0, 0.2, 0.2, 0.2, 0.0, 0.0, 0.0};
    nyx::MonteCarlo full(nyx::MvnSpacecraft::from_sigmas(t, sig), 0, "disperse_full_cartesian");
    int cnt = 0;
    for (const auto &is : full.generate_states(0, 1000))
        for (int k = 0; k < 6; ++k) cnt += std::fabs(is.second.rv[k] - t.rv[k]) > sig[k] ? 1 : 0;
    // disperse_r_mag: covariance sigma^2 r^ r^T (Jacobian pseudo-inverse of |r|, sigma = 1 km): exactly 6 samples beyond 3 sigma
    const double rm = std::sqrt(t.rv[0] * t.rv[0] + t.rv[1] * t.rv[1] + t.rv[2] * t.rv[2]);
    std::array<double, 81> cov{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) cov[i * 9 + j] = (t.rv[i] / rm) * (t.rv[j] / rm);
    nyx::MonteCarlo rmag(nyx::MvnSpacecraft(t, cov), 0, "disperse_r_mag");
    int too_far = 0;
    for (const auto &is : rmag.generate_states(0, 1000)) {
        const double m = std::sqrt(is.second.rv[0] * is.second.rv[0] + is.second.rv[1] * is.second.rv[1] + is.second.rv[2] * is.second.rv[2]);
        too_far += std::fabs(rm - m) >= 3.0 ? 1 : 0;
    }
    // resume: indices restart after the skip, the states are the tail of the same stream
    const auto all = full.generate_states(0, 6), tail = full.generate_states(4, 2);
    const bool resume_ok = tail.size() == 2 && tail[0].first == 0 && tail[1].first == 1 && tail[0].second.rv[0] == all[4].second.rv[0] &&
                           tail[1].second.rv[5] == all[5].second.rv[5];
    std::printf("dispersion known answers: %d (want 312), %d (want 6), resume %s\n", cnt / 6, too_far, resume_ok ? "ok" : "FAILED");
    return cnt / 6 == 312 && too_far == 6 && resume_ok;
}

int main() {
    if (!rng_known_answers()) return 2;
    nyx_hip_config_t cfg{};
    cfg.abi_version = NYX_HIP_ABI_VERSION;
    cfg.opts = nyx::default_options(NYX_HIP_RK89);
    cfg.central_mu_km3_s2 = 398600.435436096;
    cfg.speed_of_light_km_s = 299792.458;
    nyx_hip_body_t earth{};
    earth.naif_id = 399; earth.mu_km3_s2 = cfg.central_mu_km3_s2; earth.mean_radius_km = 6378.14;
    cfg.n_bodies = 1; cfg.bodies = &earth;
    if (nyx_hip_device_count() == 0) {
        std::printf("no device: link check only\n");
        return 0;
    }
    nyx::GpuPropagator prop(cfg);
    nyx::Spacecraft sc;
    const double init[6] = {-2436.45, -2436.45, 6891.037, 5.088611, -5.088611, 0.0};
    for (int k = 0; k < 6; ++k) sc.rv[k] = init[k];
    auto inst = prop.with(sc);
    nyx::Spacecraft fin = inst.for_duration(86400LL * 1000000000LL);
    const double want[6] = {-5971.194375461378, 3945.517831291771, 2864.6210708007134, 0.04908320163379219, -4.1850841921806206, 5.848947414864886};
    double worst = 0;
    for (int k = 0; k < 6; ++k) worst = std::fmax(worst, std::fabs(fin.rv[k] - want[k]));
    std::printf("max |delta| vs golden = %.3e km\n", worst);
    if (!(worst < 2e-9)) return 1;

    // dense output -> Traj::every / Traj::at, and a stop condition, through the same mirror
    nyx::StateBatch in(2), out(2);
    in.set(0, sc);
    in.set(1, sc);
    nyx::RunStats st(2);
    nyx::TrajBatch traj(2, 256);
    prop.many_for_duration_with_traj(in, 3600LL * 1000000000LL, out, st, traj);
    nyx::TrajBatch minute = traj.every(prop, 60LL * 1000000000LL, 61);
    std::vector<int32_t> status;
    nyx::TrajBatch last = traj.at(prop, {3600LL * 1000000000LL, 3600LL * 1000000000LL + 1}, status);
    const bool resample_ok = minute.len(0) == 61 && minute.epoch_ns(60, 1) == 3600LL * 1000000000LL && status[0] == NYX_HIP_INTERP_OK &&
                             status[2] == NYX_HIP_INTERP_NO_DATA && last.state(0, 0, 0) == out.get(0).rv[0];
    std::printf("every/at: %s (len %d)\n", resample_ok ? "ok" : "FAILED", (int)minute.len(0));
    nyx_hip_event_t apo{NYX_HIP_EV_TRUE_ANOMALY_DEG, 1, 180.0, 1e-7, 1000};
    nyx::TrajBatch etraj(2, 256);
    std::vector<int32_t> crossings;
    prop.many_until_event(in, 3 * 3600LL * 1000000000LL, apo, out, st, etraj, &crossings);
    const bool event_ok = st.status[0] == NYX_HIP_OK && crossings[0] == 1 && out.get(0).epoch_ns > 0 && out.get(0).epoch_ns < 3 * 3600LL * 1000000000LL;
    std::printf("until_event: %s (apoapsis at %.3f s)\n", event_ok ? "ok" : "FAILED", out.get(0).epoch_ns * 1e-9);

    // MonteCarlo::run_until_epoch through the mirror: 8 dispersed runs, each PropResult { state, traj }
    nyx::MonteCarlo mc(nyx::MvnSpacecraft::from_sigmas(sc, {1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3, 0.0, 0.0, 0.0}), 7, "mirror");
    nyx::Results res = mc.run_until_epoch(prop, 1800LL * 1000000000LL, 8, 256);
    bool mc_ok = res.runs.size() == 8;
    for (const auto &run : res.runs)
        mc_ok = mc_ok && run.status == NYX_HIP_OK && run.state.epoch_ns == 1800LL * 1000000000LL &&
                res.traj.state(res.traj.len((int64_t)run.index) - 1, (int64_t)run.index, 0) == run.state.rv[0];
    std::printf("MonteCarlo::run_until_epoch: %s\n", mc_ok ? "ok" : "FAILED");

    // The same Monte Carlo over SEVERAL contexts (nyx_hip_propagate_batch_sharded; here two contexts on device 0, on a node one per
    // device): contiguous index shards, run concurrently, must equal the single-context result bit for bit; the first guess of
    // the trajectory capacity (4) is far too small on purpose: the batch is re-run with what the longest run needs.
    nyx::MultiGpuPropagator multi(cfg, {0, 0});
    nyx::Results sharded = mc.run_until_epoch(multi, 1800LL * 1000000000LL, 8, 4);
    bool multi_ok = sharded.runs.size() == 8 && multi.devices() == 2 && sharded.traj.capacity() >= sharded.traj.len(0);
    for (size_t k = 0; k < sharded.runs.size() && multi_ok; ++k) {
        const auto &a = sharded.runs[k], &b = res.runs[k];
        for (int c = 0; c < 6; ++c) multi_ok = multi_ok && a.state.rv[c] == b.state.rv[c];
        multi_ok = multi_ok && a.status == NYX_HIP_OK && sharded.traj.len((int64_t)k) == res.traj.len((int64_t)k);
        for (int32_t q = 0; q < res.traj.len((int64_t)k) && multi_ok; ++q)
            multi_ok = sharded.traj.epoch_ns(q, (int64_t)k) == res.traj.epoch_ns(q, (int64_t)k) &&
                       sharded.traj.state(q, (int64_t)k, 3) == res.traj.state(q, (int64_t)k, 3);
    }
    std::printf("MultiGpuPropagator (2 contexts, capacity regrown to %d): %s\n", (int)sharded.traj.capacity(), multi_ok ? "ok" : "FAILED");
    return resample_ok && event_ok && mc_ok && multi_ok ? 0 : 1;
}
