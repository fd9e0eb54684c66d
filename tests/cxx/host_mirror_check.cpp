// Compile-and-link check of the C++ host mirror (include/nyx_hip.hpp) against libnyx_hip.so; with a GPU it
// also runs the reference's `val_two_body_dynamics` golden vector (tests/mission_design/orbitaldyn.rs:102-171).
#include <cmath>
#include <cstdio>
#include <vector>

#include "nyx_hip.hpp"

int main() {
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
    return resample_ok && event_ok ? 0 : 1;
}
