// C++ twin of the Rust host's collective step ("Several devices from one process" / the covariance reduction): the call
// sequence a rank-sharded nyx host makes around the C-ABI, WITHOUT PyTorch in the loop -
//     nyx_hip_propagate_batch_device -> nyx_hip_ensemble_moments_device -> ncclAllReduce(55 f64) + ncclAllGather(final states)
// all enqueued on the rank's launch stream - linked against librccl and libnyx_hip.so, one rank per visible device (ncclCommInitAll:
// one here, eight on a node), compared with the single-context result of the whole ensemble.  What it replaces in the reference: the
// rayon par_iter over runs of MonteCarlo::run_until_epoch (mc/montecarlo.rs:233-273) and the statistics over its results
// (mc/results.rs:60-245).  So that the first time RCCL sees N > 1 ranks is not also the first time the C-ABI meets RCCL.
// build: hipcc -std=c++17 -I include tests/cxx/rccl_twin.cpp -L nyx_amd -lnyx_hip -lrccl -o rccl_twin
// usage: rccl_twin [n trajectories = 1000] [ranks = visible devices; more ranks than devices are refused by RCCL]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nyx_hip.h"
#include "nyx_hip.hpp"

#define HIP_OK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) { std::printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)
#define NCCL_OK(x)                                                                       \
    do {                                                                                 \
        ncclResult_t r_ = (x);                                                           \
        if (r_ != ncclSuccess) { std::printf("FAILED %s: %s\n", #x, ncclGetErrorString(r_)); return 1; } \
    } while (0)
#define NYX_OK(x)                                                                        \
    do {                                                                                 \
        int32_t r_ = (x);                                                                \
        if (r_ != NYX_HIP_RC_OK) { std::printf("FAILED %s: rc %d: %s\n", #x, (int)r_, nyx_hip_last_error()); return 1; } \
    } while (0)

static const int NF = 6;   // x y z vx vy vz
static const int NG = 7;   // rows of a rank's gather block: the six, then the epoch as a double (exact below 2^53 ns)

struct Shard {  // the device-resident SoA of one rank
    int dev = 0;
    int64_t lo = 0, n = 0;
    hipStream_t stream = nullptr;
    nyx_hip_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int64_t *epoch_in = nullptr, *epoch_out = nullptr;
    double *in[NF] = {};
    double *blk = nullptr;       // [NG][pad]: this rank's final states, one block = one all-gather
    int32_t *status = nullptr;
    double *mom = nullptr;       // [55]
    double *gathered = nullptr;  // [ranks][NG][pad]
};

__global__ void epoch_to_row(const int64_t *epoch, double *row, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) row[i] = (double)epoch[i];
}

int main(int argc, char **argv) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || nyx_hip_device_count() == 0) {
        std::printf("no device: link check only (rccl header %d.%d)\n", NCCL_MAJOR, NCCL_MINOR);
        return 0;
    }
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 1000;
    const int ranks = argc > 2 ? std::atoi(argv[2]) : ndev;
    const int64_t dur = 3600LL * 1000000000LL;
    // the reference's two-body golden orbit (tests/mission_design/orbitaldyn.rs:102-171), dispersed deterministically
    nyx_hip_config_t cfg{};
    cfg.abi_version = NYX_HIP_ABI_VERSION;
    cfg.opts = nyx::default_options(NYX_HIP_RK89);
    cfg.central_mu_km3_s2 = 398600.435436096;
    cfg.speed_of_light_km_s = 299792.458;
    nyx_hip_body_t earth{};
    earth.naif_id = 399; earth.mu_km3_s2 = cfg.central_mu_km3_s2; earth.mean_radius_km = 6378.14;
    cfg.n_bodies = 1; cfg.bodies = &earth;
    const double nominal[9] = {-2436.45, -2436.45, 6891.037, 5.088611, -5.088611, 0.0, 0.0, 0.0, 0.0};
    std::vector<double> h_in[NF];
    uint64_t lcg = 0x9e3779b97f4a7c15ull;
    for (int f = 0; f < NF; ++f) h_in[f].resize((size_t)n);
    for (int64_t i = 0; i < n; ++i)
        for (int f = 0; f < NF; ++f) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            const double u = (double)(lcg >> 11) / 9007199254740992.0 - 0.5;  // [-0.5, 0.5)
            h_in[f][(size_t)i] = nominal[f] + (f < 3 ? 2.0 : 2e-3) * u;
        }
    std::vector<int64_t> h_epoch((size_t)n, 0);

    // ---- the reference run: ONE context on device 0 over the whole ensemble (host flavours of the same entry points)
    std::vector<double> ref[NF];
    std::vector<int64_t> ref_epoch((size_t)n);
    std::vector<int32_t> ref_status((size_t)n);
    double ref_mom[55], x0[9];
    {
        HIP_OK(hipSetDevice(0));
        nyx_hip_ctx *ctx = nullptr;
        NYX_OK(nyx_hip_ctx_create(&cfg, 0, &ctx));
        for (int f = 0; f < NF; ++f) ref[f].resize((size_t)n);
        nyx_hip_states_t in{}, out{};
        in.n = out.n = n; in.epoch_ns = h_epoch.data(); out.epoch_ns = ref_epoch.data();
        in.x_km = h_in[0].data(); in.y_km = h_in[1].data(); in.z_km = h_in[2].data(); in.vx_km_s = h_in[3].data(); in.vy_km_s = h_in[4].data(); in.vz_km_s = h_in[5].data();
        out.x_km = ref[0].data(); out.y_km = ref[1].data(); out.z_km = ref[2].data(); out.vx_km_s = ref[3].data(); out.vy_km_s = ref[4].data(); out.vz_km_s = ref[5].data();
        nyx_hip_step_stats_t st{};
        st.status = ref_status.data();
        NYX_OK(nyx_hip_propagate_batch(ctx, &in, dur, &out, &st));
        // x0: the first run's final state - what every rank of a sharded host can hold
        for (int f = 0; f < NF; ++f) x0[f] = ref[f][0];
        x0[6] = x0[7] = x0[8] = 0.0;
        NYX_OK(nyx_hip_ensemble_moments(ctx, &out, ref_status.data(), x0, ref_mom));
        nyx_hip_ctx_destroy(ctx);
    }

    // ---- the rank-sharded run: contiguous index shards [k n / m, (k + 1) n / m), one context + one stream + one communicator per rank
    std::vector<Shard> sh((size_t)ranks);
    std::vector<int> devs((size_t)ranks);
    std::vector<ncclComm_t> comms((size_t)ranks);
    for (int k = 0; k < ranks; ++k) devs[(size_t)k] = k % ndev;
    NCCL_OK(ncclCommInitAll(comms.data(), ranks, devs.data()));
    int64_t pad = 0;
    for (int k = 0; k < ranks; ++k) pad = std::max<int64_t>(pad, (k + 1) * n / ranks - k * n / ranks);
    for (int k = 0; k < ranks; ++k) {
        Shard &s = sh[(size_t)k];
        s.dev = devs[(size_t)k]; s.comm = comms[(size_t)k];
        s.lo = k * n / ranks; s.n = (k + 1) * n / ranks - s.lo;
        HIP_OK(hipSetDevice(s.dev));
        HIP_OK(hipStreamCreate(&s.stream));
        NYX_OK(nyx_hip_ctx_create(&cfg, s.dev, &s.ctx));
        HIP_OK(hipMalloc(&s.epoch_in, (size_t)pad * 8)); HIP_OK(hipMalloc(&s.epoch_out, (size_t)pad * 8));
        for (int f = 0; f < NF; ++f) {
            HIP_OK(hipMalloc(&s.in[f], (size_t)pad * 8));
            HIP_OK(hipMemcpy(s.in[f], h_in[f].data() + s.lo, (size_t)s.n * 8, hipMemcpyHostToDevice));
        }
        HIP_OK(hipMemcpy(s.epoch_in, h_epoch.data() + s.lo, (size_t)s.n * 8, hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&s.blk, (size_t)NG * pad * 8)); HIP_OK(hipMemset(s.blk, 0, (size_t)NG * pad * 8));
        HIP_OK(hipMalloc(&s.status, (size_t)pad * 4));
        HIP_OK(hipMalloc(&s.mom, 55 * 8));
        HIP_OK(hipMalloc(&s.gathered, (size_t)ranks * NG * pad * 8));
    }
    // enqueue: propagate -> moments -> (epoch row) on every rank's stream, then the two collectives of the step inside one group
    for (int k = 0; k < ranks; ++k) {
        Shard &s = sh[(size_t)k];
        HIP_OK(hipSetDevice(s.dev));
        nyx_hip_states_t in{}, out{};
        in.n = out.n = s.n; in.epoch_ns = s.epoch_in; out.epoch_ns = s.epoch_out;
        in.x_km = s.in[0]; in.y_km = s.in[1]; in.z_km = s.in[2]; in.vx_km_s = s.in[3]; in.vy_km_s = s.in[4]; in.vz_km_s = s.in[5];
        out.x_km = s.blk + 0 * pad; out.y_km = s.blk + 1 * pad; out.z_km = s.blk + 2 * pad;
        out.vx_km_s = s.blk + 3 * pad; out.vy_km_s = s.blk + 4 * pad; out.vz_km_s = s.blk + 5 * pad;
        nyx_hip_step_stats_t st{};
        st.status = s.status;
        NYX_OK(nyx_hip_propagate_batch_device(s.ctx, &in, dur, &out, &st, s.stream));
        NYX_OK(nyx_hip_ensemble_moments_device(s.ctx, &out, s.status, x0, s.mom, s.stream));
        hipLaunchKernelGGL(epoch_to_row, dim3((unsigned)((s.n + 255) / 256)), dim3(256), 0, s.stream, s.epoch_out, s.blk + 6 * pad, s.n);
    }
    NCCL_OK(ncclGroupStart());
    for (int k = 0; k < ranks; ++k) {
        Shard &s = sh[(size_t)k];
        NCCL_OK(ncclAllReduce(s.mom, s.mom, 55, ncclDouble, ncclSum, s.comm, s.stream));
        NCCL_OK(ncclAllGather(s.blk, s.gathered, (size_t)NG * pad, ncclDouble, s.comm, s.stream));
    }
    NCCL_OK(ncclGroupEnd());
    for (int k = 0; k < ranks; ++k) {
        HIP_OK(hipSetDevice(sh[(size_t)k].dev));
        HIP_OK(hipStreamSynchronize(sh[(size_t)k].stream));
    }

    // ---- compare, on EVERY rank: the gathered ensemble with the single-context run bit for bit (two-body: no column schedule, a shard's
    // bits do not depend on the batch), the all-reduced moments with the single-context moments (one rank: the same launches, bit for
    // bit; several: another summation order, 1e-12 of the scale)
    bool ok = true;
    double worst_mom = 0.0;
    for (int k = 0; k < ranks && ok; ++k) {
        Shard &s = sh[(size_t)k];
        HIP_OK(hipSetDevice(s.dev));
        std::vector<double> g((size_t)ranks * NG * pad), mom(55);
        HIP_OK(hipMemcpy(g.data(), s.gathered, g.size() * 8, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(mom.data(), s.mom, 55 * 8, hipMemcpyDeviceToHost));
        for (int r = 0; r < ranks && ok; ++r) {
            const int64_t lo = r * n / ranks, cnt = (r + 1) * n / ranks - lo;
            for (int64_t i = 0; i < cnt && ok; ++i) {
                for (int f = 0; f < NF; ++f) ok = ok && g[((size_t)r * NG + f) * pad + i] == ref[f][(size_t)(lo + i)];
                ok = ok && (int64_t)g[((size_t)r * NG + 6) * pad + i] == ref_epoch[(size_t)(lo + i)];
            }
        }
        if (!ok) { std::printf("FAILED: rank %d: gathered states differ from the single-context run\n", k); break; }
        for (int q = 0; q < 55; ++q) {
            const double scale = std::fabs(ref_mom[q]) + (q == 0 ? 0.0 : 1e-30);
            const double d = std::fabs(mom[(size_t)q] - ref_mom[q]);
            const double rel = scale > 0.0 ? d / scale : d;
            worst_mom = std::fmax(worst_mom, rel);
            if (ranks == 1 ? d != 0.0 : !(rel < 1e-9 || d < 1e-9)) { ok = false; std::printf("FAILED: rank %d: moment %d: %.17g vs %.17g\n", k, q, mom[(size_t)q], ref_mom[q]); }
        }
    }
    // mean and covariance from the 55 doubles: a sanity line a reader can eyeball
    {
        const double cnt = ref_mom[0];
        const double m0 = x0[0] + ref_mom[1] / cnt;
        const double s00 = ref_mom[10], mu0 = ref_mom[1] / cnt;
        std::printf("ensemble of %lld (ok runs %.0f): mean x = %.6f km, var x = %.6e km^2\n", (long long)n, cnt, m0, (s00 - cnt * mu0 * mu0) / (cnt - 1.0));
    }
    std::printf("rccl twin: %d rank(s) on %d device(s), n = %lld, all-gather of %d x %lld doubles per rank, all-reduce of 55: %s (moments rel. diff %.2e)\n",
                ranks, ndev, (long long)n, NG, (long long)pad, ok ? "ok" : "FAILED", worst_mom);
    for (int k = 0; k < ranks; ++k) {
        Shard &s = sh[(size_t)k];
        (void)hipSetDevice(s.dev);
        nyx_hip_ctx_destroy(s.ctx);
        ncclCommDestroy(s.comm);
    }
    return ok ? 0 : 1;
}
