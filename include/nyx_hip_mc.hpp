/* nyx_hip_mc.hpp — C++ host mirror of the Monte Carlo front end over the batch ABI (header-only).
 *
 * What a Rust maintainer keeps from nyx-core/src/mc: `MonteCarlo { random_state, seed, scenario }`
 * (montecarlo.rs:44-75), `generate_states(skip, num_runs, seed)` with ONE seeded stream (:277-296),
 * `run_until_epoch` (:189-273) - where the reference hands the dispersed states to a rayon par_iter of
 * `until_epoch_with_traj`, this hands the batch to `nyx_hip_propagate_batch_with_traj`.
 *
 * The random stream is the reference's: rand_pcg::Pcg64Mcg::new(seed) and rand_distr's ziggurat StandardNormal, nine
 * draws per state (mc/multivariate.rs:298-303).  Both crates are crates.io dependencies that are not part of the
 * reference tree; their published algorithms are restated here exactly as in nyx_amd/rng.py, and pinned by the
 * reference's own seeded known-answer tests (multivariate.rs:420-556; tests/cxx/host_mirror_check.cpp re-runs them).
 */
#ifndef NYX_HIP_MC_HPP
#define NYX_HIP_MC_HPP

#include <array>
#include <cmath>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "nyx_hip.hpp"

namespace nyx {

// rand_pcg::Pcg64Mcg (Mcg128Xsl64): state <- state * M mod 2^128, output = rotr64(hi ^ lo, state >> 122); new(seed) = seed | 1.
class Pcg64Mcg {
  public:
    explicit Pcg64Mcg(unsigned __int128 seed) : state_(seed | 1u) {}
    uint64_t next_u64() {
        const unsigned __int128 mult = ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
        state_ *= mult;
        const unsigned rot = (unsigned)(state_ >> 122);
        const uint64_t xsl = (uint64_t)(state_ >> 64) ^ (uint64_t)state_;
        return (xsl >> rot) | (xsl << ((64 - rot) & 63));
    }
    double random_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }  // rng.random::<f64>(): [0, 1)
    double open01() {                                                                         // Open01: (0, 1)
        const uint64_t frac = next_u64() >> 12;
        return (1.0 + (double)frac * 2.220446049250313e-16) - (1.0 - 1.1102230246251565e-16);
    }
    // rand_distr::StandardNormal: 256-layer ziggurat, symmetric; tail by Marsaglia's exponential rejection
    double standard_normal() {
        const Tables &t = tables();
        for (;;) {
            const uint64_t bits = next_u64();
            const unsigned i = (unsigned)(bits & 0xff);
            const double u = (2.0 + (double)(bits >> 12) * 4.440892098500626e-16) - 3.0;  // [2, 4) from the top 52 bits, minus 3
            const double x = u * t.x[i];
            if (std::fabs(x) < t.x[i + 1]) return x;
            if (i == 0) {
                double xx = 1.0, yy = 0.0;
                while (-2.0 * yy < xx * xx) {
                    const double a = open01(), b = open01();
                    xx = std::log(a) / R;
                    yy = std::log(b);
                }
                return u < 0.0 ? xx - R : R - xx;
            }
            if (t.f[i + 1] + (t.f[i] - t.f[i + 1]) * random_f64() < std::exp(-x * x / 2.0)) return x;
        }
    }

  private:
    static constexpr double R = 3.654152885361008796, V = 0.00492867323399;
    struct Tables {
        double x[257], f[257];
    };
    static const Tables &tables() {
        static const Tables t = [] {
            Tables q;
            auto pdf = [](double v) { return std::exp(-v * v / 2.0); };
            q.x[0] = V / pdf(R);
            q.x[1] = R;
            for (int i = 2; i < 256; ++i) q.x[i] = std::sqrt(-2.0 * std::log(V / q.x[i - 1] + pdf(q.x[i - 1])));
            q.x[256] = 0.0;
            for (int i = 0; i < 257; ++i) q.f[i] = pdf(q.x[i]);
            return q;
        }();
        return t;
    }
    unsigned __int128 state_;
};

// MvnSpacecraft for a covariance given in the state space [x y z vx vy vz Cr Cd prop-mass] (multivariate.rs:228-312): x = L z + mean,
// L = V sqrt(S).  A diagonal covariance maps component k to draw k; a general symmetric PSD one is factored by cyclic Jacobi
// (the orientation of L for repeated singular values is nalgebra's in the reference and unpinned).
class MvnSpacecraft {
  public:
    MvnSpacecraft(const Spacecraft &template_state, const std::array<double, 81> &cov_row_major, const std::array<double, 9> &mean = {})
        : template_(template_state), mean_(mean) {
        bool diagonal = true;
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j)
                if (i != j && cov_row_major[i * 9 + j] != 0.0) diagonal = false;
        l_.fill(0.0);
        if (diagonal) {
            for (int k = 0; k < 9; ++k) l_[k * 9 + k] = std::sqrt(cov_row_major[k * 9 + k]);
            return;
        }
        std::array<double, 81> a = cov_row_major, v{};
        for (int k = 0; k < 9; ++k) v[k * 9 + k] = 1.0;
        for (int sweep = 0; sweep < 60; ++sweep) {
            double off = 0.0;
            for (int p = 0; p < 9; ++p)
                for (int q = p + 1; q < 9; ++q) off += a[p * 9 + q] * a[p * 9 + q];
            if (off < 1e-300) break;
            for (int p = 0; p < 9; ++p)
                for (int q = p + 1; q < 9; ++q) {
                    if (a[p * 9 + q] == 0.0) continue;
                    const double theta = (a[q * 9 + q] - a[p * 9 + p]) / (2.0 * a[p * 9 + q]);
                    const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    const double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
                    for (int k = 0; k < 9; ++k) {
                        const double akp = a[k * 9 + p], akq = a[k * 9 + q];
                        a[k * 9 + p] = c * akp - s * akq;
                        a[k * 9 + q] = s * akp + c * akq;
                    }
                    for (int k = 0; k < 9; ++k) {
                        const double apk = a[p * 9 + k], aqk = a[q * 9 + k];
                        a[p * 9 + k] = c * apk - s * aqk;
                        a[q * 9 + k] = s * apk + c * aqk;
                    }
                    for (int k = 0; k < 9; ++k) {
                        const double vkp = v[k * 9 + p], vkq = v[k * 9 + q];
                        v[k * 9 + p] = c * vkp - s * vkq;
                        v[k * 9 + q] = s * vkp + c * vkq;
                    }
                }
        }
        // singular values in descending order (what LAPACK returns, and where the reference's seeded |r| test finds the one
        // non-zero direction: column 0)
        int order[9];
        for (int j = 0; j < 9; ++j) order[j] = j;
        for (int i = 1; i < 9; ++i)
            for (int j = i; j > 0 && a[order[j] * 9 + order[j]] > a[order[j - 1] * 9 + order[j - 1]]; --j) std::swap(order[j], order[j - 1]);
        for (int j = 0; j < 9; ++j) {
            const int src = order[j];
            const double sv = a[src * 9 + src] > 0.0 ? std::sqrt(a[src * 9 + src]) : 0.0;
            for (int i = 0; i < 9; ++i) l_[i * 9 + j] = v[i * 9 + src] * sv;
        }
    }
    static MvnSpacecraft from_sigmas(const Spacecraft &template_state, const std::array<double, 9> &sigmas) {
        std::array<double, 81> cov{};
        for (int k = 0; k < 9; ++k) cov[k * 9 + k] = sigmas[k] * sigmas[k];
        return MvnSpacecraft(template_state, cov);
    }
    // multivariate.rs:298-318: nine draws, then position += x[0..3], velocity += x[3..6], Cr += x[6], Cd += x[7], prop mass += x[8]
    Spacecraft sample(Pcg64Mcg &rng) const {
        double z[9];
        for (double &zz : z) zz = rng.standard_normal();
        Spacecraft s = template_;
        for (int i = 0; i < 9; ++i) {
            double x = mean_[i];
            for (int j = 0; j < 9; ++j) x += l_[i * 9 + j] * z[j];
            if (i < 6) s.rv[i] += x;
            else if (i == 6) s.cr += x;
            else if (i == 7) s.cd += x;
            else s.prop_mass_kg += x;
        }
        return s;
    }
    const Spacecraft &template_state() const { return template_; }

  private:
    Spacecraft template_;
    std::array<double, 9> mean_;
    std::array<double, 81> l_;
};

struct Run {  // results.rs:48-59
    size_t index;
    Spacecraft dispersed_state;
    int32_t status;    // NYX_HIP_OK or the propagation error of this run
    Spacecraft state;  // final state (PropResult::state); the trajectory is row `index` of Results::traj
};

struct Results {  // results.rs:60-71
    std::vector<Run> runs;
    std::string scenario;
    TrajBatch traj;
};

class MonteCarlo {  // montecarlo.rs:44-75
  public:
    MonteCarlo(MvnSpacecraft random_state, unsigned __int128 seed, std::string scenario = "MonteCarlo")
        : random_state_(std::move(random_state)), seed_(seed), scenario_(std::move(scenario)) {}

    // montecarlo.rs:277-296: sample_iter(rng).skip(skip).take(num_runs).enumerate() - indices restart after the skip
    std::vector<std::pair<size_t, Spacecraft>> generate_states(size_t skip, size_t num_runs) const {
        Pcg64Mcg rng(seed_);
        std::vector<std::pair<size_t, Spacecraft>> out;
        for (size_t k = 0; k < skip + num_runs; ++k) {
            Spacecraft s = random_state_.sample(rng);
            if (k >= skip) out.emplace_back(k - skip, s);
        }
        return out;
    }

    // montecarlo.rs:189-273: every run is until_epoch_with_traj.  `capacity` = accepted steps recorded per run, a first guess:
    // the kernel stops STORING at the capacity but keeps counting (traj.len), so a run that needed more is seen and the batch is
    // re-run with what the longest run needs - no caller ever sees a clipped trajectory (a 3-day low-LEO run takes > 4 096 steps).
    // P = GpuPropagator (one device) or MultiGpuPropagator (the devices of the node, contiguous index shards).
    template <typename P>
    Results resume_run_until_epoch(P &prop, size_t skip, int64_t end_epoch_ns, size_t num_runs, int64_t capacity = 4096) const {
        const auto states = generate_states(skip, num_runs);
        const int64_t n = (int64_t)states.size();
        StateBatch in(n), out(n);
        for (size_t k = 0; k < states.size(); ++k) in.set((int64_t)k, states[k].second);
        RunStats st(n);
        const int64_t duration = end_epoch_ns - random_state_.template_state().epoch_ns;
        int64_t cap = capacity > 0 ? capacity : 1;
        for (;;) {
            Results res{{}, scenario_, TrajBatch(n, cap)};
            prop.many_for_duration_with_traj(in, duration, out, st, res.traj);
            int64_t need = 0;
            for (int64_t i = 0; i < n; ++i) need = res.traj.len(i) > need ? res.traj.len(i) : need;
            if (need > cap) { cap = need; continue; }
            for (size_t k = 0; k < states.size(); ++k)
                res.runs.push_back(Run{states[k].first, states[k].second, st.status[k], out.get((int64_t)k)});
            return res;
        }
    }
    template <typename P>
    Results run_until_epoch(P &prop, int64_t end_epoch_ns, size_t num_runs, int64_t capacity = 4096) const {
        return resume_run_until_epoch(prop, 0, end_epoch_ns, num_runs, capacity);
    }

  private:
    MvnSpacecraft random_state_;
    unsigned __int128 seed_;
    std::string scenario_;
};

}  // namespace nyx
#endif
