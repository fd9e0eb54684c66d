// col_partition_check.cpp - CPU check of nyx_place_runs (nyx_amd/csrc/col_partition.h), the free-order placement of the owner's column
// runs.  Built and run by tests/test_col_partition.py with g++ (no HIP, no GPU).  Prints one line per case; exit code 0 = all good.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "../../nyx_amd/csrc/col_partition.h"

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++failures; } } while (0)

// what the linear partition of round 4 does with the same inputs (targets in descending order, whole columns at the cumulative targets):
// the largest |load - target| / weight it leaves
static double linear_worst(const std::vector<double> &cost, std::vector<double> tg, std::vector<double> wg) {
    std::vector<int> ord(tg.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return tg[a] > tg[b]; });
    double cum_t = 0.0, cum_r = 0.0, worst = 0.0;
    size_t k = 0;
    for (size_t q = 0; q < ord.size(); ++q) {
        const int w = ord[q];
        cum_t += tg[w];
        double load = 0.0;
        const bool last = q + 1 == ord.size();
        while (k < cost.size() && (last || cum_r + 0.5 * cost[k] <= cum_t)) { cum_r += cost[k]; load += cost[k]; ++k; }
        worst = std::max(worst, std::fabs(load - tg[w]) / wg[w]);
    }
    return worst;
}

static void check_case(const char *name, const std::vector<double> &cost, const std::vector<double> &tg, const std::vector<double> &wg, bool expect_found) {
    std::vector<int> sw, sk;
    double tol = -1.0;
    const bool found = nyx_place_runs(cost, tg, wg, sw, sk, &tol);
    CHECK(found == expect_found, "%s: found = %d", name, (int)found);
    if (!found) { std::printf("%-28s no placement (expected: %d)\n", name, (int)!expect_found); return; }
    const int na = (int)tg.size(), m = (int)cost.size();
    CHECK((int)sw.size() == na && (int)sk.size() == na, "%s: %zu waves placed of %d", name, sw.size(), na);
    std::vector<int> seen(na, 0);
    double worst = 0.0;
    for (int q = 0; q < na; ++q) {
        const int a = sw[q], k0 = sk[q], k1 = q + 1 < na ? sk[q + 1] : m;
        CHECK(a >= 0 && a < na && !seen[a], "%s: wave %d placed twice or out of range", name, a);
        if (a >= 0 && a < na) seen[a] = 1;
        CHECK(q == 0 ? k0 == 0 : k0 == sk[q - 1] + (k0 - sk[q - 1]), "%s: run %d", name, q);
        CHECK(k1 > k0, "%s: empty run for wave %d", name, a);            // every placed wave walks at least one column
        double load = 0.0;
        for (int k = k0; k < k1; ++k) load += cost[k];
        const double dev = std::fabs(load - tg[a]) / std::max(wg[a], 1e-3);
        CHECK(dev <= tol + 1e-9, "%s: wave %d off by %.3f weights, tolerance %.1f", name, a, dev, tol);
        worst = std::max(worst, dev);
    }
    CHECK(sk[0] == 0, "%s: the first run does not start the list", name);
    for (int q = 1; q < na; ++q) CHECK(sk[q] > sk[q - 1], "%s: runs out of order", name);   // contiguous, disjoint, covering: starts ascend, the last run ends at m
    const double lin = linear_worst(cost, tg, wg);
    std::printf("%-28s %2d waves %3d columns: tolerance %4.1f, worst deviation %6.2f weights (linear partition: %6.2f)\n", name, na, m, tol, worst, lin);
}

int main() {
    // 1. the shape the placement was built for: a 70x70 owner beside a helper that holds the 14 longest columns - columns of
    //    57 ... 1 rows, fifteen waves, weights and duties of the frozen table (model_coop_fit: rows = level * weight - duty)
    {
        std::vector<double> cost;
        for (int len = 57; len >= 1; --len) cost.push_back(len);
        const double w[15] = {1.413, 0.549, 1.95, 1.83, 1.48, 1.40, 1.23, 1.04, 0.88, 0.85, 0.62, 0.50, 0.36, 0.34, 0.16};
        const double hc[15] = {168.0, 52.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const double total = std::accumulate(cost.begin(), cost.end(), 0.0);
        double lo = 0.0, hi = 4000.0, level = 0.0;
        for (int it = 0; it < 80; ++it) {
            level = 0.5 * (lo + hi);
            double sum = 0.0;
            for (int a = 0; a < 15; ++a) sum += std::max(0.0, level * w[a] - hc[a]);
            (sum < total ? lo : hi) = level;
        }
        std::vector<double> tg, wg;
        for (int a = 0; a < 15; ++a)
            if (level * w[a] - hc[a] > 0.0) { tg.push_back(level * w[a] - hc[a]); wg.push_back(w[a]); }
        check_case("70x70 owner, frozen table", cost, tg, wg, true);
        std::vector<int> sw, sk;
        double tol = 0.0;
        nyx_place_runs(cost, tg, wg, sw, sk, &tol);
        CHECK(tol <= 6.0, "70x70: tolerance %.1f", tol);                       // (found at D = 3-4: every wave within ~6 rows of its target)
        CHECK(linear_worst(cost, tg, wg) >= 2.0 * tol, "70x70: the linear partition is not worse by 2x");
    }
    // 2. random shapes: 20 ... 120 columns of descending length, 2 ... 16 waves, random weights; targets proportional to the weights
    std::mt19937 rng(12345);
    int found_n = 0, cases = 0;
    for (int trial = 0; trial < 60; ++trial) {
        const int m = 20 + (int)(rng() % 101), na = 2 + (int)(rng() % 15);
        if (m < na) continue;
        std::vector<double> cost, tg, wg;
        for (int k = 0; k < m; ++k) cost.push_back(m - k + (trial % 3 == 0 ? 12.0 : 0.0));     // (a start-up charge per column in every third case)
        double ws = 0.0;
        for (int a = 0; a < na; ++a) { wg.push_back(0.15 + (rng() % 1000) / 500.0); ws += wg.back(); }
        const double total = std::accumulate(cost.begin(), cost.end(), 0.0);
        for (int a = 0; a < na; ++a) tg.push_back(total * wg[a] / ws);
        std::vector<int> sw, sk;
        double tol = 0.0;
        const bool found = nyx_place_runs(cost, tg, wg, sw, sk, &tol);
        ++cases;
        if (found) {
            ++found_n;
            char nm[64];
            std::snprintf(nm, sizeof nm, "random %d", trial);
            check_case(nm, cost, tg, wg, true);
        }
    }
    CHECK(found_n >= cases * 3 / 4, "only %d of %d random shapes placed", found_n, cases);
    // 3. inputs the search must refuse: one wave, more waves than columns, a mismatched weight vector
    {
        std::vector<int> sw, sk;
        CHECK(!nyx_place_runs({5, 4, 3}, {12}, {1.0}, sw, sk), "one wave accepted");
        CHECK(!nyx_place_runs({5, 4}, {3, 3, 3}, {1, 1, 1}, sw, sk), "more waves than columns accepted");
        CHECK(!nyx_place_runs({5, 4, 3}, {6, 6}, {1.0}, sw, sk), "weight vector of the wrong size accepted");
        // targets nothing can meet inside the widest tolerance: no placement, empty result
        CHECK(!nyx_place_runs({100, 100, 100}, {1, 299}, {0.01, 0.01}, sw, sk) && sw.empty() && sk.empty(), "impossible targets placed");
    }
    // 4. host time: a sixteen-wave shape with NO placement at any tolerance walks the whole budget of every D - the worst
    //    case fill_schedule can meet inside a launch - and must stay well under a second; asked again, it is a memo hit
    {
        std::vector<double> cost, tg, wg;
        for (int k = 0; k < 120; ++k) cost.push_back(120 - k);
        const double total = std::accumulate(cost.begin(), cost.end(), 0.0);
        for (int a = 0; a < 16; ++a) { wg.push_back(1.0); tg.push_back((total - 900.0) / 16.0 + 3.0 * a); }   // (every prefix fits, the last wave never does)
        std::vector<int> sw, sk;
        const auto t0 = std::chrono::steady_clock::now();
        const bool f1 = nyx_place_runs(cost, tg, wg, sw, sk);
        const double s1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const auto t1 = std::chrono::steady_clock::now();
        const bool f2 = nyx_place_runs(cost, tg, wg, sw, sk);
        const double s2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        std::printf("worst case (16 waves, 120 columns, found %d): %.3f s, again (memo): %.6f s\n", (int)f1, s1, s2);
        CHECK(f1 == f2, "memo changed the answer");
        CHECK(s1 < 1.5, "search took %.2f s", s1);
        CHECK(s2 < 0.01, "memo hit took %.4f s", s2);
    }
    std::printf("%s\n", failures ? "FAILED" : "ok");
    return failures ? 1 : 0;
}
