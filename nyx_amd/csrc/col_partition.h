// col_partition.h - placement of contiguous column runs along a column list in a FREE wave order (host only; fill_schedule in abi.cpp,
// tests/cxx/col_partition_check.cpp).
//
// A wave of a sixteen-wave owner workgroup walks ONE contiguous run of the (length-sorted) column list.  Cutting the list at the
// cumulative targets in a fixed wave order rounds every wave's load to whole columns - up to half a column (+-25 rows of ~230 for the
// waves that hold the 50-row columns of a 70x70 owner) - and a column wave is bound by its OWN issue rate, so the two or three waves
// rounded UP set the workgroup's period.  A run of j columns starting at length L holds jL - j(j-1)/2 rows: which sums exist depends on
// WHERE in the list a run sits.  nyx_place_runs() therefore searches the order: depth first over (columns consumed, waves placed) for
// the smallest tolerance D such that  |load_w - target_w| <= D * weight_w  for every wave (weight = the wave's speed: the same TIME
// error everywhere), D = 1, 2, ... d_max.  A pure function of its arguments.
#pragma once
#include <functional>
#include <set>
#include <utility>
#include <vector>

// cost[k]: cost of the k-th column of the list (rows + start-up); target[a], weight[a]: load target and speed weight of wave a
// (targets > 0; they sum to about the total cost).  On success: seq_w[q] = the wave placed q-th along the list, seq_k[q] = index of its
// first column (runs are [seq_k[q], seq_k[q + 1])), *tol = the tolerance D it was found at.
static inline bool nyx_place_runs(const std::vector<double> &cost, const std::vector<double> &target, const std::vector<double> &weight,
                                  std::vector<int> &seq_w, std::vector<int> &seq_k, double *tol = nullptr, double d_max = 40.0,
                                  long budget_per_tol = 400000) {
    const int na = (int)target.size(), m = (int)cost.size();
    seq_w.clear(); seq_k.clear();
    if (na < 2 || na > 16 || m < na || (int)weight.size() != na) return false;
    std::vector<double> pre(m + 1, 0.0);
    for (int k = 0; k < m; ++k) pre[k + 1] = pre[k] + cost[k];
    for (double D = 1.0; D <= d_max; D += 1.0) {
        std::set<std::pair<int, int>> dead;
        seq_w.clear(); seq_k.clear();
        long budget = budget_per_tol;
        std::function<bool(int, int)> dfs = [&](int k, int mask) -> bool {
            if (mask == (1 << na) - 1) return k == m;
            if (--budget < 0) return false;
            if (dead.count({k, mask})) return false;
            const int left = na - __builtin_popcount((unsigned)mask);
            for (int a = 0; a < na; ++a) {
                if (mask & (1 << a)) continue;
                const double tl = D * (weight[a] > 1e-3 ? weight[a] : 1e-3);
                // run lengths whose load meets the target within the tolerance (the last wave takes what is left)
                for (int e = k + 1; e <= m - (left - 1); ++e) {
                    const double load = pre[e] - pre[k];
                    if (load > target[a] + tl) break;
                    if (load < target[a] - tl) continue;
                    if (left == 1 && e != m) continue;
                    seq_w.push_back(a); seq_k.push_back(k);
                    if (dfs(e, mask | (1 << a))) return true;
                    seq_w.pop_back(); seq_k.pop_back();
                }
            }
            if (budget >= 0) dead.insert({k, mask});   // (a state abandoned for lack of budget is not known to be dead)
            return false;
        };
        if (dfs(0, 0)) {
            if (tol) *tol = D;
            return true;
        }
    }
    seq_w.clear(); seq_k.clear();
    return false;
}
