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
#include <map>
#include <mutex>
#include <utility>
#include <vector>

// cost[k]: cost of the k-th column of the list (rows + start-up); target[a], weight[a]: load target and speed weight of wave a
// (targets > 0; they sum to about the total cost).  On success: seq_w[q] = the wave placed q-th along the list, seq_k[q] = index of its
// first column (runs are [seq_k[q], seq_k[q + 1])), *tol = the tolerance D it was found at.
// Host time is bounded: the visited states are a flat bitmap of (m + 1) x 2^na bits (no std::set, no std::function), the
// search stops at `budget_per_tol` expansions per tolerance as before - the same states in the same order, hence the same placement -
// and the result is memoised process-wide on its arguments (launch() rebuilds a schedule whenever the helper share moves; a
// configuration without a placement at small D used to pay the whole search again on every rebuild).
struct NyxPlaceSearch {
    int na, m;
    double D;
    const std::vector<double> &pre, &target, &weight;
    std::vector<unsigned long long> dead;   // bit (k << na | mask)
    std::vector<int> &seq_w, &seq_k;
    long budget;
    bool is_dead(int k, int mask) const { const size_t b = ((size_t)k << na) | (size_t)mask; return (dead[b >> 6] >> (b & 63)) & 1ull; }
    void set_dead(int k, int mask) { const size_t b = ((size_t)k << na) | (size_t)mask; dead[b >> 6] |= 1ull << (b & 63); }
    bool dfs(int k, int mask) {
        if (mask == (1 << na) - 1) return k == m;
        if (--budget < 0) return false;
        if (is_dead(k, mask)) return false;
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
        if (budget >= 0) set_dead(k, mask);   // (a state abandoned for lack of budget is not known to be dead)
        return false;
    }
};

static inline bool nyx_place_runs(const std::vector<double> &cost, const std::vector<double> &target, const std::vector<double> &weight,
                                  std::vector<int> &seq_w, std::vector<int> &seq_k, double *tol = nullptr, double d_max = 40.0,
                                  long budget_per_tol = 400000) {
    const int na = (int)target.size(), m = (int)cost.size();
    seq_w.clear(); seq_k.clear();
    if (na < 2 || na > 16 || m < na || (int)weight.size() != na) return false;
    // memo: (cost, target, weight, d_max, budget) -> (found, tol, seq_w, seq_k)
    struct Memo { bool found; double tol; std::vector<int> w, k; };
    static std::mutex mu;
    static std::map<std::vector<double>, Memo> memo;
    std::vector<double> key;
    key.reserve(cost.size() + 2 * target.size() + 4);
    key.push_back((double)m); key.push_back((double)na); key.push_back(d_max); key.push_back((double)budget_per_tol);
    key.insert(key.end(), cost.begin(), cost.end());
    key.insert(key.end(), target.begin(), target.end());
    key.insert(key.end(), weight.begin(), weight.end());
    {
        std::lock_guard<std::mutex> lk(mu);
        const auto it = memo.find(key);
        if (it != memo.end()) {
            seq_w = it->second.w; seq_k = it->second.k;
            if (tol && it->second.found) *tol = it->second.tol;
            return it->second.found;
        }
    }
    std::vector<double> pre(m + 1, 0.0);
    for (int k = 0; k < m; ++k) pre[k + 1] = pre[k] + cost[k];
    bool found = false;
    double found_tol = 0.0;
    NyxPlaceSearch s{na, m, 1.0, pre, target, weight, {}, seq_w, seq_k, 0};
    for (double D = 1.0; D <= d_max && !found; D += 1.0) {
        s.dead.assign((((size_t)(m + 1) << na) + 63) / 64, 0ull);
        seq_w.clear(); seq_k.clear();
        s.D = D;
        s.budget = budget_per_tol;
        if (s.dfs(0, 0)) { found = true; found_tol = D; }
    }
    if (!found) { seq_w.clear(); seq_k.clear(); }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (memo.size() > 256) memo.clear();  // (a process that sweeps thousands of weight tables: tools/_exp/hill.py)
        memo[key] = Memo{found, found_tol, seq_w, seq_k};
    }
    if (tol && found) *tol = found_tol;
    return found;
}
