// hifitime_dev.h — device restatement of the hifitime conversions the path uses (shared by the propagation and the
// trajectory-evaluation kernels; see oracle/nyx_oracle.c for the reference call sites).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HT_DEVFN static __device__ __forceinline__

HT_DEVFN int64_t seconds_to_ns(double s) {
    double total = s * 1e9;
    if (total != total) return 0;
    if (total >= 9.2233720368547758e18) return INT64_MAX;
    if (total <= -9.2233720368547758e18) return INT64_MIN;
    return (int64_t)total;  // `as i64`: truncation toward zero
}

// floor-div / mod by 1e9 without the 64-bit integer divide (one f64 estimate + fix-up)
HT_DEVFN void divmod_1e9(int64_t v, int64_t &q, int64_t &r) {
    int64_t e = (int64_t)((double)v * 1e-9);
    int64_t rem = v - e * 1000000000LL;
    if (rem < 0) { e -= 1; rem += 1000000000LL; }
    if (rem < 0) { e -= 1; rem += 1000000000LL; }
    if (rem >= 1000000000LL) { e += 1; rem -= 1000000000LL; }
    if (rem >= 1000000000LL) { e += 1; rem -= 1000000000LL; }
    q = e;
    r = rem;
}

HT_DEVFN double ns_to_seconds(int64_t ns) {
    const int64_t NS_PER_CENTURY = 3155760000000000000LL;
    if (ns >= 0 && ns < NS_PER_CENTURY) {
        int64_t q, r;
        divmod_1e9(ns, q, r);
        return (double)q + (double)r * 1e-9;
    }
    int64_t cent;
    if (ns < 0) {
        cent = (ns >= -NS_PER_CENTURY) ? -1 : -2;
    } else {
        cent = (ns < 2 * NS_PER_CENTURY) ? 1 : 2;
    }
    int64_t rem = ns - cent * NS_PER_CENTURY;
    int64_t q, r;
    divmod_1e9(rem, q, r);
    return (double)cent * 3155760000.0 + (double)q + (double)r * 1e-9;
}

