// event_dev.h — device restatement of the event scalars / Event::eval the stop condition and the root search share
// (propagate_kernel.hip, traj_kernel.hip).  anise's `analysis` module is absent from the reference tree: classical
// definitions (Vallado RV2COE), see the note in include/nyx_hip.h and the independent restatement in oracle/nyx_oracle.c.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nyx_hip.h"

#define EV_DEVFN static __device__ __forceinline__

EV_DEVFN bool ev_is_angle(int scalar) { return scalar == NYX_HIP_EV_TRUE_ANOMALY_DEG; }

EV_DEVFN double ev_scalar(int scalar, double mu, const double y[6]) {
    const double rmag = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    const double vmag = sqrt(y[3] * y[3] + y[4] * y[4] + y[5] * y[5]);
    if (scalar == NYX_HIP_EV_RMAG_KM) return rmag;
    if (scalar == NYX_HIP_EV_VMAG_KM_S) return vmag;
    if (scalar >= NYX_HIP_EV_X_KM && scalar <= NYX_HIP_EV_VZ_KM_S) {
        const int k = scalar - NYX_HIP_EV_X_KM;  // static-indexed selects keep y in registers
        return k == 0 ? y[0] : k == 1 ? y[1] : k == 2 ? y[2] : k == 3 ? y[3] : k == 4 ? y[4] : y[5];
    }
    if (scalar == NYX_HIP_EV_SMA_KM) {
        const double energy = vmag * vmag / 2.0 - mu / rmag;
        return -mu / (2.0 * energy);
    }
    const double rv = y[0] * y[3] + y[1] * y[4] + y[2] * y[5];
    const double k = vmag * vmag - mu / rmag;
    const double e0 = (k * y[0] - rv * y[3]) / mu, e1 = (k * y[1] - rv * y[4]) / mu, e2 = (k * y[2] - rv * y[5]) / mu;
    const double ecc = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    if (scalar == NYX_HIP_EV_ECC) return ecc;
    // atan2 of (sin, cos) projected on the orbit plane: acos(e.r / (|e||r|)) loses half the digits at the apsides
    const double h0 = y[1] * y[5] - y[2] * y[4], h1 = y[2] * y[3] - y[0] * y[5], h2 = y[0] * y[4] - y[1] * y[3];
    const double x0 = e1 * y[2] - e2 * y[1], x1 = e2 * y[0] - e0 * y[2], x2 = e0 * y[1] - e1 * y[0];
    const double sin_part = (x0 * h0 + x1 * h1 + x2 * h2) / sqrt(h0 * h0 + h1 * h1 + h2 * h2);
    const double cos_part = e0 * y[0] + e1 * y[1] + e2 * y[2];
    const double deg = atan2(sin_part, cos_part) * (180.0 / 3.14159265358979323846);
    return deg < 0.0 ? deg + 360.0 : deg;
}

// Event::eval for Condition::Equals: value - desired, wrapped to [-180, 180) for angles
EV_DEVFN double ev_eval(int scalar, double desired, double mu, const double y[6]) {
    const double d = ev_scalar(scalar, mu, y) - desired;
    if (!ev_is_angle(scalar)) return d;
    double w = fmod(d + 180.0, 360.0);
    if (w < 0.0) w += 360.0;
    return w - 180.0;
}

// the crossing rule of the `enough_crossings` closure (propagators/event.rs:124-141)
EV_DEVFN bool ev_crossing(int scalar, double y_prev, double y_next) {
    if (ev_is_angle(scalar)) {
        const bool sp = __builtin_signbit(y_prev), sn = __builtin_signbit(y_next);  // f64::signum: -0.0 is negative
        const bool nan = (y_prev != y_prev) || (y_next != y_next);                   // NaN.signum() != anything
        return (nan || sp != sn) && fabs(y_next - y_prev) < 180.0;
    }
    return y_prev * y_next < 0.0;
}
