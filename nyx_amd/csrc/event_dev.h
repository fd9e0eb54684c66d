// event_dev.h — device restatement of the event scalars / Event::eval the stop condition and the root search share
// (propagate_kernel.hip, traj_kernel.hip).  anise's `analysis` module is absent from the reference tree: classical
// definitions (Vallado RV2COE, ECEF -> geodetic), see the note in include/nyx_hip.h and the independent restatement in
// oracle/nyx_oracle.c.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nyx_hip.h"
#include "hifitime_dev.h"

#define EV_DEVFN static __device__ __forceinline__

EV_DEVFN bool ev_is_angle(int scalar) { return scalar == NYX_HIP_EV_TRUE_ANOMALY_DEG || scalar == NYX_HIP_EV_LONGITUDE_DEG; }

// The state expressed in the event's observer frame (until_nth_event's `event_frame`, propagators/event.rs:104-117): an
// IAU-oriented body-fixed frame of the same centre.  DCM = R3(W) R1(90 - delta) R3(90 + alpha) with the trigonometric
// series, as rotation_dcm in propagate_kernel.hip; velocity R v - w x (R r), w = dW/dt about the pole.
template <typename EV>
EV_DEVFN void ev_to_frame(const EV &ev, int64_t epoch_ns, const double y[6], double yf[6]) {
    const double DEG = 3.14159265358979323846 / 180.0;
    const double HALF_PI = 1.57079632679489661923;
    const double et = ns_to_seconds(epoch_ns);
    const double d = et / 86400.0, T = et / (86400.0 * 36525.0);
    double ra = ev.frame.ra_deg[0] + ev.frame.ra_deg[1] * T + ev.frame.ra_deg[2] * T * T;
    double dec = ev.frame.dec_deg[0] + ev.frame.dec_deg[1] * T + ev.frame.dec_deg[2] * T * T;
    double w = ev.frame.w_deg[0] + ev.frame.w_deg[1] * d + ev.frame.w_deg[2] * d * d;
    double wd = ev.frame.w_deg[1] + 2.0 * ev.frame.w_deg[2] * d;
    for (int k = 0; k < ev.frame.n_nut_prec; ++k) {
        const double th = (ev.frame.nut_prec_angle_deg[k][0] + ev.frame.nut_prec_angle_deg[k][1] * T) * DEG;
        double sn, cs;
        sincos(th, &sn, &cs);
        ra = ra + ev.frame.nut_prec_ra[k] * sn;
        dec = dec + ev.frame.nut_prec_dec[k] * cs;
        w = w + ev.frame.nut_prec_w[k] * sn;
        wd = wd + ev.frame.nut_prec_w[k] * cs * (ev.frame.nut_prec_angle_deg[k][1] * DEG / 36525.0);
    }
    double s1, c1, s2, c2, s3, c3;
    sincos(HALF_PI + ra * DEG, &s1, &c1);
    sincos(HALF_PI - dec * DEG, &s2, &c2);
    sincos(w * DEG, &s3, &c3);
    const double m[9] = {c3 * c1 - s3 * c2 * s1, c3 * s1 + s3 * c2 * c1, s3 * s2, -s3 * c1 - c3 * c2 * s1, -s3 * s1 + c3 * c2 * c1, c3 * s2,
                         s2 * s1, -s2 * c1, c2};
    const double wdot = wd * DEG / 86400.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        yf[i] = m[3 * i + 0] * y[0] + m[3 * i + 1] * y[1] + m[3 * i + 2] * y[2];
        yf[3 + i] = m[3 * i + 0] * y[3] + m[3 * i + 1] * y[4] + m[3 * i + 2] * y[5];
    }
    yf[3] = yf[3] + wdot * yf[1];
    yf[4] = yf[4] - wdot * yf[0];
}

// Geodetic latitude (deg) and height (km) on the ellipsoid (a, f): the classical iteration (Vallado, Algorithm 12), to 1e-12 rad
EV_DEVFN void ev_geodetic(double a, double f, const double y[6], double &lat_deg, double &height_km) {
    const double e2 = f * (2.0 - f);
    const double r_delta = sqrt(y[0] * y[0] + y[1] * y[1]);
    double lat = atan2(y[2], r_delta);
    double c = a;
    for (int it = 0; it < 20; ++it) {
        const double sl = sin(lat);
        c = a / sqrt(1.0 - e2 * sl * sl);
        const double nl = atan2(y[2] + c * e2 * sl, r_delta);
        const bool done = fabs(nl - lat) < 1e-12;
        lat = nl;
        if (done) break;
    }
    lat_deg = lat * (180.0 / 3.14159265358979323846);
    const double sl = sin(lat), cl = cos(lat);
    c = a / sqrt(1.0 - e2 * sl * sl);
    height_km = (fabs(cl) > 1e-6) ? r_delta / cl - c : fabs(y[2]) / fabs(sl) - c * (1.0 - e2);
}

EV_DEVFN double ev_scalar(int scalar, double mu, double eq_radius, double flattening, const double y[6]) {
    const double rmag = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    const double vmag = sqrt(y[3] * y[3] + y[4] * y[4] + y[5] * y[5]);
    if (scalar == NYX_HIP_EV_RMAG_KM) return rmag;
    if (scalar == NYX_HIP_EV_VMAG_KM_S) return vmag;
    if (scalar >= NYX_HIP_EV_X_KM && scalar <= NYX_HIP_EV_VZ_KM_S) {
        const int k = scalar - NYX_HIP_EV_X_KM;  // static-indexed selects keep y in registers
        return k == 0 ? y[0] : k == 1 ? y[1] : k == 2 ? y[2] : k == 3 ? y[3] : k == 4 ? y[4] : y[5];
    }
    if (scalar == NYX_HIP_EV_LONGITUDE_DEG) {
        const double deg = atan2(y[1], y[0]) * (180.0 / 3.14159265358979323846);
        return deg < 0.0 ? deg + 360.0 : deg;
    }
    if (scalar == NYX_HIP_EV_DECLINATION_DEG) return asin(y[2] / rmag) * (180.0 / 3.14159265358979323846);
    if (scalar == NYX_HIP_EV_LATITUDE_DEG || scalar == NYX_HIP_EV_HEIGHT_KM) {
        double lat, h;
        ev_geodetic(eq_radius, flattening, y, lat, h);
        return scalar == NYX_HIP_EV_LATITUDE_DEG ? lat : h;
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
template <typename EV>
EV_DEVFN double ev_eval(const EV &ev, double mu, int64_t epoch_ns, const double y[6]) {
    double yf[6];
    if (ev.has_frame) {
        ev_to_frame(ev, epoch_ns, y, yf);
    } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) yf[q] = y[q];
    }
    const double d = ev_scalar(ev.scalar, mu, ev.frame_eq_radius_km, ev.frame_flattening, yf) - ev.desired;
    if (!ev_is_angle(ev.scalar)) return d;
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
