// pk_force_models.h - part of propagate_kernel.hip (included there, in this order; not a stand-alone header): the force models beside the harmonics: point masses, SRP with occultation, drag, solid tides; dual-number types of the STM path.
// ---------------------------------------------------------------------------------------------
// Position-dependent non-harmonic terms (master, inside the harmonics window)
// ---------------------------------------------------------------------------------------------

// PointMasses::eom, reference dynamics/orbital.rs:214-247
DEVFN void point_masses_accel(CfgPtr cfg, const double *ed, int lane, const double *r, double *acc) {
    acc[0] = acc[1] = acc[2] = 0.0;
    const int npm = cfg->n_pm;
#pragma unroll
    for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
        if (k < npm) {
            const int s = cfg->pm_slot[k];
            double pij[3];
            ed_body(cfg, ed, lane, s, pij);
            const double r_ij3 = cube(norm3(pij[0], pij[1], pij[2]));
            const double rj0 = r[0] - pij[0], rj1 = r[1] - pij[1], rj2 = r[2] - pij[2];
            const double r_j3 = cube(norm3(rj0, rj1, rj2));
            const double nmu = -cfg->slot[s].mu;
            acc[0] += nmu * (rj0 / r_j3 + pij[0] / r_ij3);
            acc[1] += nmu * (rj1 / r_j3 + pij[1] / r_ij3);
            acc[2] += nmu * (rj2 / r_j3 + pij[2] / r_ij3);
        }
    }
}

DEVFN double circ_seg_area(double r, double d) { return r * r * acos(d / r) - d * sqrt(r * r - d * d); }

// anise Occultation.percentage restated (apparent-disk overlap); see oracle for the definition.
DEVFN double occultation_pct(double r_back, double r_front, const double *r_eb, const double *r_ls) {
    const double n_ls = norm3(r_ls[0], r_ls[1], r_ls[2]), n_eb = norm3(r_eb[0], r_eb[1], r_eb[2]);
    {
        // Full sunlight and full umbra decided on COSINES, for the whole wave at once.  The exact path below compares angles -
        // d_p - ls_p > fo_p  (no occultation: 0.0 exactly)  and  fo_p > d_p + ls_p  (total: 100.0 exactly) - which costs two asin and one
        // acos per shadow body per stage, almost always to return one of those two constants.  With all three angles in [0, pi] and
        // the apparent radii below pi / 2 the same inequalities read  cos d_p < cos(ls_p + fo_p)  and  cos d_p > cos(fo_p - ls_p);
        // they are taken here only with a margin of 1e-9 in the cosine (>= 1e-9 rad in the angles, seven orders above the rounding of
        // either formulation), and only when EVERY lane of the wave is decided - then the exact path would return the same constant,
        // bit for bit; in the penumbra band, or when any lane is near a boundary, the exact path runs as before.
        const double dotq = r_ls[0] * r_eb[0] + r_ls[1] * r_eb[1] + r_ls[2] * r_eb[2];
        const double sl = r_back / n_ls, sf = r_front / n_eb;  // sines of the apparent radii
        const double cd = -dotq / (n_eb * n_ls);               // the argument of the exact path's acos, same expression
        const double cl = sqrt(1.0 - sl * sl), cf = sqrt(1.0 - sf * sf);
        const bool angles = r_back < n_ls && r_front < n_eb && cd >= -1.0 && cd <= 1.0;
        const bool lit = angles && cd < (cl * cf - sl * sf) - 1e-9;
        const bool dark = angles && sf > sl && cd > (cf * cl + sf * sl) + 1e-9;
        if (__all(lit || dark)) return lit ? 0.0 : 100.0;
    }
    const double ls_p = (r_back >= n_ls) ? r_back : asin(r_back / n_ls);
    const double fo_p = (r_front >= n_eb) ? r_front : asin(r_front / n_eb);
    const double dot = r_ls[0] * r_eb[0] + r_ls[1] * r_eb[1] + r_ls[2] * r_eb[2];
    const double d_p = acos(-dot / (n_eb * n_ls));
    double pct;
    if (d_p - ls_p > fo_p) {
        pct = 0.0;
    } else if (fo_p > d_p + ls_p) {
        pct = 100.0;
    } else if (fabs(ls_p - fo_p) < d_p && d_p < ls_p + fo_p) {
        const double d1 = (d_p * d_p - ls_p * ls_p + fo_p * fo_p) / (2.0 * d_p);
        const double d2 = (d_p * d_p + ls_p * ls_p - fo_p * fo_p) / (2.0 * d_p);
        const double shadow = circ_seg_area(fo_p, d1) + circ_seg_area(ls_p, d2);
        if (shadow != shadow) {
            pct = 100.0;
        } else {
            const double nominal = 3.14159265358979323846 * (ls_p * ls_p);
            pct = 100.0 * shadow / nominal;
        }
    } else {
        pct = 100.0 * (fo_p * fo_p) / (ls_p * ls_p);
    }
    return pct;
}

// SolarPressure::eom (reference dynamics/solarpressure.rs:135-165) + ShadowModel::compute (cosmic/eclipse.rs:69-83)
DEVFN double srp_force(CfgPtr cfg, const double *ed, int lane, const double *r, double cr, double area, double *force) {
    const int ss = cfg->sun_slot;
    double ps[3];
    ed_body(cfg, ed, lane, ss, ps);
    const double rs0 = r[0] - ps[0], rs1 = r[1] - ps[1], rs2 = r[2] - ps[2];
    const double n = norm3(rs0, rs1, rs2);
    const double u0 = rs0 / n, u1 = rs1 / n, u2 = rs2 / n;
    const double sun_radius = cfg->slot[ss].radius;
    double best = 0.0;
    const int nsh = cfg->n_shadow;
#pragma unroll
    for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
        if (k < nsh) {
            const int sb = cfg->shadow_slot[k];
            double pb[3] = {0.0, 0.0, 0.0};
            double rad = cfg->central_radius;
            if (sb >= 0) {  // uniform
                ed_body(cfg, ed, lane, sb, pb);
                rad = cfg->slot[sb].radius;
            }
            const double r_eb[3] = {r[0] - pb[0], r[1] - pb[1], r[2] - pb[2]};
            const double r_ls[3] = {ps[0] - r[0], ps[1] - r[1], ps[2] - r[2]};
            const double pct = occultation_pct(sun_radius, rad, r_eb, r_ls);
            if (pct > best) best = pct;
        }
    }
    const double occult = best / 100.0;
    const double k = fabs(occult - 1.0);
    const double r_au = n / 149597870.700;
    const double inv = 1.0 / r_au;
    const double flux = (k * cfg->phi / cfg->c_m_s) * (inv * inv);
    const double scal = 1e-3 * cr * area * flux;
    force[0] = scal * u0;
    force[1] = scal * u1;
    force[2] = scal * u2;
    return k;  // illumination factor |occultation - 1|, frozen in the partials (solarpressure.rs:194-203)
}

// f64::powi as LLVM expands it (binary method, LSB first)
DEVFN double powi_dev(double x, int n) {
    double res = 1.0, sq = x;
    bool have = false;
    while (n) {
        if (n & 1) { res = have ? res * sq : sq; have = true; }
        sq = sq * sq;
        n >>= 1;
    }
    return res;
}

// Drag::eom (reference dynamics/drag.rs:181-284) with its unit / frame quirks, as restated in the oracle (drag_eom):
// velocity in the drag frame = R v - w x (R r) with w = W_dot z_body; Exponential mixes metres and km; the relative
// velocity is (inertial velocity) - (drag-frame velocity components).  `m` = DCM inertial -> drag frame of this stage.
// dW/dt of an orientation (rad/s) without its DCM: the polynomial rate, plus the series / Chebyshev terms when there are any
DEVFN double rotation_w_rate(CfgPtr cfg, const CAS DevRot &rot, const double *records, double et_s) {
    const double DEG = 3.14159265358979323846 / 180.0;
    if (rot.kind == NYX_HIP_ROT_IAU && rot.n_np == 0) return (rot.w[1] + 2.0 * rot.w[2] * (et_s / 86400.0)) * DEG / 86400.0;
    double m[9], wr = 0.0;
    (void)rotation_dcm(cfg, rot, records, et_s, m, &wr);
    return wr;
}

DEVFN void drag_force(CfgPtr cfg, const double *records, const double *ed, int lane, double et_s, const double *r, const double *v, double cd, double area,
                      double *force) {
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) m[q] = ed[q * DEV_LANES + lane];
    const double DEG = 3.14159265358979323846 / 180.0;
    const double d = et_s / 86400.0;
    const double wdot = rotation_w_rate(cfg, cfg->d_rot, records, et_s);
    double rb[3], vb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rb[i] = m[3 * i + 0] * r[0] + m[3 * i + 1] * r[1] + m[3 * i + 2] * r[2];
        vb[i] = m[3 * i + 0] * v[0] + m[3 * i + 1] * v[1] + m[3 * i + 2] * v[2];
    }
    vb[0] = vb[0] + wdot * rb[1];
    vb[1] = vb[1] - wdot * rb[0];
    const double rmag = norm3(rb[0], rb[1], rb[2]);
    double rho;
    if (cfg->drag_density == NYX_HIP_RHO_CONSTANT) {
        rho = cfg->drag_rho0;
        const double vn = norm3(vb[0], vb[1], vb[2]);
        const double s = -0.5 * 1e3 * rho * cd * area * vn;
        force[0] = s * vb[0]; force[1] = s * vb[1]; force[2] = s * vb[2];
        return;
    } else if (cfg->drag_density == NYX_HIP_RHO_EXPONENTIAL) {
        rho = cfg->drag_rho0 * exp(-(rmag - (cfg->drag_r0 + cfg->drag_re)) / cfg->drag_ref_alt_m);
    } else {
        const double alt = rmag - cfg->drag_re;
        if (alt > cfg->drag_max_alt_m / 1000.0) {
            rho = pow(10.0, (-7e-5) * alt - 14.464);
        } else {
            const double sc = (alt - 526.8000) / 292.8563;
            const double lg = 0.34047 * powi_dev(sc, 6) - 0.5889 * powi_dev(sc, 5) - 0.5269 * powi_dev(sc, 4) + 1.0036 * powi_dev(sc, 3) +
                              0.60713 * powi_dev(sc, 2) - 2.3024 * sc - 12.575;
            rho = pow(10.0, lg);
        }
    }
    const double vel[3] = {v[0] - vb[0], v[1] - vb[1], v[2] - vb[2]};
    const double vn = norm3(vel[0], vel[1], vel[2]);
    const double s = -0.5 * 1e3 * rho * cd * area * vn;
    force[0] = s * vel[0]; force[1] = s * vel[1]; force[2] = s * vel[2];
}

