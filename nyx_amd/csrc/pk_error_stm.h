// pk_error_stm.h - part of propagate_kernel.hip (included there, in this order; not a stand-alone header): ErrorControl::estimate, the fold of the partial sums, the STM variants' gradients and per-step updates, phase C of the quad layout.
// ---------------------------------------------------------------------------------------------
// ErrorControl::estimate on the 9-vector (reference propagators/error_ctrl.rs:79-229).
// Elements 9..89 of the reference's 90-vector are zero without an STM and do not contribute.
// ---------------------------------------------------------------------------------------------

DEVFN double rss_step3(const double *e, const double *cand, const double *cur) {
    const double mag = norm3(cand[0] - cur[0], cand[1] - cur[1], cand[2] - cur[2]);
    const double err = norm3(e[0], e[1], e[2]);
    return (mag > sqrt(0.1)) ? err / mag : err;
}
DEVFN double rss_state3(const double *e, const double *cand, const double *cur) {
    const double mag = 0.5 * norm3(cand[0] + cur[0], cand[1] + cur[1], cand[2] + cur[2]);
    const double err = norm3(e[0], e[1], e[2]);
    return (mag > 0.1) ? err / mag : err;
}

// nalgebra's 8-accumulator dot over the 9 leading entries of the 90-vector: entries 0..7 land
// in acc0..acc7, entry 8 in acc0 of the second block; the remaining blocks add zeros.
DEVFN double nalgebra_norm9(const double *x) {
    const double a0 = x[0] * x[0] + x[8] * x[8];
    double res = 0.0;
    res += a0 + x[4] * x[4];
    res += x[1] * x[1] + x[5] * x[5];
    res += x[2] * x[2] + x[6] * x[6];
    res += x[3] * x[3] + x[7] * x[7];
    return sqrt(res);
}

DEVFN double error_estimate(int ec, const double *e, const double *cand, const double *cur) {
    double tmp[9];
    switch (ec) {
    case NYX_HIP_RSS_CARTESIAN_STATE: return fmax(rss_state3(e, cand, cur), rss_state3(e + 3, cand + 3, cur + 3));
    case NYX_HIP_RSS_CARTESIAN_STEP: return fmax(rss_step3(e, cand, cur), rss_step3(e + 3, cand + 3, cur + 3));
    case NYX_HIP_RSS_STATE: {
        for (int i = 0; i < 9; ++i) tmp[i] = cand[i] + cur[i];
        const double mag = 0.5 * nalgebra_norm9(tmp), err = nalgebra_norm9(e);
        return (mag > 0.1) ? err / mag : err;
    }
    case NYX_HIP_RSS_STEP: {
        for (int i = 0; i < 9; ++i) tmp[i] = cand[i] - cur[i];
        const double mag = nalgebra_norm9(tmp), err = nalgebra_norm9(e);
        return (mag > sqrt(0.1)) ? err / mag : err;
    }
    case NYX_HIP_LARGEST_ERROR: {
        double mx = 0.0;
        for (int i = 0; i < 9; ++i) {
            const double dl = cand[i] - cur[i];
            const double er = (dl > 0.1) ? fabs(e[i] / dl) : fabs(e[i]);
            if (er > mx) mx = er;
        }
        return mx;
    }
    case NYX_HIP_LARGEST_STATE: {
        double mag = 0.0, err = 0.0;
        for (int i = 0; i < 9; ++i) { mag += 0.5 * fabs(cand[i] + cur[i]); err += fabs(e[i]); }
        return (mag > 0.1) ? err / mag : err;
    }
    default: {
        double mag = 0.0, err = 0.0;
        for (int i = 0; i < 9; ++i) { mag += fabs(cand[i] - cur[i]); err += fabs(e[i]); }
        return (mag > 0.1) ? err / mag : err;
    }
    }
}

// Fold of the 15 workers' partial accelerations (fixed wave order => deterministic).  Kept out of line on purpose:
// inside the integrator role (at its 128-VGPR cap) the scheduler serialised the 60 LDS reads at one LDS latency each
// (5 k cycles on the critical path of every force evaluation); on its own the function batches them.
#ifndef FOLD_INLINE
#define FOLD_INLINE 0
#endif
#if FOLD_INLINE
// (inlined variant: the sixty reads in four batches of fifteen - one component at a time - so that they need 30 registers, not 120)
static __device__ __forceinline__ Partial4 fold_partials(LdsCPtr part, int lane, double px, double py, double pz, double pw) {
    double o[4] = {px, py, pz, pw};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v[DEV_MAX_WAVES - 1];
#pragma unroll
        for (int w = 1; w < DEV_MAX_WAVES; ++w) v[w - 1] = part[(w * 4 + q) * DEV_LANES + lane];
#pragma unroll
        for (int w = 1; w < DEV_MAX_WAVES; ++w) o[q] += v[w - 1];
    }
    Partial4 r = {o[0], o[1], o[2], o[3]};
    return r;
}
#else
static __device__ __attribute__((noinline)) Partial4 fold_partials(LdsCPtr part, int lane, double px, double py, double pz, double pw) {
    double v[4][DEV_MAX_WAVES - 1];
#pragma unroll
    for (int w = 1; w < DEV_MAX_WAVES; ++w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q][w - 1] = part[(w * 4 + q) * DEV_LANES + lane];
    }
#pragma unroll
    for (int w = 1; w < DEV_MAX_WAVES; ++w) {
        px += v[0][w - 1]; py += v[1][w - 1]; pz += v[2][w - 1]; pw += v[3][w - 1];
    }
    Partial4 r = {px, py, pz, pw};
    return r;
}
#endif

// ---------------------------------------------------------------------------------------------
// STM variant: position partials of the perturbations (perturbation wave) and the per-step update
// ---------------------------------------------------------------------------------------------

// PointMasses::gradient (orbital.rs:249-308) and SolarPressure::gradient (solarpressure.rs:167-232, k frozen).
// out[27][64]: a_pm(3), G_pm(9 row-major), f_srp/m(3), G_srp/m(9), c = (F/Cr)/m (3, zero unless `estimate`).
DEVFN void pert_gradients(CfgPtr cfg, const double *ed, int lane, const double *r, double cr, double area, double mass,
                          bool has_pm, bool has_srp, bool has_tides, double *out) {
    double o[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) o[q] = 0.0;
    if (has_pm) {
        const int npm = cfg->n_pm;
#pragma unroll
        for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
            if (k < npm) {
                const int s = cfg->pm_slot[k];
                double pb3[3];
                ed_body(cfg, ed, lane, s, pb3);
                const D3 rij[3] = {d3c(pb3[0]), d3c(pb3[1]), d3c(pb3[2])};
                const D3 rij3 = d3cube(d3norm(rij[0], rij[1], rij[2]));
                const D3 rj[3] = {{r[0] - rij[0].v, 1.0, 0.0, 0.0}, {r[1] - rij[1].v, 0.0, 1.0, 0.0}, {r[2] - rij[2].v, 0.0, 0.0, 1.0}};
                const D3 rj3 = d3cube(d3norm(rj[0], rj[1], rj[2]));
                const D3 gm = d3c(-cfg->slot[s].mu);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const D3 t = (d3div(rj[i], rj3) + d3div(rij[i], rij3)) * gm;
                    o[i] += t.v;
                    o[3 + 3 * i + 0] += t.x; o[3 + 3 * i + 1] += t.y; o[3 + 3 * i + 2] += t.z;
                }
            }
        }
    }
    if (has_tides) {  // SolidTides::gradient: added to the orbital (point-mass) block
        const D3 rd[3] = {{r[0], 1.0, 0.0, 0.0}, {r[1], 0.0, 1.0, 0.0}, {r[2], 0.0, 0.0, 1.0}};
        D3 at[3];
        tides_accel<D3>(cfg, ed, lane, rd, at);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            o[i] += at[i].v;
            o[3 + 3 * i + 0] += at[i].x; o[3 + 3 * i + 1] += at[i].y; o[3 + 3 * i + 2] += at[i].z;
        }
    }
    if (has_srp) {
        const int ss = cfg->sun_slot;
        double ps[3];
    ed_body(cfg, ed, lane, ss, ps);
        const D3 rs[3] = {{r[0] - ps[0], 1.0, 0.0, 0.0}, {r[1] - ps[1], 0.0, 1.0, 0.0}, {r[2] - ps[2], 0.0, 0.0, 1.0}};
        const D3 n = d3norm(rs[0], rs[1], rs[2]);
        // illumination factor exactly as the real path computes it (frozen in the partials)
        double f3[3];
        const double kfro = srp_force(cfg, ed, lane, r, cr, area, f3);  // real path: force and the frozen illumination factor
        const D3 r_au = n * (1.0 / 149597870.700);
        const D3 inv = d3div(d3c(1.0), r_au);
        const D3 flux = (inv * inv) * (kfro * cfg->phi / cfg->c_m_s);
        const double scal = 1e-3 * cr * area;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const D3 f = (flux * scal) * d3div(rs[i], n);
            o[12 + i] = f3[i] / mass;  // real part from the real path (spacecraft.rs:349)
            o[15 + 3 * i + 0] = f.x / mass; o[15 + 3 * i + 1] = f.y / mass; o[15 + 3 * i + 2] = f.z / mass;
            if (cfg->srp_estimate) o[24 + i] = (f3[i] / cr) / mass;  // solarpressure.rs:225-229, spacecraft.rs:355-359
        }
    }
#pragma unroll
    for (int q = 0; q < 27; ++q) out[q * DEV_LANES + lane] = o[q];
}

// Phi_next = Phi + h * Phi * A_sum with A_sum = [[0, (sum b) I, 0], [Gs, 0, cs], [0, 0, 0]]  — the reference integrates
// Phi_dot = Phi_ctx * A with the STEP-START Phi (dynamics/spacecraft.rs:214), so the RK sum factorises exactly.
// phi: this trajectory's 81 entries, column-major (cosmic/spacecraft.rs:467-471).
DEVFN bool stm_update(double *phi, double h, const double *sacc, int lane, double sumb) {
    double gs[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) gs[q] = sacc[q * DEV_LANES + lane];
    bool nan = false;
    for (int r = 0; r < 9; ++r) {
        double row[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) row[c] = phi[r + 9 * c];
        double nw[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            nw[j] = row[j] + h * (row[3] * gs[0 * 3 + j] + row[4] * gs[1 * 3 + j] + row[5] * gs[2 * 3 + j]);
            nw[3 + j] = row[3 + j] + h * (sumb * row[j]);
        }
        nw[6] = row[6] + h * (row[3] * gs[9] + row[4] * gs[10] + row[5] * gs[11]);
        nw[7] = row[7];
        nw[8] = row[8];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            nan = nan || (nw[c] != nw[c]);
            phi[r + 9 * c] = nw[c];
        }
    }
    return nan;
}

// Quad layout of the two functions above.  out[15][64]: a_pm(3), column (ql - 1) of G_pm (3), f_srp/m(3), column of
// G_srp/m (3), c(3); every expression is pert_gradients' own for the value and for ONE partial slot.
DEVFN void pert_gradients_q(CfgPtr cfg, const double *ed, int lane, int ql, const double *r, double cr, double area, double mass,
                            bool has_pm, bool has_srp, bool has_tides, int pmask, double *out) {
    double o[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) o[q] = 0.0;
    if (has_pm) {
        const int npm = cfg->n_pm;
#pragma unroll
        for (int k = 0; k < DEV_MAX_SLOTS; ++k) {
            if (k < npm) {
                const int s = cfg->pm_slot[k];
                double pb3[3];
                ed_body(cfg, ed, lane, s, pb3);
                const D1 rij[3] = {d1c(pb3[0]), d1c(pb3[1]), d1c(pb3[2])};
                const D1 rij3 = d1cube(d1norm(rij[0], rij[1], rij[2]));
                const D1 rj[3] = {d1seed(r[0] - rij[0].v, 0, ql), d1seed(r[1] - rij[1].v, 1, ql), d1seed(r[2] - rij[2].v, 2, ql)};
                const D1 rj3 = d1cube(d1norm(rj[0], rj[1], rj[2]));
                const D1 gm = d1c(-cfg->slot[s].mu);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const D1 t = (d1div(rj[i], rj3) + d1div(rij[i], rij3)) * gm;
                    o[i] += t.v;
                    o[3 + i] += t.d;
                }
            }
        }
    }
    if (has_tides) {
        const D1 rd[3] = {d1seed(r[0], 0, ql), d1seed(r[1], 1, ql), d1seed(r[2], 2, ql)};
        D1 at[3];
        tides_accel<D1>(cfg, ed, lane, rd, at);
#pragma unroll
        for (int i = 0; i < 3; ++i) { o[i] += at[i].v; o[3 + i] += at[i].d; }
    }
    if (has_srp) {
        const int ss = cfg->sun_slot;
        double ps[3];
    ed_body(cfg, ed, lane, ss, ps);
        const D1 rs[3] = {d1seed(r[0] - ps[0], 0, ql), d1seed(r[1] - ps[1], 1, ql), d1seed(r[2] - ps[2], 2, ql)};
        const D1 n = d1norm(rs[0], rs[1], rs[2]);
        double f3[3];
        const double kfro = srp_force(cfg, ed, lane, r, cr, area, f3);
        const D1 r_au = n * (1.0 / 149597870.700);
        const D1 inv = d1div(d1c(1.0), r_au);
        const D1 flux = (inv * inv) * (kfro * cfg->phi / cfg->c_m_s);
        const double scal = 1e-3 * cr * area;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const D1 f = (flux * scal) * d1div(rs[i], n);
            o[6 + i] = f3[i] / mass;
            o[9 + i] = f.d / mass;
            if (cfg->srp_estimate) o[12 + i] = (f3[i] / cr) / mass;
        }
    }
    // (role fan-out: rows 0..5 belong to the point-mass share, 6..14 to the SRP share)
#pragma unroll
    for (int q = 0; q < 15; ++q)
        if (pmask & (q < 6 ? DEV_PERT_PM : DEV_PERT_SRP)) out[q * DEV_LANES + lane] = o[q];
}

// stm_update for the quad layout: sacc rows 0..2 hold, per lane, column (ql - 1) of sum b_i G_i and rows 3..5 sum b_i c_i
// (the same in the four lanes); the nine rows of Phi are dealt over the quad's lanes.
DEVFN bool stm_update_q(double *phi, double h, const double *sacc, int lane, int ql, double sumb) {
    double gs[12];
    const int base = lane & ~3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) gs[3 * i + j] = sacc[i * DEV_LANES + base + 1 + j];
        gs[9 + i] = sacc[(3 + i) * DEV_LANES + lane];
    }
    bool nan = false;
    for (int r = ql; r < 9; r += 4) {
        double row[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) row[c] = phi[r + 9 * c];
        double nw[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            nw[j] = row[j] + h * (row[3] * gs[0 * 3 + j] + row[4] * gs[1 * 3 + j] + row[5] * gs[2 * 3 + j]);
            nw[3 + j] = row[3 + j] + h * (sumb * row[j]);
        }
        nw[6] = row[6] + h * (row[3] * gs[9] + row[4] * gs[10] + row[5] * gs[11]);
        nw[7] = row[7];
        nw[8] = row[8];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            nan = nan || (nw[c] != nw[c]);
            phi[r + 9 * c] = nw[c];
        }
    }
    return quad_or(nan ? 1 : 0) != 0;
}

// Phase C of the quad layout (assembly of f(x) and of this lane's column of A = df/dx from the partial sums, the
// perturbation rows and the position-only pieces formed in the window; accumulation of sum b_i A_i; k_i), OUT OF LINE: inside
// the integrator role (128 VGPRs = 64 doubles for everything it keeps live) it ran through scratch, 10 k cycles per
// evaluation; on its own it has the whole register file.  Everything goes through LDS: `qpre` rows 0..2 two-body
// acceleration, 3..5 this lane's column of its gradient, 6..13 the duals of s, t, u and (mu / r) / R_eq.
#define QPRE_ROWS 23  /* + rows 14..22: the DCM of the stage (the almanac wave recycles its LDS buffer in the pipelined loop) */
#define PC_HAS_PM 1
#define PC_HAS_GRAV 2
#define PC_HAS_SRP 4
// (LDS pointers are passed as such: through generic pointers every access pays an address-space test)
static __device__ __attribute__((noinline)) void phase_c_quad(LdsCPtr pertD, LdsCPtr partD, LdsCPtr qpre,
                                                            LdsCPtr ysl, LdsPtr sacc, LdsPtr kb, int kb_str, double b_i, int nw_v,
                                                            int flags_v, int lane, int ql, LdsFlagPtr gate, int gate_val_v,
                                                            int64_t *pslot = nullptr) {
    const int64_t pc0 = pslot ? (int64_t)__builtin_readcyclecounter() : 0;
    const int nw = __builtin_amdgcn_readfirstlane(nw_v);
    const int flags = __builtin_amdgcn_readfirstlane(flags_v);
    double acc[3], Gc[3], cv[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 3; ++q) { acc[q] = qpre[q * DEV_LANES + lane]; Gc[q] = qpre[(3 + q) * DEV_LANES + lane]; }
    if (flags & PC_HAS_PM) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { acc[q] += pertD[q * DEV_LANES + lane]; Gc[q] += pertD[(3 + q) * DEV_LANES + lane]; }
    }
    if (flags & PC_HAS_GRAV) {
        D1 pD[4] = {d1c(0.0), d1c(0.0), d1c(0.0), d1c(0.0)};
        for (int w0 = 0; w0 < nw; w0 += 4) {  // fixed wave order; four waves' worth of loads in flight
            double v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                LdsCPtr pp = partD + (w0 + k < nw ? w0 + k : 0) * QSLOT;
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[k][2 * q] = pp[4 * DEV_LANES + q * (DEV_LANES / 4) + (lane >> 2)]; v[k][2 * q + 1] = pp[q * DEV_LANES + lane]; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (w0 + k < nw) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { pD[q].v += v[k][2 * q]; pD[q].d += v[k][2 * q + 1]; }
                }
            }
        }
        {   // the sums are in registers: the column waves may write the next stage's into their slots
            const int gate_val = __builtin_amdgcn_readfirstlane(gate_val_v);
            if (gate_val > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) *gate = gate_val;
            }
        }
        if (pslot && lane == 0) pslot[5] += (int64_t)__builtin_readcyclecounter() - pc0;
        double m[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) m[q] = qpre[(14 + q) * DEV_LANES + lane];
        D1 aux[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { aux[q].v = qpre[(6 + 2 * q) * DEV_LANES + lane]; aux[q].d = qpre[(7 + 2 * q) * DEV_LANES + lane]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) pD[q] = pD[q] * aux[3];
        const D1 al[3] = {pD[0] + pD[3] * aux[0], pD[1] + pD[3] * aux[1], pD[2] + pD[3] * aux[2]};
        // a = R^T a_bf ; G_h = R^T G_bf R: the first product is linear in the partial slot (this lane's), the second
        // mixes the three slots: fetched from the quad's lanes 1..3; this lane forms column b = ql - 1
        double tmpc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc[a] += m[0 + a] * al[0].v + m[3 + a] * al[1].v + m[6 + a] * al[2].v;
            tmpc[a] = m[0 + a] * al[0].d + m[3 + a] * al[1].d + m[6 + a] * al[2].d;
        }
        const int b = ql > 0 ? ql - 1 : 0;
        const double mb0 = qpre[(14 + b) * DEV_LANES + lane], mb1 = qpre[(17 + b) * DEV_LANES + lane], mb2 = qpre[(20 + b) * DEV_LANES + lane];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double t0 = quad_bcast<1>(tmpc[a]), t1 = quad_bcast<2>(tmpc[a]), t2 = quad_bcast<3>(tmpc[a]);
            Gc[a] += t0 * mb0 + t1 * mb1 + t2 * mb2;
        }
    } else {
        const int gate_val = __builtin_amdgcn_readfirstlane(gate_val_v);
        if (gate_val > 0 && lane == 0) *gate = gate_val;
    }
    if (flags & PC_HAS_SRP) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            acc[q] += pertD[(6 + q) * DEV_LANES + lane]; cv[q] = pertD[(12 + q) * DEV_LANES + lane];
            Gc[q] += pertD[(9 + q) * DEV_LANES + lane];
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        sacc[q * DEV_LANES + lane] += b_i * Gc[q];
        sacc[(3 + q) * DEV_LANES + lane] += b_i * cv[q];
    }
    // k_i = [velocity of the stage state, f(x)]
#pragma unroll
    for (int e = 0; e < 3; ++e) { kb[e * kb_str] = ysl[(3 + e) * DEV_LANES + lane]; kb[(3 + e) * kb_str] = acc[e]; }
    if (pslot && lane == 0) pslot[6] += (int64_t)__builtin_readcyclecounter() - pc0;
}

// Quad layout: the position-only pieces of phase C - the two-body dual, the duals of s, t, u and (mu / r) / R_eq, the stage's DCM -
// formed inside the window and left in L.qpre for phase C.  By the integrator wave, or (DevCfg.qpre_off, round 5) by the almanac wave
// that holds DEV_ROLE_QPRE: the integrator's chain - phase C, phase A, window - is what bounds a quad workgroup's period, and this is
// 4-5 k cycles of its window that need nothing but the published position and the stage's epoch data.  Same operations on the same
// operands in the same lanes: same bits.
DEVFN void quad_pre(CfgPtr cfg, const double *edc, double y0, double y1, double y2, int ql, int lane, double *qpre, bool has_grav) {
    double q_acc[3], q_gc[3];
    D1 q_aux[4] = {d1c(0.0), d1c(0.0), d1c(0.0), d1c(0.0)};
    const D1 rad[3] = {d1seed(y0, 0, ql), d1seed(y1, 1, ql), d1seed(y2, 2, ql)};
    const D1 fac = d1div(d1c(-cfg->mu_central), d1cube(d1norm(rad[0], rad[1], rad[2])));
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const D1 a = rad[q] * fac;
        q_acc[q] = a.v; q_gc[q] = a.d;
    }
    if (has_grav) {
        double m[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) m[q] = edc[q * DEV_LANES + lane];
        double rq[3] = {y0, y1, y2};
        if (cfg->g_slot >= 0) {  // (uniform; plain stage loop then: edc is this stage's data)
            double pg[3];
            ed_body(cfg, edc, lane, cfg->g_slot, pg);
            rq[0] = y0 - pg[0]; rq[1] = y1 - pg[1]; rq[2] = y2 - pg[2];
        }
        const D1 x0 = d1seed(m[0] * rq[0] + m[1] * rq[1] + m[2] * rq[2], 0, ql);
        const D1 x1 = d1seed(m[3] * rq[0] + m[4] * rq[1] + m[5] * rq[2], 1, ql);
        const D1 x2 = d1seed(m[6] * rq[0] + m[7] * rq[1] + m[8] * rq[2], 2, ql);
        const D1 rD = d1norm(x0, x1, x2);
        q_aux[0] = d1div(x0, rD); q_aux[1] = d1div(x1, rD); q_aux[2] = d1div(x2, rD);
        q_aux[3] = d1div(d1div(d1c(cfg->g_mu), rD), d1c(cfg->g_re));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { qpre[q * DEV_LANES + lane] = q_acc[q]; qpre[(3 + q) * DEV_LANES + lane] = q_gc[q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { qpre[(6 + 2 * q) * DEV_LANES + lane] = q_aux[q].v; qpre[(7 + 2 * q) * DEV_LANES + lane] = q_aux[q].d; }
    if (has_grav) {  // the DCM of this stage, for phase C (its LDS buffer is recycled by the almanac wave in the pipelined loop)
#pragma unroll
        for (int q = 0; q < 9; ++q) qpre[(14 + q) * DEV_LANES + lane] = edc[q * DEV_LANES + lane];
    }
}

