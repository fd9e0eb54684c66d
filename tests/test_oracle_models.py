"""Oracle unit checks of the force models (CPU, no GPU)."""
import ctypes as C

import numpy as np

import nyx_amd as nx
import oracle_lib
from nyx_amd import _abi, ephem
from scenarios import EPOCH0_NS, almanac_earth, dispersed_leo_batch, leo_full_setup, leo_nominal


def _grav_struct(compiled):
    return compiled.cfg.gravity.contents


def potential(g, r_fixed):
    """Independent reference: fully-normalised geopotential U (without the central term) in the body-fixed frame."""
    from scipy.special import lpmv
    from math import factorial
    x, y, z = r_fixed
    r = np.linalg.norm(r_fixed)
    lat, lon = np.arcsin(z / r), np.arctan2(y, x)
    mu, re, N = g.mu_km3_s2, g.eq_radius_km, g.degree
    u = 0.0
    for n in range(2, N + 1):
        for m in range(0, n + 1):
            k = 1 if m == 0 else 2
            norm = np.sqrt(factorial(n - m) * (2 * n + 1) * k / factorial(n + m))
            pnm = (-1) ** m * lpmv(m, n, np.sin(lat)) * norm  # undo Condon-Shortley, normalise
            cnm, snm = g.c_nm[n * (n + 1) // 2 + m], g.s_nm[n * (n + 1) // 2 + m]
            u += mu / r * (re / r) ** n * pnm * (cnm * np.cos(m * lon) + snm * np.sin(m * lon))
    return u


def test_gravity_is_gradient_of_potential():
    prop, almanac, central = leo_full_setup(degree=12, point_masses=(), srp=False)
    compiled = prop.compile(almanac, central)
    g = _grav_struct(compiled)
    lib = oracle_lib.load()
    r = leo_nominal()[:3]
    dcm = np.zeros(9)
    lib.nyx_oracle_rotation_dcm(C.byref(g.rotation), None, EPOCH0_NS, dcm.ctypes.data_as(_abi.c_double_p), None)
    R = dcm.reshape(3, 3)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
    acc = np.zeros(3)
    lib.nyx_oracle_gravity_accel(C.byref(g), EPOCH0_NS, r.ctypes.data_as(_abi.c_double_p), acc.ctypes.data_as(_abi.c_double_p))
    h = 1e-3
    num = np.zeros(3)
    for i in range(3):
        dp, dm = r.copy(), r.copy()
        dp[i] += h
        dm[i] -= h
        num[i] = (potential(g, R @ dp) - potential(g, R @ dm)) / (2 * h)
    assert np.allclose(acc, num, rtol=0, atol=2e-11), (acc, num)
    # J2 dominates: magnitude ~ 1.5 J2 mu Re^2 / r^4
    assert 1e-6 < np.linalg.norm(acc) < 5e-5


def test_ephemeris_geometry_and_eclipse():
    prop, almanac, central = leo_full_setup(degree=0)
    compiled = prop.compile(almanac, central)
    lib = oracle_lib.load()
    pos, st = np.zeros(3), C.c_int32()
    names = {}
    for b in range(compiled.cfg.n_bodies):
        lib.nyx_oracle_body_position(C.byref(compiled.cfg), b, EPOCH0_NS, pos.ctypes.data_as(_abi.c_double_p), C.byref(st))
        assert st.value == 0
        names[compiled.cfg.bodies[b].naif_id] = (b, pos.copy())
    et = nx.to_seconds(EPOCH0_NS)
    assert np.allclose(names[nx.SUN][1], ephem.sun_geocentric(et), rtol=1e-9)
    assert np.allclose(names[nx.MOON][1], ephem.moon_geocentric(et), rtol=1e-7)
    assert 1.45e8 < np.linalg.norm(names[nx.SUN][1]) < 1.53e8 and 3.5e5 < np.linalg.norm(names[nx.MOON][1]) < 4.1e5
    # eclipse factor: 0 on the day side, 1 deep in the umbra, in (0,1) on the penumbra cone
    sun_b, earth_b = names[nx.SUN][0], names[nx.EARTH][0]
    shat = names[nx.SUN][1] / np.linalg.norm(names[nx.SUN][1])
    perp = np.cross(shat, [0, 0, 1.0])
    perp /= np.linalg.norm(perp)

    def occ(r):
        r = np.ascontiguousarray(r)
        return lib.nyx_oracle_occultation_factor(C.byref(compiled.cfg), earth_b, sun_b, EPOCH0_NS, r.ctypes.data_as(_abi.c_double_p), C.byref(st))

    # (exactly colinear geometry makes acos() see |x| > 1 by an ulp, as in the restated algorithm: offset slightly)
    assert occ(7000.0 * shat + 3.0 * perp) == 0.0
    assert occ(-7000.0 * shat + 3.0 * perp) == 1.0
    vals = [occ(-7000.0 * shat + d * perp) for d in np.linspace(6300.0, 6460.0, 33)]
    assert any(0.0 < v < 1.0 for v in vals) and vals[0] == 1.0 and vals[-1] == 0.0
    assert all(a >= b - 1e-12 for a, b in zip(vals, vals[1:]))  # monotone through the penumbra


def test_full_model_step_counts_and_stm_consistency():
    prop, almanac, central = leo_full_setup(degree=8)
    compiled = prop.compile(almanac, central)
    batch = dispersed_leo_batch(2, seed=1)
    out, st = oracle_lib.propagate(compiled, batch, 3600 * nx.NS_PER_S)
    assert (st.status == 0).all() and (out.epoch_ns == batch.epoch_ns + 3600 * nx.NS_PER_S).all()
    assert (st.n_evals == 16 * (st.n_accepted + st.n_rejected)).all()
    # dual path == real path on f(x), and A matches central differences of the real eom
    compiled_stm = prop.compile(almanac, central, stm=True)
    y9 = np.concatenate([batch.rv()[0], [1.8, 2.2, 0.0]])
    s, fx, A = oracle_lib.dual_eom(compiled_stm, EPOCH0_NS, y9, dry=100.0, srp_area=1.0)
    assert s == 0
    s, dy = oracle_lib.eom(compiled, EPOCH0_NS, 0.0, y9, dry=100.0, srp_area=1.0)
    assert s == 0 and np.allclose(fx[:6], dy[:6], rtol=0, atol=1e-18 + 1e-15 * np.abs(dy[:6]).max())
    for j in range(3):
        h = 1e-3
        yp, ym = y9.copy(), y9.copy()
        yp[j] += h
        ym[j] -= h
        _, fp = oracle_lib.eom(compiled, EPOCH0_NS, 0.0, yp, dry=100.0, srp_area=1.0)
        _, fm = oracle_lib.eom(compiled, EPOCH0_NS, 0.0, ym, dry=100.0, srp_area=1.0)
        num = (fp[3:6] - fm[3:6]) / (2 * h)
        assert np.allclose(A[3:6, j], num, rtol=1e-5, atol=1e-13), (j, A[3:6, j], num)
    # d a / d Cr (STM column 6)
    yc = y9.copy()
    yc[6] += 1e-3
    _, fc = oracle_lib.eom(compiled, EPOCH0_NS, 0.0, yc, dry=100.0, srp_area=1.0)
    assert np.allclose(A[3:6, 6], (fc[3:6] - dy[3:6]) / 1e-3, rtol=1e-6, atol=1e-20)
