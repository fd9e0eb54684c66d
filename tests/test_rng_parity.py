"""The random stream of the Monte Carlo front end against the reference's own seeded known-answer tests
(nyx-core/src/mc/multivariate.rs, `multivariate_ut`): `Pcg64Mcg::new(0)`, 1 000 samples, EXACT counts.  They pin
`rand_pcg::Pcg64Mcg`, rand_distr's ziggurat `StandardNormal` (tables included) and the order of the nine draws per state
as restated in nyx_amd/rng.py - crates that are not part of the reference tree."""
import numpy as np

import nyx_amd as nx
from nyx_amd.params import StateParameter as P
from scenarios import EPOCH0_NS, earth_frame

GMAT_EARTH_GM = 398_600.441_5


def keplerian(sma, ecc, inc_deg, raan_deg, aop_deg, ta_deg, mu):
    inc, raan, aop, ta = (np.radians(v) for v in (inc_deg, raan_deg, aop_deg, ta_deg))
    p = sma * (1 - ecc * ecc)
    r = p / (1 + ecc * np.cos(ta))
    rp = np.array([r * np.cos(ta), r * np.sin(ta), 0.0])
    vp = np.sqrt(mu / p) * np.array([-np.sin(ta), ecc + np.cos(ta), 0.0])

    def rot(axis, ang):
        c, s = np.cos(ang), np.sin(ang)
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) if axis == 3 else np.array([[1, 0, 0], [0, c, -s], [0, s, c]])

    m = rot(3, raan) @ rot(1, inc) @ rot(3, aop)
    return np.concatenate([m @ rp, m @ vp])


def template():
    frame = earth_frame(GMAT_EARTH_GM)
    return nx.Spacecraft(EPOCH0_NS, keplerian(8_191.93, 1e-6, 12.85, 306.614, 314.19, 99.887_7, GMAT_EARTH_GM), frame)


def test_pcg64mcg_is_the_published_generator():
    """First outputs of pcg_mcg_128_xsl_rr_64 seeded with state 0 | 1 (multiplier 0x2360ED051FC65DA44385DF649FCCF645)."""
    r = nx.Pcg64Mcg(0)
    state, out = 1, []
    for _ in range(3):
        state = (state * 0x2360ED051FC65DA44385DF649FCCF645) % (1 << 128)
        x, rot = ((state >> 64) ^ state) & ((1 << 64) - 1), state >> 122
        out.append(((x >> rot) | (x << (64 - rot))) & ((1 << 64) - 1))
    assert [r.next_u64() for _ in range(3)] == out
    # 0 and 1 seed the same stream (state | 1), as in rand_pcg
    a, b = nx.Pcg64Mcg(0), nx.Pcg64Mcg(1)
    assert [a.next_u64() for _ in range(4)] == [b.next_u64() for _ in range(4)]


def test_disperse_r_mag_known_answer():
    """multivariate.rs:420-476: Rmag dispersed with sigma = 1 km, seed 0: exactly 6 of 1 000 samples are 3 sigma or more
    away ("Mathematically, this should be 3!")."""
    t = template()
    gen = nx.MvnSpacecraft.new(t, [nx.StateDispersion(P.Rmag, std_dev=1.0)])
    mc = nx.MonteCarlo(gen, seed=0)
    init = np.linalg.norm(np.asarray(t.rv)[:3])
    too_far = sum(1 for _, s in mc.generate_states(0, 1000) if abs(init - np.linalg.norm(s.rv[:3])) >= 3.0)
    assert too_far == 6


def test_disperse_full_cartesian_known_answer():
    """multivariate.rs:478-556: the six Cartesian components dispersed independently, seed 0: the number of components more
    than one sigma from nominal, over 1 000 samples, divided by 6, is exactly 312."""
    t = template()
    std = [10.0, 10.0, 10.0, 0.2, 0.2, 0.2]
    gen = nx.MvnSpacecraft.new(t, [nx.StateDispersion(p, std_dev=s) for p, s in zip([P.X, P.Y, P.Z, P.VX, P.VY, P.VZ], std)])
    mc = nx.MonteCarlo(gen, seed=0)
    nominal = np.asarray(t.rv)
    cnt = sum(int((np.abs(s.rv - nominal) > std).sum()) for _, s in mc.generate_states(0, 1000))
    assert cnt // 6 == 312
    # resume: the stream is consumed sample by sample, so skipping reproduces the tail (montecarlo.rs:290-295)
    full = mc.generate_states(0, 12)
    tail = mc.generate_states(8, 4)
    assert [i for i, _ in tail] == [0, 1, 2, 3]
    np.testing.assert_array_equal([s.rv for _, s in tail], [s.rv for _, s in full[8:]])


def test_disperse_raan_only_statistics():
    """multivariate.rs:560-640: RAAN dispersed by 0.2 deg: SMA and inclination move by less than 5 %, and the 95th
    percentile of the squared Mahalanobis distance of the applied RAAN dispersions matches chi-squared(1) within 20 %."""
    frame = earth_frame(GMAT_EARTH_GM)
    t = nx.Spacecraft(EPOCH0_NS, keplerian(8_100.0, 1e-6, 12.85, 356.614, 14.19, 199.887_7, GMAT_EARTH_GM), frame)
    gen = nx.MvnSpacecraft.new(t, [nx.StateDispersion.zero_mean(P.RAAN, 0.2)])
    states = nx.MonteCarlo(gen, seed=0).generate_states(0, 1000)
    md = []
    for _, s in states:
        for param in (P.SemiMajorAxis, P.Inclination):
            orig, new = (float(nx.state_value(param, np.asarray(x.rv), GMAT_EARTH_GM)) for x in (t, s))
            assert 100.0 * abs(orig - new) / orig < 5.0
        assert s.actual_dispersions[0][0] is P.RAAN
        d = s.actual_dispersions[0][1]
        d = (d + 180.0) % 360.0 - 180.0
        md.append(d * d / 0.2 ** 2)
    p95 = sorted(md)[950]
    assert abs(p95 - 3.841458820694124) / 3.841458820694124 < 0.2     # chi-squared(1).inverse_cdf(0.95)


def test_disperse_keplerian_statistics():
    """multivariate.rs:632-720: SMA, inclination, RAAN and AoP dispersed together; 2 000 samples from seed 0: the sample mean
    stays within 1 km / km/s of nominal (norm) and the sample covariance within 20 % (Frobenius) of the generator's own
    Cartesian covariance L L^T."""
    frame = earth_frame(GMAT_EARTH_GM)
    t = nx.Spacecraft(EPOCH0_NS, keplerian(8_100.0, 1e-6, 12.85, 356.614, 14.19, 199.887_7, GMAT_EARTH_GM), frame)
    gen = nx.MvnSpacecraft.new(t, [nx.StateDispersion.zero_mean(P.SemiMajorAxis, 10.0), nx.StateDispersion.zero_mean(P.Inclination, 0.15),
                                   nx.StateDispersion.zero_mean(P.RAAN, 0.02), nx.StateDispersion.zero_mean(P.AoP, 0.02)])
    expected = (gen._sqrt_s_v @ gen._sqrt_s_v.T)[:6, :6]
    x = np.array([s.rv for _, s in nx.MonteCarlo(gen, seed=0).generate_states(0, 2000)])
    assert np.linalg.norm(x.mean(axis=0) - np.asarray(t.rv)) < 1.0
    d = x - x.mean(axis=0)
    sample_cov = d.T @ d / (len(x) - 1)
    assert np.linalg.norm(sample_cov - expected) / np.linalg.norm(expected) < 0.2
    # the requested sigmas come back out of the linearised map: J C J^T = diag(sigma^2) - for SMA, inclination and RAAN; the
    # argument of periapsis of an e = 1e-6 orbit has partials ~1e7 deg per km/s and drowns in the rounding of the 9x9 SVD
    from nyx_amd.mc import _partials
    jac = np.stack([_partials(p, np.asarray(t.rv), GMAT_EARTH_GM) for p in (P.SemiMajorAxis, P.Inclination, P.RAAN)])
    np.testing.assert_allclose(jac @ expected @ jac.T, np.diag([100.0, 0.15 ** 2, 0.02 ** 2]), rtol=1e-5, atol=1e-7)
