"""Oracle checks of the solid-tides model (dynamics/solid_tides.rs): the assertions of the reference's own unit test
(:646-688), consistency with the spherical-harmonics evaluation (the model IS a degree-3 field with time-varying
coefficients) and the gradient against central differences."""
import ctypes as C

import numpy as np

import nyx_amd as nx
import oracle_lib
from nyx_amd import _abi, ephem
from scenarios import EPOCH0_NS, dispersed_leo_batch, iau_earth_frame, leo_full_setup


def tides_only():
    prop, almanac, central = leo_full_setup(degree=0, point_masses=(), srp=False, tides=True)
    return prop, almanac, central, prop.compile(almanac, central)


def test_reference_unit_test_assertions():
    # solid_tides.rs:668-687: LEO spacecraft on the x axis; 0 < |a| < 1e-6 km/s^2; gradient's value part equals eom
    _, _, _, compiled = tides_only()
    r = np.array([7000.0, 0.0, 0.0])
    a, g, dc, ds = oracle_lib.tides_accel(compiled, EPOCH0_NS, r)
    assert 0.0 < np.linalg.norm(a) < 1e-6
    assert 1e-12 < np.linalg.norm(a) < 1e-8           # physical size of the solid-tide acceleration in LEO
    assert np.linalg.norm(g) > 0.0
    st, fx, grad = oracle_lib.dual_eom(compiled, EPOCH0_NS, np.concatenate([r, [0.0, 7.5, 0.0], [0, 0, 0]]))
    st2, dy = oracle_lib.eom(compiled, EPOCH0_NS, 0.0, np.concatenate([r, [0.0, 7.5, 0.0], [0, 0, 0]]))
    assert st == 0 and st2 == 0
    two_body = -ephem.MU_EARTH / 7000.0 ** 3 * r
    assert np.linalg.norm((fx[3:6] - two_body) - a) < 1e-18 and np.linalg.norm((dy[3:6] - two_body) - a) < 1e-18   # :684
    # only degrees 2 and 3 are populated, degree 3 by the Moon alone; orders of magnitude of the Moon/Sun tide on C20
    assert (dc[:2] == 0).all() and (ds[:2] == 0).all() and ds[2, 0] == 0 and ds[3, 0] == 0
    assert 1e-10 < abs(dc[2, 0]) < 1e-8 and abs(dc[3, 0]) < 1e-9 * abs(dc[2, 0]) * 1e9


def test_equals_a_degree3_field_with_the_delta_coefficients():
    _, _, _, compiled = tides_only()
    rng = np.random.default_rng(4)
    frame = iau_earth_frame()
    for _ in range(5):
        r = rng.standard_normal(3)
        r *= rng.uniform(6600.0, 42000.0) / np.linalg.norm(r)
        epoch = EPOCH0_NS + int(rng.uniform(0, 30 * 86400)) * nx.NS_PER_S
        a, _, dc, ds = oracle_lib.tides_accel(compiled, epoch, r)
        c = np.zeros(10); s = np.zeros(10)
        for n in (2, 3):
            for m in range(n + 1):
                c[n * (n + 1) // 2 + m], s[n * (n + 1) // 2 + m] = dc[n, m], ds[n, m]
        c[0] = 0.0   # the gravity evaluation starts at n = 1 and never reads C00
        field = nx.GravityFieldData(3, 3, c, s, frame)
        prop = nx.Propagator.default(nx.SpacecraftDynamics.new(nx.OrbitalDynamics.from_model(field)))
        comp = prop.compile(nx.Almanac(), nx.Frame(nx.EARTH, ephem.MU_EARTH, frame.mean_equatorial_radius_km, None))
        g = comp.cfg.gravity.contents
        ga = np.zeros(3)
        oracle_lib.load().nyx_oracle_gravity_accel(C.byref(g), int(epoch), np.ascontiguousarray(r).ctypes.data_as(_abi.c_double_p),
                                                   ga.ctypes.data_as(_abi.c_double_p))
        assert np.linalg.norm(ga - a) < 1e-12 * np.linalg.norm(a) + 1e-24


def test_gradient_matches_central_differences():
    _, _, _, compiled = tides_only()
    r = np.array([-2436.45, -2436.45, 6891.037])
    a, g, _, _ = oracle_lib.tides_accel(compiled, EPOCH0_NS + 5000 * nx.NS_PER_S, r)
    num = np.zeros((3, 3))
    h = 1e-2
    for j in range(3):
        dr = np.zeros(3); dr[j] = h
        ap, _, _, _ = oracle_lib.tides_accel(compiled, EPOCH0_NS + 5000 * nx.NS_PER_S, r + dr)
        am, _, _, _ = oracle_lib.tides_accel(compiled, EPOCH0_NS + 5000 * nx.NS_PER_S, r - dr)
        num[:, j] = (ap - am) / (2 * h)
    assert np.abs(g - num).max() < 1e-6 * np.abs(g).max()
    assert abs(np.trace(g)) < 1e-9 * np.abs(g).max()          # harmonic: Laplace's equation


def test_one_day_effect_is_metres():
    prop, almanac, central = leo_full_setup(degree=4, tides=True)
    prop0, _, _ = leo_full_setup(degree=4, tides=False)
    b = dispersed_leo_batch(2, seed=3)
    with_t, st = oracle_lib.propagate(prop.compile(almanac, central), b, 86400 * nx.NS_PER_S)
    without, st0 = oracle_lib.propagate(prop0.compile(almanac, central), b, 86400 * nx.NS_PER_S)
    assert (st.status == 0).all() and (st0.status == 0).all()
    d = np.linalg.norm(with_t.rv()[:, :3] - without.rv()[:, :3], axis=1)
    assert (d > 1e-4).all() and (d < 0.5).all()               # 0.1 m ... 500 m after one day in LEO
