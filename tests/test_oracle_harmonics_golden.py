"""Pins the oracle's spherical harmonics (rotation sign, coefficient indexing, normalisation, body-fixed round trip)
on the vectors the REFERENCE's tests hold for exactly this term: Monte (J2) and GMAT (JGM3 70x70),
nyx-core/tests/mission_design/orbitaldyn.rs:860-930, 1021-1121, with the reference's own tolerances."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from harmonics_cases import DAY_NS, HGOLD, initial_batch, j2_case, jgm3_case, rss_errors


def test_jgm3_fixture_is_the_reference_file():
    # the packed fixture the cases read == the reference's loader on the reference's file (io/gravity.rs:533-567)
    import os
    ref = "/root/reference/data/01_planetary/JGM3.cof.gz"
    if not os.path.exists(ref):
        pytest.skip("reference checkout absent")
    prop, _, central, _ = jgm3_case()
    f = prop.dynamics.orbital_dyn.accel_models[0]
    g = nx.GravityFieldData.from_cof(ref, 70, 70, True, f.frame)
    assert (g.degree, g.order) == (f.degree, f.order) == (70, 70)
    np.testing.assert_array_equal(g.c_nm, f.c_nm)
    np.testing.assert_array_equal(g.s_nm, f.s_nm)


def test_j2_monte():
    prop, almanac, central, g = j2_case()
    out, st = oracle_lib.propagate(prop.compile(almanac, central), initial_batch(), DAY_NS)
    assert st.status[0] == 0
    err_r, err_v = rss_errors(out.rv()[0], g["state_monte"])
    assert err_r < g["tol_r_km"], err_r
    assert err_v < g["tol_v_km_s"], err_v
    # "GMAT and Monte are within about 0.1 meters ... we're checking we're in the same bracket" (orbitaldyn.rs:861)
    gr, gv = rss_errors(out.rv()[0], g["state_gmat_commented_out"])
    print("J2 vs Monte %.3e km %.3e km/s; vs GMAT %.3e km %.3e km/s" % (err_r, err_v, gr, gv))


def _monte_error(shift_s=0.0, pole=None):
    prop, almanac, central, g = j2_case()
    if pole is not None:
        f = prop.dynamics.orbital_dyn.accel_models[0]
        r = nx.IAU_EARTH_ROTATION
        f.frame = nx.Frame(f.frame.naif_id, f.frame.mu_km3_s2, f.frame.mean_equatorial_radius_km,
                           nx.Rotation((pole[0], r.ra_deg[1], 0.0), (pole[1], r.dec_deg[1], 0.0), r.w_deg))
    b = initial_batch()
    b.epoch_ns[:] += int(round(shift_s * 1e9))
    out, _ = oracle_lib.propagate(prop.compile(almanac, central), b, DAY_NS)
    return rss_errors(out.rv()[0], g["state_monte"])[0]


def test_what_the_j2_margin_depends_on():
    """The J2 pin passes at 19.59 m of the reference's 20 m.  It is NOT at the mercy of the epoch model: a zonal field is
    axially symmetric, so the prime meridian W(t) does not enter - the TDB-TT term dropped in the fixture (< 2 ms) moves the
    result by < 1e-7 m, confusing TAI with TT (32.184 s) by 0.2 mm, a whole hour by 2 cm.  What it DOES depend on is the
    direction of the pole: the two-angle IAU polynomial puts it on the J2000 z axis at this epoch (to 4e-4 arcsec), Monte's
    and GMAT's Earth-fixed frames include nutation (~8 arcsec of pole offset on 2000-01-01), and 8 arcsec of tilt move the
    result by 3 ... 100 m depending on its azimuth.  The modelling term behind the 19.6 m is therefore the pole model, shared
    with the reference (whose own tolerance for this very comparison is 20 m), not time."""
    base = _monte_error()
    assert abs(base - 0.019588) < 2e-5
    for shift in (2e-3, -32.184, 3600.0):
        assert abs(_monte_error(shift_s=shift) - base) < 3e-5, shift          # < 3 cm for an HOUR of epoch error
    tilt = 8.0 / 3600.0
    moved = [abs(_monte_error(pole=(az, 90.0 - tilt)) - base) for az in (133.84, -133.84, 46.16, -46.16)]
    assert min(moved) > 1e-3 and max(moved) > 0.05                            # metres to a hundred metres for 8 arcsec of pole


@pytest.mark.parametrize("with_stm", [False, True])
def test_jgm3_70x70_gmat(with_stm):
    # with_stm=True is what the test NAMED _partials would exercise (in this snapshot it does not call with_stm());
    # the STM-enabled run must land on the same GMAT vector
    prop, almanac, central, g = jgm3_case()
    out, st = oracle_lib.propagate(prop.compile(almanac, central, stm=with_stm), initial_batch(with_stm=with_stm), DAY_NS)
    assert st.status[0] == 0
    err_r, err_v = rss_errors(out.rv()[0], g["state_gmat"])
    print("70x70 stm=%s vs GMAT %.3e km %.3e km/s, %d steps" % (with_stm, err_r, err_v, st.n_accepted[0]))
    assert err_r < g["tol_r_km"], err_r
    assert err_v < g["tol_v_km_s"], err_v


def _with_w(f, w0, w1=360.9856235):
    f.frame = nx.Frame(f.frame.naif_id, f.frame.mu_km3_s2, f.frame.mean_equatorial_radius_km,
                       nx.Rotation(nx.IAU_EARTH_ROTATION.ra_deg, nx.IAU_EARTH_ROTATION.dec_deg, (w0, w1, 0.0)))


def _gmat_error(mod=None, degree=70):
    prop, almanac, central, g = jgm3_case(degree)
    if mod is not None:
        mod(prop.dynamics.orbital_dyn.accel_models[0])
    out, _ = oracle_lib.propagate(prop.compile(almanac, central), initial_batch(), DAY_NS)
    return rss_errors(out.rv()[0], g["state_gmat"])[0]


def test_sensitivity_of_the_pin():
    """A tolerance is only a pin if the mistakes it should catch move the answer by more than it.  Measured (km vs GMAT):
    as is 0.113; S_nm sign 5.4; prime meridian 90 deg off 3.6; zonals only 1.9; C22/S22 dropped 0.60; truncated at 12x12
    0.21 -- all FAIL the reference's 0.2 km.  NOT caught by this vector: the sense of the Earth's spin (0.101 km with W
    running backwards: the tesseral m-dailies of this near-polar orbit average out over 15 revolutions either way);
    that is fixed by the IAU definition restated in rotation_dcm and by the next test."""
    tol = HGOLD["jgm3_70x70"]["tol_r_km"]
    assert _gmat_error() < tol

    def neg_s(f):
        f.s_nm = -f.s_nm

    def zonals_only(f):
        for n in range(f.degree + 1):
            for m in range(1, n + 1):
                f.c_nm[n * (n + 1) // 2 + m] = 0.0
                f.s_nm[n * (n + 1) // 2 + m] = 0.0

    def no_c22(f):
        f.c_nm[5] = f.s_nm[5] = 0.0

    assert _gmat_error(neg_s) > 10 * tol
    assert _gmat_error(lambda f: _with_w(f, 190.147 + 90.0)) > 10 * tol
    assert _gmat_error(zonals_only) > 5 * tol
    assert _gmat_error(no_c22) > 2 * tol
    assert _gmat_error(degree=12) > tol


def test_residual_is_the_iau_prime_meridian():
    """Where the 0.113 km comes from (not a reference-held number, a consistency check on the pin): the IAU model puts the
    prime meridian at 90 + alpha0 + W = 280.147 deg at J2000, GMAT's FK5 chain at GMST(J2000) = 280.4606 deg (IAU 1982).
    With that 0.3136 deg added to W0 the same code lands 16 m from GMAT after one day of 70x70 (nutation, polar motion,
    UT1-UTC remain), and shifting further away from it makes it worse again."""
    e = _gmat_error(lambda f: _with_w(f, 190.147 + 0.3136))
    assert e < 0.03, e
    assert _gmat_error(lambda f: _with_w(f, 190.147 + 0.3136 + 0.05)) > e
    assert _gmat_error(lambda f: _with_w(f, 190.147 + 0.3136 - 0.05)) > e


def test_real_and_dual_eoms_agree():
    # second half of val_earth_sph_harmonics_12x12 (orbitaldyn.rs:985-1015): fixed 30 s steps, 6 h, the run with the
    # hyperdual EOMs (STM on) must give the same orbit as the plain run to f64::EPSILON
    c = HGOLD["jgm3_12x12_itrf93"]
    prop, almanac, central, _ = jgm3_case(degree=12, opts=nx.IntegratorOptions.with_fixed_step_s(c["fixed_step_s"]))
    dur = c["dual_duration_s"] * nx.NS_PER_S
    real, _ = oracle_lib.propagate(prop.compile(almanac, central), initial_batch(), dur)
    dual, _ = oracle_lib.propagate(prop.compile(almanac, central, stm=True), initial_batch(with_stm=True), dur)
    err_r, err_v = rss_errors(real.rv()[0], dual.rv()[0])
    assert err_r < np.finfo(float).eps and err_v < np.finfo(float).eps, (err_r, err_v)
