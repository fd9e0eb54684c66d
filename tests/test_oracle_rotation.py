"""The orientation kinds of nyx_hip_rotation_t in the oracle: IAU phase angles with their trigonometric series (IAU_MOON's 13
terms) and Chebyshev Euler angles of a binary PCK (ITRF93 / MOON_PA style), against closed-form numpy restatements."""
import ctypes as C

import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from nyx_amd import _abi, ephem
from rotation_cases import DEG, OBLIQUITY, dcm_from_angles, euler_rotation_like, iau_angles_rad, r1
from scenarios import EPOCH0_NS, JGM3_PATH

S = nx.NS_PER_S


def oracle_dcm(compiled, rot_c, epoch_ns):
    lib = oracle_lib.load()
    dcm, rate = np.zeros(9), C.c_double()
    st = lib.nyx_oracle_rotation_dcm(C.byref(rot_c), compiled.cfg.segments, int(epoch_ns), dcm.ctypes.data_as(_abi.c_double_p), C.byref(rate))
    assert st == 0
    return dcm.reshape(3, 3), rate.value


def field_with(rot, degree=8):
    frame = nx.Frame(nx.EARTH, ephem.MU_EARTH, 6378.14, rot)
    field = nx.GravityFieldData.from_packed_file(JGM3_PATH, frame, degree, degree)
    prop = nx.Propagator(nx.SpacecraftDynamics.new(nx.OrbitalDynamics.from_model(field)), nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions())
    return prop, nx.Almanac(), nx.Frame(nx.EARTH, ephem.MU_EARTH, 6378.14, None)


def test_iau_moon_series_matches_the_published_formula():
    prop, almanac, central = field_with(nx.IAU_MOON_ROTATION)
    compiled = prop.compile(almanac, central)
    rot_c = compiled.cfg.gravity.contents.rotation
    assert rot_c.kind == _abi.ROT_IAU and rot_c.n_nut_prec == 13
    for dt_days in (0.0, 3.25, 400.0, -2000.0):
        ep = EPOCH0_NS + int(dt_days * 86400) * S
        got, rate = oracle_dcm(compiled, rot_c, ep)
        want = dcm_from_angles(iau_angles_rad(nx.IAU_MOON_ROTATION, nx.to_seconds(ep)))
        assert np.abs(got - want).max() < 1e-13
        h = 30.0
        w_num = (iau_angles_rad(nx.IAU_MOON_ROTATION, nx.to_seconds(ep) + h)[2] - iau_angles_rad(nx.IAU_MOON_ROTATION, nx.to_seconds(ep) - h)[2]) / (2 * h)
        assert abs(rate - w_num) < 1e-8 * abs(w_num)   # (central difference of an angle of ~1e4 rad: 1e-9 relative at best)
    # the series matters: the physical librations in longitude move the prime meridian by degrees
    poly = dcm_from_angles(iau_angles_rad(nx.IAU_MOON_ROTATION_POLY, nx.to_seconds(EPOCH0_NS)))
    full = dcm_from_angles(iau_angles_rad(nx.IAU_MOON_ROTATION, nx.to_seconds(EPOCH0_NS)))
    assert 0.01 < np.abs(poly - full).max() < 0.2


def test_euler_chebyshev_kind_reproduces_the_orientation_it_was_fitted_to():
    et0 = nx.to_seconds(EPOCH0_NS)
    rot_e = euler_rotation_like(nx.IAU_EARTH_ROTATION, et0 - 86400.0, 4.0)
    prop, almanac, central = field_with(rot_e)
    compiled = prop.compile(almanac, central)
    rot_c = compiled.cfg.gravity.contents.rotation
    assert rot_c.kind == _abi.ROT_EULER_CHEBY and rot_c.euler_segment == 0 and compiled.cfg.n_segments == 1
    for dt in (0.0, 1234.5, 86400.0 * 2.9):
        ep = EPOCH0_NS + int(dt * 1e9)
        got, rate = oracle_dcm(compiled, rot_c, ep)
        want = dcm_from_angles(iau_angles_rad(nx.IAU_EARTH_ROTATION, nx.to_seconds(ep)))
        assert np.abs(got - want).max() < 1e-11          # (W runs through ~6 rad per record: the degree-9 fit is good to 1e-12 rad)
        assert abs(rate - 360.9856235 * DEG / 86400.0) < 1e-13   # (2e-11 relative: derivative of the fitted series)
    # outside the segment: an error status, not an extrapolation
    lib = oracle_lib.load()
    dcm = np.zeros(9)
    assert lib.nyx_oracle_rotation_dcm(C.byref(rot_c), compiled.cfg.segments, EPOCH0_NS + 10 * 86400 * S, dcm.ctypes.data_as(_abi.c_double_p), None) == _abi.ERR_EPHEM_RANGE


def test_base_frame_rotation_is_composed():
    # an ECLIPJ2000-based file (the Earth high-precision BPCs): DCM(J2000->fixed) = E(t) * R1(obliquity)
    et0 = nx.to_seconds(EPOCH0_NS)
    base = r1(OBLIQUITY)
    rot_e = euler_rotation_like(nx.IAU_EARTH_ROTATION, et0 - 86400.0, 4.0, base=base)
    prop, almanac, central = field_with(rot_e)
    compiled = prop.compile(almanac, central)
    got, _ = oracle_dcm(compiled, compiled.cfg.gravity.contents.rotation, EPOCH0_NS)
    want = dcm_from_angles(iau_angles_rad(nx.IAU_EARTH_ROTATION, et0)) @ base
    assert np.abs(got - want).max() < 1e-11


def test_propagation_in_an_euler_oriented_frame_equals_the_iau_one():
    """The reference's ITRF93 cases (orbitaldyn.rs:934-1016, 1124-1190) run the same GravityField code with a BPC-driven
    frame; the BPC files are LFS pointers here, so the orientation is a fitted stand-in: same DCM => same trajectory."""
    from harmonics_cases import initial_batch, HGOLD
    et0 = HGOLD["epoch_et_ns"] / 1e9
    p_iau, a1, c1 = field_with(nx.IAU_EARTH_ROTATION, degree=12)
    p_eul, a2, c2 = field_with(euler_rotation_like(nx.IAU_EARTH_ROTATION, et0 - 3600.0, 1.2, ), degree=12)
    b = initial_batch(2)
    dur = 6 * 3600 * S
    r1_, s1 = oracle_lib.propagate(p_iau.compile(a1, c1), b, dur)
    r2_, s2 = oracle_lib.propagate(p_eul.compile(a2, c2), b, dur)
    assert (s1.status == 0).all() and (s2.status == 0).all()
    d = r1_.rv() - r2_.rv()
    assert np.linalg.norm(d[:, :3], axis=1).max() < 1e-7 and np.linalg.norm(d[:, 3:], axis=1).max() < 1e-10   # 0.1 mm: the fit's 1e-12 rad
    # and a run that leaves the coverage of the orientation data fails with the ephemeris-range status
    r3_, s3 = oracle_lib.propagate(p_eul.compile(a2, c2), b, 3 * 86400 * S)
    assert (s3.status == _abi.ERR_EPHEM_RANGE).all()


def test_descriptor_validation():
    from test_abi import _create_rc
    prop, almanac, central = field_with(nx.IAU_MOON_ROTATION)
    cc = prop.compile(almanac, central)
    cc.cfg.gravity.contents.rotation.n_nut_prec = 17
    assert _create_rc(cc)[0] == _abi.RC_BAD_ARG
    cc = prop.compile(almanac, central)
    cc.cfg.gravity.contents.rotation.kind = _abi.ROT_EULER_CHEBY   # no such segment
    rc, msg = _create_rc(cc)
    assert rc == _abi.RC_BAD_ARG and "euler_segment" in msg
    with pytest.raises(ValueError):
        bad = nx.Rotation(nut_prec_angles_deg=[(0.0, 1.0)] * 17)
        field_with(bad)[0].compile(nx.Almanac(), central)


def test_spin_sense_sub_synchronous_orbit_drifts_east():
    from spin_cases import check_spin_sense

    def until_event(compiled, b, max_ns, ev):
        out, st, _, _ = oracle_lib.propagate_until_event(compiled, b, max_ns, ev, trigger=1, capacity=2048)
        return out, st

    check_spin_sense(until_event)
