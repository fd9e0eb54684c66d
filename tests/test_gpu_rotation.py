"""-m gpu: the orientation kinds of nyx_hip_rotation_t on the device against the oracle: IAU_MOON with its 13-term
series under a lunar field, Chebyshev Euler angles (binary-PCK style, with an ECLIPJ2000-like base rotation) under an Earth
field, with drag (which uses dW/dt), and through the STM kernel."""
import os

import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
import scenarios as sc
from nyx_amd import _abi, ephem
from rotation_cases import OBLIQUITY, euler_rotation_like, r1

pytestmark = pytest.mark.gpu
S = nx.NS_PER_S
NCPU = os.cpu_count() or 1


def both(prop, almanac, central, batch, dur, stm=False):
    compiled = prop.compile(almanac, central, stm=stm)
    ctx = nx.GpuContext(compiled)
    out, st = ctx.propagate(batch, dur)
    ctx.close()
    ref, rst = oracle_lib.propagate(compiled, batch, dur, n_threads=NCPU)
    return out, st, ref, rst


def test_lunar_field_in_iau_moon_with_the_trig_series():
    almanac = ephem.build_moon_centered_almanac(nx.to_seconds(sc.EPOCH0_NS), 40.0)
    central = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, None)
    iau_moon = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, nx.IAU_MOON_ROTATION)
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics([nx.PointMasses([nx.EARTH, nx.SUN]), sc.kaula_field(30, seed=2, frame=iau_moon)]), [])
    prop = nx.Propagator(dyn, nx.IntegratorMethod.DormandPrince78, nx.IntegratorOptions())
    b = sc.lunar_batch(70, seed=4)
    out, st, ref, rst = both(prop, almanac, central, b, 4 * 3600 * S)
    assert (st.status == 0).all() and (rst.status == 0).all()
    dr, dv = sc.pos_vel_errors(out, ref)
    print(f"IAU_MOON (13 terms), 30x30: max dr {dr.max() * 1e3:.2e} m, dv {dv.max() * 1e6:.2e} mm/s")
    assert dr.max() < 1e-6 and dv.max() < 1e-9
    # the series is felt: the polynomial-only frame puts the field elsewhere by far more than the parity bar
    iau_poly = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, nx.IAU_MOON_ROTATION_POLY)
    dyn2 = nx.SpacecraftDynamics(nx.OrbitalDynamics([nx.PointMasses([nx.EARTH, nx.SUN]), sc.kaula_field(30, seed=2, frame=iau_poly)]), [])
    ctx = nx.GpuContext(nx.Propagator(dyn2, nx.IntegratorMethod.DormandPrince78, nx.IntegratorOptions()).compile(almanac, central))
    other, _ = ctx.propagate(b, 4 * 3600 * S)
    ctx.close()
    assert sc.pos_vel_errors(other, out)[0].max() > 1e-3


@pytest.mark.parametrize("drag", [None, "exp"])
@pytest.mark.parametrize("stm", [False, True])
def test_euler_chebyshev_orientation(drag, stm):
    if drag and stm:
        pytest.skip("drag has no partials (refused like the reference's PartialsUndefined)")
    et0 = nx.to_seconds(sc.EPOCH0_NS)
    base = r1(OBLIQUITY)
    # an orientation that is NOT the IAU one w.r.t. J2000: the segment is relative to an ecliptic base frame
    rot = euler_rotation_like(nx.IAU_EARTH_ROTATION, et0 - 7200.0, 1.0, base=base)
    frame = nx.Frame(nx.EARTH, ephem.MU_EARTH, sc.EARTH_RADIUS_KM, rot)
    almanac = sc.almanac_earth()
    accel = [nx.PointMasses([nx.SUN, nx.MOON]), nx.GravityFieldData.from_packed_file(sc.JGM3_PATH, frame, 21, 21)]
    forces = [nx.SolarPressure.default_flux(nx.EARTH)] + ([nx.Drag.earth_exp(frame)] if drag else [])
    prop = nx.Propagator(nx.SpacecraftDynamics(nx.OrbitalDynamics(accel), forces), nx.IntegratorMethod.RungeKutta89,
                         nx.IntegratorOptions.with_fixed_step_s(30.0) if stm else nx.IntegratorOptions())
    b = sc.dispersed_leo_batch(70, seed=8)
    b.drag_area_m2[:] = 2.0
    if stm:
        b.stm = np.zeros((b.n, 81))
        b.reset_stm()
    out, st, ref, rst = both(prop, almanac, sc.earth_frame(ephem.MU_EARTH), b, 2 * 3600 * S, stm=stm)
    assert (st.status == 0).all() and (rst.status == 0).all()
    dr, dv = sc.pos_vel_errors(out, ref)
    print(f"Euler/Chebyshev frame, drag={drag} stm={stm}: max dr {dr.max() * 1e3:.2e} m, dv {dv.max() * 1e6:.2e} mm/s")
    assert dr.max() < 1e-6 and dv.max() < 1e-9
    if stm:
        a, r = out.stm.reshape(-1, 81), ref.stm.reshape(-1, 81)
        scale = np.maximum(np.abs(r), 1e-6 * np.abs(r).max(axis=1, keepdims=True))
        assert (np.abs(a - r) / scale).max() < 1e-9
    # leaving the coverage of the orientation data: per-trajectory status, like an ephemeris gap
    out2, st2, _, rst2 = both(prop, almanac, sc.earth_frame(ephem.MU_EARTH), b, 2 * 86400 * S, stm=stm)
    assert (st2.status == _abi.ERR_EPHEM_RANGE).all() and (rst2.status == _abi.ERR_EPHEM_RANGE).all()


def test_spin_sense_sub_synchronous_orbit_drifts_east():
    """The sense of the Earth's spin in the device's rotation_dcm (what the harmonics pins of orbitaldyn.rs do not catch)."""
    from spin_cases import check_spin_sense

    def until_event(compiled, b, max_ns, ev):
        ctx = nx.GpuContext(compiled)
        out, st, _, _ = ctx.propagate_until_event(b, max_ns, ev, trigger=1, capacity=2048)
        ctx.close()
        return out, st

    check_spin_sense(until_event)
