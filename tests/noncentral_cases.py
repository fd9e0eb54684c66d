"""Shared set-up of the non-central gravity field tests (GravityField::eom transforms the orbit into `grav_data.frame` whatever its
centre: gravity_field.rs:150-154, and only rotates the acceleration back: :258-265).  One physical system, two formulations:

  A  Moon-centred integration: two-body Moon + the Moon's field (central, the path pinned on the reference's vectors) + Earth / Sun
     point masses;
  B  Earth-centred integration: two-body Earth + Moon / Sun point masses + THE SAME Moon field, now of a non-central body
     (`nyx_hip_gravity_field_t.offset_body`), the states given and returned Moon-centred (opts.integration_frame).

They differ by the Moon's harmonics pulling on the Earth, which neither models (1e-13 m/s^2), and by integration error."""
import nyx_amd as nx
from nyx_amd import ephem
from scenarios import EPOCH0_NS, _ALMANAC_CACHE, almanac_earth, earth_frame, kaula_field, lunar_batch

MOON_FRAME = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, None)
IAU_MOON = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, nx.IAU_MOON_ROTATION_POLY)


def moon_centred(degree, method=nx.IntegratorMethod.RungeKutta89):
    key = ("moon", 40.0)
    if key not in _ALMANAC_CACHE:
        _ALMANAC_CACHE[key] = ephem.build_moon_centered_almanac(nx.to_seconds(EPOCH0_NS), 40.0)
    accel = [nx.PointMasses([nx.EARTH, nx.SUN])] + ([kaula_field(degree, seed=1, frame=IAU_MOON)] if degree else [])
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics(accel), [])
    return nx.Propagator(dyn, method, nx.IntegratorOptions()), _ALMANAC_CACHE[key], MOON_FRAME


def earth_centred(degree, method=nx.IntegratorMethod.RungeKutta89):
    accel = [nx.PointMasses([nx.MOON, nx.SUN])] + ([kaula_field(degree, seed=1, frame=IAU_MOON)] if degree else [])
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics(accel), [])
    return nx.Propagator(dyn, method, nx.IntegratorOptions()), almanac_earth(), earth_frame(ephem.MU_EARTH)


def batch(n, seed=0):
    return lunar_batch(n, seed=seed)


def two_fields(centre, deg_earth, deg_moon, method=nx.IntegratorMethod.RungeKutta89):
    """The reference's cislunar stacking: the Earth's AND the Moon's field in one OrbitalDynamics (accel_models is a list), integrated
    around `centre` ("earth" or "moon"); the other body's field is the non-central one."""
    from scenarios import JGM3_PATH, iau_earth_frame
    earth_field = nx.GravityFieldData.from_packed_file(JGM3_PATH, iau_earth_frame(), deg_earth, deg_earth)
    moon_field = kaula_field(deg_moon, seed=1, frame=IAU_MOON)
    if centre == "earth":
        accel = [nx.PointMasses([nx.MOON, nx.SUN]), earth_field, moon_field]
        almanac, frame = almanac_earth(), earth_frame(ephem.MU_EARTH)
    else:
        key = ("moon", 40.0)
        if key not in _ALMANAC_CACHE:
            _ALMANAC_CACHE[key] = ephem.build_moon_centered_almanac(nx.to_seconds(EPOCH0_NS), 40.0)
        accel = [nx.PointMasses([nx.EARTH, nx.SUN]), moon_field, earth_field]
        almanac, frame = _ALMANAC_CACHE[key], MOON_FRAME
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics(accel), [])
    return nx.Propagator(dyn, method, nx.IntegratorOptions()), almanac, frame
