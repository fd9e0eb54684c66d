"""Shared scenario builders for the tests / bench (host logic only; no oracle import here)."""
import json
import os

import numpy as np

import nyx_amd as nx
from nyx_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "two_body_leo.json")))

EARTH_RADIUS_KM = 6378.14  # data/02_config/prop_config.dhall:31 (pck08)


def earth_frame(mu):
    return nx.Frame(nx.EARTH, mu, EARTH_RADIUS_KM, None)


def two_body_setup(method, opts, mu):
    dyn = nx.SpacecraftDynamics.new(nx.OrbitalDynamics.two_body())
    prop = nx.Propagator(dyn, method, opts)
    almanac = nx.Almanac()
    central = earth_frame(mu)
    return prop, almanac, central


def leo_batch(n=1, with_stm=False):
    b = _abi.StateBatch(n, with_stm)
    b.set_rv(np.tile(np.array(GOLDEN["initial_state"]), (n, 1)))
    if with_stm:
        b.reset_stm()
    return b


# ---------------------------------------------------------------------------------------------
# BASELINE config 2: LEO Monte Carlo, 70x70 JGM3 + Sun/Moon point masses + cannonball SRP
# (SURVEY.md section 8d).  The dispersed STATES are the contract, not the RNG stream.
# ---------------------------------------------------------------------------------------------
from nyx_amd import ephem  # noqa: E402

JGM3_PATH = os.path.join(os.path.dirname(HERE), "nyx_amd", "data", "jgm3_70x70.f64")
EPOCH0_NS = ephem.EPOCH_2024_02_29_NS
_ALMANAC_CACHE = {}


def almanac_earth(span_days=40.0):
    key = ("earth", span_days)
    if key not in _ALMANAC_CACHE:
        _ALMANAC_CACHE[key] = ephem.build_almanac(nx.to_seconds(EPOCH0_NS), span_days)
    return _ALMANAC_CACHE[key]


def iau_earth_frame(mu=ephem.MU_EARTH):
    return nx.Frame(nx.EARTH, mu, EARTH_RADIUS_KM, nx.IAU_EARTH_ROTATION)


def leo_full_setup(degree=70, order=None, point_masses=(nx.SUN, nx.MOON), srp=True, method=nx.IntegratorMethod.RungeKutta89,
                   opts=None):
    """(Propagator, Almanac, central Frame) for the north-star force model."""
    order = degree if order is None else order
    almanac = almanac_earth()
    central = earth_frame(ephem.MU_EARTH)
    accel = []
    if point_masses:
        accel.append(nx.PointMasses(list(point_masses)))
    if degree and degree > 0:
        accel.append(nx.GravityFieldData.from_packed_file(JGM3_PATH, iau_earth_frame(), degree, order))
    forces = [nx.SolarPressure.default_flux(nx.EARTH)] if srp else []
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics(accel), forces)
    return nx.Propagator(dyn, method, opts or nx.IntegratorOptions()), almanac, central


def keplerian_to_cartesian(sma, ecc, inc_deg, raan_deg, aop_deg, ta_deg, mu):
    inc, raan, aop, ta = np.radians([inc_deg, raan_deg, aop_deg, ta_deg])
    p = sma * (1 - ecc ** 2)
    r = p / (1 + ecc * np.cos(ta))
    rp = np.array([r * np.cos(ta), r * np.sin(ta), 0.0])
    vp = np.sqrt(mu / p) * np.array([-np.sin(ta), ecc + np.cos(ta), 0.0])
    cO, sO, ci, si, cw, sw = np.cos(raan), np.sin(raan), np.cos(inc), np.sin(inc), np.cos(aop), np.sin(aop)
    R = np.array([[cO * cw - sO * sw * ci, -cO * sw - sO * cw * ci, sO * si],
                  [sO * cw + cO * sw * ci, -sO * sw + cO * cw * ci, -cO * si],
                  [sw * si, cw * si, ci]])
    return np.concatenate([R @ rp, R @ vp])


def leo_nominal():
    """examples/01_orbit_prop/main.rs:43-53: try_keplerian_altitude(300 km, e 0.015, i 68.5, raan 65.2, aop 75, ta 0)."""
    return keplerian_to_cartesian(EARTH_RADIUS_KM + 300.0, 0.015, 68.5, 65.2, 75.0, 0.0, ephem.MU_EARTH)


def dispersed_leo_batch(n, seed=0, nominal=None):
    """Zero-mean Gaussian sigma = (1 km x3, 1 m/s x3); dry mass 100 kg, SRP area 1 m^2, Cr 1.8."""
    rng = np.random.default_rng(seed)
    nominal = leo_nominal() if nominal is None else np.asarray(nominal)
    disp = rng.standard_normal((n, 6)) * np.array([1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3])
    b = _abi.StateBatch(n)
    b.set_rv(nominal[None, :] + disp)
    b.epoch_ns[:] = EPOCH0_NS
    b.cr[:] = 1.8
    b.cd[:] = 2.2
    b.dry_mass_kg[:] = 100.0
    b.srp_area_m2[:] = 1.0
    return b


def pos_vel_errors(a, b):
    d = a.rv() - b.rv()
    return np.linalg.norm(d[:, :3], axis=1), np.linalg.norm(d[:, 3:], axis=1)
