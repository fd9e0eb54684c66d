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
                   opts=None, drag=None, tides=False):
    """(Propagator, Almanac, central Frame) for the north-star force model."""
    order = degree if order is None else order
    almanac = almanac_earth()
    central = earth_frame(ephem.MU_EARTH)
    accel = []
    if point_masses:
        accel.append(nx.PointMasses(list(point_masses)))
    if degree and degree > 0:
        accel.append(nx.GravityFieldData.from_packed_file(JGM3_PATH, iau_earth_frame(), degree, order))
    if tides:  # third accel model of Dynamics::build (dynamics/sequence/config.rs:116-118)
        accel.append(nx.SolidTides.earth_moon_system(iau_earth_frame(), nx.MOON, nx.SUN))
    forces = [nx.SolarPressure.default_flux(nx.EARTH)] if srp else []
    if drag == "exp":
        forces.append(nx.Drag.earth_exp(iau_earth_frame()))          # drag.rs:128-143
    elif drag == "stdatm":
        forces.append(nx.Drag.std_atm1976(iau_earth_frame()))        # drag.rs:145-160
    elif drag == "const":
        forces.append(nx.Drag(("constant", 1e-12), iau_earth_frame()))
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


# ---------------------------------------------------------------------------------------------
# BASELINE config 3 (JWST-like, point masses + SRP with Earth and Moon shadows) and
# config 5 (low lunar orbit, 150x150 synthetic field + Earth/Sun point masses)
# ---------------------------------------------------------------------------------------------

def jwst_setup(opts=None, method=nx.IntegratorMethod.RungeKutta89, jupiter=True):
    """examples/02_jwst_covar_monte_carlo/main.rs:99-146: Sun, Moon (, Jupiter barycentre) point masses + SRP with
    Earth and Moon shadows."""
    almanac = almanac_earth()
    central = earth_frame(ephem.MU_EARTH)
    bodies = [nx.MOON, nx.SUN] + ([nx.JUPITER_BARYCENTER] if jupiter else [])
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics([nx.PointMasses(bodies)]), [nx.SolarPressure([nx.EARTH, nx.MOON])])
    return nx.Propagator(dyn, method, opts or nx.IntegratorOptions()), almanac, central


def jwst_batch(n, seed=0):
    """README.md:52 state; m = 6200 kg, A = 21.197 x 14.162 m^2, Cr 1.56 (main.rs:63-70); RIC sigmas (0.5, 0.3, 1.5) km,
    (1e-4, 6e-4, 3e-3) km/s (main.rs:77-86) rotated to inertial."""
    r0 = np.array([119901.07, -1389299.67, -1041369.15])
    v0 = np.array([0.045956, -0.013168, 0.034535])
    rh = r0 / np.linalg.norm(r0)
    ch = np.cross(r0, v0)
    ch /= np.linalg.norm(ch)
    ih = np.cross(ch, rh)
    dcm = np.stack([rh, ih, ch], axis=1)  # RIC -> inertial
    rng = np.random.default_rng(seed)
    dr = rng.standard_normal((n, 3)) * np.array([0.5, 0.3, 1.5])
    dv = rng.standard_normal((n, 3)) * np.array([1e-4, 6e-4, 3e-3])
    b = _abi.StateBatch(n)
    b.set_rv(np.concatenate([r0[None, :] + dr @ dcm.T, v0[None, :] + dv @ dcm.T], axis=1))
    b.epoch_ns[:] = EPOCH0_NS
    b.cr[:] = 1.56
    b.dry_mass_kg[:] = 6200.0
    b.srp_area_m2[:] = 21.197 * 14.162
    return b


def kaula_field(degree, seed=0, frame=None):
    """Synthetic fully-normalised lunar-like field (the GRGM/JGGRX file is a missing blob): Kaula's rule
    sigma_n = 2.5e-4 / n^2, seeded; C20, C22 set to the lunar values."""
    rng = np.random.default_rng(seed)
    n_tot = (degree + 1) * (degree + 2) // 2
    c, s = np.zeros(n_tot), np.zeros(n_tot)
    for n in range(2, degree + 1):
        sig = 2.5e-4 / n ** 2
        for m in range(n + 1):
            c[n * (n + 1) // 2 + m] = sig * rng.standard_normal()
            s[n * (n + 1) // 2 + m] = 0.0 if m == 0 else sig * rng.standard_normal()
    c[3], c[5] = -9.088e-5, 3.467e-5
    return nx.GravityFieldData(degree, degree, c, s, frame)


def lunar_setup(degree=150, opts=None, method=nx.IntegratorMethod.DormandPrince78):
    key = ("moon", 40.0)
    if key not in _ALMANAC_CACHE:
        _ALMANAC_CACHE[key] = ephem.build_moon_centered_almanac(nx.to_seconds(EPOCH0_NS), 40.0)
    almanac = _ALMANAC_CACHE[key]
    central = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, None)
    iau_moon = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, nx.IAU_MOON_ROTATION_POLY)
    accel = [nx.PointMasses([nx.EARTH, nx.SUN]), kaula_field(degree, seed=1, frame=iau_moon)]
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics(accel), [])
    return nx.Propagator(dyn, method, opts or nx.IntegratorOptions()), almanac, central


def lunar_batch(n, seed=0):
    """~50 km altitude polar LLO (LRO-like, examples/04_lro_od/main.rs:121-145), sigma 100 m / 0.1 m/s."""
    nominal = keplerian_to_cartesian(ephem.R_MOON + 50.0, 0.002, 89.5, 30.0, 0.0, 40.0, ephem.MU_MOON)
    rng = np.random.default_rng(seed)
    b = _abi.StateBatch(n)
    b.set_rv(nominal[None, :] + rng.standard_normal((n, 6)) * np.array([0.1, 0.1, 0.1, 1e-4, 1e-4, 1e-4]))
    b.epoch_ns[:] = EPOCH0_NS
    b.dry_mass_kg[:] = 1000.0
    return b
