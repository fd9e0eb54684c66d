"""Builders for the reference-held spherical-harmonics validation cases (tests/golden/harmonics_earth.json:
nyx-core/tests/mission_design/orbitaldyn.rs:860-930, 1021-1121).  Host logic only; no oracle import here."""
import json
import os

import numpy as np

import nyx_amd as nx
from nyx_amd import _abi
from scenarios import HERE, JGM3_PATH

HGOLD = json.load(open(os.path.join(HERE, "golden", "harmonics_earth.json")))
DAY_NS = 86400 * nx.NS_PER_S


def _frames(mu):
    r_eq = HGOLD["earth_mean_equatorial_radius_km"]
    eme2k = nx.Frame(nx.EARTH, mu, r_eq, None)
    iau_earth = nx.Frame(nx.EARTH, mu, r_eq, nx.IAU_EARTH_ROTATION)
    return eme2k, iau_earth


def harmonics_only(field, opts=None, method=nx.IntegratorMethod.RungeKutta89):
    """SpacecraftDynamics::new(OrbitalDynamics::from_model(GravityField::new(field)))"""
    dyn = nx.SpacecraftDynamics.new(nx.OrbitalDynamics.from_model(field))
    return nx.Propagator(dyn, method, opts or nx.IntegratorOptions())


def j2_case():
    g = HGOLD["j2"]
    eme2k, iau_earth = _frames(g["mu_km3_s2"])
    return harmonics_only(nx.GravityFieldData.from_j2(g["j2_normalised_c20"], iau_earth)), nx.Almanac(), eme2k, g


def jgm3_case(degree=70, opts=None):
    g = HGOLD["jgm3_70x70"]
    eme2k, iau_earth = _frames(g["mu_km3_s2"])
    field = nx.GravityFieldData.from_packed_file(JGM3_PATH, iau_earth, degree, degree)
    return harmonics_only(field, opts), nx.Almanac(), eme2k, g


def initial_batch(n=1, with_stm=False):
    b = _abi.StateBatch(n, with_stm)
    b.set_rv(np.tile(np.array(HGOLD["initial_state"]), (n, 1)))
    b.epoch_ns[:] = HGOLD["epoch_et_ns"]
    if with_stm:
        b.reset_stm()
    return b


def rss_errors(rv, want):
    d = np.asarray(rv) - np.asarray(want)
    return float(np.linalg.norm(d[:3])), float(np.linalg.norm(d[3:]))
