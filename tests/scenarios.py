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
