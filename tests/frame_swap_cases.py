"""Shared set-up of the integration-frame swap tests (opts.integration_frame, propagators/instance.rs:117-142, 211-220): states
centred on the Moon, dynamics integrated in the Earth frame (two-body Earth + Sun / Moon point masses + SRP)."""
import numpy as np
from numpy.polynomial import chebyshev as _cheb

import nyx_amd as nx
from nyx_amd import ephem
from scenarios import EPOCH0_NS, almanac_earth, earth_frame, keplerian_to_cartesian

MOON_FRAME = nx.Frame(nx.MOON, ephem.MU_MOON, ephem.R_MOON, None)


def setup():
    almanac = almanac_earth()
    earth = earth_frame(ephem.MU_EARTH)
    dyn = nx.SpacecraftDynamics(nx.OrbitalDynamics([nx.PointMasses([nx.SUN, nx.MOON])]), [nx.SolarPressure.default_flux(nx.EARTH)])
    prop = nx.Propagator(dyn, nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions())
    return prop, almanac, earth


def moon_batch(n, seed=0):
    """Low lunar orbits, Moon-centred, J2000 orientation; ragged start epochs."""
    rng = np.random.default_rng(seed)
    b = nx._abi.StateBatch(n)
    nominal = keplerian_to_cartesian(1900.0, 0.02, 60.0, 20.0, 40.0, 10.0, ephem.MU_MOON)
    b.set_rv(nominal[None, :] + rng.standard_normal((n, 6)) * np.array([1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3]))
    b.epoch_ns[:] = EPOCH0_NS + rng.integers(0, 3600, size=n) * nx.NS_PER_S
    b.cr[:] = 1.5
    b.dry_mass_kg[:] = 200.0
    b.srp_area_m2[:] = 2.0
    return b


def chain_state_numpy(almanac, naif_id, epoch_ns):
    """Position and velocity of a body w.r.t. the almanac's centre from numpy's Chebyshev evaluation and DERIVATIVE: an
    independent check of the CHBINT-style recurrence the device and the oracle use."""
    et = epoch_ns / 1e9
    r, v = np.zeros(3), np.zeros(3)
    for seg_idx, sign in almanac.bodies[naif_id]["chain"]:
        seg = almanac.segments[seg_idx]
        idx = min(int(np.floor((et - seg.init_et_s) / seg.interval_s)), seg.records.shape[0] - 1)
        rec = seg.records[idx]
        t = (et - rec[0]) / rec[1]
        for c in range(3):
            cf = rec[2 + c * seg.n_coeffs: 2 + (c + 1) * seg.n_coeffs]
            r[c] += sign * _cheb.chebval(t, cf)
            v[c] += sign * _cheb.chebval(t, _cheb.chebder(cf)) / rec[1]
    return r, v
