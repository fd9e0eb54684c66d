"""`StateParameter` and `Spacecraft::value` for arrays of states.

Host-side mirror of what the Monte Carlo results hand to their consumers (`Results::every_value_of`,
`first_values_of`, ... nyx-core/src/mc/results.rs:86-245): the value of one state parameter for every resampled
state of every run.  `Spacecraft::value` (cosmic/spacecraft.rs:520-578) answers Cd / Cr / masses itself and
forwards `StateParameter::Element(e)` to ANISE's `OrbitalElement::evaluate` (the `analysis` feature, not part of
this reference tree): the Cartesian components and magnitudes are exact restatements; the Keplerian elements
follow ANISE's documented definitions (osculating elements from the Cartesian state and the frame's mu,
angles in degrees in [0, 360)) and are **parity unpinned** like the event scalars.
"""
from __future__ import annotations

import enum

import numpy as np


class StateParameter(enum.Enum):
    """md/param.rs:34-: the parameters this path can evaluate (Element(..) flattened to its element)."""

    X = "X"
    Y = "Y"
    Z = "Z"
    VX = "VX"
    VY = "VY"
    VZ = "VZ"
    Rmag = "Rmag"
    Vmag = "Vmag"
    Hmag = "Hmag"
    Energy = "Energy"
    SemiMajorAxis = "SemiMajorAxis"
    Eccentricity = "Eccentricity"
    Inclination = "Inclination"
    RAAN = "RAAN"
    AoP = "AoP"
    TrueAnomaly = "TrueAnomaly"
    Period = "Period"
    ApoapsisRadius = "ApoapsisRadius"
    PeriapsisRadius = "PeriapsisRadius"
    Cr = "Cr"
    Cd = "Cd"
    DryMass = "DryMass"
    PropMass = "PropMass"
    TotalMass = "TotalMass"
    # known to the reference, not available from a propagated ballistic state (StateError::Unavailable / NoThrusterAvail)
    Isp = "Isp"
    Thrust = "Thrust"


class StateError(Exception):
    """StateError::Unavailable { param } (cosmic/mod.rs)."""

    def __init__(self, param: StateParameter):
        super().__init__(f"{param.name} is unavailable for this state")
        self.param = param


_CART = {StateParameter.X: 0, StateParameter.Y: 1, StateParameter.Z: 2, StateParameter.VX: 3, StateParameter.VY: 4,
         StateParameter.VZ: 5}


def _wrap360(a):
    a = np.mod(a, 360.0)
    return np.where(a < 0.0, a + 360.0, a)


def state_value(param: StateParameter, rv: np.ndarray, mu_km3_s2: float, cr=None, cd=None, dry_mass_kg=None, prop_mass_kg=None,
                extra_mass_kg=None) -> np.ndarray:
    """Value of `param` for every row of `rv` ([..., 6], km and km/s).  Spacecraft-level parameters (Cr, Cd, masses)
    broadcast against the leading dimensions of `rv`; raises StateError for parameters a ballistic state does not have."""
    rv = np.asarray(rv, dtype=np.float64)
    lead = rv.shape[:-1]
    if param in _CART:
        return rv[..., _CART[param]].copy()
    sc = {StateParameter.Cr: cr, StateParameter.Cd: cd, StateParameter.DryMass: dry_mass_kg, StateParameter.PropMass: prop_mass_kg}
    if param in sc:
        if sc[param] is None:
            raise StateError(param)
        return np.broadcast_to(np.asarray(sc[param], dtype=np.float64), lead).copy()
    if param is StateParameter.TotalMass:
        if dry_mass_kg is None or prop_mass_kg is None:
            raise StateError(param)
        extra = 0.0 if extra_mass_kg is None else extra_mass_kg
        return np.broadcast_to(np.asarray(dry_mass_kg) + np.asarray(prop_mass_kg) + np.asarray(extra), lead).copy()
    if param in (StateParameter.Isp, StateParameter.Thrust):
        raise StateError(param)
    r, v = rv[..., :3], rv[..., 3:]
    rmag = np.linalg.norm(r, axis=-1)
    vmag = np.linalg.norm(v, axis=-1)
    if param is StateParameter.Rmag:
        return rmag
    if param is StateParameter.Vmag:
        return vmag
    h = np.cross(r, v)
    hmag = np.linalg.norm(h, axis=-1)
    if param is StateParameter.Hmag:
        return hmag
    energy = 0.5 * vmag * vmag - mu_km3_s2 / rmag
    if param is StateParameter.Energy:
        return energy
    sma = -mu_km3_s2 / (2.0 * energy)
    if param is StateParameter.SemiMajorAxis:
        return sma
    evec = ((vmag * vmag - mu_km3_s2 / rmag)[..., None] * r - np.sum(r * v, axis=-1)[..., None] * v) / mu_km3_s2
    ecc = np.linalg.norm(evec, axis=-1)
    if param is StateParameter.Eccentricity:
        return ecc
    if param is StateParameter.ApoapsisRadius:
        return sma * (1.0 + ecc)
    if param is StateParameter.PeriapsisRadius:
        return sma * (1.0 - ecc)
    if param is StateParameter.Period:
        return 2.0 * np.pi * np.sqrt(sma ** 3 / mu_km3_s2)
    if param is StateParameter.Inclination:
        return np.degrees(np.arccos(np.clip(h[..., 2] / hmag, -1.0, 1.0)))
    n = np.stack([-h[..., 1], h[..., 0], np.zeros_like(hmag)], axis=-1)  # z x h
    if param is StateParameter.RAAN:
        return _wrap360(np.degrees(np.arctan2(n[..., 1], n[..., 0])))
    if param is StateParameter.AoP:
        # angle from the node to the eccentricity vector, measured in the orbit plane around h
        cosw = np.sum(n * evec, axis=-1)
        sinw = np.sum(np.cross(n, evec) * h, axis=-1) / hmag
        return _wrap360(np.degrees(np.arctan2(sinw, cosw)))
    if param is StateParameter.TrueAnomaly:
        # atan2 form, as the device's event scalar (event_dev.h): well conditioned at the apsides
        cost = np.sum(evec * r, axis=-1)
        sint = np.sum(np.cross(evec, r) * h, axis=-1) / hmag
        return _wrap360(np.degrees(np.arctan2(sint, cost)))
    raise StateError(param)
