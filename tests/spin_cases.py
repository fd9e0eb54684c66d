"""The sense of the central body's spin, checked without a reference vector (the harmonics pins do not catch a reversed
W): a circular equatorial orbit a little BELOW the synchronous radius runs ahead of the Earth, so its sub-satellite longitude in
the body-fixed frame drifts EAST, slowly (n - w_earth = +5 deg/day).  Seen through the library's own rotation (the event frame of
until_nth_event evaluates rotation_dcm, the function the gravity field, the drag and the tides use): longitude lambda0 + 1 deg is
reached after 1 / 5 day = 4.8 h, longitude lambda0 - 1 deg is not reached within 6 h.  With the spin reversed in rotation_dcm the
longitude would race at n + w = 725 deg/day: both events within minutes."""
import numpy as np

import nyx_amd as nx
from nyx_amd import ephem
from rotation_cases import dcm_from_angles, iau_angles_rad
from scenarios import EPOCH0_NS, two_body_setup

W_EARTH_DEG_PER_DAY = nx.IAU_EARTH_ROTATION.w_deg[1]
DRIFT_DEG_PER_DAY = 5.0
IAU_EARTH = nx.Frame(nx.EARTH, ephem.MU_EARTH, 6378.14, nx.IAU_EARTH_ROTATION, flattening=(6378.14 - 6356.75) / 6378.14)


def sub_synchronous_state():
    """(batch of one, lambda0): circular orbit in the plane of the IAU equator of date, mean motion w_earth + 5 deg/day."""
    n = np.radians(W_EARTH_DEG_PER_DAY + DRIFT_DEG_PER_DAY) / 86400.0
    a = (ephem.MU_EARTH / n ** 2) ** (1.0 / 3.0)
    m = dcm_from_angles(iau_angles_rad(nx.IAU_EARTH_ROTATION, nx.to_seconds(EPOCH0_NS)))   # inertial -> body-fixed at t0
    lam0 = 40.0
    rb = a * np.array([np.cos(np.radians(lam0)), np.sin(np.radians(lam0)), 0.0])
    # inertial velocity of a circular prograde orbit about the body's pole (z of the body-fixed frame)
    pole = m.T @ np.array([0.0, 0.0, 1.0])
    r = m.T @ rb
    v = np.sqrt(ephem.MU_EARTH / a) * np.cross(pole, r) / a
    b = nx._abi.StateBatch(1)
    b.set_rv(np.concatenate([r, v])[None, :])
    b.epoch_ns[:] = EPOCH0_NS
    return b, lam0


def check_spin_sense(until_event):
    """`until_event(compiled, batch, max_duration_ns, event)` -> (out, stats): the oracle's or the device's until_nth_event."""
    prop, almanac, central = two_body_setup(nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions(), ephem.MU_EARTH)
    compiled = prop.compile(almanac, central)
    b, lam0 = sub_synchronous_state()
    six_h = 6 * 3600 * nx.NS_PER_S
    east = nx.Event(nx._abi.EV_LONGITUDE_DEG, lam0 + 1.0, frame=IAU_EARTH)
    out, st = until_event(compiled, b, six_h, east)
    assert st.status[0] == 0
    t_hours = (int(out.epoch_ns[0]) - EPOCH0_NS) / 3.6e12
    assert abs(t_hours - 24.0 / DRIFT_DEG_PER_DAY) < 0.1, t_hours          # 4.8 h: +5 deg/day EASTWARD in the body-fixed frame
    west = nx.Event(nx._abi.EV_LONGITUDE_DEG, lam0 - 1.0, frame=IAU_EARTH)
    out, st = until_event(compiled, b, six_h, west)
    assert st.status[0] == nx._abi.ERR_EVENT_NOT_FOUND                      # never westward
    return t_hours
