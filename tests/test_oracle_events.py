"""Oracle checks of until_nth_event (propagators/event.rs:88-211) against the assertions of the reference's own tests
(tests/propagation/stopcond.rs:29-150): the n-th apsis falls in the right orbit, the event value is met, successive
events are one period apart; plus the bookkeeping nyx-core fixes (the triggering state is not published, NthEventError)."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from nyx_amd import _abi
from scenarios import EPOCH0_NS, GOLDEN, earth_frame

S = nx.NS_PER_S
MU = GOLDEN["mu_gmat"]
STATE = [-2436.45, -2436.45, 6891.037, 5.088611, -5.088611, 0.01]      # stopcond.rs:33-35


def setup(n=1):
    central = earth_frame(MU)
    prop = nx.Propagator.default(nx.SpacecraftDynamics.new(nx.OrbitalDynamics.two_body()))
    compiled = prop.compile(nx.Almanac(), central)
    scs = [nx.Spacecraft(EPOCH0_NS, np.array(STATE) * (1 + 1e-3 * i), central) for i in range(n)]
    return compiled, nx.pack_spacecraft(scs, False)


def period_ns(rv):
    r, v = np.linalg.norm(rv[:3]), np.linalg.norm(rv[3:])
    a = -MU / (2 * (v * v / 2 - MU / r))
    return int(2 * np.pi * np.sqrt(a ** 3 / MU) * 1e9)


def true_anomaly_deg(rv, mu=MU):
    """Independent of the implementation under test: numpy, atan2 form (acos has a 1e-6 deg noise floor at the apsides)."""
    r, v = rv[:3], rv[3:]
    e = ((v @ v - mu / np.linalg.norm(r)) * r - (r @ v) * v) / mu
    h = np.cross(r, v)
    ta = np.degrees(np.arctan2(np.cross(e, r) @ h / np.linalg.norm(h), e @ r))
    return ta + 360.0 if ta < 0 else ta


@pytest.mark.parametrize("event,target", [(nx.Event.apoapsis(), 180.0), (nx.Event.periapsis(), 0.0)])
def test_third_apsis_like_the_reference(event, target):
    compiled, batch = setup()
    p = period_ns(batch.rv()[0])
    out, st, traj, crossings = oracle_lib.propagate_until_event(compiled, batch, 5 * p, event, trigger=3)
    assert st.status[0] == 0 and crossings[0] == 3
    # stopcond.rs:70-86: the third apsis lies in the third orbit
    assert out.epoch_ns[0] - (EPOCH0_NS + 2 * p) >= 1 and out.epoch_ns[0] - (EPOCH0_NS + 3 * p) <= 1
    ta = true_anomaly_deg(out.rv()[0])
    assert min(abs(ta - target), abs(ta - target - 360.0), abs(ta - target + 360.0)) < 1e-6      # stopcond.rs:87-90
    # the trajectory ends with the END state of the triggering step (event.rs:179), the event lies in its last interval
    ep, xs = traj.trajectory(0)
    assert ep[-2] <= out.epoch_ns[0] <= ep[-1] and traj.len[0] == st.n_accepted[0] + 1
    # successive events one period apart (stopcond.rs:56-63: 0.5 s)
    first, st1, _, _ = oracle_lib.propagate_until_event(compiled, batch, 5 * p, event, trigger=1)
    second, st2, _, _ = oracle_lib.propagate_until_event(compiled, batch, 5 * p, event, trigger=2)
    assert abs((second.epoch_ns[0] - first.epoch_ns[0]) - p) < 0.5 * S and abs((out.epoch_ns[0] - second.epoch_ns[0]) - p) < 0.5 * S


def test_not_found_is_reported_with_the_crossings_seen():
    compiled, batch = setup()
    p = period_ns(batch.rv()[0])
    out, st, traj, crossings = oracle_lib.propagate_until_event(compiled, batch, int(1.5 * p), nx.Event.apoapsis(), trigger=3)
    assert st.status[0] == _abi.ERR_EVENT_NOT_FOUND and crossings[0] in (1, 2)      # NthEventError{nth: 3, found: < 3}
    plain, _ = oracle_lib.propagate(compiled, batch, int(1.5 * p))
    np.testing.assert_array_equal(out.rv(), plain.rv())                             # the propagation itself ran to max_duration
    assert out.epoch_ns[0] == EPOCH0_NS + int(1.5 * p)


def test_non_angle_scalars_and_batches():
    compiled, batch = setup(n=3)
    p = period_ns(batch.rv()[0])
    rmag0 = np.linalg.norm(batch.rv()[:, :3], axis=1)
    target = float(rmag0.max() + 2.0)
    ev = nx.Event(_abi.EV_RMAG_KM, target)
    out, st, traj, crossings = oracle_lib.propagate_until_event(compiled, batch, 2 * p, ev, trigger=2)
    ok = st.status == 0
    assert ok.any() and (crossings[ok] == 2).all() and (st.status[~ok] == _abi.ERR_EVENT_NOT_FOUND).all()
    assert np.abs(np.linalg.norm(out.rv()[ok, :3], axis=1) - target).max() < 1e-6
    # equator crossings.  The interpolant is evaluated at f64 seconds past J2000 (0.12 us grid): with dz/dt ~ 7 km/s the
    # event value moves in steps of ~1e-6 km, so that is the precision that can be asked for (a tighter one ends as
    # "not found in the bracket", status ERR_EVENT_SEARCH)
    z = nx.Event(_abi.EV_Z_KM, 0.0, value_precision=1e-5)
    out, st, _, _ = oracle_lib.propagate_until_event(compiled, batch, 2 * p, z, trigger=1)
    assert (st.status == 0).all() and np.abs(out.rv()[:, 2]).max() < 1e-5
    tight = nx.Event(_abi.EV_Z_KM, 0.0, value_precision=1e-12)
    _, st_t, _, _ = oracle_lib.propagate_until_event(compiled, batch, 2 * p, tight, trigger=1)
    assert (st_t.status == _abi.ERR_EVENT_SEARCH).any()
    sma = nx.Event(_abi.EV_SMA_KM, 1.0)                                              # never crossed: constant of the motion
    out, st, _, crossings = oracle_lib.propagate_until_event(compiled, batch, p // 2, sma)
    assert (st.status == _abi.ERR_EVENT_NOT_FOUND).all() and (crossings == 0).all()


def test_less_than_and_greater_than_stop_where_equals_stops():
    """Condition::LessThan(v) / GreaterThan(v): until_nth_event's closure counts sign changes of the event's value in either direction
    (event.rs:124-141), so on a non-angle scalar they are `Equals(v)` to the stop condition - which is what the mirror builds."""
    import pytest
    compiled, batch = setup(n=2)
    p = period_ns(batch.rv()[0])
    target = float(np.linalg.norm(batch.rv()[:, :3], axis=1).max() + 2.0)
    ref = oracle_lib.propagate_until_event(compiled, batch, 2 * p, nx.Event(_abi.EV_RMAG_KM, target), trigger=2)
    for make in (nx.Event.less_than, nx.Event.greater_than):
        got = oracle_lib.propagate_until_event(compiled, batch, 2 * p, make(_abi.EV_RMAG_KM, target), trigger=2)
        np.testing.assert_array_equal(got[0].rv(), ref[0].rv())
        np.testing.assert_array_equal(got[0].epoch_ns, ref[0].epoch_ns)
        np.testing.assert_array_equal(got[3], ref[3])
    with pytest.raises(NotImplementedError):
        nx.Event.less_than(_abi.EV_TRUE_ANOMALY_DEG, 10.0)
    with pytest.raises(NotImplementedError):
        nx.Event.between(_abi.EV_RMAG_KM, 7000.0, 7100.0)


# ---- geometric scalars and the observer frame (tests/propagation/stopcond.rs:252-312) -------------------------------------
EPOCH_2008_02_29_NOON_UTC_NS = (2981 * 86400) * nx.NS_PER_S + 65_184_000_000   # ET past J2000 (TT - UTC = 65.184 s in 2008)
IAU_EARTH_SHAPED = nx.Frame(nx.EARTH, 398600.435436096, 6378.14, nx.IAU_EARTH_ROTATION, flattening=(6378.14 - 6356.75) / 6378.14)  # pck08 radii


def _stopcond_state(n=1):
    b = nx._abi.StateBatch(n)
    b.set_rv(np.tile(np.array([-2436.45, -2436.45, 6891.037, 5.088611, -5.088611, 0.0]), (n, 1)))
    b.epoch_ns[:] = EPOCH_2008_02_29_NOON_UTC_NS
    return b


def test_line_of_nodes_longitude_event():
    # stopcond.rs:252-281: Event(Longitude, Equals(0)), two-body, RK89 default, 3 periods, no event frame
    from scenarios import two_body_setup
    prop, almanac, central = two_body_setup(nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions(), 398600.435436096)
    compiled = prop.compile(almanac, central)
    rv0 = _stopcond_state().rv()[0]
    period = 2 * np.pi * np.sqrt(_sma(rv0, central.mu_km3_s2) ** 3 / central.mu_km3_s2)
    ev = nx.Event(nx._abi.EV_LONGITUDE_DEG, 0.0)
    out, st, traj, cr = oracle_lib.propagate_until_event(compiled, _stopcond_state(), int(3 * period * 1e9), ev, trigger=1, capacity=1024)
    assert st.status[0] == 0
    lon = np.degrees(np.arctan2(out.rv()[0, 1], out.rv()[0, 0]))
    assert abs(lon) < 1e-3                                    # the reference's assertion
    assert abs(lon) < 1e-6                                    # ... and what the search actually reaches


def test_geodetic_latitude_event_in_iau_earth():
    # stopcond.rs:283-312: Event(Latitude, Equals(2.0)) with event_frame = IAU_EARTH, DP78 default, 3 periods
    from scenarios import two_body_setup
    prop, almanac, central = two_body_setup(nx.IntegratorMethod.DormandPrince78, nx.IntegratorOptions(), 398600.435436096)
    compiled = prop.compile(almanac, central)
    rv0 = _stopcond_state().rv()[0]
    period = 2 * np.pi * np.sqrt(_sma(rv0, central.mu_km3_s2) ** 3 / central.mu_km3_s2)
    ev = nx.Event(nx._abi.EV_LATITUDE_DEG, 2.0, frame=IAU_EARTH_SHAPED)
    out, st, traj, cr = oracle_lib.propagate_until_event(compiled, _stopcond_state(), int(3 * period * 1e9), ev, trigger=1, capacity=1024)
    assert st.status[0] == 0
    # independent evaluation of the geodetic latitude of the returned state in the rotating frame (closed-form Bowring check)
    from rotation_cases import dcm_from_angles, iau_angles_rad
    m = dcm_from_angles(iau_angles_rad(nx.IAU_EARTH_ROTATION, nx.to_seconds(int(out.epoch_ns[0]))))
    rb = m @ out.rv()[0, :3]
    a, f = IAU_EARTH_SHAPED.mean_equatorial_radius_km, IAU_EARTH_SHAPED.flattening
    e2 = f * (2 - f)
    p = np.hypot(rb[0], rb[1])
    lat = np.arctan2(rb[2], p * (1 - e2))
    for _ in range(10):   # fixed-point iteration on the geodetic latitude
        n = a / np.sqrt(1 - e2 * np.sin(lat) ** 2)
        h = p / np.cos(lat) - n
        lat = np.arctan2(rb[2], p * (1 - e2 * n / (n + h)))
    assert abs(2.0 - np.degrees(lat)) < 1e-3                  # the reference's assertion
    assert abs(2.0 - np.degrees(lat)) < 1e-6
    # geodetic, not geocentric: the declination of the same state differs by the ellipsoid's ~0.01 deg at this latitude
    decl = np.degrees(np.arcsin(rb[2] / np.linalg.norm(rb)))
    assert 1e-3 < abs(decl - 2.0) < 0.05


def _sma(rv, mu):
    r, v = np.linalg.norm(rv[:3]), np.linalg.norm(rv[3:])
    return -mu / (2 * (v * v / 2 - mu / r))
