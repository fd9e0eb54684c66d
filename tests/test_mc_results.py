"""`Results` of a Monte Carlo run with trajectories (reference mc/results.rs:60-245, montecarlo.rs:208-273): the
reports `every_value_of`, `every_value_of_between`, `first_values_of`, `last_values_of`, `dispersion_values_of`.
No GPU here: propagation and trajectory evaluation are injected (the oracle); what is under test is the host logic -
report layout and order, failed runs, the template-minus-state sign of `actual_dispersions`, the clamping of
`every_between` - and `Spacecraft::value` for arrays (nyx_amd/params.py)."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from nyx_amd import ephem
from nyx_amd.params import StateParameter as P
from scenarios import EPOCH0_NS, leo_full_setup, leo_nominal


class OracleTraj:
    """What Results needs from a context: traj_at / traj_every (GpuContext's signatures)."""

    traj_at = staticmethod(oracle_lib.traj_at)
    traj_every = staticmethod(oracle_lib.traj_every)


def _mc(fail_index=None, sigmas=(1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3)):
    prop, almanac, central = leo_full_setup(degree=4)
    compiled = prop.compile(almanac, central)
    template = nx.Spacecraft(EPOCH0_NS, leo_nominal(), central, dry_mass_kg=100.0, prop_mass_kg=10.0, srp_area_m2=1.0, cr=1.8)
    mvn = nx.MvnSpacecraft.from_sigmas(template, list(sigmas))

    def fn(batch, end_epoch_ns):
        out, st, traj = oracle_lib.propagate_with_traj(compiled, batch, end_epoch_ns - int(batch.epoch_ns[0]), 256)
        if fail_index is not None:
            st.status[fail_index] = nx._abi.ERR_NAN
        return out, st, traj, OracleTraj

    return prop, almanac, compiled, nx.MonteCarlo(mvn, seed=3, propagate_fn=fn)


def test_reports_follow_the_reference_layout():
    prop, almanac, compiled, mc = _mc()
    end = EPOCH0_NS + 1800 * nx.NS_PER_S
    res = mc.run_until_epoch(prop, almanac, end, 5)
    assert [r.index for r in res.runs] == [0, 1, 2, 3, 4]
    assert all(isinstance(r.result, nx.PropResult) and r.result.traj is not None for r in res.runs)
    # PropResult { state, traj }: the state is the end of the trajectory
    for r in res.runs:
        ep, xs = r.result.traj
        assert ep[0] == EPOCH0_NS and ep[-1] == end == r.result.state.epoch_ns
        np.testing.assert_array_equal(xs[-1], r.result.state.rv)
        np.testing.assert_array_equal(xs[0], r.dispersed_state.state.rv)
    step = 300 * nx.NS_PER_S
    xs = res.every_value_of(P.X, step)
    assert len(xs) == 5 * 7                                   # 0, 300, ..., 1800 s of every run, run after run
    for k, r in enumerate(res.runs):
        _, want = r.result.traj.every(step)
        np.testing.assert_array_equal(xs[7 * k: 7 * k + 7], want[:, 0])
    np.testing.assert_array_equal(res.first_values_of(P.VY), [r.dispersed_state.rv[4] for r in res.runs])
    np.testing.assert_array_equal(res.last_values_of(P.Z), [r.result.state.rv[2] for r in res.runs])
    assert res.last_values_of(P.Cr) == [1.8] * 5 and res.first_values_of(P.TotalMass) == [110.0] * 5
    # between: clamped to the trajectory (traj.rs:153-162)
    btw = res.every_value_of_between(P.Rmag, step, EPOCH0_NS - 10 * step, EPOCH0_NS + 2 * step)
    assert len(btw) == 5 * 3
    np.testing.assert_allclose(btw[0], np.linalg.norm(res.runs[0].dispersed_state.rv[:3]), rtol=1e-15)
    assert res.every_value_of_between(P.X, step, end + step, end + 3 * step) == []
    assert res.every_value_of(P.Isp, step) == []              # unavailable and no substitute: skipped (a warning in the reference) ...
    assert res.every_value_of(P.Isp, step, value_if_run_failed=-1.0) == [-1.0] * 35   # ... or one substitute per state


def test_failed_runs_and_dispersions():
    prop, almanac, compiled, mc = _mc(fail_index=2)
    res = mc.run_until_epoch(prop, almanac, EPOCH0_NS + 600 * nx.NS_PER_S, 4)
    assert isinstance(res.runs[2].result, nx.PropagationError) and len(res.ok_runs()) == 3
    step = 300 * nx.NS_PER_S
    assert len(res.every_value_of(P.Y, step)) == 3 * 3                                   # the failed run is skipped ...
    got = res.every_value_of(P.Y, step, value_if_run_failed=0.0)
    assert len(got) == 3 * 3 + 1 and got[6] == 0.0                                        # ... or stands for ONE value
    assert len(res.last_values_of(P.X, value_if_run_failed=np.nan)) == 4
    # actual_dispersions: template.value(param) - state.value(param) (multivariate.rs:320-325), only what is dispersed
    t = mc.random_state.template
    d = res.dispersion_values_of(P.X)
    np.testing.assert_allclose(d, [t.rv[0] - r.dispersed_state.rv[0] for r in res.runs], rtol=0, atol=0)
    # a generator built from a covariance lists all nine components (multivariate.rs:274-312)
    assert [p for p, _ in res.runs[0].dispersed_state.actual_dispersions] == [P.X, P.Y, P.Z, P.VX, P.VY, P.VZ, P.Cr, P.Cd, P.PropMass]
    assert res.dispersion_values_of(P.Cr) == [0.0] * 4
    with pytest.raises(nx.StateError):
        res.dispersion_values_of(P.SemiMajorAxis)


def test_resume_skips_and_without_traj():
    prop, almanac, compiled, mc = _mc()
    end = EPOCH0_NS + 600 * nx.NS_PER_S
    full = mc.run_until_epoch(prop, almanac, end, 6)
    tail = mc.resume_run_until_epoch(prop, almanac, 4, end, 2)
    assert [r.index for r in tail.runs] == [0, 1]            # enumerate() comes after skip() in the reference: indices restart
    np.testing.assert_array_equal(tail.last_values_of(P.X), full.last_values_of(P.X)[4:])
    np.testing.assert_array_equal(tail.every_value_of(P.VZ, 200 * nx.NS_PER_S), full.every_value_of(P.VZ, 200 * nx.NS_PER_S)[4 * 4:])


def test_state_values_of_a_known_orbit():
    """Elements of a hand-built state (values from the definitions, not from ANISE: parity unpinned)."""
    mu = ephem.MU_EARTH
    a, e, inc, raan, aop, ta = 7000.0, 0.01, np.radians(51.6), np.radians(40.0), np.radians(30.0), np.radians(75.0)
    p = a * (1 - e * e)
    r = p / (1 + e * np.cos(ta))
    rp = np.array([r * np.cos(ta), r * np.sin(ta), 0.0])
    vp = np.sqrt(mu / p) * np.array([-np.sin(ta), e + np.cos(ta), 0.0])

    def rot(axis, ang):
        c, s = np.cos(ang), np.sin(ang)
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) if axis == 3 else np.array([[1, 0, 0], [0, c, -s], [0, s, c]])

    m = rot(3, raan) @ rot(1, inc) @ rot(3, aop)
    rv = np.concatenate([m @ rp, m @ vp])
    want = {P.SemiMajorAxis: a, P.Eccentricity: e, P.Inclination: 51.6, P.RAAN: 40.0, P.AoP: 30.0, P.TrueAnomaly: 75.0,
            P.Rmag: r, P.ApoapsisRadius: a * (1 + e), P.PeriapsisRadius: a * (1 - e), P.Period: 2 * np.pi * np.sqrt(a ** 3 / mu),
            P.Energy: -mu / (2 * a), P.Hmag: np.sqrt(mu * p)}
    for param, val in want.items():
        got = nx.state_value(param, np.stack([rv, rv]), mu)
        assert got.shape == (2,)
        np.testing.assert_allclose(got, val, rtol=2e-12, err_msg=param.name)
    assert nx.state_value(P.VX, rv, mu) == rv[3]
    np.testing.assert_array_equal(nx.state_value(P.Cd, np.zeros((3, 2, 6)), mu, cd=2.2), np.full((3, 2), 2.2))


def test_traj_views_follow_the_reference():
    """Traj::every_between / filter_by_epoch / filter_by_offset / resample / rebuild (md/trajectory/traj.rs:148-193, 367-407)."""
    prop, almanac, compiled, mc = _mc()
    end = EPOCH0_NS + 1800 * nx.NS_PER_S
    traj = mc.run_until_epoch(prop, almanac, end, 1).runs[0].result.traj
    assert traj.start_epoch() == EPOCH0_NS and traj.end_epoch() == end
    s = nx.NS_PER_S
    ep, xs = traj.every_between(300 * s, EPOCH0_NS - 1000 * s, EPOCH0_NS + 700 * s)      # clamped at the start: 0, 300, 600
    assert list(ep - EPOCH0_NS) == [0, 300 * s, 600 * s] and xs.shape == (3, 6)
    np.testing.assert_array_equal(xs[0], traj.first())
    assert traj.every_between(300 * s, end + s, end + 10 * s)[0].size == 0
    # resample = every(step) as a trajectory; evaluating it at its own nodes gives them back
    rs = traj.resample(450 * s)
    assert list(rs.epochs_ns - EPOCH0_NS) == [0, 450 * s, 900 * s, 1350 * s, 1800 * s]
    np.testing.assert_array_equal(rs.states, traj.every(450 * s)[1])
    np.testing.assert_allclose(rs.at(EPOCH0_NS + 900 * s), rs.states[2], rtol=0, atol=1e-9)
    # rebuild: at() for the given epochs, sorted by finalize(); an epoch outside is the error
    rb = traj.rebuild([EPOCH0_NS + 1000 * s, EPOCH0_NS + 10 * s])
    assert list(rb.epochs_ns - EPOCH0_NS) == [10 * s, 1000 * s]
    np.testing.assert_array_equal(rb.states[1], traj.at(EPOCH0_NS + 1000 * s))
    with pytest.raises(nx.TrajError):
        traj.rebuild([end + s])
    # filters keep STORED states
    mid = traj.filter_by_epoch(EPOCH0_NS + 200 * s, EPOCH0_NS + 1500 * s)
    assert len(mid) == int(((traj.epochs_ns >= EPOCH0_NS + 200 * s) & (traj.epochs_ns <= EPOCH0_NS + 1500 * s)).sum()) > 0
    assert len(traj.filter_by_epoch(None, end, end_inclusive=False)) == len(traj) - 1
    off = traj.filter_by_offset(200 * s, 1500 * s)
    np.testing.assert_array_equal(off.epochs_ns, mid.epochs_ns)
    assert len(traj.filter_by_offset()) == len(traj)
