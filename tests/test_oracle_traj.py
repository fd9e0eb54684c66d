"""Oracle checks of the Traj evaluation (md/trajectory/traj.rs:82-162, interpolatable.rs:52-108): HRMINT's published
known answer, independent Hermite interpolation (scipy Krogh), the reference's window rule and the properties the
reference's own tests assert (tests/propagation/trajectory.rs:84-135)."""
import numpy as np
import pytest
from scipy.interpolate import KroghInterpolator

import nyx_amd as nx
import oracle_lib
from nyx_amd import _abi
from scenarios import EPOCH0_NS, GOLDEN, earth_frame

LEO = [-2436.45, -2436.45, 6891.037, 5.088611, -5.088611, 0.0]


def two_body_traj(duration_s, n=1, capacity=4096, scale=None):
    central = earth_frame(GOLDEN["mu_gmat"])
    prop = nx.Propagator.default(nx.SpacecraftDynamics.new(nx.OrbitalDynamics.two_body()))
    compiled = prop.compile(nx.Almanac(), central)
    scs = []
    for i in range(n):
        rv = np.array(LEO) * (1.0 + (0.002 * i if scale is None else scale[i]))
        scs.append(nx.Spacecraft(EPOCH0_NS, rv, central))
    batch = nx.pack_spacecraft(scs, False)
    out, st, traj = oracle_lib.propagate_with_traj(compiled, batch, int(duration_s * 1e9), capacity)
    return compiled, batch, out, traj


def window(length, idx):
    """traj.rs:104-115 restated independently for the test."""
    first = max(idx - 6, 0)
    last = min(length, first + 13)
    if last == length:
        first = max(last - 12, 0)
    return first, last


def test_hrmint_documented_example():
    # NAIF HRMINT "Examples": f(x) = x^7 + 2x^2 + 5 through four points, evaluated at x = 2
    st, f, df = oracle_lib.hermite_eval([-1.0, 0.0, 3.0, 5.0], [6.0, 5.0, 2210.0, 78180.0], [3.0, 0.0, 5115.0, 109395.0], 2.0)
    assert st == 0 and f == 141.0 and df == 456.0


def test_hermite_reproduces_polynomials_and_matches_krogh():
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 12, 13):
        xs = np.sort(rng.uniform(-1.0, 1.0, n)) + np.arange(n) * 0.3
        coef = rng.standard_normal(2 * n)  # degree 2n-1: reproduced exactly in exact arithmetic
        p = np.polynomial.Polynomial(coef)
        x = rng.uniform(xs[0], xs[-1]) if n > 1 else xs[0] + 0.1
        st, f, df = oracle_lib.hermite_eval(xs, p(xs), p.deriv()(xs), x)
        scale = np.abs(coef).sum() * max(1.0, np.abs(xs).max()) ** (2 * n - 1)
        assert st == 0
        assert abs(f - p(x)) < 1e-9 * scale and abs(df - p.deriv()(x)) < 1e-8 * scale
        # independent algorithm on arbitrary data
        ys, yd = rng.standard_normal(n), rng.standard_normal(n)
        k = KroghInterpolator(np.repeat(xs, 2), np.column_stack([ys, yd]).ravel())
        st, f, df = oracle_lib.hermite_eval(xs, ys, yd, x)
        ref_f, ref_df = k(x), k.derivative(x)
        assert abs(f - ref_f) <= 1e-7 * max(1.0, abs(ref_f)) and abs(df - ref_df) <= 1e-6 * max(1.0, abs(ref_df))


def test_hermite_rejects_coincident_abscissas():
    st, _, _ = oracle_lib.hermite_eval([0.0, 1.0, 1.0], [0.0, 1.0, 1.0], [0.0, 0.0, 0.0], 0.5)
    assert st == _abi.INTERP_MATH


def test_stored_epochs_come_back_exactly_and_bounds_fail():
    # trajectory.rs:84-135: stored states are returned as they are (error == 0.0); one ns past the end is an error
    _, batch, out, traj = two_body_traj(6 * 3600.0)
    ep, xs = traj.trajectory(0)
    assert len(ep) > 100 and ep[0] == EPOCH0_NS and ep[-1] == EPOCH0_NS + 6 * 3600 * 10**9
    got, status = oracle_lib.traj_at(traj, ep)
    assert (status == 0).all()
    np.testing.assert_array_equal(got[:, 0, :], xs)
    np.testing.assert_array_equal(got[-1, 0, :], out.rv()[0])
    _, status = oracle_lib.traj_at(traj, [ep[-1] + 1, ep[0] - 1])
    assert (status == _abi.INTERP_NO_DATA).all()


def test_interpolation_error_against_repropagation():
    # independent truth: propagate to the query epochs themselves
    compiled, batch, _, traj = two_body_traj(3 * 3600.0)
    ep, _ = traj.trajectory(0)
    rng = np.random.default_rng(0)
    queries = np.sort(rng.integers(ep[0] + 1, ep[-1], size=40))
    got, status = oracle_lib.traj_at(traj, queries)
    assert (status == 0).all()
    for q, e in enumerate(queries):
        truth, _ = oracle_lib.propagate(compiled, batch, int(e - EPOCH0_NS))
        idx = int(np.searchsorted(ep, e))
        # interior: bounded by the 0.12 us granularity of f64 seconds past J2000 (x 7.6 km/s ~ 1 mm, the reference's own
        # gate, trajectory.rs:405-413); the one-sided windows of the first / last interval are worse (Runge)
        edge = idx <= 1 or idx >= len(ep) - 1
        assert np.linalg.norm(got[q, 0, :3] - truth.rv()[0, :3]) < (1e-4 if edge else 1e-6)
        assert np.linalg.norm(got[q, 0, 3:] - truth.rv()[0, 3:]) < (1e-5 if edge else 2e-8)


def test_window_rule_on_rough_data():
    # data that no polynomial fits: the result depends on exactly which stored states enter the window
    n, length = 3, 40
    rng = np.random.default_rng(5)
    traj = _abi.TrajBatch(n, 64)
    traj.len[:] = [length, 9, 13]
    base = EPOCH0_NS + np.cumsum(rng.integers(20, 90, size=64)) * 10**9
    for i in range(n):
        traj.epoch_ns[:, i] = base + i * 7
    traj.state[:] = rng.standard_normal(traj.state.shape)
    for i in range(n):
        L = int(traj.len[i])
        for idx in (1, 2, 6, 7, L // 2, L - 7, L - 6, L - 2, L - 1):
            if not 1 <= idx < L:
                continue
            e = int(traj.epoch_ns[idx - 1, i] + (traj.epoch_ns[idx, i] - traj.epoch_ns[idx - 1, i]) // 3)
            got, status = oracle_lib.traj_at(traj, [e])
            assert status[0, i] == 0
            a, b = window(L, idx)
            assert b - a == (12 if b == L and L >= 12 else min(13, L))
            xs = np.array([oracle_lib.load().nyx_oracle_ns_to_seconds(int(v)) for v in traj.epoch_ns[a:b, i]])
            x = oracle_lib.load().nyx_oracle_ns_to_seconds(e)
            for c in range(3):
                k = KroghInterpolator(np.repeat(xs - xs[0], 2), np.column_stack([traj.state[c, a:b, i], traj.state[c + 3, a:b, i]]).ravel())
                ref_f, ref_df = k(x - xs[0]), k.derivative(x - xs[0])
                assert abs(got[0, i, c] - ref_f) <= 1e-6 * max(1.0, abs(ref_f)), (i, idx, c)
                assert abs(got[0, i, c + 3] - ref_df) <= 1e-6 * max(1.0, abs(ref_df)), (i, idx, c)


def test_every_is_an_inclusive_time_series():
    _, _, _, traj = two_body_traj(3600.0, n=2)
    step = 60 * 10**9
    out = oracle_lib.traj_every(traj, step, 128)
    assert list(out.len) == [61, 61]                      # 0, 60, ..., 3600 s inclusive
    np.testing.assert_array_equal(out.epoch_ns[:61, 0], EPOCH0_NS + np.arange(61) * step)
    ep, xs = traj.trajectory(1)
    np.testing.assert_array_equal(out.state[:, 0, 1], xs[0])     # first sample = stored start state
    np.testing.assert_array_equal(out.state[:, 60, 1], xs[-1])   # last sample = stored end state
    short = oracle_lib.traj_every(traj, step, 10)
    assert list(short.len) == [61, 61]                    # produced, not stored
    np.testing.assert_array_equal(short.state[:, :10], out.state[:, :10])
    odd = oracle_lib.traj_every(traj, 7 * 60 * 10**9 + 1, 128)
    assert list(odd.len) == [9, 9]                        # floor(3600 / 420.000000001) + 1


def test_back_propagated_trajectory_reads_sorted():
    compiled, batch, out, fwd = two_body_traj(1800.0)
    # run it backwards from the end state: the same orbit stored in decreasing epochs
    back_out, _, back = oracle_lib.propagate_with_traj(compiled, out, -1800 * 10**9, 4096)
    ep, _ = back.trajectory(0)
    assert ep[0] > ep[-1]
    queries = EPOCH0_NS + np.array([1, 450, 900, 1799]) * 10**9
    a, sa = oracle_lib.traj_at(fwd, queries)
    b, sb = oracle_lib.traj_at(back, queries)
    assert (sa == 0).all() and (sb == 0).all()
    assert np.abs(a[:, 0, :3] - b[:, 0, :3]).max() < 5e-6  # two interpolants, each at the ~1 mm abscissa granularity
    ev = oracle_lib.traj_every(back, 600 * 10**9, 16)
    assert list(ev.epoch_ns[:4, 0]) == list(EPOCH0_NS + np.arange(4) * 600 * 10**9) and ev.len[0] == 4
