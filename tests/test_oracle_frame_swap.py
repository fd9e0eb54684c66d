"""opts.integration_frame in the oracle (instance.rs:117-142, 211-220): the swap is the translation by the ephemeris state of the
states' centre, in at the start epoch and out at the final one - checked against doing exactly that by hand with numpy's
Chebyshev derivative, and against the reference's own statement that the frame of the result is the frame of the input."""
import dataclasses

import numpy as np

import nyx_amd as nx
import oracle_lib
from nyx_amd import ephem
from frame_swap_cases import MOON_FRAME, chain_state_numpy, moon_batch, setup


def test_swap_equals_translating_by_hand():
    prop, almanac, earth = setup()
    b = moon_batch(6, seed=1)
    dur = 45 * 60 * nx.NS_PER_S
    swapped = prop.compile(almanac, earth, state_frame=MOON_FRAME)
    assert swapped.cfg.state_frame_body > 0
    out, st = oracle_lib.propagate(swapped, b, dur)
    assert (st.status == 0).all() and (out.epoch_ns == b.epoch_ns + dur).all()
    # by hand: into the Earth frame with numpy, the plain Earth-frame propagation, back with numpy
    plain = prop.compile(almanac, earth)
    assert plain.cfg.state_frame_body == 0
    e = b.copy()
    rv = b.rv().copy()
    for i in range(b.n):
        r, v = chain_state_numpy(almanac, nx.MOON, int(b.epoch_ns[i]))
        rv[i, :3] += r
        rv[i, 3:] += v
    e.set_rv(rv)
    eo, est = oracle_lib.propagate(plain, e, dur)
    assert (est.status == 0).all()
    back = eo.rv().copy()
    for i in range(b.n):
        r, v = chain_state_numpy(almanac, nx.MOON, int(eo.epoch_ns[i]))
        back[i, :3] -= r
        back[i, 3:] -= v
    d = out.rv() - back
    # two evaluations of the same series (the recurrence here, numpy's chebder there) differ by ~1e-11 km/s in the Moon's 1 km/s,
    # and 45 minutes carry that into the position: 0.1 mm is the noise floor of this comparison, a wrong derivative is off by km/s
    assert np.abs(d[:, :3]).max() < 2e-7 and np.abs(d[:, 3:]).max() < 2e-10
    # and the orbit stayed a lunar orbit (the result IS Moon-centred)
    assert np.all(np.linalg.norm(out.rv()[:, :3], axis=1) < 2100.0)


def test_mirror_resolves_the_frames():
    prop, almanac, earth = setup()
    sc = nx.Spacecraft(0, np.array([1900.0, 0, 0, 0, 1.6, 0]), MOON_FRAME)
    prop.opts = dataclasses.replace(prop.opts, integration_frame=earth)
    c = prop.compile(almanac, prop.opts.integration_frame, state_frame=sc.frame)
    assert c.cfg.state_frame_body > 0 and c.cfg.central_mu_km3_s2 == ephem.MU_EARTH
    # an integration frame equal to the state's frame is no swap (instance.rs:118-119)
    prop.opts = dataclasses.replace(prop.opts, integration_frame=MOON_FRAME)
    assert prop.compile(almanac, MOON_FRAME, state_frame=MOON_FRAME).cfg.state_frame_body == 0


def test_trajectory_of_a_swapped_run_is_the_references_mixture():
    """instance.rs:297-326 around :117-142 / :211-220: `start_state` is taken before propagate() translates the state, the channel is
    fed inside the loop and only the returned state is translated back - the Traj's first state is in the caller's frame, every
    other one in the integration frame.  The oracle restates that as it is."""
    prop, almanac, earth = setup()
    b = moon_batch(4, seed=2)
    dur = 30 * 60 * nx.NS_PER_S
    swapped = prop.compile(almanac, earth, state_frame=MOON_FRAME)
    out, st, traj = oracle_lib.propagate_with_traj(swapped, b, dur, 256)
    plain_out, plain_st = oracle_lib.propagate(swapped, b, dur)
    assert (st.status == 0).all()
    np.testing.assert_array_equal(out.rv(), plain_out.rv())            # recording changes nothing
    np.testing.assert_array_equal(st.n_accepted, plain_st.n_accepted)
    assert (traj.len == st.n_accepted + 1).all()
    for i in range(b.n):
        ep, rv = traj.trajectory(i)
        np.testing.assert_array_equal(rv[0], b.rv()[i])                # the start state: Moon-centred, untouched
        assert ep[0] == b.epoch_ns[i] and ep[-1] == out.epoch_ns[i]
        assert np.all(np.linalg.norm(rv[1:, :3], axis=1) > 3.0e5)      # every published state: Earth-centred
        r, v = chain_state_numpy(almanac, nx.MOON, int(ep[-1]))        # the last one, translated back by hand, is the returned state
        assert np.abs(rv[-1, :3] - r - out.rv()[i, :3]).max() < 1e-6 and np.abs(rv[-1, 3:] - v - out.rv()[i, 3:]).max() < 1e-9


def test_covariance_map_of_a_swapped_run():
    """od/process/mod.rs:453-468: every segment is one `until_epoch`, i.e. one translation in and one back.  A translation does not
    touch the STM: the mapped covariance of the Moon-centred formulation equals the one of the same states handed in Earth-centred."""
    from scenarios import EPOCH0_NS
    prop, almanac, earth = setup()
    b = moon_batch(3, seed=5)
    b.epoch_ns[:] = EPOCH0_NS
    b.stm = np.zeros((b.n, 81))
    b.reset_stm()
    rng = np.random.default_rng(0)
    p0 = np.zeros((b.n, 9, 9))
    for i in range(b.n):
        a = rng.standard_normal((9, 9)) * np.array([1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3, 1e-2, 0.0, 0.0])[:, None]
        p0[i] = a @ a.T
    end = EPOCH0_NS + 5 * 60 * nx.NS_PER_S
    swapped = prop.compile(almanac, earth, stm=True, state_frame=MOON_FRAME)
    plain = prop.compile(almanac, earth, stm=True)
    got = oracle_lib.predict_until(swapped, b, p0, end, 60 * nx.NS_PER_S, history=5)
    e = b.copy()
    rv = b.rv().copy()
    for i in range(b.n):
        r, v = chain_state_numpy(almanac, nx.MOON, int(b.epoch_ns[i]))
        rv[i, :3] += r
        rv[i, 3:] += v
    e.set_rv(rv)
    ref = oracle_lib.predict_until(plain, e, p0, end, 60 * nx.NS_PER_S, history=5)
    assert (got.stats.status == 0).all() and (got.n_updates == 5).all() and (ref.n_updates == 5).all()
    assert np.all(np.linalg.norm(got.states.rv()[:, :3], axis=1) < 2200.0)       # handed back Moon-centred
    scale = np.abs(ref.covar).max(axis=(-2, -1), keepdims=True)
    assert (np.abs(got.covar - ref.covar) / scale).max() < 1e-7                  # (positions differ by the 0.1 mm of the hand translation)
