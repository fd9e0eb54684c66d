"""Oracle checks of the covariance-mapping loop (od/process/mod.rs:440-486, od/kalman/filtering.rs:59-99,
od/snc.rs:165-283): the properties the reference's own tests assert (tests/orbit_determination/two_body.rs:1118-1188,
predict_validation.rs:109-116) and the algebra of the time update against numpy."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from nyx_amd import ephem
from scenarios import EPOCH0_NS, earth_frame, keplerian_to_cartesian

S = nx.NS_PER_S


def two_body_rk4(step_s=10.0, n=2):
    central = earth_frame(ephem.MU_EARTH)
    prop = nx.Propagator(nx.SpacecraftDynamics.new(nx.OrbitalDynamics.two_body()), nx.IntegratorMethod.RungeKutta4,
                         nx.IntegratorOptions.with_fixed_step_s(step_s))
    compiled = prop.compile(nx.Almanac(), central, stm=True)
    rv = keplerian_to_cartesian(22000.0, 0.01, 30.0, 80.0, 40.0, 0.0, ephem.MU_EARTH)   # two_body.rs:1130
    scs = [nx.Spacecraft(EPOCH0_NS, rv * (1 + 1e-3 * i), central) for i in range(n)]
    return compiled, nx.pack_spacecraft(scs, True)


INIT_COVAR = np.diag([1e-3] * 3 + [1e-6] * 3 + [0.0] * 3)      # two_body.rs:1139-1151


def test_covariance_inflates_like_the_reference_test():
    compiled, batch = two_body_rk4()
    dur = 2 * 86400 * S                                          # two_body.rs:1123: 2 days, 10 s RK4 steps, 30 s updates
    res = oracle_lib.predict_until(compiled, batch, np.tile(INIT_COVAR, (2, 1, 1)), EPOCH0_NS + dur, 30 * S, deviation_tracking=True)
    assert (res.stats.status == 0).all() and (res.n_updates == dur // (30 * S)).all()
    for i in range(2):
        d = np.diag(res.covar[i])
        assert (d[:6] >= 0).all() and (d[:3] > 1e-3).all() and (d[3:6] > 1e-6).all()      # two_body.rs:1167-1187
        assert np.allclose(res.covar[i], res.covar[i].T, rtol=1e-9, atol=1e-18)
    assert (res.states.epoch_ns == EPOCH0_NS + dur).all()
    # the instance's STM is reset after the last update (mod.rs:479)
    np.testing.assert_array_equal(res.states.stm, np.tile(np.eye(9).ravel(), (2, 1)))


def test_predicted_nominal_state_is_the_plain_propagation():
    # predict_validation.rs:109-116: RIC difference below f64::EPSILON; with a fixed step dividing max_step the two
    # step sequences are identical, so the states are bit-identical
    compiled, batch = two_body_rk4()
    dur = 3600 * S
    res = oracle_lib.predict_until(compiled, batch, np.tile(INIT_COVAR, (2, 1, 1)), EPOCH0_NS + dur, 60 * S)
    plain, _ = oracle_lib.propagate(compiled, batch, dur)
    np.testing.assert_array_equal(res.states.rv(), plain.rv())


def test_time_update_algebra_and_history():
    compiled, batch = two_body_rk4(n=1)
    rng = np.random.default_rng(0)
    a = rng.standard_normal((9, 9)) * 1e-3
    p0 = a @ a.T
    dev0 = rng.standard_normal((1, 9)) * 1e-3
    res = oracle_lib.predict_until(compiled, batch, p0[None], EPOCH0_NS + 95 * S, 30 * S, deviation_tracking=True,
                                   state_deviation=dev0, history=8)
    # 95 s with 30 s segments: the loop runs until epoch >= end, i.e. 4 full segments (mod.rs:465-484)
    assert res.n_updates[0] == 4 and list(res.epochs_ns[:4, 0]) == [EPOCH0_NS + k * 30 * S for k in (1, 2, 3, 4)]
    p, d = p0, dev0[0]
    for u in range(4):
        phi = res.stm[u, 0]
        p = phi @ p @ phi.T
        d = phi @ d
        np.testing.assert_allclose(res.covar_history[u, 0], p, rtol=1e-12, atol=1e-22)
        np.testing.assert_allclose(res.deviation_history[u, 0], d, rtol=1e-12, atol=1e-18)
        assert abs(np.linalg.det(phi[:6, :6]) - 1.0) < 1e-6       # each segment starts from the identity
    np.testing.assert_array_equal(res.covar[0], res.covar_history[3, 0])
    np.testing.assert_array_equal(res.nominal[3, 0, :6], res.states.rv()[0])
    # without deviation tracking the deviation is zeroed (filtering.rs:84-88)
    res2 = oracle_lib.predict_until(compiled, batch, p0[None], EPOCH0_NS + 95 * S, 30 * S, state_deviation=dev0, history=4)
    assert (res2.state_deviation == 0).all() and (res2.deviation_history == 0).all()
    np.testing.assert_array_equal(res2.covar, res.covar)
    # capacity smaller than the number of updates: counted, not stored
    res3 = oracle_lib.predict_until(compiled, batch, p0[None], EPOCH0_NS + 95 * S, 30 * S, history=2)
    assert res3.n_updates[0] == 4
    np.testing.assert_array_equal(res3.covar_history[:2], res2.covar_history[:2])
    # an end epoch that is already reached still performs one update (the loop body runs first)
    res4 = oracle_lib.predict_until(compiled, batch, p0[None], EPOCH0_NS - 5 * S, 30 * S, history=2)
    assert res4.n_updates[0] == 1 and res4.states.epoch_ns[0] == EPOCH0_NS + 30 * S


def gamma_q(dt, diag):
    g = np.zeros((9, 3))
    for i in range(3):
        g[i, i] = dt ** 2 / 2.0
        g[i + 3, i] = dt
    return g @ np.diag(diag) @ g.T


def test_process_noise_selection_and_gamma():
    compiled, batch = two_body_rk4(n=1)
    p0 = INIT_COVAR[None]
    base = oracle_lib.predict_until(compiled, batch, p0, EPOCH0_NS + 60 * S, 30 * S, history=2)
    q1, q2 = [1e-12, 2e-12, 3e-12], [5e-10, 5e-10, 5e-10]
    pn1 = nx.ProcessNoise3D.from_diagonal(q1, 2 * 60 * S)
    with_q = oracle_lib.predict_until(compiled, batch, p0, EPOCH0_NS + 60 * S, 30 * S, process_noise=[pn1], history=2)
    np.testing.assert_allclose(with_q.covar_history[0, 0] - base.covar_history[0, 0], gamma_q(30.0, q1), rtol=1e-9, atol=1e-24)
    phi = base.stm[1, 0]
    expect = phi @ with_q.covar_history[0, 0] @ phi.T + gamma_q(30.0, q1)
    np.testing.assert_allclose(with_q.covar_history[1, 0], expect, rtol=1e-12, atol=1e-22)
    # the LAST applicable noise wins (filtering.rs:64 iterates in reverse); one that has not started yet is skipped
    late = nx.ProcessNoise3D.with_start_time(2 * 60 * S, q2, EPOCH0_NS + 45 * S)
    both = oracle_lib.predict_until(compiled, batch, p0, EPOCH0_NS + 60 * S, 30 * S, process_noise=[pn1, late], history=2)
    np.testing.assert_array_equal(both.covar_history[0], with_q.covar_history[0])        # @30 s: `late` not started -> pn1
    phi = base.stm[1, 0]
    np.testing.assert_allclose(both.covar_history[1, 0], phi @ both.covar_history[0, 0] @ phi.T + gamma_q(30.0, q2), rtol=1e-12, atol=1e-22)
    # disabled when the update spans more than disable_time (snc.rs:248-250)
    off = nx.ProcessNoise3D.from_diagonal(q2, 20 * S)
    none = oracle_lib.predict_until(compiled, batch, p0, EPOCH0_NS + 60 * S, 30 * S, process_noise=[off], history=2)
    np.testing.assert_array_equal(none.covar_history, base.covar_history)
    # from_velocity_km_s: diag = v / duration (snc.rs:288-309)
    v = nx.ProcessNoise3D.from_velocity_km_s([1e-6, 2e-6, 3e-6], 10 * 60 * S, 60 * S)
    assert v.diag == [1e-6 / 600.0, 2e-6 / 600.0, 3e-6 / 600.0]


def local_dcm(rv, frame):
    r, v = rv[:3], rv[3:6]
    h = np.cross(r, v); h = h / np.linalg.norm(h)
    if frame == "RIC":
        e0 = r / np.linalg.norm(r); e2 = h; e1 = np.cross(e2, e0)
    else:
        e0 = v / np.linalg.norm(v); e1 = h; e2 = np.cross(e0, e1)
    return np.stack([e0, e1, e2], axis=1)


@pytest.mark.parametrize("frame", ["RIC", "VNC"])
def test_process_noise_decay_and_local_frame(frame):
    # snc.rs:145-160, 193-197 (decay since the initial estimate's epoch) and :219-239 (local frame: dcm * snc * dcm^T at the
    # nominal orbit with only the DIAGONAL kept, as the reference does) against numpy
    compiled, batch = two_body_rk4(n=2)
    p0 = np.repeat(INIT_COVAR[None], 2, axis=0)
    end = EPOCH0_NS + 90 * S
    base = oracle_lib.predict_until(compiled, batch, p0, end, 30 * S, history=3)
    q, decay = [1e-12, 4e-12, 9e-12], [1e-2, 2e-2, 0.0]
    pn = nx.ProcessNoise3D.with_decay(q, 2 * 60 * S, decay, local_frame=frame)
    got = oracle_lib.predict_until(compiled, batch, p0, end, 30 * S, process_noise=[pn], history=3)
    np.testing.assert_array_equal(got.stm, base.stm)
    # the nominal states at the three updates: plain propagation of the same batch
    for i in range(2):
        p = p0[i].copy()
        for u in range(3):
            t = 30.0 * (u + 1)
            nominal = oracle_lib.propagate(compiled, batch, int(t) * S)[0].rv()[i]
            d = np.array(q) * np.exp(-np.array(decay) * t)
            dcm = local_dcm(nominal, frame)
            d_in = np.diag(dcm @ np.diag(d) @ dcm.T)
            phi = got.stm[u, i]
            p = phi @ p @ phi.T + gamma_q(30.0, d_in)
            np.testing.assert_allclose(got.covar_history[u, i], p, rtol=1e-10, atol=1e-24)
    # an explicit init_epoch shifts the decay clock (ProcessNoise::init_epoch)
    import dataclasses
    shifted = dataclasses.replace(pn, init_epoch_ns=EPOCH0_NS - 100 * S, local_frame=None)
    plain = dataclasses.replace(pn, local_frame=None)
    a = oracle_lib.predict_until(compiled, batch, p0, EPOCH0_NS + 30 * S, 30 * S, process_noise=[shifted], history=1)
    b = oracle_lib.predict_until(compiled, batch, p0, EPOCH0_NS + 30 * S, 30 * S, process_noise=[plain], history=1)
    da = np.diag(a.covar_history[0, 0] - base.covar_history[0, 0])[3:6]
    db = np.diag(b.covar_history[0, 0] - base.covar_history[0, 0])[3:6]
    np.testing.assert_allclose(da[:2] / db[:2], np.exp(-np.array(decay[:2]) * 100.0), rtol=1e-6)
    # a frame the device path does not carry is refused by the host mirror
    with pytest.raises(NotImplementedError):
        oracle_lib.predict_until(compiled, batch, p0, end, 30 * S, process_noise=[dataclasses.replace(pn, local_frame="RCN")])


def test_product_of_segment_stms_tracks_the_long_stm():
    # sanity of the segment/reset scheme: prod(Phi_seg) ~ Phi of one uninterrupted propagation (first-order scheme:
    # agreement to the integration accuracy of Phi, not to round-off)
    compiled, batch = two_body_rk4(step_s=5.0, n=1)
    res = oracle_lib.predict_until(compiled, batch, INIT_COVAR[None], EPOCH0_NS + 600 * S, 60 * S, history=10)
    long_, _ = oracle_lib.propagate(compiled, batch, 600 * S)
    total = np.eye(9)
    for u in range(10):
        total = res.stm[u, 0] @ total
    phi_long = long_.stm[0].reshape(9, 9).T
    assert np.abs(total - phi_long).max() / np.abs(phi_long).max() < 1e-3
