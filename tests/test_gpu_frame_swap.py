"""opts.integration_frame on the device (nyx_hip_config_t.state_frame_body): Moon-centred states integrated in the Earth frame,
translated in and out by nyx_frame_shift_kernel - against the oracle twin, and the entry points that refuse a swap."""
import dataclasses

import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from frame_swap_cases import MOON_FRAME, moon_batch, setup

pytestmark = pytest.mark.gpu


def test_swap_vs_oracle_and_refusals():
    prop, almanac, earth = setup()
    b = moon_batch(100, seed=3)
    dur = 2 * 3600 * nx.NS_PER_S
    compiled = prop.compile(almanac, earth, state_frame=MOON_FRAME)
    ctx = nx.GpuContext(compiled)
    out, st = ctx.propagate(b, dur)
    ref, rst = oracle_lib.propagate(compiled, b, dur, n_threads=8)
    assert (st.status == 0).all() and (rst.status == 0).all()
    np.testing.assert_array_equal(out.epoch_ns, ref.epoch_ns)
    d = out.rv() - ref.rv()
    dr, dv = np.linalg.norm(d[:, :3], axis=1).max(), np.linalg.norm(d[:, 3:], axis=1).max()
    print(f"swap: dr {dr*1e3:.2e} m dv {dv*1e6:.2e} mm/s")
    assert dr < 1e-3 and dv < 1e-6                      # the path's parity bar: 1 m, 1 mm/s
    assert np.all(np.linalg.norm(out.rv()[:, :3], axis=1) < 2200.0)      # still Moon-centred
    # per-trajectory end epochs, same translation
    end = int(b.epoch_ns.max()) + 1800 * nx.NS_PER_S
    o2, s2 = ctx.propagate_until_epoch(b, end)
    assert (s2.status == 0).all() and (o2.epoch_ns == end).all()
    # dense output of a swapped run: the reference's mixture (instance.rs:297-326) - entry 0 in the caller's frame, the published
    # states in the integration frame, the returned state translated back - against the oracle twin, state by state
    o3, s3, traj = ctx.propagate_with_traj(b, dur, capacity=64)
    r3, rs3, rtraj = oracle_lib.propagate_with_traj(compiled, b, dur, int(traj.len.max()), n_threads=8)
    np.testing.assert_array_equal(o3.rv(), out.rv())                   # recording changes nothing
    assert (traj.len == s3.n_accepted + 1).all() and (rtraj.len == rs3.n_accepted + 1).all()
    for i in range(b.n):
        ep, rv = traj.trajectory(i)
        np.testing.assert_array_equal(rv[0], b.rv()[i])                # Moon-centred, untouched
        assert ep[0] == b.epoch_ns[i] and ep[-1] == o3.epoch_ns[i] and np.all(np.diff(ep) > 0)
        assert np.all(np.linalg.norm(rv[1:, :3], axis=1) > 3.0e5)      # Earth-centred
    # (a lunar orbit integrated in the Earth frame is a roundoff-driven step sequence - 16 s steps, the two sides take 370-510 of
    #  them per run, not the same ones: the trajectories are compared where both are defined, through the oracle's interpolation, on
    #  epochs whose 13-state windows lie past the mixed first state)
    from scenarios import EPOCH0_NS
    at = [EPOCH0_NS + int(t) * nx.NS_PER_S for t in (5000, 5600, 6200, 6800, 7100)]
    dev_at, dst = oracle_lib.traj_at(traj, at)
    ref_at, rst = oracle_lib.traj_at(rtraj, at)
    assert (dst == 0).all() and (rst == 0).all()
    worst = np.linalg.norm(dev_at[..., :3] - ref_at[..., :3], axis=-1).max()
    print(f"swap + dense output: worst dr {worst*1e3:.2e} m between the two interpolated trajectories (Earth-centred)")
    assert worst < 1e-3 and np.all(np.linalg.norm(dev_at[..., :3], axis=-1) > 3.0e5)
    # the event search still refuses states of another frame (the reference returns them untranslated from inside its loop)
    with pytest.raises(RuntimeError, match="integration-frame swap"):
        ctx.propagate_until_event(b, dur, nx.Event.apoapsis(), 1, 64)
    ctx.close()
    # outside the ephemeris: reported per run by the translation
    late = b.copy()
    late.epoch_ns[:] += 400 * 86400 * nx.NS_PER_S
    ctx = nx.GpuContext(compiled)
    _, s3 = ctx.propagate(late, 600 * nx.NS_PER_S)
    assert (s3.status == nx._abi.ERR_EPHEM_RANGE).all()
    ctx.close()


def test_mirror_many_for_duration_with_integration_frame():
    prop, almanac, earth = setup()
    prop.opts = dataclasses.replace(prop.opts, integration_frame=earth)
    b = moon_batch(5, seed=8)
    scs = [nx.Spacecraft(int(b.epoch_ns[i]), b.rv()[i], MOON_FRAME, cr=1.5, dry_mass_kg=200.0, srp_area_m2=2.0) for i in range(5)]
    res = prop.many_for_duration(scs, almanac, 1800 * nx.NS_PER_S)
    assert len(res) == 5 and all(r.frame.naif_id == nx.MOON for r in res)
    ref, _ = oracle_lib.propagate(prop.compile(almanac, earth, state_frame=MOON_FRAME), b, 1800 * nx.NS_PER_S)
    got = np.array([np.asarray(r.rv) for r in res])
    assert np.abs(got - ref.rv()).max() < 1e-6


def test_covariance_map_of_a_swapped_run_vs_oracle():
    """nyx_hip_predict_until with opts.integration_frame: one translation in and one back per segment (od/process/mod.rs:453-468 =
    one `until_epoch` each), runs that have reached their end are left alone; against the oracle twin."""
    from scenarios import EPOCH0_NS
    prop, almanac, earth = setup()
    n = 40
    b = moon_batch(n, seed=6)
    b.epoch_ns[:] = EPOCH0_NS + (np.arange(n) % 3) * 60 * nx.NS_PER_S    # ragged starts: some runs finish a segment or two early
    b.stm = np.zeros((n, 81))
    b.reset_stm()
    rng = np.random.default_rng(1)
    p0 = np.zeros((n, 9, 9))
    for i in range(n):
        a = rng.standard_normal((9, 9)) * np.array([1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3, 1e-2, 0.0, 0.0])[:, None]
        p0[i] = a @ a.T
    end = EPOCH0_NS + 8 * 60 * nx.NS_PER_S
    compiled = prop.compile(almanac, earth, stm=True, state_frame=MOON_FRAME)
    pn = [nx.ProcessNoise3D.from_diagonal([1e-14, 1e-14, 2e-14], 2 * 60 * nx.NS_PER_S, local_frame="RIC")]
    ctx = nx.GpuContext(compiled)
    got = nx.predict_until(ctx, b, p0, end, 60 * nx.NS_PER_S, process_noise=pn, history=8)
    ref = oracle_lib.predict_until(compiled, b, p0, end, 60 * nx.NS_PER_S, process_noise=pn, history=8)
    ctx.close()
    assert (got.stats.status == 0).all() and (ref.stats.status == 0).all()
    np.testing.assert_array_equal(got.n_updates, ref.n_updates)
    assert set(got.n_updates.tolist()) == {6, 7, 8}
    np.testing.assert_array_equal(got.states.epoch_ns, ref.states.epoch_ns)
    d = got.states.rv() - ref.states.rv()
    dr, dv = np.linalg.norm(d[:, :3], axis=1).max(), np.linalg.norm(d[:, 3:], axis=1).max()
    scale = np.abs(ref.covar).max(axis=(-2, -1), keepdims=True)
    e_p = (np.abs(got.covar - ref.covar) / scale).max()
    print(f"swap + covariance map: dr {dr*1e3:.2e} m dv {dv*1e6:.2e} mm/s Pbar {e_p:.2e}")
    assert dr < 1e-3 and dv < 1e-6 and e_p < 1e-9
    assert np.all(np.linalg.norm(got.states.rv()[:, :3], axis=1) < 2200.0)       # handed back Moon-centred
