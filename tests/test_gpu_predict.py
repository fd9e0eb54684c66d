"""Device covariance mapping (nyx_hip_predict_until: STM segments + predict_kernel.hip time updates, no host round trip)
against the oracle.  BASELINE config 4 pattern: GEO states, 21x21 + Sun/Moon + SRP with Cr estimated, 1-minute
segments with Phi reset (od/process/mod.rs:466-483); parity on Phi and P-bar: 1e-9 relative, element-wise."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from nyx_amd import ephem
from scenarios import EPOCH0_NS, dispersed_leo_batch, keplerian_to_cartesian, leo_full_setup

pytestmark = pytest.mark.gpu
S = nx.NS_PER_S


def geo_batch(n, seed):
    b = dispersed_leo_batch(n, seed=seed)
    geo = keplerian_to_cartesian(42164.0, 1e-5, 0.0, 163.0, 75.0, 0.0, ephem.MU_EARTH)   # examples/03_geo_analysis/drift.rs:50
    rv = b.rv()
    b.set_rv(geo[None, :] + (rv - rv.mean(axis=0)))
    return b


def init_covar(n, seed=0):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 9, 9))
    for i in range(n):
        a = rng.standard_normal((9, 9)) * np.array([1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3, 1e-2, 0.0, 0.0])[:, None]
        out[i] = a @ a.T
    return out


def rel_err(got, ref):
    scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max(axis=(-2, -1), keepdims=True))
    return (np.abs(got - ref) / scale).max()


def test_config4_geo_covariance_map_vs_oracle():
    prop, almanac, central = leo_full_setup(degree=21)
    compiled = prop.compile(almanac, central, stm=True)
    ctx = nx.GpuContext(compiled)
    n = 70
    b = geo_batch(n, seed=4)
    p0 = init_covar(n)
    pn = [nx.ProcessNoise3D.from_diagonal([1e-14, 1e-14, 2e-14], 2 * 60 * S)]
    end = EPOCH0_NS + 10 * 60 * S
    got = nx.predict_until(ctx, b, p0, end, 60 * S, process_noise=pn, deviation_tracking=True,
                           state_deviation=np.full((n, 9), 1e-3), history=10)
    ref = oracle_lib.predict_until(compiled, b, p0, end, 60 * S, process_noise=pn, deviation_tracking=True,
                                   state_deviation=np.full((n, 9), 1e-3), history=10)
    assert (got.stats.status == 0).all() and (got.n_updates == 10).all() and (ref.n_updates == 10).all()
    np.testing.assert_array_equal(got.epochs_ns, ref.epochs_ns)
    np.testing.assert_array_equal(got.states.epoch_ns, ref.states.epoch_ns)
    # 1-minute segments at GEO are single RK89 attempts at h = 60 s for both: identical step sequences
    np.testing.assert_array_equal(got.stats.n_accepted, ref.stats.n_accepted)
    dr = np.linalg.norm(got.states.rv()[:, :3] - ref.states.rv()[:, :3], axis=1).max()
    dv = np.linalg.norm(got.states.rv()[:, 3:] - ref.states.rv()[:, 3:], axis=1).max()
    e_phi, e_p = rel_err(got.stm, ref.stm), rel_err(got.covar_history, ref.covar_history)
    e_dev = np.abs(got.deviation_history - ref.deviation_history).max() / np.abs(ref.deviation_history).max()
    print(f"config 4: dr {dr*1e3:.2e} m dv {dv*1e6:.2e} mm/s Phi {e_phi:.2e} Pbar {e_p:.2e} dev {e_dev:.2e} "
          f"device time {got.kernel_ms:.2f} ms for 10 updates of {n}")
    assert dr < 1e-3 and dv < 1e-6
    assert e_phi < 1e-9 and e_p < 1e-9 and e_dev < 1e-9
    assert rel_err(got.covar, ref.covar) < 1e-9
    assert np.abs(got.stm[:, :, 3:6, 6]).max() > 0.0        # d v / d Cr: SRP with Cr estimated, sunlit GEO
    np.testing.assert_array_equal(got.states.stm, np.tile(np.eye(9).ravel(), (n, 1)))
    # the time update itself, given the device's own Phi, is exact algebra: bit-equal to numpy in nalgebra's order
    # is not required, 1e-13 relative is
    p = p0.copy()
    for u in range(10):
        phi = got.stm[u]
        p = phi @ p @ np.transpose(phi, (0, 2, 1))
        g = np.zeros((9, 3)); g[[0, 1, 2], [0, 1, 2]] = 60.0 ** 2 / 2; g[[3, 4, 5], [0, 1, 2]] = 60.0
        p = p + g @ np.diag(pn[0].diag) @ g.T
        assert rel_err(got.covar_history[u], p) < 1e-12
    ctx.close()


@pytest.mark.parametrize("frame", [None, "RIC", "VNC"])
def test_process_noise_decay_and_local_frame_vs_oracle(frame):
    """ProcessNoise::with_decay and local_frame (od/snc.rs:145-160, 193-197, 219-239) on the device against the oracle:
    same arithmetic order, so the noise contribution agrees to the last bits of Phi P Phi^T."""
    prop, almanac, central = leo_full_setup(degree=8)
    compiled = prop.compile(almanac, central, stm=True)
    ctx = nx.GpuContext(compiled)
    n = 40
    b = dispersed_leo_batch(n, seed=9)
    b.epoch_ns[: n // 2] += 17 * S                       # per-trajectory initial epochs: each has its own decay clock
    p0 = init_covar(n, seed=2)
    pns = [nx.ProcessNoise3D.from_diagonal([1e-13] * 3, 10 * 60 * S),
           nx.ProcessNoise3D.with_decay([1e-12, 4e-12, 9e-12], 10 * 60 * S, [5e-3, 1e-2, 0.0], local_frame=frame)]
    pns[1].start_time_ns = EPOCH0_NS + 100 * S          # the constant one applies first, the decaying one afterwards
    end = EPOCH0_NS + 6 * 60 * S
    got = nx.predict_until(ctx, b, p0, end, 60 * S, process_noise=pns, history=6)
    ref = oracle_lib.predict_until(compiled, b, p0, end, 60 * S, process_noise=pns, history=6)
    none = nx.predict_until(ctx, b, p0, end, 60 * S, history=6)
    assert (got.stats.status == 0).all() and (ref.n_updates == 6).all()
    np.testing.assert_array_equal(got.n_updates, ref.n_updates)
    e_p = rel_err(got.covar_history, ref.covar_history)
    # the noise term on its own (P with noise - P without) is a small difference of large numbers: compare it at the
    # precision that leaves, and make sure it is there at all
    q_got, q_ref = got.covar_history[0] - none.covar_history[0], ref.covar_history[0] - none.covar_history[0]
    assert np.abs(q_ref[:, 3, 3]).min() > 0.0
    print(f"decay + {frame}: Pbar {e_p:.2e}")
    assert e_p < 1e-9
    np.testing.assert_allclose(q_got[:, 3:6, 3:6], q_ref[:, 3:6, 3:6], rtol=1e-3, atol=1e-16)
    ctx.close()


def test_ragged_epochs_failures_and_capacity():
    prop, almanac, central = leo_full_setup(degree=4)
    compiled = prop.compile(almanac, central, stm=True)
    ctx = nx.GpuContext(compiled)
    n = 6
    b = dispersed_leo_batch(n, seed=9)
    b.epoch_ns[:] = EPOCH0_NS + np.array([0, 30, 60, 90, 200, 400]) * S     # different numbers of segments per trajectory
    b.dry_mass_kg[3] = 0.0                                                  # MasslessSpacecraft: fails in its first segment
    b.prop_mass_kg[3] = 0.0
    p0 = init_covar(n, seed=3)
    end = EPOCH0_NS + 300 * S
    got = nx.predict_until(ctx, b, p0, end, 60 * S, history=3)
    ref = oracle_lib.predict_until(compiled, b, p0, end, 60 * S, history=3)
    np.testing.assert_array_equal(got.n_updates, ref.n_updates)
    assert list(got.n_updates) == [5, 5, 4, 0, 2, 1]                        # ceil((end - epoch) / 60 s), >= 1; failure: none
    np.testing.assert_array_equal(got.stats.status, ref.stats.status)
    assert got.stats.status[3] == nx._abi.ERR_MASSLESS and (np.delete(got.stats.status, 3) == 0).all()
    np.testing.assert_array_equal(got.states.epoch_ns, ref.states.epoch_ns)
    ok = np.array([0, 1, 2, 4, 5])
    np.testing.assert_array_equal(got.stats.n_accepted[ok], ref.stats.n_accepted[ok])
    np.testing.assert_array_equal(got.covar[3], p0[3])                      # untouched
    assert rel_err(got.covar[ok], ref.covar[ok]) < 1e-9
    for i in ok:
        m = min(int(got.n_updates[i]), 3)
        np.testing.assert_array_equal(got.epochs_ns[:m, i], ref.epochs_ns[:m, i])
        assert rel_err(got.covar_history[:m, i], ref.covar_history[:m, i]) < 1e-9
    # host-loop equivalent: the same result as segment-by-segment calls through the plain propagation entry
    g = b.copy(); g.stm = np.zeros((n, 81)); g.reset_stm()
    seg, st = ctx.propagate(g, 60 * S)
    phi0 = seg.stm[0].reshape(9, 9).T
    assert rel_err(got.stm[0, 0][None], phi0[None]) < 1e-12
    ctx.close()


def test_requires_an_stm_context_and_valid_config():
    prop, almanac, central = leo_full_setup(degree=2)
    ctx = nx.GpuContext(prop.compile(almanac, central, stm=False))
    b = dispersed_leo_batch(2, seed=1)
    with pytest.raises(RuntimeError, match="NYX_HIP_FLAG_STM"):
        nx.predict_until(ctx, b, init_covar(2), EPOCH0_NS + 60 * S, 60 * S)
    ctx.close()
    ctx = nx.GpuContext(prop.compile(almanac, central, stm=True))
    with pytest.raises(RuntimeError, match="max_step_ns"):
        nx.predict_until(ctx, b, init_covar(2), EPOCH0_NS + 60 * S, 0)
    ctx.close()


def test_config4_full_size_one_hour():
    # N = 1 000 GEO states (BASELINE config 4), 60 one-minute updates in one call; properties + head vs the oracle
    prop, almanac, central = leo_full_setup(degree=21)
    compiled = prop.compile(almanac, central, stm=True)
    ctx = nx.GpuContext(compiled)
    n = 1000
    b = geo_batch(n, seed=11)
    p0 = np.tile(np.diag([1.0, 1.0, 1.0, 1e-6, 1e-6, 1e-6, 1e-4, 0.0, 0.0]), (n, 1, 1))
    end = EPOCH0_NS + 3600 * S
    got = nx.predict_until(ctx, b, p0, end, 60 * S, history=60, keep_stm=False)
    assert (got.stats.status == 0).all() and (got.n_updates == 60).all()
    assert (got.states.epoch_ns == end).all()
    d = np.diagonal(got.covar_history, axis1=2, axis2=3)
    assert (d >= 0).all() and (d[-1, :, :3] > 1.0).all()                         # positive, position variance inflates
    asym = np.abs(got.covar - np.transpose(got.covar, (0, 2, 1))).max() / np.abs(got.covar).max()
    assert asym < 1e-12
    head = nx._abi.StateBatch(8)
    for f in ["epoch_ns"] + nx._abi.F64_FIELDS:
        getattr(head, f)[:] = getattr(b, f)[:8]
    ref = oracle_lib.predict_until(compiled, head, p0[:8], end, 60 * S, history=60, keep_stm=False)
    assert rel_err(got.covar_history[:, :8], ref.covar_history) < 1e-9
    print(f"config 4 full size: 60 updates x {n} trajectories in {got.kernel_ms:.1f} ms of device time "
          f"({got.kernel_ms / 60:.2f} ms per update)")
    ctx.close()


@pytest.mark.parametrize("frame", [None, "RIC"])
def test_one_launch_loop_equals_the_launch_per_segment_loop(frame):
    """Round 6: the whole covariance-mapping loop runs in ONE launch (quad layout; the integrator workgroup stays resident and every wave
    takes a share of the Kalman time updates at a segment boundary, propagate_kernel.hip segment_update) instead of a segment launch + a
    time-update launch per segment (rounds 2-5, debug_flags 0x20000000; od/process/mod.rs:440-486, od/kalman/filtering.rs:59-99).  Same
    arithmetic on the same operands: final states, covariances, deviations, the whole history (epochs, states, STMs, covariances),
    update counts and step counters are bit-identical - ragged start epochs (trajectories leave the loop at different segments), a
    partial last workgroup, process noise with decay in a local frame."""
    prop, almanac, central = leo_full_setup(degree=21)
    compiled = prop.compile(almanac, central, stm=True)
    n = 40
    b = geo_batch(n, 13)
    b.epoch_ns[:] = EPOCH0_NS + (np.arange(n) % 4) * 45 * S        # ragged: 8 to 10 updates to the common end epoch
    b.stm = np.zeros((n, 81)); b.reset_stm()
    end = EPOCH0_NS + 600 * S
    pn = [nx.ProcessNoise3D(diag=(1e-12, 2e-12, 3e-12), disable_time_ns=3600 * S, decay_s=(1e-4, 2e-4, 0.0) if frame else None, local_frame=frame)]
    dev0 = np.random.default_rng(3).standard_normal((n, 9)) * 1e-3
    res = {}
    for name, flags in (("one launch", 0), ("per segment", 0x20000000)):
        ctx = nx.GpuContext(compiled, tuning=nx.Tuning(debug_flags=flags))
        res[name] = nx.predict_until(ctx, b, init_covar(n, 2), end, 60 * S, process_noise=pn, deviation_tracking=True, state_deviation=dev0, history=12)
        ctx.close()
    a, r = res["one launch"], res["per segment"]
    assert (a.stats.status == 0).all() and a.n_updates.min() == 8 and a.n_updates.max() == 10
    np.testing.assert_array_equal(a.n_updates, r.n_updates)
    np.testing.assert_array_equal(a.states.rv(), r.states.rv())
    np.testing.assert_array_equal(a.states.epoch_ns, r.states.epoch_ns)
    np.testing.assert_array_equal(a.states.stm, r.states.stm)
    np.testing.assert_array_equal(a.covar, r.covar)
    np.testing.assert_array_equal(a.state_deviation, r.state_deviation)
    for f in ("epochs_ns", "nominal", "covar_history", "deviation_history", "stm"):
        np.testing.assert_array_equal(getattr(a, f), getattr(r, f), err_msg=f)
    for f in ("n_accepted", "n_rejected", "n_evals"):
        np.testing.assert_array_equal(getattr(a.stats, f), getattr(r.stats, f), err_msg=f)
