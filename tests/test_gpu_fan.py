"""-m gpu: the fan-out mode of the cooperative launch (round 6; nyx_amd/csrc/propagate_kernel.hip helper_body under NYX_COOP_FAN,
abi.cpp launch()).  When the idle CUs outnumber the trajectory-owning workgroups at least two to one - what a rank runs when ONE
ensemble is cut over the GPUs of a node (mc/montecarlo.rs:233-273 is the loop being sharded), or any small Monte Carlo - every owner
gets K dedicated helper workgroups (no claims), hands them all but its shortest columns and keeps the single-part mailbox protocol; the
part-0 helper adds the other parts' sums in part order.  Checked here: the result against the CPU oracle at the shard sizes of
configs[1] (1 250 / 2 500 / 5 000 trajectories = 8 / 5 / 2 helpers per owner) and for a single workgroup; against the claim mode and
against workgroups working alone (three roundings of the same sums); the owner's fallback when the helpers never answer; helper counts."""
import os

import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from scenarios import dispersed_leo_batch, leo_full_setup

pytestmark = pytest.mark.gpu
S = nx.NS_PER_S
NO_FAN = 0x8000000


def _setup():
    prop, almanac, central = leo_full_setup(degree=70)
    return prop.compile(almanac, central)


def _err(out, ref, idx):
    d = out.rv()[idx] - ref.rv()
    return np.linalg.norm(d[:, :3], axis=1).max(), np.linalg.norm(d[:, 3:], axis=1).max()


@pytest.mark.parametrize("n, helpers", [(64, 8), (1250, 160), (2500, 200), (5000, 158)])
def test_fan_out_against_the_oracle_and_the_other_schedules(n, helpers):
    compiled = _setup()
    batch = dispersed_leo_batch(n, seed=23)
    dur = 3600 * S
    ctx = nx.GpuContext(compiled)
    out, st = ctx.propagate(batch, dur)
    assert ctx.last_coop_helpers() == helpers      # owners x min(8, idle CUs / owners) on a 256-CU device
    ctx.close()
    assert (st.status == 0).all()
    idx = np.arange(0, n, max(1, n // 48))[:48]
    ref, rst = oracle_lib.propagate(compiled, batch.take(idx), dur, n_threads=os.cpu_count() or 1)
    assert (rst.status == 0).all()
    dr, dv = _err(out, ref, idx)
    assert dr < 1e-3 and dv < 1e-6, (n, dr, dv)      # km, km/s: the 1 m / 1 mm/s bar of north_star (measured: 0.06 mm after an hour)
    # the claim mode of rounds 1-5 and workgroups working alone: other summation orders of the same terms
    for tuning in (dict(debug_flags=NO_FAN), dict(cooperative=0)):
        c2 = nx.GpuContext(compiled, tuning=nx.Tuning(**tuning))
        o2, s2 = c2.propagate(batch, dur)
        c2.close()
        assert (s2.status == 0).all()
        d = out.rv() - o2.rv()
        assert np.linalg.norm(d[:, :3], axis=1).max() < 1e-5 and np.linalg.norm(d[:, 3:], axis=1).max() < 1e-8, tuning   # 1 cm / 0.01 mm/s after an hour (measured ~0.1 mm)


def test_fan_out_owner_falls_back_when_no_helper_answers():
    """coop_mute: the helpers leave at once.  Every owner waits its 2 ms, walks every part's columns itself (coop_fallback: the parts'
    sums in part order, then integ_back_slow with the stage's DCM evaluated again) and carries on alone: same parity, no failure."""
    compiled = _setup()
    batch = dispersed_leo_batch(640, seed=5)
    dur = 1200 * S
    ctx = nx.GpuContext(compiled, tuning=nx.Tuning(coop_mute=1))
    out, st = ctx.propagate(batch, dur)
    assert ctx.last_coop_helpers() == 80
    ctx.close()
    assert (st.status == 0).all()
    idx = np.arange(0, 640, 10)
    ref, rst = oracle_lib.propagate(compiled, batch.take(idx), dur, n_threads=os.cpu_count() or 1)
    dr, dv = _err(out, ref, idx)
    assert dr < 1e-3 and dv < 1e-6, (dr, dv)
    # ... and it is the result of workgroups that never had helpers, but for the first evaluation's summation order
    c2 = nx.GpuContext(compiled, tuning=nx.Tuning(cooperative=0))
    o2, _ = c2.propagate(batch, dur)
    c2.close()
    d = out.rv() - o2.rv()
    assert np.linalg.norm(d[:, :3], axis=1).max() < 1e-6


def test_fan_out_is_reproducible_and_follows_the_batch_size():
    """Two launches of one context give the same bits; a batch that fills the chip goes back to the claim mode (99 helpers for 157 owners)."""
    compiled = _setup()
    ctx = nx.GpuContext(compiled)
    small = dispersed_leo_batch(1280, seed=9)
    a, _ = ctx.propagate(small, 900 * S)
    h1 = ctx.last_coop_helpers()
    big = dispersed_leo_batch(10000, seed=9)
    ctx.propagate(big, 300 * S)
    h2 = ctx.last_coop_helpers()
    b, _ = ctx.propagate(small, 900 * S)
    h3 = ctx.last_coop_helpers()
    ctx.close()
    assert (h1, h2, h3) == (160, 99, 160)
    np.testing.assert_array_equal(a.rv(), b.rv())


def test_fan_out_step_control_out_of_line_is_the_inline_one_and_records_the_same_trajectory():
    """Round 6: the fan-out kernel's step control is a leaf function (integ_step, pk_integrator_ool.h) and the caller writes the dense
    output from the cold state it stored; a launch with a stop condition keeps the INLINE step control of the same kernel (event_step
    is a call).  Same kernel, same column schedule, same sums: a stop condition that never fires must leave every bit where the
    out-of-line step control leaves it - final states, epochs, counters and every recorded state of every run."""
    from nyx_amd import _abi
    compiled = _setup()
    batch = dispersed_leo_batch(128, seed=5)
    dur = 45 * 60 * S
    ctx = nx.GpuContext(compiled)
    out_p, st_p = ctx.propagate(batch, dur)
    assert ctx.last_coop_helpers() == 16           # two owners x eight dedicated helpers: the fan-out kernel
    out_a, st_a, tr_a = ctx.propagate_with_traj(batch, dur, 512)
    never = nx.Event(_abi.EV_RMAG_KM, 1.0e6)      # |r| = 1e6 km: no sign change in 45 min of a LEO
    out_b, st_b, tr_b, crossings = ctx.propagate_until_event(batch, dur, never, trigger=1, capacity=512)
    ctx.close()
    assert (st_p.status == 0).all() and (st_a.status == 0).all()
    assert (st_b.status == _abi.ERR_EVENT_NOT_FOUND).all() and (crossings == 0).all()
    for o in (out_a, out_b):
        np.testing.assert_array_equal(o.rv(), out_p.rv())
        np.testing.assert_array_equal(o.epoch_ns, out_p.epoch_ns)
    for s in (st_a, st_b):
        np.testing.assert_array_equal(s.n_accepted, st_p.n_accepted)
        np.testing.assert_array_equal(s.n_rejected, st_p.n_rejected)
        np.testing.assert_array_equal(s.n_evals, st_p.n_evals)
    assert (st_p.n_rejected > 0).any()             # (both branches of the controller were walked)
    np.testing.assert_array_equal(tr_a.len, st_p.n_accepted + 1)
    for i in range(0, 128, 7):
        m = int(tr_a.len[i])
        np.testing.assert_array_equal(tr_a.epoch_ns[:m, i], tr_b.epoch_ns[:m, i])
        np.testing.assert_array_equal(tr_a.state[:, :m, i], tr_b.state[:, :m, i])
        np.testing.assert_array_equal(tr_a.state[:, m - 1, i], out_p.rv()[i])
        assert tr_a.epoch_ns[m - 1, i] == out_p.epoch_ns[i]
