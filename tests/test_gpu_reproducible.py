"""Reproducibility contract of nyx_hip_tuning_t (include/nyx_hip.h, ABI v4).  The reference's Monte Carlo is reproducible per
(seed, index) whatever the batch (mc/montecarlo.rs:208-224, 290-295: every run is seeded and propagated on its own).  Here:
default schedule = a cost model => the same bits in every context; deterministic = 1 => the same bits whatever the batch."""
import numpy as np
import pytest

import nyx_amd as nx
from scenarios import dispersed_leo_batch, leo_full_setup, pos_vel_errors

pytestmark = pytest.mark.gpu


def test_two_fresh_contexts_give_identical_bits():
    prop, almanac, central = leo_full_setup(degree=70)
    compiled = prop.compile(almanac, central)
    b = dispersed_leo_batch(10_000, seed=5)
    dur = 40 * 60 * nx.NS_PER_S
    outs = []
    for _ in range(2):
        ctx = nx.GpuContext(compiled)          # default tuning: NYX_HIP_SCHED_MODEL, cooperative mode on (99 helpers: every CU the 157 owners leave)
        out, st = ctx.propagate(b, dur)
        assert (st.status == 0).all() and ctx.last_coop_helpers() == 99
        outs.append((out.rv().copy(), st.n_evals.copy()))
        ctx.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_deterministic_mode_is_independent_of_the_batch():
    """Rank shards (ragged), a permuted batch and a single trajectory propagated alone all reproduce the full batch bit for bit."""
    prop, almanac, central = leo_full_setup(degree=70)
    compiled = prop.compile(almanac, central)
    n = 3000
    b = dispersed_leo_batch(n, seed=9)
    dur = 40 * 60 * nx.NS_PER_S
    ctx = nx.GpuContext(compiled, tuning=nx.Tuning(deterministic=1))
    full, st = ctx.propagate(b, dur)
    assert (st.status == 0).all() and ctx.last_coop_helpers() == 0
    for rank in range(3):                      # contiguous index shards of a 3-rank job
        lo, hi = nx.shard_bounds(n, rank, 3)
        part, _ = ctx.propagate(b.slice(lo, hi), dur)
        np.testing.assert_array_equal(part.rv(), full.rv()[lo:hi])
    other = nx.GpuContext(compiled, tuning=nx.Tuning(deterministic=1))   # another context, another batch composition
    perm = np.random.default_rng(0).permutation(n)
    shuffled, _ = other.propagate(b.take(perm), dur)
    np.testing.assert_array_equal(shuffled.rv(), full.rv()[perm])
    one, _ = other.propagate(b.slice(1234, 1235), dur)
    np.testing.assert_array_equal(one.rv(), full.rv()[1234:1235])
    # ... and the cooperative default differs from it only by the summation order of the harmonics
    coop_ctx = nx.GpuContext(compiled)
    coop, _ = coop_ctx.propagate(b, dur)
    assert coop_ctx.last_coop_helpers() > 0
    dr, dv = pos_vel_errors(coop, full)
    assert dr.max() < 1e-6 and dv.max() < 1e-9
    for c in (ctx, other, coop_ctx):
        c.close()


def test_calibrated_schedule_is_opt_in_and_agrees():
    prop, almanac, central = leo_full_setup(degree=70)
    compiled = prop.compile(almanac, central)
    b = dispersed_leo_batch(4096, seed=2)
    dur = 2 * 3600 * nx.NS_PER_S
    model = nx.GpuContext(compiled)
    ref, _ = model.propagate(b, dur)
    cal = nx.GpuContext(compiled, tuning=nx.Tuning(schedule=nx.SCHED_CALIBRATED))
    out, st = cal.propagate(b, dur)
    assert (st.status == 0).all()
    again, _ = cal.propagate(b, dur)
    np.testing.assert_array_equal(out.rv(), again.rv())   # a context stays deterministic once its schedule is measured
    dr, dv = pos_vel_errors(out, ref)
    assert dr.max() < 1e-6 and dv.max() < 1e-9
    model.close()
    cal.close()


def test_sharded_entry_point_equals_the_single_context_run():
    """nyx_hip_propagate_batch_sharded (the multi-device driver of the C-ABI) with three contexts on device 0 - on a node: one per
    device - against one context, deterministic tuning: bit for bit, ragged shards included."""
    prop, almanac, central = leo_full_setup(degree=21)
    compiled = prop.compile(almanac, central)
    b = dispersed_leo_batch(1000, seed=4)
    dur = 30 * 60 * nx.NS_PER_S
    tun = nx.Tuning(deterministic=1)
    one = nx.GpuContext(compiled, tuning=tun)
    ref, rst = one.propagate(b, dur)
    ctxs = [nx.GpuContext(compiled, tuning=tun) for _ in range(3)]
    out, st = nx.propagate_sharded(ctxs, b, dur)
    assert (st.status == 0).all()
    np.testing.assert_array_equal(out.rv(), ref.rv())
    np.testing.assert_array_equal(out.epoch_ns, ref.epoch_ns)
    np.testing.assert_array_equal(st.n_evals, rst.n_evals)
    for c in ctxs + [one]:
        c.close()


def test_deterministic_mode_across_the_workgroup_shape_thresholds():
    """the workgroup shape (waves per workgroup) used to follow the batch size even with deterministic = 1 - 8 waves from
    32 705 trajectories on, 16 below - and with it the column split and the last bits.  A batch above the threshold against its two
    halves below it, and against a few trajectories propagated alone: bit for bit."""
    prop, almanac, central = leo_full_setup(degree=24)   # (degree >= 24: the sixteen-wave shape is the configuration's)
    compiled = prop.compile(almanac, central)
    n = 33_000
    b = dispersed_leo_batch(n, seed=11)
    dur = 10 * 60 * nx.NS_PER_S
    ctx = nx.GpuContext(compiled, tuning=nx.Tuning(deterministic=1))
    full, st = ctx.propagate(b, dur)
    assert (st.status == 0).all()
    for lo, hi in ((0, n // 2), (n // 2, n), (777, 841)):
        part, pst = ctx.propagate(b.slice(lo, hi), dur)
        np.testing.assert_array_equal(part.rv(), full.rv()[lo:hi])
        np.testing.assert_array_equal(pst.n_evals, st.n_evals[lo:hi])
    ctx.close()


def test_deterministic_stm_layout_does_not_follow_the_batch():
    """The STM layout (quad below ~8 000 trajectories, 64-lane duals above) is part of the shape: deterministic = 1 pins it."""
    prop, almanac, central = leo_full_setup(degree=8)
    compiled = prop.compile(almanac, central, stm=True)
    n = 8704   # > 2 x 256 x 16: the automatic choice for the full batch would be the 64-lane layout, for a slice the quad layout
    b = dispersed_leo_batch(n, seed=3)
    b.stm = np.zeros((n, 81))
    b.reset_stm()
    dur = 120 * nx.NS_PER_S
    ctx = nx.GpuContext(compiled, tuning=nx.Tuning(deterministic=1))
    full, st = ctx.propagate(b, dur)
    assert (st.status == 0).all()
    part, _ = ctx.propagate(b.slice(100, 164), dur)
    np.testing.assert_array_equal(part.rv(), full.rv()[100:164])
    np.testing.assert_array_equal(part.stm, full.stm[100:164])
    ctx.close()
