"""-m gpu: the quad layout of the STM kernel (16 trajectories x 4 lanes per wave, one position partial per lane) against
the 64-lane layout with three-partial duals.  Same expressions per value and per partial => with the same column split the
two layouts agree BIT FOR BIT on states and on every element of Phi; and the quad layout meets the oracle like the D3 one.
(The default picks the quad layout for small ensembles: the other STM tests of this suite run on it.)"""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from nyx_amd import ephem
from scenarios import dispersed_leo_batch, keplerian_to_cartesian, leo_full_setup

pytestmark = pytest.mark.gpu
S = nx.NS_PER_S


def stm_batch(n, seed, geo_tail=True):
    b = dispersed_leo_batch(n, seed=seed)
    b.stm = np.zeros((n, 81))
    b.reset_stm()
    if geo_tail:
        geo = keplerian_to_cartesian(42164.0, 1e-5, 0.0, 163.0, 75.0, 0.0, ephem.MU_EARTH)
        rv = b.rv()
        k = n // 2
        rv[k:] = geo[None, :] + (rv[k:] - rv[k:].mean(axis=0))
        b.set_rv(rv)
    return b


def run(compiled, batch, dur, layout, waves=0):
    ctx = nx.GpuContext(compiled)
    ctx.set_stm_layout(layout)
    if waves:
        ctx.set_column_waves(waves)
    out, st = ctx.propagate(batch, dur)
    ms = ctx.last_kernel_ms()
    ctx.close()
    return out, st, ms


@pytest.mark.parametrize("degree,tides", [(0, False), (8, False), (21, False), (8, True)])
@pytest.mark.parametrize("fixed", [True, False])
def test_quad_layout_is_bit_identical_to_the_d3_layout(degree, tides, fixed):
    opts = nx.IntegratorOptions.with_fixed_step_s(30.0) if fixed else nx.IntegratorOptions()
    prop, almanac, central = leo_full_setup(degree=degree, opts=opts, tides=tides)
    compiled = prop.compile(almanac, central, stm=True)
    b = stm_batch(37, seed=5 + degree)          # ragged: 37 = 2 quad workgroups + 5 trajectories, one D3 workgroup
    waves = 4 if degree else 3                  # the SAME column split in both layouts (the D3 kernel has at most 4 waves)
    d3, s3, ms3 = run(compiled, b, 3600 * S, 0, waves)
    qd, sq, msq = run(compiled, b, 3600 * S, 1, waves)
    assert (s3.status == 0).all() and (sq.status == 0).all()
    np.testing.assert_array_equal(qd.epoch_ns, d3.epoch_ns)
    np.testing.assert_array_equal(sq.n_accepted, s3.n_accepted)
    np.testing.assert_array_equal(sq.n_evals, s3.n_evals)
    np.testing.assert_array_equal(qd.rv(), d3.rv())
    np.testing.assert_array_equal(qd.stm, d3.stm)
    np.testing.assert_array_equal(qd.step_ns, d3.step_ns)
    print(f"deg {degree} tides {tides} fixed {fixed}: D3 {ms3:.2f} ms, quad {msq:.2f} ms (4 waves each)")


def test_quad_layout_full_width_vs_oracle():
    """16 column waves (what the default picks for 21x21): fixed 30 s steps => same step sequence as the oracle, Phi to
    1e-9 element-wise; 150 trajectories = 10 workgroups, the last one ragged."""
    prop, almanac, central = leo_full_setup(degree=21, opts=nx.IntegratorOptions.with_fixed_step_s(30.0))
    compiled = prop.compile(almanac, central, stm=True)
    b = stm_batch(150, seed=9)
    out, st, ms = run(compiled, b, 1800 * S, 1)
    ref, rst = oracle_lib.propagate(compiled, b, 1800 * S, n_threads=8)
    assert (st.status == 0).all() and (rst.status == 0).all()
    d = out.rv() - ref.rv()
    assert np.linalg.norm(d[:, :3], axis=1).max() < 1e-6 and np.linalg.norm(d[:, 3:], axis=1).max() < 1e-9
    a, r = out.stm.reshape(-1, 81), ref.stm.reshape(-1, 81)
    scale = np.maximum(np.abs(r), 1e-6 * np.abs(r).max(axis=1, keepdims=True))
    err = (np.abs(a - r) / scale).max()
    print(f"quad, 16 waves: Phi rel err {err:.2e}, kernel {ms:.2f} ms")
    assert err < 1e-9


def test_layout_is_chosen_by_ensemble_size_and_events_and_traj_survive_it():
    """Dense output and the per-trajectory duration array (what predict_until uses) through the quad layout."""
    prop, almanac, central = leo_full_setup(degree=8)
    compiled = prop.compile(almanac, central, stm=True)
    b = stm_batch(21, seed=3)
    ctx = nx.GpuContext(compiled)
    ctx.set_column_waves(4)   # (the same column split in both layouts => the same adaptive step sequences, bit for bit)
    out, st, traj = ctx.propagate_with_traj(b, 1800 * S, 64)
    ctx.set_stm_layout(0)
    out0, st0, traj0 = ctx.propagate_with_traj(b, 1800 * S, 64)
    ctx.close()
    np.testing.assert_array_equal(traj.len, traj0.len)
    np.testing.assert_array_equal(traj.epoch_ns, traj0.epoch_ns)
    np.testing.assert_array_equal(traj.state, traj0.state)
    np.testing.assert_array_equal(out.stm, out0.stm)
