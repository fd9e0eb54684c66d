"""-m gpu: parity at BASELINE.json's FULL sizes.  Every configuration is run on the device at the size and duration
BASELINE.json quotes (config 5: 6 250 trajectories = one GPU's share of the 50 000, for the first 6 h of its 3 days), and a
sample of the ensemble is re-propagated by the oracle on all host threads: every sampled trajectory within the north-star
bar of 1 m / 1 mm/s.  Sampled indices are spread over the batch (first and last workgroup, ragged tail included)."""
import os
import time

import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
import scenarios as sc

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 1
S = nx.NS_PER_S


def spread(n, k):
    """k indices spread over [0, n): both ends and an even comb in between."""
    return np.unique(np.concatenate([np.arange(min(8, n)), np.arange(max(n - 8, 0), n), np.linspace(0, n - 1, k).astype(int)]))


def take(batch, idx):
    out = nx._abi.StateBatch(len(idx), batch.stm is not None)
    out.epoch_ns[:] = batch.epoch_ns[idx]
    for f in nx._abi.F64_FIELDS:
        getattr(out, f)[:] = getattr(batch, f)[idx]
    if batch.stm is not None:
        out.stm[:] = batch.stm[idx]
    return out


def device_vs_oracle_sample(prop, almanac, central, batch, dur_ns, k, label):
    compiled = prop.compile(almanac, central)
    ctx = nx.GpuContext(compiled)
    out, st = ctx.propagate(batch, dur_ns)
    ms = ctx.last_kernel_ms()
    helpers = ctx.last_coop_helpers()
    ctx.close()
    assert (st.status == 0).all()
    idx = spread(batch.n, k)
    t0 = time.time()
    ref, rst = oracle_lib.propagate(compiled, take(batch, idx), dur_ns, n_threads=NCPU)
    dt = time.time() - t0
    assert (rst.status == 0).all()
    d = out.rv()[idx] - ref.rv()
    dr, dv = np.linalg.norm(d[:, :3], axis=1), np.linalg.norm(d[:, 3:], axis=1)
    print(f"{label}: n = {batch.n}, kernel {ms:.1f} ms ({helpers} helper workgroups), oracle sample of {len(idx)} in {dt:.1f} s; "
          f"max |dr| {dr.max() * 1e3:.3e} m, max |dv| {dv.max() * 1e6:.3e} mm/s; steps gpu {int(st.n_accepted[idx].sum())} "
          f"cpu {int(rst.n_accepted.sum())}")
    assert (out.epoch_ns[idx] == ref.epoch_ns).all()
    assert dr.max() < 1e-3 and dv.max() < 1e-6, (dr.max(), dv.max())
    return out, st


def test_config2_full_day_10k():
    """configs[1]: 10 000 LEO trajectories, 70x70 + Sun/Moon + SRP, RK89 default options, 24 h (cooperative mode on)."""
    prop, almanac, central = sc.leo_full_setup(degree=70)
    device_vs_oracle_sample(prop, almanac, central, sc.dispersed_leo_batch(10_000, seed=0), 86400 * S, 240, "config 2")


def test_config3_jwst_5k_30_days():
    """configs[2]: 5 000 JWST states, Sun/Moon/Jupiter + SRP with two shadow bodies, 30 days."""
    prop, almanac, central = sc.jwst_setup()
    device_vs_oracle_sample(prop, almanac, central, sc.jwst_batch(5_000, seed=0), 30 * 86400 * S, 256, "config 3")


def test_config5_llo_150x150():
    """configs[4] (one GPU's 6 250 of the 50 000): 150x150 + Earth/Sun, DP78, the first 6 h."""
    prop, almanac, central = sc.lunar_setup(degree=150)
    device_vs_oracle_sample(prop, almanac, central, sc.lunar_batch(6_250, seed=0), 6 * 3600 * S, 256, "config 5")


def test_config5_llo_150x150_full_three_days():
    """configs[4] at its FULL length: 6 250 low-lunar-orbit states x 150x150 + Earth/Sun, DP78 default options, 72 h (~9 s of
    device time), every state of a 32-trajectory sample within 1 m / 1 mm/s of the oracle (~4 min of CPU on 32 threads)."""
    prop, almanac, central = sc.lunar_setup(degree=150)
    device_vs_oracle_sample(prop, almanac, central, sc.lunar_batch(6_250, seed=0), 72 * 3600 * S, 32, "config 5, 72 h")


def test_config4_geo_1k_sixty_updates():
    """configs[3]: 1 000 GEO states, 21x21 + Sun/Moon + SRP(Cr), STM, sixty 1-minute time updates; Phi-mapped covariance
    and nominal state vs the oracle's predict_until twin on a spread sample."""
    from bench import geo_batch, init_covar
    prop, almanac, central = sc.leo_full_setup(degree=21)
    compiled = prop.compile(almanac, central, stm=True)
    ctx = nx.GpuContext(compiled)
    n = 1_000
    b = geo_batch(n, seed=0)
    b.stm = np.zeros((n, 81))
    b.reset_stm()
    p0 = init_covar(n)
    end = int(b.epoch_ns[0]) + 3600 * S
    got = nx.predict_until(ctx, b, p0, end, 60 * S, history=60)
    assert (got.stats.status == 0).all() and (got.n_updates == 60).all()
    idx = spread(n, 48)
    t0 = time.time()
    ref = oracle_lib.predict_until(compiled, take(b, idx), p0[idx], end, 60 * S, history=60)
    dt = time.time() - t0
    d = got.states.rv()[idx] - ref.states.rv()
    dr, dv = np.linalg.norm(d[:, :3], axis=1).max(), np.linalg.norm(d[:, 3:], axis=1).max()

    def rel(a, r):
        scale = np.maximum(np.abs(r), 1e-6 * np.abs(r).max(axis=(-2, -1), keepdims=True))
        return float((np.abs(a - r) / scale).max())

    e_phi, e_p = rel(got.stm[:, idx], ref.stm), rel(got.covar_history[:, idx], ref.covar_history)
    print(f"config 4: n = {n}, device {got.kernel_ms:.1f} ms for 60 updates, oracle sample of {len(idx)} in {dt:.1f} s; dr {dr * 1e3:.2e} m "
          f"dv {dv * 1e6:.2e} mm/s, Phi {e_phi:.2e}, Pbar {e_p:.2e} (relative, element-wise, all 60 updates)")
    assert dr < 1e-3 and dv < 1e-6
    assert e_phi < 1e-9 and e_p < 1e-9   # "parity on Phi: relative 1e-9 element-wise vs oracle"
