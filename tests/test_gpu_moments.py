"""nyx_hip_ensemble_moments (moments_kernel.hip): count / sum(x - x0) / sum((x - x0)(x - x0)^T) of the final 9-vectors on the device
against numpy (mean, np.cov) to 1e-12, the status filter, the ragged and the empty case, run-to-run bit reproducibility, and
`Results.mean_and_covariance` going through it."""
import numpy as np
import pytest

import nyx_amd as nx
from nyx_amd import _abi
from scenarios import EPOCH0_NS, dispersed_leo_batch, leo_full_setup, leo_nominal

pytestmark = pytest.mark.gpu


def _ctx(degree=4):
    prop, almanac, central = leo_full_setup(degree=degree)
    return nx.GpuContext(prop.compile(almanac, central)), prop, almanac, central


def _x9(b):
    return np.concatenate([b.rv(), b.cr[:, None], b.cd[:, None], b.prop_mass_kg[:, None]], axis=1)


@pytest.mark.parametrize("n", [1, 63, 64, 5000, 70_001])
def test_moments_match_numpy(n):
    ctx, *_ = _ctx()
    b = dispersed_leo_batch(n, seed=n)
    rng = np.random.default_rng(n)
    b.cr[:] = 1.8 + 0.01 * rng.standard_normal(n)
    b.cd[:] = 2.2 + 0.01 * rng.standard_normal(n)
    b.prop_mass_kg[:] = 50.0 + rng.standard_normal(n)
    x = _x9(b)
    x0 = x[0].copy()
    mom = ctx.ensemble_moments(b, None, x0)
    assert mom[0] == n
    mean, cov = nx.moments_to_mean_cov(mom, x0)
    np.testing.assert_allclose(mean, x.mean(axis=0), rtol=1e-13, atol=0)
    if n > 1:
        ref = np.cov(x.T)
        scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
        assert np.max(np.abs(cov - ref) / scale) < 1e-12
        np.testing.assert_array_equal(cov, cov.T)
    else:
        assert np.isnan(cov).all()
    again = ctx.ensemble_moments(b, None, x0)
    np.testing.assert_array_equal(mom, again)                 # fixed grid, fixed order: the same bits every time
    ctx.close()


def test_status_filter_and_empty_ensemble():
    ctx, *_ = _ctx()
    n = 1000
    b = dispersed_leo_batch(n, seed=1)
    status = np.zeros(n, dtype=np.int32)
    status[::3] = _abi.ERR_NAN if hasattr(_abi, "ERR_NAN") else 5
    x = _x9(b)[status == 0]
    x0 = x.mean(axis=0)
    mean, cov = nx.moments_to_mean_cov(ctx.ensemble_moments(b, status, x0), x0)
    np.testing.assert_allclose(mean, x.mean(axis=0), rtol=1e-13)
    ref = np.cov(x.T)
    nz = np.diag(ref) > 1e-18      # (Cr, Cd, prop mass are constant in this batch: zero variance up to rounding, compared absolutely)
    sc = np.sqrt(np.outer(np.diag(ref)[nz], np.diag(ref)[nz]))
    assert np.max(np.abs(cov[np.ix_(nz, nz)] - ref[np.ix_(nz, nz)]) / sc) < 1e-12
    assert np.max(np.abs(cov[~nz][:, ~nz])) < 1e-20
    status[:] = 7
    mean, cov = nx.moments_to_mean_cov(ctx.ensemble_moments(b, status, x0), x0)
    assert np.isnan(mean).all() and np.isnan(cov).all()
    empty = _abi.StateBatch(0)
    assert ctx.ensemble_moments(empty, None, None)[0] == 0.0
    # without a reference point the sums are taken about the origin: same mean, a covariance that has lost ~8 digits to cancellation
    status[:] = 0
    mean0, cov0 = nx.moments_to_mean_cov(ctx.ensemble_moments(b, status, None), None)
    np.testing.assert_allclose(mean0, _x9(b).mean(axis=0), rtol=1e-12)
    ctx.close()


def test_monte_carlo_results_use_the_device_reduction():
    ctx, prop, almanac, central = _ctx()
    template = nx.Spacecraft(EPOCH0_NS, leo_nominal(), central, dry_mass_kg=100.0, srp_area_m2=1.0, cr=1.8)
    mc = nx.MonteCarlo(nx.MvnSpacecraft.from_sigmas(template, [1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3]), seed=3)
    res = mc.run_until_epoch(prop, almanac, EPOCH0_NS + 600 * nx.NS_PER_S, 257)
    assert hasattr(res._traj_ctx, "ensemble_moments")
    mean, cov = res.mean_and_covariance()
    x = np.array([np.concatenate([r.result.state.rv, [r.result.state.cr, r.result.state.cd, r.result.state.prop_mass_kg]]) for r in res.ok_runs()])
    np.testing.assert_allclose(mean, x.mean(axis=0), rtol=1e-13)
    ref = np.cov(x.T)
    nz = np.diag(ref) > 1e-18
    sc = np.sqrt(np.outer(np.diag(ref)[nz], np.diag(ref)[nz]))
    assert np.max(np.abs(cov[np.ix_(nz, nz)] - ref[np.ix_(nz, nz)]) / sc) < 1e-12
    ctx.close()
