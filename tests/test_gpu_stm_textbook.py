"""-m gpu: NYX_HIP_FLAG_STM_TEXTBOOK on the device (the 64-lane dual kernel replays the tableau over the stage matrices of the
accepted attempt, stm_update_textbook) against the oracle twin (the 90-vector integrated as one, oracle/nyx_oracle.c): same
operations in the same order - the states stay bit-identical to the reference form's, Phi agrees to rounding."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from scenarios import GOLDEN, dispersed_leo_batch, leo_full_setup, pos_vel_errors, two_body_setup

pytestmark = pytest.mark.gpu
S = nx.NS_PER_S


def _rel(got, want):
    scale = np.maximum(np.abs(want), 1e-6 * np.abs(want).max(axis=1, keepdims=True))
    return float(np.max(np.abs(got - want) / scale))


def test_two_body_textbook_stm_matches_the_oracle():
    g = GOLDEN["two_body_dual"]
    prop, almanac, central = two_body_setup(nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions.with_fixed_step_s(10.0), GOLDEN["mu_pck"])
    c_tb = prop.compile(almanac, central, stm=True, stm_textbook=True)
    c_ref = prop.compile(almanac, central, stm=True)
    b = nx._abi.StateBatch(70, with_stm=True)          # two workgroups of the 64-lane layout, the second ragged
    b.set_rv(np.tile(np.array(g["state"]), (70, 1)) + np.random.default_rng(3).standard_normal((70, 6)) * np.array([1, 1, 1, 1e-3, 1e-3, 1e-3]))
    b.reset_stm()
    ctx = nx.GpuContext(c_tb)
    out, st = ctx.propagate(b, 600 * S)
    ref, rst = oracle_lib.propagate(c_tb, b, 600 * S, n_threads=8)
    assert (st.status == 0).all() and (rst.status == 0).all()
    np.testing.assert_array_equal(out.rv(), ref.rv())
    e = _rel(out.stm, ref.stm)
    print(f"two-body, RK89 fixed 10 s, 600 s, textbook form: device vs oracle Phi {e:.2e} (relative, element-wise)")
    assert e < 1e-13
    # the state does not depend on the form; Phi does
    ctx2 = nx.GpuContext(c_ref)
    out2, _ = ctx2.propagate(b, 600 * S)
    np.testing.assert_array_equal(out.rv(), out2.rv())
    assert _rel(out.stm, out2.stm) > 1e-3
    ctx.close()
    ctx2.close()


@pytest.mark.parametrize("fixed", [True, False])
def test_full_model_textbook_stm_vs_oracle(fixed):
    """21x21 + Sun/Moon + SRP (Cr estimated), RK89, 30 minutes, LEO and GEO states: Phi element-wise 1e-9 wherever device and oracle
    walk the same steps (fixed steps: everywhere)."""
    opts = nx.IntegratorOptions.with_fixed_step_s(30.0) if fixed else nx.IntegratorOptions()
    prop, almanac, central = leo_full_setup(degree=21, opts=opts)
    compiled = prop.compile(almanac, central, stm=True, stm_textbook=True)
    n = 9
    b = dispersed_leo_batch(n, seed=42)
    b.stm = np.zeros((n, 81))
    b.reset_stm()
    from scenarios import keplerian_to_cartesian
    from nyx_amd import ephem
    geo = keplerian_to_cartesian(42164.0, 1e-5, 0.0, 163.0, 75.0, 0.0, ephem.MU_EARTH)
    rv = b.rv()
    rv[5:] = geo[None, :] + (rv[5:] - rv[5:].mean(axis=0))
    b.set_rv(rv)
    ctx = nx.GpuContext(compiled)
    out, st = ctx.propagate(b, 1800 * S)
    ref, rst = oracle_lib.propagate(compiled, b, 1800 * S, n_threads=8)
    assert (st.status == 0).all() and (rst.status == 0).all()
    dr, dv = pos_vel_errors(out, ref)
    scale = np.maximum(np.abs(ref.stm), 1e-6 * np.abs(ref.stm).max(axis=1, keepdims=True))
    e = (np.abs(out.stm - ref.stm) / scale).max(axis=1)
    same = (out.step_ns == ref.step_ns) & (st.n_accepted == rst.n_accepted) & (st.n_rejected == rst.n_rejected)
    print(f"textbook STM, 21x21 full model, fixed={fixed}: dr {dr.max() * 1e3:.2e} m; Phi same-steps {e[same].max() if same.any() else 0.0:.2e} "
          f"(n = {int(same.sum())}), others {e[~same].max() if (~same).any() else 0.0:.2e}")
    assert dr.max() < 1e-3 and dv.max() < 1e-6
    if fixed:
        assert same.all()
    assert same.any() and e[same].max() < 1e-9
    assert e.max() < 1e-2
    assert np.abs(out.stm[5:, 9 * 6 + 3:9 * 6 + 6]).max() > 0.0   # d v / d Cr is populated in the textbook form too
    ctx.close()


def test_textbook_form_through_the_covariance_mapping_loop():
    """nyx_hip_predict_until on a textbook context (one-minute segments, Phi reset) against the oracle's twin."""
    from bench import geo_batch, init_covar
    prop, almanac, central = leo_full_setup(degree=8)
    compiled = prop.compile(almanac, central, stm=True, stm_textbook=True)
    n = 40
    b = geo_batch(n, seed=1)
    b.stm = np.zeros((n, 81))
    b.reset_stm()
    p0 = init_covar(n)
    end = int(b.epoch_ns[0]) + 600 * S
    ctx = nx.GpuContext(compiled)
    got = nx.predict_until(ctx, b, p0, end, 60 * S, history=10)
    ref = oracle_lib.predict_until(compiled, b, p0, end, 60 * S, history=10)
    assert (got.stats.status == 0).all() and (got.n_updates == 10).all()

    def rel(a, r):
        scale = np.maximum(np.abs(r), 1e-6 * np.abs(r).max(axis=(-2, -1), keepdims=True))
        return float((np.abs(a - r) / scale).max())
    e_phi, e_p = rel(got.stm, ref.stm), rel(got.covar_history, ref.covar_history)
    print(f"textbook form, predict_until, 10 updates: Phi {e_phi:.2e}, Pbar {e_p:.2e}")
    assert e_phi < 1e-9 and e_p < 1e-9
    ctx.close()
