"""Pins the CPU oracle against the reference's own golden vectors."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from scenarios import GOLDEN, leo_batch, two_body_setup

DAY_NS = 86400 * nx.NS_PER_S


def run_oracle(method, opts, mu, duration_ns=DAY_NS, batch=None):
    prop, almanac, central = two_body_setup(method, opts, mu)
    compiled = prop.compile(almanac, central)
    batch = batch if batch is not None else leo_batch(1)
    out, st = oracle_lib.propagate(compiled, batch, duration_ns)
    assert st.status[0] == 0
    return out, st


@pytest.mark.parametrize("name", ["RungeKutta4", "Verner56", "DormandPrince45", "DormandPrince78", "RungeKutta89"])
def test_fixed_step_bit_exact(name):
    # reference: tests/propagation/propagators.rs:306-472, assert_eq! on the 6-vector
    g = GOLDEN["fixed_step"][name]
    opts = nx.IntegratorOptions.with_fixed_step_s(g["step_s"])
    out, st = run_oracle(nx.IntegratorMethod[name], opts, GOLDEN["mu_gmat"])
    assert out.epoch_ns[0] == DAY_NS
    np.testing.assert_array_equal(out.rv()[0], np.array(g["state"]))
    stages = {"RungeKutta4": 4, "Verner56": 8, "DormandPrince45": 7, "DormandPrince78": 13, "RungeKutta89": 16}[name]
    n_steps = int(86400 / g["step_s"])
    assert st.n_accepted[0] == n_steps and st.n_evals[0] == n_steps * stages


@pytest.mark.parametrize("name", ["DormandPrince78", "RungeKutta89", "DormandPrince45", "Verner56", "CashKarp45"])
def test_adaptive(name):
    # reference: tests/propagation/propagators.rs:85-302 (assert_eq! for DP78 / RK89, tolerances otherwise) and :26-81
    a = GOLDEN["adaptive"]
    g = a[name]
    opts = nx.IntegratorOptions.with_adaptive_step_s(a["min_step_s"], a["max_step_s"], a["tolerance"], nx.ErrorControl.RSSCartesianState)
    out, st = run_oracle(nx.IntegratorMethod[name], opts, GOLDEN["mu_gmat"])
    assert out.epoch_ns[0] == DAY_NS
    got, want = out.rv()[0], np.array(g["state"])
    if g["tol"] == 0.0:
        np.testing.assert_array_equal(got, want)
    else:
        assert np.max(np.abs(got - want)) < g["tol"]
    # reference: if error > accuracy the last step must be at the minimum
    if st.last_error[0] > a["tolerance"]:
        assert st.last_step_ns[0] == nx.seconds(a["min_step_s"])


def test_rk89_default_options_and_backprop():
    # reference: tests/mission_design/orbitaldyn.rs:102-171
    g = GOLDEN["rk89_default_options"]
    out, st = run_oracle(nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions(), GOLDEN["mu_pck"])
    err = np.max(np.abs(out.rv()[0] - np.array(g["state"])))
    assert err < g["tol"], err
    assert st.n_rejected[0] == 0 and 1000 < st.n_accepted[0] < 1100
    # back-propagation returns to the initial state (err_r < 1e-5 km, err_v < 1e-8 km/s)
    back, _ = run_oracle(nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions(), GOLDEN["mu_pck"], -DAY_NS, batch=out)
    assert back.epoch_ns[0] == 0
    d = back.rv()[0] - np.array(GOLDEN["initial_state"])
    assert np.linalg.norm(d[:3]) < 1e-5 and np.linalg.norm(d[3:]) < 1e-8
    fwd, _ = run_oracle(nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions(), GOLDEN["mu_pck"], DAY_NS, batch=back)
    d = fwd.rv()[0] - np.array(g["state"])
    assert np.linalg.norm(d[:3]) < 1e-5 and np.linalg.norm(d[3:]) < 1e-8


def test_two_body_dual():
    # reference: tests/mission_design/orbitaldyn.rs:671-743
    g = GOLDEN["two_body_dual"]
    prop, almanac, central = two_body_setup(nx.IntegratorMethod.RungeKutta89, nx.IntegratorOptions(), GOLDEN["mu_pck"])
    compiled = prop.compile(almanac, central, stm=True)
    y9 = np.array(g["state"] + [0.0, 0.0, 0.0])
    st, fx, grad = oracle_lib.dual_eom(compiled, 0, y9)
    assert st == 0
    assert np.linalg.norm(fx[:6] - np.array(g["fx"])) < 1e-16
    expected = np.zeros((9, 9))
    for k, v in g["grad"].items():
        i, j = (int(x) for x in k.split(","))
        expected[i, j] = v
    assert np.linalg.norm(grad - expected) < 1e-16
    # real and dual paths agree (orbitaldyn.rs:706-709)
    y90 = np.concatenate([y9, np.eye(9).reshape(-1)])
    st, dy = oracle_lib.eom(compiled, 0, 0.0, y90, ctx_stm=np.eye(9).reshape(-1))
    assert st == 0 and np.linalg.norm(dy[:6] - fx[:6]) < 1e-16
