"""-m gpu: the host-side mirror of the reference interface (Propagator / PropInstance / many_for_duration / MonteCarlo)
driving the HIP path.  These read like the reference's own tests."""
import numpy as np
import pytest

import nyx_amd as nx
import oracle_lib
from scenarios import EPOCH0_NS, GOLDEN, dispersed_leo_batch, earth_frame, leo_full_setup, leo_nominal

pytestmark = pytest.mark.gpu
DAY = 86400 * nx.NS_PER_S


def test_val_two_body_dynamics_through_prop_instance():
    # reference: tests/mission_design/orbitaldyn.rs:102-171 (val_two_body_dynamics), same sequence of calls
    eme2k = earth_frame(GOLDEN["mu_pck"])
    state = nx.Spacecraft(0, GOLDEN["initial_state"], eme2k)
    dynamics = nx.SpacecraftDynamics.new(nx.OrbitalDynamics.two_body())
    setup = nx.Propagator.rk89(dynamics, nx.IntegratorOptions())
    prop = setup.with_(state, nx.Almanac())
    prop.for_duration(DAY)
    rslt = np.array(GOLDEN["rk89_default_options"]["state"])
    assert np.max(np.abs(prop.state.rv - rslt)) < 2e-9, "two body prop failed"
    assert prop.state.epoch_ns == DAY
    prop.for_duration(-DAY)  # back-propagation on the same instance (carries the adapted step size)
    d = prop.state.rv - np.array(GOLDEN["initial_state"])
    assert np.linalg.norm(d[:3]) < 1e-5 and np.linalg.norm(d[3:]) < 1e-8 and prop.state.epoch_ns == 0
    prop.for_duration(DAY)  # forward again: repeated calls work
    d = prop.state.rv - rslt
    assert prop.state.epoch_ns == DAY and np.linalg.norm(d[:3]) < 1e-5 and np.linalg.norm(d[3:]) < 1e-8
    det = prop.latest_details()
    assert det["attempts"] >= 1 and det["error"] >= 0.0


def test_many_for_duration_drops_failed_runs_like_the_reference():
    # reference: nyx-py/src/py_md.rs:275-321 — failed runs are dropped from the returned list
    prop, almanac, central = leo_full_setup(degree=4)
    scs = [nx.Spacecraft(EPOCH0_NS, leo_nominal() + i * 1e-3, central, dry_mass_kg=100.0, srp_area_m2=1.0) for i in range(5)]
    scs[2].dry_mass_kg = 0.0  # massless with a force model -> that run errors
    finals = prop.many_for_duration(scs, almanac, 600 * nx.NS_PER_S)
    assert len(finals) == 4 and all(f.epoch_ns == EPOCH0_NS + 600 * nx.NS_PER_S for f in finals)
    kept = prop.many_for_duration(scs, almanac, 600 * nx.NS_PER_S, drop_failed=False)
    assert len(kept) == 5 and isinstance(kept[2], nx.PropagationError) and kept[2].index == 2
    until = prop.many_until_epoch(scs, almanac, EPOCH0_NS + 900 * nx.NS_PER_S)
    assert len(until) == 4 and all(f.epoch_ns == EPOCH0_NS + 900 * nx.NS_PER_S for f in until)


def test_monte_carlo_run_until_epoch_on_gpu():
    # reference shape: tests/monte_carlo/framework.rs:22-95 (MonteCarlo::run_until_epoch, 10 runs) + resume
    prop, almanac, central = leo_full_setup(degree=8)
    template = nx.Spacecraft(EPOCH0_NS, leo_nominal(), central, dry_mass_kg=100.0, srp_area_m2=1.0, cr=1.8)
    mc = nx.MonteCarlo(nx.MvnSpacecraft.from_sigmas(template, [1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3]), seed=0, scenario="demo")
    end = EPOCH0_NS + 3600 * nx.NS_PER_S
    rslts = mc.run_until_epoch(prop, almanac, end, 10)
    assert [r.index for r in rslts.runs] == list(range(10)) and all(r.result.epoch_ns == end for r in rslts.runs)
    mean, cov = rslts.mean_and_covariance()
    assert mean.shape == (6,) and cov.shape == (6, 6) and np.all(np.linalg.eigvalsh(cov) > -1e-12)
    # the GPU ensemble equals the oracle's run on the same dispersed states
    from nyx_amd.propagator import pack_spacecraft
    batch = pack_spacecraft([r.dispersed_state for r in rslts.runs], False)
    ref, _ = oracle_lib.propagate(prop.compile(almanac, central), batch, 3600 * nx.NS_PER_S)
    d = rslts.final_rv() - ref.rv()
    assert np.linalg.norm(d[:, :3], axis=1).max() < 1e-3 and np.linalg.norm(d[:, 3:], axis=1).max() < 1e-6
    # resume_run_until_epoch(skip) reproduces the tail (montecarlo.rs:208-224)
    tail = mc.resume_run_until_epoch(prop, almanac, 6, end, 4)
    np.testing.assert_array_equal(tail.final_rv(), rslts.final_rv()[6:])
