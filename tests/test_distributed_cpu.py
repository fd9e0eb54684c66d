"""N > 1 path on CPU: world_size-2 and -3 gloo runs of the sharded Monte Carlo front end.  The propagation itself is
injected (the oracle) because there is no GPU here; what is under test is the sharding, the all-gather, the index-stable
ordering (reference mc/montecarlo.rs:208-273), the all-reduce of the ensemble moments and the gathered reports."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nyx_amd as nx
from scenarios import EPOCH0_NS, earth_frame, leo_full_setup, leo_nominal
from nyx_amd import ephem


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    import oracle_lib
    prop, almanac, central = leo_full_setup(degree=4)
    compiled = prop.compile(almanac, central)
    template = nx.Spacecraft(EPOCH0_NS, leo_nominal(), central, dry_mass_kg=100.0, srp_area_m2=1.0, cr=1.8)
    mvn = nx.MvnSpacecraft.from_sigmas(template, [1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3])

    def fn(batch, arg):
        if isinstance(arg, tuple):  # ("event", max_duration, event, trigger): run_until_nth_event
            out, st, _, _ = oracle_lib.propagate_until_event(compiled, batch, arg[1], arg[2], arg[3], capacity=512)
            return out, st
        return oracle_lib.propagate(compiled, batch, arg - int(batch.epoch_ns[0]))

    return prop, almanac, nx.MonteCarlo(mvn, seed=7, propagate_fn=fn)


class _OracleTrajEval:
    """Stands in for the GpuContext as the evaluator of dense output (anything with traj_at / traj_every)."""

    def traj_every(self, tb, step_ns, capacity):
        import oracle_lib
        return oracle_lib.traj_every(tb, step_ns, capacity)

    def traj_at(self, tb, epochs_ns):
        import oracle_lib
        return oracle_lib.traj_at(tb, epochs_ns)


def _build_with_traj(seed):
    import oracle_lib
    prop, almanac, mc = _build()
    compiled = prop.compile(almanac, earth_frame(ephem.MU_EARTH))

    def fn(batch, arg):
        out, st, traj = oracle_lib.propagate_with_traj(compiled, batch, arg - int(batch.epoch_ns[0]), 64)
        return out, st, traj, _OracleTrajEval()

    mc.propagate_fn = fn
    mc.seed = seed
    return prop, almanac, mc


def _worker3(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    end = EPOCH0_NS + 900 * nx.NS_PER_S
    # seed=None (the reference's OS entropy): the ranks must still agree on ONE stream
    prop, almanac, mc = _build_with_traj(None)
    res = mc.resume_run_until_epoch(prop, almanac, 0, end, 11, dist=dist)
    disp = np.array([r.dispersed_state.rv for r in res.runs])
    fin = np.array([r.result.rv for r in res.runs])
    mean, cov = res.mean_and_covariance()                       # all-reduce of the moments of the local shards
    rep = np.array(res.every_value_of(nx.StateParameter.X, 300 * nx.NS_PER_S))    # gathered report
    last = np.array(res.last_values_of(nx.StateParameter.Rmag))
    dv = np.array(res.dispersion_values_of(nx.StateParameter.VZ))
    np.savez(os.path.join(out_dir, f"w{rank}.npz"), disp=disp, fin=fin, mean=mean, cov=cov, rep=rep, last=last, dv=dv,
             n_local=len(res._local_runs()))
    dist.destroy_process_group()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prop, almanac, mc = _build()
    res = mc.resume_run_until_epoch(prop, almanac, 0, EPOCH0_NS + 600 * nx.NS_PER_S, 11, dist=dist)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([[r.index, *r.result.rv] for r in res.runs]))
    ev = mc.resume_run_until_nth_event(prop, almanac, 0, 2 * 3600 * nx.NS_PER_S, nx.Event(nx._abi.EV_Z_KM, 0.0, value_precision=1e-5), 2, 7,
                                       dist=dist)
    np.save(os.path.join(out_dir, f"e{rank}.npy"), np.array([[r.index, r.result.epoch_ns - EPOCH0_NS, r.result.rv[2]] for r in ev.runs]))
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 10_000):
        for world in (1, 2, 3, 8):
            b = [nx.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_two_rank_gloo_monte_carlo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    np.testing.assert_array_equal(r0, r1)               # every rank holds the complete ensemble
    assert list(r0[:, 0]) == list(range(11))            # sorted by run index
    prop, almanac, mc = _build()
    single = mc.run_until_epoch(prop, almanac, EPOCH0_NS + 600 * nx.NS_PER_S, 11)
    np.testing.assert_array_equal(r0[:, 1:], single.final_rv())  # sharding does not change any trajectory
    # integer-ns epochs survive the gather exactly (they exceed 2^53, so they travel as bit patterns)
    assert all(r.result.epoch_ns == EPOCH0_NS + 600 * nx.NS_PER_S for r in single.runs)
    # the event-terminated ensemble: sharded the same way, per-run event epochs survive the gather bit for bit
    e0, e1 = np.load(tmp_path / "e0.npy"), np.load(tmp_path / "e1.npy")
    np.testing.assert_array_equal(e0, e1)
    assert list(e0[:, 0]) == list(range(7)) and (np.abs(e0[:, 2]) < 1e-5).all() and len(set(e0[:, 1])) == 7
    # resume(skip) reproduces the tail of the stream (montecarlo.rs:208-224)
    tail = mc.resume_run_until_epoch(prop, almanac, 5, EPOCH0_NS + 600 * nx.NS_PER_S, 6)
    assert [r.index for r in tail.runs] == list(range(6))      # (enumerate() after skip(): indices restart, montecarlo.rs:290-295)
    np.testing.assert_array_equal(tail.final_rv(), single.final_rv()[5:])


def test_three_rank_ragged_shards_moments_and_reports(tmp_path):
    port = _free_port()
    mp.spawn(_worker3, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    w = [np.load(tmp_path / f"w{r}.npz") for r in range(3)]
    assert [int(x["n_local"]) for x in w] == [4, 4, 3]                       # ragged last shard
    for k in ("disp", "fin", "mean", "cov", "rep", "last", "dv"):
        np.testing.assert_array_equal(w[0][k], w[1][k])
        np.testing.assert_array_equal(w[0][k], w[2][k])
    # unseeded: one stream for all ranks, and results paired with the states they were propagated from
    fin, disp = w[0]["fin"], w[0]["disp"]
    assert np.linalg.norm(fin[:, :3] - disp[:, :3], axis=1).min() > 1000.0 and len(np.unique(disp[:, 0])) == 11
    # moments from the all-reduce == numpy on the gathered final states (Cr, Cd, prop mass are constant: zero variance)
    np.testing.assert_allclose(w[0]["mean"][:6], fin.mean(axis=0), rtol=1e-13)
    np.testing.assert_allclose(w[0]["cov"][:6, :6], np.cov(fin, rowvar=False), rtol=1e-9, atol=1e-12)
    assert np.all(w[0]["cov"][6:, :] == 0.0)
    # the gathered report == the single-process report of the same (seeded) ensemble: every run contributes 4 samples
    # (0, 300, 600, 900 s), in index order
    rep = w[0]["rep"]
    assert rep.shape == (44,)
    np.testing.assert_allclose(rep[0::4], disp[:, 0], rtol=0, atol=0)        # first sample = the dispersed X of each run
    np.testing.assert_allclose(rep[3::4], fin[:, 0], rtol=0, atol=1e-9)      # last sample = the final X
    np.testing.assert_allclose(w[0]["last"], np.linalg.norm(fin[:, :3], axis=1), rtol=1e-15)
    assert w[0]["dv"].shape == (11,)


def test_single_process_moments_match_numpy():
    prop, almanac, mc = _build()
    res = mc.run_until_epoch(prop, almanac, EPOCH0_NS + 600 * nx.NS_PER_S, 9, with_traj=False)
    mean, cov = res.mean_and_covariance()
    x = res.final_rv()
    np.testing.assert_allclose(mean[:6], x.mean(axis=0), rtol=1e-13)
    np.testing.assert_allclose(cov[:6, :6], np.cov(x, rowvar=False), rtol=1e-9, atol=1e-12)


def _worker_fail(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prop, almanac, mc = _build_with_traj(3)
    res = mc.resume_run_until_epoch(prop, almanac, 0, EPOCH0_NS + 600 * nx.NS_PER_S, 6, dist=dist)
    if rank == 1:   # this rank loses its trajectories: its part of the report cannot be made
        res._traj_batch = None
    outcome = "no error"
    try:
        res.every_value_of(nx.StateParameter.X, 300 * nx.NS_PER_S)
    except ValueError as e:
        outcome = "ValueError: " + str(e)
    except RuntimeError as e:
        outcome = "RuntimeError: " + str(e)
    # ... and the process group is still usable: nobody is stuck inside a half-entered collective
    mean, _ = res.mean_and_covariance()
    with open(os.path.join(out_dir, f"f{rank}.txt"), "w") as f:
        f.write(outcome + "\n" + repr(float(mean[0])))
    dist.destroy_process_group()


def test_a_failing_rank_aborts_the_collective_on_every_rank(tmp_path):
    """a rank that raises before a gathered report must not leave the others waiting in the collective."""
    port = _free_port()
    mp.spawn(_worker_fail, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f0, f1 = open(tmp_path / "f0.txt").read().split("\n"), open(tmp_path / "f1.txt").read().split("\n")
    assert f0[0].startswith("RuntimeError: another rank failed")
    assert f1[0].startswith("ValueError: these results carry no trajectories")
    assert f0[1] == f1[1]
