"""N > 1 path on CPU: world_size-2 gloo run of the sharded Monte Carlo front end.  The propagation itself is
injected (the oracle) because there is no GPU here; what is under test is the sharding, the all-gather and the
index-stable ordering (reference mc/montecarlo.rs:208-273)."""
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
