"""Monte Carlo front end of the batched propagation path.

Mirror of the reference's `MonteCarlo` (nyx-core/src/mc/montecarlo.rs:44-296) at the level this path
needs it: states are generated on the host, one by one, from a single seeded stream
(`generate_states`, montecarlo.rs:277-296), run `index` keeps its dispersed state
(`Run{index, dispersed_state, result}`, mc/results.rs:48-59), `resume_run_until_epoch(skip, ..)`
regenerates the stream and skips the first `skip` samples (montecarlo.rs:208-224).  Where the reference
hands the states to a rayon `par_iter` (montecarlo.rs:233-253), this hands the whole batch to the GPU
through the C-ABI; with several ranks the ensemble is split into contiguous index shards and the
final states are collected with one all-gather (RCCL on GPUs, gloo in the CPU tests).

NOT restated: the reference's RNG (`Pcg64Mcg` + ziggurat `Normal`).  Dispersions here come from
`numpy.random.default_rng(seed)`; the dispersed STATES are the contract of the parity tests, not the
stream.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _abi
from .propagator import Almanac, Frame, Propagator, Spacecraft

# indices into the 9-vector [x, y, z, vx, vy, vz, Cr, Cd, prop mass] (cosmic/spacecraft.rs:451-473)
STATE_DIM = 9


@dataclass
class MvnSpacecraft:
    """Multivariate normal over the 9-state: x = L z + mean, z ~ N(0, I) (mc/multivariate.rs:298-330).
    `cov` is 9x9 (or 6x6 for an orbit-only dispersion); the factor is taken by SVD like the reference
    (`sqrt_s_v`, multivariate.rs:262-295)."""

    template: Spacecraft
    cov: np.ndarray
    mean: Optional[np.ndarray] = None

    def __post_init__(self):
        c = np.zeros((STATE_DIM, STATE_DIM))
        cov = np.asarray(self.cov, dtype=np.float64)
        c[: cov.shape[0], : cov.shape[1]] = cov
        u, s, _ = np.linalg.svd(c)
        self._sqrt_s_v = u @ np.diag(np.sqrt(s))
        m = np.zeros(STATE_DIM)
        if self.mean is not None:
            m[: len(self.mean)] = self.mean
        self._mean = m

    @classmethod
    def from_sigmas(cls, template: Spacecraft, sigmas: Sequence[float]):
        s = np.zeros(STATE_DIM)
        s[: len(sigmas)] = sigmas
        return cls(template, np.diag(s ** 2))

    def sample_vector(self, rng: np.random.Generator) -> np.ndarray:
        return self._sqrt_s_v @ rng.standard_normal(STATE_DIM) + self._mean


@dataclass
class Run:
    index: int
    dispersed_state: Spacecraft
    result: object  # Spacecraft or PropagationError


@dataclass
class Results:
    runs: List[Run]
    scenario: str

    def final_rv(self) -> np.ndarray:
        return np.array([r.result.rv for r in self.runs if isinstance(r.result, Spacecraft)])

    def mean_and_covariance(self):
        x = self.final_rv()
        return x.mean(axis=0), np.cov(x, rowvar=False)


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous index shard [lo, hi) of rank `rank` (index-stable => resume/skip semantics carry over)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class MonteCarlo:
    """montecarlo.rs:44-75."""

    random_state: MvnSpacecraft
    seed: Optional[int] = None
    scenario: str = "MonteCarlo"
    # injectable for the CPU tests (gloo): (batch, end_epoch_ns) -> (out batch, stats); default = the GPU context
    propagate_fn: Optional[Callable] = field(default=None, repr=False)

    def generate_states(self, skip: int, num_runs: int, seed: Optional[int] = None):
        """montecarlo.rs:277-296: one stream, samples drawn sequentially, first `skip` discarded."""
        rng = np.random.default_rng(self.seed if seed is None else seed)
        t = self.random_state.template
        base = np.concatenate([np.asarray(t.rv, dtype=np.float64), [t.cr, t.cd, t.prop_mass_kg]])
        out = []
        for index in range(skip + num_runs):
            v = self.random_state.sample_vector(rng)
            if index < skip:
                continue
            x = base + v
            s = Spacecraft(**{**t.__dict__})
            s.rv, s.cr, s.cd, s.prop_mass_kg = x[:6].copy(), float(x[6]), float(x[7]), float(x[8])
            out.append((index, s))
        return out

    def run_until_epoch(self, prop: Propagator, almanac: Almanac, end_epoch_ns: int, num_runs: int) -> Results:
        return self.resume_run_until_epoch(prop, almanac, 0, end_epoch_ns, num_runs)

    def resume_run_until_epoch(self, prop: Propagator, almanac: Almanac, skip: int, end_epoch_ns: int, num_runs: int,
                               dist=None) -> Results:
        """montecarlo.rs:208-273.  With `dist` (an initialised torch.distributed module) the runs are sharded by
        contiguous index range over the ranks and every rank returns the complete, index-sorted Results."""

        def run(ctx, batch):
            return ctx.propagate_until_epoch(batch, int(end_epoch_ns))

        return self._run(prop, almanac, skip, num_runs, dist, run, int(end_epoch_ns))

    def run_until_nth_event(self, prop: Propagator, almanac: Almanac, max_duration_ns: int, event, trigger: int, num_runs: int) -> Results:
        """montecarlo.rs:93-110."""
        return self.resume_run_until_nth_event(prop, almanac, 0, max_duration_ns, event, trigger, num_runs)

    def resume_run_until_nth_event(self, prop: Propagator, almanac: Almanac, skip: int, max_duration_ns: int, event, trigger: int,
                                   num_runs: int, dist=None, capacity: int = 4096) -> Results:
        """montecarlo.rs:115-186: every run stops at the `trigger`-th occurrence of `event` (or fails with NthEventError)."""

        def run(ctx, batch):
            out, st, _, _ = ctx.propagate_until_event(batch, int(max_duration_ns), event, trigger, capacity)
            return out, st

        return self._run(prop, almanac, skip, num_runs, dist, run, ("event", int(max_duration_ns), event, trigger))

    def _run(self, prop, almanac, skip, num_runs, dist, run, fn_arg) -> Results:
        from .propagator import PropagationError, pack_spacecraft

        states = self.generate_states(skip, num_runs, self.seed)
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
        lo, hi = shard_bounds(len(states), rank, world)
        mine = states[lo:hi]
        batch = pack_spacecraft([s for _, s in mine], False)
        if self.propagate_fn is not None:
            out, st = self.propagate_fn(batch, fn_arg)
        else:
            out, st = run(prop._context(almanac, self.random_state.template.frame, False), batch)
        payload = np.concatenate([out.rv(), out.cr[:, None], out.cd[:, None], out.prop_mass_kg[:, None],
                                  np.ascontiguousarray(out.epoch_ns, dtype=np.int64).view(np.float64)[:, None],  # bit pattern: ns past J2000 exceed 2^53
                                  st.status[:, None].astype(np.float64)], axis=1)
        if dist is not None and world > 1:
            payload = all_gather_rows(dist, payload, [shard_bounds(len(states), r, world) for r in range(world)])
        runs = []
        for k, (index, s) in enumerate(states):
            row = payload[k]
            status = int(row[10])
            if status == _abi.OK:
                r = Spacecraft(**{**s.__dict__})
                r.rv, r.cr, r.cd, r.prop_mass_kg, r.epoch_ns = row[:6].copy(), float(row[6]), float(row[7]), float(row[8]), int(row[9:10].view(np.int64)[0])
                runs.append(Run(index, s, r))
            else:
                runs.append(Run(index, s, PropagationError(status, index)))
        runs.sort(key=lambda r: r.index)  # par_sort_by_key(index), montecarlo.rs:267
        return Results(runs, self.scenario)


def all_gather_rows(dist, local: np.ndarray, bounds) -> np.ndarray:
    """One all-gather of the per-rank result rows (ragged shards are padded to the largest)."""
    import torch

    world = dist.get_world_size()
    width = local.shape[1]
    nmax = max(hi - lo for lo, hi in bounds)
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros((nmax, width), dtype=torch.float64, device=dev)
    buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return np.concatenate([parts[r][: hi - lo].cpu().numpy() for r, (lo, hi) in enumerate(bounds)], axis=0)
