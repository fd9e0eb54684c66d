"""Monte Carlo front end of the batched propagation path.

Mirror of the reference's `MonteCarlo` (nyx-core/src/mc/montecarlo.rs:44-296) at the level this path
needs it: states are generated on the host, one by one, from a single seeded stream
(`generate_states`, montecarlo.rs:277-296; the index is the position AFTER the skip, as there), run `index` keeps its dispersed state
(`Run{index, dispersed_state, result}`, mc/results.rs:48-59), `resume_run_until_epoch(skip, ..)`
regenerates the stream and skips the first `skip` samples (montecarlo.rs:208-224).  Where the reference
hands the states to a rayon `par_iter` (montecarlo.rs:233-253), this hands the whole batch to the GPU
through the C-ABI; with several ranks the ensemble is split into contiguous index shards and the
final states are collected with one all-gather (RCCL on GPUs, gloo in the CPU tests).

The random stream is the reference's (`rand_pcg::Pcg64Mcg::new(seed)` + ziggurat `Normal(0, 1)`, nine draws per
state: nyx_amd/rng.py, pinned by the reference's seeded known-answer tests).  What is NOT pinned is the orientation of
the covariance factor: the reference takes `V sqrt(S)` from nalgebra's `svd_unordered`, whose column order and signs for
repeated or zero singular values are nalgebra's own; here a diagonal covariance maps component k to draw k (which is what
the reference's seeded tests show for diagonal cases) and a general one goes through LAPACK's SVD - the same
distribution, possibly another labelling of the draws.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import os

import numpy as np

from . import _abi
from .rng import Pcg64Mcg
from .params import StateError, StateParameter, state_value
from .propagator import Almanac, Propagator, Spacecraft, Traj

# indices into the 9-vector [x, y, z, vx, vy, vz, Cr, Cd, prop mass] (cosmic/spacecraft.rs:451-473)
STATE_DIM = 9


@dataclass
class StateDispersion:
    """mc/dispersion.rs:29-48."""

    param: StateParameter
    mean: Optional[float] = None
    std_dev: Optional[float] = None

    @classmethod
    def zero_mean(cls, param: StateParameter, std_dev: float):
        return cls(param, 0.0, std_dev)


_ORBITAL = {StateParameter.X, StateParameter.Y, StateParameter.Z, StateParameter.VX, StateParameter.VY, StateParameter.VZ,
            StateParameter.Rmag, StateParameter.Vmag, StateParameter.Hmag, StateParameter.Energy, StateParameter.SemiMajorAxis,
            StateParameter.Eccentricity, StateParameter.Inclination, StateParameter.RAAN, StateParameter.AoP,
            StateParameter.TrueAnomaly, StateParameter.Period, StateParameter.ApoapsisRadius, StateParameter.PeriapsisRadius}


def _partials(param: StateParameter, rv: np.ndarray, mu: float) -> np.ndarray:
    """d param / d (x, y, z, vx, vy, vz) at `rv`.  The reference reads them off ANISE's `OrbitGrad` (hyperdual numbers,
    absent here): exact for the Cartesian components and the magnitudes, central differences for the Keplerian elements."""
    cart = [StateParameter.X, StateParameter.Y, StateParameter.Z, StateParameter.VX, StateParameter.VY, StateParameter.VZ]
    g = np.zeros(6)
    if param in cart:
        g[cart.index(param)] = 1.0
    elif param is StateParameter.Rmag:
        g[:3] = rv[:3] / np.linalg.norm(rv[:3])
    elif param is StateParameter.Vmag:
        g[3:] = rv[3:] / np.linalg.norm(rv[3:])
    else:
        for k in range(6):
            h = 1e-6 * (np.linalg.norm(rv[:3]) if k < 3 else np.linalg.norm(rv[3:]))
            up, dn = rv.copy(), rv.copy()
            up[k] += h
            dn[k] -= h
            d = state_value(param, up, mu) - state_value(param, dn, mu)
            if param in (StateParameter.RAAN, StateParameter.AoP, StateParameter.TrueAnomaly):
                d = (d + 180.0) % 360.0 - 180.0
            g[k] = d / (2.0 * h)
    return g


class MvnSpacecraft:
    """Multivariate normal over the 9-state: x = L z + mean, z ~ N(0, I) (mc/multivariate.rs:63-330); L = V sqrt(S) from the
    SVD of the 9x9 covariance (`sqrt_s_v`, multivariate.rs:203-218, 262-272).  Built from a covariance (`from_spacecraft_cov`,
    the constructor; `from_sigmas` for a diagonal one) or from dispersions of state parameters (`new` / `zero_mean`)."""

    def __init__(self, template: Spacecraft, cov: np.ndarray, mean: Optional[np.ndarray] = None, dispersions=None):
        self.template = template
        c = np.zeros((STATE_DIM, STATE_DIM))
        cov = np.asarray(cov, dtype=np.float64)
        c[: cov.shape[0], : cov.shape[1]] = cov
        if np.any(np.linalg.eigvalsh(0.5 * (c + c.T)) < -1e-12 * max(1.0, float(np.abs(c).max()))):
            raise ValueError("covariance matrix is not positive semi-definite")   # NyxError::CovarianceMatrixNotPsd
        self.cov = c
        if np.count_nonzero(c - np.diag(np.diagonal(c))) == 0:
            self._sqrt_s_v = np.diag(np.sqrt(np.diagonal(c)))   # V = I: component k takes draw k
        else:
            _, sv, vt = np.linalg.svd(c)
            self._sqrt_s_v = vt.T * np.sqrt(sv)[None, :]
        m = np.zeros(STATE_DIM)
        if mean is not None:
            m[: len(mean)] = mean
        self._mean = m
        self.mean = m
        if dispersions is None:   # from_spacecraft_cov lists the nine components (multivariate.rs:274-312: std_dev <- the VARIANCE)
            dispersions = [StateDispersion(p, None, float(c[k, k])) for k, p in enumerate(_VECTOR_PARAMS)]
        self.dispersions = list(dispersions)

    from_spacecraft_cov = classmethod(lambda cls, template, cov, mean=None: cls(template, cov, mean))

    @classmethod
    def from_sigmas(cls, template: Spacecraft, sigmas: Sequence[float]):
        s = np.zeros(STATE_DIM)
        s[: len(sigmas)] = sigmas
        return cls(template, np.diag(s ** 2))

    @classmethod
    def new(cls, template: Spacecraft, dispersions: Sequence[StateDispersion]):
        """multivariate.rs:78-226: the dispersions of orbital parameters are independent in THEIR space; the Jacobian J of the
        parameters w.r.t. the Cartesian state maps them back, C = J+ P J+^T with J+ the pseudo-inverse, mean = J+ means."""
        dispersions = list(dispersions)
        cov = np.zeros((STATE_DIM, STATE_DIM))
        mean = np.zeros(STATE_DIM)
        orbital = [d for d in dispersions if d.param in _ORBITAL]
        if orbital:
            rv = np.asarray(template.rv, dtype=np.float64)
            jac = np.stack([_partials(d.param, rv, float(template.frame.mu_km3_s2)) for d in orbital])
            p = np.diag([(d.std_dev or 0.0) ** 2 for d in orbital])
            means = np.array([d.mean or 0.0 for d in orbital])
            jac_inv = np.linalg.pinv(jac)
            cov[:6, :6] = jac_inv @ p @ jac_inv.T
            mean[:6] = jac_inv @ means
        for d in dispersions:
            if d.param in _ORBITAL:
                continue
            # as written in the reference (multivariate.rs:183-197): the variance is taken from `mean`, Cr lands in slot 7 and
            # Cd in slot 8 of [.., Cr, Cd, prop mass]; the mass branch indexes (9, 9) of a 9x9 matrix and panics
            if d.param is StateParameter.Cr:
                cov[7, 7] = (d.mean or 0.0) ** 2
            elif d.param is StateParameter.Cd:
                cov[8, 8] = (d.mean or 0.0) ** 2
            elif d.param in (StateParameter.DryMass, StateParameter.PropMass):
                raise IndexError("Matrix index out of bounds (multivariate.rs:193: cov[(9, 9)] of a 9x9)")
            else:
                raise StateError(d.param)   # StateError::ReadOnly
        return cls(template, cov, mean, dispersions)

    @classmethod
    def zero_mean(cls, template: Spacecraft, dispersions: Sequence[StateDispersion]):
        return cls.new(template, [StateDispersion(d.param, 0.0, d.std_dev) for d in dispersions])

    def sample_vector(self, rng) -> np.ndarray:
        """multivariate.rs:300-303: nine standard normals, in order, from the run's stream (`rng`: nyx_amd.rng.Pcg64Mcg, or a
        numpy Generator)."""
        z = np.array(rng.normal_vector(STATE_DIM)) if hasattr(rng, "normal_vector") else rng.standard_normal(STATE_DIM)
        return self._sqrt_s_v @ z + self._mean

    def disperses(self, k: int) -> bool:
        """Is component k of the 9-vector dispersed (non-zero variance or mean)?"""
        return bool(np.any(self._sqrt_s_v[k] != 0.0) or self._mean[k] != 0.0)


# the dispersed components of the 9-vector, as state parameters (multivariate.rs:298-330)
_VECTOR_PARAMS = [StateParameter.X, StateParameter.Y, StateParameter.Z, StateParameter.VX, StateParameter.VY, StateParameter.VZ,
                  StateParameter.Cr, StateParameter.Cd, StateParameter.PropMass]


class DispersedState(Spacecraft):
    """generator.rs:59-67: the dispersed state plus `actual_dispersions`, the list of (parameter, delta).  The delta is
    `template.value(param) - state.value(param)` - template MINUS dispersed, as the reference computes it
    (multivariate.rs:320-325).  Subclass of Spacecraft so that it packs like one (`.state` returns itself)."""

    actual_dispersions: list = []

    @property
    def state(self) -> "Spacecraft":
        return self


class PropResult:
    """results.rs:73-80: the final state of a run and its trajectory.  `traj` is None when the run was made without
    dense output (`with_traj=False`) or on another rank of a sharded ensemble.  Attribute access falls through to the
    state, so `run.result.rv` and `run.result.state.rv` are the same thing."""

    def __init__(self, state: Spacecraft, traj_src=None):
        self.state = state
        self._traj_src = traj_src   # (context, TrajBatch, row) - the Traj object is built on first use
        self._traj = None

    @property
    def traj(self) -> Optional[Traj]:
        if self._traj is None and self._traj_src is not None:
            ctx, batch, row = self._traj_src
            self._traj = Traj(ctx, batch, row)
        return self._traj

    def __getattr__(self, name):
        if name.startswith("_") or name in ("state", "traj"):
            raise AttributeError(name)
        return getattr(self.state, name)


@dataclass
class Run:
    index: int
    dispersed_state: DispersedState
    result: object  # PropResult or PropagationError


@dataclass
class Results:
    """mc/results.rs:60-245.  Runs are sorted by index.  When the runs of this process carry trajectories they share one
    dense-output batch, and the `every_value_of*` reports resample ALL of them with one launch of the trajectory kernel
    (`Traj::every` per run in the reference, results.rs:134-160)."""

    runs: List[Run]
    scenario: str
    mu_km3_s2: float = 0.0
    _traj_ctx: object = field(default=None, repr=False)
    _traj_batch: object = field(default=None, repr=False)
    _traj_rows: dict = field(default_factory=dict, repr=False)   # run index -> row of the dense-output batch
    # sharded ensemble: the process group and the positions [lo, hi) of `runs` this rank propagated (and holds trajectories of)
    _dist: object = field(default=None, repr=False)
    _local: Optional[tuple] = field(default=None, repr=False)

    def ok_runs(self) -> List[Run]:
        return [r for r in self.runs if isinstance(r.result, PropResult)]

    def final_rv(self) -> np.ndarray:
        return np.array([r.result.state.rv for r in self.ok_runs()])

    def _local_runs(self) -> List[Run]:
        return self.runs if self._local is None else self.runs[self._local[0]:self._local[1]]

    def mean_and_covariance(self):
        """Ensemble mean and (unbiased) covariance of the final 9-vectors [r, v, Cr, Cd, prop mass] of the successful runs
        (NINE components since round 2: Cr, Cd and the propellant mass are dispersed and integrated like r and v; rounds before
        returned the 6-vector only).  No successful run at all: NaNs.
        On a sharded ensemble each rank sums ITS runs only and one all-reduce (RCCL / gloo) of the 1 + 9 + 45 moments
        [count, sum(x - x0), upper triangle of sum((x - x0)(x - x0)^T)] completes them; x0 = the final state of the
        first successful run (every rank holds the gathered final states), which keeps the sums well conditioned.
        Returns (mean[9], cov[9, 9])."""
        d = STATE_DIM
        iu = np.triu_indices(d)
        err, mom, x0 = None, np.zeros(1 + d + len(iu[0])), np.zeros(d)
        try:
            ok = self.ok_runs()
            x0 = _vec9(ok[0].result.state) if ok else np.zeros(d)
            xs = np.array([_vec9(r.result.state) for r in self._local_runs() if isinstance(r.result, PropResult)]).reshape(-1, d)
            if hasattr(self._traj_ctx, "ensemble_moments") and len(xs):
                # the device reduction of the C-ABI (nyx_hip_ensemble_moments: moments_kernel.hip) - the 55 numbers a Rust host
                # would all-reduce; injected evaluators (the CPU tests' oracle stand-ins) have no such entry and take the numpy sums
                b = _abi.StateBatch(len(xs))
                b.set_rv(xs[:, :6])
                b.cr[:], b.cd[:], b.prop_mass_kg[:] = xs[:, 6], xs[:, 7], xs[:, 8]
                mom = self._traj_ctx.ensemble_moments(b, None, x0)
            else:
                xs = xs - x0
                mom = np.concatenate([[float(len(xs))], xs.sum(axis=0), (xs.T @ xs)[iu]])
        except Exception as e:  # noqa: BLE001 - re-raised on every rank below
            err = e
        self._sync_errors(err)
        if self._dist is not None and self._dist.get_world_size() > 1:
            mom = all_reduce_sum(self._dist, mom)
        n, sx = mom[0], mom[1:1 + d]
        if n < 1:   # no successful run anywhere: nothing to average
            return np.full(d, np.nan), np.full((d, d), np.nan)
        sxx = np.zeros((d, d))
        sxx[iu] = mom[1 + d:]
        sxx = sxx + np.triu(sxx, 1).T
        mean = sx / n
        cov = (sxx - n * np.outer(mean, mean)) / (n - 1.0) if n > 1 else np.full((d, d), np.nan)
        return mean + x0, cov

    # ---- reports (results.rs:86-245): one flat list, run after run, failed runs replaced by `value_if_run_failed`
    def _value(self, param: StateParameter, run: Run, rv: np.ndarray) -> np.ndarray:
        s = run.result.state
        return state_value(param, rv, self.mu_km3_s2, cr=s.cr, cd=s.cd, dry_mass_kg=s.dry_mass_kg, prop_mass_kg=s.prop_mass_kg,
                           extra_mass_kg=getattr(s, "extra_mass_kg", 0.0))

    def _sync_errors(self, err: Optional[BaseException]) -> None:
        """Sharded ensemble: a rank that fails BEFORE a collective would leave the others waiting in it for ever.  Every rank
        therefore reduces an error flag first and all of them raise - the failing rank its own exception, the others a
        RuntimeError naming the situation."""
        if self._dist is not None and self._dist.get_world_size() > 1:
            failed = all_reduce_sum(self._dist, np.array([1.0 if err is not None else 0.0]))[0]
            if err is None and failed > 0:
                raise RuntimeError("another rank failed while preparing this collective report; aborted on every rank")
        if err is not None:
            raise err

    def _report(self, param: StateParameter, states_of_run, value_if_run_failed: Optional[float], prepare=None) -> List[float]:
        """One flat list, run after run.  Sharded ensemble: every rank reports the runs it propagated (it holds their
        trajectories) and the pieces are gathered in rank order = index order (contiguous shards): a collective call.
        `prepare` (optional) runs first, inside the guarded section, and returns `states_of_run`."""
        report: List[float] = []
        err = None
        try:
            if prepare is not None:
                states_of_run = prepare()
            for run in self._local_runs():
                if not isinstance(run.result, PropResult):
                    if value_if_run_failed is not None:
                        report.append(float(value_if_run_failed))
                    continue
                rv = states_of_run(run)
                try:
                    report.extend(self._value(param, run, rv).ravel().tolist())
                except StateError:
                    # (the reference pushes the substitute once per state that cannot be evaluated)
                    if value_if_run_failed is not None:
                        report.extend([float(value_if_run_failed)] * len(rv))
        except Exception as e:  # noqa: BLE001 - re-raised on every rank by _sync_errors
            err = e
        self._sync_errors(err)
        if self._dist is not None and self._dist.get_world_size() > 1:
            report = all_gather_lists(self._dist, report)
        return report

    def _need_traj(self):
        if self._traj_batch is None:
            raise ValueError("these results carry no trajectories (with_traj=False)")

    def every_value_of(self, param: StateParameter, step_ns: int, value_if_run_failed: Optional[float] = None) -> List[float]:
        """results.rs:127-160: `param` of every run from the start to the end of its trajectory every `step_ns`."""
        def prepare():
            self._need_traj()
            tb = self._traj_batch
            last = np.array([tb.epoch_ns[max(min(int(tb.len[i]), tb.capacity) - 1, 0), i] for i in range(tb.n)])
            count = int(np.max(np.abs(last - tb.epoch_ns[0]) // abs(int(step_ns)))) + 1 if tb.n else 1
            res = self._traj_ctx.traj_every(tb, int(step_ns), count) if tb.n else None   # (each rank resamples ITS shard on its device)
            return lambda run: res.trajectory(self._traj_rows[run.index])[1]

        return self._report(param, None, value_if_run_failed, prepare=prepare)

    def every_value_of_between(self, param: StateParameter, step_ns: int, start_ns: int, end_ns: int,
                               value_if_run_failed: Optional[float] = None) -> List[float]:
        """results.rs:89-125: as above between max(start, first epoch) and min(end, last epoch) of each run
        (`Traj::every_between`, traj.rs:153-162; the series stops at the first epoch that cannot be interpolated)."""
        def states_of_run(run):
            self._need_traj()
            tb = self._traj_batch
            i = self._traj_rows[run.index]
            ep, _ = tb.trajectory(i)
            lo, hi = max(int(start_ns), int(ep.min())), min(int(end_ns), int(ep.max()))
            if hi < lo:
                return np.zeros((0, 6))
            q = lo + int(step_ns) * np.arange((hi - lo) // int(step_ns) + 1, dtype=np.int64)
            one = _abi.TrajBatch(1, max(len(ep), 1))
            one.len[0] = len(ep)
            one.epoch_ns[: len(ep), 0] = tb.epoch_ns[: len(ep), i]
            one.state[:, : len(ep), 0] = tb.state[:, : len(ep), i]
            states, status = self._traj_ctx.traj_at(one, q)
            bad = np.nonzero(_abi.interp_failed(status[:, 0]))[0]
            return states[: (bad[0] if len(bad) else len(q)), 0]

        return self._report(param, states_of_run, value_if_run_failed)

    def first_values_of(self, param: StateParameter, value_if_run_failed: Optional[float] = None) -> List[float]:
        """results.rs:162-190."""
        return self._report(param, None, value_if_run_failed,
                            prepare=lambda: (self._need_traj(), lambda run: np.asarray(run.result.traj.first())[None, :])[1])

    def last_values_of(self, param: StateParameter, value_if_run_failed: Optional[float] = None) -> List[float]:
        """results.rs:192-220."""
        return self._report(param, None, value_if_run_failed,
                            prepare=lambda: (self._need_traj(), lambda run: np.asarray(run.result.traj.last())[None, :])[1])

    def dispersion_values_of(self, param: StateParameter) -> List[float]:
        """results.rs:222-239: the applied dispersion of `param` for every run; StateError if it was not dispersed."""
        report = []
        for run in self.runs:
            for dparam, val in run.dispersed_state.actual_dispersions:
                if dparam == param:
                    report.append(val)
                    break
            else:
                raise StateError(param)
        return report


def _vec9(sc) -> np.ndarray:
    return np.concatenate([np.asarray(sc.rv, dtype=np.float64), [float(sc.cr), float(sc.cd), float(sc.prop_mass_kg)]])


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
    nominal_state: Optional[Spacecraft] = None   # (carried for the reports' headers in the reference, montecarlo.rs:49-51)
    # injectable for the CPU tests (gloo): (batch, end_epoch_ns) -> (out batch, stats); default = the GPU context
    propagate_fn: Optional[Callable] = field(default=None, repr=False)

    def generate_states(self, skip: int, num_runs: int, seed: Optional[int] = None):
        """montecarlo.rs:277-296: one stream, samples drawn sequentially, first `skip` discarded."""
        use = self.seed if seed is None else seed
        # Pcg64Mcg::new(seed) (montecarlo.rs:284-287); without a seed the reference takes one from the OS
        rng = Pcg64Mcg(use if use is not None else int.from_bytes(os.urandom(16), "little"))
        t = self.random_state.template
        base = np.concatenate([np.asarray(t.rv, dtype=np.float64), [t.cr, t.cd, t.prop_mass_kg]])
        out = []
        for index in range(skip + num_runs):
            v = self.random_state.sample_vector(rng)
            if index < skip:
                continue
            x = base + v
            s = DispersedState(**{f: v for f, v in t.__dict__.items() if f != "actual_dispersions"})
            s.rv, s.cr, s.cd, s.prop_mass_kg = x[:6].copy(), float(x[6]), float(x[7]), float(x[8])
            # template.value(param) - state.value(param) for every dispersed component (multivariate.rs:320-325)
            mu = float(t.frame.mu_km3_s2)

            def val(sc, p):
                return float(state_value(p, np.asarray(sc.rv, dtype=np.float64), mu, cr=sc.cr, cd=sc.cd, dry_mass_kg=sc.dry_mass_kg,
                                         prop_mass_kg=sc.prop_mass_kg, extra_mass_kg=sc.extra_mass_kg))

            s.actual_dispersions = [(d.param, val(t, d.param) - val(s, d.param)) for d in self.random_state.dispersions]
            out.append((index - skip, s))   # `.skip(skip).take(num_runs).enumerate()`: a resumed run counts from 0 again (montecarlo.rs:290-295)
        return out

    def run_until_epoch(self, prop: Propagator, almanac: Almanac, end_epoch_ns: int, num_runs: int, with_traj: bool = True,
                        capacity: int = 4096) -> Results:
        return self.resume_run_until_epoch(prop, almanac, 0, end_epoch_ns, num_runs, with_traj=with_traj, capacity=capacity)

    def resume_run_until_epoch(self, prop: Propagator, almanac: Almanac, skip: int, end_epoch_ns: int, num_runs: int,
                               dist=None, with_traj: bool = True, capacity: int = 4096) -> Results:
        """montecarlo.rs:208-273: every run is `until_epoch_with_traj` (:236-239), so each PropResult carries its
        trajectory - recorded on the device by the propagation kernel, `capacity` accepted steps per run (doubled and
        re-run if a run needs more).  `with_traj=False` skips the dense output.  With `dist` (an initialised
        torch.distributed module) the runs are sharded by contiguous index range over the ranks and every rank returns the
        complete, index-sorted Results; trajectories stay on the rank that propagated them."""
        t_epoch = int(self.random_state.template.epoch_ns)

        def run(ctx, batch):
            if not with_traj:
                return ctx.propagate_until_epoch(batch, int(end_epoch_ns))
            cap = int(capacity)
            while True:
                out, st, traj = ctx.propagate_with_traj(batch, int(end_epoch_ns) - t_epoch, cap)
                need = int(traj.len.max()) if traj.n else 0
                if need <= cap:
                    return out, st, traj, ctx
                cap = max(2 * cap, need)

        return self._run(prop, almanac, skip, num_runs, dist, run, int(end_epoch_ns))

    def run_until_nth_event(self, prop: Propagator, almanac: Almanac, max_duration_ns: int, event, trigger: int, num_runs: int) -> Results:
        """montecarlo.rs:93-110."""
        return self.resume_run_until_nth_event(prop, almanac, 0, max_duration_ns, event, trigger, num_runs)

    def resume_run_until_nth_event(self, prop: Propagator, almanac: Almanac, skip: int, max_duration_ns: int, event, trigger: int,
                                   num_runs: int, dist=None, capacity: int = 4096) -> Results:
        """montecarlo.rs:115-186: every run stops at the `trigger`-th occurrence of `event` (or fails with NthEventError)."""

        def run(ctx, batch):
            out, st, traj, _ = ctx.propagate_until_event(batch, int(max_duration_ns), event, trigger, capacity)
            return out, st, traj, ctx

        return self._run(prop, almanac, skip, num_runs, dist, run, ("event", int(max_duration_ns), event, trigger))

    def _run(self, prop, almanac, skip, num_runs, dist, run, fn_arg) -> Results:
        from .propagator import PropagationError, pack_spacecraft

        rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
        seed = self.seed
        if world > 1 and seed is None:
            # every rank regenerates the whole stream and keeps its shard: without a seed they must still draw the SAME one
            seed = broadcast_seed(dist, None)
        states = self.generate_states(skip, num_runs, seed if seed is not None else None)
        lo, hi = shard_bounds(len(states), rank, world)
        mine = states[lo:hi]
        batch = pack_spacecraft([s for _, s in mine], False)
        # (out, stats) or (out, stats, TrajBatch, evaluator of trajectories: anything with traj_at / traj_every)
        got = self.propagate_fn(batch, fn_arg) if self.propagate_fn is not None else \
            run(prop._context(almanac, self.random_state.template.frame, False), batch)
        out, st = got[0], got[1]
        traj_batch, traj_ctx = (got[2], got[3]) if len(got) == 4 else (None, None)
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
                r = Spacecraft(**{f: v for f, v in s.__dict__.items() if f != "actual_dispersions"})
                r.rv, r.cr, r.cd, r.prop_mass_kg, r.epoch_ns = row[:6].copy(), float(row[6]), float(row[7]), float(row[8]), int(row[9:10].view(np.int64)[0])
                src = (traj_ctx, traj_batch, k - lo) if (traj_batch is not None and lo <= k < hi) else None
                runs.append(Run(index, s, PropResult(r, src)))
            else:
                runs.append(Run(index, s, PropagationError(status, index)))
        rows = {index: k - lo for k, (index, _) in enumerate(states) if lo <= k < hi} if traj_batch is not None else {}
        runs.sort(key=lambda r: r.index)  # par_sort_by_key(index), montecarlo.rs:267
        # every rank holds every final state; trajectories (and hence the reports' raw material) stay on the rank that
        # propagated them: Results gathers the report pieces / reduces the moments over `dist` (index order = rank order,
        # runs being sorted by index and the shards contiguous)
        sharded = dist is not None and world > 1
        return Results(runs, self.scenario, float(self.random_state.template.frame.mu_km3_s2), traj_ctx, traj_batch, rows,
                       dist if sharded else None, (lo, hi) if sharded else None)


def _coll_device(dist):
    import torch

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def all_reduce_sum(dist, v: np.ndarray) -> np.ndarray:
    """One all-reduce (sum) of a small f64 vector: the ensemble moments (RCCL over xGMI on GPUs; latency-bound at 55 doubles)."""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(_coll_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def broadcast_seed(dist, seed: Optional[int]) -> int:
    """A 128-bit seed agreed by all ranks: rank 0's (drawn from the OS when None, as the reference does), as four int32-safe
    limbs in one int64 tensor."""
    import torch

    if dist.get_rank() == 0:
        seed = int.from_bytes(os.urandom(16), "little") if seed is None else int(seed)
    limbs = [((seed or 0) >> (32 * k)) & 0xFFFFFFFF for k in range(4)]
    t = torch.tensor(limbs, dtype=torch.int64, device=_coll_device(dist))
    dist.broadcast(t, src=0)
    return sum(int(x) << (32 * k) for k, x in enumerate(t.cpu().tolist()))


def all_gather_lists(dist, local: List[float]) -> List[float]:
    """Ragged all-gather of per-rank report pieces, concatenated in rank order: lengths first, then one padded all-gather."""
    import torch

    dev = _coll_device(dist)
    world = dist.get_world_size()
    n = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    buf = torch.zeros(max(max(lens), 1), dtype=torch.float64, device=dev)
    if local:
        buf[: len(local)] = torch.tensor(local, dtype=torch.float64).to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out: List[float] = []
    for r in range(world):
        out.extend(parts[r][: lens[r]].cpu().tolist())
    return out


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
