#!/usr/bin/env python3
"""bench.py — benchmark of the MI355X-native propagation path.

Headline (BASELINE.json `metric`, default, `--config 2`): propagated trajectories / second, 10 000-trajectory LEO Monte
Carlo, 70x70 gravity (JGM3, the model present in the reference checkout) + Sun/Moon point masses + cannonball SRP with
Earth shadow, RK89 default options, 1 day (configs[1]).  `--config 3|4|5` run BASELINE.json's other GPU configurations
with the same JSON shape:

    3  JWST covariance Monte Carlo: 5 000 dispersed halo-orbit states, Sun/Moon/Jupiter point masses + SRP with Earth and
       Moon shadows, RK89, 30 days
    4  GEO covariance mapping: 1 000 states, 21x21 + Sun/Moon + SRP (Cr estimated), 9x9 STM, sixty 1-minute time updates
       (`nyx_hip_predict_until`: segment kernel + time-update kernel on one stream)
    5  low lunar orbit: 150x150 (synthetic Kaula field, the GRGM file is a missing blob) + Earth/Sun point masses, DP78,
       3 days; 50 000 trajectories over 8 GPUs = 6 250 per GPU

A "step" is one pass of the hot path over one batch: every rank propagates its shard of dispersed states (already
resident in HBM; config 4 goes through the host-buffer entry, its only one) through the C-ABI, then the ranks exchange
the final states with one RCCL all-gather (the Monte Carlo result collection of north_star).  Weak scaling: the per-GPU
ensemble is fixed (--n), `value` = all ranks' trajectories / max-over-ranks time.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1
    python bench.py --gpus 8                      # started as ONE process: starts its 8 ranks itself (same line, n_gpus 8)
    python bench.py --gpus 8 --scaling strong     # ONE ensemble of the configuration's size cut 8 ways
    python bench.py --gpus 8 --single-process     # one process, 8 contexts through nyx_hip_propagate_batch_sharded

`--gpus N` always measures N ranks: under torchrun WORLD_SIZE must equal N, without it the script starts the ranks itself
(one process per GPU, rendezvous on 127.0.0.1, RCCL) and rank 0 prints the one JSON line with `n_gpus`, `rccl_ranks`,
`per_rank_kernel_ms` and `all_gather_ms`.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402


class _LazyTorch:
    """`import torch` costs one to two minutes on a fresh GPU box (the image pages in); tools/sweep.py and friends import this module
    for workload() only and must not pay for it: the module is imported at its first use."""

    def __getattr__(self, name):
        import torch as _t
        globals()["torch"] = _t
        return getattr(_t, name)


torch = _LazyTorch()

import nyx_amd as nx  # noqa: E402
from nyx_amd import _abi, ephem  # noqa: E402
import scenarios as sc  # noqa: E402

FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector FP64 (= matrix FP64) peak, SURVEY 8(d) / MI355X_MICROARCH.md clocks
HBM_PEAK_GBPS = 8000.0
BYTES_PER_TRAJ = 2 * 13 * 8 + 8 * 8  # 13 f64 + epoch in and out, plus status/details/counters
N_CU = 256


# ---------------------------------------------------------------------------------------------------------------------
# Algorithmic FLOP per force-model evaluation, SURVEY.md section 8(d) ("Algorithmic work per unit"): the REFERENCE
# formulation's cost (what `roofline.achieved` is quoted in); the kernel's own instruction count is lower, see
# `executed_flop_per_eval` below and DESIGN.md section 3.
# ---------------------------------------------------------------------------------------------------------------------
def harmonics_flop(n):
    """4 N(N+1)/2 [interior recursion] + 24 (N(N+1)/2 + N) [(n, m) sum] + 16 N  (SURVEY 8d)."""
    return 4 * n * (n + 1) / 2 + 24 * (n * (n + 1) / 2 + n) + 16 * n


def algorithmic_flop_per_eval(degree, n_pm, srp, n_shadow, stm):
    f = 20.0 + n_pm * (40.0 + 250.0) + (150.0 + 100.0 * max(n_shadow - 1, 0) if srp else 0.0)
    if degree > 0:
        f += 100.0 + harmonics_flop(degree)          # body-fixed rotation in and out + the double sum
    if stm:
        # "+ 1 458 (9x9x9 MAC) + partials (analytic: ~3x the non-STM harmonics cost)"; the point-mass and SRP
        # duals roughly triple their real parts as well
        f += 1458.0 + 3.0 * (harmonics_flop(degree) if degree > 0 else 0.0) + 2.0 * (n_pm * 40.0 + (150.0 if srp else 0.0))
    return f


def executed_flop_per_eval(degree, stm):
    """What the kernel issues for the harmonics (disassembly of harmonics_partial: 7 v_fma_f64 + 2 v_mul_f64 = 16 FLOP per
    table entry, (N+1)(N+2)/2 entries - columns c = 1..N+1 of N+2-c rows; the quad-layout dual variant issues 22 f64
    instructions = 37 FLOP per entry and quad of lanes) — reported beside the algorithmic figure."""
    if degree <= 0:
        return None
    entries = (degree + 1) * (degree + 2) / 2
    return entries * (37.0 if stm else 16.0)


def geo_batch(n, seed):
    b = sc.dispersed_leo_batch(n, seed=seed)
    geo = sc.keplerian_to_cartesian(42164.0, 1e-5, 0.0, 163.0, 75.0, 0.0, ephem.MU_EARTH)   # examples/03_geo_analysis/drift.rs:50
    rv = b.rv()
    b.set_rv(geo[None, :] + (rv - rv.mean(axis=0)))
    return b


def init_covar(n, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, 9, 9)) * np.array([1.0, 1.0, 1.0, 1e-3, 1e-3, 1e-3, 1e-2, 0.0, 0.0])[None, :, None]
    return a @ np.transpose(a, (0, 2, 1))


def workload(cfg_id, degree=None):
    """(propagator, almanac, central frame, batch maker, defaults, description) of one BASELINE configuration."""
    if cfg_id == 2:
        deg = 70 if degree is None else degree
        prop, almanac, central = sc.leo_full_setup(degree=deg)
        return dict(prop=prop, almanac=almanac, central=central, batch=sc.dispersed_leo_batch, n=10_000, hours=24.0, stm=False,
                    flop=algorithmic_flop_per_eval(deg, 2, True, 1, False), exec_flop=executed_flop_per_eval(deg, False), degree=deg,
                    metric="propagated trajectories/sec (10k-ensemble, 1-day RK89)",
                    label=lambda n, h: f"configs[1]: {n}-trajectory LEO Monte Carlo per GPU, {deg}x{deg} JGM3 gravity + Sun/Moon point masses + "
                                       f"cannonball SRP (Earth shadow), RK89 default options, {h:g} h")
    if cfg_id == 3:
        prop, almanac, central = sc.jwst_setup()
        return dict(prop=prop, almanac=almanac, central=central, batch=sc.jwst_batch, n=5_000, hours=30 * 24.0, stm=False,
                    flop=algorithmic_flop_per_eval(0, 3, True, 2, False), exec_flop=None, degree=0,
                    metric="propagated trajectories/sec (JWST 5k-ensemble, 30-day RK89)",
                    label=lambda n, h: f"configs[2]: {n} dispersed JWST halo-orbit states per GPU, Sun/Moon/Jupiter point masses + SRP "
                                       f"(Earth and Moon shadows), RK89 default options, {h / 24:g} days")
    if cfg_id == 4:
        deg = 21 if degree is None else degree
        prop, almanac, central = sc.leo_full_setup(degree=deg)
        return dict(prop=prop, almanac=almanac, central=central, batch=geo_batch, n=1_000, hours=1.0, stm=True,
                    flop=algorithmic_flop_per_eval(deg, 2, True, 1, True), exec_flop=executed_flop_per_eval(deg, True), degree=deg,
                    metric="covariance-mapped trajectories/sec (GEO 1k-ensemble, 9x9 STM, sixty 1-minute EKF time updates)",
                    label=lambda n, h: f"configs[3]: {n} GEO states per GPU, {deg}x{deg} JGM3 + Sun/Moon + SRP (Cr estimated), 9x9 STM, "
                                       f"{int(round(h * 60))} one-minute time updates with STM reset (predict_until), RK89")
    if cfg_id == 5:
        deg = 150 if degree is None else degree
        prop, almanac, central = sc.lunar_setup(degree=deg)
        return dict(prop=prop, almanac=almanac, central=central, batch=sc.lunar_batch, n=6_250, hours=72.0, stm=False,
                    flop=algorithmic_flop_per_eval(deg, 2, False, 0, False), exec_flop=executed_flop_per_eval(deg, False), degree=deg,
                    metric="propagated trajectories/sec (LLO 50k-ensemble over 8 GPUs, 3-day DP78)",
                    label=lambda n, h: f"configs[4]: {n} low-lunar-orbit states per GPU (50 000 over 8), {deg}x{deg} synthetic Kaula field + "
                                       f"Earth/Sun point masses, DP78 default options, {h / 24:g} days")
    raise SystemExit(f"unknown --config {cfg_id}")


def tensor_states(batch: _abi.StateBatch, dev):
    """Uploads a host SoA batch into torch device tensors and returns (tensors, States view)."""
    t = {"epoch_ns": torch.from_numpy(batch.epoch_ns).to(dev), "step_ns": torch.zeros(batch.n, dtype=torch.int64, device=dev)}
    for f in _abi.F64_FIELDS:
        t[f] = torch.from_numpy(getattr(batch, f)).to(dev)
    s = _abi.States()
    s.n = batch.n
    s.epoch_ns = C.cast(t["epoch_ns"].data_ptr(), _abi.c_int64_p)
    s.step_ns = C.cast(t["step_ns"].data_ptr(), _abi.c_int64_p)
    for f in _abi.F64_FIELDS:
        setattr(s, f, C.cast(t[f].data_ptr(), _abi.c_double_p))
    return t, s


def tensor_stats(n, dev):
    t = {"status": torch.zeros(n, dtype=torch.int32, device=dev), "last_attempts": torch.zeros(n, dtype=torch.int32, device=dev),
         "last_step_ns": torch.zeros(n, dtype=torch.int64, device=dev), "last_error": torch.zeros(n, dtype=torch.float64, device=dev),
         "n_accepted": torch.zeros(n, dtype=torch.int64, device=dev), "n_rejected": torch.zeros(n, dtype=torch.int64, device=dev),
         "n_evals": torch.zeros(n, dtype=torch.int64, device=dev)}
    s = _abi.StepStats()
    s.status = C.cast(t["status"].data_ptr(), _abi.c_int32_p)
    s.last_attempts = C.cast(t["last_attempts"].data_ptr(), _abi.c_int32_p)
    s.last_step_ns = C.cast(t["last_step_ns"].data_ptr(), _abi.c_int64_p)
    s.last_error = C.cast(t["last_error"].data_ptr(), _abi.c_double_p)
    s.n_accepted = C.cast(t["n_accepted"].data_ptr(), _abi.c_int64_p)
    s.n_rejected = C.cast(t["n_rejected"].data_ptr(), _abi.c_int64_p)
    s.n_evals = C.cast(t["n_evals"].data_ptr(), _abi.c_int64_p)
    return t, s


def host_threads():
    """(hardware threads, physical cores) of this host."""
    logical = os.cpu_count() or 1
    try:
        cores = set()
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        return logical, (len(cores) or logical)
    except OSError:
        return logical, logical


CPU_BASELINE_MIN_WALL_S = 12.0   # the timed CPU wall of every configuration: >= 10 s, so that thread start-up and scheduling noise are < 1 %


def cpu_baseline(shard, compiled, n_per_gpu, hours, min_wall=None):
    """The oracle (CPU restatement of the reference, kind = "port") timed on this box's host on a bounded sample of the SAME
    workload: one pthread per hardware thread over trajectories (rayon par_iter analogue).  Rounds of `threads` trajectories -
    the head of this rank's shard first (the states the device propagated, not a re-draw), cycling through the shard when it
    is exhausted - are propagated until the timed wall reaches CPU_BASELINE_MIN_WALL_S; full-length trajectories when one round
    fits the budget, otherwise the head of the propagation (stated in `sample`, the rate scaled by the fraction)."""
    import oracle_lib
    min_wall = CPU_BASELINE_MIN_WALL_S if min_wall is None else min_wall
    threads, phys = host_threads()
    probe = shard.slice(0, min(threads, shard.n))
    probe_h = min(0.5, hours)
    t0 = time.time()
    oracle_lib.propagate(compiled, probe, int(probe_h * 3600) * nx.NS_PER_S, n_threads=threads)
    per_hour = (time.time() - t0) / probe_h  # wall seconds per hour of propagation of one round (`threads` trajectories)
    samp_h = hours
    if per_hour * hours > 1.5 * min_wall:  # a full-length round is too long: time the first `samp_h` hours instead
        samp_h = max(probe_h, min(hours, float(int(1.2 * min_wall / per_hour * 4) / 4.0)))
    span = int(samp_h * 3600) * nx.NS_PER_S
    n1 = min(threads, shard.n)
    dt, n, evals, rounds, lo = 0.0, 0, 0.0, 0, 0
    sample = out = None
    while dt < min_wall and rounds < 400:
        if lo + n1 > shard.n:
            lo = 0   # (the shard is exhausted: the same states again - the cost does not depend on who propagates them)
        chunk = shard.slice(lo, lo + n1)
        t0 = time.time()
        o, st = oracle_lib.propagate(compiled, chunk, span, n_threads=threads)
        dt += time.time() - t0
        assert (st.status == 0).all()
        if sample is None:
            sample, out = chunk, o
        n += chunk.n
        evals += float(st.n_evals.sum())
        rounds += 1
        lo += n1
    frac = samp_h / hours
    what = f"full {hours:g} h propagation each" if frac == 1.0 else \
        f"the first {samp_h:g} h of the {hours:g} h propagation each (rate scaled by {frac:.4g}: the orbit is periodic, the cost per hour constant)"
    return {"value": n / dt * frac, "unit": "trajectories/s", "cores": threads, "physical_cores": phys, "kind": "port",
            "sample": f"{n} propagations ({rounds} rounds of {n1} of the {n_per_gpu} dispersed states), {what}, {threads} pthreads = hardware threads "
                      f"on {phys} physical cores (rayon par_iter analogue), {dt:.1f} s wall, {int(evals)} force evaluations",
            "wall_s": dt, "evals_per_s": evals / dt}, sample, out, samp_h


def cpu_baseline_predict(shard, compiled, p0, end_ns, n_per_gpu, hours, min_wall=None):
    """Config 4: the oracle's predict_until twin over independent estimates.  One OD process is sequential, but the workload
    is 1 000 independent ones, which the reference would par_iter: every hardware thread maps its own chunk of 16 estimates
    again and again (the C call releases the GIL; its arguments are prepared once per thread) until the timed wall reaches
    CPU_BASELINE_MIN_WALL_S; the one-thread rate is kept beside it."""
    import threading
    import oracle_lib
    from nyx_amd import od
    threads, phys = host_threads()
    step = 60 * nx.NS_PER_S
    chunk = 16
    lib = oracle_lib.load()
    t1 = time.time()
    oracle_lib.predict_until(compiled, shard.slice(0, chunk), p0[:chunk], end_ns, step)
    one_chunk_s = time.time() - t1
    reps = max(2, int(np.ceil((CPU_BASELINE_MIN_WALL_S if min_wall is None else min_wall) / max(one_chunk_s, 1e-3))))
    gate = threading.Barrier(threads + 1)
    done = [0] * threads

    def worker(k):
        lo = (k * chunk) % max(shard.n - chunk, 1)
        held = {}

        def capture(*args):   # (keeps the ctypes arguments - and through `keep` everything they point to - alive for the timed loop)
            held["args"] = args
            return 0
        keep = od.predict_until(None, shard.slice(lo, lo + chunk), p0[lo:lo + chunk], end_ns, step, _call=capture)
        gate.wait()
        for _ in range(reps):
            rc = lib.nyx_oracle_predict_until(C.byref(compiled.cfg), *held["args"])   # (the covariances keep being mapped on: same cost)
            assert rc == 0
            done[k] += chunk
        del keep

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
    for t in ts:
        t.start()
    gate.wait()
    t1 = time.time()
    for t in ts:
        t.join()
    dt = time.time() - t1
    total = sum(done)
    # one thread, for reference: the first 256 estimates
    ns = min(shard.n, 256)
    t1 = time.time()
    ref = oracle_lib.predict_until(compiled, shard.slice(0, ns), p0[:ns], end_ns, step)
    dt1 = time.time() - t1
    return {"value": total / dt, "unit": "trajectories/s", "cores": threads, "physical_cores": phys, "kind": "port", "wall_s": dt,
            "one_thread": {"value": ns / dt1, "unit": "trajectories/s", "wall_s": dt1},
            "sample": f"{total} covariance mappings ({threads} host threads x {reps} passes over a chunk of {chunk} of the {n_per_gpu} GEO states), all "
                      f"{int(round(hours * 60))} one-minute time updates each, the oracle's predict_until twin (independent estimates mapped in "
                      f"parallel, as a par_iter over OD processes would), {dt:.1f} s wall on {phys} physical cores"}, ref, ns


def measured_traffic(cfg_id, n, hours, degree):
    """HBM bytes per launch from the committed rocprofv3 --pmc pass of this command (profiles/*hbm_traffic*.json) - only when
    that pass was taken on THIS build: the file carries the stamp of the kernel sources it was measured on
    (tools/kernel_stamp.py) and a figure whose stamp differs from the tree's is not quoted (traffic = null, the reason in
    `traffic_source`): counters cannot be read from inside the run, a stale file must not pass for a measurement."""
    from kernel_stamp import kernel_source_stamp
    stamp = kernel_source_stamp(ROOT)
    best, stale = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json"))):
        try:
            tj = json.load(open(path))
        except Exception:
            continue
        if tj.get("config", 2) == cfg_id and tj.get("n") == n and tj.get("hours") == hours and tj.get("degree") == degree:
            rel = os.path.relpath(path, ROOT)
            if tj.get("kernel_source_stamp") == stamp:
                best = (tj.get("hbm_bytes_per_launch"), rel)
            else:
                stale = rel
    if best:
        return best
    if stale:
        return None, f"stale: {stale} was measured on kernel sources {json.load(open(os.path.join(ROOT, stale))).get('kernel_source_stamp', 'unstamped')}, this tree is {stamp}"
    return None, None


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n_ranks, argv):
    """`python bench.py --gpus N` started as ONE process: start the N ranks here (one process per GPU, rendezvous on
    127.0.0.1) and wait for them; rank 0 prints the line.  Under torchrun (WORLD_SIZE set) this is never reached."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NYX_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:   # poll ALL ranks: the one that fails is rarely the first in the list, and the others would sit in a collective for ever
            for p in list(live):
                prc = p.poll()
                if prc is None:
                    continue
                live.remove(p)
                if prc != 0 and rc == 0:
                    rc = prc
                    for q in live:   # (exact PIDs, our own children)
                        q.terminate()
            if live:
                time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise
    return rc


def selftest_launch(rank, world):
    """CPU check of the launch layer (tests/test_bench_launch.py): rendezvous, all-gather and max-over-ranks of synthetic
    numbers over gloo - everything of an N-rank run but the device work."""
    import torch.distributed as dist
    if os.environ.get("NYX_BENCH_SELFTEST_FAIL_RANK") == str(rank):   # (the test of the launcher's supervision: this rank dies before the rendezvous)
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.full((4, 7), float(rank + 1), dtype=torch.float64)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    k = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(k, torch.tensor([10.0 * (rank + 1)], dtype=torch.float64))
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "ranks": world, "backend": "gloo", "self_launched": bool(os.environ.get("NYX_BENCH_SELF_LAUNCHED")),
                          "gathered_sum": float(sum(g.sum().item() for g in got)), "max_time": float(t.item()),
                          "per_rank_kernel_ms": [float(x.item()) for x in k]}), flush=True)
    dist.destroy_process_group()


def single_process(args, w):
    """`--single-process`: ONE process drives every device through the C symbol `nyx_hip_propagate_batch_sharded` (one context
    and one host thread per device, contiguous index shards, results in place; host buffers: the staging copies are inside
    the timed call).  The ensemble is `--gpus` x the per-GPU size (weak) or the configuration's own size (strong)."""
    if w["stm"]:
        raise SystemExit("--single-process drives nyx_hip_propagate_batch_sharded: plain propagation only (configs 2, 3, 5)")
    ndev = torch.cuda.device_count()
    if args.gpus > ndev and not args.oversubscribe:
        raise SystemExit(f"--single-process --gpus {args.gpus}: {ndev} device(s) visible (--oversubscribe puts several contexts on one device)")
    n_per = args.n or w["n"]
    hours = args.hours or w["hours"]
    total = n_per if args.scaling == "strong" else n_per * args.gpus
    compiled = w["prop"].compile(w["almanac"], w["central"], stm=False)
    ctxs = [nx.GpuContext(compiled, device=k % ndev) for k in range(args.gpus)]
    full = w["batch"](total, seed=0)
    dur_ns = int(round(hours * 3600)) * nx.NS_PER_S
    for _ in range(args.warmup):
        nx.propagate_sharded(ctxs, full, dur_ns)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, st = nx.propagate_sharded(ctxs, full, dur_ns)
    elapsed = time.perf_counter() - t0
    if (st.status != 0).any():
        raise SystemExit(f"{int((st.status != 0).sum())} trajectories failed")
    k_ms = [c.last_kernel_ms() for c in ctxs]
    n_evals = int(st.n_evals.sum())
    achieved_tf = n_evals * w["flop"] / (max(k_ms) * 1e-3) / 1e12 / args.gpus   # per device, on the slowest device's kernel time
    line = {"metric": w["metric"], "value": total * args.steps / elapsed, "unit": "trajectories/s", "n_gpus": min(args.gpus, ndev), "contexts": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "launch": "single-process",
            "config": {"workload": w["label"](total // args.gpus, hours), "baseline_config": args.config, "trajectories_total": total,
                       "sharding": "nyx_hip_propagate_batch_sharded: contiguous index shards over the contexts, one host thread per device, host "
                                   "buffers in and out (PCIe-inclusive), no collective"},
            "per_rank_kernel_ms": k_ms, "force_evals_per_launch": n_evals,
            "roofline": {"bound": "valu_fp64", "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS, "traffic": None, "note": "per device, slowest device's kernel time"}}
    print(json.dumps(line), flush=True)


OTHER_CPU_WALL_S = 6.0   # timed CPU wall of the baselines the other configurations carry (the headline's own: CPU_BASELINE_MIN_WALL_S)


def measure_other(cfg_id, dev, device_index, lib, n=0, hours=0.0, tag=None, cpu=False):
    """One timed pass of another BASELINE configuration inside the headline run (`other_configs` of the JSON line): its own context,
    one short warm-up launch (code object, tables, mailboxes), then ONE launch of the full workload with inputs resident in HBM
    (configs 3, 5 and the full-chip launch of configs[1]'s force model: nyx_hip_propagate_batch_device + the ensemble moments) or
    through the host-buffer entry (config 4: nyx_hip_predict_until has no other), bracketed by synchronisations.  The driver's
    clock is around all of it."""
    t_all = time.perf_counter()
    w = workload(cfg_id)
    n = n or w["n"]
    hours = hours or w["hours"]
    compiled = w["prop"].compile(w["almanac"], w["central"], stm=w["stm"])
    ctx = nx.GpuContext(compiled, device=device_index)
    batch = w["batch"](n, seed=0)
    dur_ns = int(round(hours * 3600)) * nx.NS_PER_S
    stream = torch.cuda.current_stream(dev)
    if not w["stm"]:
        tin, sin = tensor_states(batch, dev)
        tout, sout = tensor_states(batch, dev)
        tst, sst = tensor_stats(n, dev)
        mom = torch.zeros(55, dtype=torch.float64, device=dev)
        x0 = np.concatenate([batch.rv()[0], [batch.cr[0], batch.cd[0], batch.prop_mass_kg[0]]]).astype(np.float64)

        def go(d_ns):
            rc = lib.nyx_hip_propagate_batch_device(ctx._h, C.byref(sin), d_ns, C.byref(sout), C.byref(sst), C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(_abi.last_error())
            rc = lib.nyx_hip_ensemble_moments_device(ctx._h, C.byref(sout), C.c_void_p(tst["status"].data_ptr()), x0.ctypes.data_as(_abi.c_double_p),
                                                     C.c_void_p(mom.data_ptr()), C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(_abi.last_error())
        go(min(dur_ns, 1800 * nx.NS_PER_S))   # warm-up: the first half hour
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        go(dur_ns)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        k_ms = ctx.last_kernel_ms()
        n_evals = int(tst["n_evals"].sum().item())
        n_bad = int((tst["status"] != 0).sum().item())
        helpers = ctx.last_coop_helpers()
        per_wg = 64
    else:
        batch.stm = np.zeros((n, 81))
        batch.reset_stm()
        p0 = init_covar(n)
        end_ns = int(batch.epoch_ns[0]) + dur_ns
        nx.predict_until(ctx, batch, p0, int(batch.epoch_ns[0]) + 120 * nx.NS_PER_S, 60 * nx.NS_PER_S)   # warm-up: two updates
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res = nx.predict_until(ctx, batch, p0, end_ns, 60 * nx.NS_PER_S)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        k_ms = res.kernel_ms
        n_evals, n_bad = int(res.stats.n_evals.sum()), int((res.stats.status != 0).sum())
        helpers = 0
        per_wg = 16 if (n + 15) // 16 <= 2 * N_CU else 64
    ctx.close()
    if n_bad:
        raise SystemExit(f"config {cfg_id}: {n_bad} trajectories failed")
    cb = None
    if cpu:
        # VERDICT r5, Weak 9: the CPU oracle on this box's host threads for THIS configuration too (a bounded sample, OTHER_CPU_WALL_S
        # of timed wall): context for "is this path worth a GPU at this ensemble size", not a target
        if not w["stm"]:
            cb, _, _, _ = cpu_baseline(batch, compiled, n, hours, min_wall=OTHER_CPU_WALL_S)
        else:
            cb, _, _ = cpu_baseline_predict(batch, compiled, p0, end_ns, n, hours, min_wall=OTHER_CPU_WALL_S)
        cb["gpu_over_cpu"] = (n / wall) / cb["value"]
    achieved_tf = n_evals * w["flop"] / (k_ms * 1e-3) / 1e12
    owners = (n + per_wg - 1) // per_wg
    out = {"baseline_config": cfg_id, "workload": w["label"](n, hours), "metric": w["metric"], "value": n / wall, "unit": "trajectories/s",
           "ms_per_step": wall * 1e3, "kernel_ms": k_ms, "value_device": n / (k_ms * 1e-3), "steps": 1, "force_evals_per_launch": n_evals,
           "roofline": {"bound": "valu_fp64", "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS, "algorithmic_flop_per_eval": w["flop"]},
           "occupancy": {"owner_workgroups": owners, "helper_workgroups": helpers, "cus": N_CU, "cu_fraction": min(1.0, (owners + helpers) / N_CU)},
           "inputs": "host buffers (nyx_hip_predict_until: PCIe-inclusive)" if w["stm"] else "resident in HBM",
           "wall_s_with_setup": None, "fits_in_driver_run": True}
    if tag:
        out["tag"] = tag
    if cb is not None:
        out["cpu_baseline"] = cb
    out["wall_s_with_setup"] = time.perf_counter() - t_all
    return out


class ClockSampler:
    """Shader clock of the device WHILE the timed steps run (amdsmi through torch.cuda.clock_rate, 5 Hz from a side thread; best
    effort: None where the query is not available).  Diagnostic only - a box whose clock is capped shows up here, not as a regression."""

    def __init__(self, device_index):
        import threading
        self.dev, self.samples, self._stop = device_index, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        try:   # (the first query initialises the library: not inside the timed region)
            torch.cuda.clock_rate(self.dev)
            self.ok = True
        except Exception:
            self.ok = False

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(int(torch.cuda.clock_rate(self.dev)))
            except Exception:
                return
            self._stop.wait(0.2)

    def __enter__(self):
        if self.ok:
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.ok:
            self._t.join(timeout=2.0)

    def report(self):
        if not self.samples:
            return None
        v = sorted(self.samples)
        return {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "samples": len(v), "source": "torch.cuda.clock_rate (amdsmi), sampled during the timed steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configuration (1-based; default 2 = the headline)")
    ap.add_argument("--n", type=int, default=0, help="trajectories per GPU (weak) or in total (strong); 0 = the configuration's own size")
    ap.add_argument("--hours", type=float, default=0.0, help="propagation length (0 = the configuration's own)")
    ap.add_argument("--degree", type=int, default=None)
    ap.add_argument("--waves", type=int, default=0, help="column-split waves per workgroup (0 = auto)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every GPU gets the configuration's ensemble (total = N x n); strong: ONE ensemble of that size is cut N ways")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="collective backend of the final-state exchange (nccl = RCCL)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than devices (rank r -> device r mod devices; RCCL refuses two ranks on one device: use --backend gloo)")
    ap.add_argument("--single-process", action="store_true", help="one process, every device through nyx_hip_propagate_batch_sharded")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise the process group and run the all-gather / all-reduce of every step even with ONE rank "
                         "(exercises the RCCL path on a 1-GPU box; the line says so in `launch`)")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-output", action="store_true", help="skip the extra launch with the trajectories recorded")
    ap.add_argument("--no-host-call", action="store_true", help="skip the PCIe-inclusive host-buffer call")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the one-pass measurements of configs 3, 4, 5 and of the full-chip launch that the headline run carries in `other_configs`")
    args = ap.parse_args()

    # ---- how many ranks, and who starts them.  Under torchrun (the driver's N > 1 launch) WORLD_SIZE is set and must agree
    # with --gpus; started as one process with --gpus N > 1 the ranks are started HERE (spawn_ranks) - `--gpus N` always
    # measures N ranks, however the script was started.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.single_process:
        if not args.selftest_launch:   # what every rank would find out on its own, said once and before anything is started
            ndev0 = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if ndev0 == 0:
                raise SystemExit("bench.py needs an MI355X: no HIP device visible")
            if args.gpus > ndev0 and not args.oversubscribe:
                raise SystemExit(f"--gpus {args.gpus}: {ndev0} device(s) visible (--oversubscribe --backend gloo shares devices)")
            if args.gpus > ndev0 and args.backend == "nccl":
                raise SystemExit("RCCL refuses two ranks on one device: --oversubscribe needs --backend gloo")
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_process:
        if world > 1:
            raise SystemExit("--single-process is one process by definition: do not start it under torchrun")
    elif world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest_launch:
        return selftest_launch(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    w = workload(args.config, args.degree)
    if args.single_process:
        return single_process(args, w)
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not args.oversubscribe:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {ndev} device(s) visible (--oversubscribe --backend gloo shares devices)")
        if args.backend == "nccl":
            raise SystemExit("RCCL refuses two ranks on one device: --oversubscribe needs --backend gloo")
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    coll = world > 1 or args.force_collectives   # the collectives of a step are issued (N > 1, or asked for with one rank)
    if coll:
        import torch.distributed as dist  # noqa: F811
        if world == 1 and "MASTER_ADDR" not in os.environ:   # one rank started by hand: a rendezvous of its own
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    hours = args.hours or w["hours"]
    prop, almanac, central = w["prop"], w["almanac"], w["central"]
    compiled = prop.compile(almanac, central, stm=w["stm"])
    ctx = nx.GpuContext(compiled, device=device_index)
    if args.waves:
        ctx.set_column_waves(args.waves)
    lib = _abi.load_library()

    # contiguous index shards of ONE ensemble (seed = global stream).  weak: rank r owns [r n, (r + 1) n) of N x n;
    # strong: the configuration's own ensemble cut N ways (shard_bounds: sizes differ by at most one)
    n_cfg = args.n or w["n"]
    total_traj = n_cfg if args.scaling == "strong" else n_cfg * world
    lo, hi = nx.shard_bounds(total_traj, rank, world)
    n = hi - lo
    n_max = nx.shard_bounds(total_traj, 0, world)[1]   # the largest shard (rank 0 holds one of them): all-gather pieces are padded to it
    full = w["batch"](total_traj, seed=0)
    shard = full.slice(lo, hi)
    dur_ns = int(round(hours * 3600)) * nx.NS_PER_S
    stream = torch.cuda.current_stream(dev)
    gdev = dev if args.backend == "nccl" else torch.device("cpu")
    gathered = [torch.empty((n_max, 7), dtype=torch.float64, device=gdev) for _ in range(world)] if coll else None
    gather_s = [0.0]
    host_call = None

    def exchange(final):
        """one all-gather of the final states (n x 7 f64 per rank, padded to the largest shard): RCCL over xGMI, latency-bound"""
        if final.shape[0] < n_max:
            final = torch.cat([final, torch.zeros((n_max - final.shape[0], 7), dtype=torch.float64, device=final.device)])
        if args.backend == "nccl":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dist.all_gather(gathered, final)
            e1.record(stream)
            return (e0, e1)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        dist.all_gather(gathered, final.cpu())
        gather_s[0] += time.perf_counter() - t1
        return None

    gather_events = []
    if not w["stm"]:
        tin, sin = tensor_states(shard, dev)
        tout, sout = tensor_states(shard, dev)
        tst, sst = tensor_stats(n, dev)

        # ensemble moments on the device (nyx_hip_ensemble_moments_device: count, sum(x - x0), sum((x - x0)(x - x0)^T) of the final
        # 9-vectors = 55 doubles; x0 = the nominal start state, the same on every rank): what mean and covariance of the whole
        # ensemble need from a rank is ONE all-reduce of these, no D2H of the states (north_star's "covariance reduction")
        mom = torch.zeros(55, dtype=torch.float64, device=dev)
        x0 = np.concatenate([full.rv()[0], [full.cr[0], full.cd[0], full.prop_mass_kg[0]]]).astype(np.float64)
        x0_p = x0.ctypes.data_as(_abi.c_double_p)

        def step():
            rc = lib.nyx_hip_propagate_batch_device(ctx._h, C.byref(sin), dur_ns, C.byref(sout), C.byref(sst), C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(_abi.last_error())
            rc = lib.nyx_hip_ensemble_moments_device(ctx._h, C.byref(sout), C.c_void_p(tst["status"].data_ptr()), x0_p, C.c_void_p(mom.data_ptr()),
                                                     C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(_abi.last_error())
            if coll:  # final-state collection (one all-gather) and the moments (one all-reduce of 55 doubles)
                final = torch.stack([tout[f] for f in _abi.F64_FIELDS[:6]] + [tout["epoch_ns"].to(torch.float64)], dim=1)
                gather_events.append(exchange(final))
                if args.backend == "nccl":
                    dist.all_reduce(mom)
                else:
                    mh = mom.cpu()
                    dist.all_reduce(mh)
                    mom.copy_(mh)
    else:
        # covariance mapping: the C-ABI entry takes host buffers (states, covariances) and keeps the whole segment /
        # time-update loop on one stream; the timed region therefore includes the one H2D and the one D2H of the call
        shard.stm = np.zeros((n, 81))
        shard.reset_stm()
        p0 = init_covar(total_traj)[lo:hi]
        end_ns = int(shard.epoch_ns[0]) + dur_ns
        last = {}

        def step():
            last["res"] = nx.predict_until(ctx, shard, p0, end_ns, 60 * nx.NS_PER_S)
            if coll:
                r = last["res"].states
                final = torch.from_numpy(np.concatenate([r.rv(), r.epoch_ns[:, None].astype(np.float64)], axis=1)).to(dev)
                gather_events.append(exchange(final))

    def barrier():
        if coll:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    gather_events.clear()
    gather_s[0] = 0.0
    kernel_ms = []
    clocks = ClockSampler(device_index)
    with clocks:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            kernel_ms.append(ctx.last_kernel_ms() if not w["stm"] else last["res"].kernel_ms)  # HIP events on the launch stream
        barrier()
        elapsed = time.perf_counter() - t0
    k_ms = float(np.mean(kernel_ms))
    per_rank_kernel_ms = [k_ms]
    gather_ms = None
    if coll:
        red = dev if args.backend == "nccl" else torch.device("cpu")
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if args.backend == "nccl":
            gather_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in gather_events]))
        else:
            gather_ms = gather_s[0] / args.steps * 1e3
        kall = [torch.zeros(2, dtype=torch.float64, device=red) for _ in range(world)]
        dist.all_gather(kall, torch.tensor([k_ms, gather_ms], dtype=torch.float64, device=red))
        per_rank_kernel_ms = [float(t[0].item()) for t in kall]
        gather_ms = max(float(t[1].item()) for t in kall)

    if not w["stm"]:
        n_evals = int(tst["n_evals"].sum().item())
        n_bad = int((tst["status"] != 0).sum().item())
        n_acc, n_rej = int(tst["n_accepted"].sum().item()), int(tst["n_rejected"].sum().item())
    else:
        st = last["res"].stats
        n_evals, n_bad = int(st.n_evals.sum()), int((st.status != 0).sum())
        n_acc, n_rej = int(st.n_accepted.sum()), int(st.n_rejected.sum())
    if n_bad:
        raise SystemExit(f"{n_bad} trajectories failed")
    ev_all = n_evals
    if coll:   # force evaluations of the whole job (strong scaling: the shards differ)
        red = dev if args.backend == "nccl" else torch.device("cpu")
        evt = torch.tensor([float(n_evals)], dtype=torch.float64, device=red)
        dist.all_reduce(evt, op=dist.ReduceOp.SUM)
        ev_all = int(evt.item())
    value = total_traj * args.steps / elapsed

    if rank == 0:
        flops = n_evals * w["flop"]
        achieved_tf = flops / (k_ms * 1e-3) / 1e12
        alg_gbps = n * BYTES_PER_TRAJ / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(args.config, n, hours, w["degree"])
        # CU occupancy of the launch: trajectory-owning workgroups (64 trajectories each) + cooperative-mode helpers
        # (the STM kernel's quad layout - chosen when ceil(n / 16) <= 2 x CUs - has 16 trajectories per workgroup)
        per_wg = 16 if (w["stm"] and (n + 15) // 16 <= 2 * N_CU) else 64
        owners = (n + per_wg - 1) // per_wg
        helpers = ctx.last_coop_helpers() if not w["stm"] else 0
        line = {
            "metric": w["metric"],
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "launch": "torchrun" if (world > 1 and not os.environ.get("NYX_BENCH_SELF_LAUNCHED")) else ("self-launched ranks" if world > 1 else ("single rank, collectives forced" if coll else "single rank")),
            "rccl_ranks": world if (coll and args.backend == "nccl") else 0, "backend": args.backend if coll else None,
            "devices_visible": ndev, "per_rank_kernel_ms": per_rank_kernel_ms, "all_gather_ms": gather_ms, "sclk_mhz": clocks.report(),
            "config": {"workload": w["label"](n, hours), "baseline_config": args.config,
                       "trajectories_per_gpu": n, "trajectories_total": total_traj, "column_waves": args.waves or "auto",
                       "tuning": "nyx_hip_tuning_t defaults: NYX_HIP_SCHED_MODEL (process-independent column schedule), cooperative mode auto",
                       "sharding": "contiguous index shards, no data-path collective; one all-gather of final states per step"
                                   + (" (strong: ONE ensemble of the configuration's size cut over the ranks)" if args.scaling == "strong" else "")},
            "ensemble_moments": (None if w["stm"] else {"count": float(mom[0].item()), "reduced_over_ranks": world,
                                                        "trace_cov_pos_km2": float(((mom[10] + mom[19] + mom[27]) - (mom[1] ** 2 + mom[2] ** 2 + mom[3] ** 2) / mom[0]).item() / max(float(mom[0].item()) - 1.0, 1.0)),
                                                        "note": "nyx_hip_ensemble_moments_device inside every timed step; N > 1: one all-reduce of 55 f64"}),
            "force_evals_per_s": ev_all / (elapsed / args.steps),
            "force_evals_per_launch": n_evals, "accepted_steps": n_acc, "rejected_attempts": n_rej,
            "kernel_ms": k_ms,
            "occupancy": {"workgroups": owners + helpers, "owner_workgroups": owners, "helper_workgroups": helpers, "trajectories_per_workgroup": per_wg, "cus": N_CU,
                          "cu_fraction": min(1.0, (owners + helpers) / N_CU)},
            "roofline": {"bound": "valu_fp64", "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_flop_per_eval": w["flop"],
                         "hbm": {"algorithmic": alg_gbps, "achieved": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                                 "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": ((traffic / (k_ms * 1e-3) / 1e9) if traffic else alg_gbps) / HBM_PEAK_GBPS,
                                 "algorithmic_bytes_per_trajectory": BYTES_PER_TRAJ}},
        }
        if w["exec_flop"]:
            # what the kernel itself issues for the harmonics (lower than the reference formulation's count): the f64-pipe view
            ex_tf = n_evals * w["exec_flop"] / (k_ms * 1e-3) / 1e12
            line["roofline"]["executed_harmonics_flop_per_eval"] = w["exec_flop"]
            line["roofline"]["executed_frac"] = ex_tf / FP64_VECTOR_PEAK_TFLOPS
        if world == 1 and not args.no_host_call:
            # SURVEY 8d "report both": the same batch through the HOST-buffer entry (H2D of the states, kernel, D2H of the
            # results inside the call; ctx_create excluded), timed once with the wall clock
            if not w["stm"]:
                t1 = time.perf_counter()
                out_h, st_h = ctx.propagate(shard, dur_ns)
                wall = time.perf_counter() - t1
                host_call = {"ms": wall * 1e3, "value": n / wall, "unit": "trajectories/s", "kernel_ms": ctx.last_kernel_ms(),
                             "note": "nyx_hip_propagate_batch on host buffers: PCIe-inclusive wall time of one call"}
            else:
                host_call = {"ms": elapsed / args.steps * 1e3, "value": value, "unit": "trajectories/s", "kernel_ms": k_ms,
                             "note": "nyx_hip_predict_until takes host buffers: the timed steps ARE PCIe-inclusive; kernel_ms = the "
                                     "device time of the loop (round 6: ONE launch of the quad STM kernel for all sixty time updates)"}
            line["host_call"] = host_call
        if world == 1 and not args.no_dense_output and not w["stm"] and args.config == 2:
            # The reference's Monte Carlo runs `until_epoch_with_traj` (mc/montecarlo.rs:236-239): the same launch with the
            # dense output on (every accepted state of every run appended in HBM), timed once, reported beside the headline.
            cap = int(tst["n_accepted"].max().item()) + 2
            t_ep = torch.zeros((cap, n), dtype=torch.int64, device=dev)
            t_st = torch.zeros((6, cap, n), dtype=torch.float64, device=dev)
            t_len = torch.zeros(n, dtype=torch.int32, device=dev)
            tr = _abi.Traj()
            tr.capacity = cap
            tr.epoch_ns = C.cast(t_ep.data_ptr(), _abi.c_int64_p)
            for k, f in enumerate(["x_km", "y_km", "z_km", "vx_km_s", "vy_km_s", "vz_km_s"]):
                setattr(tr, f, C.cast(t_st[k].data_ptr(), _abi.c_double_p))
            tr.len = C.cast(t_len.data_ptr(), _abi.c_int32_p)
            rc = lib.nyx_hip_propagate_batch_with_traj_device(ctx._h, C.byref(sin), dur_ns, C.byref(sout), C.byref(sst), C.byref(tr),
                                                              C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(_abi.last_error())
            torch.cuda.synchronize(dev)
            d_ms = ctx.last_kernel_ms()
            n_states = int(t_len.sum().item())
            line["dense_output"] = {"kernel_ms": d_ms, "value": n / (d_ms * 1e-3), "unit": "trajectories/s",
                                    "states_written": n_states, "bytes_written": n_states * 56,
                                    "note": "one launch of nyx_hip_propagate_batch_with_traj_device (until_epoch_with_traj of every run)"}
            del t_ep, t_st, t_len
        if world == 1 and args.config == 2 and not args.no_other_configs and not args.n and not args.hours and args.degree is None:
            # the other GPU configurations and the full-chip launch under the SAME driver clock, one timed pass each
            others = {}
            for key, kw in (("config3", dict(cfg_id=3, cpu=True)), ("config4", dict(cfg_id=4, cpu=True)), ("config5", dict(cfg_id=5, cpu=True)),
                            ("fullchip", dict(cfg_id=2, n=16384, hours=3.0, tag="configs[1]'s force model on a full chip: 16 384 trajectories (256 workgroups, no helpers), 3 h"))):
                others[key] = measure_other(dev=dev, device_index=device_index, lib=lib, **kw)
            # what ONE rank runs when the 10 000-trajectory ensemble of the metric is CUT over 2 / 4 / 8 GPUs (--scaling
            # strong), measured on this one GPU, one timed pass each of the full day (and config 3 at an eighth): since round 6 such a shard
            # runs in the fan-out mode (dedicated helper workgroups on the idle CUs).  The projection below is max rank time = shard time
            # (equal shards) + the step's collectives; it is NOT the headline `value`, which at N > 1 is weak scaling (10 000 per GPU).
            for key, kw in (("shard_5000", dict(cfg_id=2, n=5000)), ("shard_2500", dict(cfg_id=2, n=2500)), ("shard_1250", dict(cfg_id=2, n=1250)),
                            ("config3_shard_625", dict(cfg_id=3, n=625))):
                others[key] = measure_other(dev=dev, device_index=device_index, lib=lib,
                                            tag=f"what one of {10000 // kw['n'] if kw['cfg_id'] == 2 else 8} ranks runs under --scaling strong", **kw)
            line["other_configs"] = others
            coll_ms = gather_ms if gather_ms is not None else 0.0
            one = line["ms_per_step"]
            proj = {}
            for g, key in ((2, "shard_5000"), (4, "shard_2500"), (8, "shard_1250")):
                t = others[key]["ms_per_step"] + coll_ms
                proj[str(g)] = {"rank_ms": others[key]["ms_per_step"], "collectives_ms": coll_ms, "speedup_over_1_gpu": one / t,
                                "value_trajectories_per_s": 10000.0 / (t * 1e-3)}
            line["strong_scaling_projection"] = {
                "what": "configs[1]'s ONE 10 000-trajectory ensemble cut over N GPUs: time of a rank's shard measured on this GPU (equal shards: max over ranks = one "
                        "shard) + the collectives of a step; 1 GPU = this line's ms_per_step",
                "collectives": ("measured in this run (--force-collectives)" if gather_ms is not None else
                                "not in this run (0 ms assumed: one all-gather of n x 7 f64 per rank + one all-reduce of 55 f64, latency-bound; bench.py --force-collectives measures them)"),
                "n_gpus": proj,
                "config3_cut_8_ways_ms": others["config3_shard_625"]["ms_per_step"],
                "headline_value_at_n_gt_1": "weak scaling: every rank propagates its own 10 000-trajectory ensemble (the task's contract for a path that shards: per-GPU work fixed); "
                                            "--scaling strong cuts ONE ensemble N ways"}
        if not args.no_cpu_baseline and world == 1:
            if not w["stm"]:
                cb, sample, ref, samp_h = cpu_baseline(shard, compiled, n, hours)
                if samp_h == hours:
                    # the sample is the head of this rank's shard: check parity on it while we are here
                    got = np.stack([tout[f][: sample.n].cpu().numpy() for f in _abi.F64_FIELDS[:6]], axis=1)
                else:
                    # the CPU sample stops early: propagate the same head for the same span on the device for the comparison
                    o2, s2 = ctx.propagate(sample, int(samp_h * 3600) * nx.NS_PER_S)
                    assert (s2.status == 0).all()
                    got = o2.rv()
                d = got - ref.rv()
                cb["parity_on_sample"] = {"max_dr_m": float(np.linalg.norm(d[:, :3], axis=1).max() * 1e3),
                                          "max_dv_mm_s": float(np.linalg.norm(d[:, 3:], axis=1).max() * 1e6)}
            else:
                cb, ref, ns = cpu_baseline_predict(shard, compiled, p0, end_ns, n, hours)
                got = last["res"]
                scale = np.maximum(np.abs(ref.covar), 1e-6 * np.abs(ref.covar).max(axis=(-2, -1), keepdims=True))
                d = got.states.rv()[:ns] - ref.states.rv()
                cb["parity_on_sample"] = {"max_dr_m": float(np.linalg.norm(d[:, :3], axis=1).max() * 1e3),
                                          "max_dv_mm_s": float(np.linalg.norm(d[:, 3:], axis=1).max() * 1e6),
                                          "max_rel_covar": float((np.abs(got.covar[:ns] - ref.covar) / scale).max())}
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
