#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native propagation path.

Metric (BASELINE.json): propagated trajectories / second, 10 000-trajectory LEO Monte Carlo,
70x70 gravity (JGM3, the model present in the reference checkout) + Sun/Moon point masses +
cannonball SRP with Earth shadow, RK89 default options, 1 day  (configs[1]).

A "step" is one pass of the hot path over one batch: every rank propagates its shard of
dispersed states (already resident in HBM) for one day through the C-ABI
(`nyx_hip_propagate_batch_device`), then the ranks exchange the final states with one RCCL
all-gather (the Monte Carlo result collection of north_star).  Weak scaling: the per-GPU ensemble
is fixed (--n, default 10 000), `value` = all ranks' trajectories / max-over-ranks time.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import nyx_amd as nx  # noqa: E402
from nyx_amd import _abi  # noqa: E402
from scenarios import dispersed_leo_batch, leo_full_setup  # noqa: E402

FLOP_PER_EVAL = 7.3e4          # config 2, algorithmic FLOP per force-model evaluation
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X vector FP64 (= matrix FP64) peak, SURVEY 8(d) / MI355X_MICROARCH.md clocks
HBM_PEAK_GBPS = 8000.0
BYTES_PER_TRAJ = 2 * 13 * 8 + 8 * 8  # 13 f64 + epoch in and out, plus status/details/counters


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


def cpu_baseline(compiled, n_per_gpu, hours, seed):
    """The oracle (CPU restatement of the reference, kind = "port") timed on this box's host cores on a bounded
    sample of the SAME workload: full-length trajectories, as many as fit in ~20 s."""
    import oracle_lib
    cores = os.cpu_count() or 1
    probe = dispersed_leo_batch(cores, seed=seed)
    t0 = time.time()
    oracle_lib.propagate(compiled, probe, int(0.25 * 3600) * nx.NS_PER_S, n_threads=cores)
    per_traj_hour = (time.time() - t0) / 0.25  # seconds of wall per (cores trajectories) per hour of propagation
    # one round = `cores` full-length trajectories, one per thread (~13 s for 24 h on this class of host); the short probe
    # underestimates it (caches, clocks), so at most two rounds: 10-30 s of CPU work whatever the probe says
    budget_s = 25.0
    rounds = max(1, min(2, int(budget_s / max(per_traj_hour * hours, 1e-3))))
    n = cores * rounds
    sample = dispersed_leo_batch(n, seed=seed)
    t0 = time.time()
    out, st = oracle_lib.propagate(compiled, sample, int(hours * 3600) * nx.NS_PER_S, n_threads=cores)
    dt = time.time() - t0
    assert (st.status == 0).all()
    return {"value": n / dt, "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"{n} of the {n_per_gpu} dispersed LEO states, full {hours:g} h propagation each, {cores} pthreads "
                      f"(rayon par_iter analogue), {dt:.1f} s wall, {int(st.n_evals.sum())} force evaluations",
            "evals_per_s": float(st.n_evals.sum()) / dt}, sample, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=10_000, help="trajectories per GPU")
    ap.add_argument("--hours", type=float, default=24.0)
    ap.add_argument("--degree", type=int, default=70)
    ap.add_argument("--waves", type=int, default=0, help="column-split waves per workgroup (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-output", action="store_true", help="skip the extra launch with the trajectories recorded")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        dist.init_process_group("nccl", device_id=dev)

    prop, almanac, central = leo_full_setup(degree=args.degree)
    compiled = prop.compile(almanac, central)
    ctx = nx.GpuContext(compiled, device=local_rank)
    if args.waves:
        ctx.set_column_waves(args.waves)
    lib = _abi.load_library()

    # contiguous index shards of ONE ensemble: rank r owns trajectories [r*n, (r+1)*n) (seed = global stream)
    full = dispersed_leo_batch(args.n * world, seed=0)
    shard = full.slice(rank * args.n, (rank + 1) * args.n)
    tin, sin = tensor_states(shard, dev)
    tout, sout = tensor_states(shard, dev)
    tst, sst = tensor_stats(args.n, dev)
    dur_ns = int(args.hours * 3600) * nx.NS_PER_S
    stream = torch.cuda.current_stream(dev)
    gathered = [torch.empty((args.n, 7), dtype=torch.float64, device=dev) for _ in range(world)] if world > 1 else None

    def step():
        rc = lib.nyx_hip_propagate_batch_device(ctx._h, C.byref(sin), dur_ns, C.byref(sout), C.byref(sst), C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(_abi.last_error())
        if world > 1:  # final-state collection over RCCL/xGMI (one all-gather, latency-bound: n x 7 f64)
            final = torch.stack([tout[f] for f in _abi.F64_FIELDS[:6]] + [tout["epoch_ns"].to(torch.float64)], dim=1)
            dist.all_gather(gathered, final)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(ctx.last_kernel_ms())  # HIP events recorded on the launch stream around the kernel
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    n_evals = int(tst["n_evals"].sum().item())
    n_bad = int((tst["status"] != 0).sum().item())
    n_acc, n_rej = int(tst["n_accepted"].sum().item()), int(tst["n_rejected"].sum().item())
    if n_bad:
        raise SystemExit(f"{n_bad} trajectories failed")
    k_ms = float(np.mean(kernel_ms))
    total_traj = args.n * world
    value = total_traj * args.steps / elapsed

    if rank == 0:
        flops = n_evals * FLOP_PER_EVAL
        achieved_tf = flops / (k_ms * 1e-3) / 1e12
        hbm_gbps = args.n * BYTES_PER_TRAJ / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "round01_hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("n") == args.n and tj.get("hours") == args.hours and tj.get("degree") == args.degree:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "propagated trajectories/sec (10k-ensemble, 1-day RK89)",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: {args.n}-trajectory LEO Monte Carlo per GPU, {args.degree}x{args.degree} JGM3 gravity + "
                                   f"Sun/Moon point masses + cannonball SRP (Earth shadow), RK89 default options, {args.hours:g} h",
                       "trajectories_per_gpu": args.n, "column_waves": args.waves or "auto",
                       "sharding": "contiguous index shards, no data-path collective; one RCCL all-gather of final states per step"},
            "force_evals_per_s": n_evals * world / (elapsed / args.steps),
            "force_evals_per_launch": n_evals, "accepted_steps": n_acc, "rejected_attempts": n_rej,
            "kernel_ms": k_ms,
            "roofline": {"bound": "valu_fp64", "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                         "algorithmic_flop_per_eval": FLOP_PER_EVAL,
                         "hbm": {"achieved": hbm_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_gbps / HBM_PEAK_GBPS,
                                 "algorithmic_bytes_per_trajectory": BYTES_PER_TRAJ}},
        }
        if world == 1 and not args.no_dense_output:
            # The reference's Monte Carlo runs `until_epoch_with_traj` (mc/montecarlo.rs:236-239): the same launch with the
            # dense output on (every accepted state of every run appended in HBM), timed once, reported beside the headline.
            cap = int(tst["n_accepted"].max().item()) + 2
            t_ep = torch.zeros((cap, args.n), dtype=torch.int64, device=dev)
            t_st = torch.zeros((6, cap, args.n), dtype=torch.float64, device=dev)
            t_len = torch.zeros(args.n, dtype=torch.int32, device=dev)
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
            line["dense_output"] = {"kernel_ms": d_ms, "value": args.n / (d_ms * 1e-3), "unit": "trajectories/s",
                                    "states_written": n_states, "bytes_written": n_states * 56,
                                    "note": "one launch of nyx_hip_propagate_batch_with_traj_device (until_epoch_with_traj of every run)"}
            del t_ep, t_st, t_len
        if not args.no_cpu_baseline and world == 1:
            cb, sample, ref = cpu_baseline(compiled, args.n, args.hours, seed=0)
            # the sample is the head of this rank's shard: check parity on it while we are here
            got = np.stack([tout[f][: sample.n].cpu().numpy() for f in _abi.F64_FIELDS[:6]], axis=1)
            d = got - ref.rv()
            cb["parity_on_sample"] = {"max_dr_m": float(np.linalg.norm(d[:, :3], axis=1).max() * 1e3),
                                      "max_dv_mm_s": float(np.linalg.norm(d[:, 3:], axis=1).max() * 1e6)}
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
