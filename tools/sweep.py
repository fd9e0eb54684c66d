#!/usr/bin/env python3
"""A/B of tuning variants of ONE BASELINE configuration in ONE process (GPU box): the workload is built once, every variant gets
its own context (nyx_hip_tuning_t, no environment), is launched twice and the second launch is reported - device ms, evaluations,
a bit digest of the results and, with "profile": 1, the in-kernel cycle table of workgroup 0.
usage: tools/sweep.py <config 2|3|4|5> <n or 0> <hours or 0> '<json: {"name": {tuning fields...}, ...}>' [reps]
(a GPU box costs minutes to get and seconds to use: batch the questions)"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import nyx_amd as nx  # noqa: E402
import bench  # noqa: E402

cfg_id = int(sys.argv[1])
w = bench.workload(cfg_id)
n = int(sys.argv[2]) or w["n"]
hours = float(sys.argv[3]) or w["hours"]
variants = json.loads(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
check = int(sys.argv[6]) if len(sys.argv) > 6 else 0   # parity of every variant against the oracle on the first `check` trajectories
compiled = w["prop"].compile(w["almanac"], w["central"], stm=w["stm"])
b = w["batch"](n, seed=0)
dur = int(round(hours * 3600)) * nx.NS_PER_S
ref = None
if check and not w["stm"]:
    import oracle_lib
    ref, _ = oracle_lib.propagate(compiled, b.slice(0, check), dur, n_threads=os.cpu_count())
if w["stm"]:
    b.stm = np.zeros((n, 81))
    b.reset_stm()
print(f"config {cfg_id} n={n} hours={hours:g}", flush=True)
for rep in range(reps):
    for name, fields in variants.items():
        t0 = time.time()
        ctx = nx.GpuContext(compiled, tuning=nx.Tuning(**fields))
        if w["stm"]:
            for _ in range(2):
                res = nx.predict_until(ctx, b, bench.init_covar(n), int(b.epoch_ns[0]) + dur, 60 * nx.NS_PER_S)
            ms, st, arrs, ne = res.kernel_ms, res.stats, [res.states.rv(), res.covar], 16
        else:
            for _ in range(2):
                out, st = ctx.propagate(b, dur)
            ms, arrs, ne = ctx.last_kernel_ms(), [out.rv(), out.epoch_ns], int(st.n_evals[:64].max())
        h = hashlib.sha256()
        for a in arrs + [st.n_evals, st.n_rejected]:
            h.update(np.ascontiguousarray(a).tobytes())
        ev = int(st.n_evals.sum())
        print(f"{name:28s} {ms:9.3f} ms  evals {ev}  bad {(st.status != 0).sum()}  helpers {ctx.last_coop_helpers() if not w['stm'] else 0}  "
              f"frac {ev * w['flop'] / ms / 1e9 / bench.FP64_VECTOR_PEAK_TFLOPS:.4f}  digest {h.hexdigest()[:12]}  ({time.time() - t0:.1f} s)", flush=True)
        if ref is not None:
            d = out.rv()[:check] - ref.rv()
            print(f"   parity vs oracle on {check}: max dr {np.linalg.norm(d[:, :3], axis=1).max() * 1e6:.4f} mm, max dv {np.linalg.norm(d[:, 3:], axis=1).max() * 1e6:.3e} mm/s", flush=True)
        if fields.get("profile"):
            buf = (C.c_int64 * 136)()
            ctx._lib.nyx_hip_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
            if ctx._lib.nyx_hip_debug_profile(ctx._h, buf) == 0:
                p = np.array(buf[:]).reshape(17, 8)
                print("   mailbox: answers %d fallbacks %d fb_seq_sum %d posted %d helper_jobs %d" % tuple(p[16, :5]))
                print("   wg0 cycles per eval (phaseA duty harmonics phaseC stepctl | total barrier-wait), clock %.0f MHz, %d evals" % (p[0, 5] / max(p[0, 7], 1) * 100.0, ne))
                for wv in range(16):
                    if p[wv, 5]:
                        print(f"    wave {wv:2d}: " + " ".join(f"{p[wv, q] / ne:8.0f}" for q in (0, 1, 2, 3, 4)) + f" | {p[wv, 5] / ne:8.0f} {p[wv, 6] / ne:8.0f}")
        ctx.close()
