#!/usr/bin/env python3
"""Hill climbing over the explicit per-wave column weights of one BASELINE configuration (GPU box): perturb, keep what is faster.
usage: tools/_exp/hill.py <config> <n or 0> <hours or 0> <evaluations> [w0,...,w15] ['{"tuning": fields}']"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import nyx_amd as nx  # noqa: E402
import bench  # noqa: E402

cfg_id = int(sys.argv[1])
w_ = bench.workload(cfg_id)
n = int(sys.argv[2]) or w_["n"]
hours = float(sys.argv[3]) or w_["hours"]
evals = int(sys.argv[4])
w0 = np.array([float(x) for x in sys.argv[5].split(",")])
extra = json.loads(sys.argv[6]) if len(sys.argv) > 6 else {}
compiled = w_["prop"].compile(w_["almanac"], w_["central"], stm=w_["stm"])
b = w_["batch"](n, seed=0)
dur = int(round(hours * 3600)) * nx.NS_PER_S
if w_["stm"]:
    b.stm = np.zeros((n, 81))
    b.reset_stm()


def run(w, reps=3):
    ctx = nx.GpuContext(compiled, tuning=nx.Tuning(schedule=nx.SCHED_EXPLICIT, wave_weights=list(w), **extra))
    best = 1e9
    for _ in range(reps):
        if w_["stm"]:
            res = nx.predict_until(ctx, b, bench.init_covar(n), int(b.epoch_ns[0]) + dur, 60 * nx.NS_PER_S)
            ms = res.kernel_ms
        else:
            ctx.propagate(b, dur)
            ms = ctx.last_kernel_ms()
        best = min(best, ms)
    rows = (C.c_int32 * 16)()
    ctx._lib.nyx_hip_debug_schedule_rows.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    sched = 1 if (not w_["stm"] and ctx.last_coop_helpers() > 0) else 0
    ctx._lib.nyx_hip_debug_schedule_rows(ctx._h, sched, rows, None)
    ctx.close()
    return best, tuple(rows[:])


rng = np.random.default_rng(1)
cur, (cur_ms, cur_rows) = w0.copy(), run(w0)
print(f"start {cur_ms:.3f} ms rows {cur_rows}", flush=True)
seen = {cur_rows: cur_ms}
step = float(os.environ.get("HILL_STEP", "0.15"))
for it in range(evals):
    cand = cur.copy()
    for k in rng.choice(np.arange(1, 16), size=rng.integers(1, 4), replace=False):
        cand[k] = max(0.05, cand[k] * (1.0 + step * rng.standard_normal()))
    ms, rows = run(cand)
    tag = ""
    if ms < cur_ms * 0.998:
        cur, cur_ms, cur_rows, tag = cand, ms, rows, " *"
    print(f"{it:3d} {ms:8.3f} ms{tag} rows {rows}", flush=True)
print("best %.3f ms: %s\n rows %s" % (cur_ms, ",".join(f"{x:.3f}" for x in cur), cur_rows))
