#!/usr/bin/env python3
"""Fits the per-wave speed weights of the column schedule on the GPU box (what the frozen tables model_coop / model_solo / ... of
fill_schedule() in nyx_amd/csrc/abi.cpp hold): runs one BASELINE configuration with explicit weights and the in-kernel cycle
accounting, reads every wave's window (role duty + its columns) and the rows it walked, solves for the row counts that would make
all windows equal, and feeds them back as weights (damped); the best kernel time wins.  One process, a few seconds.
usage: tools/tune_schedule.py <config 2|4|5> <n or 0> <hours or 0> [iterations] ['{"tuning": fields}'] [w0,w1,...,w15]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import nyx_amd as nx  # noqa: E402
import bench  # noqa: E402

cfg_id = int(sys.argv[1])
w_ = bench.workload(cfg_id)
n = int(sys.argv[2]) or w_["n"]
hours = float(sys.argv[3]) or w_["hours"]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 12
extra = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
compiled = w_["prop"].compile(w_["almanac"], w_["central"], stm=w_["stm"])
b = w_["batch"](n, seed=0)
dur = int(round(hours * 3600)) * nx.NS_PER_S
if w_["stm"]:
    b.stm = np.zeros((n, 81))
    b.reset_stm()


def run(tuning):
    ctx = nx.GpuContext(compiled, tuning=tuning)
    lib = ctx._lib
    if w_["stm"]:
        for _ in range(2):
            res = nx.predict_until(ctx, b, bench.init_covar(n), int(b.epoch_ns[0]) + dur, 60 * nx.NS_PER_S)
        ms, ne = res.kernel_ms, 16.0
    else:
        for _ in range(2):
            out, st = ctx.propagate(b, dur)
        ms, ne = ctx.last_kernel_ms(), float(st.n_evals[:64].max())
    buf = (C.c_int64 * 136)()
    lib.nyx_hip_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.nyx_hip_debug_profile(ctx._h, buf)
    p = np.array(buf[:]).reshape(17, 8)[:16].astype(float) / ne
    rows = (C.c_int32 * 16)()
    duties = (C.c_double * 3)()
    lib.nyx_hip_debug_schedule_rows.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    sched = 1 if (not w_["stm"] and ctx.last_coop_helpers() > 0) else 0
    lib.nyx_hip_debug_schedule_rows(ctx._h, sched, rows, duties)
    ctx.close()
    return ms, p, np.array(rows[:], dtype=float), np.array(duties[:])


ms0, p, rows, duties = run(nx.Tuning(profile=1, **extra))
print(f"config {cfg_id} n={n} hours={hours:g}: model schedule {ms0:.3f} ms; duties (model) {duties}")
print("  rows   " + " ".join(f"{x:5.0f}" for x in rows))
print("  window " + " ".join(f"{(p[k, 1] + p[k, 2]) / 1e3:5.1f}" for k in range(16)))
# role of each wave is not exposed: a wave with a window duty > 500 cycles is a role wave, charged the model duty of its kind by rank
w = None
if len(sys.argv) > 6:
    w = np.array([float(x) for x in sys.argv[6].split(",")])
best = (ms0, None)
for it in range(iters):
    duty, harm = p[:, 1].copy(), p[:, 2].copy()
    act = rows > 0
    cpr = np.where(act, harm / np.maximum(rows, 1.0), np.nan)          # cycles per row as THIS wave sees them
    cpr = np.where(np.isnan(cpr), np.nanmean(cpr), cpr)
    total = rows.sum()
    lo, hi = 0.0, 10.0 * (duty + harm).max()
    for _ in range(80):
        T = 0.5 * (lo + hi)
        want = np.maximum(0.0, (T - duty) / cpr)
        want[0] = 0.0
        if want.sum() < total:
            lo = T
        else:
            hi = T
    hc = np.zeros(16)
    role = np.argsort(-duty[1:])[:2] + 1                                 # the two role waves beside the integrator
    alm, prt = (role[0], role[1]) if duty[role[0]] >= duty[role[1]] else (role[1], role[0])
    if duty[alm] > 500:
        hc[alm] = duties[1]
    if duty[prt] > 500:
        hc[prt] = duties[2]
    target = want + hc
    target[0] = 1.0
    target = target / target[1:].mean()
    w = target if w is None else w ** 0.5 * target ** 0.5
    w[0] = 1.0
    ms, p, rows, _ = run(nx.Tuning(schedule=nx.SCHED_EXPLICIT, wave_weights=list(w), profile=1, **extra))
    win = p[:, 1] + p[:, 2]
    print(f"iter {it:2d}: {ms:8.3f} ms  T* {T / 1e3:5.1f}k  spread {(win[1:].max() - win[1:].min()) / 1e3:4.1f}k  weights " + ",".join(f"{x:.3f}" for x in w))
    print("         rows   " + " ".join(f"{x:5.0f}" for x in rows))
    print("         window " + " ".join(f"{x / 1e3:5.1f}" for x in win))
    if ms < best[0]:
        best = (ms, ",".join(f"{x:.3f}" for x in w))
print(f"best {best[0]:.3f} ms (model {ms0:.3f}): {best[1]}")
