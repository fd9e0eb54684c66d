#!/usr/bin/env python3
"""Calibrates the per-wave column weights of the 16-wave schedule on the GPU: runs the north-star force model with the
in-kernel cycle accounting, reads how long each wave of workgroup 0 spends in its window (role duty + harmonics), and
moves weight from the late waves to the early ones.  usage: tools/tune_wave_weights.py [iterations] [coop 0/1]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
os.environ["NYX_HIP_COOP"] = sys.argv[2] if len(sys.argv) > 2 else "1"
os.environ["NYX_HIP_PROFILE"] = "1"
os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx  # noqa: E402
from scenarios import dispersed_leo_batch, leo_full_setup  # noqa: E402

prop, almanac, central = leo_full_setup(degree=70)
compiled = prop.compile(almanac, central)
batch = dispersed_leo_batch(10_000, seed=0)
dur = 2 * 3600 * nx.NS_PER_S
w = np.array([1.0] + [1.3 * 1.0, 1.3, 1.3] + [1.1 * 1.25, 1.1, 1.1, 1.1] + [0.9 * 1.25, 0.9, 0.9, 0.9] + [0.7 * 1.25, 0.7, 0.7, 0.7])
if len(sys.argv) > 3:
    w = np.array([float(x) for x in sys.argv[3].split(",")])
best = (1e9, None)
for it in range(iters):
    os.environ["NYX_HIP_WAVE_WEIGHTS"] = ",".join(f"{x:.4f}" for x in w)
    ctx = nx.GpuContext(compiled)
    ms = []
    for rep in range(2):
        out, st = ctx.propagate(batch, dur)
        ms.append(ctx.last_kernel_ms())
    buf = (C.c_int64 * 136)()
    ctx._lib.nyx_hip_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    ctx._lib.nyx_hip_debug_profile(ctx._h, buf)
    p = np.array(buf[:]).reshape(17, 8)[:16].astype(float)
    ne = float(st.n_evals[:64].max())
    t = (p[:, 1] + p[:, 2]) / ne          # duty + harmonics per evaluation
    t[0] = np.nan
    print(f"iter {it}: kernel {min(ms):.2f} ms; window per wave (k cycles): " + " ".join(f"{x/1e3:.1f}" for x in t[1:]) +
          f" | spread {np.nanmax(t) - np.nanmin(t):.0f}")
    print("   weights " + os.environ["NYX_HIP_WAVE_WEIGHTS"])
    if min(ms) < best[0]:
        best = (min(ms), os.environ["NYX_HIP_WAVE_WEIGHTS"])
    mean = np.nanmean(t[1:])
    w[1:] = w[1:] * (mean / t[1:]) ** 0.6
    ctx.close()
print("best", best)
