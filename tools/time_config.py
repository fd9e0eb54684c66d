#!/usr/bin/env python3
"""Kernel time and in-kernel cycle accounting (NYX_HIP_PROFILE=1) of one BASELINE configuration on the GPU box.
usage: tools/time_config.py <config 2|3|4|5> [n] [hours] [waves]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx  # noqa: E402
import bench  # noqa: E402

cfg_id = int(sys.argv[1])
w = bench.workload(cfg_id)
n = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) else w["n"]
hours = float(sys.argv[3]) if len(sys.argv) > 3 and float(sys.argv[3]) else w["hours"]
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 0
compiled = w["prop"].compile(w["almanac"], w["central"], stm=w["stm"])
ctx = nx.GpuContext(compiled)
if waves:
    ctx.set_column_waves(waves)
b = w["batch"](n, seed=0)
dur = int(round(hours * 3600)) * nx.NS_PER_S
if w["stm"]:
    b.stm = np.zeros((n, 81))
    b.reset_stm()
    res = nx.predict_until(ctx, b, bench.init_covar(n), int(b.epoch_ns[0]) + dur, 60 * nx.NS_PER_S)
    res = nx.predict_until(ctx, b, bench.init_covar(n), int(b.epoch_ns[0]) + dur, 60 * nx.NS_PER_S)
    ms, st = res.kernel_ms, res.stats
    evals_last_launch = 16
else:
    out, st = ctx.propagate(b, dur)
    out, st = ctx.propagate(b, dur)
    ms = ctx.last_kernel_ms()
    evals_last_launch = int(st.n_evals[:64].max())
ev = int(st.n_evals.sum())
if os.environ.get("NYX_DIGEST"):  # bit-level digest of the results: compare builds (NYX_HIP_LIB) or switches across processes
    import hashlib
    h = hashlib.sha256()
    for a in ([res.states.rv(), res.covar] if w["stm"] else [out.rv(), out.epoch_ns]) + [st.n_evals, st.n_rejected, st.last_error]:
        h.update(np.ascontiguousarray(a).tobytes())
    print(f"digest {h.hexdigest()[:16]}")
print(f"config {cfg_id}: n={n} hours={hours:g} waves={waves or 'auto'}: device {ms:.2f} ms, evals {ev} -> {ev / ms * 1e3:.3e} evals/s, "
      f"{ms * 1e3 / max(ev / n, 1):.2f} us per evaluation per trajectory-lane, acc {int(st.n_accepted.sum())} rej {int(st.n_rejected.sum())} "
      f"bad {(st.status != 0).sum()}, algorithmic {ev * w['flop'] / ms / 1e9:.2f} TFLOP/s, helpers {ctx.last_coop_helpers()}")
wb = (C.c_double * 17)()
ctx._lib.nyx_hip_debug_weights.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
if ctx._lib.nyx_hip_debug_weights(ctx._h, wb) == 0:
    print("  column weights " + " ".join(f"{x:.2f}" for x in wb[:16]) + f" | window spread at calibration {wb[16]:.3f}")
if os.environ.get("NYX_HIP_PROFILE"):
    buf = (C.c_int64 * 136)()
    ctx._lib.nyx_hip_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    if ctx._lib.nyx_hip_debug_profile(ctx._h, buf) == 0:
        p = np.array(buf[:]).reshape(17, 8)
        ne = evals_last_launch
        print("  wg0 cycles per eval (phaseA, duty, harmonics, phaseC, stepctl | total, barrier-wait) clock %.0f MHz, %d evals" %
              (p[0, 5] / max(p[0, 7], 1) * 100.0, ne))
        print(f"  (phase C of the quad layout: fold of the partial sums done after {p[16, 5] / ne:.0f} cycles, function {p[16, 6] / ne:.0f} cycles)")
        for wv in range(16):
            if p[wv, 5]:
                print(f"   wave {wv:2d}: " + " ".join(f"{p[wv, q] / ne:9.0f}" for q in (0, 1, 2, 3, 4)) + f" | {p[wv, 5] / ne:9.0f} {p[wv, 6] / ne:9.0f}")
