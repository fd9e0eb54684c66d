#!/usr/bin/env python3
"""Ad-hoc timing of the device path (GPU box): kernel ms, force evaluations/s, parity vs oracle on a few runs.
usage: tools/time_gpu.py [n] [hours] [degree] [waves] [check]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx  # noqa: E402
from scenarios import dispersed_leo_batch, leo_full_setup, pos_vel_errors  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
hours = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
degree = int(sys.argv[3]) if len(sys.argv) > 3 else 70
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 0
check = int(sys.argv[5]) if len(sys.argv) > 5 else 0

prop, almanac, central = leo_full_setup(degree=degree)
compiled = prop.compile(almanac, central)
ctx = nx.GpuContext(compiled)
if waves:
    ctx.set_column_waves(waves)
batch = dispersed_leo_batch(n, seed=0)
dur = int(hours * 3600) * nx.NS_PER_S
t0 = time.time()
out, st = ctx.propagate(batch, dur)
wall = time.time() - t0
ms = ctx.last_kernel_ms()
ev = int(st.n_evals.sum())
print(f"n={n} hours={hours} deg={degree} waves={waves or 'auto'}: kernel {ms:.1f} ms (wall {wall*1e3:.1f}), evals {ev} -> {ev/ms*1e3:.3e} evals/s, "
      f"{n/ms*1e3*(24/hours):.1f} traj-days/s equiv, acc {st.n_accepted.sum()} rej {st.n_rejected.sum()} status!=0: {(st.status!=0).sum()}")
import ctypes as C
lay = (C.c_int32 * 8)()
ctx._lib.nyx_hip_debug_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
if ctx._lib.nyx_hip_debug_layout(ctx._h, lay) == 0:
    print("  layout: waves %d pipelined %d carried epoch fields %d chained attempts %d records in LDS %d (%d doubles) almanac waves %d LDS %d B" % tuple(lay))
if os.environ.get("NYX_HIP_PROFILE"):
    buf = (C.c_int64 * 136)()
    ctx._lib.nyx_hip_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    if ctx._lib.nyx_hip_debug_profile(ctx._h, buf) == 0:
        p = np.array(buf[:]).reshape(17, 8)
        print('  mailbox: answers %d fallbacks %d fb_seq_sum %d posted %d helper_jobs %d' % tuple(p[16, :5]))
        ne = int(st.n_evals[:64].max())
        print("  wg0 cycles per eval (phaseA, duty, harmonics, phaseC, stepctl | total, barrier-wait) clock %.0f MHz" % (p[0, 5] / max(p[0, 7], 1) * 100.0))
        for w in range(16):
            if p[w, 5]:
                print(f"   wave {w:2d}: " + " ".join(f"{p[w, q] / ne:9.0f}" for q in (0, 1, 2, 3, 4)) + f" | {p[w, 5] / ne:9.0f} {p[w, 6] / ne:9.0f}")
if check:
    import oracle_lib
    sub = batch.slice(0, check)
    ref, _ = oracle_lib.propagate(compiled, sub, dur, n_threads=os.cpu_count())
    dr, dv = pos_vel_errors(out.slice(0, check), ref)
    print(f"  parity on {check}: max dr {dr.max()*1e3:.3e} m, max dv {dv.max()*1e6:.3e} mm/s")
