#!/usr/bin/env python3
"""Calibrated column schedule of one BASELINE configuration (speed weights, duties, window spread) - the numbers behind the cost
model of nyx_hip_tuning_t.schedule = NYX_HIP_SCHED_MODEL.  usage: tools/dump_weights.py <config> [n] [hours] [coop 0|1]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import nyx_amd as nx  # noqa: E402
import bench  # noqa: E402

cfg_id = int(sys.argv[1])
w = bench.workload(cfg_id)
n = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) else w["n"]
hours = float(sys.argv[3]) if len(sys.argv) > 3 and float(sys.argv[3]) else min(w["hours"], 3.0)
coop = int(sys.argv[4]) if len(sys.argv) > 4 else -1
compiled = w["prop"].compile(w["almanac"], w["central"], stm=w["stm"])
b = w["batch"](n, seed=0)
dur = int(round(hours * 3600)) * nx.NS_PER_S
for trial in range(2):
    ctx = nx.GpuContext(compiled, tuning=nx.Tuning(schedule=nx.SCHED_CALIBRATED, cooperative=coop))
    if w["stm"]:
        b.stm = np.zeros((n, 81))
        b.reset_stm()
        res = nx.predict_until(ctx, b, bench.init_covar(n), int(b.epoch_ns[0]) + dur, 60 * nx.NS_PER_S)
        ms = res.kernel_ms
    else:
        ctx.propagate(b, dur)
        ctx.propagate(b, dur)
        ms = ctx.last_kernel_ms()
    out = (C.c_double * 33)()
    ctx._lib.nyx_hip_debug_schedule_weights.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    ctx._lib.nyx_hip_debug_schedule_weights(ctx._h, out)
    lay = (C.c_int32 * 8)()
    ctx._lib.nyx_hip_debug_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    ctx._lib.nyx_hip_debug_layout(ctx._h, lay)
    print(f"config {cfg_id} n={n} coop={coop} helpers={ctx.last_coop_helpers()} waves={lay[0]} pipe={lay[1]}: {ms:.2f} ms, spread {out[32]:.3f}")
    print("  weights {" + ", ".join(f"{x:.3f}" for x in out[:16]) + "}")
    print("  duties  {" + ", ".join(f"{x:.1f}" for x in out[16:32]) + "}")
    ctx.close()
