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
        fields = dict(fields)
        env = fields.pop("env", None)   # experiment knobs that travel through the environment (read at ctx_create, only with NYX_HIP_TUNING_ENV)
        show = fields.pop("show_sched", 0)
        saved = {}
        if env:
            env = dict(env, NYX_HIP_TUNING_ENV="1")
            for k, v in env.items():
                saved[k] = os.environ.get(k)
                os.environ[k] = str(v)
        ctx = nx.GpuContext(compiled, tuning=nx.Tuning(**fields))
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if w["stm"]:
            for _ in range(2):
                res = nx.predict_until(ctx, b, bench.init_covar(n), int(b.epoch_ns[0]) + dur, 60 * nx.NS_PER_S)
            ms, st, arrs, ne = res.kernel_ms, res.stats, [res.states.rv(), res.covar], int(st_max) if (st_max := res.stats.n_evals[:16].max()) else 16   # (the accounting sums over every segment of the loop)
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
        if show:
            rows = (C.c_int32 * 16)()
            ctx._lib.nyx_hip_debug_schedule_rows.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
            for sc_id, nm in ((0, "solo"), (1, "owner"), (2, "helper")):
                ctx._lib.nyx_hip_debug_schedule_rows(ctx._h, sc_id, rows, None)
                print(f"   {nm} rows/wave " + " ".join(f"{x:4d}" for x in rows[:]) + f"  sum {sum(rows[:])}")
        if show:
            ro = (C.c_int32 * 32)()
            if hasattr(ctx._lib, "nyx_hip_debug_roles"):
                ctx._lib.nyx_hip_debug_roles.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
                ctx._lib.nyx_hip_debug_roles(ctx._h, ro)
                print("   roles (kind:mask) " + " ".join(f"{ro[w]}:{ro[16 + w]:x}" for w in range(16)))
        if fields.get("profile"):
            buf = (C.c_int64 * 136)()
            ctx._lib.nyx_hip_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
            if ctx._lib.nyx_hip_debug_profile(ctx._h, buf) == 0:
                p = np.array(buf[:]).reshape(17, 8)
                print("   mailbox: answers %d fallbacks %d fb_seq_sum %d posted %d helper_jobs %d paired_passes %d" % tuple(p[16, :6]))
                print("   wg0 cycles per eval (phaseA duty harmonics phaseC stepctl | total barrier-wait), clock %.0f MHz, %d evals" % (p[0, 5] / max(p[0, 7], 1) * 100.0, ne))
                for wv in range(16):
                    if p[wv, 5]:
                        print(f"    wave {wv:2d}: " + " ".join(f"{p[wv, q] / ne:8.0f}" for q in (0, 1, 2, 3, 4)) + f" | {p[wv, 5] / ne:8.0f} {p[wv, 6] / ne:8.0f}")
            hb = (C.c_int64 * 152)()
            if hasattr(ctx._lib, "nyx_hip_debug_profile_helper"):
                ctx._lib.nyx_hip_debug_profile_helper.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
                if ctx._lib.nyx_hip_debug_profile_helper(ctx._h, hb) == 0:
                    hp = np.array(hb[:]).reshape(19, 8)
                    sgv = hp[17:19].reshape(-1)[:16]
                    if sgv.sum():
                        names = ['back-edge', 'phase A', 'next position', 'DCM wait', 'rotate+inputs', 'post', 'two-body+sums', 'barrier', 'C to fold', 'answer', 'C rest', 'step ctl (rest)', 'sc cold state', 'sc sums', 'sc decide', 'sc open next']
                        print('   integrator per eval: ' + ', '.join(f'{n} {v / ne:.0f}' for n, v in zip(names, sgv)))
                    if w["stm"] and hp[1].sum():
                        segs = max(1, int(st.n_evals[:16].max()) // 16)
                        print('   integrator boundary per segment: ' + ', '.join(f'{n} {v / segs:.0f}' for n, v in zip(['step control', 'open next', 'B0', 'time updates', 're-arm', 'epoch data + Bp', 'phase A + B1'], hp[1, :7])) + f'  ({segs} segments)')
                    if hp[16, 3]:
                        print(f"   owner latency loop (wg0, per posted job): wait for the answer {hp[16, 0] / hp[16, 3]:.0f}, answer in hand -> post {hp[16, 1] / hp[16, 3]:.0f}, post {hp[16, 2] / hp[16, 3]:.0f} cycles ({hp[16, 3]} jobs)")
                    if hp[0, 2]:
                        print(f"   first helper: producer {hp[0, 2]} jobs, per job: wait-for-slot {hp[0, 0] / hp[0, 2]:.0f}, scan+claim+fetch {hp[0, 1] / hp[0, 2]:.0f}, lost claims {hp[0, 3] / hp[0, 2]:.2f}; total {hp[0, 5] / hp[0, 2]:.0f} cycles/job")
                        for wv in range(1, 16):
                            if hp[wv, 2]:
                                print(f"    hwave {wv:2d}: busy {hp[wv, 0] / hp[wv, 2]:8.0f}  wait {hp[wv, 1] / hp[wv, 2]:8.0f}  per job ({hp[wv, 2]} jobs)")
        ctx.close()
