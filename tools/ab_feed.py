#!/usr/bin/env python3
"""A/B of the harmonics table feed on the GPU box: scalar stream (0) against the hybrid scalar + DPP feed (1) - bit identity of
the final states and kernel time.  usage: tools/ab_feed.py [config 2|5] [n] [hours] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx  # noqa: E402
import bench  # noqa: E402

cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = bench.workload(cfg_id)
n = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) else w["n"]
hours = float(sys.argv[3]) if len(sys.argv) > 3 and float(sys.argv[3]) else w["hours"]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
compiled = w["prop"].compile(w["almanac"], w["central"], stm=w["stm"])
b = w["batch"](n, seed=0)
dur = int(round(hours * 3600)) * nx.NS_PER_S
res = {}
for feed in (0, 1):
    os.environ["NYX_HIP_HARM_FEED"] = str(feed)
    os.environ["NYX_HIP_CALIBRATE"] = os.environ.get("AB_CALIBRATE", "0")  # same (structural) schedule on both sides: same summation order
    ctx = nx.GpuContext(compiled)
    ms = []
    for _ in range(reps):
        out, st = ctx.propagate(b, dur)
        ms.append(ctx.last_kernel_ms())
    res[feed] = (out, st, ms)
    print(f"feed {feed}: kernel ms {' '.join(f'{m:.2f}' for m in ms)}; evals {int(st.n_evals.sum())}, bad {(st.status != 0).sum()}")
    ctx.close()
o0, o1 = res[0][0], res[1][0]
same = np.array_equal(o0.rv(), o1.rv()) and np.array_equal(o0.epoch_ns, o1.epoch_ns)
dr = np.linalg.norm((o0.rv() - o1.rv())[:, :3], axis=1).max()
print(f"bit-identical final states: {same} (max |dr| {dr * 1e6:.3e} mm); speed-up {min(res[0][2]) / min(res[1][2]):.3f}x")
