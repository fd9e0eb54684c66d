#!/usr/bin/env python3
"""How much of a workgroup's run is spent on its slowest lane?  (GPU box)  usage: tools/lane_divergence.py [n] [hours]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx  # noqa: E402
from scenarios import dispersed_leo_batch, leo_full_setup  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
hours = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
prop, almanac, central = leo_full_setup(degree=70)
ctx = nx.GpuContext(prop.compile(almanac, central))
b = dispersed_leo_batch(n, seed=0)
out, st = ctx.propagate(b, int(hours * 3600) * nx.NS_PER_S)
att = (st.n_accepted + st.n_rejected).astype(np.float64)
pad = (-n) % 64
a = np.concatenate([att, np.zeros(pad)]).reshape(-1, 64)
wg_max = a.max(axis=1)
print(f"kernel {ctx.last_kernel_ms():.1f} ms; attempts per lane: mean {att.mean():.1f} min {att.min():.0f} max {att.max():.0f}")
print(f"per workgroup: max-lane attempts mean {wg_max.mean():.1f}, min {wg_max.min():.0f}, max {wg_max.max():.0f}")
print(f"lane efficiency inside workgroups (mean lane / workgroup max): {att.sum() / (wg_max[:, None] * (a > 0)).sum():.4f}")
print(f"workgroup balance (mean workgroup / slowest workgroup): {wg_max.mean() / wg_max.max():.4f}")
srt = np.sort(att)[::-1]
s = np.concatenate([srt, np.zeros(pad)]).reshape(-1, 64)
print(f"if lanes were grouped by attempt count: efficiency {att.sum() / (s.max(axis=1)[:, None] * (s > 0)).sum():.4f}, balance {s.max(axis=1).mean() / s.max():.4f}")
