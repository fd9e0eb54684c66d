#!/usr/bin/env python3
"""Latency of repeated short calls (the OD pattern: 1-minute STM segments with Phi reset, od/process/mod.rs:466-483)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx
from scenarios import dispersed_leo_batch, leo_full_setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 21
stm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prop, almanac, central = leo_full_setup(degree=deg)
ctx = nx.GpuContext(prop.compile(almanac, central, stm=bool(stm)))
b = dispersed_leo_batch(n, seed=0)
if stm:
    b.stm = np.zeros((n, 81)); b.reset_stm()
ks, ws = [], []
for i in range(12):
    t0 = time.perf_counter()
    b, st = ctx.propagate(b, 60 * nx.NS_PER_S)
    ws.append((time.perf_counter() - t0) * 1e3)
    ks.append(ctx.last_kernel_ms())
    if stm:
        b.reset_stm()
print(f"n={n} deg={deg} stm={stm}: per 60-s segment kernel ms {np.round(ks, 2)}, wall ms {np.round(ws, 2)}")
