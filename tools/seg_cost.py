#!/usr/bin/env python3
"""What a segment launch of the covariance-mapping loop (config 4: 1 000 GEO states, 21x21 + Sun/Moon + SRP, 9x9 STM, quad layout) costs
beyond its force evaluations: the same STM kernel with FIXED 60-s steps, 1 / 2 / 4 / 8 steps per launch - the
slope is a step's sixteen evaluations, the intercept what every launch pays (LDS zeroing and table staging, the first attempt's
barriers, reading and writing the 81-double STMs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nyx_amd as nx
import scenarios as sc
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
prop, almanac, central = sc.leo_full_setup(degree=21, opts=nx.IntegratorOptions.with_fixed_step(60 * nx.NS_PER_S))
ctx = nx.GpuContext(prop.compile(almanac, central, stm=True))
b = bench.geo_batch(n, 0)
b.stm = np.zeros((n, 81)); b.reset_stm()
rows = []
for steps in (1, 2, 4, 8, 16):
    ks = []
    for rep in range(6):
        out, st = ctx.propagate(b, steps * 60 * nx.NS_PER_S)
        ks.append(ctx.last_kernel_ms())
    rows.append((steps, float(np.median(ks[1:])), int(st.n_evals.max())))
    print(f"{steps:2d} fixed 60-s steps per launch: kernel {rows[-1][1] * 1e3:8.1f} us  ({rows[-1][2]} evaluations per trajectory)", flush=True)
x = np.array([r[0] for r in rows], dtype=float); y = np.array([r[1] for r in rows]) * 1e3
slope, icpt = np.polyfit(x, y, 1)
print(f"fit: {slope:.1f} us per step (16 evaluations: {slope / 16 * 1e3:.0f} ns each) + {icpt:.1f} us per launch")
