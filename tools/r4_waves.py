#!/usr/bin/env python3
"""Workgroup shape (column waves) against the ensemble size on configs[1]'s force model: kernel ms for forced 4 / 8 / 16 waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nyx_amd as nx
import bench
w = bench.workload(2)
compiled = w["prop"].compile(w["almanac"], w["central"])
dur = 3600 * nx.NS_PER_S
for n in [int(x) for x in sys.argv[1].split(",")]:
    b = w["batch"](n, seed=0)
    for waves in (0, 4, 8, 16):
        ctx = nx.GpuContext(compiled)
        if waves:
            ctx.set_column_waves(waves)
        for _ in range(2):
            out, st = ctx.propagate(b, dur)
        ms = ctx.last_kernel_ms(); ev = int(st.n_evals.sum())
        print(f"n={n:7d} waves={waves or 'auto':>4}: {ms:9.3f} ms  frac {ev * w['flop'] / ms / 1e9 / 78.6:.4f}  helpers {ctx.last_coop_helpers()}", flush=True)
        ctx.close()
