#!/usr/bin/env python3
"""Bisect helper: D3 vs quad vs oracle on a small STM case (deg, waves from argv)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx
import oracle_lib
from scenarios import leo_full_setup
from test_gpu_stm_quad import stm_batch, run
deg = int(sys.argv[1]); waves = int(sys.argv[2]); S = nx.NS_PER_S
prop, almanac, central = leo_full_setup(degree=deg, opts=nx.IntegratorOptions.with_fixed_step_s(30.0))
compiled = prop.compile(almanac, central, stm=True)
b = stm_batch(37, seed=5 + deg)
dur = 600 * S
ref, _ = oracle_lib.propagate(compiled, b, dur, n_threads=8)
for layout in (0, 1):
    out, st, ms = run(compiled, b, dur, layout, waves)
    d = out.rv() - ref.rv()
    a, r = out.stm.reshape(-1, 81), ref.stm.reshape(-1, 81)
    scale = np.maximum(np.abs(r), 1e-6 * np.abs(r).max(axis=1, keepdims=True))
    print(f"  deg {deg} waves {waves} layout {layout}: status {int((st.status != 0).sum())} bad, dr vs oracle {np.linalg.norm(d[:, :3], axis=1).max():.3e} km, "
          f"Phi rel {(np.abs(a - r) / scale).max():.3e}, {ms:.2f} ms")
