#!/usr/bin/env python3
"""Ad-hoc timing of a drag configuration (GPU box): usage tools/time_drag.py [n] [hours] [model]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx  # noqa: E402
from scenarios import dispersed_leo_batch, leo_full_setup  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
hours = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
model = sys.argv[3] if len(sys.argv) > 3 else "exp"
prop, almanac, central = leo_full_setup(degree=70, drag=model)
ctx = nx.GpuContext(prop.compile(almanac, central))
b = dispersed_leo_batch(n, seed=0)
b.drag_area_m2[:] = 2.0
out, st = ctx.propagate(b, int(hours * 3600) * nx.NS_PER_S)
out, st = ctx.propagate(b, int(hours * 3600) * nx.NS_PER_S)
print(f"drag {model} n={n} {hours} h: kernel {ctx.last_kernel_ms():.1f} ms, helpers {ctx.last_coop_helpers()}, status!=0 {(st.status != 0).sum()}")
