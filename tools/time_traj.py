"""Times the trajectory-evaluation kernel: n trajectories x `hours` of dense output resampled every `step_s`.
usage: python tools/time_traj.py [n] [hours] [step_s]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

os.environ.setdefault("NYX_HIP_TUNING_ENV", "1")  # the A/B switches of these tools travel through the environment
import nyx_amd as nx
from scenarios import dispersed_leo_batch, leo_full_setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
hours = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
step_s = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
prop, almanac, central = leo_full_setup(degree=8)
ctx = nx.GpuContext(prop.compile(almanac, central))
b = dispersed_leo_batch(n, seed=0)
dur = int(hours * 3600) * nx.NS_PER_S
cap = int(hours * 3600 / 40) + 64
t0 = time.time()
out, st, traj = ctx.propagate_with_traj(b, dur, capacity=cap)
print(f"propagate_with_traj: {time.time() - t0:.2f} s wall, kernel {ctx.last_kernel_ms():.1f} ms, stored states max {traj.len.max()} (cap {cap})")
count = int(hours * 3600 / step_s) + 1
for rep in range(2):
    t0 = time.time()
    ev = ctx.traj_every(traj, int(step_s * 1e9), count)
    wall = time.time() - t0
    ms = ctx.last_kernel_ms()
    samples = int(np.minimum(ev.len, count).sum())
    print(f"traj_every: {samples} samples, kernel {ms:.2f} ms = {samples / ms / 1e3:.1f} M samples/s "
          f"({samples * 56 / ms / 1e6:.1f} GB/s written), host call {wall:.2f} s")
ctx.close()
