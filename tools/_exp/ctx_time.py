import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nyx_amd as nx
from scenarios import dispersed_leo_batch, leo_full_setup
batch = dispersed_leo_batch(640, seed=1)
for deg in list(range(40, 97, 3)) + [70]:
    prop, almanac, central = leo_full_setup(degree=deg)
    compiled = prop.compile(almanac, central)
    for flags in (0, 0x2000000):
        t0 = time.time()
        ctx = nx.GpuContext(compiled, tuning=nx.Tuning(debug_flags=flags))
        t1 = time.time()
        ctx.propagate(batch, 60 * nx.NS_PER_S)
        t2 = time.time()
        ctx.close()
        print(f"degree {deg} flags {flags:#x}: create {t1 - t0:.3f} s, first launch {t2 - t1:.3f} s, helpers {ctx.last_coop_helpers() if False else ''}", flush=True)
