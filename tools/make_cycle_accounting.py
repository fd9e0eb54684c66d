#!/usr/bin/env python3
"""profiles/round06_cycle_accounting.md from the logs of tools/r6_call20.sh (gpurun_out/r6_cycle_accounting.log: one tools/sweep.py run per
workload, product kernel and accounting twin) - the header states what the tables are, the numbers are read off the tables themselves.
usage: tools/make_cycle_accounting.py [out.md]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_stamp import kernel_source_stamp  # noqa: E402

log = open(os.path.join(ROOT, "gpurun_out", "r6_cycle_accounting.log")).read()
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "round06_cycle_accounting.md")
blocks = ["config " + b.strip() for b in log.split("\nconfig ") if b.strip()]
blocks[0] = blocks[0].replace("config config ", "config ")


def wave_rows(block):
    rows = {}
    for m in re.finditer(r"^\s+wave\s+(\d+):\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+) \|\s+(\d+)\s+(\d+)", block, re.M):
        rows[int(m.group(1))] = [int(x) for x in m.groups()[1:]]
    return rows


def first(block, pat):
    m = re.search(pat, block)
    return m.groups() if m else None


c2 = next(b for b in blocks if b.startswith("config 2 n=10000"))
sh = next((b for b in blocks if b.startswith("config 2 n=1250")), "")
w = wave_rows(c2)
integ = w[0]
busy = integ[1] + integ[2] + integ[3] + integ[4]
col = sorted((w[k][2], k) for k in w if k >= 3)[-3:]
lat = first(c2, r"wait for the answer (\d+), answer in hand -> post (\d+), post (\d+)")
hp = first(c2, r"wait-for-slot (\d+), scan\+claim\+fetch (\d+), lost claims ([\d.]+); total (\d+)")
hw = [int(x) for x in re.findall(r"hwave\s+\d+: busy\s+(\d+)", c2)]
ws = wave_rows(sh) if sh else {}
hdr = f"""# round06: in-kernel cycle accounting of the final build (tools/sweep.py, `tuning.profile = 1`; kernel sources {kernel_source_stamp(ROOT)})

One process per configuration; `product` = the product kernel (no accounting), `accounting` = its `_prof` twin (the in-kernel
counters cost 1-20 % - compare the two `ms`; the fan-out shard's integrator chain pays the most).  Cycles per force evaluation of
workgroup 0: integ_front (sixteen-wave plain kernels; phase A elsewhere), window duty (role work; the integrator's includes integ_front),
harmonics walk, phase C, step control | total, barrier wait; then the owner's latency loop and the first helper workgroup (cooperative
launches).  `rows/wave`: table rows each wave walks under the solo / owner / helper schedule.  MI355X, one box.
Workloads: configs[1] (10 000 x 24 h), its 8-GPU shard (1 250 x 24 h: fan-out mode, 160 dedicated helpers), config 3 (5 000 x 30 d),
config 4 (1 000 x 60 updates, ONE launch), config 5 (6 250 x 6 h of the 72), configs[1]'s force model on a full chip (16 384 x 3 h).

Reading it (round 6).  **configs[1]**: the integrator wave is busy {integ[1] / 1e3:.1f} k (window, integ_front's {integ[0] / 1e3:.1f} k included) + {integ[3] / 1e3:.1f} k (fold +
integ_back) + {integ[4] / 1e3:.1f} k (step control) + {integ[2] / 1e3:.1f} k = **{busy / 1e3:.1f} k cycles per evaluation** (round 5: 27.0 k) and waits {integ[6] / 1e3:.1f} k at the
barrier; the period ({integ[5] / 1e3:.1f} k on this box, 28.3-29.3 k over the boxes of the round) is the column waves': {' / '.join(f'{c[0] / 1e3:.1f}' for c in reversed(col))} k of walking on the three
slowest (waves {', '.join(str(c[1]) for c in reversed(col))}) + {min(w[c[1]][6] for c in col) / 1e3:.1f}-{max(w[c[1]][6] for c in col) / 1e3:.1f} k at the barrier; the owner waits {int(lat[0]) / 1e3:.1f} k for a helper's answer - one uncached
read -; the first helper's producer holds the next job, claimed and fetched ({int(hp[1]) / 1e3:.1f} k, {hp[2]} lost claims), while it waits for a free slot
({int(hp[0]) / 1e3:.1f} k): its three hops are under the column waves' work, which are busy {min(hw) / 1e3:.1f}-{max(hw) / 1e3:.1f} k of a {int(hp[3]) / 1e3:.1f} k job period - the owners' posting
rate x 157 / 99.  **The 1 250-trajectory shard**: the thirteen column waves hold three rows and skip their walk; the period ({ws.get(0, [0] * 7)[5] / 1e3:.1f} k with the
accounting) is the integrator's chain (busy {sum(ws.get(0, [0] * 7)[1:5]) / 1e3:.1f} k) against the turnaround of a helper job.  **Config 4**: per SEGMENT (NYX_SEG_PROF twin, GPU call 19) the
integrator's boundary is step control 11.4 k, next attempt opened 1.7 k, B0 0.2 k, time updates 15.9 k, re-arming 2.5 k, stage-0 epoch data + Bp 5.9 k,
phase A + B1 3.6 k: 41 k of the 343 k cycles of a segment.  **Step control in pieces** (NYX_SEG_PROF twin of the plain kernel, GPU call 9, cycles
per attempt): cold state 1.5 k, the two sums 4.8 k, error estimate + decision + state update 13.0 k (10.7 k with one pow), next attempt opened 1.6 k, rest 1.2 k.

```
"""
open(out, "w").write(hdr + "\n\n".join(blocks) + "\n```\n")
print(f"wrote {out}: {len(blocks)} workloads")
