#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd (.db) result: per-kernel stats (like --stats CSV) and PMC counter sums per dispatch.
usage: tools/rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
out = []
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
rows = con.execute("select * from kernels").fetchall()
stats = {}
for r in rows:
    d = dict(zip(cols, r))
    name = d.get("name") or d.get("kernel_name")
    dur = d.get("duration") or (d["end"] - d["start"])
    s = stats.setdefault(name, [])
    s.append((dur, d.get("grid_x", d.get("grid_size_x")), d.get("workgroup_x", d.get("workgroup_size_x")), d.get("lds_size", d.get("lds_block_size")),
              d.get("scratch_size"), d.get("vgpr_count", d.get("arch_vgpr_count")), d.get("sgpr_count")))
# The first launch of a context calibrates the column schedule with a few SHORT launches of the same kernel (abi.cpp calibrate()):
# list them apart from the timed launches, whose average is the number bench.py reports.
split = {}
for name, v in stats.items():
    longest = max(x[0] for x in v)
    short = [x for x in v if x[0] < 0.5 * longest]
    if "nyx_propagate" in str(name) and short and len(short) < len(v):
        long_ = [x for x in v if x[0] >= 0.5 * longest]
        # (the timed launches are the cluster that holds most of the time: the day-long launch against its short calibration runs,
        #  or the 60 one-minute segments of predict_until against the four longer calibration runs of that shape)
        timed, cal = (long_, short) if sum(x[0] for x in long_) >= sum(x[0] for x in short) else (short, long_)
        split[f"{name} [timed launches]"] = timed
        split[f"{name} [calibration launches of the context's first call]"] = cal
    else:
        split[name] = v
stats = split
tot = sum(sum(x[0] for x in v) for v in stats.values()) or 1
out.append("| kernel | calls | total ms | avg ms | min ms | max ms | % | grid | wg | lds | scratch | vgpr | sgpr |")
out.append("|---|---:|---:|---:|---:|---:|---:|---|---|---|---|---|---|")
for name, v in sorted(stats.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    ds = [x[0] for x in v]
    out.append(f"| `{name[:90]}` | {len(v)} | {sum(ds)/1e6:.3f} | {sum(ds)/len(ds)/1e6:.3f} | {min(ds)/1e6:.3f} | {max(ds)/1e6:.3f} | {100*sum(ds)/tot:.1f} | "
               f"{v[0][1]} | {v[0][2]} | {v[0][3]} | {v[0][4]} | {v[0][5]} | {v[0][6]} |")
try:
    pc = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    prow = con.execute("select * from counters_collection").fetchall()
    if prow:
        out.append("")
        out.append("| kernel | dispatch | counter | value |")
        out.append("|---|---:|---|---:|")
        for r in prow:
            d = dict(zip(pc, r))
            out.append(f"| `{str(d.get('kernel_name', d.get('name')))[:50]}` | {d.get('dispatch_id')} | {d.get('counter_name')} | {d.get('value')} |")
except Exception as e:  # noqa: BLE001
    out.append(f"(no counters: {e})")
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
