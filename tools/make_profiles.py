#!/usr/bin/env python3
"""Turns the scratch output of tools/profile_round.sh (gpurun_out/prof_cfgN/) into the tracked summaries under profiles/:
roundNN_cfgN_bench.json, _kernel_trace_stats.md, _pmc.md and _hbm_traffic.json (what bench.py's roofline.traffic reads).
usage: tools/make_profiles.py <round tag, e.g. round02> [configs...]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_stamp import kernel_source_stamp  # noqa: E402
tag = sys.argv[1]
# configurations by number, or "fullchip" (round 5: configs[1]'s force model on a full chip, 16 384 trajectories x 3 h: gpurun_out/prof_fullchip)
items = sys.argv[2:] or ["2", "3", "4", "5"]
CLOCK_HZ, SIMDS = 2.4e9, 1024
degree_of = {2: 70, 3: 0, 4: 21, 5: 150}
for item in items:
    cfg = 2 if item == "fullchip" else int(item)
    name = "fullchip" if item == "fullchip" else f"cfg{cfg}"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{name}")
    if not os.path.isdir(src):
        continue
    line = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    bench_out = os.path.join(ROOT, "profiles", f"{tag}_{name}_bench.json")
    shutil.copy(os.path.join(src, "kernel_trace_stats.md"), os.path.join(ROOT, "profiles", f"{tag}_{name}_kernel_trace_stats.md"))
    if name == "cfg2" and os.path.exists(os.path.join(src, "bench_with_others.json")):   # the default run, `other_configs` included
        full = json.loads(open(os.path.join(src, "bench_with_others.json")).read().strip().splitlines()[-1])
        json.dump(full, open(os.path.join(ROOT, "profiles", f"{tag}_cfg2_bench_with_other_configs.json"), "w"), indent=1)
    per = collections.defaultdict(list)
    # (gpurun merges every pass of a configuration into the same scratch directory: the newest file of each counter set counts)
    newest = {}
    for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
        key = f[len(src):].split(os.sep)[1]
        if key not in newest or os.path.getmtime(f) > os.path.getmtime(newest[key]):
            newest[key] = f
    for f in newest.values():
        for r in csv.DictReader(open(f)):
            if "nyx_propagate" in r.get("Kernel_Name", ""):
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the dominant dispatch of the command = the timed launch (the short calibration launches of a fresh context are the others);
    # config 4 launches one segment kernel per time update: all alike, take the mean
    # (rounds 2-5: config 4 launched one segment kernel per time update - all alike, the mean was taken; round 6: the covariance-mapping
    #  loop is ONE launch of the quad STM kernel, the largest dispatch like everywhere else.  PER_SEGMENT=1 restores the old reading)
    per_segment = cfg == 4 and os.environ.get("PER_SEGMENT") == "1"
    pick = (lambda v: sum(v) / len(v)) if per_segment else max
    c = {k: pick(v) for k, v in per.items()}
    launches = 60 if per_segment else 1
    k_ms = line["kernel_ms"] / launches
    simd_cycles = SIMDS * k_ms * 1e-3 * CLOCK_HZ
    fetch_b, write_b = c.get("FETCH_SIZE", 0.0) * 1024.0, c.get("WRITE_SIZE", 0.0) * 1024.0
    traffic = 2.0 * fetch_b + write_b   # MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half of the bytes on gfx950; WRITE_SIZE as is
    extra = " --n 16384 --hours 3" if name == "fullchip" else ""
    out = [f"# {tag}, BASELINE config {cfg}{' on a full chip (16 384 trajectories, 3 h)' if name == 'fullchip' else ''}: PMC counters of the dominant kernel (rocprofv3 --pmc, one counter set per pass)", "",
           f"command: `python bench.py --config {cfg} --steps 1 --warmup 0 --no-cpu-baseline --no-dense-output --no-host-call --no-other-configs{extra}`; "
           f"values per launch of `{'nyx_propagate_kernel_stmq' if cfg == 4 else 'nyx_propagate_kernel'}` "
           f"({'mean over the segment launches' if per_segment else 'the timed launch: the largest dispatch of the command'}); "
           f"kernel time {k_ms:.3f} ms (bench line of the same build).", "",
           "| counter | per launch |", "|---|---:|"]
    for k in sorted(c):
        out.append(f"| {k} | {c[k]:.6g} |")
    out += ["", "Derived:", ""]
    if "SQ_INSTS_FLAT" in c and "SQ_INSTS_VALU" in c:
        out.append(f"* memory instructions per launch: FLAT (scratch spills and generic pointers) {c['SQ_INSTS_FLAT']:.4g}, VMEM reads {c.get('SQ_INSTS_VMEM_RD', 0):.4g}, "
                   f"VMEM writes {c.get('SQ_INSTS_VMEM_WR', 0):.4g}, LDS {c.get('SQ_INSTS_LDS', 0):.4g} against {c['SQ_INSTS_VALU']:.4g} VALU instructions "
                   f"(FLAT / VALU = {c['SQ_INSTS_FLAT'] / c['SQ_INSTS_VALU']:.4f})")
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        out.append(f"* SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f} of the wave-cycles are spent waiting on an instruction's operands or pipe")
    if "SQ_ACTIVE_INST_VALU" in c:
        out.append(f"* VALU issue: SQ_ACTIVE_INST_VALU x 4 cycles = {c['SQ_ACTIVE_INST_VALU'] * 4:.4g} of {simd_cycles:.4g} SIMD-cycles "
                   f"(1024 SIMDs x kernel time x 2.4 GHz) = **{c['SQ_ACTIVE_INST_VALU'] * 4 / simd_cycles:.3f}** of all issue slots on the chip "
                   f"(every VALU instruction, not only f64)")
    if degree_of.get(cfg, 0) > 0:
        nd = degree_of[cfg]
        entries = (nd + 1) * (nd + 2) / 2
        instr, traj_per_wave = (22.0, 16.0) if cfg == 4 else (9.0, 64.0)
        ex = line["force_evals_per_launch"] / launches / traj_per_wave * entries * instr * 4.0
        out.append(f"* f64 issue of the harmonics alone ({instr:.0f} f64 instructions per table entry per wave of {traj_per_wave:.0f} trajectories, "
                   f"{entries:.0f} entries, 4 cycles each): {ex:.4g} SIMD-cycles = **{ex / simd_cycles:.3f}** of the chip's")
    out.append(f"* HBM: FETCH_SIZE {fetch_b / 1e6:.4g} MB (x2 on gfx950 per MI355X_MICROARCH.md) + WRITE_SIZE {write_b / 1e6:.4g} MB = "
               f"**{traffic / 1e6:.5g} MB per launch** = {traffic / (k_ms * 1e-3) / 1e9:.4g} GB/s = {traffic / (k_ms * 1e-3) / 8e12:.2e} of the 8 TB/s peak; "
               f"algorithmic: {line['config']['trajectories_per_gpu'] * 272 / 1e6:.3g} MB")
    open(os.path.join(ROOT, "profiles", f"{tag}_{name}_pmc.md"), "w").write("\n".join(out) + "\n")
    n = line["config"]["trajectories_per_gpu"]
    w = line["config"]["workload"]
    hours = 3.0 if name == "fullchip" else {2: 24.0, 3: 720.0, 4: 1.0, 5: 72.0}[cfg]
    degree = {2: 70, 3: 0, 4: 21, 5: 150}[cfg]
    json.dump({"config": cfg, "n": n, "hours": hours, "degree": degree, "hbm_bytes_per_launch": traffic * launches,
               "kernel_source_stamp": kernel_source_stamp(ROOT),
               "fetch_size_bytes_raw": fetch_b * launches, "write_size_bytes": write_b * launches,
               "note": "2 x FETCH_SIZE + WRITE_SIZE of the timed step (gfx950 correction of MI355X_MICROARCH.md), rocprofv3 --pmc, separate passes",
               "workload": w}, open(os.path.join(ROOT, "profiles", f"{tag}_{name}_hbm_traffic.json"), "w"), indent=1)
    # the bench line was printed before the counter passes of this round existed: give it THIS round's traffic (same command, same build)
    line["roofline"]["traffic"] = traffic * launches
    line["roofline"]["traffic_source"] = f"profiles/{tag}_{name}_hbm_traffic.json"
    line["roofline"]["hbm"]["achieved"] = traffic * launches / (line["kernel_ms"] * 1e-3) / 1e9
    line["roofline"]["hbm"]["frac"] = line["roofline"]["hbm"]["achieved"] / line["roofline"]["hbm"]["peak"]
    json.dump(line, open(bench_out, "w"), indent=1)
    print(f"{name}: value {line['value']:.1f} {line['unit']}, kernel {line['kernel_ms']:.2f} ms, frac {line['roofline']['frac']:.4f}, traffic {traffic * launches / 1e6:.4g} MB")
