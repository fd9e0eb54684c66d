#!/usr/bin/env python3
"""What tests/test_code_budget.py measures on the built library, and the tool that writes the committed budgets
(tests/golden/code_budget.json): `python tools/code_budget.py --update [slack]` = measured figures x (1 + slack, default 0.08),
to be run - and its diff read - when a change of the kernels is INTENDED to move them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_meta  # noqa: E402
import kernel_roles  # noqa: E402

PROPAGATION = ("nyx_propagate_kernel", "nyx_propagate_kernel_w8", "nyx_propagate_kernel_w8n", "nyx_propagate_kernel_stm", "nyx_propagate_kernel_stmq",
               "nyx_propagate_kernel_stmq_w8", "nyx_propagate_kernel_p2", "nyx_propagate_kernel_fan")
PROPAGATION = PROPAGATION + tuple(k + "_prof" for k in PROPAGATION)   # the twins that carry the in-kernel accounting (NYX_PROF)
ROLE_KERNELS = ("nyx_propagate_kernel", "nyx_propagate_kernel_stmq", "nyx_propagate_kernel_w8n")


def measure(lib):
    out = {"kernels": {}, "roles": {}}
    for k in kernel_meta.kernels(lib):
        name = k.get("name", "")
        if not name.startswith("nyx_") and "nyx_" not in name:
            continue
        short = name if name in PROPAGATION else name.split("nyx_")[1].split("kernel")[0].rstrip("_") + "_kernel"
        out["kernels"][short if name not in PROPAGATION else name] = {
            "vgprs": int(k.get("vgpr_count", 0)) + int(k.get("agpr_count", 0)), "scratch_bytes": int(k.get("private_segment_fixed_size", 0)),
            "vgpr_spills": int(k.get("vgpr_spill_count", 0)), "sgpr_spills": int(k.get("sgpr_spill_count", 0)), "text_bytes": int(k.get("text_bytes", 0))}
    for text in kernel_roles.code_objects(lib):
        for kernel, rows in kernel_roles.census(text).items():
            if kernel not in ROLE_KERNELS:
                continue
            for rid, c, l in rows:
                key = f"{kernel}: {kernel_roles.role_name(rid)}"
                out["roles"][key] = {"loop_instructions": l.get("total", 0), "loop_scratch_loads": l.get("scratch_ld", 0), "loop_scratch_stores": l.get("scratch_st", 0),
                                     "loop_lane_moves": l.get("sgpr_spill", 0), "instructions": c.get("total", 0)}
    return out


def toolchain():
    """The compiler the figures belong to: budgets are register-allocation outcomes of ONE hipcc (tests/test_code_budget.py skips under another)."""
    import subprocess
    try:
        out = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout
    except OSError:
        return None
    for line in out.splitlines():
        if line.startswith("HIP version:"):
            return line.split(":", 1)[1].strip()
    return None


if __name__ == "__main__":
    lib = os.path.join(ROOT, "nyx_amd", "libnyx_hip.so")
    m = measure(lib)
    if "--update" in sys.argv:
        i = sys.argv.index("--update")
        slack = float(sys.argv[i + 1]) if len(sys.argv) > i + 1 else 0.08
        up = lambda v: int(v * (1.0 + slack)) + (4 if v else 0)
        b = {"note": "budgets = the figures of the build they were written from x (1 + slack); see tests/test_code_budget.py", "slack": slack,
             "hipcc": toolchain(),
             "kernels": {k: {f: (v if f == "vgprs" else up(v)) for f, v in d.items()} for k, d in m["kernels"].items()},
             "roles": {k: {f: up(v) for f, v in d.items()} for k, d in m["roles"].items()}}
        json.dump(b, open(os.path.join(ROOT, "tests", "golden", "code_budget.json"), "w"), indent=1, sort_keys=True)
        print("wrote tests/golden/code_budget.json")
    else:
        print(json.dumps(m, indent=1, sort_keys=True))
