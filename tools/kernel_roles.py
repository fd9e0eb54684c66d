#!/usr/bin/env python3
"""Per-ROLE instruction census of the propagation kernels ("which role owns the scratch traffic").

Every role of a workgroup (integrator, almanac, perturbations, column waves; plain and pipelined stage loop) is an instantiation
of role_loop<> inlined into ONE kernel, so the code object's metadata (scratch bytes, spill counts) cannot say whose they are.
role_loop brackets its code with marker pairs (`s_nop 13; s_nop <id>` ... `s_nop 13; s_nop 15`, id = INTEG | ALMANAC << 1 |
PERT << 2 | PIPE << 3); this tool disassembles every gfx950 code object of the library, cuts each kernel at the markers and counts,
per role: instructions, VALU, scratch loads / stores (VGPR spills and private arrays), v_readlane / v_writelane (SGPR spills),
LDS and global / flat memory instructions, calls.  Static counts - how often a piece runs is not in the object: the stage loop's
share is what matters, so the census also reports the counts INSIDE the innermost region between the first and the last
s_barrier of a role (the stage loop) separately.

usage: tools/kernel_roles.py [lib.so] [--md out.md]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
ROLE = {1: "integrator", 2: "almanac", 4: "perturbations", 6: "almanac + perturbations", 7: "all roles (one wave)", 0: "column wave"}


def role_name(i):
    return ROLE.get(i & 7, f"roles {i & 7}") + (", pipelined loop" if i & 8 else ", plain loop")


def code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
        for k, st in enumerate(starts):
            piece = os.path.join(td, f"b{k}.bin")
            open(piece, "wb").write(data[st:(starts[k + 1] if k + 1 < len(starts) else len(data))])
            co = os.path.join(td, f"k{k}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            yield subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout


def classify(op):
    if op.startswith("scratch_load"):
        return "scratch_ld"
    if op.startswith("scratch_store"):
        return "scratch_st"
    if op in ("v_readlane_b32", "v_writelane_b32"):
        return "sgpr_spill"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("s_swappc"):
        return "calls"
    if op.startswith("v_"):
        return "valu"
    return "other"


def census(text):
    """{kernel: [(role id, counts, stage-loop counts)]}"""
    out = {}
    kernel = None
    cur = None
    pend13 = False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            kernel = m.group(1)
            cur = None
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", line)
        if not m or kernel is None:
            continue
        op, args = m.group(1), m.group(2)
        if op == "s_nop":
            n = int(args, 0)
            if pend13:
                pend13 = False
                if n == 15 and cur is not None:
                    out.setdefault(kernel, []).append(cur)
                    cur = None
                    continue
                if n != 13:
                    cur = {"id": n, "all": {}, "ops": []}
                    continue
            if n == 13:
                pend13 = True
                continue
        pend13 = False
        if cur is not None:
            cur["ops"].append(op)
    res = {}
    for k, roles in out.items():
        rows = []
        for r in roles:
            ops = r["ops"]
            bars = [i for i, o in enumerate(ops) if o == "s_barrier"]
            loop = ops[bars[0]:bars[-1] + 1] if len(bars) >= 2 else []

            def count(seq):
                c = {}
                for o in seq:
                    c[classify(o)] = c.get(classify(o), 0) + 1
                c["total"] = len(seq)
                return c
            rows.append((r["id"], count(ops), count(loop)))
        res[k] = rows
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "nyx_amd", "libnyx_hip.so")
    md = sys.argv[sys.argv.index("--md") + 1] if "--md" in sys.argv else None
    lines = ["| kernel | role | instructions | VALU | scratch loads | scratch stores | v_readlane / v_writelane | LDS | global / flat | calls | of which between the role's first and last barrier: instructions / scratch ld / scratch st / lane moves |",
             "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---|"]
    for text in code_objects(lib):
        for kernel, rows in census(text).items():
            for rid, c, l in rows:
                g = lambda d, k: d.get(k, 0)
                lines.append(f"| `{kernel}` | {role_name(rid)} | {c['total']} | {g(c, 'valu')} | {g(c, 'scratch_ld')} | {g(c, 'scratch_st')} | {g(c, 'sgpr_spill')} | "
                             f"{g(c, 'lds')} | {g(c, 'vmem')} | {g(c, 'calls')} | {g(l, 'total')} / {g(l, 'scratch_ld')} / {g(l, 'scratch_st')} / {g(l, 'sgpr_spill')} |")
    text = "\n".join(lines) + "\n"
    if md:
        open(md, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
