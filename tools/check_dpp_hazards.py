#!/usr/bin/env python3
"""Scan the gfx950 disassembly of libnyx_hip.so for the two DPP hazards the assembler cannot see inside inline asm:
a VALU write of a VGPR that a v_*_dpp reads as its DPP operand within the next two instructions, and a VALU write of EXEC
within five instructions before a DPP instruction.  usage: tools/check_dpp_hazards.py [lib.so]; exit code 1 on a finding."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def disassemble(lib):
    """Disassembly of every gfx950 code object of the library (one offload bundle per translation unit)."""
    text = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
        for k, st in enumerate(starts):
            piece, co = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"k{k}.co")
            open(piece, "wb").write(data[st:(starts[k + 1] if k + 1 < len(starts) else len(data))])
            r = subprocess.run([BUNDLER, "--unbundle", "--type=o", f"--input={piece}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            text.append(subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout)
    return "\n".join(text)


def vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(text):
    findings, n_dpp = [], 0
    window = []  # (mnemonic, dst tokens) of the last instructions of the current function
    for line in text.splitlines():
        line = line.split("//")[0].strip()
        if not line or line.endswith(":"):
            if line.endswith(":"):
                window = []
            continue
        parts = line.replace(",", " ").split()
        mn, ops = parts[0], parts[1:]
        if "_dpp" in mn:
            n_dpp += 1
            src0 = vregs(ops[1]) if len(ops) > 1 else set()
            for back, (pmn, pops) in enumerate(reversed(window[-5:]), 1):
                valu = pmn.startswith("v_")
                if valu and back <= 2 and pops and vregs(pops[0]) & src0:
                    findings.append(f"VGPR hazard: '{pmn} {' '.join(pops)}' {back} before '{line}'")
                if valu and pops and pops[0] in ("exec", "exec_lo", "exec_hi"):
                    findings.append(f"EXEC hazard: '{pmn} {' '.join(pops)}' {back} before '{line}'")
        if not mn.startswith("s_nop"):
            window.append((mn, ops))
        else:  # s_nop N counts as N + 1 wait states
            m = re.search(r"\d+", line)
            window.extend([("s_nop", [])] * ((int(m.group(0)) if m else 0) + 1))
    return n_dpp, findings


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "nyx_amd", "libnyx_hip.so")
    n, f = scan(disassemble(lib))
    print(f"{n} DPP instructions, {len(f)} hazard finding(s)")
    for x in f[:20]:
        print("  " + x)
    sys.exit(1 if f else 0)
