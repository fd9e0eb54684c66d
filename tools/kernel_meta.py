#!/usr/bin/env python3
"""Code-object metadata of every gfx950 kernel in libnyx_hip.so: registers, scratch, static spill counts, LDS.
usage: tools/kernel_meta.py [lib.so]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        # the section concatenates one offload bundle per translation unit: split at the magic strings
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
        for k, st in enumerate(starts):
            piece = os.path.join(td, f"b{k}.bin")
            open(piece, "wb").write(data[st:(starts[k + 1] if k + 1 < len(starts) else len(data))])
            co = os.path.join(td, f"k{k}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            # .text bytes of every function symbol of this code object (kernels and the out-of-line device functions they call)
            sizes = {}
            for ln in subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", co], capture_output=True, text=True).stdout.splitlines():
                f = ln.split()
                if len(f) >= 8 and f[3] == "FUNC":
                    sizes[f[7]] = int(f[2])
            first = len(out)
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if key == "agpr_count" and cur.get("name"):
                    out.append(cur)
                    cur = {}
                if key in ("name", "vgpr_count", "sgpr_count", "agpr_count", "private_segment_fixed_size", "sgpr_spill_count",
                           "vgpr_spill_count", "group_segment_fixed_size", "max_flat_workgroup_size"):
                    cur[key] = val
            if cur.get("name"):
                out.append(cur)
            for k in out[first:]:
                k["text_bytes"] = str(sizes.get(k.get("name", ""), 0))
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "nyx_amd", "libnyx_hip.so")
    print(f"{'kernel':36s} {'wg':>5s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch B':>9s} {'sgpr spills':>11s} {'vgpr spills':>11s} {'.text B':>9s}")
    for k in kernels(lib):
        print(f"{k.get('name', '?'):36s} {k.get('max_flat_workgroup_size', '?'):>5s} {k.get('vgpr_count', '?'):>5s} {k.get('agpr_count', '?'):>5s} "
              f"{k.get('sgpr_count', '?'):>5s} {k.get('private_segment_fixed_size', '?'):>9s} {k.get('sgpr_spill_count', '?'):>11s} {k.get('vgpr_spill_count', '?'):>11s} {k.get('text_bytes', '?'):>9s}")
