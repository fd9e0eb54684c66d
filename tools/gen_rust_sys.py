#!/usr/bin/env python3
"""Generates the Rust `-sys` module (the `#[repr(C)]` structs, constants and `extern "C"` block a maintainer adds as
nyx-core/src/propagators/gpu/sys.rs) from include/nyx_hip.h, so that the text in INTEGRATION.md cannot drift from the header.

    python tools/gen_rust_sys.py            # prints sys.rs
    python tools/gen_rust_sys.py --update   # rewrites the block between the markers in INTEGRATION.md

tests/test_rust_binding.py regenerates it, diffs it against INTEGRATION.md, and checks the struct sizes this parser derives
(C layout rules on x86-64 / the Rust repr(C) rules) against nyx_hip_abi_sizeof() of the built library."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nyx_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED sys.rs (tools/gen_rust_sys.py) -->", "<!-- END GENERATED sys.rs -->"

PRIM = {"int32_t": ("i32", 4), "uint32_t": ("u32", 4), "int64_t": ("i64", 8), "uint64_t": ("u64", 8), "double": ("f64", 8),
        "char": ("c_char", 1), "void": ("c_void", 0)}


def parse(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    defines = {m.group(1): int(m.group(2).rstrip("u"), 0) for m in re.finditer(r"#define\s+(NYX_HIP_\w+)\s+(0x[0-9a-fA-F]+u?|\d+u?)\s*$", text, flags=re.M)}
    enums = []
    for m in re.finditer(r"enum\s+(\w+)\s*\{(.*?)\}\s*;", text, flags=re.S):
        items, nxt = [], 0
        for it in m.group(2).split(","):
            it = it.strip()
            if not it:
                continue
            if "=" in it:
                name, val = [x.strip() for x in it.split("=")]
                nxt = int(val, 0)
            else:
                name = it
            items.append((name, nxt))
            nxt += 1
        enums.append((m.group(1), items))
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            dm = re.match(r"(const\s+)?(struct\s+)?(\w+)\s+(.*)$", decl)
            const, base, rest = bool(dm.group(1)), dm.group(3), dm.group(4)
            for d in rest.split(","):
                d = d.strip()
                ptr = d.count("*")
                d = d.replace("*", "").strip()
                dims = [x for x in re.findall(r"\[(\w+)\]", d)]
                name = re.match(r"\w+", d).group(0)
                fields.append(dict(name=name, base=base, const=const, ptr=ptr, dims=[defines[x] if x in defines else int(x) for x in dims]))
        structs.append((m.group(3), fields))
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text)
    funcs = []
    for m in re.finditer(r"^\s*(const\s+char\s*\*|int32_t|int64_t|void|double)\s*(nyx_hip_\w+)\s*\((.*?)\)\s*;", text, flags=re.S | re.M):
        args = []
        a = " ".join(m.group(3).split())
        if a != "void":
            for arg in a.split(","):
                am = re.match(r"\s*(const\s+)?(struct\s+)?(\w+)\s*((?:\*\s*(?:const\s*)?)*)(\w+)\s*$", arg)
                # `T *const *p`: the const AFTER a star qualifies the pointer of that level (seen from the next level out)
                lv_const = [bool(x) for x in re.findall(r"\*\s*(const)?", am.group(4))]
                args.append(dict(const=bool(am.group(1)), base=am.group(3), ptr=am.group(4).count("*"), name=am.group(5), lv_const=lv_const))
        funcs.append((m.group(2), " ".join(m.group(1).split()), args))
    return defines, enums, structs, opaque, funcs


def rust_type(base, const, ptr, known, lv_const=None):
    t = PRIM[base][0] if base in PRIM else base
    for k in range(ptr):
        # `const T *` is a pointer to const; further levels (T **) are out-parameters unless the level below is `*const`
        pointee_const = (const and k == 0) or (k > 0 and lv_const and lv_const[k - 1])
        t = ("*const " if pointee_const else "*mut ") + t
    return t


def layout(structs):
    """(size, align) of every struct under the x86-64 SysV / repr(C) rules."""
    out = {}
    for name, fields in structs:
        off, al = 0, 1
        for f in fields:
            if f["ptr"]:
                sz, a = 8, 8
            elif f["base"] in PRIM:
                sz = a = PRIM[f["base"]][1]
            else:
                sz, a = out[f["base"]]
            n = 1
            for d in f["dims"]:
                n *= d
            off = (off + a - 1) // a * a
            off += sz * n
            al = max(al, a)
        out[name] = ((off + al - 1) // al * al, al)
    return out


def field_type(f):
    t = rust_type(f["base"], f["const"], f["ptr"], None)
    for d in reversed(f["dims"]):
        t = f"[{t}; {d}]"
    return t


def generate():
    defines, enums, structs, opaque, funcs = parse(open(HEADER).read())
    L = ["// nyx-core/src/propagators/gpu/sys.rs — GENERATED from include/nyx_hip.h by tools/gen_rust_sys.py; do not edit.",
         "#![allow(non_camel_case_types, non_upper_case_globals)]", "use std::os::raw::{c_char, c_void};", ""]
    for k, v in defines.items():
        ty = "u32" if k.startswith("NYX_HIP_FLAG") or k == "NYX_HIP_ABI_VERSION" else "usize"
        L.append(f"pub const {k}: {ty} = {v};")
    L.append("")
    for ename, items in enums:
        L.append(f"// enum {ename}")
        L += [f"pub const {n}: i32 = {v};" for n, v in items]
    L.append("")
    sizes = layout(structs)
    for name, fields in structs:
        L.append(f"#[repr(C)] #[derive(Clone, Copy)] pub struct {name} {{   // {sizes[name][0]} bytes")
        row = "   "
        for f in fields:
            item = f" pub {f['name']}: {field_type(f)},"
            if len(row) + len(item) > 118:
                L.append(row)
                row = "   "
            row += item
        L.append(row)
        L.append("}")
    for _, alias in opaque:
        L.append(f"pub enum {alias} {{}}   // opaque")
    L += ["", '#[link(name = "nyx_hip")]', 'extern "C" {']
    for fname, ret, args in funcs:
        a = ", ".join(f"{x['name'] if x['name'] not in ('in', 'type') else x['name'] + '_'}: {rust_type(x['base'], x['const'], x['ptr'], None, x.get('lv_const'))}" for x in args)
        r = {"int32_t": " -> i32", "int64_t": " -> i64", "double": " -> f64", "void": "", "const char *": " -> *const c_char"}[ret]
        line = f"    pub fn {fname}({a}){r};"
        while len(line) > 120:
            cut = line.rfind(", ", 0, 118)
            L.append(line[:cut + 1])
            line = "            " + line[cut + 2:]
        L.append(line)
    L.append("}")
    return "\n".join(L) + "\n", sizes


def main():
    text, _ = generate()
    if "--update" in sys.argv:
        doc = open(DOC).read()
        i, j = doc.index(BEGIN), doc.index(END)
        doc = doc[: i + len(BEGIN)] + "\n```rust\n" + text + "```\n" + doc[j:]
        open(DOC, "w").write(doc)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
