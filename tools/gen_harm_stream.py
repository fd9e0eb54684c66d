#!/usr/bin/env python3
"""Generates nyx_amd/csrc/harm_stream_asm.h: the hand-scheduled gfx950 loop of the hybrid harmonics feed (one asm statement).

Why assembly: the loop wants (i) a boundary test in front of EVERY row (a column may end anywhere in the stream) whose rare
branch must not cost the common path anything, (ii) two statically alternating sets of table registers fed by loads that run one
group ahead, (iii) the a1 / a2 pair of the recursion ping-ponging between two registers.  Written in C++ the register allocator
answers (i) with eight register copies per row and (ii) with a rotation of the sets through a third one (measured: slower than
the scalar loop it was to replace).  Here every register is named.

Register plan (all hard-coded ones are declared as clobbers; caller-saved VGPR ranges only):
  v[48:49] X, v[50:51] Y   a1 / a2 of the recursion, swapping roles every row
  v[52:53] T, v[54:55] U, v[84:85] W   temporaries
  v[64:71] S1..S4, v[80:83] S5, S6     the six sums of the column
  v[96:103] A0..A3, v[112:119] B0..B3  the two table register sets (16 rows x {t3..t6} each)
  s[40:87]   one scalar batch: 8 rows x {g, t1, t2}
  s[88:89] return address of the boundary code, s[90:91] c sqrt2 of the current column, s[92:99] header of the NEXT column
  s[100:101] scalar-side pointer, s[36:37] vector-side pointer, s[38:39] header pointer
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "nyx_amd", "csrc", "harm_stream_asm.h")

X, Y, T, U, W = "v[48:49]", "v[50:51]", "v[52:53]", "v[54:55]", "v[84:85]"
S = ["v[64:65]", "v[66:67]", "v[68:69]", "v[70:71]", "v[80:81]", "v[82:83]"]
A = ["v[96:97]", "v[98:99]", "v[100:101]", "v[102:103]"]
B = ["v[112:113]", "v[114:115]", "v[116:117]", "v[118:119]"]
EP, VP, HP, RET, CSC = "s[100:101]", "s[36:37]", "s[38:39]", "s[88:89]", "s[90:91]"
HDR_SCALE, HDR_DIAG, HDR_ROWS = "s[94:95]", "s[96:97]", "s98"
BATCH0 = 40
DPP = "row_mask:0xf bank_mask:0xf"

lines = []


def emit(s):
    lines.append(s)


def add64(reg_lo, reg_hi, imm):
    emit(f"s_add_u32 {reg_lo}, {reg_lo}, {imm}")
    emit(f"s_addc_u32 {reg_hi}, {reg_hi}, 0")


def prefetch(regs):
    for j, r in enumerate(regs):
        emit(f"global_load_dwordx2 {r}, %[voff], {VP} offset:{128 * j}")
    add64("s36", "s37", 512)


def batch_load():
    emit(f"s_load_dwordx16 s[{BATCH0}:{BATCH0 + 15}], {EP}, 0x0")
    emit(f"s_load_dwordx16 s[{BATCH0 + 16}:{BATCH0 + 31}], {EP}, 0x40")
    emit(f"s_load_dwordx16 s[{BATCH0 + 32}:{BATCH0 + 47}], {EP}, 0x80")
    emit("s_waitcnt lgkmcnt(0)")
    for off in (0xc0, 0x100, 0x140):  # touch the three lines of the next batch: its loads will hit the scalar cache
        emit(f"s_load_dword %[sink], {EP}, {hex(off)}")
    add64("s100", "s101", 0xc0)


def row(k, regs):
    e, lane = k % 8, k % 16
    a1, a2 = (X, Y) if k % 2 == 0 else (Y, X)
    g = BATCH0 + 6 * e
    emit(f"s_cmp_eq_u32 %[left], 0")
    emit(f"s_cbranch_scc1 .Lhs_stub{k}_%=")
    emit(f".Lhs_cont{k}_%=:")
    emit(f"v_mul_f64 {T}, %[rho2], s[{g}:{g + 1}]")
    emit(f"v_mul_f64 {a2}, {T}, -{a2}")
    emit(f"v_fmac_f64 {a2}, %[rho_u], {a1}")  # a_n, now in a2's register: the roles swap
    emit(f"v_fmac_f64 {S[0]}, s[{g + 2}:{g + 3}], {a2}")
    emit(f"v_fmac_f64 {S[1]}, s[{g + 4}:{g + 5}], {a2}")
    for j in range(4):
        emit(f"v_fmac_f64_dpp {S[2 + j]}, {regs[j]}, {a2} row_newbcast:{lane} {DPP}")
    emit(f"s_add_i32 %[left], %[left], -1")


def stub(k):
    a1, a2 = (X, Y) if k % 2 == 0 else (Y, X)
    emit(f".Lhs_stub{k}_%=:")
    emit(f"s_getpc_b64 {RET}")
    emit(f".Lhs_pc{k}_%=:")
    emit(f"s_add_u32 s88, s88, .Lhs_back{k}_%=-.Lhs_pc{k}_%=")
    emit(f"s_addc_u32 s89, s89, 0")
    emit(f"s_branch .Lhs_boundary_%=")
    emit(f".Lhs_back{k}_%=:")
    emit(f"v_mov_b64 {a1}, 0")
    emit(f"v_mul_f64 {a2}, %[inv_rho], {HDR_DIAG}")
    emit(f"s_load_dwordx8 s[92:99], {HP}, 0x0")  # the header after this one; waited for at the next boundary (or batch load)
    add64("s38", "s39", 32)
    emit(f"s_branch .Lhs_cont{k}_%=")


# ---- prologue
emit(f"s_mov_b64 {EP}, %[e]")
emit(f"s_mov_b64 {VP}, %[vp]")
emit(f"s_mov_b64 {HP}, %[hp]")
emit(f"s_load_dwordx8 s[92:99], {HP}, 0x0")
add64("s38", "s39", 32)
prefetch(A)
for r in [X, Y] + S:
    emit(f"v_mov_b64 {r}, 0")
prefetch(B)
emit("s_cmp_eq_u32 %[low_half], 0")
emit("s_cbranch_scc1 .Lhs_half_%=")
emit(".Lhs_top_%=:")
batch_load()
emit("s_waitcnt vmcnt(4)")
for k in range(0, 8):
    row(k, A)
emit(".Lhs_half_%=:")
batch_load()
emit("s_waitcnt vmcnt(4)")
for k in range(8, 16):
    row(k, A)
prefetch(A)
batch_load()
emit("s_waitcnt vmcnt(4)")
for k in range(16, 24):
    row(k, B)
batch_load()
for k in range(24, 32):
    row(k, B)
prefetch(B)
emit("s_branch .Lhs_top_%=")
# ---- out of line
for k in range(32):
    stub(k)
emit(".Lhs_boundary_%=:")
emit("s_cmp_eq_u32 %[first], 0")
emit("s_cbranch_scc0 .Lhs_nofold_%=")
emit(f"v_mul_f64 {T}, %[rho], {CSC}")           # rho * c * sqrt(2)
emit(f"v_mul_f64 {U}, %[ic], {S[1]}")
emit(f"v_fmac_f64 {U}, %[rc], {S[0]}")
emit(f"v_fmac_f64 %[px], {T}, {U}")
emit(f"v_mul_f64 {U}, %[ic], -{S[0]}")
emit(f"v_fmac_f64 {U}, %[rc], {S[1]}")
emit(f"v_fmac_f64 %[py], {T}, {U}")
emit(f"v_mul_f64 {U}, %[ic], {S[3]}")
emit(f"v_fmac_f64 {U}, %[rc], {S[2]}")
emit(f"v_fmac_f64 %[pz], %[rho], {U}")
emit(f"v_mul_f64 {U}, %[ic], {S[5]}")
emit(f"v_fmac_f64 {U}, %[rc], {S[4]}")
emit(f"v_add_f64 %[pw], %[pw], -{U}")
emit(f"v_mul_f64 {T}, %[rc], %[zr]")            # (rc + i ic) *= (zr + i zi), products and sums unfused as in cpow_uniform's caller
emit(f"v_mul_f64 {U}, %[ic], %[zi]")
emit(f"v_add_f64 {T}, {T}, -{U}")
emit(f"v_mul_f64 {U}, %[rc], %[zi]")
emit(f"v_mul_f64 {W}, %[ic], %[zr]")
emit(f"v_add_f64 %[ic], {U}, {W}")
emit(f"v_mov_b64 %[rc], {T}")
emit(".Lhs_nofold_%=:")
emit("s_mov_b32 %[first], 0")
emit("s_cmp_eq_u32 %[cols_left], 0")
emit("s_cbranch_scc1 .Lhs_done_%=")
emit("s_add_i32 %[cols_left], %[cols_left], -1")
emit("s_waitcnt lgkmcnt(0)")
emit(f"s_mov_b32 %[left], {HDR_ROWS}")
emit(f"s_mov_b64 {CSC}, {HDR_SCALE}")
for r in S:
    emit(f"v_mov_b64 {r}, 0")
emit(f"s_setpc_b64 {RET}")
emit(".Lhs_done_%=:")
emit("s_waitcnt vmcnt(0) lgkmcnt(0)")

clob = [f"s{i}" for i in range(36, 102)] + [f"v{i}" for r in ((48, 56), (64, 72), (80, 88), (96, 104), (112, 120)) for i in range(*r)]
clob += ["vcc", "scc", "memory"]

import io  # noqa: E402

f = io.StringIO()
f.write("// GENERATED by tools/gen_harm_stream.py - do not edit.  The hybrid-feed walk of one column range (see the generator's header).\n")
f.write("#define HARM_STREAM_ASM(e, vp, hp, voff, left, cols_left, first, low_half, sink, rho_u, rho2, rho, inv_rho, zr, zi, px, py, pz, pw, rc, ic) \\\n")
f.write("    asm volatile( \\\n")
for ln in lines:
    f.write(f'        "{ln}\\n\\t" \\\n')
f.write('        : [left] "+s"(left), [cols_left] "+s"(cols_left), [first] "+s"(first), [sink] "=&s"(sink), \\\n')
f.write('          [px] "+v"(px), [py] "+v"(py), [pz] "+v"(pz), [pw] "+v"(pw), [rc] "+v"(rc), [ic] "+v"(ic) \\\n')
f.write('        : [e] "s"(e), [vp] "s"(vp), [hp] "s"(hp), [voff] "v"(voff), [low_half] "s"(low_half), \\\n')
f.write('          [rho_u] "v"(rho_u), [rho2] "v"(rho2), [rho] "v"(rho), [inv_rho] "v"(inv_rho), [zr] "v"(zr), [zi] "v"(zi) \\\n')
f.write("        : " + ", ".join(f'"{c}"' for c in clob) + ")\n")
text = f.getvalue()
# written only when the content changes: the header's mtime is a build dependency of every kernel object (__graft_entry__.build),
# and tests/test_codegen.py imports this module on every run
if not os.path.exists(OUT) or open(OUT).read() != text:
    with open(OUT, "w") as out_f:
        out_f.write(text)
print(f"wrote {OUT}: {len(lines)} instructions / labels")
