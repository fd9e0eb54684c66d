// harm_microbench.hip — the column recursion of propagate_kernel.hip in isolation: what does ONE batch of table entries cost a wave,
// as a function of the waves per SIMD and of how the table reaches the scalar registers?  Standalone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/harm_microbench.hip -o gpurun_out/harm_microbench && gpurun_out/harm_microbench
// Every workgroup (one per CU: 150 KB of LDS requested) streams the same 143 KB table, wave w of a workgroup its own sixteenth, like the
// column waves of the propagation kernel; nothing is reused in the 16 KB scalar cache between passes (15 other waves went through it).
// Variants: 0 = five entries per wait + touch of the next batch (the kernel's loop), 1 = the same without the touch, 2 = two windows of
// three entries, the next one in flight while this one is evaluated, 3 = variant 0 with TWO lane sets per table pass, 4 = variant 2 with
// two lane sets, 8 = the table through VGPRs, sixteen doubles per register pair (one per lane of a 16-lane row), every operand picked
// by the DPP row_newbcast of v_fmac_f64 (no scalar data path at all), 16 entries = 7 coalesced 128-byte loads per batch, double-buffered;
// 9 = variant 8 with two lane sets.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
#define CAS __attribute__((address_space(4)))
typedef const double CAS *TabPtr;
#define D(v, i) __builtin_bit_cast(double, (v2i){(v)[(i)], (v)[(i) + 1]})

struct B5 { v16i q0, q1, q2, q3; v4i q4; v2i q5; };
struct B3 { v16i q0, q1; v8i q2; v2i q3; };

__device__ __forceinline__ void load5(TabPtr e, B5 &b) {
    asm volatile("s_load_dwordx16 %0, %6, 0x0\n\ts_load_dwordx16 %1, %6, 0x40\n\ts_load_dwordx16 %2, %6, 0x80\n\t"
                 "s_load_dwordx16 %3, %6, 0xc0\n\ts_load_dwordx4 %4, %6, 0x100\n\ts_load_dwordx2 %5, %6, 0x110\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(b.q0), "=&s"(b.q1), "=&s"(b.q2), "=&s"(b.q3), "=&s"(b.q4), "=&s"(b.q5) : "s"(e) : "memory");
}
__device__ __forceinline__ void touch5(TabPtr e, int &sink) {
    asm volatile("s_load_dword %0, %1, 0x118\n\ts_load_dword %0, %1, 0x158\n\ts_load_dword %0, %1, 0x198\n\ts_load_dword %0, %1, 0x1d8\n\t"
                 "s_load_dword %0, %1, 0x218\n\ts_load_dword %0, %1, 0x22c" : "+&s"(sink) : "s"(e) : "memory");
}
// 168 bytes, no wait.  `pin` (the head of the recursion) is named as in/out by the issue and by the wait, so that the evaluation of the
// OTHER window, which hangs on it, stays between the two: left alone the scheduler hoists it above the issue or sinks it below the wait
__device__ __forceinline__ void issue3(TabPtr e, B3 &b, double &pin) {
    asm volatile("s_load_dwordx16 %0, %5, 0x0\n\ts_load_dwordx16 %1, %5, 0x40\n\ts_load_dwordx8 %2, %5, 0x80\n\ts_load_dwordx2 %3, %5, 0xa0"
                 : "=&s"(b.q0), "=&s"(b.q1), "=&s"(b.q2), "=&s"(b.q3), "+v"(pin) : "s"(e) : "memory");
}
// the wait names the window as in/out: its registers are not read before, and not reallocated across, this point
__device__ __forceinline__ void wait3(B3 &b, double &pin, double &pin2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b.q0), "+s"(b.q1), "+s"(b.q2), "+s"(b.q3), "+v"(pin), "+v"(pin2) : : "memory");
}

struct B6 { v16i q0, q1, q2, q3; v8i q4; };  // six 48-byte entries
__device__ __forceinline__ void load6(TabPtr e, B6 &b) {
    asm volatile("s_load_dwordx16 %0, %5, 0x0\n\ts_load_dwordx16 %1, %5, 0x40\n\ts_load_dwordx16 %2, %5, 0x80\n\t"
                 "s_load_dwordx16 %3, %5, 0xc0\n\ts_load_dwordx8 %4, %5, 0x100\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(b.q0), "=&s"(b.q1), "=&s"(b.q2), "=&s"(b.q3), "=&s"(b.q4) : "s"(e) : "memory");
}

template <int L>
struct Acc { double a1[L], a2[L], s1[L], s2[L], s3[L], s4[L], s5[L], s6[L], rho_u[L], rho2[L]; };

#define TERM(g, t1, t2, t3, t4, t5, t6)                                                        \
    _Pragma("unroll") for (int l = 0; l < L; ++l) {                                            \
        const double an = __builtin_fma(A.rho_u[l], A.a1[l], -((A.rho2[l] * (g)) * A.a2[l]));  \
        A.s1[l] = __builtin_fma(an, (t1), A.s1[l]); A.s2[l] = __builtin_fma(an, (t2), A.s2[l]); \
        A.s3[l] = __builtin_fma(an, (t3), A.s3[l]); A.s4[l] = __builtin_fma(an, (t4), A.s4[l]); \
        A.s5[l] = __builtin_fma(an, (t5), A.s5[l]); A.s6[l] = __builtin_fma(an, (t6), A.s6[l]); \
        A.a2[l] = A.a1[l]; A.a1[l] = an;                                                       \
    }
#define TERMS5(b)                                                                                            \
    TERM(D(b.q0, 0), D(b.q0, 2), D(b.q0, 4), D(b.q0, 6), D(b.q0, 8), D(b.q0, 10), D(b.q0, 12))               \
    TERM(D(b.q0, 14), D(b.q1, 0), D(b.q1, 2), D(b.q1, 4), D(b.q1, 6), D(b.q1, 8), D(b.q1, 10))               \
    TERM(D(b.q1, 12), D(b.q1, 14), D(b.q2, 0), D(b.q2, 2), D(b.q2, 4), D(b.q2, 6), D(b.q2, 8))               \
    TERM(D(b.q2, 10), D(b.q2, 12), D(b.q2, 14), D(b.q3, 0), D(b.q3, 2), D(b.q3, 4), D(b.q3, 6))              \
    TERM(D(b.q3, 8), D(b.q3, 10), D(b.q3, 12), D(b.q3, 14), D(b.q4, 0), D(b.q4, 2), D(b.q5, 0))
#define TERMS3(b)                                                                                            \
    TERM(D(b.q0, 0), D(b.q0, 2), D(b.q0, 4), D(b.q0, 6), D(b.q0, 8), D(b.q0, 10), D(b.q0, 12))               \
    TERM(D(b.q0, 14), D(b.q1, 0), D(b.q1, 2), D(b.q1, 4), D(b.q1, 6), D(b.q1, 8), D(b.q1, 10))               \
    TERM(D(b.q1, 12), D(b.q1, 14), D(b.q2, 0), D(b.q2, 2), D(b.q2, 4), D(b.q2, 6), D(b.q3, 0))

// 48-byte entry (g, t1, t2, t3, t4, kappa): the w sums reuse the PREVIOUS row's (t3, t4) with the row's own factor kappa - ten operations
#define TERM48(g, t1, t2, t3, t4, kap, p3, p4)                                                 \
    _Pragma("unroll") for (int l = 0; l < L; ++l) {                                            \
        const double an = __builtin_fma(A.rho_u[l], A.a1[l], -((A.rho2[l] * (g)) * A.a2[l]));  \
        const double ak = an * (kap);                                                          \
        A.s1[l] = __builtin_fma(an, (t1), A.s1[l]); A.s2[l] = __builtin_fma(an, (t2), A.s2[l]); \
        A.s3[l] = __builtin_fma(an, (t3), A.s3[l]); A.s4[l] = __builtin_fma(an, (t4), A.s4[l]); \
        A.s5[l] = __builtin_fma(ak, (p3), A.s5[l]); A.s6[l] = __builtin_fma(ak, (p4), A.s6[l]); \
        A.a2[l] = A.a1[l]; A.a1[l] = an;                                                       \
    }
#define E48(b0, i0, b1, i1, b2, i2, b3, i3, b4, i4, b5, i5, p3, p4) TERM48(D(b0, i0), D(b1, i1), D(b2, i2), D(b3, i3), D(b4, i4), D(b5, i5), p3, p4)


// ---- DPP-fed variant: the table lives in VGPRs, 16 doubles per register pair (lane l of every 16-lane row holds double l of a 128-byte
// line), and v_fmac_f64_dpp row_newbcast:k multiplies by double k of the row: wave-uniform operands without the scalar data path.
// a_n = rho_u a_{n-1} + g_n (-rho2 a_{n-2}): mul, fmac(dpp), mul, 6 fmac(dpp) = nine operations per entry, like the scalar form.
struct V7 { double r[7]; };
template <int K>
__device__ __forceinline__ void fmac_bc(double &acc, double tab, double x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x), "n"(K));
}
template <int L, int E>
__device__ __forceinline__ void term_dpp(Acc<L> &A, double (&np1)[L], double (&np2)[L], const V7 &v) {
    constexpr int f = 7 * E;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        double an = A.rho_u[l] * A.a1[l];
        fmac_bc<(f + 0) & 15>(an, v.r[(f + 0) >> 4], np2[l]);
        fmac_bc<(f + 1) & 15>(A.s1[l], v.r[(f + 1) >> 4], an);
        fmac_bc<(f + 2) & 15>(A.s2[l], v.r[(f + 2) >> 4], an);
        fmac_bc<(f + 3) & 15>(A.s3[l], v.r[(f + 3) >> 4], an);
        fmac_bc<(f + 4) & 15>(A.s4[l], v.r[(f + 4) >> 4], an);
        fmac_bc<(f + 5) & 15>(A.s5[l], v.r[(f + 5) >> 4], an);
        fmac_bc<(f + 6) & 15>(A.s6[l], v.r[(f + 6) >> 4], an);
        np2[l] = np1[l]; np1[l] = A.rho2[l] * an;  // (rho2 holds -rho^2 here)
        A.a1[l] = an;
    }
}
template <int L>
__device__ __forceinline__ void terms16_dpp(Acc<L> &A, double (&np1)[L], double (&np2)[L], const V7 &v) {
    term_dpp<L, 0>(A, np1, np2, v); term_dpp<L, 1>(A, np1, np2, v); term_dpp<L, 2>(A, np1, np2, v); term_dpp<L, 3>(A, np1, np2, v);
    term_dpp<L, 4>(A, np1, np2, v); term_dpp<L, 5>(A, np1, np2, v); term_dpp<L, 6>(A, np1, np2, v); term_dpp<L, 7>(A, np1, np2, v);
    term_dpp<L, 8>(A, np1, np2, v); term_dpp<L, 9>(A, np1, np2, v); term_dpp<L, 10>(A, np1, np2, v); term_dpp<L, 11>(A, np1, np2, v);
    term_dpp<L, 12>(A, np1, np2, v); term_dpp<L, 13>(A, np1, np2, v); term_dpp<L, 14>(A, np1, np2, v); term_dpp<L, 15>(A, np1, np2, v);
}
typedef const __attribute__((address_space(1))) double *GTab;
__device__ __forceinline__ void vload7(GTab p, V7 &v) {
#pragma unroll
    for (int j = 0; j < 7; ++j) v.r[j] = p[16 * j];
}

// ---- hybrid: KS of the seven values of an entry through the scalar path (g first), the other 7 - KS through the DPP broadcast.
// Scalar side: 8 entries per wait (KS * 64 bytes); vector side: 16 entries per register pair (value j of entry e = lane e of pair j).
template <int KS> struct SB;  // KS x 16 dwords
template <> struct SB<3> { v16i q[3]; };
template <> struct SB<4> { v16i q[4]; };
template <> struct SB<5> { v16i q[5]; };
__device__ __forceinline__ void sload(TabPtr e, SB<3> &b) {
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx16 %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(b.q[0]), "=&s"(b.q[1]), "=&s"(b.q[2]) : "s"(e) : "memory");
}
__device__ __forceinline__ void sload(TabPtr e, SB<4> &b) {
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(b.q[0]), "=&s"(b.q[1]), "=&s"(b.q[2]), "=&s"(b.q[3]) : "s"(e) : "memory");
}
__device__ __forceinline__ void sload(TabPtr e, SB<5> &b) {
    asm volatile("s_load_dwordx16 %0, %5, 0x0\n\ts_load_dwordx16 %1, %5, 0x40\n\ts_load_dwordx16 %2, %5, 0x80\n\ts_load_dwordx16 %3, %5, 0xc0\n\t"
                 "s_load_dwordx16 %4, %5, 0x100\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(b.q[0]), "=&s"(b.q[1]), "=&s"(b.q[2]), "=&s"(b.q[3]), "=&s"(b.q[4]) : "s"(e) : "memory");
}
template <int KS> struct VD { double r[7 - KS]; };
template <int KS>
__device__ __forceinline__ void vloadk(GTab p, VD<KS> &v) {
#pragma unroll
    for (int j = 0; j < 7 - KS; ++j) v.r[j] = p[16 * j];
}
// scalar value i (0 .. KS-1) of entry E (0..7) of a half batch: dword 2 * (KS * E + i)
#define SVAL(b, E, i) D((b).q[(2 * (KS * (E) + (i))) >> 4], (2 * (KS * (E) + (i))) & 15)
template <int L, int KS, int E, int LANE0>
__device__ __forceinline__ void term_hyb(Acc<L> &A, const SB<KS> &b, const VD<KS> &v) {
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const double an = __builtin_fma(A.rho_u[l], A.a1[l], -((A.rho2[l] * SVAL(b, E, 0)) * A.a2[l]));
        double *acc[6] = {&A.s1[l], &A.s2[l], &A.s3[l], &A.s4[l], &A.s5[l], &A.s6[l]};
        if (KS > 1) *acc[0] = __builtin_fma(an, SVAL(b, E, 1 < KS ? 1 : 0), *acc[0]);
        if (KS > 2) *acc[1] = __builtin_fma(an, SVAL(b, E, 2 < KS ? 2 : 0), *acc[1]);
        if (KS > 3) *acc[2] = __builtin_fma(an, SVAL(b, E, 3 < KS ? 3 : 0), *acc[2]);
        if (KS > 4) *acc[3] = __builtin_fma(an, SVAL(b, E, 4 < KS ? 4 : 0), *acc[3]);
#pragma unroll
        for (int j = 0; j < 7 - KS; ++j) fmac_bc<LANE0 + E>(*acc[KS - 1 + j], v.r[j], an);
        A.a2[l] = A.a1[l]; A.a1[l] = an;
    }
}
template <int L, int KS, int LANE0>
__device__ __forceinline__ void terms8_hyb(Acc<L> &A, const SB<KS> &b, const VD<KS> &v) {
    term_hyb<L, KS, 0, LANE0>(A, b, v); term_hyb<L, KS, 1, LANE0>(A, b, v); term_hyb<L, KS, 2, LANE0>(A, b, v); term_hyb<L, KS, 3, LANE0>(A, b, v);
    term_hyb<L, KS, 4, LANE0>(A, b, v); term_hyb<L, KS, 5, LANE0>(A, b, v); term_hyb<L, KS, 6, LANE0>(A, b, v); term_hyb<L, KS, 7, LANE0>(A, b, v);
}
template <int L, int KS>
__device__ __forceinline__ void hybrid_pass(Acc<L> &A, uint64_t base, int lane, int entries_per_wave) {
    // scalar part: KS * 8 bytes per entry at `base`; vector part: (7 - KS) * 8 bytes per entry behind it (same wave's share of the table)
    TabPtr e = (TabPtr)base;
    GTab pv = (GTab)(base + (uint64_t)entries_per_wave * KS * 8) + (lane & 15);
    VD<KS> va, vb;
    vloadk<KS>(pv, va);
    const int nbt = entries_per_wave / 16;
    for (int b = 0; b + 1 < nbt; b += 2, pv += 2 * 16 * (7 - KS)) {
        vloadk<KS>(pv + 16 * (7 - KS), vb);
        { SB<KS> w; sload(e, w); terms8_hyb<L, KS, 0>(A, w, va); e += 8 * KS; }
        { SB<KS> w; sload(e, w); terms8_hyb<L, KS, 8>(A, w, va); e += 8 * KS; }
        vloadk<KS>(pv + 2 * 16 * (7 - KS), va);
        { SB<KS> w; sload(e, w); terms8_hyb<L, KS, 0>(A, w, vb); e += 8 * KS; }
        { SB<KS> w; sload(e, w); terms8_hyb<L, KS, 8>(A, w, vb); e += 8 * KS; }
    }
    if (nbt & 1) {
        { SB<KS> w; sload(e, w); terms8_hyb<L, KS, 0>(A, w, va); e += 8 * KS; }
        { SB<KS> w; sload(e, w); terms8_hyb<L, KS, 8>(A, w, va); e += 8 * KS; }
    }
}

// entries: multiples of 180 so that every variant evaluates the same rows
template <int VARIANT, int L>
__global__ __launch_bounds__(1024) void bench(const double *tab_g, int entries_per_wave, int passes, double *out) {
    extern __shared__ double lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    Acc<L> A;
    for (int l = 0; l < L; ++l) {
        A.rho_u[l] = 0.3 + 1e-3 * lane + 0.01 * l; A.rho2[l] = (VARIANT == 8 || VARIANT == 9) ? -0.81 : 0.81; A.a1[l] = 0.0; A.a2[l] = 1.0 + 1e-3 * lane;
        A.s1[l] = A.s2[l] = A.s3[l] = A.s4[l] = A.s5[l] = A.s6[l] = 0.0;
    }
    const uint64_t base = (uint64_t)tab_g + (VARIANT == 7 ? 0 : (uint64_t)(wave & 15) * (uint64_t)entries_per_wave * 56);
    for (int p = 0; p < passes; ++p) {
        TabPtr e = (TabPtr)base;
        if (VARIANT >= 10) {
            hybrid_pass<L, VARIANT == 10 || VARIANT == 13 ? 4 : VARIANT == 11 ? 5 : 3>(A, base, lane, entries_per_wave);
        } else if (VARIANT == 8 || VARIANT == 9) {
            GTab pl = (GTab)base + (lane & 15);
            double np1[L], np2[L];
            for (int l = 0; l < L; ++l) { np1[l] = A.rho2[l] * A.a1[l]; np2[l] = A.rho2[l] * A.a2[l]; }
            V7 va, vb;
            vload7(pl, va);
            const int nbt = entries_per_wave / 16;  // whole batches of 16 entries (112 doubles = 7 lines of 128 bytes)
            for (int b = 0; b + 1 < nbt; b += 2, pl += 224) {
                vload7(pl + 112, vb);
                terms16_dpp<L>(A, np1, np2, va);
                vload7(pl + 224, va);  // (padded table: the last one reads past the wave's share)
                terms16_dpp<L>(A, np1, np2, vb);
            }
            if (nbt & 1) terms16_dpp<L>(A, np1, np2, va);
        } else if (VARIANT == 5 || VARIANT == 6) {  // 48-byte entries, six per wait (the same number of BYTES per wave as the others: 7/6 of the rows)
            double c3 = 1e-4, c4 = -1e-4;    // (t3, t4) of the last row of the previous batch: scalar registers
            for (int b = 0; b < entries_per_wave * 7 / 36; ++b, e += 36) {
                B6 w;
                load6(e, w);
                E48(w.q0, 0, w.q0, 2, w.q0, 4, w.q0, 6, w.q0, 8, w.q0, 10, c3, c4)
                E48(w.q0, 12, w.q0, 14, w.q1, 0, w.q1, 2, w.q1, 4, w.q1, 6, D(w.q0, 6), D(w.q0, 8))
                E48(w.q1, 8, w.q1, 10, w.q1, 12, w.q1, 14, w.q2, 0, w.q2, 2, D(w.q1, 2), D(w.q1, 4))
                E48(w.q2, 4, w.q2, 6, w.q2, 8, w.q2, 10, w.q2, 12, w.q2, 14, D(w.q1, 14), D(w.q2, 0))
                E48(w.q3, 0, w.q3, 2, w.q3, 4, w.q3, 6, w.q3, 8, w.q3, 10, D(w.q2, 10), D(w.q2, 12))
                E48(w.q3, 12, w.q3, 14, w.q4, 0, w.q4, 2, w.q4, 4, w.q4, 6, D(w.q3, 6), D(w.q3, 8))
                c3 = D(w.q4, 2); c4 = D(w.q4, 4);
            }
        } else if (VARIANT == 0 || VARIANT == 1 || VARIANT == 3 || VARIANT == 7) {
            int sink = 0;
            for (int b = 0; b < entries_per_wave / 5; ++b, e += 35) {
                if (VARIANT == 7 && (b & 7) == 0) e = (TabPtr)base;  // 8 batches = 2.2 KB, every wave the same bytes: scalar-cache hits
                B5 w;
                load5(e, w);
                if (VARIANT != 1) touch5(e, sink);
                TERMS5(w)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sink) : : "memory");
        } else {
            B3 wa, wb;
            issue3(e, wa, A.a1[0]);
            wait3(wa, A.a1[0], A.s6[0]);
            for (int b = 0; b < entries_per_wave / 6; ++b, e += 42) {
                issue3(e + 21, wb, A.a1[0]);
                TERMS3(wa)
                wait3(wb, A.a1[L - 1], A.s6[L - 1]);
                issue3(e + 42, wa, A.a1[0]);  // (the table is padded: the last issue of a pass reads past the wave's share)
                TERMS3(wb)
                wait3(wa, A.a1[L - 1], A.s6[L - 1]);
            }
        }
    }
    double r = 0.0;
    for (int l = 0; l < L; ++l) r += A.s1[l] + A.s2[l] + A.s3[l] + A.s4[l] + A.s5[l] + A.s6[l] + A.a1[l];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && out == nullptr) lds[0] = r;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VARIANT, int L>
static void run(const char *name, const double *tab, double *out, int waves, int blocks, int entries, int passes) {
    const size_t lds = 150 * 1024;
    CHECK(hipFuncSetAttribute((const void *)bench<VARIANT, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    bench<VARIANT, L><<<blocks, waves * 64, lds>>>(tab, entries, 2, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(t0));
    bench<VARIANT, L><<<blocks, waves * 64, lds>>>(tab, entries, passes, out);
    CHECK(hipEventRecord(t1));
    CHECK(hipEventSynchronize(t1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, t0, t1));
    const double clk = 2.4e9, cyc = ms * 1e-3 * clk;
    const double ent = (double)(VARIANT >= 8 ? entries / 16 * 16 : entries) * passes;  // entries per wave
    const double rows = VARIANT == 5 || VARIANT == 6 ? 7.0 / 6.0 : 1.0, ops = VARIANT == 5 || VARIANT == 6 ? 10.0 : 9.0;
    const double valu = ent * rows * ops * 4.0 * L * (waves / 4.0);   // VALU cycles a SIMD must issue
    double chk = 0.0;
    std::vector<double> h((size_t)blocks * waves * 64);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    for (double v : h) chk += v;
    printf("%-28s waves/SIMD %d lane sets %d: %8.3f ms, %7.1f cycles per 5 entries and wave, %6.1f cycles per entry and lane set on a SIMD, "
           "f64 issue %.3f, table %.2f B/cycle/CU  (chk %.6e)\n",
           name, waves / 4, L, ms, cyc / (ent * rows) * 5.0, cyc / (ent * rows * L * (waves / 4.0)), valu / cyc, ent * 56.0 * waves / cyc, chk);
}

int main() {
    const int entries = 180, passes = 400;  // 16 x 180 x 56 B = 161 KB table (config 2: 2 556 entries = 143 KB)
    const size_t n = (size_t)16 * entries * 7 + 4096;
    std::vector<double> h(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (i % 7 == 0) ? 0.2 + 0.1 * (rand() / (double)RAND_MAX) : (rand() / (double)RAND_MAX - 0.5) * 1e-3;
    double *tab, *out;
    CHECK(hipMalloc(&tab, n * 8));
    CHECK(hipMemcpy(tab, h.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)256 * 1024 * 8));
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    const int blocks = pr.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz (2.4 GHz assumed in the cycle figures)\n", pr.name, blocks, pr.clockRate);
    for (int waves : {4, 8, 12, 16}) {
        run<0, 1>("5 per wait + touch", tab, out, waves, blocks, entries, passes);
        run<1, 1>("5 per wait", tab, out, waves, blocks, entries, passes);
        run<2, 1>("3 + 3 double-buffered", tab, out, waves, blocks, entries, passes);
        run<3, 2>("5 per wait + touch, 2 sets", tab, out, waves, blocks, entries, passes);
        run<4, 2>("3 + 3, 2 sets", tab, out, waves, blocks, entries, passes);
        run<5, 1>("48-byte entries, 6 per wait", tab, out, waves, blocks, entries, passes);
        run<6, 2>("48-byte entries, 2 sets", tab, out, waves, blocks, entries, passes);
        run<7, 1>("scalar-cache HITS (5 + touch)", tab, out, waves, blocks, entries, passes);
        run<8, 1>("DPP row_newbcast, 16 per batch", tab, out, waves, blocks, entries, passes);
        run<9, 2>("DPP row_newbcast, 2 sets", tab, out, waves, blocks, entries, passes);
        run<12, 1>("hybrid 3 scalar + 4 DPP", tab, out, waves, blocks, entries, passes);
        run<10, 1>("hybrid 4 scalar + 3 DPP", tab, out, waves, blocks, entries, passes);
        run<11, 1>("hybrid 5 scalar + 2 DPP", tab, out, waves, blocks, entries, passes);
        run<13, 2>("hybrid 4 + 3, 2 sets", tab, out, waves, blocks, entries, passes);
    }
    // one workgroup only: the table path with no other CU on the L2
    run<0, 1>("ONE workgroup: 5 + touch", tab, out, 16, 1, entries, passes);
    run<2, 1>("ONE workgroup: 3 + 3", tab, out, 16, 1, entries, passes);
    return 0;
}
