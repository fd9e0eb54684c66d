// dpp_rate.hip - issue cost of v_fmac_f64 by operand source on gfx950: SGPR operand, VGPR operand, DPP row_newbcast operand, and mixes.
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_rate.hip -o tools/_bin/dpp_rate && tools/_bin/dpp_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define DPP(acc, tab, x, k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x))
#define VV(acc, tab, x) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(tab), "v"(x))
#define SV(acc, s, x) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "s"(s), "v"(x))
#define MOVD(dst, tab, k) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(tab))

template <int MODE>
__global__ __launch_bounds__(1024) void k(const double *in, double *out, long long *cyc, int iters) {
    const int lane = threadIdx.x & 63;
    double t0 = in[lane & 15], t1 = in[16 + (lane & 15)];
    double x = in[32 + lane], y = in[96 + lane];
    double s = in[200];
    s = __builtin_bit_cast(double, ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(__builtin_bit_cast(unsigned long long, s) >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(unsigned long long, s)));
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    __syncthreads();
    const long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 16 DPP ops, 8 independent accumulators
            DPP(a0, t0, x, 0); DPP(a1, t0, x, 1); DPP(a2, t0, x, 2); DPP(a3, t0, x, 3); DPP(a4, t0, x, 4); DPP(a5, t0, x, 5); DPP(a6, t0, x, 6); DPP(a7, t0, x, 7);
            DPP(a0, t1, y, 8); DPP(a1, t1, y, 9); DPP(a2, t1, y, 10); DPP(a3, t1, y, 11); DPP(a4, t1, y, 12); DPP(a5, t1, y, 13); DPP(a6, t1, y, 14); DPP(a7, t1, y, 15);
        } else if (MODE == 1) {  // VGPR operands
            VV(a0, t0, x); VV(a1, t0, x); VV(a2, t0, x); VV(a3, t0, x); VV(a4, t0, x); VV(a5, t0, x); VV(a6, t0, x); VV(a7, t0, x);
            VV(a0, t1, y); VV(a1, t1, y); VV(a2, t1, y); VV(a3, t1, y); VV(a4, t1, y); VV(a5, t1, y); VV(a6, t1, y); VV(a7, t1, y);
        } else if (MODE == 2) {  // SGPR operand
            SV(a0, s, x); SV(a1, s, x); SV(a2, s, x); SV(a3, s, x); SV(a4, s, x); SV(a5, s, x); SV(a6, s, x); SV(a7, s, x);
            SV(a0, s, y); SV(a1, s, y); SV(a2, s, y); SV(a3, s, y); SV(a4, s, y); SV(a5, s, y); SV(a6, s, y); SV(a7, s, y);
        } else if (MODE == 3) {  // alternate DPP / SGPR
            DPP(a0, t0, x, 0); SV(a1, s, x); DPP(a2, t0, x, 2); SV(a3, s, x); DPP(a4, t0, x, 4); SV(a5, s, x); DPP(a6, t0, x, 6); SV(a7, s, x);
            DPP(a0, t1, y, 8); SV(a1, s, y); DPP(a2, t1, y, 10); SV(a3, s, y); DPP(a4, t1, y, 12); SV(a5, s, y); DPP(a6, t1, y, 14); SV(a7, s, y);
        } else if (MODE == 4) {  // v_mov_b64_dpp broadcast then plain fmac: 8 movs + 16 fmacs (two lane sets would share the movs)
            double b0, b1, b2, b3, b4, b5, b6, b7;
            MOVD(b0, t0, 0); MOVD(b1, t0, 1); MOVD(b2, t0, 2); MOVD(b3, t0, 3); MOVD(b4, t0, 4); MOVD(b5, t0, 5); MOVD(b6, t0, 6); MOVD(b7, t0, 7);
            VV(a0, b0, x); VV(a1, b1, x); VV(a2, b2, x); VV(a3, b3, x); VV(a4, b4, x); VV(a5, b5, x); VV(a6, b6, x); VV(a7, b7, x);
            VV(a0, b0, y); VV(a1, b1, y); VV(a2, b2, y); VV(a3, b3, y); VV(a4, b4, y); VV(a5, b5, y); VV(a6, b6, y); VV(a7, b7, y);
        } else if (MODE == 5) {  // 16 v_mov_b64_dpp only
            double b0, b1, b2, b3, b4, b5, b6, b7;
            MOVD(b0, t0, 0); MOVD(b1, t0, 1); MOVD(b2, t0, 2); MOVD(b3, t0, 3); MOVD(b4, t0, 4); MOVD(b5, t0, 5); MOVD(b6, t0, 6); MOVD(b7, t0, 7);
            a0 += b0; 
            MOVD(b0, t1, 0); MOVD(b1, t1, 1); MOVD(b2, t1, 2); MOVD(b3, t1, 3); MOVD(b4, t1, 4); MOVD(b5, t1, 5); MOVD(b6, t1, 6); MOVD(b7, t1, 7);
            a1 += b7;
        } else if (MODE == 6) {  // DPP with the SAME accumulator chain (dependent)
            DPP(a0, t0, x, 0); DPP(a0, t0, x, 1); DPP(a0, t0, x, 2); DPP(a0, t0, x, 3); DPP(a0, t0, x, 4); DPP(a0, t0, x, 5); DPP(a0, t0, x, 6); DPP(a0, t0, x, 7);
            DPP(a0, t1, y, 8); DPP(a0, t1, y, 9); DPP(a0, t1, y, 10); DPP(a0, t1, y, 11); DPP(a0, t1, y, 12); DPP(a0, t1, y, 13); DPP(a0, t1, y, 14); DPP(a0, t1, y, 15);
        } else if (MODE == 7) {  // plain dependent chain
            VV(a0, t0, x); VV(a0, t0, x); VV(a0, t0, x); VV(a0, t0, x); VV(a0, t0, x); VV(a0, t0, x); VV(a0, t0, x); VV(a0, t0, x);
            VV(a0, t1, y); VV(a0, t1, y); VV(a0, t1, y); VV(a0, t1, y); VV(a0, t1, y); VV(a0, t1, y); VV(a0, t1, y); VV(a0, t1, y);
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0;
}

static int g_blocks = 256;
template <int MODE>
static void run(const char *name, int nops, const double *in, double *out, long long *cyc) {
    for (int waves : {4, 16}) {
        const int iters = 200000, blocks = g_blocks;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k<MODE><<<blocks, waves * 64>>>(in, out, cyc, 100);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<MODE><<<blocks, waves * 64>>>(in, out, cyc, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long h[16];
        CHECK(hipMemcpy(h, cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
        // wall cycles at 2.4 GHz per op and SIMD: ms * 2.4e6 / (iters * nops * waves/4)
        printf("[%3d workgroups] %-30s waves/SIMD %d: %.2f SIMD cycles per op (events, 2.4 GHz), wave-0 counter %lld per iteration\n", blocks, name, waves / 4,
               ms * 2.4e6 / ((double)iters * nops * (waves / 4)), h[0] / iters);
    }
}

int main(int argc, char **argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    double h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1e-3 * (i + 1);
    double *in, *out; long long *cyc;
    CHECK(hipMalloc(&in, sizeof(h))); CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 256 * 1024 * 8)); CHECK(hipMalloc(&cyc, 256 * 16 * 8));
    run<0>("v_fmac_f64_dpp row_newbcast", 16, in, out, cyc);
    run<1>("v_fmac_f64 vgpr,vgpr", 16, in, out, cyc);
    run<2>("v_fmac_f64 sgpr,vgpr", 16, in, out, cyc);
    run<3>("alternating dpp / sgpr", 16, in, out, cyc);
    run<4>("8 v_mov_b64_dpp + 16 fmac", 24, in, out, cyc);
    run<5>("v_mov_b64_dpp (16) + 2 add", 18, in, out, cyc);
    run<6>("dpp dependent chain", 16, in, out, cyc);
    run<7>("vgpr dependent chain", 16, in, out, cyc);
    return 0;
}
