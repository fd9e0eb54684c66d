// Does a wave64 VALU instruction cost fewer cycles when only the low 32 / 16 lanes are active (EXEC)?  One wave per workgroup,
// a chain of independent v_fma_f64 / dependent v_fma_f64 / v_rcp_f64, timed with the shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(double *out, int64_t *cyc, int active, int mode, int iters) {
    const int lane = threadIdx.x & 63;
    double a0 = 1.0 + lane * 1e-3, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
    const double b = 1.0000001, c = 1e-9;
    int64_t t0 = 0, t1 = 0;
    if (lane < active) {
        t0 = __builtin_readcyclecounter();
        if (mode == 0) {
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
                a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
            }
        } else if (mode == 1) {
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c);
                a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c);
            }
        } else {
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
                a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7);
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double *out; int64_t *cyc;
    hipMalloc(&out, 64 * 8 * 1024); hipMalloc(&cyc, 8 * 1024);
    const int iters = 20000;
    const char *names[3] = {"8 independent v_fma_f64", "dependent v_fma_f64 chain", "8 independent v_rcp_f64"};
    for (int waves = 1; waves <= 2; ++waves)
    for (int mode = 0; mode < 3; ++mode)
        for (int active : {64, 32, 16, 8}) {
            int64_t h[4];
            // `waves` waves on the same SIMD: a workgroup of 64 * (4 * (waves - 1) + 1) threads puts waves 0 and 4 on SIMD 0
            const int threads = 64 * (4 * (waves - 1) + 1);
            for (int rep = 0; rep < 2; ++rep) k<<<1, threads>>>(out, cyc, active, mode, iters);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s)/SIMD  %-28s active lanes %2d: %.2f cycles per instruction (wave 0)\n", waves, names[mode], active, (double)h[0] / (8.0 * iters));
        }
    return 0;
}
