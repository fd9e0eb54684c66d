// moments_kernel.hip — ensemble moments of the final states, on the device.
//
// What the consumers of a Monte Carlo result reduce the ensemble to (reference mc/results.rs:60-245: the dispersions and
// the covariance of the final states; north_star: "RCCL ... only for the final trajectory/covariance reduction"):
//   out[0]       count of the runs whose status is OK
//   out[1..9]    sum (x - x0)            x = [x y z vx vy vz Cr Cd prop_mass]
//   out[10..54]  upper triangle (row-major, i <= j) of sum (x - x0)(x - x0)^T
// 55 doubles: a rank-sharded host adds them up with ONE ncclAllReduce(sum) and no D2H of the ensemble; mean = x0 + s / n,
// cov = (S - n m m^T) / (n - 1).  x0 is the caller's reference point (any state near the ensemble: it keeps the sums well
// conditioned - the nominal, or the first successful run, which every rank can hold).
//
// HBM-bound by construction (9 coalesced f64 reads + a status word per trajectory, 76 B; 380 KB for 5 000 states: one
// launch of microseconds).  Lane <-> trajectory, 55 accumulators in registers, a fixed grid of MOM_BLOCKS x 256 threads
// with a strided walk, butterfly sums inside a wave, the four waves of a block and the blocks added in index order by a
// second one-block launch: the result depends on n only - bit-reproducible from run to run, no atomics.
#include <hip/hip_runtime.h>

#include "moments_args.h"

__global__ __launch_bounds__(MOM_THREADS) void nyx_moments_partial_kernel(MomArgs a) {
    __shared__ double red[MOM_THREADS / 64][MOM_N];
    double acc[MOM_N];
#pragma unroll
    for (int q = 0; q < MOM_N; ++q) acc[q] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * MOM_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * MOM_THREADS + threadIdx.x; i < a.n; i += stride) {
        if (a.status && a.status[i] != 0) continue;
        double d[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = (a.f[k] ? a.f[k][i] : 0.0) - a.x0[k];
        acc[0] += 1.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[1 + k] += d[k];
        int q = 10;
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = r; c < 9; ++c) acc[q++] += d[r] * d[c];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < MOM_N; ++q) {
        double v = acc[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < MOM_N) {
        double s = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < MOM_THREADS / 64; ++w) s += red[w][threadIdx.x];
        a.partial[(int64_t)blockIdx.x * MOM_N + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(64) void nyx_moments_final_kernel(const double *partial, int n_blocks, double *out) {
    if (threadIdx.x < MOM_N) {
        double s = 0.0;
        for (int b = 0; b < n_blocks; ++b) s += partial[(int64_t)b * MOM_N + threadIdx.x];
        out[threadIdx.x] = s;
    }
}

extern "C" hipError_t nyx_launch_moments(const MomArgs &a, double *out, hipStream_t stream) {
    int blocks = (int)((a.n + MOM_THREADS - 1) / MOM_THREADS);
    blocks = blocks < 1 ? 1 : (blocks > MOM_BLOCKS ? MOM_BLOCKS : blocks);
    hipLaunchKernelGGL(nyx_moments_partial_kernel, dim3((unsigned)blocks), dim3(MOM_THREADS), 0, stream, a);
    hipLaunchKernelGGL(nyx_moments_final_kernel, dim3(1), dim3(64), 0, stream, (const double *)a.partial, blocks, out);
    return hipGetLastError();
}
