// moments_args.h — launch descriptor of nyx_moments_partial_kernel (moments_kernel.hip), shared with abi.cpp.
#ifndef NYX_AMD_MOMENTS_ARGS_H
#define NYX_AMD_MOMENTS_ARGS_H
#include <stdint.h>

#define MOM_N 55        /* count, 9 sums, 45 products (upper triangle, row-major) */
#define MOM_BLOCKS 256  /* fixed grid: one block per CU at most, strided walk */
#define MOM_THREADS 256

struct MomArgs {
    int64_t n;
    const double *f[9];     /* x y z vx vy vz cr cd prop_mass (device, SoA; NULL = zeros) */
    const int32_t *status;  /* device, NULL = every run counts */
    double x0[9];
    double *partial;        /* [MOM_BLOCKS][MOM_N] scratch (device) */
};
#endif
