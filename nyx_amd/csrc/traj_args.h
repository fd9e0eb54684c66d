// traj_args.h — launch arguments of the trajectory-evaluation kernels (traj_kernel.hip), shared with abi.cpp.
#pragma once
#include <stdint.h>

#include "../../include/nyx_hip.h"

enum { TRAJ_MODE_AT = 0, TRAJ_MODE_EVERY = 1 };

struct TrajEvalArgs {
    nyx_hip_traj_t src;   // device pointers, step-major [k * n + i]
    nyx_hip_traj_t dst;   // device pointers, sample-major [q * n + i]
    int64_t n;            // trajectories
    const int64_t *query; // AT: m shared epochs (device)
    int64_t m;
    int64_t step_ns;      // EVERY
    int32_t *status;      // AT: [m * n]
    int32_t mode;
    int64_t samples_per_block;  // filled by the launcher
};
