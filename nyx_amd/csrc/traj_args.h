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

// nyx_event_search_kernel: Brent on the interpolant between the last published state and the end state
// (propagators/event.rs:178-197), then out = traj.at(event epoch).
struct EventSearchArgs {
    nyx_hip_traj_t traj;  // device; the end state is appended here (event.rs:179)
    int64_t n;
    nyx_hip_event_t ev;
    double mu;
    const int32_t *found;  // [n] set by the propagation kernel
    int32_t *status;       // [n] in: propagation status; out: + EVENT_NOT_FOUND / EVENT_SEARCH
    int64_t *epoch_ns;     // [n] in: end-state epoch; out: event epoch
    double *state[6];      // [n] in: end state; out: interpolated state at the event
};
