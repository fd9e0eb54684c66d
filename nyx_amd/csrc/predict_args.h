// predict_args.h — launch arguments of the covariance-mapping kernels (predict_kernel.hip), shared with abi.cpp.
#pragma once
#include <stdint.h>

#include "../../include/nyx_hip.h"

struct PredictArgs {
    int64_t n;
    nyx_hip_predict_t cfg;
    // nominal states of the segment just propagated (device SoA of ctx->out) and its per-launch stats
    const int64_t *epoch;
    const double *s9[9];  // x, y, z, vx, vy, vz, Cr, Cd, prop mass
    double *stm;          // [n][81] column-major: read, then reset to identity
    const int32_t *seg_status;
    const int64_t *seg_n_acc, *seg_n_rej, *seg_n_evals;
    // filter state, trajectory-major
    double *covar;        // [n][81]
    double *state_dev;    // [n][9]
    int64_t *prev_epoch;  // [n] epoch of the previous estimate
    int64_t *init_epoch;  // [n] epoch of the initial estimate (ProcessNoise::init_epoch when the descriptor leaves it open)
    int64_t *dur;         // [n] duration of the NEXT segment (0 = trajectory finished or failed)
    int32_t *status;      // [n] first failing status
    int64_t *acc_n_acc, *acc_n_rej, *acc_n_evals;
    nyx_hip_predict_history_t hist;  // device pointers (any array may be NULL)
};
