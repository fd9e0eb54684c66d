// abi.cpp — host side of the C-ABI (include/nyx_hip.h): context creation (table building,
// column scheduling, upload), batch staging and kernel launch.  Compiled with hipcc.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/nyx_hip.h"
#include "butcher.h"
#include "devcfg.h"
#include "col_partition.h"
#include "predict_args.h"
#include "traj_args.h"
#include "moments_args.h"

extern "C" size_t nyx_kernel_lds_bytes(int n_waves, int rec_doubles, int stm, int reuse_fields);
extern "C" hipError_t nyx_launch_predict_init(const PredictArgs *a, const int64_t *epoch0, hipStream_t stream);
extern "C" hipError_t nyx_launch_time_update(const PredictArgs *a, hipStream_t stream);
extern "C" hipError_t nyx_launch_event_search(const EventSearchArgs *args, hipStream_t stream);
extern "C" hipError_t nyx_launch_traj_eval(const TrajEvalArgs *args, hipStream_t stream);
extern "C" hipError_t nyx_launch_moments(const MomArgs &a, double *out, hipStream_t stream);
extern "C" hipError_t nyx_launch_frame_shift(const DevCfg *cfg, const double *records, const int32_t *chain_seg, const double *chain_sign,
                                             int n_chain, int64_t n, const int64_t *epoch_ns, double *x, double *y, double *z, double *vx,
                                             double *vy, double *vz, double dir, int32_t *status, const int32_t *prior, const int64_t *dur_ns,
                                             hipStream_t stream);
extern "C" hipError_t nyx_launch_propagate(const DevBatch &bt, const DevCfg *cfg, const HarmEntry *htab,
                                           const ColHdr *cols, const double *records, int n_waves, int rec_lds_doubles,
                                           int reuse_fields, hipStream_t stream, int quad, int no_body_fixed);

// ---------------------------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------------------------

static thread_local char g_err[512] = "";

void nyx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *nyx_hip_last_error(void) { return g_err; }

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            nyx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return NYX_HIP_RC_HIP_ERROR;                                                   \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------

struct DevArrays {  // one SoA batch resident on the device: ONE device block, ONE pinned host mirror => one copy each way
    int64_t cap = 0;
    char *dblock = nullptr, *hblock = nullptr;
    size_t bytes = 0;
    int64_t *epoch = nullptr, *step = nullptr;
    double *f[13] = {nullptr};
    double *stm = nullptr;  // [cap][81], allocated on first STM use
    int64_t stm_cap = 0;
    int64_t *last_step = nullptr, *n_acc = nullptr, *n_rej = nullptr, *n_evals = nullptr;
    double *last_error = nullptr;
    int32_t *status = nullptr, *last_attempts = nullptr;
    template <typename T> T *host(T *dev) const { return (T *)(hblock + ((char *)dev - dblock)); }
};

struct nyx_hip_ctx {
    int device = 0;
    nyx_hip_tuning_t tune = NYX_HIP_TUNING_DEFAULT;  // config.tuning, resolved (see resolve_tuning)
    DevCfg host_cfg;
    DevCfg *d_cfg = nullptr;
    // opts.integration_frame: the states of a batch are centred on another body (its chain w.r.t. the integration centre)
    int swap_n_chain = 0;
    int32_t swap_seg[4] = {0, 0, 0, 0};
    double swap_sign[4] = {0.0, 0.0, 0.0, 0.0};
    double *d_mom = nullptr;   // scratch of the ensemble-moments reduction: [MOM_BLOCKS][MOM_N] block sums, then MOM_N results (host flavour)
    double *d_stm_hist = nullptr;  // NYX_HIP_FLAG_STM_TEXTBOOK: [16][12][stm_hist_cap] stage matrices of the attempt in flight (DevBatch.stm_hist)
    int64_t stm_hist_cap = 0;
    double *d_swap = nullptr;  // six rows of swap_cap doubles: the translated copy of a batch's Cartesian state
    int64_t swap_cap = 0;
    int ed_reuse_fit = 0;  // fields of stage-0 epoch data an unchained pipelined loop may carry between attempts (LDS room)
    HarmEntry *d_htab = nullptr;
    HarmEntry *d_htab2 = nullptr;  // second gravity field
    int terms2 = 0;                // its table rows
    ColHdr *d_cols2 = nullptr;
    double *d_hyb = nullptr;  // the same table in the hybrid-feed layout (devcfg.h HYB_*)
    // Run streams (DevCfg.rs_*): the table once more per column schedule - [0] SOLO, [1] PRIMARY, [2] the helpers' - with every RANGE of
    // a wave starting a sixteen-row group of its own.  Rebuilt when a schedule changes (build_schedule sets rs_dirty), uploaded by
    // launch() before a launch that streams the table.
    std::vector<HarmEntry> h_tab;  // host copy of the entry table (without its tail padding)
    std::vector<ColHdr> h_cols;
    std::vector<double> h_rs[3];   // (kept: the asynchronous upload reads them)
    std::vector<ColHdr> h_rs_cols[3];
    double *d_rs[3] = {nullptr, nullptr, nullptr};
    size_t rs_cap[3] = {0, 0, 0};
    ColHdr *d_rs_cols[3] = {nullptr, nullptr, nullptr};
    bool rs_dirty = true;
    // experiment knobs of the helper dealing (environment, only with NYX_HIP_TUNING_ENV: tools/sweep.py)
    double coop_fast_weight = 4.0 / 3.0;  // speed of a helper column wave on a SIMD that hosts three of them (beside the producer / the answering wave)
    double coop_start_rows = 4.0;         // what the start of one more column on a helper wave is charged, in rows
    int coop_deal = 1;                    // 1: balanced dealing (build_schedule), 0: the longest columns, one per wave
    ColHdr *d_cols = nullptr;
    double *d_records = nullptr;
    std::vector<int32_t> col_len;  // rows per column (index = c)
    int n_waves = 1;
    int forced_waves = 0;
    bool sched_quad = false;  // the schedule / roles in host_cfg were built for the quad layout
    bool sched_dirty = false; // weights changed: rebuild the schedule at the next launch
    // Per-wave column weights, calibrated on this device for every workgroup shape this context has launched (see calibrate()):
    // key = (waves per workgroup, pipelined loop, quad layout, cooperative share in tenths or -1 when working alone)
    typedef std::tuple<int, int, int, int> WKey;
    std::map<WKey, std::array<double, 2 * DEV_MAX_WAVES>> weights;  // [0..16): speed weights, [16..32): measured duties (harmonics-term units)
    std::map<WKey, double> weight_spread;  // (max - min) / mean of the per-wave windows after calibration
    WKey last_key = WKey(0, 0, 0, 0);      // shape of the last launch
    DevArrays cal;                         // scratch outputs of the calibration launches
    bool block_schedule = true;  // one contiguous run of columns per wave where the owner streams the table (fill_schedule)
    bool fit_big = false;        // (tools: NYX_HIP_FIT_BIG - the free-order placement for the large cooperative shape too)
    bool fit_quad = false;       // (tools: NYX_HIP_FIT_QUAD - ... and for the sixteen-wave quad STM shape)
    bool fit_solo = false;       // (tools: NYX_HIP_FIT_SOLO - ... and for workgroups that walk every column themselves)
    bool fit_partition = true;   // ... placed along the column list in a free wave order so that every wave meets its target (fill_schedule)
    bool block_force = false;
    int coop_parts = 1;  // sub-jobs per evaluation of the schedules in host_cfg (1, or 2: two helper workgroups per owner and evaluation; fan-out: 2 .. DEV_FAN_MAX)
    const PredictArgs *fused_pred = nullptr;  // set around the ONE launch of a fused covariance-mapping loop (nyx_hip_predict_until): DEVICE copy of its arguments
    bool coop_fan = false;  // the schedules in host_cfg are those of the fan-out mode: coop_parts DEDICATED helper workgroups per owner (small shards, see launch())
    int forced_quad = -1;  // STM layout: -1 = by ensemble size, 0 = 64 trajectories x D3 per workgroup, 1 = quad layout (16 x 4 lanes, D1)
    double role_handicap[3] = {0.0, 0.0, 0.0};  // integrator, almanac, perturbations (harmonics-term units)
    int role_place[8] = {-1, -1, -1, -1, -1, -1, -1, -1}, role_place_sums = -1, role_place_twobody = -1;  // (tools only: ExpKnobs)
    DevArrays in, out;
    int64_t *d_prof = nullptr;
    CoopBox *d_coop = nullptr;  // cooperative-mode mailboxes, one per trajectory-owning workgroup, then the packed scan words
    int64_t coop_cap = 0;
    int n_cu = 0;
    int last_coop_helpers = 0;  // helpers of the last launch (0 = solo)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = -1.0;
    // Concurrency contract (nyx_hip.h): every entry point locks `mu`, so host threads may share a context; on the DEVICE the
    // launches of one context are chained through `ev_done` (each launch waits for the previous one, whatever its stream),
    // because they share d_cfg, the cooperative-mode mailboxes and the staging blocks.  Concurrent kernels => one ctx each.
    std::recursive_mutex mu;
    hipEvent_t ev_done = nullptr;
    bool launched = false;
};
#define CTX_LOCK(ctx) std::lock_guard<std::recursive_mutex> lock_((ctx)->mu)

// Cooperative-mode mailboxes live in UNCACHED device memory (hipExtMallocWithFlags).  The blocks are pooled per process and device:
// a context borrows one at its first cooperative launch and returns it when it is destroyed; a pooled block is never freed before
// the process ends.  History: in round 3 the ~17th cooperative context of one process never finished (only behind ~90 other tests),
// the pool and the zeroing of a workgroup's LDS went in together, and the hang was gone - attributed, without a reproducer, to
// allocate / free churn of this kind of memory.  Round 4 looked for it and did NOT find it there: tools/uncached_churn.hip (the
// exchange alone: 400 allocate / free cycles x 96 pairs x 300 round trips, with and without unrelated allocator traffic, no stall)
// and tests/test_gpu_coop_contexts.py (40 cooperative contexts in one process with the pool switched OFF, debug_flags 0x100000:
// every launch completes, bit-identical results).  What the library relies on is stated in include/nyx_hip.h ("Cooperative mode");
// the pool stays because it saves an allocation and a 7 MB memset per context, not because freeing is known to be unsafe.
struct MailboxBlock { int device; void *ptr; int64_t cap; };
static std::mutex g_mailbox_mu;
static std::vector<MailboxBlock> g_mailbox_free;
static void *mailbox_acquire(int device, int64_t want_cap, int64_t *cap_out, bool pooled = true) {
    if (pooled) {
        std::lock_guard<std::mutex> lk(g_mailbox_mu);
        for (size_t k = 0; k < g_mailbox_free.size(); ++k)
            if (g_mailbox_free[k].device == device && g_mailbox_free[k].cap >= want_cap) {
                void *p = g_mailbox_free[k].ptr;
                *cap_out = g_mailbox_free[k].cap;
                g_mailbox_free.erase(g_mailbox_free.begin() + (long)k);
                return p;
            }
    }
    const int64_t cap = (want_cap + 255) / 256 * 256;
    const size_t bytes = (size_t)cap * (sizeof(CoopBox) + sizeof(CoopOut)) + 3 * (size_t)(cap + 64) * sizeof(uint32_t);  // sets of 16: <= cap + 16 words; the part-1 answers last
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *cap_out = cap;
    return p;
}
static void mailbox_release(int device, void *ptr, int64_t cap, bool pooled = true) {
    if (!ptr) return;
    if (!pooled) { (void)hipFree(ptr); return; }
    std::lock_guard<std::mutex> lk(g_mailbox_mu);
    g_mailbox_free.push_back(MailboxBlock{device, ptr, cap});
}

static void free_arrays(DevArrays &a) {
    (void)hipFree(a.dblock);
    (void)hipHostFree(a.hblock);
    (void)hipFree(a.stm);
    a = DevArrays();
}

static int ensure_arrays(DevArrays &a, int64_t n, bool stats) {
    if (n <= a.cap) return NYX_HIP_RC_OK;
    free_arrays(a);
    const int64_t cap = (std::max<int64_t>(n, 1024) + 63) / 64 * 64;
    const size_t slot = (size_t)cap * 8;
    const size_t n64 = 15 + (stats ? 5 : 0);  // epoch, step, 13 f64 (+ last_step, n_acc, n_rej, n_evals, last_error)
    a.bytes = n64 * slot + (stats ? 2 * (size_t)cap * 4 : 0);
    HIP_TRY(hipMalloc((void **)&a.dblock, a.bytes));
    HIP_TRY(hipMemset(a.dblock, 0, a.bytes));  // (never hand the kernel recycled device memory it might read before writing)
    HIP_TRY(hipHostMalloc((void **)&a.hblock, a.bytes, hipHostMallocDefault));
    char *p = a.dblock;
    a.epoch = (int64_t *)p; p += slot;
    a.step = (int64_t *)p; p += slot;
    for (auto &q : a.f) { q = (double *)p; p += slot; }
    if (stats) {
        a.last_step = (int64_t *)p; p += slot;
        a.n_acc = (int64_t *)p; p += slot;
        a.n_rej = (int64_t *)p; p += slot;
        a.n_evals = (int64_t *)p; p += slot;
        a.last_error = (double *)p; p += slot;
        a.status = (int32_t *)p; p += (size_t)cap * 4;
        a.last_attempts = (int32_t *)p; p += (size_t)cap * 4;
    }
    a.cap = cap;
    return NYX_HIP_RC_OK;
}

extern "C" int32_t nyx_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int64_t nyx_hip_abi_sizeof(int32_t which) {
    switch (which) {
    case 0: return sizeof(nyx_hip_integ_opts_t);
    case 1: return sizeof(nyx_hip_cheby_segment_t);
    case 2: return sizeof(nyx_hip_body_t);
    case 3: return sizeof(nyx_hip_rotation_t);
    case 4: return sizeof(nyx_hip_gravity_field_t);
    case 5: return sizeof(nyx_hip_srp_t);
    case 6: return sizeof(nyx_hip_drag_t);
    case 7: return sizeof(nyx_hip_config_t);
    case 8: return sizeof(nyx_hip_states_t);
    case 9: return sizeof(nyx_hip_step_stats_t);
    case 10: return sizeof(nyx_hip_traj_t);
    case 11: return sizeof(nyx_hip_solid_tides_t);
    case 12: return sizeof(nyx_hip_predict_t);
    case 13: return sizeof(nyx_hip_predict_history_t);
    case 14: return sizeof(nyx_hip_process_noise_t);
    case 15: return sizeof(nyx_hip_tuning_t);
    default: return -1;
    }
}

// nyx_hip_rotation_t -> DevRot (validated by check_rotation() first)
static void copy_rotation(DevRot &d, const nyx_hip_rotation_t &r) {
    std::memset(&d, 0, sizeof d);
    for (int k = 0; k < 3; ++k) { d.ra[k] = r.ra_deg[k]; d.dec[k] = r.dec_deg[k]; d.w[k] = r.w_deg[k]; }
    d.kind = r.kind; d.n_np = r.n_nut_prec;
    for (int k = 0; k < r.n_nut_prec; ++k) {
        d.np_ang[k][0] = r.nut_prec_angle_deg[k][0]; d.np_ang[k][1] = r.nut_prec_angle_deg[k][1];
        d.np_ra[k] = r.nut_prec_ra[k]; d.np_dec[k] = r.nut_prec_dec[k]; d.np_w[k] = r.nut_prec_w[k];
    }
    d.euler_seg = r.euler_segment;
    for (int k = 0; k < 9; ++k) d.base[k] = r.base_dcm[k];
}
static const char *check_rotation(const nyx_hip_rotation_t &r, int n_segments) {
    if (r.kind != NYX_HIP_ROT_IAU && r.kind != NYX_HIP_ROT_EULER_CHEBY) return "unknown orientation kind";
    if (r.n_nut_prec < 0 || r.n_nut_prec > NYX_HIP_MAX_NUT_PREC) return "n_nut_prec outside 0..NYX_HIP_MAX_NUT_PREC";
    if (r.kind == NYX_HIP_ROT_EULER_CHEBY && (r.euler_segment < 0 || r.euler_segment >= n_segments)) return "euler_segment is not one of config.segments";
    return nullptr;
}

static double ns_to_seconds_host(int64_t ns) {  // Duration::to_seconds for |ns| < 1 century, ns >= 0
    int64_t q = ns / 1000000000LL, r = ns % 1000000000LL;
    return (double)q + (double)r * 1e-9;
}

// config.tuning -> the context's copy.  The process environment is consulted ONLY when NYX_HIP_TUNING_ENV is set (the A/B
// tools of this repository: tools/*.py, tools/*.sh): a library behind a C-ABI takes its switches through its config struct.
struct ExpKnobs {  // experiment knobs of tools/sweep.py that have no field in nyx_hip_tuning_t (the helper dealing, round 5); < 0 / 0: unset
    double fast_weight = 0.0, start_rows = -1.0;
    int deal = -1;
    int place[8] = {-1, -1, -1, -1, -1, -1, -1, -1};  // role fan-out: the wave of the k-th duty (duties heaviest first), assign_roles
    int place_sums = -1, place_twobody = -1;         // ... and of the two offloaded integrator pieces
    bool fit_big = false, fit_quad = false, fit_solo = false;  // the free-order column placement for the large cooperative / the quad STM / the solo shape too
};
static nyx_hip_tuning_t resolve_tuning(const nyx_hip_tuning_t *t, ExpKnobs *xk = nullptr) {
    nyx_hip_tuning_t r = NYX_HIP_TUNING_DEFAULT;
    if (t) r = *t;
    if (!std::getenv("NYX_HIP_TUNING_ENV")) return r;
    if (xk) {
        if (const char *e = std::getenv("NYX_HIP_COOP_FASTW")) xk->fast_weight = std::atof(e);
        if (const char *e = std::getenv("NYX_HIP_COOP_START")) xk->start_rows = std::atof(e);
        if (const char *e = std::getenv("NYX_HIP_COOP_DEAL")) xk->deal = std::atoi(e);
        if (const char *e = std::getenv("NYX_HIP_ROLE_PLACE")) {
            const char *q = e;
            for (int k = 0; k < 8 && *q; ++k) { xk->place[k] = (int)std::strtol(q, (char **)&q, 10); if (*q == ',') ++q; }
        }
        if (const char *e = std::getenv("NYX_HIP_ROLE_OFFLOAD")) (void)std::sscanf(e, "%d,%d", &xk->place_sums, &xk->place_twobody);
        xk->fit_big = std::getenv("NYX_HIP_FIT_BIG") != nullptr;
        xk->fit_quad = std::getenv("NYX_HIP_FIT_QUAD") != nullptr;
        xk->fit_solo = std::getenv("NYX_HIP_FIT_SOLO") != nullptr;
    }
    auto geti = [](const char *name, int32_t &dst) { if (const char *e = std::getenv(name)) dst = (int32_t)std::strtol(e, nullptr, 0); };
    auto getd = [](const char *name, double &dst) { if (const char *e = std::getenv(name)) dst = std::atof(e); };
    int32_t cal = -1;
    geti("NYX_HIP_CALIBRATE", cal);
    if (cal == 1) r.schedule = NYX_HIP_SCHED_CALIBRATED;
    if (cal == 0) r.schedule = NYX_HIP_SCHED_MODEL;
    geti("NYX_HIP_DETERMINISTIC", r.deterministic);
    geti("NYX_HIP_COOP", r.cooperative);
    geti("NYX_HIP_PIPE", r.pipelined);
    geti("NYX_HIP_SPEC", r.chained_attempts);
    geti("NYX_HIP_ED_REUSE", r.epoch_data_reuse);
    geti("NYX_HIP_FANOUT", r.role_fanout);
    if (std::getenv("NYX_HIP_MERGE_ROLES")) r.merge_roles = 1;
    geti("NYX_HIP_STM_QUAD", r.stm_quad);
    geti("NYX_HIP_HARM_FEED", r.harmonics_feed);
    geti("NYX_HIP_COOP_COLS", r.coop_max_columns);
    if (std::getenv("NYX_HIP_COOP_MUTE")) r.coop_mute = 1;
    if (std::getenv("NYX_HIP_PROFILE")) r.profile = 1;
    geti("NYX_HIP_DEBUG", r.debug_flags);
    getd("NYX_HIP_COOP_FRAC", r.coop_fraction);
    getd("NYX_HIP_COOP_HELPERS", r.coop_helper_ratio);
    getd("NYX_HIP_COL_FIX", r.column_start_cost);
    if (const char *e = std::getenv("NYX_HIP_ROLE_HANDICAP")) (void)std::sscanf(e, "%lf,%lf,%lf", &r.role_duties[0], &r.role_duties[1], &r.role_duties[2]);
    if (const char *e = std::getenv("NYX_HIP_AGE_WEIGHTS"))
        if (std::sscanf(e, "%lf,%lf,%lf,%lf", &r.age_weights[0], &r.age_weights[1], &r.age_weights[2], &r.age_weights[3]) == 4) r.schedule = NYX_HIP_SCHED_EXPLICIT;
    if (const char *e = std::getenv("NYX_HIP_WAVE_WEIGHTS")) {
        const char *q = e;
        for (int w = 0; w < 16 && *q; ++w) { r.wave_weights[w] = std::strtod(q, (char **)&q); if (*q == ',') ++q; }
        r.schedule = NYX_HIP_SCHED_EXPLICIT;
    }
    return r;
}
static bool any_nonzero(const double *v, int n) { for (int k = 0; k < n; ++k) if (v[k] != 0.0) return true; return false; }

// GravityField::new (reference dynamics/gravity_field.rs:52-132) re-expressed as the per-column
// entry table the kernel streams (see HarmEntry in devcfg.h).
static void build_harmonics(const nyx_hip_gravity_field_t *g, std::vector<HarmEntry> &tab, std::vector<ColHdr> &cols,
                            std::vector<int32_t> &col_len, int &n_cols) {
    const int N = g->degree, M = std::min(g->order, g->degree);
    auto C = [&](int n, int m) -> double { return (n < 0 || m < 0 || n > N || m > n || m > M) ? 0.0 : g->c_nm[(size_t)n * (n + 1) / 2 + m]; };
    auto S = [&](int n, int m) -> double { return (n < 0 || m < 0 || n > N || m > n || m > M) ? 0.0 : g->s_nm[(size_t)n * (n + 1) / 2 + m]; };
    auto vr01 = [&](int n, int m) -> double {
        double nf = n, mf = m;
        double v = std::sqrt((nf - mf) * (nf + mf + 1.0));
        return m == 0 ? v / std::sqrt(2.0) : v;
    };
    auto vr11 = [&](int n, int m) -> double {
        double nf = n, mf = m;
        double v = std::sqrt(((2.0 * nf + 1.0) * (nf + mf + 2.0) * (nf + mf + 1.0)) / (2.0 * nf + 3.0));
        return m == 0 ? v / std::sqrt(2.0) : v;
    };
    auto bnm = [&](int n, int m) -> double {
        double nf = n, mf = m;
        return std::sqrt(((2.0 * nf + 1.0) * (2.0 * nf - 1.0)) / ((nf + mf) * (nf - mf)));
    };
    auto cnm = [&](int n, int m) -> double {
        double nf = n, mf = m;
        return std::sqrt(((2.0 * nf + 1.0) * (nf + mf - 1.0) * (nf - mf - 1.0)) / ((nf - mf) * (nf + mf) * (2.0 * nf - 3.0)));
    };
    // diagonal A[n][n]
    std::vector<double> diag(N + 3);
    diag[0] = 1.0;
    for (int n = 1; n <= N + 2; ++n) diag[n] = std::sqrt(1.0 + 1.0 / (2.0 * (double)n)) * diag[n - 1];
    // column c carries x/y terms of order m = c (c <= M) and z/w terms of order m = c - 1 (c - 1 <= M)
    n_cols = std::min(N + 1, M + 1);
    const double SQ2 = std::sqrt(2.0);
    ColHdr zero_hdr;
    std::memset(&zero_hdr, 0, sizeof zero_hdr);
    cols.assign(n_cols + 3, zero_hdr);  // spare tail entries: the kernel prefetches header c + 1
    col_len.assign(n_cols + 2, 0);
    tab.clear();
    for (int c = 1; c <= n_cols; ++c) {
        const int rows = N + 2 - c;
        const int nb = rows / HARM_BATCH, rem = rows % HARM_BATCH;  // full batches, then `rem` rows one at a time
        cols[c].start = (int32_t)tab.size();
        cols[c].nb = nb | (rem << 16);
        cols[c].rows = rows;
        cols[c].scale = (double)c * SQ2;
        cols[c].diag = diag[c];
        col_len[c] = rows;
        double B = 1.0;  // prod of b[k][c], k = c+1 .. n: the scale of the carried recursion variable (see HarmEntry)
        for (int n = c; n <= N + 1; ++n) {
            HarmEntry e;
            if (n == c) {
                e.g = -1.0;
            } else if (n == c + 1) {
                e.g = 0.0;
                B *= bnm(n, c);
            } else {
                e.g = cnm(n, c) / (bnm(n, c) * bnm(n - 1, c));
                B *= bnm(n, c);
            }
            e.t1 = B * C(n, c);
            e.t2 = B * S(n, c);
            // z: (n, m = c-1), n in 1..N
            const bool zok = (n >= 1 && n <= N);
            e.t3 = zok ? B * (SQ2 * vr01(n, c - 1) * C(n, c - 1)) : 0.0;
            e.t4 = zok ? B * (SQ2 * vr01(n, c - 1) * S(n, c - 1)) : 0.0;
            // w: (n-1, m = c-1), n-1 in 1..N
            const bool wok = (n - 1 >= 1 && n - 1 <= N && n - 1 >= c - 1);
            e.t5 = wok ? B * (SQ2 * vr11(n - 1, c - 1) * C(n - 1, c - 1)) : 0.0;
            e.t6 = wok ? B * (SQ2 * vr11(n - 1, c - 1) * S(n - 1, c - 1)) : 0.0;
            tab.push_back(e);
        }
    }
}

// The same table as ONE stream for the hybrid feed (devcfg.h, HYB_*): stream row r = entry r of `tab` (the columns' rows back to
// back).  Scalar side: {g, t1, t2}, 24 bytes per row; vector side, behind it: groups of sixteen rows, [t3..t6][16].
static void build_hybrid(const std::vector<HarmEntry> &tab, std::vector<double> &hyb, int64_t &vec_off) {
    // whole groups, plus two more: the walk fetches one batch / one group past its last row
    const size_t n = tab.size(), padded = (n + 15) / 16 * 16 + 2 * 16;
    const HarmEntry z = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    hyb.clear();
    for (size_t r = 0; r < padded; ++r) {
        const HarmEntry &e = r < n ? tab[r] : z;
        hyb.push_back(e.g); hyb.push_back(e.t1); hyb.push_back(e.t2);
    }
    vec_off = (int64_t)hyb.size();  // 3 * padded doubles = a multiple of 48: the vector side starts 128-byte aligned
    for (size_t g = 0; g < padded / 16; ++g)
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 16; ++l) {
                const size_t r = g * 16 + l;
                const HarmEntry &e = r < n ? tab[r] : z;
                hyb.push_back(j == 0 ? e.t3 : j == 1 ? e.t4 : j == 2 ? e.t5 : e.t6);
            }
}

// A run stream (DevCfg.rs_*): the rows of the schedules `ids`, every range of every wave laid out as one contiguous piece that
// starts a sixteen-row group of its own - a wave's walk then begins on its range's first row (in the common stream it begins at the
// batch that holds it, with up to seven rows of the previous column in front, through the recursion with a zero state: at 70x70
// 3 % of an owner's rows, and most of a short range).  `cols_x` = the column headers with `start` pointing into this stream.
// Same rows, same operations: bit-identical sums.
static void build_run_stream(const nyx_hip_ctx *ctx, std::initializer_list<int> ids, std::vector<double> &hyb, int64_t &vec_off, std::vector<ColHdr> &cols_x) {
    cols_x = ctx->h_cols;
    std::vector<HarmEntry> t;
    const HarmEntry z = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    std::vector<char> seen(cols_x.size(), 0);
    for (int k : ids) {
        const DevSched &sd = ctx->host_cfg.sched[k];
        for (int w = 0; w < DEV_MAX_WAVES; ++w)
            for (int r = 0; r < sd.n_ranges[w]; ++r) {
                t.resize((t.size() + 15) / 16 * 16, z);
                for (int c = sd.range_c0[w][r]; c < sd.range_c0[w][r] + sd.range_cnt[w][r]; ++c) {
                    if (c < 1 || c >= (int)cols_x.size() || seen[c]) continue;
                    seen[c] = 1;
                    const int32_t src = ctx->h_cols[c].start;
                    cols_x[c].start = (int32_t)t.size();
                    for (int q = 0; q < ctx->h_cols[c].rows; ++q) t.push_back(ctx->h_tab[(size_t)src + q]);
                }
            }
    }
    build_hybrid(t, hyb, vec_off);
}

// Column schedule: wave w walks at most two contiguous ranges — long columns from the low-c end,
// topped up with short columns from the high-c end — so that one complex power per range suffices.
// Waves 0/1/2 also carry the integrator / almanac / perturbation duties (`role_handicap`, in units of
// one harmonics term), so they receive a reduced share of the columns, possibly none.
// Water-filling of the columns [c_lo, c_hi] over `n_waves` waves with per-wave handicaps hc[] (work a wave does besides
// its columns, in harmonics-term units) and SIMD age weights.  Wave 0 takes what is left.
// `list`: the columns to distribute, ascending (= longest first).  Returns false if a wave would need more than
// DEV_MAX_RANGES contiguous ranges.
static bool fill_schedule(const nyx_hip_ctx *ctx, DevSched &sd, int n_waves, const std::vector<int> &list, const double *hc_model,
                          bool all_columns) {
    double hc[DEV_MAX_WAVES];
    for (int w = 0; w < DEV_MAX_WAVES; ++w) hc[w] = hc_model[w];
    for (int w = 0; w < DEV_MAX_WAVES; ++w) sd.n_ranges[w] = 0;
    if (list.empty()) return true;
    // cost of a column in rows: its length plus what it costs to START one (header, complex power of the range, a cold first batch).
    // With the short columns of a small field in the quad layout that start is most of a column: 21x21, 1 000 trajectories, sixty
    // segments: 11.45 ms with 0 rows, 11.1 with 4, 10.83 with 6, 10.95 with 8 (four runs each, +-0.03).  70x70 plain kernel: no effect
    // up to 6, slower beyond (the measured per-wave weights already carry it there).  Round 4, with the roles fanned out over eight of
    // the sixteen waves: 6 rows left the oldest pure column wave without a column (the two-ended fill ran out of columns before it
    // reached wave 4) and the youngest ones with the longest; config 4, three runs each: 10.10 ms with 6, 9.83 with 8, 9.55-9.60 with
    // 9 ... 18 (a plateau: every wave holds one or two columns then) - same bits, the quad layout's sums do not depend on the split.
    double col_fix = (ctx->sched_quad && n_waves == DEV_MAX_WAVES) ? 12.0 : 0.0;  // (the quad layout's production shape: sixteen waves)
    if (ctx->tune.column_start_cost >= 0.0) col_fix = ctx->tune.column_start_cost;
    auto cost = [&](int c) { return (double)ctx->col_len[c] + col_fix; };
    double terms = 0.0;
    for (int c : list) terms += cost(c);
    // Per-wave weights.  The four waves that share a SIMD (w, w+4, w+8, w+12) are arbitrated oldest-first, so with equal
    // shares the oldest finishes early and the youngest runs the tail alone, with nothing to hide its scalar-load latency;
    // role waves carry their duty besides.  The weights are MEASURED: calibrate() runs the workload's own first steps with
    // the in-kernel cycle accounting and moves columns from the late waves to the early ones until the windows agree;
    // before that (and with calibration off) a structural guess by age class is used.
    double per_wave[DEV_MAX_WAVES];
    bool fit = false;  // (the runs of a block schedule placed along the list in a free wave order: the cooperative 70x70 shape, see below)
    {
        const nyx_hip_ctx::WKey key(n_waves, (ctx->host_cfg.pipe && (!(ctx->host_cfg.flags & NYX_HIP_FLAG_STM) || ctx->sched_quad)) ? 1 : 0, ctx->sched_quad ? 1 : 0,
                                    all_columns ? -1 : (int)(ctx->host_cfg.coop_frac * 10.0 + 0.5));
        const auto it = ctx->weights.find(key);
        // The cost model of NYX_HIP_SCHED_MODEL: the speed of a wave is a property of its place in the workgroup (the four waves of a
        // SIMD are arbitrated oldest first; role waves and their SIMD-mates run differently) and of the workgroup's shape, not of
        // the force model.  Measured once with the calibration below on the BASELINE workloads (tools/dump_weights.py, two contexts
        // each, agreement ~2 %) and frozen here, so that the default schedule - hence the summation order, hence every bit of the
        // result - is the same in every context, process and rank.
        static const double model_coop[16] = {1.70, 1.66, 1.48, 2.14, 2.08, 1.70, 1.65, 1.67, 1.45, 0.97, 1.05, 1.02, 0.70, 0.53, 0.56, 0.55};
        static const double model_solo[16] = {1.41, 1.41, 1.26, 1.61, 1.59, 1.29, 1.375, 1.23, 1.06, 0.98, 0.98, 0.98, 0.77, 0.69, 0.70, 0.70};
        // (quad table, round 5: re-fitted by hill-climbing the explicit weights on config 4 with the position-only pieces of phase C on
        //  the DCM wave (assign_roles, DEV_ROLE_QPRE) - with the integrator's window shorter the column waves are the period again:
        //  8.99 ms with the round-4 table {0.70 x 4, 1.26, 2.03, 1.78, 1.59, 1.58, 1.68, 0.96, 1.02, 0.875, 0.93, 0.86, 0.91}, 8.61 with this)
        static const double model_quad[16] = {0.700, 0.700, 0.505, 0.876, 1.173, 1.490, 1.795, 2.380, 3.445, 2.528, 0.927, 1.020, 0.875, 0.930, 0.941, 1.124};
        // Round 4, the shapes that deal ONE contiguous run of columns per wave (below) and stream the table in the trajectory-owning
        // workgroups: fitted with tools/tune_schedule.py (windows of every wave -> rows that would equalise them -> weights, best
        // kernel time of 8-14 iterations, two boxes) on configs[1] at 10 000 (cooperative) and 16 384 trajectories (alone) and on
        // configs[4] (150x150, cooperative, helper jobs of several columns).  The role duties of the water-filling are unchanged.
        // (cooperative 70x70 table: fitted on the FULL day of configs[1] - the perturbation wave's duty grows over the day, 14 k -> 20 k cycles
        //  per evaluation once the lanes' eclipse transitions no longer coincide, and a table fitted on the first three hours overloaded it:
        //  719 -> 680 ms per 10 000 x 24 h, same box)
        static const double model_coop_blk[16] = {1.00, 1.413, 0.549, 1.946, 1.892, 1.523, 1.588, 1.292, 1.066, 0.828, 0.937, 0.692, 0.430, 0.347, 0.357, 0.142};
        static const double model_coop_big_blk[16] = {1.00, 1.755, 1.706, 1.802, 1.733, 1.22, 1.246, 1.181, 1.087, 0.672, 0.621, 0.604, 0.549, 0.279, 0.284, 0.255};
        static const double model_solo_blk[16] = {1.00, 1.612, 1.263, 1.906, 1.764, 1.346, 1.331, 1.198, 1.113, 0.757, 0.659, 0.568, 0.513, 0.398, 0.291, 0.27};
        // Round 5, the cooperative 70x70 shape with its runs placed in a free wave order (`fit`, below): the column waves' weights as
        // tools/tune_schedule.py settles on them with that partition (full day of configs[1]; the role waves keep the table's values -
        // every row more on them costs the integrator's chain: 614 ms with these, 662 with ten rows more on each of the two).
        // Same box, product kernel, 24 h: 625.0 ms linear partition, 617.0 free order with the old table, 614.0 with this one.
        static const double model_coop_fit[16] = {1.00, 1.413, 0.549, 1.95, 1.83, 1.48, 1.40, 1.23, 1.04, 0.88, 0.85, 0.62, 0.50, 0.36, 0.34, 0.16};
        const bool blk = ctx->block_schedule && n_waves == DEV_MAX_WAVES && !ctx->sched_quad && ((ctx->host_cfg.harm_feed & 1) || ctx->block_force);
        fit = (blk && (!all_columns || ctx->fit_solo) && (ctx->host_cfg.n_cols <= 96 || ctx->fit_big) && ctx->fit_partition) || (ctx->fit_quad && ctx->sched_quad && n_waves == DEV_MAX_WAVES);
        const double *model = ctx->sched_quad ? model_quad
                              : (blk ? (all_columns ? model_solo_blk : (ctx->host_cfg.n_cols > 96 ? model_coop_big_blk : (fit ? model_coop_fit : model_coop_blk)))
                                     : (all_columns ? model_solo : model_coop));
        for (int w = 0; w < DEV_MAX_WAVES; ++w)
            per_wave[w] = it != ctx->weights.end() ? it->second[w] : (n_waves == 16 ? model[w] : 1.0);
        if (it != ctx->weights.end())  // measured duties replace the model's (the integrator keeps its window free: hc_model[0])
            for (int w = 0; w < n_waves; ++w)
                if (hc_model[w] < 1e8) hc[w] = it->second[DEV_MAX_WAVES + w];
    }
    if (ctx->tune.schedule == NYX_HIP_SCHED_EXPLICIT) {  // explicit weights: per wave, or one per SIMD age class
        const bool per = any_nonzero(ctx->tune.wave_weights, 16), age_on = any_nonzero(ctx->tune.age_weights, 4);
        for (int w = 0; w < DEV_MAX_WAVES; ++w) {
            if (per) per_wave[w] = ctx->tune.wave_weights[w];
            else if (age_on && n_waves == 16) per_wave[w] = ctx->tune.age_weights[w / 4];
        }
    }
    auto wgt = [&](int w) { return per_wave[w]; };
    // water-filling: level such that sum_w max(0, level * weight_w - hc[w]) = terms
    double level = 0.0;
    {
        double hsum = 0.0;
        for (int w = 0; w < n_waves; ++w) hsum += std::min(hc[w], 1e6);
        double lo = 0.0, hi = 4.0 * (terms + hsum);
        for (int it = 0; it < 80; ++it) {
            level = 0.5 * (lo + hi);
            double sum = 0.0;
            for (int w = 0; w < n_waves; ++w) sum += std::max(0.0, level * wgt(w) - hc[w]);
            if (sum < terms) lo = level; else hi = level;
        }
    }
    if ((ctx->block_schedule && n_waves == DEV_MAX_WAVES && !ctx->sched_quad && ((ctx->host_cfg.harm_feed & 1) || ctx->block_force)) || (fit && ctx->sched_quad)) {
        // ONE contiguous run of columns per wave (the hybrid stream walks a run as one piece of the table: every range START costs it a
        // pipeline fill, the complex power of the range and up to seven rows in front of the run - with two or three ranges per wave
        // and evaluation that is a third of a 70x70 owner's work).  Linear partition of the list (longest columns first) at the
        // cumulative targets; the waves with the largest targets take the long columns, role waves the short ones at the end, where
        // the granularity is finest.  The integrator wave (target 0 in the pipelined loop) gets nothing.
        std::vector<int> order;
        for (int w = 0; w < n_waves; ++w) order.push_back(w);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::max(0.0, level * wgt(a) - hc[a]) > std::max(0.0, level * wgt(b) - hc[b]); });
        if (fit) {
            // Round 5, the same contiguous runs in a FREE wave order.  The linear partition above deals whole columns at the cumulative
            // targets in descending target order, so a wave's load is off by up to half a column - +-25 rows of ~230 for the waves
            // that hold the 50-row columns of a 70x70 owner, and a column wave is bound by its OWN issue rate (one VALU instruction per
            // ~9 cycles and wave, tools/_exp/exec_rate.hip): the two or three waves rounded UP set the workgroup's period.  A run of j
            // columns starting at length L sums to jL - j(j-1)/2: which sums exist depends on WHERE in the list a run sits, so the
            // waves are placed along the list in whatever order lets every one of them meet its target - a depth-first search over
            // (columns consumed, waves placed) for the smallest tolerance D with |load_w - target_w| <= D * weight_w for every wave
            // (the weight is the wave's speed: the same TIME error everywhere; col_partition.h).  A pure function of the configuration, like the rest.
            std::vector<int> act;
            for (int w : order) if (std::max(0.0, level * wgt(w) - hc[w]) > 0.0) act.push_back(w);
            std::vector<double> cst, tg, wg;
            for (int c : list) cst.push_back(cost(c));
            for (int w : act) { tg.push_back(std::max(0.0, level * wgt(w) - hc[w])); wg.push_back(wgt(w)); }
            std::vector<int> seq_w, seq_k;   // the placement found: wave index (into act) and its first column, in list order
            const bool found = nyx_place_runs(cst, tg, wg, seq_w, seq_k);
            if (found) {
                for (size_t q = 0; q < seq_w.size(); ++q) {
                    const int w = act[seq_w[q]];
                    const size_t k0 = (size_t)seq_k[q], k1 = q + 1 < seq_w.size() ? (size_t)seq_k[q + 1] : list.size();
                    int nr = 0;
                    for (size_t a = k0; a < k1;) {
                        size_t e = a + 1;
                        while (e < k1 && list[e] == list[e - 1] + 1) ++e;
                        if (nr >= DEV_MAX_RANGES) return false;
                        sd.range_c0[w][nr] = list[a]; sd.range_cnt[w][nr] = (int)(e - a); ++nr;
                        a = e;
                    }
                    sd.n_ranges[w] = nr;
                }
                return true;
            }
            // (no placement within the widest tolerance: the linear partition below)
        }
        double cum_t = 0.0, cum_r = 0.0;
        size_t k = 0;
        for (size_t q = 0; q < order.size(); ++q) {
            const int w = order[q];
            cum_t += std::max(0.0, level * wgt(w) - hc[w]);
            const size_t k0 = k;
            const bool last = q + 1 == order.size() || std::max(0.0, level * wgt(order[q + 1]) - hc[order[q + 1]]) <= 0.0;
            while (k < list.size() && (last || cum_r + 0.5 * cost(list[k]) <= cum_t)) { cum_r += cost(list[k]); ++k; }
            if (k > k0) {
                // (the list is ascending in column number but may have gaps - the helper's columns: split the run at every gap)
                int nr = 0;
                for (size_t a = k0; a < k;) {
                    size_t e = a + 1;
                    while (e < k && list[e] == list[e - 1] + 1) ++e;
                    if (nr >= DEV_MAX_RANGES) return false;
                    sd.range_c0[w][nr] = list[a]; sd.range_cnt[w][nr] = (int)(e - a); ++nr;
                    a = e;
                }
                sd.n_ranges[w] = nr;
            }
            if (last) break;
        }
        return true;
    }
    int lo = 0, hi = (int)list.size() - 1;  // indices into `list`
    // plain column workers first (highest wave index), role waves last so they take what is left
    // (round 6: what is left goes to the last wave that WALKS columns - the integrator wave of a pipelined workgroup walks none (hc = 1e9),
    //  and a list of two or three short columns, the owner's share in the fan-out mode of a field below degree 40, is all "left over":
    //  dealt to wave 0 it was never evaluated - 59 m after two hours, tests/test_gpu_rotation.py)
    const int last_w = (hc[0] >= 1e8 && n_waves > 1) ? 1 : 0;
    for (int w = n_waves - 1; w >= last_w; --w) {
        const double tgt = std::max(0.0, level * wgt(w) - hc[w]);
        std::vector<int> mine;
        if (w == last_w) {
            for (int k = lo; k <= hi; ++k) mine.push_back(list[k]);
            lo = hi + 1;
        } else {
            double load = 0.0;
            while (lo <= hi && load + 0.5 * cost(list[lo]) <= tgt) { load += cost(list[lo]); mine.push_back(list[lo++]); }
            std::vector<int> tail;
            while (lo <= hi && load + 0.5 * cost(list[hi]) <= tgt) { load += cost(list[hi]); tail.push_back(list[hi--]); }
            mine.insert(mine.end(), tail.rbegin(), tail.rend());
        }
        // contiguous runs of column numbers -> ranges
        int nr = 0;
        for (size_t k = 0; k < mine.size();) {
            size_t e = k + 1;
            while (e < mine.size() && mine[e] == mine[e - 1] + 1) ++e;
            if (nr >= DEV_MAX_RANGES) return false;
            sd.range_c0[w][nr] = mine[k]; sd.range_cnt[w][nr] = (int)(e - k); ++nr;
            k = e;
        }
        sd.n_ranges[w] = nr;
        if (w == last_w) break;
    }
    return true;
}

// Role fan-out (small ensembles: the STM quad layout, and dynamics without a gravity field): with few workgroups on the
// chip what counts is the latency of ONE force evaluation, and the almanac and perturbation duties are its longest serial
// pieces.  They are dealt over several waves - the DCM and the body slots (or the distinct ephemeris segments, fanout_almanac_units) over up to DEV_MAX_ALM almanac waves (longest
// first), point masses (+ tides) and SRP (+ drag) over two perturbation waves - each writing its own LDS rows, so the
// arithmetic and its order do not change.  Costs in harmonics-term units, as `role_handicap`.
// Units of the almanac duty: the DCM, then either the DISTINCT ephemeris segments of all chains (segment mode: Earth -> EMB sits on
// every chain of an Earth-centred run and is evaluated once; the readers sum the chains, ed_bp() in the kernel) when their vectors
// fit the body rows of the epoch data - 4 with a DCM, 7 without - or the body slots.
static int distinct_segments(const DevCfg &dc, int *useg_seg = nullptr) {
    int n = 0, list[DEV_MAX_SEG];
    for (int s = 0; s < dc.n_slots; ++s)
        for (int k = 0; k < dc.slot[s].n_chain; ++k) {
            bool seen = false;
            for (int q = 0; q < n; ++q) seen = seen || list[q] == dc.slot[s].seg[k];
            if (!seen && n < DEV_MAX_SEG) list[n++] = dc.slot[s].seg[k];
        }
    if (useg_seg) for (int q = 0; q < n; ++q) useg_seg[q] = list[q];
    return n;
}
static bool segment_units_fit(const DevCfg &dc) {
    const bool dcm = dc.has_grav || dc.has_drag || dc.has_tides;
    const int nu = distinct_segments(dc);
    return nu >= 2 && nu <= (dcm ? DEV_MAX_SLOTS : DEV_MAX_SLOTS + 3);
}
static int fanout_almanac_units(const DevCfg &dc, int *unit_mask, double *unit_cost) {
    int n = 0;
    if (dc.has_grav || dc.has_drag || dc.has_tides) { unit_mask[n] = DEV_ROLE_DCM; unit_cost[n] = 18.0; ++n; }
    if (segment_units_fit(dc)) {
        int us[DEV_MAX_SEG];
        const int nu = distinct_segments(dc, us);
        for (int u = 0; u < nu; ++u) { unit_mask[n] = 1 << u; unit_cost[n] = 3.0 + 0.75 * dc.seg[us[u]].n_coef; ++n; }
        return n;
    }
    for (int s = 0; s < dc.n_slots; ++s) { unit_mask[n] = 1 << s; unit_cost[n] = 12.0 * dc.slot[s].n_chain; ++n; }
    return n;
}
static bool want_fanout(const nyx_hip_ctx *ctx, bool quad) {
    if (ctx->tune.role_fanout >= 0) return ctx->tune.role_fanout != 0;
    return quad || !ctx->host_cfg.has_grav;
}
static int fanout_role_waves(const nyx_hip_ctx *ctx, int *n_alm_out = nullptr, int *n_pert_out = nullptr) {
    const DevCfg &dc = ctx->host_cfg;
    int um[10]; double uc[10];
    const int units = fanout_almanac_units(dc, um, uc);
    const int n_alm = std::min(DEV_MAX_ALM, std::max(units, 0));
    const int n_pert = ((dc.n_pm > 0 || dc.has_tides || dc.has_grav2) ? 1 : 0) + ((dc.has_srp || dc.has_drag) ? 1 : 0);
    if (n_alm_out) *n_alm_out = n_alm;
    if (n_pert_out) *n_pert_out = n_pert;
    return 1 + n_alm + n_pert;
}

// Deals the roles of an n_waves workgroup (DevCfg.role_*) and returns the serial duty of every wave in `hc`.
static void assign_roles(nyx_hip_ctx *ctx, int n_waves, bool fanout, double *hc) {
    DevCfg &dc = ctx->host_cfg;
    const int all_alm = DEV_ROLE_DCM | ((1 << dc.n_slots) - 1);
    const int all_pert = (DEV_PERT_PM | DEV_PERT_SRP) << 16;
    for (int w = 0; w < DEV_MAX_WAVES; ++w) { dc.role_kind[w] = DEV_ROLE_COLUMNS; dc.role_mask[w] = 0; dc.role_slot[w] = 0; hc[w] = 0.0; }
    dc.n_alm = 1;
    dc.seg_mode = 0;
    dc.offload = 0;
    dc.qpre_off = 0;
    const double *rh = ctx->role_handicap;
    if (n_waves == 1) { dc.role_kind[0] = DEV_ROLE_ALL; dc.role_mask[0] = all_alm | all_pert; hc[0] = rh[0] + rh[1] + rh[2]; return; }
    dc.role_kind[0] = DEV_ROLE_INTEG; hc[0] = rh[0];
    if (n_waves == 2 || dc.merge_roles) { dc.role_kind[1] = DEV_ROLE_ALMANAC_PERT; dc.role_mask[1] = all_alm | all_pert; hc[1] = rh[1] + rh[2]; return; }
    int n_alm = 1, n_pert = 1;
    if (fanout && fanout_role_waves(ctx, &n_alm, &n_pert) <= n_waves && n_alm >= 1 && n_pert >= 1) {
        const bool stm = (dc.flags & NYX_HIP_FLAG_STM) != 0;
        // the duties: almanac shares (longest unit first onto the least loaded share), then the perturbation shares
        struct Duty { int kind, mask, slot; double cost; };
        std::vector<Duty> duties;
        {
            int um[10]; double uc[10];
            const int units = fanout_almanac_units(dc, um, uc);
            int order[10];
            for (int k = 0; k < units; ++k) order[k] = k;
            std::sort(order, order + units, [&](int a, int b) { return uc[a] > uc[b]; });
            double load[DEV_MAX_ALM] = {0.0};
            int amask[DEV_MAX_ALM] = {0};
            for (int k = 0; k < units; ++k) {
                int best = 0;
                for (int a = 1; a < n_alm; ++a) if (load[a] < load[best]) best = a;
                amask[best] |= um[order[k]];
                load[best] += uc[order[k]];
            }
            for (int a = 0; a < n_alm; ++a) duties.push_back({DEV_ROLE_ALMANAC, amask[a], a, load[a]});
            dc.n_alm = n_alm;
        }
        // (measured on the device, in units of ~250 cycles: a plain point mass 4, a dual one 11; SRP with its occultation 12 + 8 per
        //  shadow body, dual 25 + 25; drag 10; tides 14 + 8 per perturber)
        const double pm_cost = (stm ? 11.0 : 4.0) * dc.n_pm + (dc.has_tides ? (stm ? 3.0 : 1.0) * (14.0 + 8.0 * dc.t_n) : 0.0) +
                               (dc.has_grav2 ? 8.0 + 0.2 * ctx->terms2 : 0.0);
        const double srp_cost = (dc.has_srp ? (stm ? 25.0 + 25.0 * dc.n_shadow : 12.0 + 8.0 * dc.n_shadow) : 0.0) + (dc.has_drag ? 10.0 : 0.0);
        if (n_pert == 2) {
            duties.push_back({DEV_ROLE_PERT, DEV_PERT_PM << 16, 0, pm_cost});
            duties.push_back({DEV_ROLE_PERT, DEV_PERT_SRP << 16, 0, srp_cost});
        } else {
            duties.push_back({DEV_ROLE_PERT, all_pert, 0, pm_cost + srp_cost});
        }
        // Placement: wave w runs on SIMD w % 4, and a force evaluation is bound by the busiest SIMD's role work (the role code
        // is VALU-heavy: sincos, Chebyshev chains, divisions).  Heaviest duty first onto the least loaded SIMD; the integrator
        // (wave 0, ~50 units with its phases A and C) sits on SIMD 0.
        double simd_load[4] = {stm ? 52.0 : 26.0, 0.0, 0.0, 0.0};
        bool taken[DEV_MAX_WAVES] = {true};
        std::sort(duties.begin(), duties.end(), [](const Duty &a, const Duty &b) { return a.cost > b.cost; });
        bool placed_all = true;
        int duty_no = 0;
        for (const Duty &d : duties) {
            int best_w = -1;
            double best_load = 1e300;
            for (int sd = 0; sd < 4; ++sd) {
                int w = -1;
                for (int k = sd; k < n_waves; k += 4) if (!taken[k]) { w = k; break; }
                if (w >= 0 && simd_load[sd] < best_load) { best_load = simd_load[sd]; best_w = w; }
            }
            if (duty_no < 8 && ctx->role_place[duty_no] > 0 && ctx->role_place[duty_no] < n_waves && !taken[ctx->role_place[duty_no]]) best_w = ctx->role_place[duty_no];  // (tools: NYX_HIP_ROLE_PLACE)
            ++duty_no;
            if (best_w < 0) { placed_all = false; break; }
            taken[best_w] = true;
            simd_load[best_w % 4] += d.cost;
            dc.role_kind[best_w] = d.kind; dc.role_mask[best_w] = d.mask; dc.role_slot[best_w] = d.slot; hc[best_w] = d.cost;
        }
        if (placed_all) {
            if (dc.pipe && !dc.has_grav && !stm && (ctx->tune.debug_flags & 0x800)) {  // (0x800: A/B switch, same results)
                // pipelined, no column waves: the two lightest almanac shares take the two-body term and the head of the stage sums off
                // the integrator wave (DevCfg.offload).  Rounds 3-4 default; OFF since round 5: the integrator's publish-first window
                // (role_loop, `fastp`) is shorter than the offloaded one and the almanac SIMDs are the busiest of the workgroup
                // (config 3: 46.7 -> 44.7 ms without it, same bits)
                // (what an almanac wave has to spare depends on whom it shares its SIMD with: the integrator's SIMD last, then by the SIMD's load)
                int w1 = -1, w2 = -1;
                auto spare = [&](int w) { return (w % 4 == 0 ? 1e6 : 0.0) + simd_load[w % 4] + hc[w]; };
                for (int w = 1; w < n_waves; ++w) {
                    if (dc.role_kind[w] != DEV_ROLE_ALMANAC) continue;
                    if (w1 < 0 || spare(w) < spare(w1)) { w2 = w1; w1 = w; }
                    else if (w2 < 0 || spare(w) < spare(w2)) w2 = w;
                }
                if (ctx->role_place_sums > 0 && ctx->role_place_sums < n_waves && dc.role_kind[ctx->role_place_sums] == DEV_ROLE_ALMANAC) w1 = ctx->role_place_sums;  // (tools: NYX_HIP_ROLE_OFFLOAD)
                if (ctx->role_place_twobody > 0 && ctx->role_place_twobody < n_waves && dc.role_kind[ctx->role_place_twobody] == DEV_ROLE_ALMANAC) w2 = ctx->role_place_twobody;
                if (w1 >= 0) {
                    if (w2 < 0) w2 = w1;
                    dc.role_mask[w1] |= DEV_ROLE_SUMS; hc[w1] += 4.0;
                    dc.role_mask[w2] |= DEV_ROLE_TWOBODY; hc[w2] += 2.0;
                    dc.offload = 1;
                }
            }
            if (stm && ctx->sched_quad && !(ctx->tune.debug_flags & 0x4000000)) {  // (0x4000000: A/B switch, same results)
                // quad STM layout: the position-only pieces of phase C (quad_pre, 4-5 k cycles of the integrator's window per evaluation)
                // go to the almanac wave with the most time to spare - the integrator's chain is what bounds such a workgroup
                // (the wave that holds the DCM when there is one: in the sixteen-wave shape it walks no columns, the segment waves do)
                int wq = -1;
                for (int w = 1; w < n_waves; ++w)
                    if (dc.role_kind[w] == DEV_ROLE_ALMANAC && (dc.role_mask[w] & DEV_ROLE_DCM)) wq = w;
                if (wq < 0)
                    for (int w = 1; w < n_waves; ++w) {
                        if (dc.role_kind[w] != DEV_ROLE_ALMANAC) continue;
                        if (wq < 0 || simd_load[w % 4] + hc[w] < simd_load[wq % 4] + hc[wq]) wq = w;
                    }
                if (wq >= 0) { dc.role_mask[wq] |= DEV_ROLE_QPRE; hc[wq] += 18.0; simd_load[wq % 4] += 18.0; dc.qpre_off = 1; }
            }
            if (segment_units_fit(dc)) {  // the almanac shares above are distinct segments: tell the kernel where their vectors live
                dc.seg_mode = 1;
                dc.n_useg = distinct_segments(dc, dc.useg_seg);
                dc.ed_seg_base = (dc.has_grav || dc.has_drag || dc.has_tides) ? 9 : 0;
                for (int sl = 0; sl < dc.n_slots; ++sl)
                    for (int k = 0; k < dc.slot[sl].n_chain; ++k)
                        for (int u = 0; u < dc.n_useg; ++u)
                            if (dc.useg_seg[u] == dc.slot[sl].seg[k]) dc.slot[sl].useg[k] = u;
            }
            return;
        }
        for (int w = 1; w < DEV_MAX_WAVES; ++w) { dc.role_kind[w] = DEV_ROLE_COLUMNS; dc.role_mask[w] = 0; dc.role_slot[w] = 0; hc[w] = 0.0; }
        dc.n_alm = 1;
    }
    dc.role_kind[1] = DEV_ROLE_ALMANAC; dc.role_mask[1] = all_alm; hc[1] = rh[1];
    dc.role_kind[2] = DEV_ROLE_PERT; dc.role_mask[2] = all_pert; hc[2] = rh[2];
    // ONE almanac wave (the sixteen-wave column shapes): distinct-segment units pay here too - Earth -> EMB sits on the chain of
    // every body of an Earth-centred run and was evaluated once per BODY per stage (five Chebyshev evaluations for Sun + Moon where
    // four segments are distinct).  The wave evaluates every distinct segment once, the readers sum the chains (ed_body(): the same
    // additions in the same order, bit-identical).  (0x1000: A/B switch, same results)
    int chain_evals = 0;
    for (int sl = 0; sl < dc.n_slots; ++sl) chain_evals += dc.slot[sl].n_chain;
    if (segment_units_fit(dc) && distinct_segments(dc) < chain_evals && !(ctx->tune.debug_flags & 0x1000)) {
        dc.seg_mode = 1;
        dc.n_useg = distinct_segments(dc, dc.useg_seg);
        dc.ed_seg_base = (dc.has_grav || dc.has_drag || dc.has_tides) ? 9 : 0;
        for (int sl = 0; sl < dc.n_slots; ++sl)
            for (int k = 0; k < dc.slot[sl].n_chain; ++k)
                for (int u = 0; u < dc.n_useg; ++u)
                    if (dc.useg_seg[u] == dc.slot[sl].seg[k]) dc.slot[sl].useg[k] = u;
        dc.role_mask[1] = DEV_ROLE_DCM | ((1 << dc.n_useg) - 1);
    }
}

static void build_schedule(nyx_hip_ctx *ctx, int n_waves, bool quad = false) {
    DevCfg &dc = ctx->host_cfg;
    ctx->rs_dirty = true;  // (the run streams follow the schedules: rebuilt before the next launch that streams the table)
    const int nc = dc.n_cols;
    for (int k = 0; k < DEV_N_SCHED; ++k)
        for (int w = 0; w < DEV_MAX_WAVES; ++w) dc.sched[k].n_ranges[w] = 0;
    dc.n_waves = n_waves;
    if (dc.has_grav2)  // the second field: every column, for whichever wave walks it (the perturbation wave with the point-mass share)
        for (int w = 0; w < DEV_MAX_WAVES; ++w) {
            dc.sched[DEV_SCHED_SECOND].n_ranges[w] = 1;
            dc.sched[DEV_SCHED_SECOND].range_c0[w][0] = 1;
            dc.sched[DEV_SCHED_SECOND].range_cnt[w][0] = dc.n_cols2;
        }
    dc.merge_roles = (ctx->tune.merge_roles && n_waves >= 8) ? 1 : 0;
    // pipelined stage loop: sixteen-wave workgroups (the column waves go from one stage's harmonics into the next's), and - plain
    // kernel - any workgroup of dynamics without a gravity field that has the integrator in a wave of its own: the perturbation
    // waves need the POSITION of the next stage only, which the integrator publishes inside the window, so its phases A and C run
    // beside the almanac / perturbation duties instead of in front of them
    const bool stm_cfg = (dc.flags & NYX_HIP_FLAG_STM) != 0;
    // (not with a non-central gravity field: its inputs need the body's position of the stage, which the almanac waves write late in the window)
    dc.pipe = (!dc.merge_roles && ctx->tune.pipelined != 0 && !(dc.has_grav && dc.g_slot >= 0) &&
               ((n_waves == DEV_MAX_WAVES && dc.has_grav) || (!dc.has_grav && !stm_cfg && n_waves >= 2 && n_waves <= 8))) ? 1 : 0;
    // (a workgroup of more than eight waves WITHOUT a gravity field exists only when the caller forces it - nyx_hip_ctx_set_column_waves -
    //  and runs the plain loop: the pipelined integrator of the sixteen-wave kernels is compiled for the gravity-field shape, INTEG_OOL)
    // roles of this workgroup shape and their serial duties (merged roles when there are fewer than three waves)
    double hc[DEV_MAX_WAVES] = {0};
    assign_roles(ctx, n_waves, want_fanout(ctx, quad), hc);
    // speculative stage 0 (role_loop): the pipelined plain kernel with ONE almanac wave and an even stage count (the last window
    // then leaves the buffers of stage parity 0 free for the epoch data of t + h)
    // (with a gravity field: one almanac wave; without: any fan-out, almanac and perturbation duties in waves of their own)
    dc.spec = (dc.pipe && !(dc.flags & NYX_HIP_FLAG_STM) && dc.stages % 2 == 0 &&
               (dc.has_grav ? (dc.n_alm == 1 || (ctx->tune.debug_flags & 0x10000000) != 0) : (n_waves >= 3 && dc.role_kind[1] != DEV_ROLE_ALMANAC_PERT && (dc.n_slots > 0 || dc.has_drag || dc.has_tides))) &&
               !dc.has_grav2 &&  // (the second field's wave reads the attempt's epoch at stage 0: it would have to wait for step control)
               ctx->tune.chained_attempts != 0) ? 1 : 0;
    dc.ed_reuse = (dc.spec || dc.seg_mode) ? 0 : ctx->ed_reuse_fit;  // (chained attempts need no copy of the stage-0 epoch data: a rejected lane keeps its k_0)
    if (!dc.has_grav || nc == 0) return;
    // with enough column workers the integrator keeps its window free: its serial phases A / C gate every other wave
    // (pipelined loop: the integrator wave walks NO columns at all - role_loop skips its walk -, whatever duties the caller states:
    //  round 5 found tuning.role_duties handing it 57 rows that nobody then evaluated)
    if (n_waves >= 8 && (dc.pipe || !any_nonzero(ctx->tune.role_duties, 3))) hc[0] = 1e9;
    std::vector<int> all;
    for (int c = 1; c <= nc; ++c) all.push_back(c);
    (void)fill_schedule(ctx, dc.sched[DEV_SCHED_SOLO], n_waves, all, hc, true);
    // Cooperative mode (16-wave workgroups only).  The helper takes the LONGEST columns, at most one per column wave: its job
    // time is then one long column (~18 batches), which is within 17 % of the ideal x * terms / 16 for x <= 0.35, and the
    // owner keeps the many short columns that let it balance its fifteen waves.  (Interleaving the two sets column by
    // column was measured 10-25 % slower: the helper's waves then hold a long AND a short column each.)
    dc.sums_wave1 = 0;
    if (n_waves == DEV_MAX_WAVES && nc >= 8 && ctx->coop_fan) {
        // FAN-OUT mode (launch(): the idle CUs outnumber the owners at least two to one - a shard of an ensemble, a small Monte Carlo).
        // Every owner has K = coop_parts dedicated helper workgroups (propagate_kernel.hip, helper_body under NYX_COOP_FAN); the owner's
        // period is then bounded by its integrator's chain, not by column work, so the helpers take everything but the shortest columns:
        // the K * cpp longest, dealt round-robin over the parts (every part a mix of long and short: equal jobs), one column per wave,
        // the waves of a part taken round-robin over the SIMDs (eight columns = two waves per SIMD, which finish in ~10 k cycles where
        // four per SIMD need ~17 k).  The owner keeps at least two columns (its PRIMARY schedule must not be empty).
        const int K = std::min(std::max(ctx->coop_parts, 2), DEV_FAN_MAX);
        const int col_waves = DEV_MAX_WAVES - 2;
        int cpp = std::min(col_waves, (nc - 2 + K - 1) / K);
        if (ctx->tune.coop_max_columns > 0) cpp = std::max(1, std::min(cpp, (int)ctx->tune.coop_max_columns / K));
        const int n_help = std::min(nc - 2, K * cpp);
        std::vector<int> own;
        for (int c = n_help + 1; c <= nc; ++c) own.push_back(c);
        for (int k = 0; k < n_help; ++k) {
            const int part = k % K, pos = k / K;       // (ascending column number = descending length)
            const int w = 1 + pos;                      // waves 1 .. 14 sit on SIMDs 1 2 3 0 1 2 3 0 ...: any prefix is balanced
            DevSched &hs = dc.sched[DEV_SCHED_FAN0 + part];
            const int r = hs.n_ranges[w]++;
            hs.range_c0[w][r] = 1 + k; hs.range_cnt[w][r] = 1;
        }
        if (n_help > 0 && !own.empty() && fill_schedule(ctx, dc.sched[DEV_SCHED_PRIMARY], n_waves, own, hc, false)) {
            dc.coop_ok = 1;
            // The sums wave (round 6, fan_sums): the owner's period is its integrator's chain, of which the two stage sums of a window are
            // ~40 %; a column wave that holds no column of the PRIMARY schedule forms them beside it - on a SIMD that hosts no role wave
            // when there is one (waves 3, 7, 11, 15).  debug_flags 0x40000000: the integrator forms them itself (A/B, same bits).
#if NYX_FAN_SUMS
            if (dc.pipe && !dc.has_drag && !(ctx->tune.debug_flags & 0x40000000)) {  // (the six values live in the drag rows of the perturbation buffers)
                static const int order[] = {15, 11, 7, 3, 14, 13, 12, 10, 9, 8, 6, 5, 4};
                for (int w : order)
                    if (dc.role_kind[w] == DEV_ROLE_COLUMNS && dc.sched[DEV_SCHED_PRIMARY].n_ranges[w] == 0) { dc.sums_wave1 = w + 1; break; }
            }
#endif
        } else {
            dc.coop_ok = 0;
            for (int k = 0; k < DEV_N_SCHED; ++k)
                if (k == DEV_SCHED_PRIMARY || k >= DEV_SCHED_FAN0)
                    for (int w = 0; w < DEV_MAX_WAVES; ++w) dc.sched[k].n_ranges[w] = 0;
        }
    } else
    if (n_waves == DEV_MAX_WAVES && nc >= 8) {
        double terms = 0.0, given = 0.0;
        for (int c = 1; c <= nc; ++c) terms += ctx->col_len[c];
        std::vector<int> own, help;
        const int col_waves = DEV_MAX_WAVES - 2;  // a helper's wave 0 claims jobs, its last wave answers
        // One column per helper wave is the rule for short evaluation periods (70x70: a second column makes the job longer than the
        // owner can wait, 302 ms against 182 ms).  A large field turns that around: at 150x150 the owner's period is 140 k cycles,
        // a job of one 150-row column 34 k + the hand-off, and fourteen columns are 18 % of the terms where the helpers could take
        // half - so when the one-column rule leaves the helpers below HALF of their share, their waves take up to
        // DEV_MAX_RANGES columns each (config 5: 12.05 s with 14 columns, 11.36 s with 21, 10.31 s with 28).
        int max_cols = col_waves;
        const int parts_cfg = ctx->coop_parts == 2 ? 2 : 1;
        double share = dc.coop_frac;
        {
            double first = 0.0;
            for (int c = 1; c <= std::min(nc, col_waves); ++c) first += ctx->col_len[c];
            if (first < 0.5 * dc.coop_frac * terms) {
                max_cols = DEV_MAX_RANGES * col_waves;
                share = 0.92 * dc.coop_frac;  // (several columns per wave: a job is longer for the same share; 35 / 38 / 42 columns at 150x150: 9.24 / 9.04 / 9.72 s)
            }
        }
        if (parts_cfg == 2 && max_cols > col_waves) max_cols = 2 * DEV_MAX_RANGES * col_waves;  // (each part has its own DEV_MAX_RANGES per wave)
        if (ctx->tune.coop_max_columns > 0) max_cols = std::min(parts_cfg * DEV_MAX_RANGES * col_waves, (int)ctx->tune.coop_max_columns);
        // Balanced dealing (round 5; one-part hand-off of a field whose helper jobs hold ONE long column per wave, i.e. 70x70):
        // the column waves of a helper are not alike - the two SIMDs that host the producer and the answering wave run three of them,
        // the other two four - and with the streamed table a helper is bound by its SIMDs' issue, so a wave of a three-wave SIMD walks
        // 4/3 the rows of the others in the same time.  The longest columns still go one per wave; when the share asks for more than
        // those, the FAST waves get a second, medium column each out of one contiguous block of the table (the owners keep contiguous
        // runs on either side), chosen so that every SIMD of the helper finishes together.  (debug_flags 0x400000: the old dealing.)
        const bool balanced = ctx->coop_deal != 0 && parts_cfg == 1 && max_cols == col_waves && nc > 3 * col_waves;
        std::vector<int> topup;
        if (balanced) {
            double first = 0.0;
            for (int c = 1; c <= col_waves; ++c) first += ctx->col_len[c];
            const double extra = share * terms - first;
            const int n_fast = 6;
            const double per = extra / n_fast - ctx->coop_start_rows;  // rows of the second column of a fast wave
            if (per >= 6.0) {
                // columns of `per` rows: col_len[c] = deg + 2 - c
                int c_mid = dc.deg + 2 - (int)(per + 0.5);
                int c_lo = c_mid - n_fast / 2, c_hi = c_lo + n_fast - 1;
                if (c_lo <= col_waves) { c_lo = col_waves + 1; c_hi = c_lo + n_fast - 1; }
                if (c_hi > nc - 2) { c_hi = nc - 2; c_lo = c_hi - n_fast + 1; }
                if (c_lo > col_waves)
                    for (int c = c_lo; c <= c_hi; ++c) topup.push_back(c);
            }
        }
        int n_long = 0;  // columns taken from the head of the table (the longest)
        for (int c = 1; c <= nc; ++c) {
            const bool is_top = std::find(topup.begin(), topup.end(), c) != topup.end();
            // (with a second column on the fast waves the long block is the full first round: one column per wave)
            const bool long_ok = n_long < max_cols && c < nc - 1 && (!topup.empty() || given + 0.5 * ctx->col_len[c] <= share * terms) && (topup.empty() || c <= col_waves);
            if (is_top || long_ok) {
                help.push_back(c);
                given += ctx->col_len[c];
                if (!is_top) ++n_long;
            } else {
                own.push_back(c);
            }
        }
        // helper: one column per wave, longest first; the two SIMDs that also host the producer and the answering wave have
        // three column waves (4 8 12 / 3 7 11) and take the six longest, the other two SIMDs four each
        static const int wave_order[DEV_MAX_WAVES - 2] = {4, 3, 8, 7, 12, 11, 1, 2, 5, 6, 9, 10, 13, 14};
        // Two-part hand-off (ctx->coop_parts == 2, chosen by launch() when the idle CUs outnumber the owners and the helpers' jobs hold
        // several columns per wave): the helpers' columns are dealt alternately into two sub-jobs that two DIFFERENT helper workgroups
        // claim - half the job per helper, so the turnaround the owner waits for halves and twice the helpers find work.
        const int parts = ctx->coop_parts == 2 ? 2 : 1;
        for (int part = 0; part < 2; ++part) {
            DevSched &hs = dc.sched[part ? DEV_SCHED_HELPER2 : DEV_SCHED_HELPER];
            for (int w = 0; w < DEV_MAX_WAVES; ++w) hs.n_ranges[w] = 0;
            if (part >= parts) continue;
            std::vector<int> mine;
            for (size_t k = 0; k < help.size(); ++k) if ((int)(k % (size_t)parts) == part) mine.push_back(help[k]);
            if (ctx->coop_deal >= 2 && parts_cfg == 1 && max_cols == col_waves) {
                // EXPERIMENT (tools only: NYX_HIP_COOP_DEAL=2 / 3), measured in round 5 and LOST: two waves per SIMD, each with a pair of
                // columns of equal sum (k-th longest + k-th shortest of the helper's set) - on the microbenchmark two waves saturate a
                // SIMD's issue, so four one-column waves that finish at 8 / 10 / 13 / 16-17 k cycles looked like oldest-first skew a pair
                // of two-column waves would avoid.  They do not: a column walk is a dependent chain per row (108 cycles per row and wave
                // whatever the SIMD's load), 129 rows on one wave are 15.6 k cycles - 95.4 ms per 3 h of configs[1] against 77.4 (scalar
                // feed), 102.3 against 77.0 (streamed).  One column per wave on fourteen waves stays.
                static const int pair_waves[12] = {1, 2, 4, 3, 5, 6, 8, 7, 9, 10, 12, 11};
                const int nm = (int)mine.size();
                const int use = ctx->coop_deal == 3 ? 12 : 8;   // (3: three waves per SIMD, for comparison)
                int q = 0;
                for (int a = 0, b = nm - 1; a <= b; ++a, --b, ++q) {
                    const int w = pair_waves[q % use];
                    if (hs.n_ranges[w] + 2 > DEV_MAX_RANGES) break;
                    int r = hs.n_ranges[w]++;
                    hs.range_c0[w][r] = mine[a]; hs.range_cnt[w][r] = 1;
                    if (b > a) { r = hs.n_ranges[w]++; hs.range_c0[w][r] = mine[b]; hs.range_cnt[w][r] = 1; }
                }
                continue;
            }
            if (balanced) {
                // longest column first onto the wave that would finish it soonest: load / speed, speed = coop_fast_weight on the SIMDs with
                // three column waves (waves 4 8 12 beside the producer, 3 7 11 beside the answering wave)
                double load[DEV_MAX_WAVES] = {0.0};
                for (int c : mine) {  // (ascending column number = descending length)
                    int best = -1;
                    double best_t = 1e300;
                    for (int q = 0; q < col_waves; ++q) {
                        const int w = wave_order[q];
                        if (hs.n_ranges[w] >= DEV_MAX_RANGES) continue;
                        const double speed = (w % 4 == 0 || w % 4 == 3) ? ctx->coop_fast_weight : 1.0;
                        const double t = (load[w] + ctx->col_len[c] + (hs.n_ranges[w] > 0 ? ctx->coop_start_rows : 0.0)) / speed;
                        if (t < best_t - 1e-9) { best_t = t; best = w; }
                    }
                    if (best < 0) break;
                    load[best] += ctx->col_len[c] + (hs.n_ranges[best] > 0 ? ctx->coop_start_rows : 0.0);
                    const int r = hs.n_ranges[best]++;
                    hs.range_c0[best][r] = c; hs.range_cnt[best][r] = 1;
                }
                continue;
            }
            for (size_t k = 0; k < mine.size(); ++k) {
                // further rounds are dealt in alternating directions: every wave's set has about the same length
                const int round = (int)k / col_waves, pos = (int)k % col_waves;
                const int w = (round & 1) ? wave_order[col_waves - 1 - pos] : wave_order[pos];
                const int r = hs.n_ranges[w]++;
                hs.range_c0[w][r] = mine[k]; hs.range_cnt[w][r] = 1;
            }
        }
        if (!help.empty() && !own.empty() && fill_schedule(ctx, dc.sched[DEV_SCHED_PRIMARY], n_waves, own, hc, false)) {
            dc.coop_ok = 1;
        } else {
            dc.coop_ok = 0;
            for (int w = 0; w < DEV_MAX_WAVES; ++w) dc.sched[DEV_SCHED_PRIMARY].n_ranges[w] = dc.sched[DEV_SCHED_HELPER].n_ranges[w] = dc.sched[DEV_SCHED_HELPER2].n_ranges[w] = 0;
        }
    } else {
        dc.coop_ok = 0;
    }
}

// STM layout by ensemble size.  The quad layout spends 4 lanes per trajectory (1.6x the f64 issue slots of the D3 layout
// per trajectory) to get 4x the workgroups and 4x the waves per workgroup: it wins while the D3 layout would leave most of
// the chip without a workgroup, i.e. up to ~2 quad workgroups per CU.
static bool pick_quad(const nyx_hip_ctx *ctx, int64_t n) {
    if (!(ctx->host_cfg.flags & NYX_HIP_FLAG_STM)) return false;
    if (ctx->host_cfg.flags & NYX_HIP_FLAG_STM_TEXTBOOK) return false;  // (the variational equations are integrated by the 64-lane layout: one trajectory's k-buffer column per lane)
    if (ctx->forced_quad >= 0) return ctx->forced_quad != 0;
    if (ctx->tune.stm_quad >= 0) return ctx->tune.stm_quad != 0;
    // deterministic: the layout fixes the column split, hence the bits - it must not follow the batch size (a shard is a smaller batch)
    if (ctx->tune.deterministic) return true;
    const int64_t cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
    return (n + 15) / 16 <= 2 * cus;
}

static int pick_waves(const nyx_hip_ctx *ctx, int64_t n) {
    const bool stm = (ctx->host_cfg.flags & NYX_HIP_FLAG_STM) != 0;
    if (stm && pick_quad(ctx, n)) {  // quad layout: 128 VGPRs per wave like the plain kernel
        if (ctx->forced_waves > 0) return std::min(ctx->forced_waves, DEV_MAX_WAVES);
        if (!ctx->host_cfg.has_grav) return want_fanout(ctx, true) ? (fanout_role_waves(ctx) > 4 ? 8 : std::max(3, fanout_role_waves(ctx))) : 3;
        return ctx->host_cfg.deg < 8 ? 8 : 16;
    }
    if (stm) {  // dual-number variant: 256 VGPRs per wave, at most DEV_MAX_WAVES_STM waves
        if (ctx->forced_waves > 0) return std::min(ctx->forced_waves, DEV_MAX_WAVES_STM);
        return ctx->host_cfg.has_grav ? DEV_MAX_WAVES_STM : 3;
    }
    if (ctx->forced_waves > 0) return std::min(ctx->forced_waves, DEV_MAX_WAVES);
    // no harmonics: integrator + almanac + perturbation waves form a 3-stage pipeline
    if (!ctx->host_cfg.has_grav) {
        if (!(ctx->host_cfg.n_slots > 0 || ctx->host_cfg.has_drag || ctx->host_cfg.has_tides)) return 1;
        return want_fanout(ctx, false) ? (fanout_role_waves(ctx) > 4 ? 8 : std::max(3, fanout_role_waves(ctx))) : 3;  // (8: two role waves per SIMD can be placed)
    }
    // Sixteen waves whatever the ensemble size: a workgroup's LDS (~150 KB) gives it a CU to itself, so the column split is what puts
    // four waves on every SIMD.  (Rounds 1-3 went down to eight and four waves for >= 32 705 / >= 131 009 trajectories, sized when a
    // workgroup was small enough to share a CU; measured in round 4 on configs[1]'s force model, 1 h: 32 768 trajectories 104.2 ms
    // with eight waves against 69.1 with sixteen, 131 072: 521 (four) / 410 (eight) / 277 ms (sixteen) - 0.44 / 0.56 / 0.83 of the
    // FP64 peak.)  The shape therefore depends on the configuration alone, which is also what tuning.deterministic promises.
    (void)n;
    const int deg = ctx->host_cfg.deg;
    int want = 16;
    if (deg < 8) want = std::min(want, 4);
    else if (deg < 24) want = std::min(want, 8);
    return want;
}

extern "C" int32_t nyx_hip_ctx_set_column_waves(nyx_hip_ctx *ctx, int32_t waves) {
    if (!ctx || waves < 0 || waves > DEV_MAX_WAVES) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    ctx->forced_waves = waves;
    return NYX_HIP_RC_OK;
}

extern "C" int32_t nyx_hip_last_coop_helpers(nyx_hip_ctx *ctx) { return ctx ? ctx->last_coop_helpers : 0; }

// Introspection: the per-wave column weights of the last launch's workgroup shape and the spread (max - min) / mean of the
// per-wave windows measured when they were calibrated (-1 = structural default weights, never calibrated).  out[17].
extern "C" int32_t nyx_hip_debug_weights(nyx_hip_ctx *ctx, double *out) {
    if (!ctx || !out) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    const auto it = ctx->weights.find(ctx->last_key);
    for (int w = 0; w < DEV_MAX_WAVES; ++w) out[w] = it != ctx->weights.end() ? it->second[w] : 0.0;  // (speed weights; the duties follow in the table)
    const auto sp = ctx->weight_spread.find(ctx->last_key);
    out[DEV_MAX_WAVES] = sp != ctx->weight_spread.end() ? sp->second : -1.0;
    return NYX_HIP_RC_OK;
}

// Shape of the last launch's workgroups (tools): waves, pipelined loop, carried epoch-data fields, chained attempts, ephemeris
// records in LDS, their size in doubles, almanac waves, LDS bytes of the plain kernel.
extern "C" int32_t nyx_hip_debug_layout(nyx_hip_ctx *ctx, int32_t *out) {
    if (!ctx || !out) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    const DevCfg &dc = ctx->host_cfg;
    out[0] = dc.n_waves; out[1] = dc.pipe; out[2] = dc.ed_reuse; out[3] = dc.spec; out[4] = dc.rec_in_lds; out[5] = dc.rec_doubles;
    out[6] = dc.n_alm;
    out[7] = (int32_t)nyx_kernel_lds_bytes(DEV_MAX_WAVES, dc.rec_in_lds ? dc.rec_doubles : 0, 0, dc.ed_reuse);
    return NYX_HIP_RC_OK;
}

// Roles of the last launch's workgroup shape (tools): out[w] = role_kind, out[16 + w] = role_mask of wave w.
extern "C" int32_t nyx_hip_debug_roles(nyx_hip_ctx *ctx, int32_t *out) {
    if (!ctx || !out) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    const DevCfg &dc = ctx->host_cfg;
    for (int w = 0; w < DEV_MAX_WAVES; ++w) { out[w] = dc.role_kind[w]; out[DEV_MAX_WAVES + w] = dc.role_mask[w]; }
    return NYX_HIP_RC_OK;
}

// Calibration of the column weights: 1 = on the device, once per workgroup shape, 0 = the model's weights (tuning.schedule).
extern "C" int32_t nyx_hip_debug_set_calibration(nyx_hip_ctx *ctx, int32_t mode) {
    if (!ctx || mode < 0 || mode > 1) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    ctx->tune.schedule = mode ? NYX_HIP_SCHED_CALIBRATED : NYX_HIP_SCHED_MODEL;
    return NYX_HIP_RC_OK;
}

// The launch-time part of the tuning (cooperative mode, determinism, schedule kind and explicit weights, helper ratio / share /
// mute, profiling) may be changed between launches; the create-time part (stage loop, role layout, feed) is fixed with the context.
extern "C" int32_t nyx_hip_ctx_set_tuning(nyx_hip_ctx *ctx, const nyx_hip_tuning_t *t) {
    if (!ctx) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    const nyx_hip_tuning_t n = resolve_tuning(t), &o = ctx->tune;
    if (n.pipelined != o.pipelined || n.chained_attempts != o.chained_attempts || n.epoch_data_reuse != o.epoch_data_reuse ||
        n.role_fanout != o.role_fanout || n.merge_roles != o.merge_roles || n.harmonics_feed != o.harmonics_feed ||
        n.debug_flags != o.debug_flags || std::memcmp(n.role_duties, o.role_duties, sizeof n.role_duties) != 0) {
        nyx_set_error("nyx_hip_ctx_set_tuning: stage loop, role layout, feed and debug switches are fixed at nyx_hip_ctx_create");
        return NYX_HIP_RC_BAD_ARG;
    }
    ctx->tune = n;
    // the helper share the schedules are built for follows the new request (auto: 0.30 until a cooperative launch derives it from
    // its helper / owner ratio); the rebuilt descriptor is uploaded by the next launch (sched_dirty)
    ctx->host_cfg.coop_frac = n.coop_fraction > 0.0 ? std::min(0.9, std::max(0.05, n.coop_fraction)) : 0.30;
    ctx->weights.clear();
    ctx->weight_spread.clear();
    ctx->sched_dirty = true;
    return NYX_HIP_RC_OK;
}

// Speed weights [0..16), duties [16..32) and the window spread [32] of the last launch's workgroup shape (tools).
extern "C" int32_t nyx_hip_debug_schedule_weights(nyx_hip_ctx *ctx, double *out) {
    if (!ctx || !out) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    const auto it = ctx->weights.find(ctx->last_key);
    for (int w = 0; w < 2 * DEV_MAX_WAVES; ++w) out[w] = it != ctx->weights.end() ? it->second[w] : 0.0;
    const auto sp = ctx->weight_spread.find(ctx->last_key);
    out[2 * DEV_MAX_WAVES] = sp != ctx->weight_spread.end() ? sp->second : -1.0;
    return NYX_HIP_RC_OK;
}

// Table rows each wave walks under schedule `sched` (DEV_SCHED_*) of the current descriptor, and the role duties the water-filling
// charged (integrator, almanac, perturbations; harmonics-term units): what tools/tune_schedule.py turns measured windows into weights with.
extern "C" int32_t nyx_hip_debug_schedule_rows(nyx_hip_ctx *ctx, int32_t sched, int32_t *rows16, double *duties3) {
    if (!ctx || !rows16 || sched < 0 || sched >= DEV_N_SCHED) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    const DevSched &sd = ctx->host_cfg.sched[sched];
    for (int w = 0; w < DEV_MAX_WAVES; ++w) {
        int rows = 0;
        for (int q = 0; q < sd.n_ranges[w]; ++q)
            for (int c = sd.range_c0[w][q]; c < sd.range_c0[w][q] + sd.range_cnt[w][q]; ++c)
                rows += (c >= 0 && c < (int)ctx->col_len.size()) ? ctx->col_len[c] : 0;
        rows16[w] = rows;
    }
    if (duties3) for (int k = 0; k < 3; ++k) duties3[k] = ctx->role_handicap[k];
    return NYX_HIP_RC_OK;
}

// Test / tuning hook: STM layout of the following launches (-1 = by ensemble size, 0 = D3 64-lane, 1 = quad).
extern "C" int32_t nyx_hip_debug_set_stm_layout(nyx_hip_ctx *ctx, int32_t quad) {
    if (!ctx || quad < -1 || quad > 1) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    ctx->forced_quad = quad;
    return NYX_HIP_RC_OK;
}

extern "C" double nyx_hip_last_kernel_ms(nyx_hip_ctx *ctx) {
    if (!ctx || !ctx->ev1) return -1.0;
    CTX_LOCK(ctx);
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0;
    ctx->last_ms = ms;
    return ms;
}

// Cycle accounting of workgroup 0 of the last launch (NYX_HIP_PROFILE=1): out[17][8], see the kernel (row 16: mailbox counters).
extern "C" int32_t nyx_hip_debug_profile_helper(nyx_hip_ctx *ctx, int64_t *out /* [19][8] */) {
    if (!ctx || !ctx->d_prof) return NYX_HIP_RC_BAD_ARG;
    if (hipMemcpy(out, ctx->d_prof + 17 * 8, 19 * 8 * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return NYX_HIP_RC_HIP_ERROR;  // (row 16 of `out` = row 33: the owner's latency loop; rows 17-18 = the integrator's stage in pieces)
    return NYX_HIP_RC_OK;
}
extern "C" int32_t nyx_hip_debug_profile(nyx_hip_ctx *ctx, int64_t *out) {
    if (!ctx || !ctx->d_prof) return NYX_HIP_RC_BAD_ARG;
    if (hipMemcpy(out, ctx->d_prof, 17 * 8 * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return NYX_HIP_RC_HIP_ERROR;
    return NYX_HIP_RC_OK;
}

extern "C" void nyx_hip_ctx_destroy(nyx_hip_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipFree(ctx->d_cfg); hipFree(ctx->d_htab); hipFree(ctx->d_htab2); hipFree(ctx->d_cols2); hipFree(ctx->d_stm_hist); hipFree(ctx->d_hyb); for (int k = 0; k < 3; ++k) { hipFree(ctx->d_rs[k]); hipFree(ctx->d_rs_cols[k]); } hipFree(ctx->d_cols); hipFree(ctx->d_records);
    free_arrays(ctx->in);
    free_arrays(ctx->out);
    free_arrays(ctx->cal);
    (void)hipFree(ctx->d_swap);
    (void)hipFree(ctx->d_mom);
    mailbox_release(ctx->device, ctx->d_coop, ctx->coop_cap, !(ctx->tune.debug_flags & 0x100000));
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->ev_done) hipEventDestroy(ctx->ev_done);
    delete ctx;
}

extern "C" int32_t nyx_hip_ctx_create(const nyx_hip_config_t *cfg, int32_t device, nyx_hip_ctx **out) {
    if (!cfg || !out) { nyx_set_error("null argument"); return NYX_HIP_RC_BAD_ARG; }
    *out = nullptr;
    if (cfg->abi_version != NYX_HIP_ABI_VERSION) { nyx_set_error("ABI version mismatch"); return NYX_HIP_RC_BAD_ARG; }
    const nyx_hip_integ_opts_t &o = cfg->opts;
    if (o.method < 0 || o.method > 5 || o.error_ctrl < 0 || o.error_ctrl > 6) { nyx_set_error("bad method / error_ctrl"); return NYX_HIP_RC_BAD_ARG; }
    if ((cfg->flags & NYX_HIP_FLAG_STM) && o.error_ctrl != NYX_HIP_RSS_CARTESIAN_STEP && o.error_ctrl != NYX_HIP_RSS_CARTESIAN_STATE) {
        nyx_set_error("STM propagation on the device supports the RSSCartesianStep / RSSCartesianState error controls only");
        return NYX_HIP_RC_UNSUPPORTED;
    }
    if ((cfg->flags & NYX_HIP_FLAG_STM_TEXTBOOK) && !(cfg->flags & NYX_HIP_FLAG_STM)) { nyx_set_error("NYX_HIP_FLAG_STM_TEXTBOOK without NYX_HIP_FLAG_STM"); return NYX_HIP_RC_BAD_ARG; }
    if (cfg->drag && (cfg->flags & NYX_HIP_FLAG_STM)) {  // PartialsUndefined in the reference too (drag.rs:286-294)
        nyx_set_error("drag has no partials: STM propagation with drag is undefined");
        return NYX_HIP_RC_UNSUPPORTED;
    }
    if (cfg->drag && cfg->gravity && std::memcmp(&cfg->drag->rotation, &cfg->gravity->rotation, sizeof(nyx_hip_rotation_t)) != 0) {
        nyx_set_error("device path: the drag frame must be the gravity-field frame when both are present");
        return NYX_HIP_RC_UNSUPPORTED;
    }
    if (cfg->tides) {
        const nyx_hip_rotation_t *other = cfg->gravity ? &cfg->gravity->rotation : (cfg->drag ? &cfg->drag->rotation : nullptr);
        if (other && std::memcmp(&cfg->tides->rotation, other, sizeof(nyx_hip_rotation_t)) != 0) {
            nyx_set_error("device path: the tidal frame must be the gravity-field / drag frame when they are present");
            return NYX_HIP_RC_UNSUPPORTED;
        }
        if (cfg->tides->n_perturbers < 0 || cfg->tides->n_perturbers > NYX_HIP_MAX_BODIES || !(cfg->tides->mu_km3_s2 > 0.0) ||
            !(cfg->tides->eq_radius_km > 0.0)) {
            nyx_set_error("bad solid-tides model");
            return NYX_HIP_RC_BAD_ARG;
        }
    }
    // ---- plain-data validation: nothing below may index past what the device code assumes
    if (cfg->n_bodies < 0 || cfg->n_bodies > NYX_HIP_MAX_BODIES || (cfg->n_bodies > 0 && !cfg->bodies) || cfg->n_segments < 0 ||
        (cfg->n_segments > 0 && !cfg->segments) || cfg->n_point_masses < 0 || cfg->n_point_masses > NYX_HIP_MAX_BODIES) {
        nyx_set_error("bad body / segment / point-mass counts"); return NYX_HIP_RC_BAD_ARG;
    }
    for (int b = 0; b < cfg->n_bodies; ++b)
        if (cfg->bodies[b].n_chain < 0 || cfg->bodies[b].n_chain > NYX_HIP_MAX_CHAIN) {
            nyx_set_error("body %d: n_chain %d outside 0..%d", b, cfg->bodies[b].n_chain, NYX_HIP_MAX_CHAIN); return NYX_HIP_RC_BAD_ARG;
        }
    for (int i = 0; i < cfg->n_segments; ++i) {
        const nyx_hip_cheby_segment_t &sg = cfg->segments[i];
        if (sg.n_coeffs < 1 || sg.n_records < 1 || !(sg.interval_s > 0.0) || !sg.records) {
            nyx_set_error("segment %d: n_coeffs >= 1, n_records >= 1, interval_s > 0 and records are required", i); return NYX_HIP_RC_BAD_ARG;
        }
        if (sg.n_coeffs > NYX_HIP_MAX_CHEBY_COEFFS) {  // (cheby_eval: a 16-wide register window, a rolled loop up to this limit)
            nyx_set_error("segment %d: %d Chebyshev coefficients per component, the device path evaluates at most %d", i, sg.n_coeffs, NYX_HIP_MAX_CHEBY_COEFFS);
            return NYX_HIP_RC_UNSUPPORTED;
        }
    }
    for (const nyx_hip_rotation_t *r : {cfg->gravity ? &cfg->gravity->rotation : nullptr, cfg->drag ? &cfg->drag->rotation : nullptr,
                                        cfg->tides ? &cfg->tides->rotation : nullptr, cfg->gravity2 ? &cfg->gravity2->rotation : nullptr})
        if (r)
            if (const char *why = check_rotation(*r, cfg->n_segments)) { nyx_set_error("body-fixed orientation: %s", why); return NYX_HIP_RC_BAD_ARG; }
    if (nyx_hip_device_count() <= device || device < 0) { nyx_set_error("no HIP device %d", device); return NYX_HIP_RC_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device));

    nyx_hip_ctx *ctx = new nyx_hip_ctx();
    ctx->device = device;
    ExpKnobs xk;
    ctx->tune = resolve_tuning(cfg->tuning, &xk);
    ctx->block_schedule = (ctx->tune.debug_flags & 0x8000) == 0;  // (0x8000: the two-ended column fill of rounds 1-3 everywhere)
    ctx->block_force = (ctx->tune.debug_flags & 0x10000) != 0;    // (0x10000: contiguous runs whatever the feed - the A/B partner of the streamed walk)
    ctx->fit_big = xk.fit_big; ctx->fit_quad = xk.fit_quad; ctx->fit_solo = xk.fit_solo;
    ctx->fit_partition = (ctx->tune.debug_flags & 0x2000000) == 0;  // (0x2000000: the linear partition of round 4 for the cooperative 70x70 shape too, fill_schedule)
    ctx->coop_deal = (ctx->tune.debug_flags & 0x400000) ? 0 : 1;  // (0x400000: the helper dealing of rounds 1-4 - the longest columns, one per wave)
    if (xk.fast_weight > 0.0) ctx->coop_fast_weight = xk.fast_weight;  // (experiment knobs of the tools, never of a caller: resolve_tuning)
    if (xk.start_rows >= 0.0) ctx->coop_start_rows = xk.start_rows;
    if (xk.deal >= 0) ctx->coop_deal = xk.deal;
    for (int k = 0; k < 8; ++k) ctx->role_place[k] = xk.place[k];
    ctx->role_place_sums = xk.place_sums; ctx->role_place_twobody = xk.place_twobody;
    DevCfg &dc = ctx->host_cfg;
    std::memset(&dc, 0, sizeof dc);
    const NyxTableau &tb = NYX_TABLEAUX[o.method];
    dc.stages = tb.stages; dc.order = tb.order;
    dc.fixed_step = o.fixed_step; dc.error_ctrl = o.error_ctrl; dc.attempts = o.attempts; dc.flags = (int32_t)cfg->flags;
    dc.flags |= ctx->tune.debug_flags & 0xff00;  // timing-only switches
    dc.tol = o.tolerance;
    dc.init_step_ns = o.init_step_ns; dc.min_step_ns = o.min_step_ns; dc.max_step_ns = o.max_step_ns;
    dc.min_step_s = ns_to_seconds_host(o.min_step_ns);
    dc.max_step_s = ns_to_seconds_host(o.max_step_ns);
    dc.inv_order = 1.0 / (double)tb.order;
    dc.inv_order_m1 = 1.0 / (double)(tb.order - 1);
    {
        int a_idx = 0;
        dc.c[0] = 0.0;
        for (int i = 0; i < tb.stages - 1; ++i) {  // c_i = running sum of row i (reference instance.rs:379-387)
            double ci = 0.0;
            for (int j = 0; j <= i; ++j) { dc.a[a_idx] = tb.a[a_idx]; ci += tb.a[a_idx]; ++a_idx; }
            dc.c[i + 1] = ci;
        }
        for (int i = 0; i < tb.stages; ++i) { dc.b[i] = tb.b[i]; dc.bdiff[i] = tb.b[i] - tb.b[i + tb.stages]; }
    }
    dc.mu_central = cfg->central_mu_km3_s2;
    dc.g_slot = -1;

    // ---- bodies -> slots (every non-central body referenced by a model)
    std::vector<int> slot_of(cfg->n_bodies, -1);
    auto slot_for = [&](int b) -> int {
        if (b < 0 || b >= cfg->n_bodies) return -2;
        if (cfg->bodies[b].n_chain == 0) return -1;  // the integration centre
        if (slot_of[b] >= 0) return slot_of[b];
        if (dc.n_slots >= DEV_MAX_SLOTS) return -2;
        const nyx_hip_body_t &bd = cfg->bodies[b];
        DevSlot &s = dc.slot[dc.n_slots];
        s.mu = bd.mu_km3_s2; s.radius = bd.mean_radius_km; s.n_chain = bd.n_chain;
        for (int k = 0; k < bd.n_chain; ++k) { s.seg[k] = bd.chain_segment[k]; s.sign[k] = (double)bd.chain_sign[k]; }
        slot_of[b] = dc.n_slots++;
        return slot_of[b];
    };
    for (int b = 0; b < cfg->n_bodies; ++b)
        if (cfg->bodies[b].n_chain == 0) dc.central_radius = cfg->bodies[b].mean_radius_km;
    if (cfg->state_frame_body != 0) {  // opts.integration_frame: the body the states are centred on
        const int b = cfg->state_frame_body;
        if (b < 0 || b >= cfg->n_bodies || cfg->bodies[b].n_chain > 4) { delete ctx; nyx_set_error("state_frame_body: not a body of this configuration"); return NYX_HIP_RC_BAD_ARG; }
        ctx->swap_n_chain = cfg->bodies[b].n_chain;
        for (int k = 0; k < ctx->swap_n_chain; ++k) {
            const int sgi = cfg->bodies[b].chain_segment[k];
            if (sgi < 0 || sgi >= cfg->n_segments) { delete ctx; nyx_set_error("state_frame_body: bad chain segment index"); return NYX_HIP_RC_BAD_ARG; }
            ctx->swap_seg[k] = sgi;
            ctx->swap_sign[k] = (double)cfg->bodies[b].chain_sign[k];
        }
    }
    for (int k = 0; k < cfg->n_point_masses; ++k) {
        int s = slot_for(cfg->point_mass_body[k]);
        if (s == -2) { delete ctx; nyx_set_error("too many / invalid point-mass bodies"); return NYX_HIP_RC_BAD_ARG; }
        if (s == -1) continue;  // central body is skipped by PointMasses::eom (orbital.rs:219-222)
        dc.pm_slot[dc.n_pm++] = s;
    }
    if (cfg->srp) {
        dc.has_srp = 1;
        dc.srp_estimate = cfg->srp->estimate;
        dc.phi = cfg->srp->phi_w_m2;
        dc.c_m_s = cfg->speed_of_light_km_s * 1e3;
        int s = slot_for(cfg->srp->sun_body);
        if (s < 0) { delete ctx; nyx_set_error("SRP light source must be a non-central body with an ephemeris"); return NYX_HIP_RC_BAD_ARG; }
        dc.sun_slot = s;
        dc.n_shadow = cfg->srp->n_shadow_bodies;
        if (dc.n_shadow > DEV_MAX_SLOTS) { delete ctx; nyx_set_error("too many shadow bodies"); return NYX_HIP_RC_BAD_ARG; }
        for (int k = 0; k < dc.n_shadow; ++k) {
            int sb = slot_for(cfg->srp->shadow_body[k]);
            if (sb == -2) { delete ctx; nyx_set_error("invalid shadow body"); return NYX_HIP_RC_BAD_ARG; }
            dc.shadow_slot[k] = sb;
        }
    }
    if (cfg->tides) {
        const nyx_hip_solid_tides_t *td = cfg->tides;
        dc.has_tides = 1;
        dc.t_k2_5 = td->k2 / (2.0 * 2.0 + 1.0);
        dc.t_k3_7 = td->k3 / (2.0 * 3.0 + 1.0);
        dc.t_mu = td->mu_km3_s2; dc.t_re = td->eq_radius_km;
        copy_rotation(dc.t_rot, td->rotation);
        for (int j = 0; j < td->n_perturbers; ++j) {
            const int b = td->perturber_body[j];
            const int sl = slot_for(b);
            if (sl < 0) { delete ctx; nyx_set_error("tidal perturbers must be non-central bodies with an ephemeris (and fit the %d slots)", DEV_MAX_SLOTS); return NYX_HIP_RC_BAD_ARG; }
            dc.t_slot[dc.t_n] = sl;
            dc.t_deg3[dc.t_n] = td->compute_degree_3[j] ? 1 : 0;
            dc.t_gm_ratio[dc.t_n] = cfg->bodies[b].mu_km3_s2 / td->mu_km3_s2;
            dc.t_n++;
        }
    }
    // ---- segments
    if (cfg->n_segments > DEV_MAX_SEG) { delete ctx; nyx_set_error("too many ephemeris segments"); return NYX_HIP_RC_BAD_ARG; }
    std::vector<double> records;
    dc.n_seg = cfg->n_segments;
    // Device layout of the records.  cheby_eval() works on a sixteen-coefficient register window and has to blank the entries past a
    // segment's own count (two v_cndmask per coefficient on the almanac wave, every stage).  When the whole table stays small the
    // records of segments with <= 16 coefficients are therefore laid out SIXTEEN wide, zero-padded: the zeros are in the table, the
    // selects go (DevSeg.stride = 50 tells the kernel; same values, same bits).  (0x2000: A/B switch, same results)
    const int kChebWin = 16;
    bool pad16 = !(ctx->tune.debug_flags & 0x2000);
    {
        size_t packed = 0, padded = 0;
        for (int i = 0; i < cfg->n_segments; ++i) {
            const nyx_hip_cheby_segment_t &sg = cfg->segments[i];
            packed += (size_t)sg.n_records * (size_t)(2 + 3 * sg.n_coeffs);
            padded += (size_t)sg.n_records * (size_t)(2 + 3 * (sg.n_coeffs <= kChebWin ? kChebWin : sg.n_coeffs));
        }
        auto fits_lds = [&](size_t doubles) {  // the staging rule further down (rec_in_lds), for this context's kernel family
            const int rd = (int)doubles + 16;
            if ((size_t)rd * sizeof(double) > 24 * 1024) return false;
            return (cfg->flags & NYX_HIP_FLAG_STM) ? nyx_kernel_lds_bytes(DEV_MAX_WAVES_STM, rd, 1, 0) <= 160 * 1024
                                                   : nyx_kernel_lds_bytes(DEV_MAX_WAVES, rd, 0, 0) <= 160 * 1024;
        };
        if (fits_lds(packed) && !fits_lds(padded)) pad16 = false;  // (never push the table out of LDS)
        if (padded * sizeof(double) > (size_t)8 << 20) pad16 = false;
    }
    for (int i = 0; i < cfg->n_segments; ++i) {
        const nyx_hip_cheby_segment_t &sg = cfg->segments[i];
        DevSeg &d = dc.seg[i];
        d.init_et = sg.init_et_s; d.interval = sg.interval_s; d.n_rec = sg.n_records; d.n_coef = sg.n_coeffs;
        d.end_et = sg.init_et_s + sg.interval_s * (double)sg.n_records;
        const int src_stride = 2 + 3 * sg.n_coeffs;
        d.offset = (int32_t)records.size();
        if (pad16 && sg.n_coeffs < kChebWin) {
            d.stride = 2 + 3 * kChebWin;
            for (int r = 0; r < sg.n_records; ++r) {
                const double *src = sg.records + (size_t)r * src_stride;
                records.push_back(src[0]); records.push_back(src[1]);
                for (int c = 0; c < 3; ++c)
                    for (int j = 0; j < kChebWin; ++j) records.push_back(j < sg.n_coeffs ? src[2 + c * sg.n_coeffs + j] : 0.0);
            }
        } else {
            d.stride = src_stride;
            records.insert(records.end(), sg.records, sg.records + (size_t)sg.n_records * d.stride);
        }
    }
    for (int s = 0; s < dc.n_slots; ++s)
        for (int k = 0; k < dc.slot[s].n_chain; ++k)
            if (dc.slot[s].seg[k] < 0 || dc.slot[s].seg[k] >= dc.n_seg) { delete ctx; nyx_set_error("bad chain segment index"); return NYX_HIP_RC_BAD_ARG; }

    // ---- gravity field
    std::vector<HarmEntry> tab;
    std::vector<ColHdr> cols;
    if (cfg->gravity) {
        const nyx_hip_gravity_field_t *g = cfg->gravity;
        if (g->degree < 1 || !g->c_nm || !g->s_nm) { delete ctx; nyx_set_error("bad gravity field"); return NYX_HIP_RC_BAD_ARG; }
        dc.has_grav = 1; dc.deg = g->degree; dc.ord = std::min(g->order, g->degree);
        dc.g_slot = -1;
        if (g->offset_body != 0) {  // the field of another body than the integration centre (gravity_field.rs:150-154)
            const int sl = slot_for(g->offset_body - 1);
            if (sl == -2) { delete ctx; nyx_set_error("gravity field: offset_body is not a body of this configuration (or the %d body slots are taken)", DEV_MAX_SLOTS); return NYX_HIP_RC_BAD_ARG; }
            dc.g_slot = sl;  // (-1: offset_body names the integration centre itself)
            if (sl >= 0)
                for (int k = 0; k < dc.slot[sl].n_chain; ++k)
                    if (dc.slot[sl].seg[k] < 0 || dc.slot[sl].seg[k] >= dc.n_seg) { delete ctx; nyx_set_error("bad chain segment index"); return NYX_HIP_RC_BAD_ARG; }
        }
        dc.g_mu = g->mu_km3_s2; dc.g_re = g->eq_radius_km; dc.g_inv_re = 1.0 / g->eq_radius_km;
        copy_rotation(dc.g_rot, g->rotation);
        int n_cols = 0;
        build_harmonics(g, tab, cols, ctx->col_len, n_cols);
        dc.n_cols = n_cols;
        ctx->h_tab = tab;
        ctx->h_cols = cols;
    }
    std::vector<HarmEntry> tab2;
    std::vector<ColHdr> cols2;
    int terms2 = 0;
    if (cfg->gravity2) {
        const nyx_hip_gravity_field_t *g = cfg->gravity2;
        if (!cfg->gravity) { delete ctx; nyx_set_error("gravity2 without gravity: a single field goes into `gravity`"); return NYX_HIP_RC_BAD_ARG; }
        if (g->degree < 1 || !g->c_nm || !g->s_nm) { delete ctx; nyx_set_error("bad second gravity field"); return NYX_HIP_RC_BAD_ARG; }
        dc.has_grav2 = 1;
        dc.g2_mu = g->mu_km3_s2; dc.g2_re = g->eq_radius_km; dc.g2_inv_re = 1.0 / g->eq_radius_km;
        copy_rotation(dc.g2_rot, g->rotation);
        dc.g2_slot = -1;
        if (g->offset_body != 0) {
            const int sl = slot_for(g->offset_body - 1);
            if (sl == -2) { delete ctx; nyx_set_error("second gravity field: offset_body is not a body of this configuration (or the %d body slots are taken)", DEV_MAX_SLOTS); return NYX_HIP_RC_BAD_ARG; }
            dc.g2_slot = sl;
            if (sl >= 0)
                for (int k = 0; k < dc.slot[sl].n_chain; ++k)
                    if (dc.slot[sl].seg[k] < 0 || dc.slot[sl].seg[k] >= dc.n_seg) { delete ctx; nyx_set_error("bad chain segment index"); return NYX_HIP_RC_BAD_ARG; }
        }
        std::vector<int32_t> len2;
        int n_cols2 = 0;
        build_harmonics(g, tab2, cols2, len2, n_cols2);
        dc.n_cols2 = n_cols2;
        for (int32_t l : len2) terms2 += l;
        ctx->terms2 = terms2;
    }
    if (cfg->drag) {
        const nyx_hip_drag_t *dg = cfg->drag;
        if (dg->density < 0 || dg->density > 2) { delete ctx; nyx_set_error("bad drag density model"); return NYX_HIP_RC_BAD_ARG; }
        dc.has_drag = 1; dc.drag_density = dg->density;
        dc.drag_rho0 = dg->rho0; dc.drag_r0 = dg->r0; dc.drag_ref_alt_m = dg->ref_alt_m; dc.drag_max_alt_m = dg->max_alt_m;
        dc.drag_re = dg->eq_radius_km;
        copy_rotation(dc.d_rot, dg->rotation);
    }
    // serial duties of the role waves per force evaluation, in units of one harmonics term (~10 f64 ops):
    // integrator: stage combination, body-fixed transform, fold of the partials; almanac: 3 sincos + Chebyshev
    // chains; perturbations: third-body and SRP/eclipse terms.
    {
        int nseg_eval = 0;
        for (int s = 0; s < dc.n_slots; ++s) nseg_eval += dc.slot[s].n_chain;
        // (refitted in round 3 to the duties the calibration measures on the BASELINE workloads: 70x70 + Sun / Moon + SRP gives
        //  integrator 66, almanac 140, perturbations 52 harmonics-term units - the first formulas were 2.2x too low)
        ctx->role_handicap[0] = 60.0;
        ctx->role_handicap[1] = 26.0 * nseg_eval + (dc.has_grav ? 38.0 : 0.0);
        ctx->role_handicap[2] = (dc.has_grav2 ? 38.0 + 1.1 * terms2 : 0.0) + 13.0 * dc.n_pm + (dc.has_srp ? 13.0 + 13.0 * dc.n_shadow : 0.0) + (dc.has_drag ? 22.0 : 0.0) +
                                (dc.has_tides ? 30.0 + 17.0 * dc.t_n : 0.0);
        if (any_nonzero(ctx->tune.role_duties, 3))
            for (int k = 0; k < 3; ++k) ctx->role_handicap[k] = ctx->tune.role_duties[k];
    }
    {
        // the body-fixed frame of the epoch data (the kernel's choice: gravity field, else drag, else tides): a polynomial IAU
        // orientation is advanced from a base epoch instead of being evaluated with three full-range sincos per stage
        const DevRot &er = dc.has_grav ? dc.g_rot : (dc.has_drag ? dc.d_rot : dc.t_rot);
        // (plain kernels only: the STM tests hold the device to the oracle's step sequence, bit for bit)
        dc.dcm_incr = ((dc.has_grav || dc.has_drag || dc.has_tides) && er.kind == NYX_HIP_ROT_IAU && er.n_np == 0 && !(cfg->flags & NYX_HIP_FLAG_STM) &&
                       !(ctx->tune.debug_flags & 0x4000)) ? 1 : 0;
    }
    records.resize(records.size() + 16, 0.0);  // padding for the 16-wide coefficient window
    dc.rec_doubles = (int32_t)records.size();
    dc.rec_in_lds = (records.size() * sizeof(double) <= 24 * 1024) ? 1 : 0;
    if ((cfg->flags & NYX_HIP_FLAG_STM) && nyx_kernel_lds_bytes(DEV_MAX_WAVES_STM, dc.rec_doubles, 1, 0) > 160 * 1024) dc.rec_in_lds = 0;
    if (!(cfg->flags & NYX_HIP_FLAG_STM) && nyx_kernel_lds_bytes(DEV_MAX_WAVES, dc.rec_doubles, 0, 0) > 160 * 1024) dc.rec_in_lds = 0;
    // stage-0 epoch data carried between attempts (see role_loop): needs an even stage count (the last stage's window
    // then leaves buffer 0 free) and 9 + 3 * n_slots doubles + 20 bytes of LDS per lane
    dc.ed_reuse = 0;
    if (!(cfg->flags & NYX_HIP_FLAG_STM) && dc.stages % 2 == 0 && ctx->tune.epoch_data_reuse != 0) {
        const int nf = 9 + 3 * dc.n_slots;
        if (nyx_kernel_lds_bytes(DEV_MAX_WAVES, dc.rec_in_lds ? dc.rec_doubles : 0, 0, nf) <= 160 * 1024) dc.ed_reuse = nf;
    }
    ctx->ed_reuse_fit = dc.ed_reuse;
    dc.coop_late = (ctx->tune.debug_flags & 0x1000000) ? 0 : 1;  // (0x1000000: the answer collected inside the window, rounds 1-4)
    dc.coop_frac = 0.30;  // measured optimum with two owners per helper (10 000 trajectories, 70x70): 0.28-0.33 is flat
    if (ctx->tune.coop_fraction > 0.0) dc.coop_frac = std::min(0.9, std::max(0.05, ctx->tune.coop_fraction));
    {
        hipDeviceProp_t prop;
        ctx->n_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 0;
    }
    build_schedule(ctx, 1);

    // ---- upload
    dc.hyb = 0;
    dc.harm_feed = 0;
    if (!tab.empty()) {
        std::vector<double> hyb;
        int64_t vec_off = 0;
        build_hybrid(tab, hyb, vec_off);
        HIP_TRY(hipMalloc(&ctx->d_hyb, hyb.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(ctx->d_hyb, hyb.data(), hyb.size() * sizeof(double), hipMemcpyHostToDevice));
        dc.hyb = (uint64_t)ctx->d_hyb;
        dc.hyb_v = (uint64_t)(ctx->d_hyb + vec_off);
        // measured (same box, calibrated): 150x150 cooperative 373 -> 334 ms per 6 250 x 3 h (1.12x); 70x70 alone 1.02-1.11x;
        // 70x70 cooperative (one column per helper wave and job: the walk's start-up weighs more) 0-2 % slower
        // (DevCfg.harm_feed: bit 0 = the trajectory-owning workgroups, bit 1 = the helpers and the owner's fallback for them)
        // Round 4: the trajectory-owning workgroups stream the table from degree 40 on (with ONE contiguous run of columns per wave,
        // fill_schedule: the start-up of a run is what the walk costs at 70x70), the helpers - one column per wave and job - above 95
        dc.harm_feed = dc.n_cols > 96 ? 3 : (dc.n_cols > 40 ? 1 : 0);
        if (ctx->tune.harmonics_feed >= 0) dc.harm_feed = ctx->tune.harmonics_feed == 0 ? 0 : (ctx->tune.harmonics_feed == 2 ? 1 : (ctx->tune.harmonics_feed == 3 ? 2 : 3));
    }
    if (!tab2.empty()) {
        tab2.resize(tab2.size() + 4 * HARM_BATCH, HarmEntry{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0});
        HIP_TRY(hipMalloc(&ctx->d_htab2, tab2.size() * sizeof(HarmEntry)));
        HIP_TRY(hipMemcpy(ctx->d_htab2, tab2.data(), tab2.size() * sizeof(HarmEntry), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&ctx->d_cols2, cols2.size() * sizeof(ColHdr)));
        HIP_TRY(hipMemcpy(ctx->d_cols2, cols2.data(), cols2.size() * sizeof(ColHdr), hipMemcpyHostToDevice));
        dc.htab2 = (uint64_t)ctx->d_htab2;
        dc.cols2 = (uint64_t)ctx->d_cols2;
    }
    HIP_TRY(hipMalloc(&ctx->d_cfg, sizeof(DevCfg)));
    HIP_TRY(hipMemcpy(ctx->d_cfg, &dc, sizeof(DevCfg), hipMemcpyHostToDevice));
    if (!tab.empty()) {
        tab.resize(tab.size() + 4 * HARM_BATCH, HarmEntry{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0});  // the kernel touches a few batches ahead
        HIP_TRY(hipMalloc(&ctx->d_htab, tab.size() * sizeof(HarmEntry)));
        HIP_TRY(hipMemcpy(ctx->d_htab, tab.data(), tab.size() * sizeof(HarmEntry), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&ctx->d_cols, cols.size() * sizeof(ColHdr)));
        HIP_TRY(hipMemcpy(ctx->d_cols, cols.data(), cols.size() * sizeof(ColHdr), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&ctx->d_records, records.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(ctx->d_records, records.data(), records.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipEventCreate(&ctx->ev0));
    HIP_TRY(hipEventCreate(&ctx->ev1));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming));
    *out = ctx;
    return NYX_HIP_RC_OK;
}

// ---------------------------------------------------------------------------------------------
// propagate
// ---------------------------------------------------------------------------------------------

static bool calibration_on(const nyx_hip_ctx *ctx) { return ctx->tune.schedule == NYX_HIP_SCHED_CALIBRATED; }

static int launch(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, nyx_hip_states_t *out, nyx_hip_step_stats_t *st,
                  int64_t duration_ns, int64_t end_epoch_ns, int use_end, hipStream_t stream, bool time_it,
                  const nyx_hip_traj_t *traj, const int64_t *dur_ns, const DevBatch *ev, bool calibrating, bool swapped = false);

// Device views of a DevArrays block (outputs + stats).
static void views_of(DevArrays &d, int64_t n, bool stm, nyx_hip_states_t &so, nyx_hip_step_stats_t &ss) {
    std::memset(&so, 0, sizeof so);
    so.n = n; so.epoch_ns = d.epoch;
    double **f[13] = {&so.x_km, &so.y_km, &so.z_km, &so.vx_km_s, &so.vy_km_s, &so.vz_km_s, &so.cr, &so.cd,
                      &so.prop_mass_kg, &so.dry_mass_kg, &so.extra_mass_kg, &so.srp_area_m2, &so.drag_area_m2};
    for (int k = 0; k < 13; ++k) *f[k] = d.f[k];
    so.step_ns = d.step;
    so.stm = stm ? d.stm : nullptr;
    ss = {d.status, d.last_step, d.last_error, d.last_attempts, d.n_acc, d.n_rej, d.n_evals};
}

// On-device calibration of the column schedule of the shape the NEXT launch of `in` will have (replaces tables fitted
// offline to one force model, and the guessed role handicaps).  Up to four short launches of the workload's own first
// steps (30 steps; 4 with the STM) into scratch outputs, with the in-kernel cycle accounting on.  Per wave of workgroup 0:
// duty[w] = cycles of role work inside the window, harm[w] = cycles in its columns, hence c[w] = harm / table entries = what
// a table entry costs THIS wave (the four waves of a SIMD are arbitrated oldest first: the young ones are slower).  The
// water-filling then gets the speed weight cbar / c[w] and the handicap duty[w] / c[w] (in entries), which makes
// duty + columns equal across the waves; iterated with damping because the shares interact through the shared SIMDs.
// Rounded to 1/64 and kept for the life of the context: launches of one context are deterministic.
static int calibrate(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, hipStream_t stream, bool backward = false) {
    const bool stm = (ctx->host_cfg.flags & NYX_HIP_FLAG_STM) != 0;
    const int64_t n = in->n;
    if (int rc = ensure_arrays(ctx->cal, n, true)) return rc;
    if (stm && ctx->cal.stm_cap < n) {
        (void)hipFree(ctx->cal.stm);
        ctx->cal.stm = nullptr;
        HIP_TRY(hipMalloc(&ctx->cal.stm, (size_t)std::max<int64_t>(n, 1024) * 81 * sizeof(double)));
        ctx->cal.stm_cap = std::max<int64_t>(n, 1024);
    }
    nyx_hip_states_t so;
    nyx_hip_step_stats_t ss;
    views_of(ctx->cal, n, stm, so, ss);
    const int64_t dur = (backward ? -1 : 1) * (stm ? 4 : 30) * ctx->host_cfg.init_step_ns;  // the direction of the real request: the ephemerides may end either way
    std::vector<int32_t> cal_status((size_t)n);
    std::array<double, 2 * DEV_MAX_WAVES> w;
    bool have = false;
    nyx_hip_ctx::WKey key(0, 0, 0, 0);
    double spread = 0.0;
    std::vector<int64_t> prof(17 * 8);
    for (int it = 0; it < 4; ++it) {
        if (have) { ctx->weights[key] = w; ctx->sched_dirty = true; }
        if (int rc = launch(ctx, in, &so, &ss, dur, 0, 0, stream, false, nullptr, nullptr, nullptr, true)) return rc;
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(prof.data(), ctx->d_prof, prof.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (ss.status) {  // lanes that died early produce garbage cycle counts: keep the model's weights then
            HIP_TRY(hipMemcpy(cal_status.data(), ss.status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
            bool bad = false;
            for (int64_t q = 0; q < n && !bad; ++q) bad = cal_status[(size_t)q] != NYX_HIP_OK;
            if (bad) { have = false; ctx->weights.erase(ctx->last_key); ctx->sched_dirty = true; break; }
        }
        key = ctx->last_key;
        const int nw = std::get<0>(key);
        const DevSched &sd = ctx->host_cfg.sched[std::get<3>(key) >= 0 ? DEV_SCHED_PRIMARY : DEV_SCHED_SOLO];
        // what the launch measured, per wave of workgroup 0: duty = role work inside the window, harm = its columns
        double duty[DEV_MAX_WAVES], harm[DEV_MAX_WAVES], ent[DEV_MAX_WAVES], cpe[DEV_MAX_WAVES];
        double csum = 0.0, lo = 1e300, hi = 0.0, tmean = 0.0;
        int ccnt = 0, tcnt = 0;
        for (int q = 0; q < nw; ++q) {
            duty[q] = (double)prof[q * 8 + 1];
            harm[q] = (double)prof[q * 8 + 2];
            ent[q] = 0.0;
            for (int r = 0; r < sd.n_ranges[q]; ++r)
                for (int c = sd.range_c0[q][r]; c < sd.range_c0[q][r] + sd.range_cnt[q][r]; ++c) ent[q] += ctx->col_len[c];
            cpe[q] = (ent[q] > 0.0 && harm[q] > 0.0) ? harm[q] / ent[q] : 0.0;   // cycles per table entry, as this wave sees them
            if (cpe[q] > 0.0) { csum += cpe[q]; ++ccnt; }
            if (q > 0 || nw < 8) {
                const double t = duty[q] + (ent[q] > 0.0 ? harm[q] : 0.0);
                if (t > 0.0) { lo = std::min(lo, t); hi = std::max(hi, t); tmean += t; ++tcnt; }
            }
        }
        if (ccnt < 2 || tcnt < 2) break;
        const double cbar = csum / ccnt;
        spread = (hi - lo) / (tmean / tcnt);
        std::array<double, 2 * DEV_MAX_WAVES> nwgt;
        for (int q = 0; q < DEV_MAX_WAVES; ++q) {
            // a wave that carried no columns this time is given the speed of its SIMD age class (waves q, q+4, q+8, q+12 share a SIMD)
            double c = q < nw ? cpe[q] : 0.0;
            if (!(c > 0.0)) {
                double a = 0.0; int an = 0;
                for (int k = (q / 4) * 4; k < (q / 4) * 4 + 4 && k < nw; ++k) if (cpe[k] > 0.0) { a += cpe[k]; ++an; }
                c = an ? a / an : cbar;
            }
            nwgt[q] = cbar / c;
            nwgt[DEV_MAX_WAVES + q] = q < nw ? duty[q] / c : 0.0;
        }
        for (int q = 0; q < 2 * DEV_MAX_WAVES; ++q) {
            const double v = have ? 0.5 * (w[q] + nwgt[q]) : nwgt[q];  // damped: the shares interact through the shared SIMDs
            w[q] = std::round(v * 64.0) / 64.0;
        }
        have = true;
        if (it > 0 && spread < 0.08) break;
    }
    if (have) {
        ctx->weights[key] = w;
        ctx->weight_spread[key] = spread;
        ctx->sched_dirty = true;
    }
    return NYX_HIP_RC_OK;
}

static int launch(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, nyx_hip_states_t *out, nyx_hip_step_stats_t *st,
                  int64_t duration_ns, int64_t end_epoch_ns, int use_end, hipStream_t stream, bool time_it,
                  const nyx_hip_traj_t *traj = nullptr, const int64_t *dur_ns = nullptr, const DevBatch *ev = nullptr, bool calibrating = false,
                  bool swapped) {
    CTX_LOCK(ctx);
    if (ctx->swap_n_chain > 0 && !swapped && !calibrating) {
        // opts.integration_frame (instance.rs:117-142, 211-220): translate a COPY of the Cartesian state into the integration frame
        // at the start epochs, propagate that, translate the final states back at their own epochs
        // Dense output and per-trajectory durations go through (round 4).  What the reference's `Traj` holds then (instance.rs:297-326
        // around :117-142): the START state as it was handed in - its own frame -, every published state in the INTEGRATION frame (the
        // channel is fed inside the loop, the translation back is applied to the returned state only, :211-220).  Reproduced as is:
        // entry 0 of the dense output is rewritten with the caller's state below.  The event search is still refused: there the
        // reference returns from inside the loop without translating back (:243-250) - a state whose frame depends on how the run ended.
        if (ev) {
            nyx_set_error("integration-frame swap: the event search does not take states of another frame");
            return NYX_HIP_RC_UNSUPPORTED;
        }
        if (ctx->launched) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_done, 0));  // (the copy below is shared by the launches of this context)
        if (ctx->swap_cap < in->n) {
            (void)hipFree(ctx->d_swap);
            ctx->d_swap = nullptr; ctx->swap_cap = 0;
            HIP_TRY(hipMalloc(&ctx->d_swap, (size_t)7 * (size_t)in->n * sizeof(double)));  // (row 6: the forward shift's status words)
            ctx->swap_cap = in->n;
        }
        nyx_hip_states_t in2 = *in;
        double *rows[6];
        const double *src[6] = {in->x_km, in->y_km, in->z_km, in->vx_km_s, in->vy_km_s, in->vz_km_s};
        for (int q = 0; q < 6; ++q) {
            rows[q] = ctx->d_swap + (size_t)q * (size_t)ctx->swap_cap;
            HIP_TRY(hipMemcpyAsync(rows[q], src[q], (size_t)in->n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
        in2.x_km = rows[0]; in2.y_km = rows[1]; in2.z_km = rows[2]; in2.vx_km_s = rows[3]; in2.vy_km_s = rows[4]; in2.vz_km_s = rows[5];
        // a start epoch outside the swap body's ephemeris gives a clamped, i.e. WRONG, translation: its status is kept aside and
        // merged into the run's status by the back-translation (the propagation launch rewrites the status array in between)
        int32_t *fwd_status = (int32_t *)(ctx->d_swap + (size_t)6 * (size_t)ctx->swap_cap);
        HIP_TRY(hipMemsetAsync(fwd_status, 0, (size_t)in->n * sizeof(int32_t), stream));
        HIP_TRY(nyx_launch_frame_shift(ctx->d_cfg, ctx->d_records, ctx->swap_seg, ctx->swap_sign, ctx->swap_n_chain, in->n, in->epoch_ns,
                                       rows[0], rows[1], rows[2], rows[3], rows[4], rows[5], +1.0, fwd_status, nullptr, dur_ns, stream));
        if (int rc = launch(ctx, &in2, out, st, duration_ns, end_epoch_ns, use_end, stream, time_it, traj, dur_ns, nullptr, false, true)) return rc;
        if (traj && traj->capacity > 0) {  // entry 0 (step-major: the first n elements of every array) = the state in the caller's frame
            double *dst[6] = {traj->x_km, traj->y_km, traj->z_km, traj->vx_km_s, traj->vy_km_s, traj->vz_km_s};
            for (int q = 0; q < 6; ++q)
                if (dst[q]) HIP_TRY(hipMemcpyAsync(dst[q], src[q], (size_t)in->n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
        HIP_TRY(nyx_launch_frame_shift(ctx->d_cfg, ctx->d_records, ctx->swap_seg, ctx->swap_sign, ctx->swap_n_chain, in->n, out->epoch_ns,
                                       out->x_km, out->y_km, out->z_km, out->vx_km_s, out->vy_km_s, out->vz_km_s, -1.0, st ? st->status : nullptr,
                                       fwd_status, dur_ns, stream));
        HIP_TRY(hipEventRecord(ctx->ev_done, stream));
        return NYX_HIP_RC_OK;
    }
    if (ctx->launched) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_done, 0));  // one launch of a context at a time on the device
    const int nw = pick_waves(ctx, in->n);
    {
        // launch-shape dependent parts of the descriptor: column schedule (waves per workgroup) and whether the ephemeris
        // records fit in LDS next to this layout's buffers (the quad layout is smaller than the D3 one)
        const bool stm_l = (ctx->host_cfg.flags & NYX_HIP_FLAG_STM) != 0;
        const int kind = stm_l ? (pick_quad(ctx, in->n) ? 2 : 1) : 0;
        bool dirty = false;
        if (nw != ctx->host_cfg.n_waves || (kind == 2) != ctx->sched_quad || ctx->sched_dirty) {
            ctx->sched_quad = kind == 2;
            build_schedule(ctx, nw, kind == 2);
            ctx->sched_dirty = false;
            dirty = true;
        }
        const int rd = ctx->host_cfg.rec_doubles;  // (after the schedule: chained attempts give the carried epoch data's LDS back)
        const int want_rec = ((size_t)rd * sizeof(double) <= 24 * 1024 &&
                              nyx_kernel_lds_bytes(DEV_MAX_WAVES, rd, kind, kind == 0 ? ctx->host_cfg.ed_reuse : 0) <= 160 * 1024) ? 1 : 0;
        dirty = dirty || want_rec != ctx->host_cfg.rec_in_lds;
        ctx->host_cfg.rec_in_lds = want_rec;
        if (dirty) HIP_TRY(hipMemcpyAsync(ctx->d_cfg, &ctx->host_cfg, sizeof(DevCfg), hipMemcpyHostToDevice, stream));
    }
    DevBatch bt;
    std::memset(&bt, 0, sizeof bt);
    bt.n = in->n;
    bt.duration_ns = duration_ns; bt.end_epoch_ns = end_epoch_ns; bt.use_end_epoch = use_end;
    bt.epoch_ns = in->epoch_ns;
    bt.x = in->x_km; bt.y = in->y_km; bt.z = in->z_km; bt.vx = in->vx_km_s; bt.vy = in->vy_km_s; bt.vz = in->vz_km_s;
    bt.cr = in->cr; bt.cd = in->cd; bt.mprop = in->prop_mass_kg; bt.mdry = in->dry_mass_kg; bt.mextra = in->extra_mass_kg;
    bt.asrp = in->srp_area_m2; bt.adrag = in->drag_area_m2; bt.step_in = in->step_ns;
    bt.dur_ns = dur_ns;
    bt.pred = ctx->fused_pred;
    if (ev) {  // stop condition: only the ev_* fields of `ev` are read
        bt.ev_on = 1; bt.ev = ev->ev; bt.ev_mu = ev->ev_mu;
        bt.ev_prev = ev->ev_prev; bt.ev_count = ev->ev_count; bt.ev_found = ev->ev_found;
    }
    if (ctx->host_cfg.flags & NYX_HIP_FLAG_STM) {
        if (!in->stm || !out->stm) { nyx_set_error("STM context: in->stm and out->stm are mandatory"); return NYX_HIP_RC_BAD_ARG; }
        bt.stm = in->stm; bt.o_stm = out->stm;
        if (ctx->host_cfg.flags & NYX_HIP_FLAG_STM_TEXTBOOK) {
            if (ctx->stm_hist_cap < in->n) {
                if (ctx->launched) HIP_TRY(hipEventSynchronize(ctx->ev_done));
                (void)hipFree(ctx->d_stm_hist);
                ctx->d_stm_hist = nullptr; ctx->stm_hist_cap = 0;
                const int64_t cap = (in->n + 63) / 64 * 64;
                HIP_TRY(hipMalloc(&ctx->d_stm_hist, (size_t)DEV_MAX_STAGES * 12 * (size_t)cap * sizeof(double)));
                ctx->stm_hist_cap = cap;
            }
            bt.stm_hist = ctx->d_stm_hist;
            bt.stm_hist_stride = ctx->stm_hist_cap;
        }
    }
    bt.o_epoch_ns = out->epoch_ns;
    bt.o_x = out->x_km; bt.o_y = out->y_km; bt.o_z = out->z_km; bt.o_vx = out->vx_km_s; bt.o_vy = out->vy_km_s; bt.o_vz = out->vz_km_s;
    bt.o_cr = out->cr; bt.o_cd = out->cd; bt.o_mprop = out->prop_mass_kg; bt.o_mdry = out->dry_mass_kg;
    bt.o_mextra = out->extra_mass_kg; bt.o_asrp = out->srp_area_m2; bt.o_adrag = out->drag_area_m2; bt.o_step = out->step_ns;
    if (traj && traj->capacity > 0) {
        if (!traj->epoch_ns || !traj->x_km || !traj->y_km || !traj->z_km || !traj->vx_km_s || !traj->vy_km_s || !traj->vz_km_s || !traj->len) {
            nyx_set_error("traj: every array is mandatory");
            return NYX_HIP_RC_BAD_ARG;
        }
        bt.traj_cap = traj->capacity; bt.t_epoch = traj->epoch_ns; bt.t_len = traj->len;
        bt.t_state[0] = traj->x_km; bt.t_state[1] = traj->y_km; bt.t_state[2] = traj->z_km;
        bt.t_state[3] = traj->vx_km_s; bt.t_state[4] = traj->vy_km_s; bt.t_state[5] = traj->vz_km_s;
    }
    if (st) {
        bt.status = st->status; bt.last_step_ns = st->last_step_ns; bt.last_error = st->last_error;
        bt.last_attempts = st->last_attempts; bt.n_acc = st->n_accepted; bt.n_rej = st->n_rejected; bt.n_evals = st->n_evals;
    }
    // Cooperative mode: when the trajectory-owning workgroups leave CUs idle, helper workgroups take over a share of
    // the harmonics columns (propagate_kernel.hip).  Owners and helpers that talk to each other get block indices that
    // agree modulo 8 (round-robin XCD dispatch: same L2); every owner needs a helper for the split to pay off.
    ctx->last_coop_helpers = 0;
    {
        // on by default; tuning.cooperative = 0 (or .deterministic: the split follows the batch size) makes every workgroup work alone
        const bool want = ctx->tune.cooperative != 0 && !ctx->tune.deterministic;
        const int64_t n_own = (in->n + DEV_LANES - 1) / DEV_LANES;
        // Helpers start right behind the owners and fill every CU that is left (round 4: 99 helpers instead of 96 for 157 owners is
        // 3.9 % of the north-star run - the helpers' queues are what the owners wait in; rounds 1-3 rounded both to multiples of
        // eight for XCD affinity, which buys nothing measurable: debug_flags 0x20000 restores it).
        const bool pack = (ctx->tune.debug_flags & 0x20000) == 0;
        const int64_t base = pack ? n_own : (n_own + 7) / 8 * 8;
        const bool stm_ctx = (ctx->host_cfg.flags & NYX_HIP_FLAG_STM) != 0;
        if (want && !stm_ctx && ctx->host_cfg.has_grav && ctx->host_cfg.g_slot < 0 && nw == DEV_MAX_WAVES && ctx->host_cfg.coop_ok &&
            base + 8 <= ctx->n_cu) {
            // (more helpers than owners: the jobs are claimed, not assigned, so extra helpers shorten the queue of a set)
            // Two-part hand-off: when the idle CUs outnumber the owners by a quarter and a helper job holds several columns per wave
            // (large fields), every evaluation's hand-off is split in two sub-jobs for two helper workgroups (see build_schedule); the
            // helper count then goes up to two per owner.  (debug_flags 0x40000 / 0x80000 force two parts / one part.)
            const int64_t free_cus = pack ? ctx->n_cu - base : (ctx->n_cu - base) / 8 * 8;
            int parts = (ctx->host_cfg.n_cols > 96 && 4 * free_cus >= 5 * n_own) ? 2 : 1;
            if (ctx->tune.debug_flags & 0x40000) parts = 2;
            if (ctx->tune.debug_flags & 0x80000) parts = 1;
            // Fan-out mode (round 6): when the idle CUs outnumber the owners at least two to one - what a rank runs when ONE ensemble is
            // cut over the GPUs of a node (configs[1] over 2 / 4 / 8 ranks: 79 / 40 / 20 owners), or a small Monte Carlo - every owner gets
            // K = idle CUs / owners (<= DEV_FAN_MAX) DEDICATED helper workgroups and hands them all but its shortest columns.  Measured
            // before it existed (round 6, profiles/round06_shard_sizes_before.log): 5 000 / 2 500 / 1 250 trajectories x 24 h ran
            // 648 / 642 / 639 ms against 615 for 10 000 - a rank of 8 was no faster than one GPU alone.  Fields up to degree 95 (larger
            // ones keep the two-part claim mode, whose jobs hold several columns per wave).  debug_flags 0x8000000 switches it off.
            bool fan = ctx->host_cfg.n_cols <= 96 && free_cus >= 2 * n_own && n_own >= 1 && !(ctx->tune.debug_flags & 0x8000000) &&
                       !(ctx->tune.debug_flags & (0x40000 | 0x80000)) && !(ctx->tune.coop_helper_ratio > 0.0);
            if (fan) parts = (int)std::min<int64_t>(DEV_FAN_MAX, free_cus / n_own);
            double h_ratio = parts == 2 ? 2.0 : 1.0;
            if (ctx->tune.coop_helper_ratio > 0.0) h_ratio = std::min(3.0, std::max(0.25, ctx->tune.coop_helper_ratio));
            const int64_t helpers = fan ? n_own * parts : std::min<int64_t>((int64_t)((double)n_own * h_ratio), free_cus);
            if (parts != ctx->coop_parts || fan != ctx->coop_fan) {
                ctx->coop_parts = parts;
                ctx->coop_fan = fan;
                build_schedule(ctx, nw, false);
                HIP_TRY(hipMemcpyAsync(ctx->d_cfg, &ctx->host_cfg, sizeof(DevCfg), hipMemcpyHostToDevice, stream));
            }
            if (fan ? (ctx->host_cfg.coop_ok != 0) : (helpers >= 8 && 4 * helpers >= n_own)) {
                // share of the terms the helpers take: owners keep (1 - x), each helper does x * owners / helpers jobs' worth
                // per evaluation period, plus its hand-off overhead: x ~ 0.95 r / (1 + r) with r = helpers / owners
                if (!fan && !(ctx->tune.coop_fraction > 0.0)) {
                    const double r = (double)helpers / (double)n_own;
                    // (two parts, measured on configs[4] with 158 helpers for 98 owners: 0.55 / 0.60 / 0.65 / 0.70 / 0.75 of the terms ->
                    //  98.1 / 97.9 / 93.4 / 92.9 / 102.6 ms per hour of the ensemble - half a job per helper takes the knee further out)
                    const double x = ctx->coop_parts == 2 ? std::min(0.68, std::max(0.10, 1.10 * r / (1.0 + r)))
                                                          : std::min(0.55, std::max(0.10, 0.95 * r / (1.0 + r)));
                    if (std::fabs(x - ctx->host_cfg.coop_frac) > 0.01) {
                        ctx->host_cfg.coop_frac = x;
                        build_schedule(ctx, nw, false);
                        HIP_TRY(hipMemcpyAsync(ctx->d_cfg, &ctx->host_cfg, sizeof(DevCfg), hipMemcpyHostToDevice, stream));
                    }
                }
                bool have_boxes = ctx->coop_cap >= n_own;
                if (!have_boxes) {
                    // the block goes back to the process-wide pool, where another context (another host thread) may take and clear it
                    // at once: not before every launch of THIS context that uses it has finished (the device entry points are asynchronous)
                    if (ctx->d_coop && ctx->launched) HIP_TRY(hipEventSynchronize(ctx->ev_done));
                    mailbox_release(ctx->device, ctx->d_coop, ctx->coop_cap, !(ctx->tune.debug_flags & 0x100000));
                    ctx->d_coop = nullptr;
                    ctx->coop_cap = 0;
                    // uncached device memory: the mailboxes are coherent across the XCDs' L2s without any cache
                    // write-back / invalidate in the kernel (those would also flush the harmonics table out of L2)
                    int64_t got = 0;
                    ctx->d_coop = (CoopBox *)mailbox_acquire(ctx->device, std::max<int64_t>(n_own, 256), &got, !(ctx->tune.debug_flags & 0x100000));
                    if (ctx->d_coop) {
                        ctx->coop_cap = got;
                        have_boxes = true;
                    }  // (else: no such memory here, every workgroup works alone)
                }
                if (have_boxes && ctx->host_cfg.coop_ok) {
                    // (only what this launch touches: n_own mailboxes, the scan words, and the part-1 answers when there are two parts)
                    HIP_TRY(hipMemsetAsync(ctx->d_coop, 0, (size_t)n_own * sizeof(CoopBox), stream));
                    HIP_TRY(hipMemsetAsync(ctx->d_coop + ctx->coop_cap, 0, 3 * (size_t)(ctx->coop_cap + 64) * sizeof(uint32_t), stream));
                    CoopOut *out2 = (CoopOut *)((char *)(ctx->d_coop + ctx->coop_cap) + 3 * (size_t)(ctx->coop_cap + 64) * sizeof(uint32_t));
                    // (the array behind the scan words holds coop_cap >= 256 answer blocks: part-1 answers of the two-part claim mode, one per
                    //  owner; in the fan-out mode the answers of every (owner, part), owners * parts <= idle CUs < 256)
                    if (ctx->coop_fan) HIP_TRY(hipMemsetAsync(out2, 0, (size_t)std::min<int64_t>(n_own * ctx->coop_parts, ctx->coop_cap) * sizeof(CoopOut), stream));
                    else if (ctx->coop_parts == 2) HIP_TRY(hipMemsetAsync(out2, 0, (size_t)n_own * sizeof(CoopOut), stream));
                    bt.coop_out2 = (ctx->coop_parts == 2 || ctx->coop_fan) ? out2 : nullptr;
                    bt.coop_fan = ctx->coop_fan ? 1 : 0;
                    bt.coop_helpers = (int32_t)helpers; bt.coop_base = (int32_t)base; bt.coop_box = ctx->d_coop;
                    uint32_t *words = (uint32_t *)(ctx->d_coop + ctx->coop_cap);
                    bt.coop_posted = words; bt.coop_claimed = words + (ctx->coop_cap + 64); bt.coop_finished = words + 2 * (ctx->coop_cap + 64);
                    bt.coop_sets = (int32_t)((n_own + 15) / 16);
                    bt.coop_parts = ctx->coop_parts;
                    bt.coop_mute = (ctx->tune.coop_mute ? 1 : 0) | ((ctx->tune.debug_flags & 0x200000) ? 2 : 0);  // (bit 1: helpers fetch a job's inputs speculatively, before they know they won its claim)
                    ctx->last_coop_helpers = (int)helpers;
                }
            }
        }
    }
    if (!ctx->h_tab.empty() && ctx->host_cfg.harm_feed != 0 && !(ctx->host_cfg.flags & NYX_HIP_FLAG_STM)) {
        // workgroups that stream the table walk run streams: one per schedule, every range of a wave at the head of a sixteen-row
        // group (DevCfg.rs_*; debug_flags 0x800000: the common stream, as in round 4 - same bits)
        const bool want = (ctx->tune.debug_flags & 0x800000) == 0;
        if (want && ctx->rs_dirty) {
            for (int k = 0; k < 3; ++k) {
                const bool used = k == 0 ? (ctx->host_cfg.harm_feed & 1) != 0
                                  : (ctx->host_cfg.coop_ok != 0 && (k == 1 ? (ctx->host_cfg.harm_feed & 1) != 0 : (ctx->host_cfg.harm_feed & 2) != 0));
                ctx->host_cfg.rs_hyb[k] = 0;
                if (!used) continue;
                int64_t vec_off = 0;
                if (k == 0) build_run_stream(ctx, {DEV_SCHED_SOLO}, ctx->h_rs[k], vec_off, ctx->h_rs_cols[k]);
                else if (k == 1) build_run_stream(ctx, {DEV_SCHED_PRIMARY}, ctx->h_rs[k], vec_off, ctx->h_rs_cols[k]);
                else build_run_stream(ctx, {DEV_SCHED_HELPER, DEV_SCHED_HELPER2}, ctx->h_rs[k], vec_off, ctx->h_rs_cols[k]);
                if (ctx->h_rs[k].size() > ctx->rs_cap[k]) {
                    if (ctx->launched) HIP_TRY(hipEventSynchronize(ctx->ev_done));
                    (void)hipFree(ctx->d_rs[k]);
                    ctx->d_rs[k] = nullptr; ctx->rs_cap[k] = 0;
                    HIP_TRY(hipMalloc(&ctx->d_rs[k], ctx->h_rs[k].size() * sizeof(double)));
                    ctx->rs_cap[k] = ctx->h_rs[k].size();
                }
                if (!ctx->d_rs_cols[k]) HIP_TRY(hipMalloc(&ctx->d_rs_cols[k], ctx->h_rs_cols[k].size() * sizeof(ColHdr)));
                HIP_TRY(hipMemcpyAsync(ctx->d_rs[k], ctx->h_rs[k].data(), ctx->h_rs[k].size() * sizeof(double), hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(ctx->d_rs_cols[k], ctx->h_rs_cols[k].data(), ctx->h_rs_cols[k].size() * sizeof(ColHdr), hipMemcpyHostToDevice, stream));
                ctx->host_cfg.rs_hyb[k] = (uint64_t)ctx->d_rs[k];
                ctx->host_cfg.rs_hyb_v[k] = (uint64_t)(ctx->d_rs[k] + vec_off);
                ctx->host_cfg.rs_cols[k] = (uint64_t)ctx->d_rs_cols[k];
            }
            ctx->rs_dirty = false;
            HIP_TRY(hipMemcpyAsync(ctx->d_cfg, &ctx->host_cfg, sizeof(DevCfg), hipMemcpyHostToDevice, stream));
        } else if (!want && (ctx->host_cfg.rs_hyb[0] | ctx->host_cfg.rs_hyb[1] | ctx->host_cfg.rs_hyb[2]) != 0) {
            ctx->host_cfg.rs_hyb[0] = ctx->host_cfg.rs_hyb[1] = ctx->host_cfg.rs_hyb[2] = 0;
            HIP_TRY(hipMemcpyAsync(ctx->d_cfg, &ctx->host_cfg, sizeof(DevCfg), hipMemcpyHostToDevice, stream));
        }
    }
    {
        // the column weights of this launch's workgroup shape: measured once per context (see calibrate())
        const bool stm_k = (ctx->host_cfg.flags & NYX_HIP_FLAG_STM) != 0;
        const nyx_hip_ctx::WKey key(nw, (ctx->host_cfg.pipe && (!stm_k || ctx->sched_quad)) ? 1 : 0, ctx->sched_quad ? 1 : 0,
                                    bt.coop_helpers > 0 ? (int)(ctx->host_cfg.coop_frac * 10.0 + 0.5) : -1);
        ctx->last_key = key;
        const int64_t span = use_end ? INT64_MAX : (duration_ns < 0 ? -duration_ns : duration_ns);
        if (!calibrating && calibration_on(ctx) && ctx->host_cfg.has_grav && nw >= 8 && in->n >= 64 && !traj && !dur_ns && !ev &&
            span >= 100 * ctx->host_cfg.init_step_ns && !ctx->weights.count(key)) {
            if (int rc = calibrate(ctx, in, stream, !use_end && duration_ns < 0)) return rc;
            return launch(ctx, in, out, st, duration_ns, end_epoch_ns, use_end, stream, time_it, traj, dur_ns, ev, false, swapped);
        }
    }
    if (ctx->tune.profile || calibrating) {
        if (!ctx->d_prof) HIP_TRY(hipMalloc(&ctx->d_prof, 36 * 8 * sizeof(int64_t)));  // rows 0-15 owner workgroup 0, 16 mailbox counts, 17-32 the first helper workgroup
        HIP_TRY(hipMemsetAsync(ctx->d_prof, 0, 36 * 8 * sizeof(int64_t), stream));
        bt.prof = ctx->d_prof;
    }
    if (time_it) HIP_TRY(hipEventRecord(ctx->ev0, stream));
    const bool quad = pick_quad(ctx, in->n);
    // (the LDS staging of the ephemeris records was decided at ctx_create for the D3 layout; the quad layout is smaller)
    HIP_TRY(nyx_launch_propagate(bt, ctx->d_cfg, ctx->d_htab, ctx->d_cols, ctx->d_records, nw,
                                 ctx->host_cfg.rec_in_lds ? ctx->host_cfg.rec_doubles : 0, ctx->host_cfg.ed_reuse, stream, quad ? 1 : 0,
                                 (!ctx->host_cfg.has_grav && !ctx->host_cfg.has_drag && !ctx->host_cfg.has_tides && !ctx->host_cfg.has_grav2) ? 1 : 0));
    if (time_it) HIP_TRY(hipEventRecord(ctx->ev1, stream));
    HIP_TRY(hipEventRecord(ctx->ev_done, stream));
    ctx->launched = true;
    return NYX_HIP_RC_OK;
}

static int check_states(const nyx_hip_states_t *s, const char *what) {
    if (!s || s->n < 0 || !s->epoch_ns || !s->x_km || !s->y_km || !s->z_km || !s->vx_km_s || !s->vy_km_s || !s->vz_km_s) {
        nyx_set_error("%s: epoch and the six Cartesian arrays are mandatory", what);
        return NYX_HIP_RC_BAD_ARG;
    }
    return NYX_HIP_RC_OK;
}

extern "C" int32_t nyx_hip_propagate_batch_device(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                                  nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, void *hip_stream) {
    if (!ctx) { nyx_set_error("null ctx"); return NYX_HIP_RC_BAD_ARG; }
    if (int rc = check_states(in, "in")) return rc;
    if (int rc = check_states(out, "out")) return rc;
    if (in->n == 0) return NYX_HIP_RC_OK;
    CTX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = launch(ctx, in, out, stats, duration_ns, 0, 0, (hipStream_t)hip_stream, true);
    return rc;
}

extern "C" int32_t nyx_hip_propagate_batch_with_traj_device(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                                            nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, nyx_hip_traj_t *traj,
                                                            void *hip_stream) {
    if (!ctx || !traj) { nyx_set_error("null ctx / traj"); return NYX_HIP_RC_BAD_ARG; }
    if (int rc = check_states(in, "in")) return rc;
    if (int rc = check_states(out, "out")) return rc;
    if (in->n == 0) return NYX_HIP_RC_OK;
    CTX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    return launch(ctx, in, out, stats, duration_ns, 0, 0, (hipStream_t)hip_stream, true, traj);
}

// Device views of a staged host batch: `din` lives in ctx->in, `dout` / `dst` in ctx->out.
struct Staged {
    nyx_hip_states_t din, dout;
    nyx_hip_step_stats_t dst;
    bool stm = false;
};

// H2D of a host batch through the pinned mirror (one copy for the SoA block, one for the STMs).
// `upload_stm` false leaves ctx->in.stm allocated but unwritten (covariance mapping starts from identity).
static int stage_batch(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, const nyx_hip_states_t *out, Staged &sg, bool upload_stm = true) {
    const int64_t n = in->n;
    if (int rc = ensure_arrays(ctx->in, n, false)) return rc;
    if (int rc = ensure_arrays(ctx->out, n, true)) return rc;
    DevArrays &di = ctx->in, &dq = ctx->out;
    const double *hin[13] = {in->x_km, in->y_km, in->z_km, in->vx_km_s, in->vy_km_s, in->vz_km_s, in->cr, in->cd,
                             in->prop_mass_kg, in->dry_mass_kg, in->extra_mass_kg, in->srp_area_m2, in->drag_area_m2};
    std::memcpy(di.host(di.epoch), in->epoch_ns, n * sizeof(int64_t));
    for (int k = 0; k < 13; ++k)
        if (hin[k]) std::memcpy(di.host(di.f[k]), hin[k], n * sizeof(double));
    if (in->step_ns) std::memcpy(di.host(di.step), in->step_ns, n * sizeof(int64_t));
    HIP_TRY(hipMemcpy(di.dblock, di.hblock, (size_t)((char *)(di.f[12] + di.cap) - di.dblock), hipMemcpyHostToDevice));
    sg.stm = (ctx->host_cfg.flags & NYX_HIP_FLAG_STM) != 0;
    if (sg.stm) {
        if (upload_stm && (!in->stm || !out->stm)) { nyx_set_error("STM context: in->stm and out->stm are mandatory"); return NYX_HIP_RC_BAD_ARG; }
        for (DevArrays *d : {&di, &dq}) {
            if (d->stm_cap < n) {
                hipFree(d->stm);
                d->stm = nullptr;
                HIP_TRY(hipMalloc(&d->stm, (size_t)std::max<int64_t>(n, 1024) * 81 * sizeof(double)));
                d->stm_cap = std::max<int64_t>(n, 1024);
            }
        }
        if (upload_stm) HIP_TRY(hipMemcpy(di.stm, in->stm, (size_t)n * 81 * sizeof(double), hipMemcpyHostToDevice));
    }
    nyx_hip_states_t &din = sg.din;
    std::memset(&din, 0, sizeof din);
    din.n = n; din.epoch_ns = di.epoch;
    double **dinf[13] = {&din.x_km, &din.y_km, &din.z_km, &din.vx_km_s, &din.vy_km_s, &din.vz_km_s, &din.cr, &din.cd,
                         &din.prop_mass_kg, &din.dry_mass_kg, &din.extra_mass_kg, &din.srp_area_m2, &din.drag_area_m2};
    for (int k = 0; k < 13; ++k) *dinf[k] = hin[k] ? di.f[k] : nullptr;
    din.step_ns = in->step_ns ? di.step : nullptr;
    din.stm = sg.stm ? di.stm : nullptr;
    nyx_hip_states_t &dout = sg.dout;
    std::memset(&dout, 0, sizeof dout);
    dout.n = n; dout.epoch_ns = dq.epoch;
    double **doutf[13] = {&dout.x_km, &dout.y_km, &dout.z_km, &dout.vx_km_s, &dout.vy_km_s, &dout.vz_km_s, &dout.cr, &dout.cd,
                          &dout.prop_mass_kg, &dout.dry_mass_kg, &dout.extra_mass_kg, &dout.srp_area_m2, &dout.drag_area_m2};
    for (int k = 0; k < 13; ++k) *doutf[k] = dq.f[k];
    dout.step_ns = dq.step;
    dout.stm = sg.stm ? dq.stm : nullptr;
    sg.dst = {dq.status, dq.last_step, dq.last_error, dq.last_attempts, dq.n_acc, dq.n_rej, dq.n_evals};
    return NYX_HIP_RC_OK;
}

// D2H of ctx->out (states + stats) through the pinned mirror, unpacked into the caller's arrays.
static int fetch_batch(nyx_hip_ctx *ctx, int64_t n, bool stm, nyx_hip_states_t *out, nyx_hip_step_stats_t *stats) {
    DevArrays &dq = ctx->out;
    HIP_TRY(hipMemcpy(dq.hblock, dq.dblock, dq.bytes, hipMemcpyDeviceToHost));
    double *hout[13] = {out->x_km, out->y_km, out->z_km, out->vx_km_s, out->vy_km_s, out->vz_km_s, out->cr, out->cd,
                        out->prop_mass_kg, out->dry_mass_kg, out->extra_mass_kg, out->srp_area_m2, out->drag_area_m2};
    std::memcpy(out->epoch_ns, dq.host(dq.epoch), n * sizeof(int64_t));
    for (int k = 0; k < 13; ++k)
        if (hout[k]) std::memcpy(hout[k], dq.host(dq.f[k]), n * sizeof(double));
    if (out->step_ns) std::memcpy(out->step_ns, dq.host(dq.step), n * sizeof(int64_t));
    if (stm && out->stm) HIP_TRY(hipMemcpy(out->stm, dq.stm, (size_t)n * 81 * sizeof(double), hipMemcpyDeviceToHost));
    if (stats) {
        if (stats->status) std::memcpy(stats->status, dq.host(dq.status), n * sizeof(int32_t));
        if (stats->last_step_ns) std::memcpy(stats->last_step_ns, dq.host(dq.last_step), n * sizeof(int64_t));
        if (stats->last_error) std::memcpy(stats->last_error, dq.host(dq.last_error), n * sizeof(double));
        if (stats->last_attempts) std::memcpy(stats->last_attempts, dq.host(dq.last_attempts), n * sizeof(int32_t));
        if (stats->n_accepted) std::memcpy(stats->n_accepted, dq.host(dq.n_acc), n * sizeof(int64_t));
        if (stats->n_rejected) std::memcpy(stats->n_rejected, dq.host(dq.n_rej), n * sizeof(int64_t));
        if (stats->n_evals) std::memcpy(stats->n_evals, dq.host(dq.n_evals), n * sizeof(int64_t));
    }
    return NYX_HIP_RC_OK;
}

static int host_propagate(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns, int64_t end_epoch_ns, int use_end,
                          nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, nyx_hip_traj_t *traj = nullptr) {
    if (!ctx) { nyx_set_error("null ctx"); return NYX_HIP_RC_BAD_ARG; }
    if (int rc = check_states(in, "in")) return rc;
    if (int rc = check_states(out, "out")) return rc;
    const int64_t n = in->n;
    if (n == 0) return NYX_HIP_RC_OK;
    if (out->n < n) { nyx_set_error("out batch smaller than in batch"); return NYX_HIP_RC_BAD_ARG; }
    CTX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    const bool trace = (ctx->tune.debug_flags & 0x400) != 0;  // (tuning.debug_flags 0x400: milestones of the host path on stderr)
    if (trace) std::fprintf(stderr, "[nyx_hip] host_propagate n=%lld: staging\n", (long long)n);
    Staged sg;
    if (int rc = stage_batch(ctx, in, out, sg)) return rc;
    if (trace) std::fprintf(stderr, "[nyx_hip] staged, launching\n");

    nyx_hip_traj_t dtraj;
    std::memset(&dtraj, 0, sizeof dtraj);
    void *traj_block = nullptr;
    if (traj && traj->capacity > 0) {
        const size_t slots = (size_t)traj->capacity * (size_t)n;
        HIP_TRY(hipMalloc(&traj_block, slots * 7 * sizeof(double) + (size_t)n * sizeof(int32_t)));
        dtraj.capacity = traj->capacity;
        dtraj.epoch_ns = (int64_t *)traj_block;
        double *base = (double *)traj_block + slots;
        dtraj.x_km = base; dtraj.y_km = base + slots; dtraj.z_km = base + 2 * slots;
        dtraj.vx_km_s = base + 3 * slots; dtraj.vy_km_s = base + 4 * slots; dtraj.vz_km_s = base + 5 * slots;
        dtraj.len = (int32_t *)(base + 6 * slots);
    }
    if (int rc = launch(ctx, &sg.din, &sg.dout, &sg.dst, duration_ns, end_epoch_ns, use_end, nullptr, true, traj_block ? &dtraj : nullptr)) {
        if (traj_block) (void)hipFree(traj_block);
        return rc;
    }
    if (trace) std::fprintf(stderr, "[nyx_hip] launched (%d helpers, %d waves), synchronising\n", ctx->last_coop_helpers, ctx->host_cfg.n_waves);
    HIP_TRY(hipDeviceSynchronize());
    if (trace) std::fprintf(stderr, "[nyx_hip] kernel done\n");
    if (traj_block) {
        const size_t slots = (size_t)traj->capacity * (size_t)n;
        HIP_TRY(hipMemcpy(traj->len, dtraj.len, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(traj->epoch_ns, dtraj.epoch_ns, slots * sizeof(int64_t), hipMemcpyDeviceToHost));
        double *hdst[6] = {traj->x_km, traj->y_km, traj->z_km, traj->vx_km_s, traj->vy_km_s, traj->vz_km_s};
        double *dsrc[6] = {dtraj.x_km, dtraj.y_km, dtraj.z_km, dtraj.vx_km_s, dtraj.vy_km_s, dtraj.vz_km_s};
        for (int k = 0; k < 6; ++k) HIP_TRY(hipMemcpy(hdst[k], dsrc[k], slots * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(traj_block));
    }
    {
        float ms = 0.f;
        ctx->last_ms = (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ? ms : -1.0;
    }
    return fetch_batch(ctx, n, sg.stm, out, stats);
}

extern "C" int32_t nyx_hip_propagate_batch(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                           nyx_hip_states_t *out, nyx_hip_step_stats_t *stats) {
    return host_propagate(ctx, in, duration_ns, 0, 0, out, stats);
}

extern "C" int32_t nyx_hip_propagate_batch_with_traj(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                                     nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, nyx_hip_traj_t *traj) {
    if (!traj || traj->capacity < 1) { nyx_set_error("traj with capacity >= 1 required"); return NYX_HIP_RC_BAD_ARG; }
    return host_propagate(ctx, in, duration_ns, 0, 0, out, stats, traj);
}

extern "C" int32_t nyx_hip_propagate_until_epoch(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t end_epoch_ns,
                                                 nyx_hip_states_t *out, nyx_hip_step_stats_t *stats) {
    return host_propagate(ctx, in, 0, end_epoch_ns, 1, out, stats);
}

// ---------------------------------------------------------------------------------------------
// One batch over several contexts = several devices of one node, from ONE process (what replaces the rayon par_iter of
// mc/montecarlo.rs:233-253 when the host is not sharded by rank): contiguous index shards (shard k of m holds
// [k n / m, (k + 1) n / m), the rule of nyx_amd.shard_bounds), one host thread per context so that the staging copies
// and the kernels of the devices overlap, results written in place at the shard's offset - the "gather" is free because the
// arrays are structure-of-arrays.  With a trajectory the shards record into private buffers (the step-major layout has the
// batch size as its stride) and are scattered afterwards.  Trajectories are independent: no device-to-device traffic at all.
// ---------------------------------------------------------------------------------------------
static nyx_hip_states_t states_at(const nyx_hip_states_t &s, int64_t lo, int64_t n) {
    nyx_hip_states_t v = s;
    v.n = n;
    auto off = [&](auto *p) { return p ? p + lo : p; };
    v.epoch_ns = off(s.epoch_ns);
    v.x_km = off(s.x_km); v.y_km = off(s.y_km); v.z_km = off(s.z_km);
    v.vx_km_s = off(s.vx_km_s); v.vy_km_s = off(s.vy_km_s); v.vz_km_s = off(s.vz_km_s);
    v.cr = off(s.cr); v.cd = off(s.cd); v.prop_mass_kg = off(s.prop_mass_kg); v.dry_mass_kg = off(s.dry_mass_kg);
    v.extra_mass_kg = off(s.extra_mass_kg); v.srp_area_m2 = off(s.srp_area_m2); v.drag_area_m2 = off(s.drag_area_m2);
    v.stm = s.stm ? s.stm + lo * 81 : nullptr;
    v.step_ns = off(s.step_ns);
    return v;
}
static nyx_hip_step_stats_t stats_at(const nyx_hip_step_stats_t &s, int64_t lo) {
    nyx_hip_step_stats_t v = s;
    auto off = [&](auto *p) { return p ? p + lo : p; };
    v.status = off(s.status); v.last_step_ns = off(s.last_step_ns); v.last_error = off(s.last_error);
    v.last_attempts = off(s.last_attempts); v.n_accepted = off(s.n_accepted); v.n_rejected = off(s.n_rejected); v.n_evals = off(s.n_evals);
    return v;
}

extern "C" int32_t nyx_hip_propagate_batch_sharded(nyx_hip_ctx *const *ctxs, int32_t n_ctx, const nyx_hip_states_t *in, int64_t duration_ns,
                                                   nyx_hip_states_t *out, nyx_hip_step_stats_t *stats, nyx_hip_traj_t *traj) {
    if (!ctxs || n_ctx < 1 || !in || !out) { nyx_set_error("sharded: null argument or no context"); return NYX_HIP_RC_BAD_ARG; }
    for (int32_t k = 0; k < n_ctx; ++k)
        if (!ctxs[k]) { nyx_set_error("sharded: null context %d", k); return NYX_HIP_RC_BAD_ARG; }
    if (out->n != in->n) { nyx_set_error("sharded: out->n != in->n"); return NYX_HIP_RC_BAD_ARG; }
    if (traj && traj->capacity < 1) { nyx_set_error("traj with capacity >= 1 required"); return NYX_HIP_RC_BAD_ARG; }
    const int64_t n = in->n, cap = traj ? traj->capacity : 0;
    std::vector<int32_t> rcs((size_t)n_ctx, NYX_HIP_RC_OK);
    std::vector<std::string> errs((size_t)n_ctx);
    struct ShardTraj { std::vector<int64_t> ep; std::vector<double> f[6]; std::vector<int32_t> len; };
    std::vector<ShardTraj> st((size_t)n_ctx);
    std::vector<std::thread> threads;
    for (int32_t k = 0; k < n_ctx; ++k) {
        const int64_t lo = n * k / n_ctx, hi = n * (k + 1) / n_ctx;
        if (hi == lo) continue;
        threads.emplace_back([&, k, lo, hi]() {
            const nyx_hip_states_t vi = states_at(*in, lo, hi - lo);
            nyx_hip_states_t vo = states_at(*out, lo, hi - lo);
            nyx_hip_step_stats_t vs{};
            if (stats) vs = stats_at(*stats, lo);
            int32_t rc;
            if (traj) {
                ShardTraj &t = st[(size_t)k];
                const size_t cells = (size_t)((hi - lo) * cap);
                t.ep.resize(cells); t.len.assign((size_t)(hi - lo), 0);
                for (auto &f : t.f) f.resize(cells);
                nyx_hip_traj_t vt{cap, t.ep.data(), t.f[0].data(), t.f[1].data(), t.f[2].data(), t.f[3].data(), t.f[4].data(), t.f[5].data(), t.len.data()};
                rc = nyx_hip_propagate_batch_with_traj(ctxs[k], &vi, duration_ns, &vo, stats ? &vs : nullptr, &vt);
            } else {
                rc = nyx_hip_propagate_batch(ctxs[k], &vi, duration_ns, &vo, stats ? &vs : nullptr);
            }
            rcs[(size_t)k] = rc;
            if (rc != NYX_HIP_RC_OK) errs[(size_t)k] = nyx_hip_last_error();  // (the message is thread-local)
        });
    }
    for (auto &t : threads) t.join();
    for (int32_t k = 0; k < n_ctx; ++k)
        if (rcs[(size_t)k] != NYX_HIP_RC_OK) { nyx_set_error("shard %d: %s", k, errs[(size_t)k].c_str()); return rcs[(size_t)k]; }
    if (traj) {  // scatter the shards' step-major blocks (stride = shard size) into the batch's (stride = n)
        for (int32_t k = 0; k < n_ctx; ++k) {
            const int64_t lo = n * k / n_ctx, hi = n * (k + 1) / n_ctx, m = hi - lo;
            const ShardTraj &t = st[(size_t)k];
            for (int64_t i = 0; i < m; ++i) traj->len[lo + i] = t.len[(size_t)i];
            for (int64_t s2 = 0; s2 < cap; ++s2) {
                std::memcpy(traj->epoch_ns + s2 * n + lo, t.ep.data() + s2 * m, (size_t)m * sizeof(int64_t));
                double *dst[6] = {traj->x_km, traj->y_km, traj->z_km, traj->vx_km_s, traj->vy_km_s, traj->vz_km_s};
                for (int c = 0; c < 6; ++c) std::memcpy(dst[c] + s2 * n + lo, t.f[c].data() + s2 * m, (size_t)m * sizeof(double));
            }
        }
    }
    return NYX_HIP_RC_OK;
}

// ---------------------------------------------------------------------------------------------
// Traj evaluation (md/trajectory/traj.rs:82-162): traj_kernel.hip
// ---------------------------------------------------------------------------------------------
static int check_traj(const nyx_hip_traj_t *t, const char *what, bool need_epochs) {
    if (!t || t->capacity < 0 || !t->len || !t->x_km || !t->y_km || !t->z_km || !t->vx_km_s || !t->vy_km_s || !t->vz_km_s ||
        (need_epochs && !t->epoch_ns)) {
        nyx_set_error("%s: null array or negative capacity", what);
        return NYX_HIP_RC_BAD_ARG;
    }
    return NYX_HIP_RC_OK;
}

static int traj_eval_device(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, const int64_t *query, int64_t m,
                            int64_t step_ns, nyx_hip_traj_t *out, int32_t *status, int mode, hipStream_t stream) {
    if (!ctx) { nyx_set_error("null ctx"); return NYX_HIP_RC_BAD_ARG; }
    if (int rc = check_traj(traj, "traj", true)) return rc;
    if (int rc = check_traj(out, "out", true)) return rc;
    if (n < 0) { nyx_set_error("negative n"); return NYX_HIP_RC_BAD_ARG; }
    if (mode == TRAJ_MODE_AT) {
        if (m < 0 || (m > 0 && (!query || !status))) { nyx_set_error("traj_at: query/status arrays required"); return NYX_HIP_RC_BAD_ARG; }
        if (out->capacity < m) { nyx_set_error("traj_at: out->capacity < m"); return NYX_HIP_RC_BAD_ARG; }
    } else if (step_ns <= 0) {
        nyx_set_error("traj_every: step_ns must be > 0 (TimeSeries with a positive step)");
        return NYX_HIP_RC_BAD_ARG;
    }
    if (n == 0) return NYX_HIP_RC_OK;
    CTX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    TrajEvalArgs a;
    std::memset(&a, 0, sizeof a);
    a.src = *traj; a.dst = *out; a.n = n; a.query = query; a.m = m; a.step_ns = step_ns; a.status = status; a.mode = mode;
    if (ctx->launched) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_done, 0));
    HIP_TRY(hipEventRecord(ctx->ev0, stream));
    HIP_TRY(nyx_launch_traj_eval(&a, stream));
    HIP_TRY(hipEventRecord(ctx->ev1, stream));
    HIP_TRY(hipEventRecord(ctx->ev_done, stream));
    ctx->launched = true;
    return NYX_HIP_RC_OK;
}

extern "C" int32_t nyx_hip_traj_at_device(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, const int64_t *query_epoch_ns,
                                          int64_t m, nyx_hip_traj_t *out, int32_t *status, void *hip_stream) {
    return traj_eval_device(ctx, traj, n, query_epoch_ns, m, 0, out, status, TRAJ_MODE_AT, (hipStream_t)hip_stream);
}

extern "C" int32_t nyx_hip_traj_every_device(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, int64_t step_ns,
                                             nyx_hip_traj_t *out, void *hip_stream) {
    return traj_eval_device(ctx, traj, n, nullptr, 0, step_ns, out, nullptr, TRAJ_MODE_EVERY, (hipStream_t)hip_stream);
}

// ---------------------------------------------------------------------------------------------
// Ensemble moments of the final states (mc/results.rs:60-245 consumers): moments_kernel.hip
// ---------------------------------------------------------------------------------------------
static int moments_device(nyx_hip_ctx *ctx, const nyx_hip_states_t *s, const int32_t *status, const double *x0, double *out55, hipStream_t stream) {
    if (!ctx || !s || !out55 || s->n < 0 || (s->n > 0 && (!s->x_km || !s->y_km || !s->z_km || !s->vx_km_s || !s->vy_km_s || !s->vz_km_s))) {
        nyx_set_error("ensemble_moments: ctx, states (six Cartesian arrays) and out are mandatory");
        return NYX_HIP_RC_BAD_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->d_mom) HIP_TRY(hipMalloc((void **)&ctx->d_mom, (size_t)(MOM_BLOCKS + 1) * MOM_N * sizeof(double)));
    MomArgs a;
    std::memset(&a, 0, sizeof a);
    a.n = s->n;
    const double *f[9] = {s->x_km, s->y_km, s->z_km, s->vx_km_s, s->vy_km_s, s->vz_km_s, s->cr, s->cd, s->prop_mass_kg};
    for (int k = 0; k < 9; ++k) { a.f[k] = f[k]; a.x0[k] = x0 ? x0[k] : 0.0; }
    a.status = status;
    a.partial = ctx->d_mom;
    if (ctx->launched) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_done, 0));  // (the scratch is the context's: one reduction at a time)
    HIP_TRY(nyx_launch_moments(a, out55, stream));
    HIP_TRY(hipEventRecord(ctx->ev_done, stream));
    ctx->launched = true;
    return NYX_HIP_RC_OK;
}

extern "C" int32_t nyx_hip_ensemble_moments_device(nyx_hip_ctx *ctx, const nyx_hip_states_t *states, const int32_t *status, const double *x0,
                                                   double *out55, void *hip_stream) {
    if (!ctx) { nyx_set_error("null ctx"); return NYX_HIP_RC_BAD_ARG; }
    CTX_LOCK(ctx);
    return moments_device(ctx, states, status, x0, out55, (hipStream_t)hip_stream);
}

extern "C" int32_t nyx_hip_ensemble_moments(nyx_hip_ctx *ctx, const nyx_hip_states_t *states, const int32_t *status, const double *x0, double *out55) {
    if (!ctx || !states || !out55 || states->n < 0) { nyx_set_error("ensemble_moments: null argument"); return NYX_HIP_RC_BAD_ARG; }
    CTX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    const int64_t n = states->n;
    if (n > 0 && (!states->x_km || !states->y_km || !states->z_km || !states->vx_km_s || !states->vy_km_s || !states->vz_km_s)) {
        nyx_set_error("ensemble_moments: the six Cartesian arrays are mandatory");
        return NYX_HIP_RC_BAD_ARG;
    }
    // host arrays: one device block of 9 rows (+ the status words), copied row by row; the reduction itself is two small launches
    char *blk = nullptr;
    const size_t row = (size_t)std::max<int64_t>(n, 1) * sizeof(double);
    HIP_TRY(hipMalloc((void **)&blk, 9 * row + (size_t)std::max<int64_t>(n, 1) * sizeof(int32_t)));
    nyx_hip_states_t d;
    std::memset(&d, 0, sizeof d);
    d.n = n;
    const double *h[9] = {states->x_km, states->y_km, states->z_km, states->vx_km_s, states->vy_km_s, states->vz_km_s, states->cr, states->cd, states->prop_mass_kg};
    double **dp[9] = {&d.x_km, &d.y_km, &d.z_km, &d.vx_km_s, &d.vy_km_s, &d.vz_km_s, &d.cr, &d.cd, &d.prop_mass_kg};
    int rc = NYX_HIP_RC_OK;
    for (int k = 0; k < 9 && rc == NYX_HIP_RC_OK; ++k) {
        if (!h[k]) continue;
        *dp[k] = (double *)(blk + (size_t)k * row);
        if (n > 0 && hipMemcpy(*dp[k], h[k], (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = NYX_HIP_RC_HIP_ERROR;
    }
    int32_t *dst = nullptr;
    if (rc == NYX_HIP_RC_OK && status) {
        dst = (int32_t *)(blk + 9 * row);
        if (n > 0 && hipMemcpy(dst, status, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) rc = NYX_HIP_RC_HIP_ERROR;
    }
    if (rc == NYX_HIP_RC_OK) {
        if (!ctx->d_mom && hipMalloc((void **)&ctx->d_mom, (size_t)(MOM_BLOCKS + 1) * MOM_N * sizeof(double)) != hipSuccess) rc = NYX_HIP_RC_HIP_ERROR;
    }
    if (rc == NYX_HIP_RC_OK) rc = moments_device(ctx, &d, dst, x0, ctx->d_mom + (size_t)MOM_BLOCKS * MOM_N, nullptr);
    if (rc == NYX_HIP_RC_OK && hipMemcpy(out55, ctx->d_mom + (size_t)MOM_BLOCKS * MOM_N, MOM_N * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = NYX_HIP_RC_HIP_ERROR;
    if (rc == NYX_HIP_RC_HIP_ERROR) nyx_set_error("ensemble_moments: HIP error: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(blk);
    return rc;
}

// A nyx_hip_traj_t whose arrays are one device allocation (RAII).
struct DevTraj {
    nyx_hip_traj_t t;
    void *block = nullptr;
    size_t slots = 0;
    int64_t n = 0;
    ~DevTraj() { if (block) (void)hipFree(block); }
    int alloc(int64_t capacity, int64_t n_) {
        n = n_;
        slots = (size_t)capacity * (size_t)n;
        std::memset(&t, 0, sizeof t);
        if (hipMalloc(&block, std::max<size_t>(slots, 1) * 7 * sizeof(double) + (size_t)n * sizeof(int32_t)) != hipSuccess) {
            block = nullptr;
            nyx_set_error("hipMalloc of the trajectory staging block failed");
            return NYX_HIP_RC_HIP_ERROR;
        }
        t.capacity = capacity;
        t.epoch_ns = (int64_t *)block;
        double *base = (double *)block + std::max<size_t>(slots, 1);
        t.x_km = base; t.y_km = base + slots; t.z_km = base + 2 * slots;
        t.vx_km_s = base + 3 * slots; t.vy_km_s = base + 4 * slots; t.vz_km_s = base + 5 * slots;
        t.len = (int32_t *)(base + 6 * std::max<size_t>(slots, 1));
        return NYX_HIP_RC_OK;
    }
    int upload(const nyx_hip_traj_t *h) {
        HIP_TRY(hipMemcpy(t.len, h->len, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!slots) return NYX_HIP_RC_OK;
        HIP_TRY(hipMemcpy(t.epoch_ns, h->epoch_ns, slots * sizeof(int64_t), hipMemcpyHostToDevice));
        const double *hsrc[6] = {h->x_km, h->y_km, h->z_km, h->vx_km_s, h->vy_km_s, h->vz_km_s};
        double *ddst[6] = {t.x_km, t.y_km, t.z_km, t.vx_km_s, t.vy_km_s, t.vz_km_s};
        for (int k = 0; k < 6; ++k) HIP_TRY(hipMemcpy(ddst[k], hsrc[k], slots * sizeof(double), hipMemcpyHostToDevice));
        return NYX_HIP_RC_OK;
    }
    int download(nyx_hip_traj_t *h) const {
        HIP_TRY(hipMemcpy(h->len, t.len, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (!slots) return NYX_HIP_RC_OK;
        HIP_TRY(hipMemcpy(h->epoch_ns, t.epoch_ns, slots * sizeof(int64_t), hipMemcpyDeviceToHost));
        double *hdst[6] = {h->x_km, h->y_km, h->z_km, h->vx_km_s, h->vy_km_s, h->vz_km_s};
        const double *dsrc[6] = {t.x_km, t.y_km, t.z_km, t.vx_km_s, t.vy_km_s, t.vz_km_s};
        for (int k = 0; k < 6; ++k) HIP_TRY(hipMemcpy(hdst[k], dsrc[k], slots * sizeof(double), hipMemcpyDeviceToHost));
        return NYX_HIP_RC_OK;
    }
};

static int traj_eval_host(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, const int64_t *query, int64_t m,
                          int64_t step_ns, nyx_hip_traj_t *out, int32_t *status, int mode) {
    if (!ctx) { nyx_set_error("null ctx"); return NYX_HIP_RC_BAD_ARG; }
    if (int rc = check_traj(traj, "traj", true)) return rc;
    if (int rc = check_traj(out, "out", true)) return rc;
    if (n <= 0) return n == 0 ? NYX_HIP_RC_OK : NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    DevTraj src, dst;
    if (int rc = src.alloc(traj->capacity, n)) return rc;
    if (int rc = dst.alloc(out->capacity, n)) return rc;
    if (int rc = src.upload(traj)) return rc;
    int64_t *d_query = nullptr;
    int32_t *d_status = nullptr;
    int rc = NYX_HIP_RC_OK;
    if (mode == TRAJ_MODE_AT && m > 0) {
        if (!query || !status) { nyx_set_error("traj_at: query/status arrays required"); return NYX_HIP_RC_BAD_ARG; }
        if (hipMalloc(&d_query, (size_t)m * sizeof(int64_t)) != hipSuccess ||
            hipMalloc(&d_status, (size_t)m * (size_t)n * sizeof(int32_t)) != hipSuccess ||
            hipMemcpy(d_query, query, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) {
            nyx_set_error("traj_at: staging of the query epochs failed");
            rc = NYX_HIP_RC_HIP_ERROR;
        }
    }
    if (!rc) rc = traj_eval_device(ctx, &src.t, n, d_query, m, step_ns, &dst.t, d_status, mode, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) { nyx_set_error("trajectory evaluation kernel failed"); rc = NYX_HIP_RC_HIP_ERROR; }
    if (!rc) {
        float ms = 0.f;
        ctx->last_ms = (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ? ms : -1.0;
        rc = dst.download(out);
    }
    if (!rc && d_status && hipMemcpy(status, d_status, (size_t)m * (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
        nyx_set_error("traj_at: D2H of the statuses failed");
        rc = NYX_HIP_RC_HIP_ERROR;
    }
    (void)hipFree(d_query);
    (void)hipFree(d_status);
    return rc;
}

extern "C" int32_t nyx_hip_traj_at(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, const int64_t *query_epoch_ns, int64_t m,
                                   nyx_hip_traj_t *out, int32_t *status) {
    return traj_eval_host(ctx, traj, n, query_epoch_ns, m, 0, out, status, TRAJ_MODE_AT);
}

extern "C" int32_t nyx_hip_traj_every(nyx_hip_ctx *ctx, const nyx_hip_traj_t *traj, int64_t n, int64_t step_ns, nyx_hip_traj_t *out) {
    return traj_eval_host(ctx, traj, n, nullptr, 0, step_ns, out, nullptr, TRAJ_MODE_EVERY);
}

struct DevBuf {  // RAII device allocation
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (hipMalloc(&p, std::max<size_t>(bytes, 8)) != hipSuccess) {
            p = nullptr;
            nyx_set_error("hipMalloc of %zu bytes failed", bytes);
            return NYX_HIP_RC_HIP_ERROR;
        }
        return NYX_HIP_RC_OK;
    }
    template <typename T> T *as() const { return (T *)p; }
};

// ---------------------------------------------------------------------------------------------
// Stop conditions (propagators/event.rs:88-211): propagation with the crossing counter, then the root search
// ---------------------------------------------------------------------------------------------
extern "C" int32_t nyx_hip_propagate_until_event(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, int64_t max_duration_ns,
                                                 const nyx_hip_event_t *event, nyx_hip_states_t *out, nyx_hip_step_stats_t *stats,
                                                 nyx_hip_traj_t *traj, int32_t *crossings) {
    if (!ctx || !event) { nyx_set_error("until_event: ctx and event are mandatory"); return NYX_HIP_RC_BAD_ARG; }
    if (event->has_frame && (event->frame.kind != NYX_HIP_ROT_IAU || event->frame.n_nut_prec < 0 || event->frame.n_nut_prec > NYX_HIP_MAX_NUT_PREC)) {
        nyx_set_error("until_event: the event frame must be an IAU-oriented frame (NYX_HIP_ROT_IAU)");
        return NYX_HIP_RC_UNSUPPORTED;
    }
    if ((event->scalar == NYX_HIP_EV_LATITUDE_DEG || event->scalar == NYX_HIP_EV_HEIGHT_KM) &&
        !(event->frame_eq_radius_km > 0.0 && event->frame_flattening >= 0.0 && event->frame_flattening < 1.0)) {
        nyx_set_error("until_event: geodetic scalars need the frame's ellipsoid (frame_eq_radius_km > 0, 0 <= flattening < 1)");
        return NYX_HIP_RC_BAD_ARG;
    }
    if (event->scalar < NYX_HIP_EV_TRUE_ANOMALY_DEG || event->scalar > NYX_HIP_EV_HEIGHT_KM || event->trigger < 1 ||
        event->epoch_precision_ns < 0 || !(event->value_precision >= 0.0)) {
        nyx_set_error("until_event: bad event (scalar, trigger >= 1, precisions >= 0)");
        return NYX_HIP_RC_BAD_ARG;
    }
    if (int rc = check_traj(traj, "traj", true)) return rc;
    if (traj->capacity < 2) { nyx_set_error("until_event: traj->capacity >= 2 required (the search needs the bracket)"); return NYX_HIP_RC_BAD_ARG; }
    if (int rc = check_states(in, "in")) return rc;
    if (int rc = check_states(out, "out")) return rc;
    const int64_t n = in->n;
    if (n == 0) return NYX_HIP_RC_OK;
    if (out->n < n) { nyx_set_error("out batch smaller than in batch"); return NYX_HIP_RC_BAD_ARG; }
    CTX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    Staged sg;
    if (int rc = stage_batch(ctx, in, out, sg)) return rc;
    DevTraj dtraj;
    if (int rc = dtraj.alloc(traj->capacity, n)) return rc;
    DevBuf evbuf, evdesc;  // prev (f64), count, found (i32); the event descriptor itself
    if (int rc = evbuf.alloc((size_t)n * 16)) return rc;
    if (int rc = evdesc.alloc(sizeof(nyx_hip_event_t))) return rc;
    HIP_TRY(hipMemcpy(evdesc.p, event, sizeof(nyx_hip_event_t), hipMemcpyHostToDevice));
    DevBatch ev;
    std::memset(&ev, 0, sizeof ev);
    ev.ev = evdesc.as<nyx_hip_event_t>();
    ev.ev_mu = ctx->host_cfg.mu_central;
    ev.ev_prev = evbuf.as<double>();
    ev.ev_count = (int32_t *)(evbuf.as<double>() + n);
    ev.ev_found = ev.ev_count + n;
    HIP_TRY(hipMemset(evbuf.p, 0, (size_t)n * 16));
    if (int rc = launch(ctx, &sg.din, &sg.dout, &sg.dst, max_duration_ns, 0, 0, nullptr, true, &dtraj.t, nullptr, &ev)) return rc;
    EventSearchArgs a;
    std::memset(&a, 0, sizeof a);
    a.traj = dtraj.t; a.n = n; a.ev = *event; a.mu = ctx->host_cfg.mu_central;
    a.found = ev.ev_found; a.status = sg.dst.status; a.epoch_ns = sg.dout.epoch_ns;
    double *st6[6] = {sg.dout.x_km, sg.dout.y_km, sg.dout.z_km, sg.dout.vx_km_s, sg.dout.vy_km_s, sg.dout.vz_km_s};
    for (int c = 0; c < 6; ++c) a.state[c] = st6[c];
    HIP_TRY(nyx_launch_event_search(&a, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    {
        float ms = 0.f;
        ctx->last_ms = (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ? ms : -1.0;
    }
    if (int rc = fetch_batch(ctx, n, sg.stm, out, stats)) return rc;
    if (int rc = dtraj.download(traj)) return rc;
    if (crossings) HIP_TRY(hipMemcpy(crossings, ev.ev_count, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return NYX_HIP_RC_OK;
}

// ---------------------------------------------------------------------------------------------
// Covariance mapping (od/process/mod.rs:440-486): segment launches + predict_kernel.hip, one stream, no host round trip
// ---------------------------------------------------------------------------------------------

extern "C" int32_t nyx_hip_predict_until(nyx_hip_ctx *ctx, const nyx_hip_states_t *in, const nyx_hip_predict_t *cfg,
                                         nyx_hip_estimates_t *est, nyx_hip_states_t *out, nyx_hip_step_stats_t *stats,
                                         nyx_hip_predict_history_t *hist) {
    if (!ctx || !cfg || !est || !est->covar) { nyx_set_error("predict: ctx, cfg and est->covar are mandatory"); return NYX_HIP_RC_BAD_ARG; }
    if (!(ctx->host_cfg.flags & NYX_HIP_FLAG_STM)) { nyx_set_error("predict: the context must be created with NYX_HIP_FLAG_STM"); return NYX_HIP_RC_BAD_ARG; }
    if (cfg->max_step_ns <= 0) { nyx_set_error("predict: max_step_ns must be > 0"); return NYX_HIP_RC_BAD_ARG; }
    if (cfg->n_process_noise < 0 || cfg->n_process_noise > NYX_HIP_MAX_PROCESS_NOISE) { nyx_set_error("predict: n_process_noise out of range"); return NYX_HIP_RC_BAD_ARG; }
    for (int q = 0; q < cfg->n_process_noise; ++q)
        if (cfg->process_noise[q].local_frame < NYX_HIP_FRAME_INERTIAL || cfg->process_noise[q].local_frame > NYX_HIP_FRAME_VNC) {
            nyx_set_error("predict: process noise %d: local frame not on the device path (inertial, RIC, VNC)", q);
            return NYX_HIP_RC_UNSUPPORTED;
        }
    if (hist && (hist->capacity < 0 || !hist->n_updates)) { nyx_set_error("predict: hist->n_updates is mandatory, capacity >= 0"); return NYX_HIP_RC_BAD_ARG; }
    if (int rc = check_states(in, "in")) return rc;
    if (int rc = check_states(out, "out")) return rc;
    const int64_t n = in->n;
    if (n == 0) return NYX_HIP_RC_OK;
    if (out->n < n) { nyx_set_error("out batch smaller than in batch"); return NYX_HIP_RC_BAD_ARG; }
    CTX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    Staged sg;
    if (int rc = stage_batch(ctx, in, out, sg, /*upload_stm=*/false)) return rc;
    // segments needed: the longest trajectory decides (the others idle with duration 0)
    int64_t n_seg = 1;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t span = cfg->end_epoch_ns - in->epoch_ns[i];
        if (span > 0) n_seg = std::max(n_seg, (span + cfg->max_step_ns - 1) / cfg->max_step_ns);
    }
    const int64_t cap = hist ? hist->capacity : 0;
    const size_t slots = (size_t)cap * (size_t)n;
    DevBuf covar, sdev, work, h_epoch, h_state, h_stm, h_covar, h_sdev;
    // work: prev_epoch, dur, acc x3, init_epoch (int64) then status, n_updates (int32)
    if (int rc = covar.alloc((size_t)n * 81 * 8)) return rc;
    if (int rc = sdev.alloc((size_t)n * 9 * 8)) return rc;
    if (int rc = work.alloc((size_t)n * (6 * 8 + 2 * 4))) return rc;
    HIP_TRY(hipMemcpy(covar.p, est->covar, (size_t)n * 81 * 8, hipMemcpyHostToDevice));
    if (est->state_dev) HIP_TRY(hipMemcpy(sdev.p, est->state_dev, (size_t)n * 9 * 8, hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(sdev.p, 0, (size_t)n * 9 * 8));
    PredictArgs a;
    std::memset(&a, 0, sizeof a);
    a.n = n; a.cfg = *cfg;
    a.epoch = sg.dout.epoch_ns;
    const double *s9[9] = {sg.dout.x_km, sg.dout.y_km, sg.dout.z_km, sg.dout.vx_km_s, sg.dout.vy_km_s, sg.dout.vz_km_s,
                           sg.dout.cr, sg.dout.cd, sg.dout.prop_mass_kg};
    for (int k = 0; k < 9; ++k) a.s9[k] = s9[k];
    a.seg_status = sg.dst.status; a.seg_n_acc = sg.dst.n_accepted; a.seg_n_rej = sg.dst.n_rejected; a.seg_n_evals = sg.dst.n_evals;
    a.covar = covar.as<double>(); a.state_dev = sdev.as<double>();
    int64_t *w64 = work.as<int64_t>();
    a.prev_epoch = w64; a.dur = w64 + n; a.acc_n_acc = w64 + 2 * n; a.acc_n_rej = w64 + 3 * n; a.acc_n_evals = w64 + 4 * n;
    a.init_epoch = w64 + 5 * n;
    a.status = (int32_t *)(w64 + 6 * n); a.hist.n_updates = a.status + n;
    a.hist.capacity = cap;
    if (hist && slots) {
        if (hist->epoch_ns) { if (int rc = h_epoch.alloc(slots * 8)) return rc; a.hist.epoch_ns = h_epoch.as<int64_t>(); }
        if (hist->state) { if (int rc = h_state.alloc(slots * 9 * 8)) return rc; a.hist.state = h_state.as<double>(); }
        if (hist->stm) { if (int rc = h_stm.alloc(slots * 81 * 8)) return rc; a.hist.stm = h_stm.as<double>(); }
        if (hist->covar) { if (int rc = h_covar.alloc(slots * 81 * 8)) return rc; a.hist.covar = h_covar.as<double>(); }
        if (hist->state_dev) { if (int rc = h_sdev.alloc(slots * 9 * 8)) return rc; a.hist.state_dev = h_sdev.as<double>(); }
    }
    hipStream_t stream = nullptr;
    HIP_TRY(hipEventRecord(ctx->ev0, stream));
    // segment 0 reads the caller's states (ctx->in) with an identity STM and writes ctx->out; later segments run in place
    a.stm = sg.din.stm;
    HIP_TRY(nyx_launch_predict_init(&a, sg.din.epoch_ns, stream));
    {
        // the segment launches carry a per-trajectory duration array and are too short to calibrate on themselves: measure
        // the column weights of their workgroup shape once per context, on the staged states (identity STM set above)
        const int nw_c = pick_waves(ctx, n);
        const bool quad_c = pick_quad(ctx, n);
        const bool pipe_c = quad_c && nw_c == DEV_MAX_WAVES && ctx->tune.pipelined != 0;
        const nyx_hip_ctx::WKey key(nw_c, pipe_c ? 1 : 0, quad_c ? 1 : 0, -1);
        if (calibration_on(ctx) && ctx->host_cfg.has_grav && nw_c >= 8 && n >= 16 && !ctx->weights.count(key)) {
            if (int rc = calibrate(ctx, &sg.din, stream)) return rc;
            HIP_TRY(hipEventRecord(ctx->ev0, stream));  // (the timed region is the segment loop, not the one-off calibration)
        }
    }
    a.stm = sg.dout.stm;
    // Round 6: the whole loop in ONE launch - the workgroups stay resident, the integrator wave performs the time updates of its
    // trajectories at every segment boundary (propagate_kernel.hip, segment_update; DevBatch.pred = a device copy of `a`).  A segment
    // launch cost ~30 us beyond its force evaluations (tools/seg_cost.py), a fifth of BASELINE config 4's loop.  The launch-per-segment
    // loop of rounds 2-5 stays for the integration-frame swap (translated in and out per segment, od/process/mod.rs:453-468) and as
    // the A/B reference (debug_flags 0x20000000): same states, STMs and covariances.
    DevBuf pa_dev;
    // (quad layout only: sixteen waves share sixteen trajectories' updates; the 64-lane layout - four waves, sixty-four trajectories per
    //  workgroup - is the large-ensemble shape, where the per-launch cost is a small share and sixteen serial updates per wave cost more:
    //  measured 26.2 ms fused against 24.3 ms per segment at n = 1 000)
    const bool fused = ctx->swap_n_chain == 0 && !(ctx->tune.debug_flags & 0x20000000) && pick_quad(ctx, n);
    if (fused) {
        if (int rc = pa_dev.alloc(sizeof(PredictArgs))) return rc;
        HIP_TRY(hipMemcpyAsync(pa_dev.p, &a, sizeof(PredictArgs), hipMemcpyHostToDevice, stream));
        ctx->fused_pred = (const PredictArgs *)pa_dev.p;
        const int rc = launch(ctx, &sg.din, &sg.dout, &sg.dst, 0, 0, 0, stream, false, nullptr, a.dur);
        ctx->fused_pred = nullptr;
        if (rc) return rc;
        // the kernel's counters run over the whole loop
        HIP_TRY(hipMemcpyAsync(a.acc_n_acc, sg.dst.n_accepted, (size_t)n * 8, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemcpyAsync(a.acc_n_rej, sg.dst.n_rejected, (size_t)n * 8, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemcpyAsync(a.acc_n_evals, sg.dst.n_evals, (size_t)n * 8, hipMemcpyDeviceToDevice, stream));
    } else
    for (int64_t s = 0; s < n_seg; ++s) {
        const nyx_hip_states_t *src = s == 0 ? &sg.din : &sg.dout;
        if (int rc = launch(ctx, src, &sg.dout, &sg.dst, 0, 0, 0, stream, false, nullptr, a.dur)) return rc;
        HIP_TRY(nyx_launch_time_update(&a, stream));
    }
    HIP_TRY(hipEventRecord(ctx->ev1, stream));
    HIP_TRY(hipDeviceSynchronize());
    {
        float ms = 0.f;
        ctx->last_ms = (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ? ms : -1.0;
    }
    if (int rc = fetch_batch(ctx, n, true, out, stats)) return rc;
    HIP_TRY(hipMemcpy(est->covar, covar.p, (size_t)n * 81 * 8, hipMemcpyDeviceToHost));
    if (est->state_dev) HIP_TRY(hipMemcpy(est->state_dev, sdev.p, (size_t)n * 9 * 8, hipMemcpyDeviceToHost));
    if (stats) {  // first failing status, counters summed over the segments
        if (stats->status) HIP_TRY(hipMemcpy(stats->status, a.status, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (stats->n_accepted) HIP_TRY(hipMemcpy(stats->n_accepted, a.acc_n_acc, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (stats->n_rejected) HIP_TRY(hipMemcpy(stats->n_rejected, a.acc_n_rej, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (stats->n_evals) HIP_TRY(hipMemcpy(stats->n_evals, a.acc_n_evals, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    if (hist) {
        HIP_TRY(hipMemcpy(hist->n_updates, a.hist.n_updates, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (a.hist.epoch_ns) HIP_TRY(hipMemcpy(hist->epoch_ns, a.hist.epoch_ns, slots * 8, hipMemcpyDeviceToHost));
        if (a.hist.state) HIP_TRY(hipMemcpy(hist->state, a.hist.state, slots * 9 * 8, hipMemcpyDeviceToHost));
        if (a.hist.stm) HIP_TRY(hipMemcpy(hist->stm, a.hist.stm, slots * 81 * 8, hipMemcpyDeviceToHost));
        if (a.hist.covar) HIP_TRY(hipMemcpy(hist->covar, a.hist.covar, slots * 81 * 8, hipMemcpyDeviceToHost));
        if (a.hist.state_dev) HIP_TRY(hipMemcpy(hist->state_dev, a.hist.state_dev, slots * 9 * 8, hipMemcpyDeviceToHost));
    }
    return NYX_HIP_RC_OK;
}

// Introspection for tests / DESIGN.md: column schedule of the current context.
extern "C" int32_t nyx_hip_debug_schedule(nyx_hip_ctx *ctx, int32_t n_waves, int32_t *loads /* [8] */) {
    if (!ctx || n_waves < 1 || n_waves > DEV_MAX_WAVES) return NYX_HIP_RC_BAD_ARG;
    CTX_LOCK(ctx);
    const int keep = ctx->host_cfg.n_waves;
    build_schedule(ctx, n_waves);
    for (int w = 0; w < DEV_MAX_WAVES; ++w) {
        int l = 0;
        const DevSched &sd = ctx->host_cfg.sched[DEV_SCHED_SOLO];
        for (int q = 0; q < sd.n_ranges[w]; ++q)
            for (int c = sd.range_c0[w][q]; c < sd.range_c0[w][q] + sd.range_cnt[w][q]; ++c) l += ctx->col_len[c];
        loads[w] = l;
    }
    build_schedule(ctx, keep);
    return NYX_HIP_RC_OK;
}
