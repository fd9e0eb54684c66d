// pk_cooperative.h - part of propagate_kernel.hip (included there, in this order; not a stand-alone header): cooperative mode: tagged-granule mailboxes, post / wait / fallback of an owner, the helper workgroup (claimed jobs; fan-out mode).
// ---------------------------------------------------------------------------------------------
// Cooperative mode: idle CUs lend a hand.
//
// A workgroup holds 64 trajectories and fills one CU; an ensemble of 10 000 therefore occupies 157 of the 256 CUs.
// When workgroups are fewer than CUs, the launch adds HELPER workgroups on the idle CUs.  For every force evaluation
// the trajectory-owning workgroup posts the five per-lane inputs of the column recursion (2.5 KB) in a mailbox in
// global memory, keeps the columns of DEV_SCHED_PRIMARY for itself, and its helper evaluates the columns of
// DEV_SCHED_HELPER for the same 64 lanes and answers with four partial sums per lane (2 KB).  The exchange overlaps
// the owner's own window; in the pipelined stage loop the job of stage i+1 is posted inside the window of stage i
// (mailbox halves by the parity of the sequence number: an owner has up to two jobs outstanding, claimed in order).
// Deadlock-free without any residency assumption: the owner waits a bounded time for an answer, and if none comes it
// evaluates the helper's columns itself (walking DEV_SCHED_HELPER) and goes back to DEV_SCHED_SOLO for the rest of the
// launch; helpers leave when every workgroup they serve has finished.  The owner adds the helper's partial after its
// own sixteen, in a fixed order: results are deterministic for a given split.
// ---------------------------------------------------------------------------------------------
// The mailboxes live in UNCACHED device memory and are only touched with device-scope relaxed atomics (loads and stores
// that go past the L1 / the XCD's L2), ordered by workgroup-scope fences, i.e. s_waitcnt on the wave's own accesses: no
// cache write-back or invalidate anywhere (a device-scope fence per evaluation also throws the harmonics table out of
// L2 and doubled the run time).
// (through GLOBAL-qualified pointers: a generic pointer makes these flat_load / flat_store, which count on lgkmcnt as well as on vmcnt -
//  every LDS wait behind a post then also waited for the stores' round trip to uncached memory)
#define GAS __attribute__((address_space(1)))
DEVFN uint32_t coop_load(const uint32_t *p) { return __hip_atomic_load((const GAS uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN void coop_store(uint32_t *p, uint32_t v) { __hip_atomic_store((GAS uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN void coop_release() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }  // (inline asm: never elided by the compiler)
DEVFN void coop_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#define COOP_TIMEOUT_TICKS 200000LL  /* 2 ms of the 100 MHz realtime counter */
#define COOP_SET 16                  /* owners per set */

DEVFN uint64_t coop_loadu(const uint64_t *p) { return __hip_atomic_load((const GAS uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN void coop_storeu(uint64_t *p, uint64_t v) { __hip_atomic_store((GAS uint64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a double as two tagged granules (CoopBox): g[0] = {low half | seq << 32}, g[DEV_LANES] = {high half | seq << 32}
DEVFN void coop_put(uint64_t *g, double v, uint32_t seq) {
    const uint64_t b = (uint64_t)__double_as_longlong(v), t = (uint64_t)seq << 32;
    coop_storeu(g, (b & 0xffffffffull) | t);
    coop_storeu(g + DEV_LANES, (b >> 32) | t);
}
DEVFN bool coop_get(const uint64_t *g, uint32_t seq, double &v) {
    const uint64_t lo = coop_loadu(g), hi = coop_loadu(g + DEV_LANES);
    v = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
    return (uint32_t)(lo >> 32) == seq && (uint32_t)(hi >> 32) == seq;
}

// Posting happens from the LDS copy of the inputs, when the integrator wave has nothing else to do (start of the
// window in the plain loop, right after the next stage's inputs are formed in the pipelined one).  The inputs are tagged
// granules: the sequence number is written right behind them, with no wait in between - a helper that sees it before the
// data simply polls the granules until their tags agree.
// `parts` sub-jobs per evaluation (1, or 2: the helpers' columns in two halves, claimed by two helper workgroups): the words the
// helpers scan count SUB-JOBS - posted = parts * seq; sub-job c (1, 2, ...) is part (c - 1) % parts of evaluation (c + parts - 1) / parts.
// The single-part functions are kept exactly as small as they were before the two-part hand-off existed, and the two-part ones are
// their own functions behind a uniform branch at the call site: measured on the north-star run (8 h of propagation), folding both into
// one function with a run-time part count cost 2.8 % - the integrator's role code is register-allocated around these calls.
// (round 5: the five rows are read from LDS through an LDS-qualified pointer and all at once, THEN stored.  Through the generic pointer
//  of rounds 1-4 every row was a flat_load behind `s_waitcnt vmcnt(0) lgkmcnt(0)`, i.e. behind the previous row's stores to uncached
//  memory: five serial round trips, 4.7 k cycles of the integrator's window per evaluation.)
#ifndef COOP_INLINE
#define COOP_INLINE 0
#endif
#if COOP_INLINE
#define COOP_FN static __device__ __forceinline__
#else
#define COOP_FN static __device__ __attribute__((noinline))
#endif
// (`mult`: what the scan words count - sub-jobs: 1 per evaluation, or 2 with the two-part hand-off)
DEVFN void coop_post_inl(CoopBox *box, uint32_t *posted, int lane, uint32_t seq, LdsCPtr inb, uint32_t mult) {
    double v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = inb[q * DEV_LANES + lane];
#pragma unroll
    for (int q = 0; q < 5; ++q) coop_put(&box->in[seq & 1u][q][0][lane], v[q], seq);
    if (lane == 0) coop_store(posted, mult * seq);
}
COOP_FN void coop_post(CoopBox *box, uint32_t *posted, int lane, uint32_t seq, LdsCPtr inb) {
    double v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = inb[q * DEV_LANES + lane];
#pragma unroll
    for (int q = 0; q < 5; ++q) coop_put(&box->in[seq & 1u][q][0][lane], v[q], seq);
    if (lane == 0) coop_store(posted, seq);
}
static __device__ __attribute__((noinline)) void coop_post2(CoopBox *box, uint32_t *posted, int lane, uint32_t seq, LdsCPtr inb) {
    double v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = inb[q * DEV_LANES + lane];
#pragma unroll
    for (int q = 0; q < 5; ++q) coop_put(&box->in[seq & 1u][q][0][lane], v[q], seq);
    if (lane == 0) coop_store(posted, 2u * seq);  // (the scan words count SUB-JOBS)
}

struct CoopAnswer {
    double x, y, z, w;
    int ok;
};
// The answer needs no flag: every lane polls the LAST granule the helper writes for it, and when all of them carry this
// evaluation's tag the other seven are read and checked the same way (they were stored earlier, but nothing orders them).
DEVFN CoopAnswer coop_wait_inl(CoopBox *box, int lane, uint32_t seq) {
    CoopAnswer a = {0.0, 0.0, 0.0, 0.0, 0};
    const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
    const unsigned par = seq & 1u;
    // Round 6: the answer is read OPTIMISTICALLY first - all eight granules of every lane in one batch of loads, one round trip.  With the
    // late collection (phase C, behind the stage barrier) the answer is almost always in the mailbox by the time the integrator asks
    // (in-kernel accounting, round 5: "wait for the answer" 3.6 k cycles = exactly the three SERIAL uncached loads this function used to
    // make - poll lane 0's last granule, every lane's last granule, then the eight - with nothing to wait for), and this wave's
    // chain answer -> next post is what bounds a cooperative owner's period.  Only when the optimistic read misses does it fall back to
    // the light poll (ONE granule, one request: the traffic of 157 polling owners is not free) and then reads again.
    {
        const bool ok = coop_get(&box->out[par][0][0][lane], seq, a.x) & coop_get(&box->out[par][1][0][lane], seq, a.y) &
                        coop_get(&box->out[par][2][0][lane], seq, a.z) & coop_get(&box->out[par][3][0][lane], seq, a.w);
        if (__all(ok)) { a.ok = 1; return a; }
    }
    while ((uint32_t)(coop_loadu(&box->out[par][3][1][0]) >> 32) != seq) {
        if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > COOP_TIMEOUT_TICKS) { a.x = a.y = a.z = a.w = 0.0; return a; }
        __builtin_amdgcn_s_sleep(1);
    }
    for (;;) {
        const bool ok = coop_get(&box->out[par][0][0][lane], seq, a.x) & coop_get(&box->out[par][1][0][lane], seq, a.y) &
                        coop_get(&box->out[par][2][0][lane], seq, a.z) & coop_get(&box->out[par][3][0][lane], seq, a.w);
        if (__all(ok)) break;
        if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > COOP_TIMEOUT_TICKS) { a.x = a.y = a.z = a.w = 0.0; return a; }
        __builtin_amdgcn_s_sleep(1);
    }
    a.ok = 1;
    return a;
}
COOP_FN CoopAnswer coop_wait(CoopBox *box, int lane, uint32_t seq) { return coop_wait_inl(box, lane, seq); }
// two parts: part 0 from the mailbox, part 1 from the array of second answers, added in that order whichever helper answered first
DEVFN CoopAnswer coop_wait2_inl(CoopBox *box, CoopOut *out2, int lane, uint32_t seq) {
    CoopAnswer a = {0.0, 0.0, 0.0, 0.0, 0};
    const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
    const unsigned par = seq & 1u;
    for (int part = 0; part < 2; ++part) {
        uint64_t *o = part ? &out2->out[par][0][0][0] : &box->out[par][0][0][0];  // [4][2][64] granules of this part
        double x, y, z, w;
        while ((uint32_t)(coop_loadu(o + (3 * 2 + 1) * DEV_LANES) >> 32) != seq) {  // (one granule, one request: see coop_wait)
            if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > COOP_TIMEOUT_TICKS) return a;
            __builtin_amdgcn_s_sleep(1);
        }
        for (;;) {
            const bool there = (uint32_t)(coop_loadu(o + (3 * 2 + 1) * DEV_LANES + lane) >> 32) == seq;
            if (__all(there)) {
                const bool ok = coop_get(o + 0 * 2 * DEV_LANES + lane, seq, x) & coop_get(o + 1 * 2 * DEV_LANES + lane, seq, y) &
                                coop_get(o + 2 * 2 * DEV_LANES + lane, seq, z) & coop_get(o + 3 * 2 * DEV_LANES + lane, seq, w);
                if (__all(ok)) break;
            }
            if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > COOP_TIMEOUT_TICKS) { a.x = a.y = a.z = a.w = 0.0; return a; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (part == 0) { a.x = x; a.y = y; a.z = z; a.w = w; }
        else { a.x += x; a.y += y; a.z += z; a.w += w; }
    }
    a.ok = 1;
    return a;
}
static __device__ __attribute__((noinline)) CoopAnswer coop_wait2(CoopBox *box, CoopOut *out2, int lane, uint32_t seq) { return coop_wait2_inl(box, out2, lane, seq); }

// What the owner does when no helper answers: the helper's sixteen wave slots one after the other, summed in the
// helper's fold order, i.e. bit for bit the answer it did not get.  Out of line: a rare path must not cost the
// integrator role registers.
static __device__ __attribute__((noinline)) Partial4 coop_fallback(uint64_t cfg_u, uint64_t htab_u, uint64_t cols_u, const double *inb, int lane_p) {
#ifdef NYX_COOP_FAN
    const int lane = lane_p & 0xff, parts = (lane_p >> 8) & 0xf;   // (bits 8-11 of the lane argument: the parts of the fan-out)
#else
    const int lane = lane_p & 0xff, parts = (lane_p & 0x100) ? 2 : 1;  // (bit 8 of the lane argument: two parts)
#endif
    const double v0 = inb[0 * DEV_LANES + lane], v1 = inb[1 * DEV_LANES + lane], v2 = inb[2 * DEV_LANES + lane],
                 v3 = inb[3 * DEV_LANES + lane], v4 = inb[4 * DEV_LANES + lane];
    Partial4 tot = {0.0, 0.0, 0.0, 0.0};
    for (int part = 0; part < parts; ++part) {  // (every part summed on its own, then added in part order: what coop_wait does with the answers)
#ifdef NYX_COOP_FAN
        const int sched = DEV_SCHED_FAN0 + part;
#else
        const int sched = part ? DEV_SCHED_HELPER2 : DEV_SCHED_HELPER;
#endif
        Partial4 o = {0.0, 0.0, 0.0, 0.0};
        for (int hw = 0; hw < DEV_MAX_WAVES; ++hw) {
            const Partial4 p = (((CfgPtr)uniform_u64(cfg_u))->harm_feed & 2) ? harmonics_stream(cfg_u, cols_u, hw, sched, v0, v1, v2, v3, v4)
                                                                             : harmonics_partial(cfg_u, htab_u, cols_u, hw, sched, v0, v1, v2, v3, v4);
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        if (part == 0) tot = o;
        else { tot.x += o.x; tot.y += o.y; tot.z += o.z; tot.w += o.w; }
    }
    return tot;
}

// Helper workgroup.  Jobs are CLAIMED, not assigned: the owners are dealt into sets of at most 16, a helper watches
// one set (lane l < 16 of its wave 0 <-> one owner: two 64-byte loads scan the set), and whichever helper of the set is
// free takes the next posted job with a compare-and-swap on claimed[owner].  The load evens out by itself whatever the
// ratio of helpers to owners.
//
// Inside the workgroup the job is a two-slot software pipeline with no workgroup barrier: wave 0 is the PRODUCER (it
// claims job j+1 and fetches its five input rows from the mailbox into LDS while the others work on job j), waves
// 1..14 are the column waves (one column each), and wave 15 ANSWERS: it waits for the fourteen partial sums, folds
// them in the fixed wave order and writes the answer.  So the two memory round trips of a job (inputs in, answer out,
// ~2 us each on uncached memory) overlap the arithmetic of its neighbours: a helper's job period is its longest
// column, not column + latencies.  LDS words: ready[s] / answered[s] = 1 + number of the job last published /
// answered in slot s, cnt[s] = column waves that have delivered.
#ifndef NYX_SEG_PROF
#define NYX_SEG_PROF 0  /* 1 adds the integrator's per-piece timers (rows 34-35); off in the product build, they cost registers */
#endif
#ifndef STEP_ONE_POW
/* step control: one pow in front of the accept / reject branches (round 6).  The sixteen-wave plain kernels only - measured same box,
 * three interleaved pairs each: 24 h of configs[1] 595.1 -> 592.1 ms (the decision 13.0 k -> 10.7 k cycles per attempt); the eight-wave
 * kernel of config 3, whose integrator shares its SIMD with one almanac wave, 46.8 ms with two pows against 47.2 with one */
#define STEP_ONE_POW ((NYX_EMIT & (NYX_EMIT_PLAIN16 | NYX_EMIT_PLAIN16_P2 | NYX_EMIT_PLAIN16_FAN)) ? 1 : 0)
#endif
#ifndef FAN_POLL_FETCH
#define FAN_POLL_FETCH 0  /* 1: the fan-out producer's poll fetches all ten granules (one round trip from post to inputs instead of two); measured neutral, round 6: 1 250 x 24 h 357.2-359.9 ms against 357.7-359.5 */
#endif
#ifndef FAN_SKIP
#define FAN_SKIP 1  /* fan-out kernel: a wave without columns under the schedule in force skips its walk (five LDS reads, a call, the schedule lookup) */
#endif
#ifndef FAN_SUMS
#define FAN_SUMS NYX_FAN_SUMS  /* devcfg.h: fan-out mode, the integrator's two stage sums formed by a column wave of their own (fan_sums, DevCfg.sums_wave1) */
#endif
#ifndef STEP_OOL
#ifdef NYX_COOP_FAN
#define STEP_OOL 1          /* step control out of line (integ_step, round 6): the fan-out kernel, whose period IS the integrator's chain (1 250 x 24 h: 391 -> 382.5 ms) */
#else
#define STEP_OOL 0          /* the other INTEG_OOL kernels keep it inline: measured same box, 24 h of configs[1]: 601.9 ms out of line against 597.9 inline (three interleaved pairs; step control 19 k -> 11.9 k cycles per attempt either way, but the period there is the column waves') */
#endif
#endif
#ifndef STEP_SUMS_UNROLL
#define STEP_SUMS_UNROLL 0  /* step control: unroll factor of the loop over the stages of its two sums (0: as the compiler leaves it) */
#endif
#ifndef COOP_AFFINITY
#define COOP_AFFINITY 1  /* helpers take a job of their own first (see helper_body) */
#endif
#ifndef HELPER_SLOTS
#define HELPER_SLOTS 2  /* jobs in flight inside a helper (see helper_body: three and four were measured, slower) */
#endif
#define HELPER_LDS_BYTES ((HELPER_SLOTS * DEV_MAX_WAVES * 4 * DEV_LANES + HELPER_SLOTS * 5 * DEV_LANES) * 8 + 64 * 4)
DEVFN void helper_body(const DevBatch &bt, CfgPtr cfg, HarmPtr htab, ColPtr cols, char *smem, int lane, int wave) {
    // HELPER_SLOTS jobs in flight.  Round 5 measured three and four (in-kernel accounting of a helper, tools/sweep.py "profile"): with
    // two slots the producer waits ~9 k cycles per job for a slot and only then scans, claims and fetches (~10 k cycles of uncached
    // round trips); more slots do move the claim under the arithmetic - and lose, 84.9 -> 92.3 -> 103.3 ms per 3 h of configs[1]:
    // a job claimed early queues INSIDE this helper behind two or three others while another helper would have been free sooner
    // (lost claims per job 2.4 -> 2.6 -> 4.0): the rate of jobs is the owners', what counts is each job's turnaround.
    constexpr int NS = HELPER_SLOTS;
    double *part = (double *)smem;                                  // [NS][16][4][64]
    double *inl = part + NS * DEV_MAX_WAVES * 4 * DEV_LANES;        // [NS][5][64]
    int *ctl = (int *)(inl + NS * 5 * DEV_LANES);
    const LdsFlagPtr ready = (LdsFlagPtr)ctl, answered = (LdsFlagPtr)ctl + 4, jown = (LdsFlagPtr)ctl + 8, jseq = (LdsFlagPtr)ctl + 12, jpart = (LdsFlagPtr)ctl + 20;
    int *cnt = ctl + 16;
    constexpr int parts = COOP_PARTS_HERE;
    const int answer_wave = (int)(blockDim.x / DEV_LANES) - 1;
    const int n_col_waves = answer_wave - 1;
    if (wave == 0 || wave == answer_wave) {
        if (wave == 0 && lane < 32) ctl[lane] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) part[((sl * DEV_MAX_WAVES + wave) * 4 + q) * DEV_LANES + lane] = 0.0;
        }
    }
    __syncthreads();
#ifdef NYX_COOP_FAN
    // FAN-OUT mode (small shards: the idle CUs outnumber the owners at least two to one).  Helper h is DEDICATED to owner h % owners and
    // evaluates part h / owners of that owner's hand-off - no scan words, no claim, no lost race: its producer polls the tag of the
    // owner's input rows (the poll is half of the fetch) and the columns of an evaluation are dealt over coop_parts helper workgroups,
    // so a job is a fraction of a column set (two waves per SIMD or fewer finish in ~10 k cycles where fourteen need ~17 k) and the
    // owner keeps next to nothing.  The owner's side is the single-part protocol unchanged - one post, one answer in its mailbox -:
    // the helpers of the parts 1.. write their sums to coop_out2[owner * parts + part], the part-0 helper (the LEAD) waits for them,
    // adds them to its own in part order and answers.  That hop is on no critical path: the owner asks for the answer ~1.5 periods
    // after the post.  Nothing assumes residency: a part that never answers makes the lead give up, the owner time out after 2 ms and
    // walk every part's columns itself (coop_fallback: the same sums in the same order).
    const int fan_h = (int)blockIdx.x - bt.coop_base;
    const int fan_own_n = (int)((bt.n + DEV_LANES - 1) / DEV_LANES);
    const int fan_owner = fan_h % fan_own_n, fan_part = fan_h / fan_own_n;
    const int fan_parts = bt.coop_parts;
    if (fan_part >= fan_parts) return;
    if (wave == 0) {
        const int fan_widx = bt.coop_sets > 0 ? (fan_owner % bt.coop_sets) * COOP_SET + fan_owner / bt.coop_sets : 0;  // (the owner's coop_widx)
        const CoopBox *b = bt.coop_box + fan_owner;
        for (int j = 0;; ++j) {
            const int s = j % NS;
            const uint32_t seq = (uint32_t)j + 1u;   // the owner's evaluations, in order: every one of them is this helper's job
            const unsigned par = seq & 1u;
            int owner = fan_owner;
            const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
            bool slot_free = j < NS;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
            for (int it = 0;; ++it) {
                if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > 5000 * COOP_TIMEOUT_TICKS) { owner = -1; break; }  // 10 s: never spin forever
                if (!slot_free) {  // the job that used this slot NS rounds ago has been answered
                    slot_free = answered[s] == j - (NS - 1);
                    if (!slot_free) { __builtin_amdgcn_s_sleep(4); continue; }
                }
                // the poll IS the fetch (FAN_POLL_FETCH): all ten granules of this lane every time, accepted when every tag of every lane
                // carries the sequence number - one uncached round trip from the owner's post to the inputs in hand instead of two (a poll of
                // one granule, then the fetch), on the turnaround that bounds the owner's period in this mode
#if FAN_POLL_FETCH
                {
#else
                if ((uint32_t)(coop_loadu(&b->in[par][4][1][0]) >> 32) == seq) {
#endif
                    const bool got = coop_get(&b->in[par][0][0][lane], seq, v0) & coop_get(&b->in[par][1][0][lane], seq, v1) &
                                     coop_get(&b->in[par][2][0][lane], seq, v2) & coop_get(&b->in[par][3][0][lane], seq, v3) &
                                     coop_get(&b->in[par][4][0][lane], seq, v4);
                    if (__all(got)) break;
#if !FAN_POLL_FETCH
                    continue;
#endif
                }
                if ((it & 7) == 7 && coop_load(bt.coop_finished + fan_widx) != 0u) { owner = -1; break; }  // the owner is done (or carries on alone)
                __builtin_amdgcn_s_sleep(2);
            }
            if (owner >= 0) {
                double *il = inl + s * 5 * DEV_LANES;
                il[0 * DEV_LANES + lane] = v0; il[1 * DEV_LANES + lane] = v1; il[2 * DEV_LANES + lane] = v2;
                il[3 * DEV_LANES + lane] = v3; il[4 * DEV_LANES + lane] = v4;
            }
            if (lane == 0) { jown[s] = owner; jseq[s] = (int)seq; jpart[s] = fan_part; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) ready[s] = j + 1;
            if (owner < 0) break;
        }
        return;
    }
#else
    if (wave == 0) {
        const int h = (int)blockIdx.x - bt.coop_base;
        const int64_t n_own = (bt.n + DEV_LANES - 1) / DEV_LANES;
        const int n_sets = bt.coop_sets;
        const int set = h % n_sets;
        const int64_t mine = (int64_t)set + (int64_t)lane * n_sets;  // the owner this lane watches (lanes 0..15)
        const bool has = lane < COOP_SET && mine < n_own;
        const int widx = set * COOP_SET + lane;                       // its scan words
        unsigned turn = (unsigned)h;
        const bool pprof = NYX_PROF && bt.prof != nullptr && (int)blockIdx.x == bt.coop_base;
        int64_t pp_slot = 0, pp_scan = 0, pp_jobs = 0, pp_lost = 0;
        const int64_t pp_start = pprof ? (int64_t)__builtin_readcyclecounter() : 0;
        for (int j = 0;; ++j) {
            const int s = j % NS;
            int owner = -1, sub = 0;
            uint32_t seq = 0;
            const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
            const int64_t pc0 = pprof ? (int64_t)__builtin_readcyclecounter() : 0;
            int64_t pc1 = pc0;
            bool slot_free = j < NS;
            for (;;) {
                if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > 5000 * COOP_TIMEOUT_TICKS) { owner = -1; break; }  // 10 s: never spin forever
                if (!slot_free) {  // the job that used this slot NS rounds ago has been answered
                    slot_free = answered[s] == j - (NS - 1);
                    if (!slot_free) { __builtin_amdgcn_s_sleep(4); continue; }
                    if (pprof) pc1 = (int64_t)__builtin_readcyclecounter();
                }
                // the two words are read by independent loads: a pair (old posted, new claimed) is possible and must not look
                // like a job, hence "posted is AHEAD of claimed", not "differs from"
                const uint32_t posted = has ? coop_load(bt.coop_posted + widx) : 0u;
                const uint32_t claimed = has ? coop_load(bt.coop_claimed + widx) : 0u;
                const uint64_t cand = __ballot(has && (int32_t)(posted - claimed) > 0);
                if (cand) {
                    // first candidate at or after a rotating start lane, so that the helpers of a set spread over the jobs
                    const unsigned rot = turn++ & 63u;
                    const uint64_t hi = cand >> rot;
                    int pick = hi ? (int)rot + __builtin_ctzll(hi) : __builtin_ctzll(cand);
#if COOP_AFFINITY
                    // ... but a job has a PREFERRED helper - (owner slot + job number) mod the set's helpers, so that an owner's consecutive
                    // jobs go round the set - and a helper takes one of its own first: two idle helpers of a set that see the same jobs no
                    // longer go for the same one.  Round 5, 3 h of configs[1], same box, alternating: 82.6 / 83.3 ms without, 79.3 / 79.2 with
                    // (lost claims per job 1.75 -> 1.3; the results are the same bits).  Measured and dropped: a static owner -> helper
                    // preference (80.7-81.2), waiting one more scan for a job of its own (82.8-83.3: lost claims 0.55, but the wait is on the
                    // job's path), every helper taking the waiting job NEAREST to its rank (83.1-84.3: it takes its neighbour's).
                    {
                        const int hs = (bt.coop_helpers - set + n_sets - 1) / n_sets;   // helpers watching this set
                        const int rank = h / n_sets;
                        const uint64_t pref = __ballot(has && (int32_t)(posted - claimed) > 0 && hs > 0 && (int)(((unsigned)lane + claimed) % (unsigned)hs) == rank);
                        if (pref) pick = __builtin_ctzll(pref);
                    }
#endif
                    // jobs are taken in order, one at a time: an owner may have two outstanding (the pipelined loop posts
                    // stage i+1 before it has read the answer of stage i).  The five input rows of the job are fetched in the
                    // shadow of the compare-and-swap (they were complete before `posted` moved): one memory round trip, not two.
                    const int owner_c = (int)__shfl((int)mine, pick);
                    const uint32_t sub_c = (uint32_t)__shfl((int)claimed, pick) + 1u;           // the sub-job being claimed (1, 2, ...)
                    const uint32_t seq_c = parts == 2 ? (sub_c + 1u) >> 1 : sub_c;             // its evaluation ...
                    const int part_c = parts == 2 ? (int)((sub_c - 1u) & 1u) : 0;              // ... and which part of the hand-off
                    int won = 0;
                    if (lane == pick) {
                        uint32_t expect = claimed;
                        won = __hip_atomic_compare_exchange_strong(bt.coop_claimed + widx, &expect, claimed + 1u, __ATOMIC_RELAXED,
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                    }
                    const CoopBox *b = bt.coop_box + owner_c;
                    const unsigned par = seq_c & 1u;
                    // (measured: fetching only after the claim has succeeded costs 7 % of the north-star run - the helper's job
                    //  latency is what bounds its share)
                    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
                    bool got = false;
                    // Fetch the inputs only AFTER the claim has succeeded.  (Rounds 1-3 fetched them in the shadow of the compare-and-swap -
                    // measured then as 7 % faster; with the tagged-granule transport the opposite holds: every lost race was 5 KB of
                    // uncached reads, and the north-star run is 5.5 % FASTER without them - 719.5 -> 679.8 ms, same box.  coop_mute bit 1
                    // = debug_flags 0x200000 restores the speculative fetch.)
                    const bool lazy = (bt.coop_mute & 2) == 0;
                    if (!lazy)
                        got = coop_get(&b->in[par][0][0][lane], seq_c, v0) & coop_get(&b->in[par][1][0][lane], seq_c, v1) &
                              coop_get(&b->in[par][2][0][lane], seq_c, v2) & coop_get(&b->in[par][3][0][lane], seq_c, v3) &
                              coop_get(&b->in[par][4][0][lane], seq_c, v4);
                    if (__shfl(won, pick)) {
                        // the job is ours; its inputs were stored before the sequence number, but nothing orders the two: poll until
                        // every granule carries the tag (normally the first look already does)
                        const int64_t tw = (int64_t)__builtin_amdgcn_s_memrealtime();
                        bool first = lazy;
                        while (!__all(got)) {
                            if ((int64_t)__builtin_amdgcn_s_memrealtime() - tw > 100 * COOP_TIMEOUT_TICKS) break;  // (0.2 s: the owner has long given up on us)
                            if (!first) __builtin_amdgcn_s_sleep(1);
                            first = false;
                            got = coop_get(&b->in[par][0][0][lane], seq_c, v0) & coop_get(&b->in[par][1][0][lane], seq_c, v1) &
                                  coop_get(&b->in[par][2][0][lane], seq_c, v2) & coop_get(&b->in[par][3][0][lane], seq_c, v3) &
                                  coop_get(&b->in[par][4][0][lane], seq_c, v4);
                        }
                        // the poll timed out: the inputs were never seen whole.  The job is NOT worked on - a tagged answer vouches for
                        // the data it was computed from, and this one would be computed from torn or zero inputs; the owner gave up
                        // waiting 2 ms in, walks these columns itself and never looks at the mailbox again
                        if (!__all(got)) continue;
                        owner = owner_c;
                        seq = seq_c;
                        sub = part_c;
                        double *il = inl + s * 5 * DEV_LANES;
                        il[0 * DEV_LANES + lane] = v0; il[1 * DEV_LANES + lane] = v1; il[2 * DEV_LANES + lane] = v2;
                        il[3 * DEV_LANES + lane] = v3; il[4 * DEV_LANES + lane] = v4;
                        break;
                    }
                    if (pprof) ++pp_lost;
                    continue;  // another helper was faster: look again
                }
                const uint32_t fin = has ? coop_load(bt.coop_finished + widx) : 1u;
                if (__all(fin != 0u)) { owner = -1; break; }
                __builtin_amdgcn_s_sleep(8);  // ~0.2 us between scans: the set's words are one memory line shared by ~10 helpers (scanning 2-5x less often: no change)
            }
            if (lane == 0) { jown[s] = owner; jseq[s] = (int)seq; jpart[s] = sub; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) ready[s] = j + 1;
            if (pprof) { const int64_t now = (int64_t)__builtin_readcyclecounter(); pp_slot += pc1 - pc0; pp_scan += now - pc1; ++pp_jobs; }
            if (owner < 0) break;
        }
        if (pprof && lane == 0) {  // [0] cycles waiting for a free slot (the column waves are behind), [1] cycles from a free slot to a won and fetched job, [2] jobs, [3] lost claims
            int64_t *row = bt.prof + 17 * 8;
            row[0] = pp_slot; row[1] = pp_scan; row[2] = pp_jobs; row[3] = pp_lost; row[5] = (int64_t)__builtin_readcyclecounter() - pp_start;
        }
        return;
    }
#endif  // NYX_COOP_FAN
    // optional accounting of the FIRST helper workgroup (NYX_HIP_PROFILE; rows 17.. of the profile, one per wave): [0] cycles in the
    // column walk, [1] cycles waiting for a job, [2] jobs, [3] cycles from a job's publication in LDS to this wave's delivery, [5] total
#ifdef HELPER_PRIO
    // issue priority against the arbiter's oldest-first rule: the four waves of a SIMD start a job together, and served oldest first the
    // oldest is done after half the job's time and runs ahead into the next job while the youngest - whose column the answer waits
    // for - gets what is left
    if (wave != answer_wave) {
        const int pr = HELPER_PRIO == 1 ? (wave >> 2) : (3 - (wave >> 2));
        if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
#endif
    const bool hprof = NYX_PROF && bt.prof != nullptr && (int)blockIdx.x == bt.coop_base;
    int64_t hp_busy = 0, hp_wait = 0, hp_jobs = 0;
    const int64_t hp_start = hprof ? (int64_t)__builtin_readcyclecounter() : 0;
    for (int j = 0;; ++j) {
        const int s = j % NS;
        {
            const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
            const int64_t c0 = hprof ? (int64_t)__builtin_readcyclecounter() : 0;
            while (ready[s] != j + 1) {
                if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > 6000 * COOP_TIMEOUT_TICKS) return;  // (the producer gives up after 10 s)
                __builtin_amdgcn_s_sleep(4);
            }
            if (hprof) hp_wait += (int64_t)__builtin_readcyclecounter() - c0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int owner = jown[s];
        const uint32_t seq = (uint32_t)jseq[s];
        const int sub = jpart[s];
        if (owner < 0) break;
        double *ps = part + s * DEV_MAX_WAVES * 4 * DEV_LANES;
        const int64_t hp_c0 = hprof ? (int64_t)__builtin_readcyclecounter() : 0;
        if (wave != answer_wave) {
            const double *il = inl + s * 5 * DEV_LANES;
            const double v0 = il[0 * DEV_LANES + lane], v1 = il[1 * DEV_LANES + lane], v2 = il[2 * DEV_LANES + lane],
                         v3 = il[3 * DEV_LANES + lane], v4 = il[4 * DEV_LANES + lane];
#ifdef NYX_COOP_FAN
            const int hsched = DEV_SCHED_FAN0 + sub;
#else
            const int hsched = sub ? DEV_SCHED_HELPER2 : DEV_SCHED_HELPER;
#endif
            const Partial4 pr = (cfg->harm_feed & 2) ? harmonics_stream((uint64_t)cfg, (uint64_t)cols, wave, hsched, v0, v1, v2, v3, v4)
                                               : harmonics_partial((uint64_t)cfg, (uint64_t)htab, (uint64_t)cols, wave, hsched, v0, v1, v2, v3, v4);
            double *pp = ps + wave * 4 * DEV_LANES;
            pp[0 * DEV_LANES + lane] = pr.x; pp[1 * DEV_LANES + lane] = pr.y; pp[2 * DEV_LANES + lane] = pr.z; pp[3 * DEV_LANES + lane] = pr.w;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) (void)__hip_atomic_fetch_add(cnt + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (hprof) { hp_busy += (int64_t)__builtin_readcyclecounter() - hp_c0; ++hp_jobs; }
            continue;
        }
        // ---- the answering wave: wait for the column waves, fold in the fixed wave order (the slots of the producer and of
        // this wave hold zeros), answer.  None of this is on a column wave's path.
        {
            const int64_t t0 = (int64_t)__builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(cnt + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != n_col_waves) {
                if ((int64_t)__builtin_amdgcn_s_memrealtime() - t0 > 6000 * COOP_TIMEOUT_TICKS) return;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        CoopBox *b = bt.coop_box + owner;
        const unsigned par = seq & 1u;
        double o[4] = {0.0, 0.0, 0.0, 0.0};
        for (int w = 0; w < DEV_MAX_WAVES; ++w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] += ps[(w * 4 + q) * DEV_LANES + lane];
        }
#ifdef NYX_COOP_FAN
        bool fan_ok = true;
        if (sub == 0) {
            // the lead: the sums of the parts 1.., in part order (what coop_fallback adds up when the owner walks the parts itself).
            // Four parts per batch: their thirty-two granules are requested together and the tags checked afterwards - one uncached round
            // trip per batch where a part at a time cost one each (seven in a row for eight parts, ~8 k cycles on the job's turnaround,
            // which bounds the owner's period in this mode: round 6, GPU call 26).  The additions stay in part order.
            const int64_t tl = (int64_t)__builtin_amdgcn_s_memrealtime();
            for (int p0 = 1; p0 < fan_parts && fan_ok; p0 += 4) {
                uint64_t raw[4][8];
                bool pend[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pend[q] = p0 + q < fan_parts;  // (uniform)
#pragma unroll
                    for (int r = 0; r < 8; ++r) raw[q][r] = 0;
                }
                for (;;) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (pend[q]) {
                            const uint64_t *o2 = &bt.coop_out2[owner * fan_parts + p0 + q].out[par][0][0][0] + lane;
#pragma unroll
                            for (int r = 0; r < 8; ++r) raw[q][r] = coop_loadu(o2 + (r >> 1) * 2 * DEV_LANES + (r & 1) * DEV_LANES);
                        }
                    }
                    bool any = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (pend[q]) {
                            bool got = true;
#pragma unroll
                            for (int r = 0; r < 8; ++r) got = got && (uint32_t)(raw[q][r] >> 32) == seq;
                            if (__all(got)) pend[q] = false; else any = true;
                        }
                    }
                    if (!any) break;
                    if ((int64_t)__builtin_amdgcn_s_memrealtime() - tl > COOP_TIMEOUT_TICKS) { fan_ok = false; break; }  // (the owner gives up at the same age)
                    __builtin_amdgcn_s_sleep(2);
                }
                if (fan_ok) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (p0 + q < fan_parts) {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                o[k] += __longlong_as_double((long long)((raw[q][2 * k] & 0xffffffffull) | (raw[q][2 * k + 1] << 32)));
                        }
                    }
                }
            }
        }
        if (fan_ok) {
            uint64_t *og = sub ? &bt.coop_out2[owner * fan_parts + sub].out[par][0][0][0] : &b->out[par][0][0][0];
#pragma unroll
            for (int q = 0; q < 4; ++q) coop_put(og + q * 2 * DEV_LANES + lane, o[q], seq);
        }
#else
        {
            uint64_t *og = (sub && bt.coop_out2) ? &bt.coop_out2[owner].out[par][0][0][0] : &b->out[par][0][0][0];
#pragma unroll
            for (int q = 0; q < 4; ++q) coop_put(og + q * 2 * DEV_LANES + lane, o[q], seq);  // tagged granules: no drain, no flag (the owner polls the last one)
        }
#endif
        if (lane == 0) __hip_atomic_store(cnt + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) answered[s] = j + 1;  // the slot may be refilled: its partial sums are in registers
        if (lane == 0 && bt.prof != nullptr) atomicAdd((unsigned long long *)bt.prof + 16 * 8 + 4, 1ull);
        if (hprof) { hp_busy += (int64_t)__builtin_readcyclecounter() - hp_c0; ++hp_jobs; }
    }
    if (hprof && lane == 0) {
        int64_t *row = bt.prof + (17 + wave) * 8;
        row[0] = hp_busy; row[1] = hp_wait; row[2] = hp_jobs; row[5] = (int64_t)__builtin_readcyclecounter() - hp_start;
    }
}

