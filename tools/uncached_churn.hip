// uncached_churn.hip — does allocate / free churn of UNCACHED device memory break a two-workgroup mailbox exchange?
//
// Round 3 found the ~17th cooperative context of one process never finishing (behind ~90 other tests) and worked around it with a
// per-process pool of mailbox blocks that is never freed (nyx_amd/csrc/abi.cpp, mailbox_acquire).  This is the stand-alone
// reproducer the verdict asked for: no library, no kernel of the product - only what the mailboxes rely on:
//
//   * a block from hipExtMallocWithFlags(hipDeviceMallocUncached), zeroed with hipMemsetAsync;
//   * a launch of 2 x PAIRS workgroups (owners and helpers, each pair on different CUs) that play ROUNDS rounds of the product's
//     exchange - owner: 5 x 64 tagged 8-byte granules {half | seq << 32} + a sequence word; helper: polls the word, reads the
//     granules until their tags agree, answers 4 x 64 tagged granules; owner: polls the last granule, checks all tags - with
//     device-scope relaxed 8-byte atomics only, every spin bounded (a stall is REPORTED, never hung on);
//   * hipFree of the block (mode "churn") or reuse of one block (mode "pool"), CYCLES times, with optional unrelated allocations
//     of ordinary memory in between ("noise": other tests' contexts come and go).
//
// usage: uncached_churn [cycles=64] [mode: churn|pool] [noise 0|1] [pairs=96] [rounds=2000]
// Output: one line per cycle that stalled or failed, and a summary.  Exit code 0 = every cycle completed.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)

struct Box {
    uint64_t in[2][5][2][64];
    uint64_t out[2][4][2][64];
    uint32_t posted, pad[15];
};

__device__ __forceinline__ uint64_t ld(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#define SPIN_LIMIT 20000000LL /* 0.2 s of the 100 MHz realtime counter per wait */

// blocks [0, pairs): owners; [pairs, 2 pairs): helpers.  result[pair] = rounds completed by the owner (== rounds when all went well),
// result[pairs + pair] = first round a wait timed out in (0 = none), stall_kind[pair]: 1 owner waited for the answer, 2 helper for the job
__global__ __launch_bounds__(64) void exchange(Box *box, int pairs, int rounds, int *result, int *stall_kind) {
    const int lane = threadIdx.x;
    const bool owner = (int)blockIdx.x < pairs;
    const int pair = owner ? (int)blockIdx.x : (int)blockIdx.x - pairs;
    Box *b = box + pair;
    int done = 0;
    for (uint32_t seq = 1; seq <= (uint32_t)rounds; ++seq) {
        const unsigned par = seq & 1u;
        const uint64_t tag = (uint64_t)seq << 32;
        if (owner) {
            for (int q = 0; q < 5; ++q) {
                const uint64_t v = (uint64_t)(seq * 1000003u + q * 101u + lane);
                st(&b->in[par][q][0][lane], (v & 0xffffffffull) | tag);
                st(&b->in[par][q][1][lane], ((v * 7u) & 0xffffffffull) | tag);
            }
            if (lane == 0) st32(&b->posted, seq);
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            bool ok = false;
            for (;;) {
                const bool there = (uint32_t)(ld(&b->out[par][3][1][lane]) >> 32) == seq;
                if (__all(there)) {
                    bool good = true;
                    for (int q = 0; q < 4; ++q) {
                        const uint64_t lo = ld(&b->out[par][q][0][lane]), hi = ld(&b->out[par][q][1][lane]);
                        const uint64_t want = (uint64_t)(seq * 1000003u + q * 101u + lane) + 5u;
                        good = good && (lo >> 32) == seq && (hi >> 32) == seq && (lo & 0xffffffffull) == (want & 0xffffffffull);
                    }
                    if (__all(good)) { ok = true; break; }
                }
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) {
                if (lane == 0) { result[pairs + pair] = (int)seq; stall_kind[pair] = 1; }
                break;
            }
            ++done;
        } else {
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            bool ok = false;
            uint64_t v0 = 0;
            for (;;) {
                if (ld32(&b->posted) >= seq) {
                    bool good = true;
                    for (int q = 0; q < 5; ++q) {
                        const uint64_t lo = ld(&b->in[par][q][0][lane]), hi = ld(&b->in[par][q][1][lane]);
                        good = good && (lo >> 32) == seq && (hi >> 32) == seq;
                        if (q == 0) v0 = lo & 0xffffffffull;
                    }
                    if (__all(good)) { ok = true; break; }
                }
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) {
                if (lane == 0 && stall_kind[pair] == 0) stall_kind[pair] = 2;
                break;
            }
            for (int q = 0; q < 4; ++q) {
                const uint64_t v = (uint64_t)(seq * 1000003u + q * 101u + lane) + 5u;
                st(&b->out[par][q][0][lane], (v & 0xffffffffull) | tag);
                st(&b->out[par][q][1][lane], ((v0 + q) & 0xffffffffull) | tag);
            }
        }
    }
    if (owner && lane == 0) result[pair] = done;
}

int main(int argc, char **argv) {
    const int cycles = argc > 1 ? std::atoi(argv[1]) : 64;
    const bool churn = argc > 2 ? std::strcmp(argv[2], "pool") != 0 : true;
    const bool noise = argc > 3 ? std::atoi(argv[3]) != 0 : true;
    const int pairs = argc > 4 ? std::atoi(argv[4]) : 96;
    const int rounds = argc > 5 ? std::atoi(argv[5]) : 2000;
    int *d_result = nullptr, *d_kind = nullptr;
    CHECK(hipMalloc(&d_result, 2 * pairs * sizeof(int)));
    CHECK(hipMalloc(&d_kind, pairs * sizeof(int)));
    std::vector<int> result(2 * pairs), kind(pairs);
    Box *pool = nullptr;
    const size_t bytes = (size_t)pairs * sizeof(Box);
    int bad_cycles = 0;
    float total_ms = 0.f;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<void *> junk;
    for (int c = 0; c < cycles; ++c) {
        Box *box = pool;
        if (churn || !box) {
            CHECK(hipExtMallocWithFlags((void **)&box, bytes, hipDeviceMallocUncached));
            if (!churn) pool = box;
        }
        if (noise) {  // unrelated traffic of the allocator between two mailbox blocks, as other contexts would cause
            void *p = nullptr;
            CHECK(hipMalloc(&p, (size_t)(1 + c % 7) << 20));
            junk.push_back(p);
            if (junk.size() > 3) { CHECK(hipFree(junk.front())); junk.erase(junk.begin()); }
        }
        CHECK(hipMemsetAsync(box, 0, bytes, nullptr));
        CHECK(hipMemsetAsync(d_result, 0, 2 * pairs * sizeof(int), nullptr));
        CHECK(hipMemsetAsync(d_kind, 0, pairs * sizeof(int), nullptr));
        CHECK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(exchange, dim3(2 * pairs), dim3(64), 0, nullptr, box, pairs, rounds, d_result, d_kind);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        CHECK(hipMemcpy(result.data(), d_result, 2 * pairs * sizeof(int), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(kind.data(), d_kind, pairs * sizeof(int), hipMemcpyDeviceToHost));
        int stalled = 0, first_pair = -1, first_round = 0, first_kind = 0;
        for (int p = 0; p < pairs; ++p)
            if (result[p] != rounds) { ++stalled; if (first_pair < 0) { first_pair = p; first_round = result[pairs + p]; first_kind = kind[p]; } }
        if (stalled) {
            ++bad_cycles;
            std::printf("cycle %d (block %p): %d of %d pairs did not finish; first: pair %d at round %d (%s), %.1f ms\n", c, (void *)box, stalled, pairs,
                        first_pair, first_round, first_kind == 1 ? "owner never saw the answer" : (first_kind == 2 ? "helper never saw the job" : "?"), ms);
        }
        if (churn) CHECK(hipFree(box));
    }
    std::printf("%s, noise %d: %d cycles x %d pairs x %d rounds: %d cycles with a stalled exchange; %.2f us per round trip on average\n",
                churn ? "churn (allocate / free every cycle)" : "pool (one block)", noise ? 1 : 0, cycles, pairs, rounds, bad_cycles,
                total_ms * 1e3f / ((float)cycles * (float)rounds));
    return bad_cycles ? 1 : 0;
}
