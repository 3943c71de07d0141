// ub_handoff3.hip -- round 3: what moves the floor of the all-to-all hand-off of scratch/ub_handoff.hip (mode 0: every workgroup publishes
// n / nwg granules with one sc1 store and sweeps the whole n-granule vector with sc1 loads until every tag matches)?
//   * DEPTH sweeps in flight per poller (1 = the engine of round 2: a sweep is issued when the previous one has failed; 2, 3: a new sweep is
//     issued as soon as the oldest has been checked, so the vector is sampled every RTT / DEPTH instead of every RTT)
//   * the memory the vector lives in: hipMalloc (coarse-grained), fine-grained, uncached (hipExtMallocWithFlags)
//   * the number of resident workgroups (256 / 128 / 64: producers and readers alike)
//   * a fixed "compute" delay between a workgroup's READY and its publish (0 / 1 us), as in ub_handoff.hip
// Reported: us per phase, and for workgroup 0 the mean round-trip time of one sweep (issue -> all loads returned).
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ub_handoff3 scratch/ub_handoff3.hip && scratch/ub_handoff3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ uint32_t tags_bad(u32x4 g, uint32_t tag) { return ((g.x >> 16) ^ tag) | ((g.y >> 16) ^ tag) | ((g.z >> 16) ^ tag) | ((g.w >> 16) ^ tag); }

struct Args {
    uint32_t* glob;  // [2][n] granules (two buffers alternate)
    int* err;
    unsigned long long* log;  // [0] total ticks, [1] sum of sweep round trips of workgroup 0, [2] their count, [3] sweeps of wg 0
    int n, nphase, epoch, work_ns, nwg, delay_ns, jitter_ns;
};

template <int NLD, int DEPTH>
__global__ void __launch_bounds__(64) handoff_kernel(const Args a) {
    const int lane = threadIdx.x, wg = blockIdx.x, n = a.n;
    unsigned long long t0 = 0, rtt = 0, nrt = 0;
    const int per = n / a.nwg; /* granules per workgroup: a multiple of 4 */
    uint32_t keep = 0;
    for (int p = 0; p < a.nphase; p++) {
        const uint32_t tag = (uint32_t)(a.epoch * 1024 + p + 1) & 0xffffu;
        uint32_t* const gbuf = a.glob + (size_t)(p & 1) * n;
        if (4 * lane < per) {
            const uint32_t v = (uint32_t)(wg * 7 + p + lane + keep) & 0xfff0u;
            u32x4 o = {(tag << 16) | v, (tag << 16) | (v + 1), (tag << 16) | (v + 2), (tag << 16) | (v + 3)};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrc(gbuf + wg * per + 4 * lane, 16), 0, 0, 16 /* sc1 */);
        }
        if (p == 4 && wg == 0 && lane == 0) t0 = __builtin_amdgcn_s_memrealtime();
        const __amdgpu_buffer_rsrc_t rs = rsrc(gbuf, (uint32_t)n * 4u);
        if (a.delay_ns > 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime() + (unsigned long long)a.delay_ns / 10;
            while (__builtin_amdgcn_s_memrealtime() < t1) __builtin_amdgcn_s_sleep(1);
        }
        u32x4 g[DEPTH][NLD];
        unsigned long long ts[DEPTH];
        auto issue = [&](int k) {
            ts[k] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
            for (int r = 0; r < NLD; r++) g[k][r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
        };
#pragma unroll
        for (int k = 0; k < DEPTH; k++) issue(k);
        bool done = false;
        for (int spins = 0; !done; spins++) {
#pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                if (done) break;
                uint32_t bad = 0;
#pragma unroll
                for (int r = 0; r < NLD; r++) bad |= tags_bad(g[k][r], tag);
                const bool ok = __all(bad == 0);
                if (wg == 0 && p >= 4) rtt += __builtin_amdgcn_s_memrealtime() - ts[k], nrt++;
                if (ok) {
#pragma unroll
                    for (int r = 0; r < NLD; r++) keep += g[k][r].x & 1u;
                    done = true;
                } else {
                    issue(k);
                }
            }
            if (spins > (1 << 16)) {
                if (lane == 0) atomicAdd(a.err, 1);
                done = true;
            }
        }
        if (a.work_ns > 0) {
            const unsigned int jit = a.jitter_ns > 0 ? ((unsigned int)(wg * 2654435761u + p * 40503u) >> 8) % (unsigned int)a.jitter_ns : 0u;
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(a.work_ns + jit) / 10;
            while (__builtin_amdgcn_s_memrealtime() < t1) __builtin_amdgcn_s_sleep(1);
        }
    }
    if (keep == 0x12345678u && lane == 0) a.err[1] = 1;
    if (wg == 0 && lane == 0) a.log[0] = __builtin_amdgcn_s_memrealtime() - t0, a.log[1] = rtt, a.log[2] = nrt;
}

template <int NLD, int DEPTH>
static void go(const Args& a, hipStream_t st) {
    hipLaunchKernelGGL((handoff_kernel<NLD, DEPTH>), dim3(a.nwg), dim3(64), 0, st, a);
}

int main(int argc, char** argv) {
    const int nphase = 404, reps = 12;
    int* err;
    unsigned long long* log;
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&log, 64));
    uint32_t* bufs[3];
    const char* mname[3] = {"coarse (hipMalloc)", "fine-grained", "uncached"};
    CK(hipMalloc(&bufs[0], 2 * 4096 * 4));
    if (hipExtMallocWithFlags((void**)&bufs[1], 2 * 4096 * 4, hipDeviceMallocFinegrained) != hipSuccess) bufs[1] = nullptr;
    if (hipExtMallocWithFlags((void**)&bufs[2], 2 * 4096 * 4, hipDeviceMallocUncached) != hipSuccess) bufs[2] = nullptr;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int epoch = 1;
    for (int n : {1024, 2048, 3072})
        for (int jitter : {0, 300, 600})
            for (int mem : {0, 2})
                for (int delay : {0, 200, 300, 400, 500, 600}) {
                    const int nwg = 256, work = 1000, depth = 1;
                    if (!bufs[mem]) continue;
                    if (mem == 0 && delay != 0 && delay != 400) continue;
                    CK(hipMemset(err, 0, 64));
                    CK(hipMemset(bufs[mem], 0xff, 2 * 4096 * 4));
                    double us = 0, rt = 0, ns = 0;
                    for (int r = 0; r < reps; r++) {
                        Args a{bufs[mem], err, log, n, nphase, epoch++, work, nwg, delay, jitter};
                        if (n == 1024) go<4, 1>(a, st);
                        if (n == 2048) go<8, 1>(a, st);
                        if (n == 3072) go<12, 1>(a, st);
                        CK(hipStreamSynchronize(st));
                        unsigned long long t[3];
                        CK(hipMemcpy(t, log, 24, hipMemcpyDeviceToHost));
                        if (r >= 2) us += t[0] / 100.0 / (nphase - 4), rt += (double)t[1] / 100.0 / (double)t[2], ns += (double)t[2] / (nphase - 4);
                    }
                    int e[2];
                    CK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost));
                    printf("n %4d  work %4d + jitter %3d ns  %-20s delay %3d: %.3f us per phase (hand-off alone %.3f)  sweep rtt %.3f us, %.1f sweeps per phase  timeouts %d\n", n, work, jitter,
                           mname[mem], delay, us / (reps - 2), us / (reps - 2) - work / 1000.0 - jitter / 2000.0, rt / (reps - 2), ns / (reps - 2), e[0]);
                    fflush(stdout);
                    (void)depth;
                }
    return 0;
}
