// ub_handoff.hip -- the floor of an all-to-all hand-off between resident workgroups on MI355X (256 CUs, 8 XCDs with private L2s).
//
// 256 workgroups (one per CU) run `nphase` dependent phases.  In every phase each workgroup publishes 4 granules (16 bytes: ONE sc1 store)
// of an n-granule vector, and every workgroup needs the whole vector before it may publish the next phase -- the communication pattern of
// the persistent decode engine (kf_engine.hip) with the arithmetic removed.  Variants of how the vector reaches the consumers:
//   0  every workgroup sweeps the global vector with sc1 loads (what kf_engine.hip does)
//   1  one leader per XCD (elected by XCC_ID ticket) sweeps the global vector and republishes it with PLAIN 16-byte stores into a buffer of
//      its XCD (the lines stay in that XCD's L2); the other workgroups of the XCD sweep that buffer with sc1 loads (L1 bypass, L2 hits)
//   2  as 1, but the followers' sweeps only start after a 4-byte "ready" word of the XCD buffer carries the phase tag (one load per poll)
//   3  no cross-XCD traffic at all: the workgroups of an XCD exchange an n-granule vector among themselves (rank r of the XCD publishes
//      granules [r*n/32, (r+1)*n/32) with plain or sc1 stores, everyone sweeps with sc1 loads that hit the XCD's L2): the price of a hand-off
//      that stays inside one XCD (what a tensor-parallel split of the layer over the 8 XCDs would pay for most of its phases)
// Reported: microseconds per phase (wall clock over all phases / nphase).
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ub_handoff scratch/ub_handoff.hip && scratch/ub_handoff
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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
__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 7u);
}

struct Args {
    uint32_t* glob;   // [2][n] granules (two buffers alternate)
    uint32_t* loc;    // [8][2][n + 32] per-XCD copies (+ a ready word on its own line)
    int* tickets;     // [8 * 32] one ticket word per XCD, 128 bytes apart
    int* err;
    unsigned long long* log;
    int n, nphase, mode, epoch, work_ns;
};

template <int NLD>
__device__ __forceinline__ bool sweep(const uint32_t* src, int n, uint32_t tag, u32x4 (&g)[NLD], int lane, int* err) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, (uint32_t)n * 4u);
    for (int spins = 0;; spins++) {
        uint32_t bad = 0;
#pragma unroll
        for (int r = 0; r < NLD; r++) g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
#pragma unroll
        for (int r = 0; r < NLD; r++) bad |= tags_bad(g[r], tag);
        if (__all(bad == 0)) return true;
        if (spins > (1 << 16)) {
            if (lane == 0) atomicAdd(err, 1);
            return false;
        }
    }
}

// one wave per workgroup does the communication (the engine's poller); NLD = n / 256
template <int NLD>
__global__ void __launch_bounds__(64) handoff_kernel(const Args a) {
    const int lane = threadIdx.x, wg = blockIdx.x, n = a.n;
    const int xcc = xcc_id();
    // leader election: first ticket of the XCD (tickets are re-zeroed by the host between launches)
    int rank = 0;
    if (lane == 0) rank = __hip_atomic_fetch_add(a.tickets + xcc * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rank = __builtin_amdgcn_readfirstlane(rank);
    const bool leader = rank == 0;
    uint32_t* const myloc = a.loc + (size_t)xcc * 2 * (n + 32);
    unsigned long long t0 = 0;
    for (int p = 0; p < a.nphase; p++) {
        const uint32_t tag = (uint32_t)(a.epoch * 1024 + p + 1) & 0xffffu;
        uint32_t* const gbuf = a.glob + (size_t)(p & 1) * n;
        uint32_t* const lbuf = myloc + (size_t)(p & 1) * (n + 32);
        // publish this workgroup's 4 granules (values: something that depends on the previous phase so nothing is hoisted)
        if (a.mode < 3 && lane < n / 1024) { /* n / 256 granules per workgroup, 16 bytes per lane */
            const uint32_t v = (uint32_t)(wg * 7 + p + lane) & 0xfff0u;
            u32x4 o = {(tag << 16) | v, (tag << 16) | (v + 1), (tag << 16) | (v + 2), (tag << 16) | (v + 3)};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrc(gbuf + wg * (n / 256) + 4 * lane, 16), 0, 0, 16 /* sc1 */);
        }
        if (p == 4 && wg == 0 && lane == 0) t0 = __builtin_amdgcn_s_memrealtime();
        u32x4 g[NLD];
        if (a.mode >= 3) { /* intra-XCD all-to-all: rank publishes n/32 granules into the XCD buffer (mode 3 plain stores, mode 4 sc1 stores) */
            const int per = n / 32; /* granules per rank: 32 (n 1024) or 96 (n 3072) */
            if (rank < 32 && 4 * lane < per) {
                const uint32_t v = (uint32_t)(wg * 7 + p + lane) & 0xfff0u;
                u32x4 o = {(tag << 16) | v, (tag << 16) | (v + 1), (tag << 16) | (v + 2), (tag << 16) | (v + 3)};
                if (a.mode == 3)
                    *reinterpret_cast<u32x4*>(lbuf + rank * per + 4 * lane) = o;
                else
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsrc(lbuf + rank * per + 4 * lane, 16), 0, 0, 16 /* sc1 */);
            }
            sweep<NLD>(lbuf, n, tag, g, lane, a.err);
        } else if (a.mode == 0 || leader) {
            sweep<NLD>(gbuf, n, tag, g, lane, a.err);
            if (a.mode != 0) { /* republish into this XCD's L2: plain stores keep the lines there */
#pragma unroll
                for (int r = 0; r < NLD; r++) *reinterpret_cast<u32x4*>(lbuf + (r * 64 + lane) * 4) = g[r];
                if (a.mode == 2) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(lbuf + n, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } else {
            if (a.mode == 2) {
                for (int spins = 0;; spins++) {
                    const uint32_t f = __hip_atomic_load(lbuf + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (f == tag) break;
                    if (spins > (1 << 20)) {
                        if (lane == 0) atomicAdd(a.err, 1);
                        break;
                    }
                }
            }
            sweep<NLD>(lbuf, n, tag, g, lane, a.err);
        }
        // "compute": a fixed delay standing in for a phase's arithmetic
        if (a.work_ns > 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime() + (unsigned long long)a.work_ns / 10;
            while (__builtin_amdgcn_s_memrealtime() < t1) __builtin_amdgcn_s_sleep(1);
        }
        // keep the swept values alive
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < NLD; r++) acc += g[r].x + g[r].w;
        if (acc == 0x12345678u && lane == 0) a.err[1] = 1;
    }
    if (wg == 0 && lane == 0) a.log[0] = __builtin_amdgcn_s_memrealtime() - t0;
}

int main(int argc, char** argv) {
    const int nphase = 404, reps = 20;
    int *tickets, *err;
    uint32_t *glob, *loc;
    unsigned long long* log;
    CK(hipMalloc(&tickets, 8 * 32 * 4));
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&log, 64));
    CK(hipMalloc(&glob, 2 * 4096 * 4));
    CK(hipMalloc(&loc, 8 * 2 * (4096 + 32) * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int epoch = 1;
    for (int n : {1024, 3072}) {
        for (int work : {0, 1000}) {
            for (int mode = 0; mode < 5; mode++) {
                CK(hipMemset(err, 0, 64));
                CK(hipMemset(glob, 0xff, 2 * 4096 * 4));
                CK(hipMemset(loc, 0xff, 8 * 2 * (4096 + 32) * 4));
                double us = 0;
                for (int r = 0; r < reps; r++) {
                    CK(hipMemsetAsync(tickets, 0, 8 * 32 * 4, st));
                    Args a{glob, loc, tickets, err, log, n, nphase, mode, epoch++, work};
                    if (n == 1024)
                        hipLaunchKernelGGL(handoff_kernel<4>, dim3(256), dim3(64), 0, st, a);
                    else
                        hipLaunchKernelGGL(handoff_kernel<12>, dim3(256), dim3(64), 0, st, a);
                    CK(hipStreamSynchronize(st));
                    unsigned long long t;
                    CK(hipMemcpy(&t, log, 8, hipMemcpyDeviceToHost));
                    if (r >= 2) us += t / 100.0 / (nphase - 4);
                }
                int e[2];
                CK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost));
                printf("n %4d  work %4d ns  mode %d (%s): %.3f us per phase  (hand-off alone %.3f)   timeouts %d\n", n, work, mode,
                       mode == 0 ? "all 256 sweep the fabric" : (mode == 1 ? "XCD leaders sweep + republish in L2" : (mode == 2 ? "leaders + ready word" : (mode == 3 ? "inside each XCD, plain stores" : "inside each XCD, sc1 stores"))), us / (reps - 2),
                       us / (reps - 2) - work / 1000.0, e[0]);
                fflush(stdout);
            }
        }
    }
    return 0;
}
