// ub_handoff2.hip -- hand-off skeleton of ONE decode layer in two organisations, arithmetic removed (256 resident workgroups, one poller wave each):
//   mode 0  the engine of kf_engine.hip: six all-to-all edges per layer -- x 4 KB, qkv (to the kv-head's workgroups: modelled as intra-XCD 2 KB),
//           partials->merge (intra 2.5 KB), ao 8 KB all-to-all, x 4 KB all-to-all, act 12 KB all-to-all
//   mode 1  the layer split over the 8 XCDs (kv-head c, q heads 2c..2c+1, ffn rows 384c.. on XCD c; o_proj / down_proj as K-slices with a cross-XCD
//           reduce): x 4 KB all-to-all, qkv intra 2 KB, partials intra 2.5 KB, ao-slice intra 1 KB, reduce-scatter (each workgroup reads 8 x 32 B
//           of fp32 partial granules written by 8 XCDs), x 4 KB all-to-all, act-slice intra 1.5 KB, reduce-scatter
// Intra-XCD vectors: plain 16-byte stores into a per-XCD buffer + sc1 loads (they hit that XCD's L2); cross-XCD: sc1 stores + sc1 loads.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ub_handoff2 scratch/ub_handoff2.hip && scratch/ub_handoff2
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
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ uint32_t tags_bad(u32x4 g, uint32_t tag) { return ((g.x >> 16) ^ tag) | ((g.y >> 16) ^ tag) | ((g.z >> 16) ^ tag) | ((g.w >> 16) ^ tag); }
__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 7u);
}

struct Args {
    uint32_t* glob;  // [2][4096] granules: all-to-all vectors (ping-pong)
    uint32_t* loc;   // [8][2][4096]: per-XCD vectors
    uint32_t* part;  // [2][8][2048] 8-byte granules {f32, tag} as pairs of dwords: reduce-scatter partials
    int* tickets;
    int* err;
    unsigned long long* log;
    int nlayer, mode, epoch;
};

// sweep n granules (n a multiple of 256, <= 3072) of a vector until every tag matches; `sc1` loads
__device__ __forceinline__ void sweep(const uint32_t* src, int n, uint32_t tag, int lane, int* err, uint32_t& acc) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, (uint32_t)n * 4u);
    const int nld = n / 256;
    for (int spins = 0;; spins++) {
        asm volatile("" ::: "memory"); /* the poll loads must be re-issued every pass */
        uint32_t bad = 0;
        u32x4 g[12];
#pragma unroll
        for (int r = 0; r < 12; r++)
            if (r < nld) g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
#pragma unroll
        for (int r = 0; r < 12; r++)
            if (r < nld) bad |= tags_bad(g[r], tag), acc += g[r].x;
        if (__all(bad == 0)) return;
        if (spins > (1 << 16)) {
            if (lane == 0) atomicAdd(err, 1);
            if (lane == 0 && err[4] == 0) err[4] = 1, err[5] = (int)g[0].x, err[6] = (int)tag, err[7] = n, err[8] = (int)bad, err[9] = (int)g[nld - 1].w;
            return;
        }
    }
}
// publish `cnt` granules (a multiple of 4) starting at granule `at`: lanes < cnt / 4 store 16 bytes each
__device__ __forceinline__ void publish(uint32_t* dst, int at, int cnt, uint32_t tag, int lane, bool plain, uint32_t v) {
    if (4 * lane < cnt) {
        u32x4 o = {(tag << 16) | (v & 0xffffu), (tag << 16) | ((v + 1) & 0xffffu), (tag << 16) | ((v + 2) & 0xffffu), (tag << 16) | ((v + 3) & 0xffffu)};
        if (plain)
            *reinterpret_cast<u32x4*>(dst + at + 4 * lane) = o;
        else
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrc(dst + at + 4 * lane, 16), 0, 0, 16 /* sc1 */);
    }
}

__global__ void __launch_bounds__(64) layer_kernel(const Args a) {
    const int lane = threadIdx.x, wg = blockIdx.x;
    const int xcc = xcc_id();
    int rank = 0;
    if (lane == 0) rank = __hip_atomic_fetch_add(a.tickets + xcc * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rank = __builtin_amdgcn_readfirstlane(rank);
    if (rank >= 32) {
        if (lane == 0) atomicAdd(a.err + 2, 1); /* more than 32 workgroups landed on one XCD */
        rank &= 31;
    }
    const int gid = xcc * 32 + rank; /* position in the XCD-major order */
    uint32_t* const myloc = a.loc + (size_t)xcc * 2 * 4096;
    unsigned long long t0 = 0;
    uint32_t acc = 0;
    int ph = 0; /* running phase counter -> tag and ping-pong slot */
    auto all2all = [&](int n) { /* every workgroup publishes n / 256 granules (sc1), every workgroup sweeps n */
        const uint32_t tag = (uint32_t)(a.epoch * 4096 + ph + 1) & 0xffffu;
        uint32_t* buf = a.glob + (size_t)(ph & 1) * 4096;
        publish(buf, wg * (n / 256), n / 256, tag, lane, false, acc);
        sweep(buf, n, tag, lane, a.err, acc);
        ph++;
    };
    auto intra = [&](int n) { /* n granules per XCD, n / 32 per workgroup (plain stores), swept with sc1 loads: n a multiple of 256 here */
        const uint32_t tag = (uint32_t)(a.epoch * 4096 + ph + 1) & 0xffffu;
        uint32_t* buf = myloc + (size_t)(ph & 1) * 4096;
        publish(buf, rank * (n / 32), n / 32, tag, lane, true, acc);
        sweep(buf, n, tag, lane, a.err, acc);
        ph++;
    };
    auto reduce_scatter = [&]() { /* XCD c, rank r publishes 32 fp32 partial rows (8-byte granules); workgroup gid sums rows 4 gid .. 4 gid + 3 over the 8 XCDs */
        const uint32_t tag = (uint32_t)(a.epoch * 4096 + ph + 1);
        uint32_t* buf = a.part + (size_t)(ph & 1) * 8 * 2048;
        if (lane < 16) { /* 32 rows x 8 B = 256 B: 16 lanes x 16 B */
            u32x4 o = {acc + lane, tag, acc + lane + 1, tag};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrc(buf + (size_t)xcc * 2048 + rank * 64 + 4 * lane, 16), 0, 0, 16);
        }
        /* lane = (c, half): c = lane >> 1 < 8, half = lane & 1: 16 bytes = two row granules */
        for (int spins = 0;; spins++) {
            asm volatile("" ::: "memory");
            uint32_t bad = 0;
            u32x4 g = {0, tag, 0, tag};
            if (lane < 16) g = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(buf + (size_t)(lane >> 1) * 2048 + gid * 8 + 4 * (lane & 1), 16), 0, 0, 16));
            bad = (g.y ^ tag) | (g.w ^ tag);
            acc += g.x;
            if (__all(bad == 0)) break;
            if (spins > (1 << 16)) {
                if (lane == 0) atomicAdd(a.err, 1);
                break;
            }
        }
        ph++;
    };
    for (int l = 0; l < a.nlayer; l++) {
        if (l == 2 && wg == 0 && lane == 0) t0 = __builtin_amdgcn_s_memrealtime();
        if (a.mode == 0) {
            all2all(1024);  /* x -> P1 */
            intra(512);     /* q,k,v -> attention units of the kv head */
            intra(768);     /* slice partials -> merge */
            all2all(2048);  /* ao -> o_proj */
            all2all(1024);  /* x -> P5 */
            all2all(3072);  /* act -> down_proj */
        } else {
            all2all(1024);  /* x -> P1 */
            intra(512);     /* q,k,v of the XCD's kv head */
            intra(768);     /* slice partials -> merge */
            intra(256);     /* ao slice (2 heads) -> o_proj K-slice */
            reduce_scatter();
            all2all(1024);  /* x -> P5 */
            intra(512);     /* act slice (384 values, padded) -> down_proj K-slice */
            reduce_scatter();
        }
    }
    if (wg == 0 && lane == 0) a.log[0] = __builtin_amdgcn_s_memrealtime() - t0;
    if (acc == 0x12345678u && lane == 0) a.err[1] = 1;
}

int main() {
    const int nlayer = 58, reps = 20;
    int *tickets, *err;
    uint32_t *glob, *loc, *part;
    unsigned long long* log;
    CK(hipMalloc(&tickets, 8 * 32 * 4));
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&log, 64));
    CK(hipMalloc(&glob, 2 * 4096 * 4));
    CK(hipMalloc(&loc, 8 * 2 * 4096 * 4));
    CK(hipMalloc(&part, 2 * 8 * 2048 * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int epoch = 1;
    for (int mode = 0; mode < 2; mode++) {
        CK(hipMemset(err, 0, 64));
        CK(hipMemset(glob, 0xff, 2 * 4096 * 4));
        CK(hipMemset(loc, 0xff, 8 * 2 * 4096 * 4));
        CK(hipMemset(part, 0xff, 2 * 8 * 2048 * 4));
        double us = 0;
        for (int r = 0; r < reps; r++) {
            CK(hipMemsetAsync(tickets, 0, 8 * 32 * 4, st));
            Args a{glob, loc, part, tickets, err, log, nlayer, mode, epoch++};
            hipLaunchKernelGGL(layer_kernel, dim3(256), dim3(64), 0, st, a);
            CK(hipStreamSynchronize(st));
            unsigned long long t;
            CK(hipMemcpy(&t, log, 8, hipMemcpyDeviceToHost));
            if (r >= 2) us += t / 100.0 / (nlayer - 2);
        }
        int e[10];
        CK(hipMemcpy(e, err, 40, hipMemcpyDeviceToHost));
        if (e[4]) printf("  first timeout: g0.x %08x tag %04x n %d bad %08x last.w %08x\n", e[5], e[6], e[7], e[8], e[9]);
        printf("mode %d (%s): %.2f us of hand-offs per layer   timeouts %d, XCD overflow %d\n", mode,
               mode == 0 ? "six edges, four of them all-to-all (the engine today)" : "layer split over the XCDs: two all-to-all, four intra-XCD, two reduce-scatters", us / (reps - 2), e[0], e[2]);
    }
    return 0;
}
