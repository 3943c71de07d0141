// ub_xcd_stream.hip -- round 5: the aggregate ceiling of EIGHT XCD-confined streaming groups.
//
// 256 workgroups, one per CU; a workgroup learns its XCD from HW_REG_XCC_ID and takes a ticket there, so the 32 workgroups of an XCD form one
// group (placement-independent, as the engine does it).  Every group streams its own "token": NPH phases of PH_ROWS x 1024 4-bit weights
// (PackedQ 16-byte blocks + per-group bf16 step / zero: 8.9 MB per phase, ~780 MB per token -- the bytes of a Qwen3-0.6B decode step), and
// between two phases the group does the engine's hand-off INSIDE the XCD: every workgroup publishes 32 tagged granules of a 1024-granule
// vector with a plain store (the lines live in that XCD's L2), the poller wave sweeps the vector with sc1 loads until all tags match and
// stages it into LDS as the next phase's fp32 activations.
//
// Knobs (template / argv): compute waves per workgroup (7 + poller, 15 + poller), 16-byte loads in flight per lane (DEPTH), what runs on the
// stream (MODE 0: a xor of the words -- pure bandwidth; 1: the engine's canonical 4-bit unpack + two v_pk_fma_f32 chains per lane
// (BlockDotF<FMT_Q4P>) + lane tree), whether the 8 groups read the SAME weights (the real decoders share them: 545 of 779 MB) or private
// copies (the K/V share), hand-off on / off.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I koifish_amd/csrc -o scratch/ub_xcd_stream scratch/ub_xcd_stream.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kf_gemv_blocks.h"

using namespace kf;

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

#define KF_GLOBAL __attribute__((address_space(1)))
constexpr int K = 1024, NBLK = K / 32, PH_ROWS = 16384; /* 16384 x 1024 4-bit weights = 8.39 MB of blocks + 0.52 MB of step / zero per phase */
constexpr size_t PH_BLOCKS = (size_t)PH_ROWS * NBLK, PH_GROUPS = PH_BLOCKS / 4;
constexpr size_t PH_BYTES = PH_BLOCKS * 16 + PH_GROUPS * 4;

struct Args {
    const u32x4* w;        /* [nph][PH_BLOCKS] (+ group_stride per XCD group when private) */
    const uint16_t* zero;  /* [nph][PH_GROUPS] */
    const uint16_t* step;
    size_t grp_stride_blocks; /* 0: the 8 groups share the weights */
    uint32_t* loc;         /* per XCD: [0..31] tickets area (stride 256 dwords), then the hand-off vector [2][1024] granules */
    int* err;
    unsigned long long* log;
    int nph, ntok, epoch, handoff, ngroups, mixed; /* ngroups: XCDs 0 .. ngroups - 1 stream, the others leave at once; mixed: odd XCDs run the xor stream whatever MODE */
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 7u);
}

constexpr int LOC_STRIDE = 4096; /* dwords per XCD: [0] ticket, [1024 ..] two vectors of 1024 granules */

template <int NWV, int DEPTH, int MODE>
__global__ void __launch_bounds__(NWV * 64) xcd_stream_kernel(const Args a) {
    constexpr int NCW = NWV - 1, LPR = 32, RPS = 2;
    __shared__ __attribute__((aligned(16))) u32x4 xs[K / 4];   /* fp32 activations, chunk layout [8][NBLK] */
    __shared__ uint32_t outb[64];
    __shared__ int ids[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) {
        const int x = xcc_id();
        ids[0] = x, ids[1] = __hip_atomic_fetch_add(a.loc + (size_t)x * LOC_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = tid; i < K / 4; i += NWV * 64) xs[i] = u32x4{0x3f800000u, 0x3f000000u, 0xbf800000u, 0x3e800000u};
    __syncthreads();
    const int xcc = ids[0], r = ids[1] & 31;
    if (ids[1] >= 32 && tid == 0) atomicAdd(a.err, 1);
    if (xcc >= a.ngroups) {
        if (tid == 0 && r == 0) a.loc[(size_t)xcc * LOC_STRIDE] = 0;
        return;
    }
    const bool plain = MODE == 0 || (a.mixed && (xcc & 1));
    uint32_t* const vec = a.loc + (size_t)xcc * LOC_STRIDE + 1024;
    const u32x4 KF_GLOBAL* wbase = (const u32x4 KF_GLOBAL*)(a.w + (size_t)xcc * a.grp_stride_blocks);
    // this workgroup's rows of a phase: PH_ROWS / 32, contiguous; a slot = RPS rows x LPR lanes = 64 consecutive blocks (1 KB): wave cw takes slots cw, cw + NCW, ...
    constexpr int ROWS_WG = PH_ROWS / 32, SLOTS_WG = ROWS_WG / RPS;
    const int nmine = wave < NCW ? (SLOTS_WG - wave + NCW - 1) / NCW : 0;
    unsigned long long t0 = 0;
    float keep = 0.f;
    uint32_t keepx = 0;
    u32x4 buf[DEPTH];
    uint16_t bst[DEPTH], bze[DEPTH];
    auto issue = [&](int ph, int i, int d) {
        int ii = i < nmine ? i : (nmine > 0 ? nmine - 1 : 0);
        const size_t blk = (size_t)ph * PH_BLOCKS + ((size_t)r * SLOTS_WG + (size_t)(wave + ii * NCW)) * 64 + lane;
        buf[d] = __builtin_nontemporal_load(wbase + blk);
        if (MODE == 1) {
            const size_t g = blk >> 2;
            bst[d] = ((const uint16_t KF_GLOBAL*)a.step)[g], bze[d] = ((const uint16_t KF_GLOBAL*)a.zero)[g];
        }
    };
    if (wave < NCW) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) issue(0, d, d);
    }
    for (int tok = 0; tok < a.ntok; tok++) {
        for (int ph = 0; ph < a.nph; ph++) {
            const int gph = tok * a.nph + ph;
            const uint32_t tag = (uint32_t)(a.epoch * 4096 + gph + 1) & 0xffffu;
            if (gph == 2 && r == 0 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
            if (wave < NCW) {
                const int nph_next = ph + 1 < a.nph ? ph + 1 : 0;
                for (int i = 0; i < nmine; i += DEPTH) {
#pragma unroll
                    for (int d = 0; d < DEPTH; d++) {
                        const u32x4 w = buf[d];
                        const uint16_t st = bst[d], ze = bze[d];
                        // the block DEPTH slots ahead: past the end of this phase, the first blocks of the next one (they do not depend on the hand-off)
                        if (i + d + DEPTH < nmine) issue(ph, i + d + DEPTH, d);
                        else issue(nph_next, i + d + DEPTH - nmine, d);
                        if (i + d < nmine) {
                            if (plain) {
                                keepx ^= w.x ^ w.y ^ w.z ^ w.w;
                            } else {
                                const float s = bf2f(st);
                                f32x2_t acc = BlockDotF<FMT_Q4P>::run(w, reinterpret_cast<const f32x4*>(xs), lane & (LPR - 1), NBLK, s, bf2f(ze), -(8.0f * s), f32x2_t{0.f, 0.f});
                                const float v = group_sum(acc.x + acc.y, 5);
                                keep += v;
                            }
                        }
                    }
                }
                if (lane < 32 && wave == 0) outb[lane] = (tag << 16) | (!plain ? (uint32_t)f2bf(keep) : (keepx & 0xffffu)); /* the workgroup's piece */
            }
            __syncthreads(); /* the workgroup's rows are done */
            if (a.handoff) {
                uint32_t* const v = vec + (gph & 1) * 1024;
                if (wave == 0 && lane < 8) *reinterpret_cast<u32x4*>(v + r * 32 + 4 * lane) = *reinterpret_cast<const u32x4*>(outb + 4 * lane);
                if (wave == NWV - 1) { /* the poller: sweep the group's vector (4 KB, this XCD's L2) until every tag matches, stage it as fp32 chunks */
                    const __amdgpu_buffer_rsrc_t rs = rsrc(v, 4096);
                    const uint32_t tagw = tag << 16;
                    u32x4 g[4];
                    for (int spins = 0;; spins++) {
                        uint32_t bad = 0;
#pragma unroll
                        for (int q = 0; q < 4; q++) g[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (q * 64 + lane) * 16, 0, 16 /* sc1 */));
#pragma unroll
                        for (int q = 0; q < 4; q++) bad |= ((g[q].x ^ tagw) | (g[q].y ^ tagw) | (g[q].z ^ tagw) | (g[q].w ^ tagw)) & 0xffff0000u;
                        if (__all(bad == 0)) break;
                        if (spins > (1 << 18)) {
                            if (lane == 0) atomicAdd(a.err, 1);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) { /* element e0 = 4 (64 q + lane): chunk (e0 / 4) -> [j][c] with c = block, j = chunk inside the block */
                        const int ch = q * 64 + lane, c = ch >> 3, j = ch & 7;
                        // keep the values tame: the activations stay the constants of the start (only the traffic and the dependency are modelled)
                        xs[j * NBLK + c] = u32x4{0x3f800000u | (g[q].x & 1u), 0x3f000000u, 0xbf800000u, 0x3e800000u};
                    }
                }
                __syncthreads();
            }
        }
    }
    if (r == 0 && tid == 0) a.log[xcc] = __builtin_amdgcn_s_memrealtime() - t0; /* per group: ticks (10 ns) from its phase 2 to its end */
    if ((keep == 1.2345f || keepx == 0x12345u) && lane == 0) a.err[1] = 1;
    if (r == 0 && tid == 0) a.loc[(size_t)xcc * LOC_STRIDE] = 0; /* every workgroup of the group took its ticket long ago */
}

template <int NWV, int DEPTH, int MODE>
static void go(const Args& a, hipStream_t st) {
    hipLaunchKernelGGL((xcd_stream_kernel<NWV, DEPTH, MODE>), dim3(256), dim3(NWV * 64), 0, st, a);
}

int main(int argc, char** argv) {
    const int nph = 88, ntok = 3; /* 88 x 8.9 MB = 784 MB per group and token */
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    int* err;
    unsigned long long* log;
    uint32_t* loc;
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&log, 64));
    CK(hipMalloc(&loc, 8 * LOC_STRIDE * 4));
    const size_t blocks_grp = (size_t)nph * PH_BLOCKS, groups_grp = (size_t)nph * PH_GROUPS;
    u32x4* w;
    uint16_t *ze, *stp;
    CK(hipMalloc(&w, blocks_grp * 16 * 8)); /* 8 private copies: 6.3 GB */
    CK(hipMalloc(&ze, groups_grp * 2));
    CK(hipMalloc(&stp, groups_grp * 2));
    CK(hipMemset(w, 0x5a, blocks_grp * 16 * 8));
    {
        uint16_t* h = (uint16_t*)malloc(groups_grp * 2);
        for (size_t i = 0; i < groups_grp; i++) h[i] = 0x3c00 + (uint16_t)(i & 63); /* bf16 ~ 0.0078 .. */
        CK(hipMemcpy(stp, h, groups_grp * 2, hipMemcpyHostToDevice));
        for (size_t i = 0; i < groups_grp; i++) h[i] = 0x3a00 + (uint16_t)(i & 31);
        CK(hipMemcpy(ze, h, groups_grp * 2, hipMemcpyHostToDevice));
        free(h);
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int epoch = 1;
    printf("# 8 XCD-confined groups of 32 workgroups; per group and token: %d phases x %.2f MB = %.1f MB; %d tokens per launch\n", nph, PH_BYTES / 1e6, nph * PH_BYTES / 1e6, ntok);
    printf("# agg = active groups x bytes / time (algorithmic bytes; shared weights may be served by the memory-side cache); per-group times from the device clock\n");
    struct Run { int mode, priv, handoff, nw, dp, ngroups, mixed; };
    std::vector<Run> runs;
    for (int mode : {0, 1})
        for (int handoff : {0, 1})
            for (int cfg = 0; cfg < 5; cfg++) {
                static const int NW[5] = {8, 9, 13, 12, 16}, DP[5] = {8, 8, 6, 6, 6};
                runs.push_back({mode, 0, handoff, NW[cfg], DP[cfg], 8, 0});
            }
    for (int ng : {1, 2, 4}) runs.push_back({0, 1, 1, 16, 6, ng, 0}), runs.push_back({1, 1, 1, 16, 6, ng, 0});
    runs.push_back({1, 0, 1, 16, 6, 8, 1});
    runs.push_back({1, 1, 1, 16, 6, 8, 1});
    runs.push_back({1, 1, 1, 16, 6, 8, 0});
    runs.push_back({0, 1, 1, 16, 6, 8, 0});
    for (const Run& R : runs) {
        double best = 1e30;
        int e[2] = {0, 0};
        unsigned long long ticks[8] = {0};
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(err, 0, 64));
            CK(hipMemset(log, 0, 64));
            CK(hipMemset(loc, 0xff, 8 * LOC_STRIDE * 4));
            for (int i = 0; i < 8; i++) CK(hipMemset(loc + (size_t)i * LOC_STRIDE, 0, 4));
            Args a{w, ze, stp, R.priv ? blocks_grp : 0, loc, err, log, nph, ntok, epoch++, R.handoff, R.ngroups, R.mixed};
            CK(hipEventRecord(e0, st));
#define GO(N, D)                                 \
    if (R.nw == N && R.dp == D) {                \
        if (R.mode == 0) go<N, D, 0>(a, st);     \
        else go<N, D, 1>(a, st);                 \
    }
            GO(8, 8) GO(9, 8) GO(13, 6) GO(12, 6) GO(16, 6)
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) {
                best = ms;
                CK(hipMemcpy(ticks, log, 64, hipMemcpyDeviceToHost));
            }
            CK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost));
            if (e[0]) break;
        }
        const double bytes = (double)R.ngroups * ntok * nph * (R.mode == 1 && !R.mixed ? (double)PH_BYTES : (R.mixed ? ((double)PH_BYTES + (double)PH_BLOCKS * 16) / 2 : (double)PH_BLOCKS * 16));
        printf("%-20s %-7s hand-off %d groups %d%s waves %2d depth %2d: %8.3f ms per launch, %6.2f us per phase, agg %7.1f GB/s = %.3f of 8 TB/s  timeouts %d | per-group ms:", R.mode ? "q4 unpack + pk_fma" : "xor (bandwidth only)",
               R.priv ? "private" : "shared", R.handoff, R.ngroups, R.mixed ? " (odd XCDs xor)" : "", R.nw, R.dp, best, best * 1e3 / (ntok * nph), bytes / best / 1e6, bytes / best / 1e6 / 8000.0, e[0]);
        for (int i = 0; i < 8; i++) printf(" %.2f", ticks[i] / 1e5);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
