// kf_engine_common.h -- what the two persistent decode engines share: kf_engine.hip (one sequence on all 256 CUs, weights in registers ahead of every hand-off) and
// kf_xengine.hip (round 5: eight independent sequences, one per XCD, every hand-off through that XCD's L2, weights streamed): the device tables, the tagged-granule
// accessors, and the mat-vec phase geometry as compile-time types (the lanes per row, rows per wave step and steps per row gemv_launch picks for the same matrices: the
// canonical summation order is a property of the matrix shape, not of how many workgroups share the rows).
#pragma once
#include <stdint.h>

#include "kf_attn_common.h"
#include "kf_gemv_blocks.h"

namespace kf {

constexpr int ENG_MAXLD = 24;             /* 1 KiB granule pieces (256 values) per sweep: vectors up to 6144 (Qwen3-1.7B's ffn) */
constexpr int ENG_NWG = 256, ENG_NWV = 8; /* MI355X: 256 CUs, one 8-wave workgroup each (7 compute waves + the poller) */
#ifndef KF_SWEEP_SLEEP
#define KF_SWEEP_SLEEP 1 /* s_sleep units between two sweeps of a hand-off vector that was not whole yet */
#endif
constexpr int ENG_SPIN_MAX = 1 << 17;     /* sweeps before a poll gives up (~0.1 s): sets the error word, never hangs */

// device tables hold GLOBAL pointers (address space 1): read back from LDS they would otherwise be generic, and every access through them a
// flat_load, which the hardware returns out of order, so every wait behind one is a drain (vmcnt(0) + lgkmcnt(0))
#define KF_GLOBAL __attribute__((address_space(1)))
typedef const u32x4 KF_GLOBAL* g_u32x4;
typedef const uint16_t KF_GLOBAL* g_u16;
typedef uint16_t KF_GLOBAL* g_u16w;
typedef const int32_t KF_GLOBAL* g_i32;
struct EngMat {
    g_u32x4 w;
    g_u16 zero;
    g_u16 step;
};
struct EngLayer {
    EngMat m[7]; /* q k v o gate up down */
    g_u16 norm_in, norm_post, norm_q, norm_k;
    g_u16w kcache, vcache; /* layer base */
    g_i32 hot;             /* sparse forward: CS_Picker's hot[ffn] (1 = the gate / up row is computed, D_matmul_sparse); NULL: dense */
};
struct EngPlan { /* host copy of one mat-vec phase: the geometry gemv_launch would pick for the same matrices (checked against PlanT) */
    int K, nBlk, lpr_log2, iters, gshift, njobs;
    int M[3], slot0[3], qBias[3];
    int total_slots, spg; /* slots per workgroup (contiguous) */
    int pad_[3];
};

constexpr int eng_gran_dw(int n) { return ((n * 4 + 255) & ~255) / 4; }
constexpr int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t eng_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void st_gran(uint32_t* p, uint32_t tag16, uint16_t v) {
    __hip_atomic_store(p, (tag16 << 16) | (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int eng_xcc() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 7u);
}
__device__ __forceinline__ void st_gran64(unsigned long long* p, uint32_t gen, float v) {
    __hip_atomic_store(p, ((unsigned long long)gen << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_gran64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// tag mismatch bits of one 16-byte piece (4 granules), accumulated in a vector register (tagw = tag << 16): as compares into lane masks the
// compiler issues a v_cmp + s_and pair per granule, each a VALU -> SALU hand-over
__device__ __forceinline__ uint32_t tags_bad(u32x4 g, uint32_t tagw, uint32_t bad) {
    bad |= (g.x ^ tagw) & 0xffff0000u;
    bad |= (g.y ^ tagw) & 0xffff0000u;
    bad |= (g.z ^ tagw) & 0xffff0000u;
    bad |= (g.w ^ tagw) & 0xffff0000u;
    return bad;
}
__device__ __forceinline__ bool all_good(uint32_t bad) {
    asm volatile("" : "+v"(bad)); /* keep the accumulated word opaque: one compare per sweep */
    return __all(bad == 0);
}

// wait until this workgroup's own counter `pub` has reached `want` (its rows of the feeding phase are on their way), then `delay` sleep units
__device__ __forceinline__ void eng_wait_pub(const int* pub, int want, int delay, bool& dead) {
    if (pub) {
        for (int spins = 0; spins < (1 << 22) && !dead; spins++) {
            if (__hip_atomic_load(pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    for (int z = 0; z < delay; z++) __builtin_amdgcn_s_sleep(1);
}

// ---- poller: sweep the NLD * 256 granules of a vector until every tag matches, then stage the vector into LDS: xs as fp32 in the engine's
// chunk layout [XCH][nBlk] (element e = c*EPB + j*4 + i -> chunk j*nBlk + c; F32X = false: bf16 chunks of 8, the mat-vec kernel's layout), optionally RMS-normalised (rms_norm_kernel,
// layernorm.cuh:800-847: fp64 sum of squares, (x*mul)*w, one bf16 store), and the raw vector in natural order into xraw (the residual
// the phase after next adds).  Lane l of load r owns elements 4*(64r + l) .. +3.  PLAIN: the vector is plain bf16 written by an
// earlier launch (layer 0's embedding row).  Straight-line code: the vector lengths are template parameters of the kernel.
template <int XCH, int NLD, int NBLK, bool NORM, bool PLAIN, bool F32X = true>
__device__ __forceinline__ void eng_poll_stage(const uint32_t* gsrc, const uint16_t* plain, uint32_t tag, g_u16 norm_w, float eps, u32x4* xs, uint16_t* xraw, int lane, int* ws,
                                               bool& dead, int* nsweeps, const int* pub, int want, int delay) {
    constexpr int n = NLD * 256;
    uint32_t p0[NLD], p1[NLD], w0[NLD], w1[NLD];
    if (NORM) { /* constants: requested in front of the sweep */
#pragma unroll
        for (int r = 0; r < NLD; r++) {
            const u32x2 t = *reinterpret_cast<const u32x2 KF_GLOBAL*>(norm_w + 4 * (r * 64 + lane));
            w0[r] = t.x, w1[r] = t.y;
        }
    }
    if (PLAIN) {
#pragma unroll
        for (int r = 0; r < NLD; r++) {
            const u32x2 t = *reinterpret_cast<const u32x2*>(plain + 4 * (r * 64 + lane));
            p0[r] = t.x, p1[r] = t.y;
        }
    } else {
        const __amdgpu_buffer_rsrc_t rs = eng_rsrc(gsrc, (uint32_t)n * 4u);
        u32x4 g[NLD];
        const uint32_t tagw = tag << 16;
        eng_wait_pub(pub, want, delay, dead);
        for (int spins = 0;; spins++) {
            uint32_t bad = 0;
#pragma unroll
            for (int r = 0; r < NLD; r++) g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
#pragma unroll
            for (int r = 0; r < NLD; r++) bad = tags_bad(g[r], tagw, bad);
            if (all_good(bad)) {
                if (nsweeps) *nsweeps = spins + 1;
                break;
            }
            if (dead || spins > ENG_SPIN_MAX) {
                if (!dead && lane == 0) atomicOr(ws + 1, 1);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
        }
#pragma unroll
        for (int r = 0; r < NLD; r++) p0[r] = (g[r].x & 0xffffu) | (g[r].y << 16), p1[r] = (g[r].z & 0xffffu) | (g[r].w << 16);
    }
    float mul = 1.0f;
    if (NORM) {
        double ss = 0.0;
#pragma unroll
        for (int r = 0; r < NLD; r++) {
            const double a = (double)bf_lo(p0[r]), b = (double)bf_hi(p0[r]), c = (double)bf_lo(p1[r]), d = (double)bf_hi(p1[r]);
            ss = fma(a, a, ss), ss = fma(b, b, ss), ss = fma(c, c, ss), ss = fma(d, d, ss);
        }
        const double tot = wave_sum_f64_fast(ss);
        mul = 1.0f / sqrtf(fmaf((float)tot, 1.0f / (float)n, eps));
    }
#pragma unroll
    for (int r = 0; r < NLD; r++) {
        const int e0 = 4 * (r * 64 + lane);
        uint32_t o0 = p0[r], o1 = p1[r];
        if (xraw) *reinterpret_cast<u32x2*>(xraw + e0) = u32x2{o0, o1};
        if (NORM) {
            o0 = pack_bf16x2((bf_lo(o0) * mul) * bf_lo(w0[r]), (bf_hi(o0) * mul) * bf_hi(w0[r]));
            o1 = pack_bf16x2((bf_lo(o1) * mul) * bf_lo(w1[r]), (bf_hi(o1) * mul) * bf_hi(w1[r]));
        }
        if (F32X) { /* the mat-vec phases multiply fp32 activations (BlockDotF): chunk = these 4 elements as floats, [XCH chunks per block][NBLK] */
            const int q = e0 >> 2, c = q / XCH, j = q - c * XCH;
            xs[j * NBLK + c] = u32x4{o0 << 16, o0 & 0xffff0000u, o1 << 16, o1 & 0xffff0000u};
        } else {
            const int q = e0 >> 3, c = q / XCH, j = q - c * XCH;
            reinterpret_cast<u32x2*>(xs + j * NBLK + c)[(e0 >> 2) & 1] = u32x2{o0, o1};
        }
    }
}

// a long vector without a norm (attention output, SwiGLU output), PIECE x 256 granules at a time: every piece swept until whole, then staged -- the sweep's registers are
// those of one piece (a 9728-wide vector in one sweep: 152 registers)
template <int XCH, int NLD, int NBLK, int PIECE, int R0 = 0>
__device__ __forceinline__ void eng_poll_stage_long(const uint32_t* gsrc, uint32_t tag, u32x4* xs, int lane, int* ws, bool& dead, int* nsweeps) {
    if constexpr (R0 < NLD) {
        constexpr int NP = (NLD - R0) < PIECE ? (NLD - R0) : PIECE;
        const __amdgpu_buffer_rsrc_t rs = eng_rsrc(gsrc + (size_t)R0 * 256, (uint32_t)NP * 1024u);
        u32x4 g[NP];
        const uint32_t tagw = tag << 16;
        for (int spins = 0;; spins++) {
            uint32_t bad = 0;
#pragma unroll
            for (int r = 0; r < NP; r++) g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
#pragma unroll
            for (int r = 0; r < NP; r++) bad = tags_bad(g[r], tagw, bad);
            if (all_good(bad)) {
                if (nsweeps && R0 == 0) *nsweeps = spins + 1;
                break;
            }
            if (dead || spins > ENG_SPIN_MAX) {
                if (!dead && lane == 0) atomicOr(ws + 1, 1);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
        }
#pragma unroll
        for (int r = 0; r < NP; r++) {
            const int e0 = 4 * ((R0 + r) * 64 + lane), q = e0 >> 2, c = q / XCH, j = q - c * XCH;
            const uint32_t o0 = (g[r].x & 0xffffu) | (g[r].y << 16), o1 = (g[r].z & 0xffffu) | (g[r].w << 16);
            xs[j * NBLK + c] = u32x4{o0 << 16, o0 & 0xffff0000u, o1 << 16, o1 & 0xffff0000u};
        }
        eng_poll_stage_long<XCH, NLD, NBLK, PIECE, R0 + NP>(gsrc, tag, xs, lane, ws, dead, nsweeps);
    }
}

// a long vector WITH the RMS norm (a 5120-wide residual stream: the one-sweep form would hold 160 registers): pass 1 sweeps PIECE x 256 granules at a time (or reads the
// plain bf16 row), leaves the raw vector in xraw (natural order, required) and sums the squares in the lane order of eng_poll_stage (load r = 0, 1, ... : the same fp64 sum);
// pass 2 reads xraw back, normalises and stages (fp32 chunks, or bf16 chunks of 8: F32X = false)
template <int XCH, int NLD, int NBLK, bool PLAIN, bool F32X, int PIECE, int R0 = 0>
__device__ __forceinline__ void eng_norm_long_pass1(const uint32_t* gsrc, const uint16_t* plain, uint32_t tag, uint16_t* xraw, int lane, int* ws, bool& dead, double& ss) {
    if constexpr (R0 < NLD) {
        constexpr int NP = (NLD - R0) < PIECE ? (NLD - R0) : PIECE;
        uint32_t p0[NP], p1[NP];
        if constexpr (PLAIN) {
#pragma unroll
            for (int r = 0; r < NP; r++) {
                const u32x2 t = *reinterpret_cast<const u32x2*>(plain + 4 * ((R0 + r) * 64 + lane));
                p0[r] = t.x, p1[r] = t.y;
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = eng_rsrc(gsrc + (size_t)R0 * 256, (uint32_t)NP * 1024u);
            u32x4 g[NP];
            const uint32_t tagw = tag << 16;
            for (int spins = 0;; spins++) {
                uint32_t bad = 0;
#pragma unroll
                for (int r = 0; r < NP; r++) g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
#pragma unroll
                for (int r = 0; r < NP; r++) bad = tags_bad(g[r], tagw, bad);
                if (all_good(bad)) break;
                if (dead || spins > ENG_SPIN_MAX) {
                    if (!dead && lane == 0) atomicOr(ws + 1, 1);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
            }
#pragma unroll
            for (int r = 0; r < NP; r++) p0[r] = (g[r].x & 0xffffu) | (g[r].y << 16), p1[r] = (g[r].z & 0xffffu) | (g[r].w << 16);
        }
#pragma unroll
        for (int r = 0; r < NP; r++) {
            const double a = (double)bf_lo(p0[r]), b = (double)bf_hi(p0[r]), c = (double)bf_lo(p1[r]), d = (double)bf_hi(p1[r]);
            ss = fma(a, a, ss), ss = fma(b, b, ss), ss = fma(c, c, ss), ss = fma(d, d, ss);
            *reinterpret_cast<u32x2*>(xraw + 4 * ((R0 + r) * 64 + lane)) = u32x2{p0[r], p1[r]};
        }
        eng_norm_long_pass1<XCH, NLD, NBLK, PLAIN, F32X, PIECE, R0 + NP>(gsrc, plain, tag, xraw, lane, ws, dead, ss);
    }
}
template <int XCH, int NLD, int NBLK, bool PLAIN, bool F32X, int PIECE>
__device__ __forceinline__ void eng_poll_stage_norm_long(const uint32_t* gsrc, const uint16_t* plain, uint32_t tag, g_u16 norm_w, float eps, u32x4* xs, uint16_t* xraw, int lane, int* ws, bool& dead) {
    constexpr int n = NLD * 256;
    u32x2 wn[NLD]; /* the norm weights: requested in front of the sweep (inside pass 2 each batch of them would be a round trip to the L2 on the hand-off's critical path) */
#pragma unroll
    for (int r = 0; r < NLD; r++) wn[r] = *reinterpret_cast<const u32x2 KF_GLOBAL*>(norm_w + 4 * (r * 64 + lane));
    double ss = 0.0;
    eng_norm_long_pass1<XCH, NLD, NBLK, PLAIN, F32X, PIECE>(gsrc, plain, tag, xraw, lane, ws, dead, ss);
    const double tot = wave_sum_f64_fast(ss);
    const float mul = 1.0f / sqrtf(fmaf((float)tot, 1.0f / (float)n, eps));
#pragma unroll
    for (int r = 0; r < NLD; r++) {
        const int e0 = 4 * (r * 64 + lane);
        const u32x2 raw = *reinterpret_cast<const u32x2*>(xraw + e0); /* this lane's own words of pass 1 */
        const u32x2 w = wn[r];
        const uint32_t o0 = pack_bf16x2((bf_lo(raw.x) * mul) * bf_lo(w.x), (bf_hi(raw.x) * mul) * bf_hi(w.x));
        const uint32_t o1 = pack_bf16x2((bf_lo(raw.y) * mul) * bf_lo(w.y), (bf_hi(raw.y) * mul) * bf_hi(w.y));
        if (F32X) {
            const int q = e0 >> 2, c = q / XCH, j = q - c * XCH;
            xs[j * NBLK + c] = u32x4{o0 << 16, o0 & 0xffff0000u, o1 << 16, o1 & 0xffff0000u};
        } else {
            const int q = e0 >> 3, c = q / XCH, j = q - c * XCH;
            reinterpret_cast<u32x2*>(xs + j * NBLK + c)[(e0 >> 2) & 1] = u32x2{o0, o1};
        }
    }
}

struct MvAt {
    int row, col;
    bool ok;
};
// ---- mat-vec phase geometry: compile-time constants of the model shape (the same lanes per row, rows per wave step and steps per row
// gemv_launch picks for these matrices: engine_build checks the two against each other), so a wave's loads are unconditional, their number
// is static and the compiler can wait with counted vmcnt(N) instead of draining.
constexpr int c_lpr_log2(int nBlk, long rows) { /* = gemv_lpr_log2 (kf_gemv.hip) */
    int l = 6;
    while (l > 0 && (nBlk % (1 << l)) != 0) l--;
    if ((1 << l) < 16) {
        l = 6;
        while ((1 << l) > nBlk) l--;
    }
    while (l < 6 && nBlk > (1 << l) && (rows << l) / 64 < 1024 && (nBlk + (2 << l) - 1) / (2 << l) < (nBlk + (1 << l) - 1) / (1 << l)) l++;
    return l;
}
struct CPlan {
    int K, nBlk, lpr_log2, iters, njobs, M[3], slot0[3], total, spg;
};
constexpr CPlan c_plan(int K, int epb, int m0, int m1, int m2, bool paired, int nwg) {
    CPlan P{};
    P.K = K, P.nBlk = K / epb;
    const long rows = (long)m0 + (paired ? 0 : m1 + m2);
    P.lpr_log2 = c_lpr_log2(P.nBlk, rows);
    const int LPR = 1 << P.lpr_log2, RPS = 64 / LPR;
    P.iters = (P.nBlk + LPR - 1) / LPR;
    P.njobs = paired ? 1 : (m2 > 0 ? 3 : (m1 > 0 ? 2 : 1));
    P.M[0] = m0, P.M[1] = m1, P.M[2] = m2;
    P.slot0[0] = 0;
    P.slot0[1] = (m0 + RPS - 1) / RPS;
    P.slot0[2] = P.slot0[1] + (paired ? 0 : (m1 + RPS - 1) / RPS);
    P.total = paired ? P.slot0[1] : P.slot0[2] + (m2 + RPS - 1) / RPS;
    if (P.njobs < 3) P.slot0[2] = 0x7fffffff;
    if (P.njobs < 2) P.slot0[1] = 0x7fffffff;
    P.spg = (P.total + nwg - 1) / nwg;
    return P;
}
// the plan as a TYPE: every figure a static constant (no object, nothing to index at run time)
template <int K_, int EPB_, int M0_, int M1_, int M2_, bool PAIRED_, int NWG_>
struct PlanT {
    static constexpr int K = K_, nBlk = K_ / EPB_;
    static constexpr int lpr_log2 = c_plan(K_, EPB_, M0_, M1_, M2_, PAIRED_, NWG_).lpr_log2, LPR = 1 << lpr_log2, RPS = 64 >> lpr_log2;
    static constexpr int iters = c_plan(K_, EPB_, M0_, M1_, M2_, PAIRED_, NWG_).iters, njobs = c_plan(K_, EPB_, M0_, M1_, M2_, PAIRED_, NWG_).njobs;
    static constexpr int M0 = M0_, M1 = M1_, M2 = M2_;
    static constexpr int S1 = c_plan(K_, EPB_, M0_, M1_, M2_, PAIRED_, NWG_).slot0[1], S2 = c_plan(K_, EPB_, M0_, M1_, M2_, PAIRED_, NWG_).slot0[2];
    static constexpr int total = c_plan(K_, EPB_, M0_, M1_, M2_, PAIRED_, NWG_).total, spg = c_plan(K_, EPB_, M0_, M1_, M2_, PAIRED_, NWG_).spg;
    static constexpr int R = spg * RPS; /* rows of one workgroup (contiguous in the phase's output vector) */
    static constexpr bool PAIRED = PAIRED_;
    static constexpr CPlan plan() { return c_plan(K_, EPB_, M0_, M1_, M2_, PAIRED_, NWG_); }
};
// the four phases of a model shape
// elements per block as the engine walks a matrix: the storage's 16-byte block, except 1-bit and 2-bit -- one DWORD of a 128-element block, resp. one 8-byte half of a
// 64-element block (32 weights either way) per lane, so that such a matrix has the lanes, slots and chains of a 4-bit one (BlockPrep<FMT_Q1T>, <FMT_Q2T>)
template <int FMT>
constexpr int eng_vepb() { return (FMT == FMT_Q1T || FMT == FMT_Q2T) ? 32 : BlockDot<FMT>::EPB; }
template <int FMT, int DIM, int QD, int KVD, int FFN, int NWG, int NWG1 = NWG> /* NWG1: the workgroups that own q | k | v rows (kf_xengine.hip: GQA-4 shapes) */
struct EngShape {
    static constexpr int EPB = eng_vepb<FMT>();
    using P1 = PlanT<DIM, EPB, QD, KVD, KVD, false, NWG1>;
    using P4 = PlanT<QD, EPB, DIM, 0, 0, false, NWG>;
    using P5 = PlanT<DIM, EPB, FFN, FFN, 0, true, NWG>;
    using P6 = PlanT<FFN, EPB, DIM, 0, 0, false, NWG>;
};

}  // namespace kf
