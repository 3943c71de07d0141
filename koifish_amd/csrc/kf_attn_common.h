// kf_attn_common.h -- pieces of the decode attention shared by kf_attn.hip and the persistent decode engine (kf_engine.hip):
// per-head q/k RMSNorm + rotate-half RoPE (ROPE::cuInfer, rope.cu:645-672), the hardware-exp2 softmax exponent, sc1 accessors.
#pragma once
#include "kf_kernels.h"

namespace kf {

// sum over the 2^lg (<= 16) lanes of each aligned lane group, DPP only
__device__ __forceinline__ float group_sum16(float v, int lg) {
    if (lg >= 1) v += dpp_f<0xB1>(v);
    if (lg >= 2) v += dpp_f<0x4E>(v);
    if (lg >= 3) v += dpp_f<0x141>(v);
    if (lg >= 4) v += dpp_f<0x140>(v);
    return v;
}
// One head as loaded (raw bf16 bits, so that the loads can be issued long before their use): lane j holds the pair (j, j + hd/2)
struct HeadRaw {
    uint16_t x0, x1, w0, w1;
};
// Unconditional loads (clamped lane, the head itself standing in for a missing norm weight): a conditional load into a preset
// register makes the compiler wait for it -- and for every K/V load issued before it -- at the join.
__device__ __forceinline__ HeadRaw load_head(const uint16_t* __restrict__ src, const uint16_t* __restrict__ wn, int hd) {
    const int half = hd >> 1;
    int j = threadIdx.x & 63;
    j = j < half ? j : half - 1;
    const uint16_t* w = wn ? wn : src;
    return HeadRaw{src[j], src[j + half], w[j], w[j + half]};
}
// Prepare one head: optional per-head RMSNorm (s rounded to bf16 first, then (a*s)*w, RN store) and rotate-half
// RoPE from the host-built (cos,sin) table.  One wave per head; lane j handles the pair (j, j + hd/2).
// Result: bf16 bits in dst[hd].
// (c, sn) = the table's pair for (position, j), already in registers; rope = false: no rotation
__device__ __forceinline__ void prep_head_cs(const HeadRaw r, bool norm, bool rope, float c, float sn, int hd, float eps, uint16_t* dst, float* rstd_out = nullptr, int lane_in = -1) {
    const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63), half = hd >> 1;
    const int j = lane; /* hd <= 128: one trip covers the head */
    const bool act = j < half;
    float x0 = act ? bf2f(r.x0) : 0.f, x1 = act ? bf2f(r.x1) : 0.f; /* lanes past hd/2 hold a clamped copy */
    if (norm) {
        const float w0 = bf2f(r.w0), w1 = bf2f(r.w1);
        const double ss = wave_sum_f64_fast(fma((double)x0, (double)x0, (double)x1 * (double)x1));
        const float s0 = 1.0f / sqrtf((float)ss / (float)hd + eps);
        const float s = round_bf16(s0);
        if (rstd_out && lane == 0) *rstd_out = s0; /* the un-rounded 1/rms, for the training path's backward */
        x0 = round_bf16(x0 * s * w0);
        x1 = round_bf16(x1 * s * w1);
    }
    if (rope) {
        const float a = x0 * c, b = x1 * sn, cc = x0 * sn, d = x1 * c;
        x0 = round_bf16(a - b);
        x1 = round_bf16(cc + d);
    }
    if (act) dst[j] = f2bf(x0), dst[j + half] = f2bf(x1);
}
__device__ __forceinline__ void prep_head(const HeadRaw r, bool norm, const float* __restrict__ tab_pos, int hd, float eps, uint16_t* dst, float* rstd_out = nullptr) {
    const int j = threadIdx.x & 63;
    float c = 1.f, sn = 0.f;
    if (tab_pos && j < (hd >> 1)) c = tab_pos[2 * j], sn = tab_pos[2 * j + 1];
    prep_head_cs(r, norm, tab_pos != nullptr, c, sn, hd, eps, dst, rstd_out);
}

// e^x for x <= 0 through the hardware exp2 (v_exp_f32); exp2(-inf) = 0
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f); }

constexpr int ATTN_U = 4; /* key tiles kept in flight per wave */

// ------------------------------------------------------------------------------------------------ canonical decode attention
// The order kernels and oracle share (oracle/kf_oracle.c section 6, mode CANON), free of launch geometry:
//   score   s = bf16(d * (1 / sqrtf(hd))), d = this key's q.k dot: per lane a chain of 8 fused multiply-adds over its 8 elements, joined over the key's
//           hd / 8 lanes by the balanced tree of group_sum16;
//   weight  p = f * 2^(n - m), (f, n) = kf_exp2_parts(s * log2 e); m = ANY integer >= every n it is compared with: a power of two rescales exactly, so every
//           lane / wave / slice works against a maximum of its own and the partial sums are rescaled when they meet;
//   sums    fp64 (every term p and p * v is exact in fp64: the sums do not depend on their order to far below an fp32 ulp), out = bf16((float)(O / L)).
constexpr float KF_LOG2E = 1.44269502162933349609375f;
__device__ __forceinline__ float canon_score(const float (&q)[8], u32x4 kw, int lpk_log2, float rden) {
    const uint32_t k4[4] = {kw.x, kw.y, kw.z, kw.w};
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        d = fmaf(q[2 * i], bf_lo(k4[i]), d);
        d = fmaf(q[2 * i + 1], bf_hi(k4[i]), d);
    }
    d = group_sum16(d, lpk_log2);
    return round_bf16(d * rden);
}
// exponent of an exact rescale: d = n - m <= 0 (integer-valued), -inf or NaN (both sides -inf): clamped so that the scaled value underflows to 0
__device__ __forceinline__ int canon_shift(float d) { return (int)fmaxf(d, -1022.0f); }
__device__ __forceinline__ double ldexp_d(double v, int e) { return __builtin_ldexp(v, e); }
// a + b of the two halves of a 32-lane / 16-lane swap, fp64 (the row-swap reduce-scatter of the key-group sums)
__device__ __forceinline__ double swap32_add_d(double a, double b) {
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]) + __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ double swap16_add_d(double a, double b) {
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]) + __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
// One wave's share of a slice: running fp64 sums of its keys against the wave's own reference exponent.
template <int GQ>
struct CanonAcc {
    double o[GQ][8], l[GQ];
    float m[GQ]; /* integer-valued, -inf before the first valid key */
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            m[hq] = -__builtin_inff(), l[hq] = 0.0;
#pragma unroll
            for (int i = 0; i < 8; i++) o[hq][i] = 0.0;
        }
    }
};
// row broadcast (gfx90a+: row_newbcast:K -- every lane of a row of 16 gets lane K's value) and row rotate (row_ror:K: lane i gets lane i + K (mod 16)... of its row)
template <int K>
__device__ __forceinline__ uint32_t row_bcast_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + K, 0xF, 0xF, true);
}
template <int K>
__device__ __forceinline__ float row_bcast_f(float v) { return __uint_as_float(row_bcast_u<K>(__float_as_uint(v))); }
template <int K>
__device__ __forceinline__ double row_bcast_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    return __builtin_bit_cast(double, ((unsigned long long)row_bcast_u<K>((uint32_t)(u >> 32)) << 32) | row_bcast_u<K>((uint32_t)u));
}
// the q.k dot of canon_score without the scale and the rounding
__device__ __forceinline__ float canon_dot(const float (&q)[8], u32x4 kw, int lpk_log2) {
    const uint32_t k4[4] = {kw.x, kw.y, kw.z, kw.w};
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        d = fmaf(q[2 * i], bf_lo(k4[i]), d);
        d = fmaf(q[2 * i + 1], bf_hi(k4[i]), d);
    }
    return group_sum16(d, lpk_log2);
}
template <int GQ, int U, int I = 1>
__device__ __forceinline__ void canon_pick_item(const float (&d)[U][GQ], const bool (&valid)[U], int lig, float& mine, bool& vmine) {
    if constexpr (I < U * GQ) {
        mine = lig == I ? d[I / GQ][I % GQ] : mine;
        vmine = lig == I ? valid[I / GQ] : vmine;
        canon_pick_item<GQ, U, I + 1>(d, valid, lig, mine, vmine);
    }
}
// max over the lanes lig, lig + GQ, lig + 2 GQ, ... (U of them) of a row: after the steps lane i holds the maximum over i, i + GQ, ..., i + (U - 1) GQ (mod 16)
template <int GQ, int U, int STEP = 1>
__device__ __forceinline__ float canon_items_max(float v) {
    if constexpr (STEP < U) {
        v = fmaxf(v, dpp_f<0x120 + ((16 - GQ * STEP) & 15)>(v)); /* row_ror:(16 - s): lane i reads lane i + s (mod 16) */
        return canon_items_max<GQ, U, STEP * 2>(v);
    } else {
        return v;
    }
}
template <int GQ, int HQ = 0>
__device__ __forceinline__ void canon_heads_max(float nm, float (&bn)[GQ]) {
    if constexpr (HQ < GQ) {
        bn[HQ] = row_bcast_f<HQ>(nm); /* lane hq of the row: the maximum over the head's items hq, hq + GQ, ... */
        canon_heads_max<GQ, HQ + 1>(nm, bn);
    }
}
template <int GQ, int U, int I = 0>
__device__ __forceinline__ void canon_apply_items(CanonAcc<GQ>& A, double p, const double (&vd)[U][8]) {
    if constexpr (I < U * GQ) {
        constexpr int u = I / GQ, hq = I % GQ;
        const double pi = row_bcast_d<I>(p);
        A.l[hq] += pi;
#pragma unroll
        for (int i = 0; i < 8; i++) A.o[hq][i] = fma(pi, vd[u][i], A.o[hq][i]);
        canon_apply_items<GQ, U, I + 1>(A, p, vd);
    }
}
// (Round 6, measured and removed: two query heads per v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 in the score chain and in the exponential's polynomial -- bit-identical (179
//  canonical tests), half the vector instructions of those two parts, and slower: the single-sequence engine 2304 -> 2103 tokens/s, 32 batched sequences 7770 -> 7410: the operand
//  pairs have to be assembled in even-aligned register pairs per use.)
#ifndef KF_CANON_ITEMS
#define KF_CANON_ITEMS 0 /* scratch/build_variant.py A/B.  1 = the item-per-lane form below: bit for bit the same (194 canonical tests), a quarter fewer vector instructions per key
                            tile on paper -- and SLOWER on the part: 32 sequences 7770 -> 6360 tokens/s, 16: 6240 -> 5710, 8: 4625 -> 4450, the single-sequence engine unchanged (each
                            broadcast is a DPP move with its wait states in front of a dependent fp64 chain; the replicated form keeps sixteen independent lanes busy instead).  Off. */
#endif
// one batch of U key tiles of this lane's key group: k / v tiles (8 elements of this lane), validity per tile; q as 8 floats per head
// KF_CANON_ITEMS (round 6, measured and left off): hd = 128, a key group is a row of 16 lanes -- the U * GQ dots of a group stand in all of its 16 lanes after the tree; the
// score's scale and rounding, the exponential's parts and the exact power-of-two weight p worked out ONCE, item i = u * GQ + hq in lane i of the group, p_i handed to the
// group's lanes by a row broadcast: the same operations on the same values as the per-lane form.
template <int GQ, int LPK, int U = ATTN_U>
__device__ __forceinline__ void canon_batch(CanonAcc<GQ>& A, const float (&qf)[GQ][8], const u32x4 (&kk)[U], const u32x4 (&vv)[U], const bool (&valid)[U], int lpk_log2, float rden) {
    if constexpr (KF_CANON_ITEMS && LPK == 16 && U * GQ <= 16 && (GQ & (GQ - 1)) == 0 && (U & (U - 1)) == 0) {
        constexpr int NI = U * GQ;
        const int lig = (int)(threadIdx.x & 15);
        float d[U][GQ];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int hq = 0; hq < GQ; hq++) d[u][hq] = canon_dot(qf[hq], kk[u], lpk_log2);
        float mine = d[0][0];
        bool vmine = valid[0];
        canon_pick_item<GQ, U>(d, valid, lig, mine, vmine);
        float f, n;
        kf_exp2_parts(round_bf16(mine * rden) * KF_LOG2E, f, n);
        vmine = vmine && lig < NI;
        n = vmine ? n : -__builtin_inff();
        float bn[GQ];
        canon_heads_max<GQ>(canon_items_max<GQ, U>(n), bn);
        float m_mine = A.m[0];
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) { /* the wave's maximum exponent of this batch */
            bn[hq] = xmax32(xmax16(bn[hq]));
            if (bn[hq] > A.m[hq]) { /* exact rescale of what has been summed so far (wave-uniform branch; the first batch has nothing to rescale) */
                if (A.m[hq] > -__builtin_inff()) {
                    const int e = canon_shift(A.m[hq] - bn[hq]);
                    A.l[hq] = ldexp_d(A.l[hq], e);
#pragma unroll
                    for (int i = 0; i < 8; i++) A.o[hq][i] = ldexp_d(A.o[hq][i], e);
                }
                A.m[hq] = bn[hq];
            }
            m_mine = (lig & (GQ - 1)) == hq ? A.m[hq] : m_mine;
        }
        const double p = vmine ? ldexp_d((double)f, canon_shift(n - m_mine)) : 0.0;
        double vd[U][8];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) vd[u][2 * i] = (double)bf_lo(vw[i]), vd[u][2 * i + 1] = (double)bf_hi(vw[i]);
        }
        canon_apply_items<GQ, U>(A, p, vd);
        return;
    }
    float f[U][GQ], n[U][GQ], bn[GQ];
#pragma unroll
    for (int hq = 0; hq < GQ; hq++) bn[hq] = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            const float s = canon_score(qf[hq], kk[u], lpk_log2, rden);
            kf_exp2_parts(s * KF_LOG2E, f[u][hq], n[u][hq]);
            n[u][hq] = valid[u] ? n[u][hq] : -__builtin_inff();
            bn[hq] = fmaxf(bn[hq], n[u][hq]);
        }
    }
#pragma unroll
    for (int hq = 0; hq < GQ; hq++) { /* the wave's maximum exponent of this batch */
        bn[hq] = xmax32(xmax16(bn[hq]));
        if (LPK < 16) bn[hq] = fmaxf(bn[hq], dpp_f<0x128>(bn[hq]));
        if (bn[hq] > A.m[hq]) { /* exact rescale of what has been summed so far (wave-uniform branch; the first batch has nothing to rescale) */
            if (A.m[hq] > -__builtin_inff()) {
                const int e = canon_shift(A.m[hq] - bn[hq]);
                A.l[hq] = ldexp_d(A.l[hq], e);
#pragma unroll
                for (int i = 0; i < 8; i++) A.o[hq][i] = ldexp_d(A.o[hq][i], e);
            }
            A.m[hq] = bn[hq];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
        double vd[8];
#pragma unroll
        for (int i = 0; i < 4; i++) vd[2 * i] = (double)bf_lo(vw[i]), vd[2 * i + 1] = (double)bf_hi(vw[i]);
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            const double p = valid[u] ? ldexp_d((double)f[u][hq], canon_shift(n[u][hq] - A.m[hq])) : 0.0;
            A.l[hq] += p;
#pragma unroll
            for (int i = 0; i < 8; i++) A.o[hq][i] = fma(p, vd[i], A.o[hq][i]);
        }
    }
}
// sums over the wave's key groups, left in LDS: c[d] for d < hd, c[hd] = l, c[hd + 1] = m (as a double); c = this wave's [hq] row of hd + 2 doubles
template <int GQ, int LPK>
__device__ __forceinline__ void canon_wave_to_lds(const CanonAcc<GQ>& A, double* comb_wave /* [GQ][hd + 2] */, int hd, int lane, int d0) {
    const int row = lane >> 4;
#pragma unroll
    for (int hq = 0; hq < GQ; hq++) {
        double s1[4], r2[2];
#pragma unroll
        for (int i = 0; i < 4; i++) s1[i] = swap32_add_d(A.o[hq][i], A.o[hq][i + 4]);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            r2[i] = swap16_add_d(s1[i], s1[i + 2]);
            if (LPK < 16) r2[i] += dpp_d<0x128>(r2[i]); /* two key groups per row: row_ror:8 */
        }
        double lt = xsum16_d(xsum32_d(A.l[hq]));
        if (LPK < 16) lt += dpp_d<0x128>(lt);
        double* c = comb_wave + (size_t)hq * (hd + 2);
        if (LPK == 16 || (lane & 8) == 0) c[d0 + 2 * row] = r2[0], c[d0 + 2 * row + 1] = r2[1];
        if (lane == 0) c[hd] = lt, c[hd + 1] = (double)A.m[hq];
    }
}

// ------------------------------------------------------------------------------------------------ the same slice bookkeeping in fp32 (the engine's default mode)
// Scores on packed bf16 pairs (v_dot2c_f32_bf16), weights p = v_exp_f32(s * log2 e - m) against an INTEGER reference exponent m (the ceiling of the wave's largest
// s * log2 e), fp32 sums.  Integer exponents keep every rescale an exact power of two, so a wave's (O, L, m) meets the other waves' -- and the slices' in the fp64
// merge, which takes them converted -- exactly as the canonical form's do; what differs from it is the approximate exponential and fp32 summation (the decode
// kernel's default arithmetic, held to the tolerance tests).
template <int GQ>
struct FastAcc {
    float o[GQ][8], l[GQ], m[GQ];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            m[hq] = -__builtin_inff(), l[hq] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) o[hq][i] = 0.f;
        }
    }
};
__device__ __forceinline__ int fast_shift(float d) { return (int)fmaxf(d, -200.0f); } /* d = m_a - m_b <= 0, integer-valued, -inf or NaN: clamped so that the scaled value underflows to 0 */
template <int GQ, int LPK, int U>
__device__ __forceinline__ void fast_batch(FastAcc<GQ>& A, const u32x4 (&q)[GQ], const u32x4 (&kk)[U], const u32x4 (&vv)[U], const bool (&valid)[U], int lpk_log2, float rden) {
    float t[U][GQ], bn[GQ];
#pragma unroll
    for (int hq = 0; hq < GQ; hq++) bn[hq] = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            float d = dot2_bf16(q[hq].x, kk[u].x, 0.f);
            d = dot2_bf16(q[hq].y, kk[u].y, d);
            d = dot2_bf16(q[hq].z, kk[u].z, d);
            d = dot2_bf16(q[hq].w, kk[u].w, d);
            d = group_sum16(d, lpk_log2);
            t[u][hq] = valid[u] ? round_bf16(d * rden) * KF_LOG2E : -__builtin_inff();
            bn[hq] = fmaxf(bn[hq], t[u][hq]);
        }
#pragma unroll
    for (int hq = 0; hq < GQ; hq++) {
        bn[hq] = xmax32(xmax16(bn[hq]));
        if (LPK < 16) bn[hq] = fmaxf(bn[hq], dpp_f<0x128>(bn[hq]));
        bn[hq] = ceilf(bn[hq]);
        if (bn[hq] > A.m[hq]) { /* wave-uniform */
            if (A.m[hq] > -__builtin_inff()) {
                const int e = fast_shift(A.m[hq] - bn[hq]);
                A.l[hq] = __builtin_ldexpf(A.l[hq], e);
#pragma unroll
                for (int i = 0; i < 8; i++) A.o[hq][i] = __builtin_ldexpf(A.o[hq][i], e);
            }
            A.m[hq] = bn[hq];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
        float vf[8];
#pragma unroll
        for (int i = 0; i < 4; i++) vf[2 * i] = bf_lo(vw[i]), vf[2 * i + 1] = bf_hi(vw[i]);
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            const float p = valid[u] ? __builtin_amdgcn_exp2f(t[u][hq] - A.m[hq]) : 0.f;
            A.l[hq] += p;
#pragma unroll
            for (int i = 0; i < 8; i++) A.o[hq][i] = fmaf(p, vf[i], A.o[hq][i]);
        }
    }
}
__device__ __forceinline__ float swap32_add_f(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add_f(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sums over the wave's key groups, left in LDS: c[d] for d < hd, c[hd] = l, c[hd + 1] = m; c = this wave's [hq] row of hd + 2 floats (canon_wave_to_lds in fp32)
template <int GQ, int LPK>
__device__ __forceinline__ void fast_wave_to_lds(const FastAcc<GQ>& A, float* comb_wave /* [GQ][hd + 2] */, int hd, int lane, int d0) {
    const int row = lane >> 4;
#pragma unroll
    for (int hq = 0; hq < GQ; hq++) {
        float s1[4], r2[2];
#pragma unroll
        for (int i = 0; i < 4; i++) s1[i] = swap32_add_f(A.o[hq][i], A.o[hq][i + 4]);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            r2[i] = swap16_add_f(s1[i], s1[i + 2]);
            if (LPK < 16) r2[i] += dpp_f<0x128>(r2[i]); /* two key groups per row: row_ror:8 */
        }
        float lt = xsum16(xsum32(A.l[hq]));
        if (LPK < 16) lt += dpp_f<0x128>(lt);
        float* c = comb_wave + (size_t)hq * (hd + 2);
        if (LPK == 16 || (lane & 8) == 0) *reinterpret_cast<float2*>(c + d0 + 2 * row) = float2{r2[0], r2[1]};
        if (lane == 0) c[hd] = lt, c[hd + 1] = A.m[hq];
    }
}

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace kf
