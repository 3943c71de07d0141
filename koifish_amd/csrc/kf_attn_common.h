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
__device__ __forceinline__ void prep_head(const HeadRaw r, bool norm, const float* __restrict__ tab_pos, int hd, float eps, uint16_t* dst, float* rstd_out = nullptr) {
    const int lane = threadIdx.x & 63, half = hd >> 1;
    const int j = lane; /* hd <= 128: one trip covers the head */
    const bool act = j < half;
    float c = 1.f, sn = 0.f;
    if (tab_pos && act) c = tab_pos[2 * j], sn = tab_pos[2 * j + 1];
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
    if (tab_pos) {
        const float a = x0 * c, b = x1 * sn, cc = x0 * sn, d = x1 * c;
        x0 = round_bf16(a - b);
        x1 = round_bf16(cc + d);
    }
    if (act) dst[j] = f2bf(x0), dst[j + half] = f2bf(x1);
}

// e^x for x <= 0 through the hardware exp2 (v_exp_f32); exp2(-inf) = 0
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f); }

constexpr int ATTN_U = 4; /* key tiles kept in flight per wave */

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace kf
