// kf_device.h -- device-side helpers shared by the gfx950 kernels (wave64, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kf {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN stays NaN: hipcc lowers the cast to v_cvt_pk_bf16_f32
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float round_bf16(float f) { return bf2f(f2bf(f)); }
// two floats -> packed bf16 pair (lo = a, hi = b), one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
// acc + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}

// IEEE half bit pattern -> f32 (v_cvt_f32_f16)
__device__ __forceinline__ float half_bits_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }

// streamed-once data (weights, KV): non-temporal 16-byte load
__device__ __forceinline__ u32x4 ld_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

// Fixed fp32 exp: the same operation sequence as oracle/kfo_math.h kfo_expf (independent statement of one
// recipe: Cody-Waite reduction, degree-7 Taylor/Horner in fma form, two-step power-of-two scaling), so the
// softmax numerators and SwiGLU agree bit for bit with the CPU oracle.  Needs -ffp-contract=off.
__device__ __forceinline__ float kf_expf(float x) {
    if (x > 88.72283f) return __builtin_inff();
    if (x < -87.33654f) return 0.0f;
    if (x != x) return x;
    float t = x * 1.44269502162933349609375f;
    float n = (t + 12582912.0f) - 12582912.0f;
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187045e-06f, r);
    float p = 1.98412701138295233249664306640625e-4f;
    p = fmaf(p, r, 1.38888892251998186111450195312500e-3f);
    p = fmaf(p, r, 8.33333376795053482055664062500000e-3f);
    p = fmaf(p, r, 4.16666679084300994873046875000000e-2f);
    p = fmaf(p, r, 1.66666671633720397949218750000000e-1f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    // the oracle scales in two exact-then-rounded steps ((p * 2^e1) * 2^e2, e1 = e/2): the first product is exact, so the pair rounds once,
    // exactly like one v_ldexp_f32
    return __builtin_amdgcn_ldexpf(p, (int)n);
}

// Fixed fp32 natural log (same operation sequence as oracle/kfo_math.h kfo_logf): x = m * 2^e with m in [sqrt(1/2), sqrt(2)),
// s = (m-1)/(m+1), log m = 2s + 2s*z*P(z), z = s^2, P = 1/3 + z/5 + z^2/7 + z^3/9 + z^4/11 (Horner, fma form), e*ln2 split hi/lo.
// <= 2 ulp; log(0) = -inf, log(x < 0) = NaN, subnormals scaled by 2^23 first.  Used for the cross-entropy loss of kf_fused_classifier.
__device__ __forceinline__ float kf_logf(float x) {
    if (x != x || x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int e = 0;
    uint32_t u = __float_as_uint(x);
    if (u < 0x00800000u) x *= 8388608.0f, e = -23, u = __float_as_uint(x);
    e += (int)(u >> 23) - 127;
    u = (u & 0x007fffffu) | 0x3f800000u;
    float m = __uint_as_float(u);
    if (m > 1.41421353816986083984375f) m *= 0.5f, e += 1;
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float p = 9.0909093618392944336e-2f;
    p = fmaf(p, z, 1.1111111193895339966e-1f);
    p = fmaf(p, z, 1.4285714924335479736e-1f);
    p = fmaf(p, z, 2.0000000298023223877e-1f);
    p = fmaf(p, z, 3.3333334326744079590e-1f);
    const float s2 = s + s;
    const float fe = (float)e;
    const float lo = fmaf(s2 * z, p, fe * 9.058001351536227e-6f);
    return fmaf(fe, 6.9313812255859375e-1f, s2 + lo);
}

// 2^y as f * 2^n: n = the integer nearest to y (ties to even), f = 2^(y - n) from a fixed degree-7 polynomial (the recipe of oracle/kfo_math.h kfo_exp2_parts,
// stated independently).  The canonical softmax of the decode attention keeps the parts apart: a power of two rescales exactly.
__device__ __forceinline__ void kf_exp2_parts(float y, float& f, float& n) {
    const float nn = (y + 12582912.0f) - 12582912.0f;
    const float r = y - nn;
    float p = 1.52527338e-5f;
    p = fmaf(p, r, 1.54035304e-4f);
    p = fmaf(p, r, 1.33335581e-3f);
    p = fmaf(p, r, 9.61812911e-3f);
    p = fmaf(p, r, 5.55041087e-2f);
    p = fmaf(p, r, 2.40226507e-1f);
    p = fmaf(p, r, 6.93147181e-1f);
    p = fmaf(p, r, 1.0f);
    f = p, n = nn;
}

// DPP cross-lane moves (row = 16 lanes): quad_perm xor1 = 0xB1, xor2 = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sums over lanes l ^ 32 and l ^ 16 without LDS (gfx950 row swaps: with both operands equal the two results are the two halves /
// row pairs broadcast, so their sum is the butterfly sum in every lane)
__device__ __forceinline__ float xsum32(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xsum16(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ double xsum32_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]) + __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ double xsum16_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]) + __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
// fp64 sum over the whole wave: 4 DPP steps inside each 16-lane row, 2 cross-row swaps
__device__ __forceinline__ double wave_sum_f64_fast(double v) {
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    return xsum32_d(xsum16_d(v));
}

__device__ __forceinline__ double wave_sum_f64(double v) { return wave_sum_f64_fast(v); }
// fp32 sum / max over the whole wave, VALU only (no LDS round trips): DPP inside the rows, row swaps across them
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return xsum32(xsum16(v));
}
__device__ __forceinline__ float xmax32(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xmax16(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return xmax32(xmax16(v));
}

// Sum of squares of a bf16 vector in fp64 over a whole workgroup (nthreads a multiple of 64, <= 1024).
// fp64 keeps the result independent of the reduction tree (products of bf16 values are exact, the fp64 sum of
// <= 2^16 of them is exact or correctly rounded to far below an fp32 ulp), which is what lets the oracle and the
// kernels agree bit for bit on RMSNorm.  `red` is >= 16 doubles of LDS.
__device__ __forceinline__ double block_sumsq_bf16(const uint16_t* __restrict__ x, int n, double* red) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double a = (double)bf2f(x[i]);
        acc = fma(a, a, acc);
    }
    acc = wave_sum_f64(acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < nw; w++) tot += red[w];
    __syncthreads();
    return tot;
}

// ---- 16-entry bf16 table in registers, looked up with v_perm_b32 (4-bit weights; see kf_gemv_lut.hip / kf_gemv.hip)
struct PermLut {
    uint32_t tl[4], th[4];
};
__device__ __forceinline__ void build_perm_lut(PermLut& t, float step, float zero, float nb) {
    uint32_t P[8];
#pragma unroll
    for (int q2 = 0; q2 < 8; q2++) {
        const uint32_t r = pack_bf16x2(fmaf((float)(2 * q2), step, nb), fmaf((float)(2 * q2 + 1), step, nb));
        P[q2] = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        t.tl[k] = __builtin_amdgcn_perm(P[2 * k + 1], P[2 * k], 0x06040200u);
        t.th[k] = __builtin_amdgcn_perm(P[2 * k + 1], P[2 * k], 0x07050301u);
    }
}
// s: four indices 0..15, one per byte (the high nibble of every byte may hold anything) -> the four entries as a low-byte word and a
// high-byte word (same byte positions).  Two 8-entry v_perm_b32 per plane on index bits 0..2, then a third v_perm_b32 picks, byte by byte, the
// lower or the upper half by index bit 3 (selector byte k = k + 4 * bit3_k, formed by one shift and one v_and_or_b32): 9 instructions per
// 4 weights.  (The first version built a byte mask from bit 3 and merged with v_bfi_b32; the compiler turned the mask's shift-and-subtract into
// a quarter-rate v_mul_lo_u32 by 255: 14 issue slots per 4 weights.)
__device__ __forceinline__ void perm_lookup4(uint32_t s, const PermLut& t, uint32_t& lo, uint32_t& hi) {
    const uint32_t idx = s & 0x07070707u;
    uint32_t sel; /* ((s >> 1) & 0x04040404) | 0x03020100 as ONE v_and_or_b32: VOP3 takes no literals on gfx9, so the compiler would issue and + or */
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(s >> 1), "s"(0x04040404u), "v"(0x03020100u));
    const uint32_t la = __builtin_amdgcn_perm(t.tl[1], t.tl[0], idx), lb = __builtin_amdgcn_perm(t.tl[3], t.tl[2], idx);
    const uint32_t ha = __builtin_amdgcn_perm(t.th[1], t.th[0], idx), hb = __builtin_amdgcn_perm(t.th[3], t.th[2], idx);
    lo = __builtin_amdgcn_perm(lb, la, sel);
    hi = __builtin_amdgcn_perm(hb, ha, sel);
}
// X here is staged as {(x0,x2), (x4,x6), (x1,x3), (x5,x7)}
__device__ __forceinline__ float perm_dot_dword(uint32_t D, u32x4 X, const PermLut& t, float acc) {
    uint32_t lo, hi;
    perm_lookup4(D >> 4, t, lo, hi); /* bytes 3..0 = elements 0, 2, 4, 6 */
    acc = dot2_bf16(__builtin_amdgcn_perm(hi, lo, 0x06020703u), X.x, acc);
    acc = dot2_bf16(__builtin_amdgcn_perm(hi, lo, 0x04000501u), X.y, acc);
    perm_lookup4(D, t, lo, hi); /* elements 1, 3, 5, 7 */
    acc = dot2_bf16(__builtin_amdgcn_perm(hi, lo, 0x06020703u), X.z, acc);
    acc = dot2_bf16(__builtin_amdgcn_perm(hi, lo, 0x04000501u), X.w, acc);
    return acc;
}
// The same lookup with the weights paired as the arithmetic form pairs them -- (e0,e1), (e2,e3), (e4,e5), (e6,e7) against X.x .. X.w in natural
// order -- so that the fp32 sum is formed in exactly the order of dot_q4_dword (kf_gemv.hip): two extra v_perm_b32 per dword gather the index bytes
// (low nibble of a byte of D >> 4 = an even element, of D = an odd element; the lookup ignores the high nibbles).
// (the pair product of kf_gemv_blocks.h, stated here for the lookup below)
// ---- the accumulator of one lane's mat-vec chain.  Default order: one float fed by v_dot2c_f32_bf16.  Canonical order (oracle/kf_oracle.c section 4c, round 4):
// TWO fused multiply-add chains per lane -- the even-indexed elements of the lane's blocks into .x, the odd-indexed into .y, both through one v_pk_fma_f32 per
// weight pair (each half is an IEEE fma: fmaf on the host) -- joined as x + y in front of the lane tree.
template <bool CANON>
struct AccOf {
    typedef float T;
};
template <>
struct AccOf<true> {
    typedef f32x2_t T;
};
template <bool CANON>
using acc_t = typename AccOf<CANON>::T;
__device__ __forceinline__ float acc_join(float a) { return a; }
__device__ __forceinline__ float acc_join(f32x2_t a) { return a.x + a.y; }
__device__ __forceinline__ float acc_pick(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ f32x2_t acc_pick(bool c, f32x2_t a, f32x2_t b) { return f32x2_t{c ? a.x : b.x, c ? a.y : b.y}; }
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
// 2-bit storage, canonical order: a chain pair per 32-element half of the lane's 64-element blocks
struct Acc2 {
    f32x2_t s[2];
};
__device__ __forceinline__ float acc_join(const Acc2& a) { return (a.s[0].x + a.s[0].y) + (a.s[1].x + a.s[1].y); }
__device__ __forceinline__ Acc2 acc_pick(bool c, const Acc2& a, const Acc2& b) { return Acc2{{acc_pick(c, a.s[0], b.s[0]), acc_pick(c, a.s[1], b.s[1])}}; }
// 1-bit storage, canonical order: one chain pair per dword position of the lane's 128-element blocks (oracle/kf_oracle.c section 4c); joined as a balanced tree
struct Acc4 {
    f32x2_t s[4];
};
__device__ __forceinline__ float acc_join(const Acc4& a) { return ((a.s[0].x + a.s[0].y) + (a.s[1].x + a.s[1].y)) + ((a.s[2].x + a.s[2].y) + (a.s[3].x + a.s[3].y)); }
__device__ __forceinline__ Acc4 acc_pick(bool c, const Acc4& a, const Acc4& b) {
    Acc4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.s[i] = acc_pick(c, a.s[i], b.s[i]);
    return r;
}
template <bool CANON>
__device__ __forceinline__ acc_t<CANON> dotp_dev(uint32_t w, uint32_t x, acc_t<CANON> acc) {
    if constexpr (CANON) {
        return pk_fma(f32x2_t{bf_lo(w), bf_hi(w)}, f32x2_t{bf_lo(x), bf_hi(x)}, acc);
    } else {
        return dot2_bf16(w, x, acc);
    }
}
template <bool CANON = false>
__device__ __forceinline__ acc_t<CANON> perm_dot_dword_nat(uint32_t D, u32x4 X, const PermLut& t, acc_t<CANON> acc) {
    const uint32_t even = D >> 4; /* bytes 3..0: elements 0,2,4,6 in the low nibbles; D itself: 1,3,5,7 */
    uint32_t lo, hi;
    perm_lookup4(__builtin_amdgcn_perm(even, D, 0x02060307u), t, lo, hi); /* bytes 0..3 = elements 0, 1, 2, 3 */
    acc = dotp_dev<CANON>(__builtin_amdgcn_perm(hi, lo, 0x05010400u), X.x, acc);
    acc = dotp_dev<CANON>(__builtin_amdgcn_perm(hi, lo, 0x07030602u), X.y, acc);
    perm_lookup4(__builtin_amdgcn_perm(even, D, 0x00040105u), t, lo, hi); /* elements 4, 5, 6, 7 */
    acc = dotp_dev<CANON>(__builtin_amdgcn_perm(hi, lo, 0x05010400u), X.z, acc);
    acc = dotp_dev<CANON>(__builtin_amdgcn_perm(hi, lo, 0x07030602u), X.w, acc);
    return acc;
}
__device__ __forceinline__ u32x4 perm_x_order(u32x4 o) {
    return u32x4{__builtin_amdgcn_perm(o.y, o.x, 0x05040100u), __builtin_amdgcn_perm(o.w, o.z, 0x05040100u), __builtin_amdgcn_perm(o.y, o.x, 0x07060302u),
                 __builtin_amdgcn_perm(o.w, o.z, 0x07060302u)};
}

// quad broadcast (DPP quad_perm k,k,k,k): every lane of an aligned group of 4 gets lane k's value
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true);
}

}  // namespace kf
