// kf_gemm_common.h -- pieces shared by the token-batch GEMM kernels (kf_gemm.hip, kf_gemm2.hip)
#pragma once
#include "kf_kernels.h"

namespace kf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM_TOK = 128;         /* tokens per workgroup tile (4 MFMA column blocks) */
constexpr int GM_KT = 128;          /* k per staged x tile */
constexpr int GM_XS = GM_KT + 8;    /* padded LDS row in bf16 elements: 272 B, ds_read_b128 of 32 rows is conflict-free */

struct GemmArgs {
    const unsigned char* w;
    const uint16_t* zero;
    const uint16_t* step;
    float qBias;
    int M, K, nBlk, gshift;
    const uint16_t* x;
    long long ldx;
    int n;
    uint16_t* y;
    long long ldy;
    const uint16_t* bias;
    const uint16_t* residual;
    long long ldr;
    float alpha, beta;
    // direct kernel only: further weights sharing x and K -- jobs 1, 2 of one launch (Q/K/V), or the `up` matrix of a paired SwiGLU launch
    int njobs;      /* 1..3 */
    int rb_end[3];  /* cumulative count of 32-row blocks per job */
    const unsigned char* xw[2];
    const uint16_t* xzero[2];
    const uint16_t* xstep[2];
    float xqBias[2];
    int xM[2];
    uint16_t* xy[2];
    long long xldy[2];
    // kf_gemm3.hip only, the 128 x 128 tile, head_dim 128 (a tile = one head): ROPE::cuInfer (q/k-norm CU_rms_forward_v2 + rotate-half RoPE) in the epilogue of the stacked
    // Q | K | V launch -- job 0's rows normed with qk_norm[0], job 1's with qk_norm[1], job 2 (V) stored as it is; position of token t = rope_pos0 + t
    int qkrope;
    const uint16_t* qk_norm[2];
    const float* rope_table; /* [n_ctx][64][2] (cos, sin) */
    int rope_pos0;
    int rope_seq; /* > 0: the rows are sequences of rope_seq tokens back to back (a batch of prompts, a training batch): position of token t = rope_pos0 + t % rope_seq */
    float qk_eps;
    int swiglu; /* kf_gemm3.hip only: W = gate | up interleaved in blocks of 16 rows (dequant_launch ilv_n = 2), M = 2 ffn; the epilogue stores SwiGLU(gate, up) of every FFN row: y[n, ffn] */
};

// ---- 8 consecutive weights -> 4 packed bf16 pairs (element 2i in the low half of word i)
__device__ __forceinline__ u32x4 frag_q4(uint32_t D, float step, float step16, float nb, float zero) {
    uint32_t H = D & 0xF0F0F0F0u, L = D & 0x0F0F0F0Fu;
    asm("" : "+v"(H)); /* opaque masks keep the byte extractions as v_cvt_f32_ubyteN (see kf_gemv.hip) */
    asm("" : "+v"(L));
    u32x4 o;
    uint32_t r;
    r = pack_bf16x2(fmaf((float)(H >> 24), step16, nb), fmaf((float)(L >> 24), step, nb));
    o.x = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    r = pack_bf16x2(fmaf((float)((H >> 16) & 0xffu), step16, nb), fmaf((float)((L >> 16) & 0xffu), step, nb));
    o.y = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    r = pack_bf16x2(fmaf((float)((H >> 8) & 0xffu), step16, nb), fmaf((float)((L >> 8) & 0xffu), step, nb));
    o.z = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    r = pack_bf16x2(fmaf((float)(H & 0xffu), step16, nb), fmaf((float)(L & 0xffu), step, nb));
    o.w = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    return o;
}
// row-codebook 4-bit stream (KF_QUANT_ROW_LUT): D = 4 stream bytes (element 2b in the high nibble of byte b), ta / tb = the row's 16 bf16 entries
__device__ __forceinline__ u32x4 frag_q4r(uint32_t D, u32x4 ta, u32x4 tb) {
    const uint32_t P[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
    PermLut t;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        t.tl[k] = __builtin_amdgcn_perm(P[2 * k + 1], P[2 * k], 0x06040200u);
        t.th[k] = __builtin_amdgcn_perm(P[2 * k + 1], P[2 * k], 0x07050301u);
    }
    const uint32_t even = D >> 4, odd = D; /* low nibble of byte b: element 2b / 2b + 1 (perm_lookup4 ignores the high nibbles) */
    u32x4 o;
    uint32_t lo, hi;
    perm_lookup4(__builtin_amdgcn_perm(even, odd, 0x01050004u), t, lo, hi); /* bytes 0..3 = elements 0, 1, 2, 3 */
    o.x = __builtin_amdgcn_perm(hi, lo, 0x05010400u), o.y = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
    perm_lookup4(__builtin_amdgcn_perm(even, odd, 0x03070206u), t, lo, hi); /* elements 4, 5, 6, 7 */
    o.z = __builtin_amdgcn_perm(hi, lo, 0x05010400u), o.w = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
    return o;
}
// 16 bits, element 0 in bits 15..14
__device__ __forceinline__ u32x4 frag_q2(uint32_t v, float step, float nb, float zero) {
    uint32_t o[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const float q0 = (float)((v >> (14 - 4 * p)) & 3u), q1 = (float)((v >> (12 - 4 * p)) & 3u);
        const uint32_t r = pack_bf16x2(fmaf(q0, step, nb), fmaf(q1, step, nb));
        o[p] = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}
// 8 bits, element 0 in bit 7; w0 / w1 = dequant(0) / dequant(1) as bf16 bit patterns
__device__ __forceinline__ u32x4 frag_q1(uint32_t b, uint32_t w0, uint32_t w1) {
    uint32_t o[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const uint32_t lo = ((b >> (7 - 2 * p)) & 1u) ? w1 : w0, hi = ((b >> (6 - 2 * p)) & 1u) ? w1 : w0;
        o[p] = lo | (hi << 16);
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}
// 8 f8e5m2 bytes (element i = byte i): value = half(byte << 8), exact in bf16
__device__ __forceinline__ u32x4 frag_f8(uint32_t D0, uint32_t D1) {
    // v_cvt_pk_f32_bf8 (gfx950, OCP E5M2): two bytes -> two fp32, the same values as half(byte << 8)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_bf8((int)D0, false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)D0, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_bf8((int)D1, false), d = __builtin_amdgcn_cvt_pk_f32_bf8((int)D1, true);
    return u32x4{pack_bf16x2(a.x, a.y), pack_bf16x2(b.x, b.y), pack_bf16x2(c.x, c.y), pack_bf16x2(d.x, d.y)};
}

// ---- epilogue: lane holds token (tb*32 + r), rows row_base + 8g + 4h + j
template <int TB>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[TB], const GemmArgs& a, int tok0, int row_base, int r, int h) {
    const bool vec_ok = ((a.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 7) == 0);
#pragma unroll
    for (int tb = 0; tb < TB; tb++) {
        const int tok = tok0 + tb * 32 + r;
        if (tok >= a.n) continue;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int rg = row_base + 8 * g + 4 * h;
            if (rg >= a.M) continue;
            uint16_t* yp = a.y + (size_t)tok * a.ldy + rg;
            uint16_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[tb][4 * g + j];
                if (rg + j < a.M) {
                    if (a.alpha != 1.0f) v = a.alpha * v;
                    if (a.beta != 0.0f) v = v + a.beta * bf2f(yp[j]);
                    if (a.bias) v = v + bf2f(a.bias[rg + j]);
                    uint16_t q = f2bf(v);
                    if (a.residual) q = f2bf(bf2f(a.residual[(size_t)tok * a.ldr + rg + j]) + bf2f(q));
                    o[j] = q;
                } else {
                    o[j] = 0;
                }
            }
            if (vec_ok && rg + 3 < a.M) {
                *reinterpret_cast<u32x2*>(yp) = u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (rg + j < a.M) yp[j] = o[j];
            }
        }
    }
}

// kf_gemm2.hip: large-batch tile kernel; KF_OK launched, 1 = not for this kernel
int gemm2_launch(hipStream_t st, int fmt, const GemmArgs& a);
// kf_gemm3.hip: 256 x 256 x 64 bf16 tiles staged by global_load_lds; KF_OK launched, 1 = not for this kernel
int gemm3_launch(hipStream_t st, int fmt, const GemmArgs& a, void* ws = nullptr, size_t ws_bytes = 0); /* ws: lends the 128 x 128 form its split-K slots (gemm3_sk_ws_bytes) */

}  // namespace kf
