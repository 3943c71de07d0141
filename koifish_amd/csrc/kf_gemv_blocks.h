// kf_gemv_blocks.h -- the per-block dot products (PackedQ unpack + bf16-stepwise dequant + v_dot2c_f32_bf16 chain) and the lane-group
// reduction shared by the mat-vec kernel (kf_gemv.hip) and the persistent decode engine (kf_engine.hip): ONE statement of the arithmetic,
// so both paths form every output bit the same way.
#pragma once
#include <type_traits>

#include "kf_kernels.h"

namespace kf {

// One pair of products added to an accumulator (acc_t, kf_device.h).  CANON = false: v_dot2c_f32_bf16 (one instruction; its internal rounding has no bit-exact
// CPU model).  CANON = true: the canonical order kernels and oracle share (oracle/kf_oracle.c section 4c): the pair's low (even-indexed) element goes into the
// lane's even chain, its high element into the odd chain -- one v_pk_fma_f32, each half an IEEE fma, every bit reproducible with fmaf on the host.  All block
// dots below add their pairs in element order, so the two chains of a lane are "the even / the odd elements of a block in index order, block after block".
template <bool CANON>
__device__ __forceinline__ acc_t<CANON> dotp(uint32_t w, uint32_t x, acc_t<CANON> acc) {
    return dotp_dev<CANON>(w, x, acc);
}

// ------------------------------------------------------------------------------------------------ block dots
// Q4: Packed128 memory image (PackedQ.hpp:99-183): dword3 (bytes 12..15) holds elements 0..7 with element 0 in
// bits 28..31; dword2 -> 8..15; dword1 -> 16..23; dword0 -> 24..31.
// dequant (T.cu:274, all bf16 operators):  w = bf16( bf16(step * (q - qBias)) - zero ).
//   step*(q-qBias) is formed exactly in fp32 by one fma (q <= 15, step has 8 significant bits), rounded to
//   bf16 by v_cvt_pk_bf16_f32 (two at a time), widened, zero subtracted in fp32 (exact operands), rounded again.
template <bool CANON>
__device__ __forceinline__ acc_t<CANON> dot_q4_dword(uint32_t D, u32x4 X, float step, float step16, float nb, float zero, acc_t<CANON> acc) {
    uint32_t H = D & 0xF0F0F0F0u, L = D & 0x0F0F0F0Fu;
    // keep the two masks opaque: folded into the byte extraction below they would defeat v_cvt_f32_ubyte1..3 (one op per nibble)
    asm("" : "+v"(H));
    asm("" : "+v"(L));
    uint32_t r, w;
    r = pack_bf16x2(fmaf((float)(H >> 24), step16, nb), fmaf((float)(L >> 24), step, nb));
    w = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    acc = dotp<CANON>(w, X.x, acc);
    r = pack_bf16x2(fmaf((float)((H >> 16) & 0xffu), step16, nb), fmaf((float)((L >> 16) & 0xffu), step, nb));
    w = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    acc = dotp<CANON>(w, X.y, acc);
    r = pack_bf16x2(fmaf((float)((H >> 8) & 0xffu), step16, nb), fmaf((float)((L >> 8) & 0xffu), step, nb));
    w = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    acc = dotp<CANON>(w, X.z, acc);
    r = pack_bf16x2(fmaf((float)(H & 0xffu), step16, nb), fmaf((float)(L & 0xffu), step, nb));
    w = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    acc = dotp<CANON>(w, X.w, acc);
    return acc;
}

template <int FMT, bool CANON = false>
struct BlockDot;
// BlockDot<FMT, CANON>::Acc: the lane's accumulator -- acc_t<CANON> (a float, or the canonical even / odd pair), except 1-bit storage in the canonical order: Acc4, a pair per
// dword position of the block

template <bool CANON>
struct BlockDot<FMT_BF16, CANON> {
    static constexpr int EPB = 8, XCH = 1;
    static constexpr bool HAS_GAMA = false;
    using Acc = acc_t<CANON>;
    __device__ static __forceinline__ acc_t<CANON> run(u32x4 w, const u32x4* xs, int col, int nBlk, float, float, float, acc_t<CANON> acc) {
        u32x4 X = xs[col];
        acc = dotp<CANON>(w.x, X.x, acc);
        acc = dotp<CANON>(w.y, X.y, acc);
        acc = dotp<CANON>(w.z, X.z, acc);
        acc = dotp<CANON>(w.w, X.w, acc);
        return acc;
    }
};

// F8E5M2: byte i of the block = element i; value = half(byte << 8) (g_float.hpp:355-383), exact in bf16.
__device__ __forceinline__ float f8_to_f32(uint32_t hbits) { return half_bits_to_f32(hbits); }
template <bool CANON>
__device__ __forceinline__ acc_t<CANON> dot_f8_dword(uint32_t D, uint32_t X0, uint32_t X1, acc_t<CANON> acc) {
    // gfx950 converts two OCP E5M2 bytes to fp32 in one instruction (v_cvt_pk_f32_bf8): the same values as half(byte << 8), exactly
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)D, false), hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)D, true);
    uint32_t w0 = pack_bf16x2(lo.x, lo.y);
    uint32_t w1 = pack_bf16x2(hi.x, hi.y);
    acc = dotp<CANON>(w0, X0, acc);
    acc = dotp<CANON>(w1, X1, acc);
    return acc;
}
template <bool CANON>
struct BlockDot<FMT_F8, CANON> {
    static constexpr int EPB = 16, XCH = 2;
    static constexpr bool HAS_GAMA = false;
    using Acc = acc_t<CANON>;
    __device__ static __forceinline__ acc_t<CANON> run(u32x4 w, const u32x4* xs, int col, int nBlk, float, float, float, acc_t<CANON> acc) {
        u32x4 X0 = xs[col], X1 = xs[nBlk + col];
        acc = dot_f8_dword<CANON>(w.x, X0.x, X0.y, acc);
        acc = dot_f8_dword<CANON>(w.y, X0.z, X0.w, acc);
        acc = dot_f8_dword<CANON>(w.z, X1.x, X1.y, acc);
        acc = dot_f8_dword<CANON>(w.w, X1.z, X1.w, acc);
        return acc;
    }
};

template <bool CANON>
struct BlockDot<FMT_Q4, CANON> {
    static constexpr int EPB = 32, XCH = 4;
    static constexpr bool HAS_GAMA = true;
    using Acc = acc_t<CANON>;
    // nb = -qBias*step (exact)
    __device__ static __forceinline__ acc_t<CANON> run(u32x4 w, const u32x4* xs, int col, int nBlk, float step, float zero, float nb, acc_t<CANON> acc) {
        const float step16 = step * 0.0625f;
        acc = dot_q4_dword<CANON>(w.w, xs[col], step, step16, nb, zero, acc);
        acc = dot_q4_dword<CANON>(w.z, xs[nBlk + col], step, step16, nb, zero, acc);
        acc = dot_q4_dword<CANON>(w.y, xs[2 * nBlk + col], step, step16, nb, zero, acc);
        acc = dot_q4_dword<CANON>(w.x, xs[3 * nBlk + col], step, step16, nb, zero, acc);
        return acc;
    }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2_t perm_fma_dword(uint32_t D, f32x4 X0, f32x4 X1, const PermLut& t, f32x2_t acc); /* below: the lookup with fp32 results */

// 4-bit through a per-group table (register byte planes + v_perm_b32, kf_device.h): the 4 lanes that hold the 4 blocks of a 128-weight group
// build the 16 entries together -- lane i of the quad forms entries 4i..4i+3 with the arithmetic above, packs them into one low-byte and one
// high-byte plane word and the quad exchanges the 8 plane words by DPP broadcasts -- then every weight costs a lookup instead of the
// fma / round / subtract / round chain: ~4.7 VALU instructions per weight instead of 6.25.  The launcher picks this form only when a group
// is exactly one aligned lane quad (lGroup 128, K a multiple of 128, at least 4 lanes per row).  The lookup pairs the weights as the arithmetic
// form does (perm_dot_dword_nat), so the fp32 sums -- and every output bit -- are those of BlockDot<FMT_Q4>.
template <bool CANON>
struct BlockDot<FMT_Q4P, CANON> {
    static constexpr int EPB = 32, XCH = 4;
    static constexpr bool HAS_GAMA = true;
    using Acc = acc_t<CANON>;
    __device__ static __forceinline__ acc_t<CANON> run(u32x4 w, const u32x4* xs, int col, int nBlk, float step, float zero, float nb, acc_t<CANON> acc) {
        const float q0 = (float)((threadIdx.x & 3) << 2);
        uint32_t r = pack_bf16x2(fmaf(q0, step, nb), fmaf(q0 + 1.0f, step, nb));
        const uint32_t P0 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        r = pack_bf16x2(fmaf(q0 + 2.0f, step, nb), fmaf(q0 + 3.0f, step, nb));
        const uint32_t P1 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        const uint32_t tlm = __builtin_amdgcn_perm(P1, P0, 0x06040200u), thm = __builtin_amdgcn_perm(P1, P0, 0x07050301u);
        PermLut t;
        t.tl[0] = quad_bcast<0>(tlm), t.tl[1] = quad_bcast<1>(tlm), t.tl[2] = quad_bcast<2>(tlm), t.tl[3] = quad_bcast<3>(tlm);
        t.th[0] = quad_bcast<0>(thm), t.th[1] = quad_bcast<1>(thm), t.th[2] = quad_bcast<2>(thm), t.th[3] = quad_bcast<3>(thm);
#ifndef KF_Q4P_CANON_OLD
#define KF_Q4P_CANON_OLD 0 /* scratch/build_variant.py A/B: 1 = the paired-word form (perm_dot_dword_nat) */
#endif
        if constexpr (CANON && !KF_Q4P_CANON_OLD) { /* canonical order on bf16 activations (vectors too long to stage as fp32: the 25600-wide down_proj of Qwen3-32B): the weights come out of the lookup
                                  as fp32 operands of the two chains (perm_fma_dword: no index gather, no pair word to widen) -- only x is widened; same products, same chains */
            const u32x4 X[4] = {xs[col], xs[nBlk + col], xs[2 * nBlk + col], xs[3 * nBlk + col]};
            const uint32_t D[4] = {w.w, w.z, w.y, w.x};
#pragma unroll
            for (int i = 0; i < 4; i++)
                acc = perm_fma_dword(D[i], f32x4{bf_lo(X[i].x), bf_hi(X[i].x), bf_lo(X[i].y), bf_hi(X[i].y)}, f32x4{bf_lo(X[i].z), bf_hi(X[i].z), bf_lo(X[i].w), bf_hi(X[i].w)}, t, acc);
            return acc;
        } else {
            acc = perm_dot_dword_nat<CANON>(w.w, xs[col], t, acc);
            acc = perm_dot_dword_nat<CANON>(w.z, xs[nBlk + col], t, acc);
            acc = perm_dot_dword_nat<CANON>(w.y, xs[2 * nBlk + col], t, acc);
            acc = perm_dot_dword_nat<CANON>(w.x, xs[3 * nBlk + col], t, acc);
            return acc;
        }
    }
};

// 4-bit row codebook (KF_QUANT_ROW_LUT; CU_Q42X_NF4 / CU_Q42X_lut, quantizer.cu:583-652): the row's nibbles as BIT_SET_k streams them
// (element 2b in the high nibble of byte b), a weight = lut[row][nibble].  The 16 bf16 entries sit in two 16-byte words per row; they
// become the byte planes of the register lookup above, and a byte swap turns a stream dword into the Packed128 nibble order it expects.
template <bool CANON>
struct BlockDot<FMT_Q4R, CANON> {
    static constexpr int EPB = 32, XCH = 4;
    static constexpr bool HAS_GAMA = false;
    using Acc = acc_t<CANON>;
    __device__ static __forceinline__ acc_t<CANON> run(u32x4, const u32x4*, int, int, float, float, float, acc_t<CANON> acc) { return acc; }
    __device__ static __forceinline__ acc_t<CANON> run_lut(u32x4 w, const u32x4* xs, int col, int nBlk, u32x4 ta, u32x4 tb, acc_t<CANON> acc) {
        const uint32_t P[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
        PermLut t;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            t.tl[k] = __builtin_amdgcn_perm(P[2 * k + 1], P[2 * k], 0x06040200u);
            t.th[k] = __builtin_amdgcn_perm(P[2 * k + 1], P[2 * k], 0x07050301u);
        }
        acc = perm_dot_dword_nat<CANON>(__builtin_amdgcn_perm(0u, w.x, 0x00010203u), xs[col], t, acc);
        acc = perm_dot_dword_nat<CANON>(__builtin_amdgcn_perm(0u, w.y, 0x00010203u), xs[nBlk + col], t, acc);
        acc = perm_dot_dword_nat<CANON>(__builtin_amdgcn_perm(0u, w.z, 0x00010203u), xs[2 * nBlk + col], t, acc);
        acc = perm_dot_dword_nat<CANON>(__builtin_amdgcn_perm(0u, w.w, 0x00010203u), xs[3 * nBlk + col], t, acc);
        return acc;
    }
};

// 2-bit (T_SIGN ternary / generic CU_Q128toX_<T,64>): element i < 32 at high >> (62-2i) (PackedQ.hpp:185-226):
// dword3 -> elements 0..15 (element 0 in bits 30..31), dword2 -> 16..31, dword1 -> 32..47, dword0 -> 48..63.
template <bool CANON>
__device__ __forceinline__ acc_t<CANON> dot_q2_dword(uint32_t D, u32x4 Xa, u32x4 Xb, float step, float nb, float zero, acc_t<CANON> acc) {
    const uint32_t xw[8] = {Xa.x, Xa.y, Xa.z, Xa.w, Xb.x, Xb.y, Xb.z, Xb.w};
#pragma unroll
    for (int p = 0; p < 8; p++) {
        float q0 = (float)((D >> (30 - 4 * p)) & 3u), q1 = (float)((D >> (28 - 4 * p)) & 3u);
        uint32_t r = pack_bf16x2(fmaf(q0, step, nb), fmaf(q1, step, nb));
        uint32_t w = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        acc = dotp<CANON>(w, xw[p], acc);
    }
    return acc;
}
template <bool CANON>
struct BlockDot<FMT_Q2, CANON> {
    static constexpr int EPB = 64, XCH = 8;
    static constexpr bool HAS_GAMA = true;
    using Acc = std::conditional_t<CANON, Acc2, float>;
    __device__ static __forceinline__ Acc run(u32x4 w, const u32x4* xs, int col, int nBlk, float step, float zero, float nb, Acc acc) {
        if constexpr (CANON) { /* a chain pair per 32-element half */
            acc.s[0] = dot_q2_dword<true>(w.w, xs[col], xs[nBlk + col], step, nb, zero, acc.s[0]);
            acc.s[0] = dot_q2_dword<true>(w.z, xs[2 * nBlk + col], xs[3 * nBlk + col], step, nb, zero, acc.s[0]);
            acc.s[1] = dot_q2_dword<true>(w.y, xs[4 * nBlk + col], xs[5 * nBlk + col], step, nb, zero, acc.s[1]);
            acc.s[1] = dot_q2_dword<true>(w.x, xs[6 * nBlk + col], xs[7 * nBlk + col], step, nb, zero, acc.s[1]);
        } else {
            acc = dot_q2_dword<false>(w.w, xs[col], xs[nBlk + col], step, nb, zero, acc);
            acc = dot_q2_dword<false>(w.z, xs[2 * nBlk + col], xs[3 * nBlk + col], step, nb, zero, acc);
            acc = dot_q2_dword<false>(w.y, xs[4 * nBlk + col], xs[5 * nBlk + col], step, nb, zero, acc);
            acc = dot_q2_dword<false>(w.x, xs[6 * nBlk + col], xs[7 * nBlk + col], step, nb, zero, acc);
        }
        return acc;
    }
};

// 1-bit (BOOL1 / T_BINARY, CU_Q128toX_<T,128>): element i < 64 at high >> (63-i) (PackedQ.hpp:200-239):
// dword3 -> elements 0..31 (element 0 = bit 31), dword2 -> 32..63, dword1 -> 64..95, dword0 -> 96..127.
// w in {w0, w1} = {dequant(0), dequant(1)}: both are formed once per block, then selected per bit.
template <bool CANON>
__device__ __forceinline__ acc_t<CANON> dot_q1_dword(uint32_t D, const u32x4* xs, int base, int nBlk, int col, uint32_t w0, uint32_t w1, acc_t<CANON> acc) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        u32x4 X = xs[(base + c) * nBlk + col];
        const uint32_t xw[4] = {X.x, X.y, X.z, X.w};
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int k = c * 8 + p * 2; /* elements k, k+1 of this dword */
            uint32_t lo = ((D >> (31 - k)) & 1u) ? w1 : w0, hi = ((D >> (30 - k)) & 1u) ? w1 : w0;
            acc = dotp<CANON>((lo & 0xffffu) | (hi << 16), xw[p], acc);
        }
    }
    return acc;
}
template <bool CANON>
struct BlockDot<FMT_Q1, CANON> {
    static constexpr int EPB = 128, XCH = 16;
    static constexpr bool HAS_GAMA = true;
    using Acc = std::conditional_t<CANON, Acc4, float>;
    __device__ static __forceinline__ Acc run(u32x4 w, const u32x4* xs, int col, int nBlk, float step, float zero, float nb, Acc acc) {
        // dequant(q) for q = 0, 1 (q - qBias folded into nb)
        uint32_t r = pack_bf16x2(fmaf(0.0f, step, nb), fmaf(1.0f, step, nb));
        uint32_t ww = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        const uint32_t w0 = ww & 0xffffu, w1 = ww >> 16;
        if constexpr (CANON) { /* a chain pair per dword position */
            acc.s[0] = dot_q1_dword<true>(w.w, xs, 0, nBlk, col, w0, w1, acc.s[0]);
            acc.s[1] = dot_q1_dword<true>(w.z, xs, 4, nBlk, col, w0, w1, acc.s[1]);
            acc.s[2] = dot_q1_dword<true>(w.y, xs, 8, nBlk, col, w0, w1, acc.s[2]);
            acc.s[3] = dot_q1_dword<true>(w.x, xs, 12, nBlk, col, w0, w1, acc.s[3]);
        } else {
            acc = dot_q1_dword<false>(w.w, xs, 0, nBlk, col, w0, w1, acc);
            acc = dot_q1_dword<false>(w.z, xs, 4, nBlk, col, w0, w1, acc);
            acc = dot_q1_dword<false>(w.y, xs, 8, nBlk, col, w0, w1, acc);
            acc = dot_q1_dword<false>(w.x, xs, 12, nBlk, col, w0, w1, acc);
        }
        return acc;
    }
};

// 1-bit through a table: a weight byte (8 weights) indexes 16 v_perm_b32 selector bytes in LDS (256 entries x 16 B, filled by the workgroup
// before the x prologue's barrier); each selector dword picks {w0, w1} for a weight pair out of the register (w1 << 16 | w0).  The same weights,
// pairs and summation order as BlockDot<FMT_Q1> -- every output bit is equal -- at 1 ds_read_b128 + 4 v_perm + 4 dot2 per 8 weights instead of
// ~32 VALU instructions.
template <bool CANON>
struct BlockDot<FMT_Q1T, CANON> {
    static constexpr int EPB = 128, XCH = 16;
    static constexpr bool HAS_GAMA = true;
    using Acc = std::conditional_t<CANON, Acc4, float>;
    __device__ static __forceinline__ Acc run(u32x4 w, const u32x4* xs, int col, int nBlk, float step, float zero, float nb, Acc acc) {
        return run_tab(w, xs, col, nBlk, step, zero, nb, xs + nBlk * 16 + 16 /* the mat-vec kernel keeps the table behind x (K * 2 bytes) and the 256-byte reduction scratch */, acc);
    }
    // tab: the 256-entry selector table
    __device__ static __forceinline__ Acc run_tab(u32x4 w, const u32x4* xs, int col, int nBlk, float step, float zero, float nb, const u32x4* tab, Acc acc) {
        const uint32_t r = pack_bf16x2(fmaf(0.0f, step, nb), fmaf(1.0f, step, nb));
        const uint32_t ww = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero); /* bytes 0,1 = dequant(0); bytes 2,3 = dequant(1) */
        const uint32_t dw[4] = {w.w, w.z, w.y, w.x};                       /* dword3 holds elements 0..31, element 0 = bit 31 */
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const u32x4 S = tab[(dw[d] >> (24 - 8 * c)) & 0xffu];
                const u32x4 X = xs[(4 * d + c) * nBlk + col];
                if constexpr (CANON) { /* dword d's chain pair */
                    acc.s[d] = dotp<true>(__builtin_amdgcn_perm(0u, ww, S.x), X.x, acc.s[d]);
                    acc.s[d] = dotp<true>(__builtin_amdgcn_perm(0u, ww, S.y), X.y, acc.s[d]);
                    acc.s[d] = dotp<true>(__builtin_amdgcn_perm(0u, ww, S.z), X.z, acc.s[d]);
                    acc.s[d] = dotp<true>(__builtin_amdgcn_perm(0u, ww, S.w), X.w, acc.s[d]);
                } else {
                    acc = dotp<false>(__builtin_amdgcn_perm(0u, ww, S.x), X.x, acc);
                    acc = dotp<false>(__builtin_amdgcn_perm(0u, ww, S.y), X.y, acc);
                    acc = dotp<false>(__builtin_amdgcn_perm(0u, ww, S.z), X.z, acc);
                    acc = dotp<false>(__builtin_amdgcn_perm(0u, ww, S.w), X.w, acc);
                }
            }
        return acc;
    }
};

// 2-bit the same way: a weight byte (4 weights) indexes 8 selector bytes (256 entries x 8 B); the four dequantised levels sit in two registers
// {dequant(1) << 16 | dequant(0)}, {dequant(3) << 16 | dequant(2)} that v_perm_b32 reads as one 8-byte source.  1 ds_read_b64 + 2 v_perm + 2 dot2
// per 4 weights instead of ~22 VALU instructions; weights, pairs and summation order are those of BlockDot<FMT_Q2>.
template <bool CANON>
struct BlockDot<FMT_Q2T, CANON> {
    static constexpr int EPB = 64, XCH = 8;
    static constexpr bool HAS_GAMA = true;
    using Acc = std::conditional_t<CANON, Acc2, float>;
    __device__ static __forceinline__ Acc run(u32x4 w, const u32x4* xs, int col, int nBlk, float step, float zero, float nb, Acc acc) {
        uint32_t r = pack_bf16x2(fmaf(0.0f, step, nb), fmaf(1.0f, step, nb));
        const uint32_t T01 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        r = pack_bf16x2(fmaf(2.0f, step, nb), fmaf(3.0f, step, nb));
        const uint32_t T23 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        const u32x2* tab = reinterpret_cast<const u32x2*>(xs + nBlk * 8 + 16); /* behind x (K * 2 bytes) and the 256-byte reduction scratch */
        const uint32_t dw[4] = {w.w, w.z, w.y, w.x};                           /* dword3 holds elements 0..15, element 0 = bits 31..30 */
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const u32x4 Xa = xs[(2 * d) * nBlk + col], Xb = xs[(2 * d + 1) * nBlk + col];
            const uint32_t xw[8] = {Xa.x, Xa.y, Xa.z, Xa.w, Xb.x, Xb.y, Xb.z, Xb.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const u32x2 S = tab[(dw[d] >> (24 - 8 * c)) & 0xffu];
                if constexpr (CANON) { /* dwords 0, 1 -> the first half's chain pair; 2, 3 -> the second's */
                    acc.s[d >> 1] = dotp<true>(__builtin_amdgcn_perm(T23, T01, S.x), xw[2 * c], acc.s[d >> 1]);
                    acc.s[d >> 1] = dotp<true>(__builtin_amdgcn_perm(T23, T01, S.y), xw[2 * c + 1], acc.s[d >> 1]);
                } else {
                    acc = dotp<false>(__builtin_amdgcn_perm(T23, T01, S.x), xw[2 * c], acc);
                    acc = dotp<false>(__builtin_amdgcn_perm(T23, T01, S.y), xw[2 * c + 1], acc);
                }
            }
        }
        return acc;
    }
};


// ------------------------------------------------------------------------------------------------ canonical block dots on fp32 activations
// The persistent engine's form of the canonical order (same products, same chain: element after element of a block, one v_fma_f32 each): the activation
// vector is staged in LDS as fp32 -- chunks of 4 floats laid out [EPB / 4][nBlk] -- and a weight is formed as an fp32 register directly (the table lookup
// assembles {0, 0, low byte, high byte} with one v_perm_b32), so neither operand is unpacked from a bf16 pair per product.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2_t perm_fma_dword(uint32_t D, f32x4 X0, f32x4 X1, const PermLut& t, f32x2_t acc) {
    // the even-indexed elements (low nibbles of D >> 4, element 0 in byte 3) and the odd-indexed ones (low nibbles of D) are looked up as they lie: a pair's two fp32
    // operands come from the two lookups at the same byte position, so the index bytes need no gather (round 3 gathered elements 0..3 / 4..7 first: 2 perms more per dword)
    uint32_t le, he, lo, ho;
    perm_lookup4(D >> 4, t, le, he);
    perm_lookup4(D, t, lo, ho);
    acc = pk_fma(f32x2_t{__uint_as_float(__builtin_amdgcn_perm(he, le, 0x07030c0cu)), __uint_as_float(__builtin_amdgcn_perm(ho, lo, 0x07030c0cu))}, f32x2_t{X0.x, X0.y}, acc); /* elements 0, 1 */
    acc = pk_fma(f32x2_t{__uint_as_float(__builtin_amdgcn_perm(he, le, 0x06020c0cu)), __uint_as_float(__builtin_amdgcn_perm(ho, lo, 0x06020c0cu))}, f32x2_t{X0.z, X0.w}, acc); /* 2, 3 */
    acc = pk_fma(f32x2_t{__uint_as_float(__builtin_amdgcn_perm(he, le, 0x05010c0cu)), __uint_as_float(__builtin_amdgcn_perm(ho, lo, 0x05010c0cu))}, f32x2_t{X1.x, X1.y}, acc); /* 4, 5 */
    acc = pk_fma(f32x2_t{__uint_as_float(__builtin_amdgcn_perm(he, le, 0x04000c0cu)), __uint_as_float(__builtin_amdgcn_perm(ho, lo, 0x04000c0cu))}, f32x2_t{X1.z, X1.w}, acc); /* 6, 7 */
    return acc;
}
// arithmetic form: w = bf16(bf16(step * (q - qBias)) - zero) per nibble, as dot_q4_dword forms it, kept as fp32
__device__ __forceinline__ f32x2_t arith_fma_dword(uint32_t D, f32x4 X0, f32x4 X1, float step, float step16, float nb, float zero, f32x2_t acc) {
    uint32_t H = D & 0xF0F0F0F0u, Lw = D & 0x0F0F0F0Fu;
    asm("" : "+v"(H));
    asm("" : "+v"(Lw));
    const float xs8[8] = {X0.x, X0.y, X0.z, X0.w, X1.x, X1.y, X1.z, X1.w};
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int sh = 24 - 8 * p;
        const uint32_t r = pack_bf16x2(fmaf((float)((H >> sh) & 0xffu), step16, nb), fmaf((float)((Lw >> sh) & 0xffu), step, nb));
        const uint32_t w = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        acc = pk_fma(f32x2_t{bf_lo(w), bf_hi(w)}, f32x2_t{xs8[2 * p], xs8[2 * p + 1]}, acc);
    }
    return acc;
}
template <int FMT>
struct BlockDotF;
template <>
struct BlockDotF<FMT_Q4P> {
    static constexpr int EPB = 32, XCH = 8;
    static constexpr bool HAS_GAMA = true;
    __device__ static __forceinline__ f32x2_t run(u32x4 w, const f32x4* xs, int col, int nBlk, float step, float zero, float nb, f32x2_t acc) {
        const float q0 = (float)((threadIdx.x & 3) << 2);
        uint32_t r = pack_bf16x2(fmaf(q0, step, nb), fmaf(q0 + 1.0f, step, nb));
        const uint32_t P0 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        r = pack_bf16x2(fmaf(q0 + 2.0f, step, nb), fmaf(q0 + 3.0f, step, nb));
        const uint32_t P1 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        const uint32_t tlm = __builtin_amdgcn_perm(P1, P0, 0x06040200u), thm = __builtin_amdgcn_perm(P1, P0, 0x07050301u);
        PermLut t;
        t.tl[0] = quad_bcast<0>(tlm), t.tl[1] = quad_bcast<1>(tlm), t.tl[2] = quad_bcast<2>(tlm), t.tl[3] = quad_bcast<3>(tlm);
        t.th[0] = quad_bcast<0>(thm), t.th[1] = quad_bcast<1>(thm), t.th[2] = quad_bcast<2>(thm), t.th[3] = quad_bcast<3>(thm);
        acc = perm_fma_dword(w.w, xs[col], xs[nBlk + col], t, acc);
        acc = perm_fma_dword(w.z, xs[2 * nBlk + col], xs[3 * nBlk + col], t, acc);
        acc = perm_fma_dword(w.y, xs[4 * nBlk + col], xs[5 * nBlk + col], t, acc);
        acc = perm_fma_dword(w.x, xs[6 * nBlk + col], xs[7 * nBlk + col], t, acc);
        return acc;
    }
};
template <>
struct BlockDotF<FMT_Q4> {
    static constexpr int EPB = 32, XCH = 8;
    static constexpr bool HAS_GAMA = true;
    __device__ static __forceinline__ f32x2_t run(u32x4 w, const f32x4* xs, int col, int nBlk, float step, float zero, float nb, f32x2_t acc) {
        const float step16 = step * 0.0625f;
        acc = arith_fma_dword(w.w, xs[col], xs[nBlk + col], step, step16, nb, zero, acc);
        acc = arith_fma_dword(w.z, xs[2 * nBlk + col], xs[3 * nBlk + col], step, step16, nb, zero, acc);
        acc = arith_fma_dword(w.y, xs[4 * nBlk + col], xs[5 * nBlk + col], step, step16, nb, zero, acc);
        acc = arith_fma_dword(w.x, xs[6 * nBlk + col], xs[7 * nBlk + col], step, step16, nb, zero, acc);
        return acc;
    }
};

// ------------------------------------------------------------------------------------------------ dequantise ahead, multiply later
// The persistent engine holds a phase's packed blocks in registers before the phase's activations exist.  Everything that does not depend on x -- the group
// table, the lookups, the pairing -- can therefore run while the wave would otherwise wait for the hand-off: BlockPrep turns a block into its 16 bf16 pair
// words in element order (word 4 d + j = elements 8 d + 2 j, 8 d + 2 j + 1; the very words BlockDot / BlockDotF feed to their products), pairs_dot multiplies
// them with the staged activations in the order of BlockDot<FMT, CANON> (CANON = false, bf16 chunks) or BlockDotF<FMT> (CANON = true, fp32 chunks): every
// output bit is unchanged, the vector work behind the hand-off shrinks from ~6 to 0.5 (2 canonical) instructions per weight.
template <int FMT>
struct BlockPrep;
template <>
struct BlockPrep<FMT_Q4P> {
    __device__ static __forceinline__ void prep(u32x4 w, float step, float zero, float nb, int lane, uint32_t (&o)[16], const u32x4* = nullptr) {
        const float q0 = (float)((lane & 3) << 2);
        uint32_t r = pack_bf16x2(fmaf(q0, step, nb), fmaf(q0 + 1.0f, step, nb));
        const uint32_t P0 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        r = pack_bf16x2(fmaf(q0 + 2.0f, step, nb), fmaf(q0 + 3.0f, step, nb));
        const uint32_t P1 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        const uint32_t tlm = __builtin_amdgcn_perm(P1, P0, 0x06040200u), thm = __builtin_amdgcn_perm(P1, P0, 0x07050301u);
        PermLut t;
        t.tl[0] = quad_bcast<0>(tlm), t.tl[1] = quad_bcast<1>(tlm), t.tl[2] = quad_bcast<2>(tlm), t.tl[3] = quad_bcast<3>(tlm);
        t.th[0] = quad_bcast<0>(thm), t.th[1] = quad_bcast<1>(thm), t.th[2] = quad_bcast<2>(thm), t.th[3] = quad_bcast<3>(thm);
        const uint32_t dw[4] = {w.w, w.z, w.y, w.x};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t D = dw[d], even = D >> 4;
            uint32_t lo, hi;
            perm_lookup4(__builtin_amdgcn_perm(even, D, 0x02060307u), t, lo, hi); /* bytes 0..3 = elements 0, 1, 2, 3 */
            o[4 * d] = __builtin_amdgcn_perm(hi, lo, 0x05010400u), o[4 * d + 1] = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
            perm_lookup4(__builtin_amdgcn_perm(even, D, 0x00040105u), t, lo, hi); /* elements 4, 5, 6, 7 */
            o[4 * d + 2] = __builtin_amdgcn_perm(hi, lo, 0x05010400u), o[4 * d + 3] = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
        }
    }
};
template <>
struct BlockPrep<FMT_Q4> {
    __device__ static __forceinline__ void prep(u32x4 w, float step, float zero, float nb, int, uint32_t (&o)[16], const u32x4* = nullptr) {
        const float step16 = step * 0.0625f;
        const uint32_t dw[4] = {w.w, w.z, w.y, w.x};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t H = dw[d] & 0xF0F0F0F0u, Lw = dw[d] & 0x0F0F0F0Fu;
            asm("" : "+v"(H));
            asm("" : "+v"(Lw));
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int sh = 24 - 8 * p;
                const uint32_t r = pack_bf16x2(fmaf((float)((H >> sh) & 0xffu), step16, nb), fmaf((float)((Lw >> sh) & 0xffu), step, nb));
                o[4 * d + p] = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
            }
        }
    }
};
// 1-bit through the selector table, for the persistent engine: ONE DWORD of a 128-element block (32 weights, element 0 = bit 31) per lane -- the engine deals the four dwords
// of a block to four neighbouring lanes (the canonical order keeps a chain pair per dword position) and walks a 1-bit matrix with the geometry of a 4-bit one.
// tab: the 256-entry table of BlockDot<FMT_Q1T> (a weight byte -> 4 selector dwords = 4 weight pairs)
template <>
struct BlockPrep<FMT_Q1T> {
    __device__ static __forceinline__ void prep(u32x4 w, float step, float zero, float nb, int, uint32_t (&o)[16], const u32x4* tab) {
        const uint32_t r = pack_bf16x2(fmaf(0.0f, step, nb), fmaf(1.0f, step, nb));
        const uint32_t ww = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero); /* bytes 0,1 = dequant(0); bytes 2,3 = dequant(1) */
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const u32x4 S = tab[(w.x >> (24 - 8 * c)) & 0xffu];
            o[4 * c] = __builtin_amdgcn_perm(0u, ww, S.x), o[4 * c + 1] = __builtin_amdgcn_perm(0u, ww, S.y);
            o[4 * c + 2] = __builtin_amdgcn_perm(0u, ww, S.z), o[4 * c + 3] = __builtin_amdgcn_perm(0u, ww, S.w);
        }
    }
};
// 2-bit the same way: one 32-element HALF of a 64-element block per lane (two dwords: w.y = elements 0 .. 15, element 0 in bits 31..30; w.x = elements 16 .. 31);
// tab: the 256-entry table of BlockDot<FMT_Q2T> (a weight byte -> 2 selector dwords = 2 weight pairs out of the {T01, T23} pool)
template <>
struct BlockPrep<FMT_Q2T> {
    __device__ static __forceinline__ void prep(u32x4 w, float step, float zero, float nb, int, uint32_t (&o)[16], const u32x4* tabv) {
        uint32_t r = pack_bf16x2(fmaf(0.0f, step, nb), fmaf(1.0f, step, nb));
        const uint32_t T01 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        r = pack_bf16x2(fmaf(2.0f, step, nb), fmaf(3.0f, step, nb));
        const uint32_t T23 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        const u32x2* tab = reinterpret_cast<const u32x2*>(tabv);
        const uint32_t dw[2] = {w.y, w.x};
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const u32x2 S = tab[(dw[d] >> (24 - 8 * c)) & 0xffu];
                o[8 * d + 2 * c] = __builtin_amdgcn_perm(T23, T01, S.x), o[8 * d + 2 * c + 1] = __builtin_amdgcn_perm(T23, T01, S.y);
            }
    }
};
template <bool CANON>
__device__ __forceinline__ acc_t<CANON> pairs_dot(const uint32_t (&p)[16], const u32x4* xs, int col, int nBlk, acc_t<CANON> acc) {
    if constexpr (CANON) {
        const f32x4* xf = reinterpret_cast<const f32x4*>(xs);
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const f32x4 X0 = xf[(2 * d) * nBlk + col], X1 = xf[(2 * d + 1) * nBlk + col];
            acc = pk_fma(f32x2_t{bf_lo(p[4 * d]), bf_hi(p[4 * d])}, f32x2_t{X0.x, X0.y}, acc);
            acc = pk_fma(f32x2_t{bf_lo(p[4 * d + 1]), bf_hi(p[4 * d + 1])}, f32x2_t{X0.z, X0.w}, acc);
            acc = pk_fma(f32x2_t{bf_lo(p[4 * d + 2]), bf_hi(p[4 * d + 2])}, f32x2_t{X1.x, X1.y}, acc);
            acc = pk_fma(f32x2_t{bf_lo(p[4 * d + 3]), bf_hi(p[4 * d + 3])}, f32x2_t{X1.z, X1.w}, acc);
        }
    } else {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const u32x4 X = xs[d * nBlk + col];
            acc = dot2_bf16(p[4 * d], X.x, acc);
            acc = dot2_bf16(p[4 * d + 1], X.y, acc);
            acc = dot2_bf16(p[4 * d + 2], X.z, acc);
            acc = dot2_bf16(p[4 * d + 3], X.w, acc);
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------ kernel
// sum over the 2^lg lanes of each aligned lane group (lg wave-uniform); every lane of the group gets the sum.
// DPP inside a 16-lane row (quad_perm xor1, xor2, row_half_mirror, row_mirror), two cross-row swaps above it.
__device__ __forceinline__ float group_sum(float v, int lg) {
    if (lg >= 1) v += dpp_f<0xB1>(v);
    if (lg >= 2) v += dpp_f<0x4E>(v);
    if (lg >= 3) v += dpp_f<0x141>(v);
    if (lg >= 4) v += dpp_f<0x140>(v);
    if (lg >= 5) v = xsum16(v);
    if (lg >= 6) v = xsum32(v);
    return v;
}

}  // namespace kf
