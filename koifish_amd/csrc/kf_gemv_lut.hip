// kf_gemv_lut.hip -- 4-bit PackedQ mat-vec for LARGE matrices: per-group dequant table in LDS, gfx950 / wave64.
//
// kf_gemv.hip's arithmetic block dot spends ~7.7 VALU instructions per weight on the reference's bf16-stepwise dequant
// bf16(bf16(step*q) - zero) (T.cu:274) and is VALU-bound at ~2.3 TB/s on 25600x5120 (profiles/: SQ counters).  A 4-bit group
// has only 16 distinct weights, so here ONE lane owns ONE 128-weight group: it forms the 16 bf16 table entries once (exactly the
// same arithmetic), writes them to a lane-private LDS column -- entry q of lane l at dword (q*64 + l), i.e. bank l mod 32
// whatever q is, so the 64 lanes never conflict -- and then turns each nibble into an LDS address (shift, and-or with the column
// base: the table of a wave is 4 KiB-aligned so the OR is an add), reads its entry, packs two of them and feeds v_dot2c_f32_bf16.
// Mapping: a row of K has K/128 groups; 8 lanes walk a row, 8 rows side by side form a workgroup slot, the NW waves of the
// workgroup split the K/1024 iterations of the slot among themselves and combine through LDS in a fixed order.
#include <stdlib.h>

#include "kf_kernels.h"

namespace kf {

typedef __attribute__((address_space(3))) uint32_t lds_u32;

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

struct LutStep {
    u32x4 w[4];
    float st, ze;
};

// 16 table entries of one group -> this lane's LDS column (dword slots, low 16 bits = bf16 weight); base = LDS byte address
__device__ __forceinline__ void build_lut(uint32_t base, float step, float zero, float nb) {
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
        const uint32_t r = pack_bf16x2(fmaf((float)q, step, nb), fmaf((float)(q + 1), step, nb));
        const uint32_t w = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        *((lds_u32*)(size_t)(base + q * 256)) = w & 0xffffu;
        *((lds_u32*)(size_t)(base + (q + 1) * 256)) = w >> 16;
    }
}

__device__ __forceinline__ uint32_t lut_at(uint32_t addr) { return *((const lds_u32*)(size_t)addr); }

// 8 weights of one dword against 8 activations (X: 4 packed bf16 pairs).  Element 0 is the top nibble (PackedQ.hpp:143-183).
// base has bits 8..11 clear (4 KiB-aligned table + lane*4), so (nibble << 8) | base is the entry's address.
__device__ __forceinline__ float lut_dot_dword(uint32_t D, u32x4 X, uint32_t base, float acc) {
    const uint32_t w0 = lut_at(((D >> 20) & 0xF00u) | base), w1 = lut_at(((D >> 16) & 0xF00u) | base);
    const uint32_t w2 = lut_at(((D >> 12) & 0xF00u) | base), w3 = lut_at(((D >> 8) & 0xF00u) | base);
    const uint32_t w4 = lut_at(((D >> 4) & 0xF00u) | base), w5 = lut_at((D & 0xF00u) | base);
    const uint32_t w6 = lut_at(((D << 4) & 0xF00u) | base), w7 = lut_at(((D << 8) & 0xF00u) | base);
    acc = dot2_bf16(w0 | (w1 << 16), X.x, acc);
    acc = dot2_bf16(w2 | (w3 << 16), X.y, acc);
    acc = dot2_bf16(w4 | (w5 << 16), X.z, acc);
    acc = dot2_bf16(w6 | (w7 << 16), X.w, acc);
    return acc;
}

// ---- the same table in REGISTERS, looked up with v_perm_b32 (KF_Q4_LUT=2): the 16 bf16 entries of a group as two byte planes
// (low bytes tl[0..3], high bytes th[0..3]; plane word k holds entries 4k..4k+3).  A v_perm_b32 picks 4 bytes out of 8, so an
// index word (one 4-bit index per byte) is looked up in entries 0..7 and in entries 8..15 with its low 3 bits, and bit 3 selects
// per byte between the two (v_bfi_b32); two more perms interleave the planes into bf16 pairs.  3.9 VALU instructions per weight
// instead of 5.75 + dot, no LDS traffic; the pairs come out as (e0,e2) (e4,e6) (e1,e3) (e5,e7), so x is staged in that order.
// LDS: tables NW x 4 KiB (first: 4 KiB alignment) | x as u32x4 chunks [16 chunks of a group][nGrp] (K*2 bytes) | reduce scratch
template <int NW, int MODE, bool PERM>
__global__ void __launch_bounds__(NW * 64) gemv_q4lut_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(4096))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr size_t TAB = PERM ? 0 : (size_t)NW * 4096; /* the register-table form keeps no tables in LDS */
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw + TAB);
    float* red = reinterpret_cast<float*>(smem_raw + TAB + (size_t)a.K * 2); /* [NW][8 rows] floats, also fp64 scratch */
    const uint32_t lds0 = (uint32_t)(size_t)((lds_u32*)smem_raw); /* LDS byte address of the dynamic segment */
    const uint32_t base = lds0 + wave * 4096 + lane * 4;

    const int nGrp = a.nBlk, iters = a.iters; /* nBlk carries the number of 128-weight groups per row here; LPR = 8 */
    const int sub = lane >> 3, ll = lane & 7;
    const long s_begin = (long)blockIdx.x * a.spw;
    long s_end = s_begin + a.spw;
    if (s_end > a.total_slots) s_end = a.total_slots;

    const u32x4* const jw = reinterpret_cast<const u32x4*>(a.job[0].w);
    const uint16_t* const jstep = a.job[0].step;
    const uint16_t* const jzero = a.job[0].zero;
    const int jM = a.job[0].M;
    const float jqb = (float)a.job[0].qBias;

    // this wave's steps: iterations wave, wave+NW, ... of every slot of the workgroup
    const int my_its = iters > wave ? (iters - wave + NW - 1) / NW : 0;
    const int nsteps = (s_end > s_begin) ? (int)(s_end - s_begin) * my_its : 0;
    auto load = [&](long s, int it, LutStep& b) {
        const int row = (int)s * 8 + sub, c = it * 8 + ll;
        b.w[0] = b.w[1] = b.w[2] = b.w[3] = u32x4{0, 0, 0, 0};
        b.st = b.ze = 0.f;
        if (row < jM) {
            const uint32_t g = (uint32_t)row * (uint32_t)nGrp + (uint32_t)c;
            const u32x4* p = jw + (size_t)g * 4;
            b.w[0] = ld_nt(p), b.w[1] = ld_nt(p + 1), b.w[2] = ld_nt(p + 2), b.w[3] = ld_nt(p + 3);
            b.st = bf2f(jstep[g]), b.ze = bf2f(jzero[g]);
        }
    };

    LutStep cur, nxt;
    if (nsteps > 0) load(s_begin, wave, cur);
    const int pos = a.d_pos ? *a.d_pos : a.pos;

    // ---- prologue: stage x (optionally RMS-normalised) into LDS: element e = c*128 + j*8 + i -> chunk (j*nGrp + c)
    {
        const int nch = a.K >> 3;
        float mul = 1.0f;
        if (a.norm_w) {
            const double ss = block_sumsq_bf16(a.x, a.K, reinterpret_cast<double*>(red));
            mul = 1.0f / sqrtf(fmaf((float)ss, a.inv_dim, a.eps));
        }
        for (int e8 = tid; e8 < nch; e8 += NW * 64) {
            const int c = e8 >> 4, j = e8 & 15;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(a.x + (size_t)e8 * 8);
            u32x4 o = raw;
            if (a.norm_w) {
                const u32x4 nw = *reinterpret_cast<const u32x4*>(a.norm_w + (size_t)e8 * 8);
                const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w}, ww[4] = {nw.x, nw.y, nw.z, nw.w};
                uint32_t ow[4];
#pragma unroll
                for (int k = 0; k < 4; k++) ow[k] = pack_bf16x2((bf_lo(rw[k]) * mul) * bf_lo(ww[k]), (bf_hi(rw[k]) * mul) * bf_hi(ww[k]));
                o = u32x4{ow[0], ow[1], ow[2], ow[3]};
            }
            xs[j * nGrp + c] = PERM ? perm_x_order(o) : o;
        }
        __syncthreads();
    }

    float best_v = -__builtin_inff();
    int best_i = 0x7fffffff;
    for (long s = s_begin; s < s_end; s++) {
        float acc = 0.f;
        for (int it = wave; it < iters; it += NW) {
            // prefetch the next step of this wave (next iteration of this slot, else first iteration of the next slot)
            long ns = s;
            int nit = it + NW;
            if (nit >= iters) ns = s + 1, nit = wave;
            if (ns < s_end) load(ns, nit, nxt);
            const int c = it * 8 + ll;
            if constexpr (PERM) {
                PermLut t;
                build_perm_lut(t, cur.st, cur.ze, -(jqb * cur.st));
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const u32x4 w = cur.w[b];
                    acc = perm_dot_dword(w.w, xs[(4 * b + 0) * nGrp + c], t, acc);
                    acc = perm_dot_dword(w.z, xs[(4 * b + 1) * nGrp + c], t, acc);
                    acc = perm_dot_dword(w.y, xs[(4 * b + 2) * nGrp + c], t, acc);
                    acc = perm_dot_dword(w.x, xs[(4 * b + 3) * nGrp + c], t, acc);
                }
            } else {
                build_lut(base, cur.st, cur.ze, -(jqb * cur.st));
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const u32x4 w = cur.w[b];
                    acc = lut_dot_dword(w.w, xs[(4 * b + 0) * nGrp + c], base, acc);
                    acc = lut_dot_dword(w.z, xs[(4 * b + 1) * nGrp + c], base, acc);
                    acc = lut_dot_dword(w.y, xs[(4 * b + 2) * nGrp + c], base, acc);
                    acc = lut_dot_dword(w.x, xs[(4 * b + 3) * nGrp + c], base, acc);
                }
            }
            cur = nxt;
        }
        // the 8 lanes of a row, then the NW waves (fixed order)
        acc += dppf<0xB1>(acc);
        acc += dppf<0x4E>(acc);
        acc += dppf<0x141>(acc);
        if (ll == 0) red[wave * 8 + sub] = acc;
        __syncthreads();
        if (tid < 8) {
            float v = red[tid];
#pragma unroll
            for (int w2 = 1; w2 < NW; w2++) v += red[w2 * 8 + tid];
            const int r = (int)s * 8 + tid;
            if (r < jM) {
                if (a.yf) {
                    a.yf[r] = v;
                } else {
                    uint16_t* y = a.job[0].y + (size_t)pos * a.job[0].y_pos_stride;
                    if (a.alpha != 1.0f) v = a.alpha * v;
                    if (a.beta != 0.0f) v = v + a.beta * bf2f(y[r]);
                    if (a.bias) v = v + bf2f(a.bias[r]);
                    uint16_t o = f2bf(v);
                    if (a.residual) o = f2bf(bf2f(a.residual[r]) + bf2f(o));
                    y[r] = o;
                    if (MODE == GEMV_ARGMAX) {
                        const float fv = bf2f(o);
                        if (fv > best_v || (fv == best_v && r < best_i)) best_v = fv, best_i = r;
                    }
                }
            }
        }
        __syncthreads();
    }

    if (MODE == GEMV_ARGMAX) { /* only lanes 0..7 of wave 0 hold candidates */
        if (wave == 0) {
#pragma unroll
            for (int m = 4; m > 0; m >>= 1) {
                float ov = __shfl_xor(best_v, m, 64);
                int oi = __shfl_xor(best_i, m, 64);
                if (ov > best_v || (ov == best_v && oi < best_i)) best_v = ov, best_i = oi;
            }
            if (tid == 0) a.amax_val[blockIdx.x] = best_v, a.amax_idx[blockIdx.x] = best_i;
        }
    }
}

// ---- register-table form without any cross-wave step (KF_Q4_LUT=3): every wave walks whole rows on its own -- 8 lanes per row, 8 rows
// per slot, all K/1024 iterations of the slot, a 3-step DPP sum at the end -- so the only barrier is the one after x is staged.
template <int MODE>
__global__ void __launch_bounds__(256) gemv_q4perm_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw + (size_t)a.K * 2);
    const int nGrp = a.nBlk, iters = a.iters;
    const int sub = lane >> 3, ll = lane & 7;
    const long gwave = (long)blockIdx.x * 4 + wave;
    const long s_begin = gwave * a.spw;
    long s_end = s_begin + a.spw;
    if (s_end > a.total_slots) s_end = a.total_slots;
    const u32x4* const jw = reinterpret_cast<const u32x4*>(a.job[0].w);
    const uint16_t* const jstep = a.job[0].step;
    const uint16_t* const jzero = a.job[0].zero;
    const int jM = a.job[0].M;
    const float jqb = (float)a.job[0].qBias;
    auto load = [&](long s, int it, LutStep& b) {
        int row = (int)s * 8 + sub;
        if (row >= jM) row = jM - 1; /* rows past the end recompute the last row (no branch around the loads); never stored */
        const uint32_t g = (uint32_t)row * (uint32_t)nGrp + (uint32_t)(it * 8 + ll);
        const u32x4* p = jw + (size_t)g * 4;
        b.w[0] = ld_nt(p), b.w[1] = ld_nt(p + 1), b.w[2] = ld_nt(p + 2), b.w[3] = ld_nt(p + 3);
        b.st = bf2f(jstep[g]), b.ze = bf2f(jzero[g]);
    };
    // two groups in flight behind the one being multiplied (one was not enough: the waves sat on s_waitcnt)
    LutStep cur, nxt, nx2;
    const long nstep = (s_end > s_begin ? (s_end - s_begin) : 0) * iters;
    auto load_step = [&](long k, LutStep& b) { /* step k of this wave = (slot s_begin + k / iters, iteration k % iters); clamped past the end */
        if (k >= nstep) k = nstep - 1;
        load(s_begin + k / iters, (int)(k % iters), b);
    };
    if (nstep > 0) {
        load_step(0, cur);
        load_step(1, nxt);
    }
    const int pos = a.d_pos ? *a.d_pos : a.pos;
    {
        const int nch = a.K >> 3;
        float mul = 1.0f;
        if (a.norm_w) {
            const double ss = block_sumsq_bf16(a.x, a.K, red);
            mul = 1.0f / sqrtf(fmaf((float)ss, a.inv_dim, a.eps));
        }
        for (int e8 = tid; e8 < nch; e8 += 256) {
            const int c = e8 >> 4, j = e8 & 15;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(a.x + (size_t)e8 * 8);
            u32x4 o = raw;
            if (a.norm_w) {
                const u32x4 nw = *reinterpret_cast<const u32x4*>(a.norm_w + (size_t)e8 * 8);
                const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w}, ww[4] = {nw.x, nw.y, nw.z, nw.w};
                uint32_t ow[4];
#pragma unroll
                for (int k = 0; k < 4; k++) ow[k] = pack_bf16x2((bf_lo(rw[k]) * mul) * bf_lo(ww[k]), (bf_hi(rw[k]) * mul) * bf_hi(ww[k]));
                o = u32x4{ow[0], ow[1], ow[2], ow[3]};
            }
            xs[j * nGrp + c] = perm_x_order(o);
        }
        __syncthreads();
    }
    float best_v = -__builtin_inff();
    int best_i = 0x7fffffff;
    long kstep = 0;
    for (long s = s_begin; s < s_end; s++) {
        float acc = 0.f;
        for (int it = 0; it < iters; it++, kstep++) {
            load_step(kstep + 2, nx2);
            const int c = it * 8 + ll;
            PermLut t;
            build_perm_lut(t, cur.st, cur.ze, -(jqb * cur.st));
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const u32x4 w = cur.w[b];
                acc = perm_dot_dword(w.w, xs[(4 * b + 0) * nGrp + c], t, acc);
                acc = perm_dot_dword(w.z, xs[(4 * b + 1) * nGrp + c], t, acc);
                acc = perm_dot_dword(w.y, xs[(4 * b + 2) * nGrp + c], t, acc);
                acc = perm_dot_dword(w.x, xs[(4 * b + 3) * nGrp + c], t, acc);
            }
            cur = nxt;
            nxt = nx2;
        }
        acc += dppf<0xB1>(acc);
        acc += dppf<0x4E>(acc);
        acc += dppf<0x141>(acc);
        const int r = (int)s * 8 + sub;
        if (ll == 0 && r < jM) {
            float v = acc;
            if (a.yf) {
                a.yf[r] = v;
            } else {
                uint16_t* y = a.job[0].y + (size_t)pos * a.job[0].y_pos_stride;
                if (a.alpha != 1.0f) v = a.alpha * v;
                if (a.beta != 0.0f) v = v + a.beta * bf2f(y[r]);
                if (a.bias) v = v + bf2f(a.bias[r]);
                uint16_t o = f2bf(v);
                if (a.residual) o = f2bf(bf2f(a.residual[r]) + bf2f(o));
                y[r] = o;
                if (MODE == GEMV_ARGMAX) {
                    const float fv = bf2f(o);
                    if (fv > best_v || (fv == best_v && r < best_i)) best_v = fv, best_i = r;
                }
            }
        }
    }
    if (MODE == GEMV_ARGMAX) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const float ov = __shfl_xor(best_v, m, 64);
            const int oi = __shfl_xor(best_i, m, 64);
            if (ov > best_v || (ov == best_v && oi < best_i)) best_v = ov, best_i = oi;
        }
        float* rv = reinterpret_cast<float*>(red);
        int* ri = reinterpret_cast<int*>(rv + 16);
        __syncthreads();
        if (lane == 0) rv[wave] = best_v, ri[wave] = best_i;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; w++)
                if (rv[w] > best_v || (rv[w] == best_v && ri[w] < best_i)) best_v = rv[w], best_i = ri[w];
            a.amax_val[blockIdx.x] = best_v;
            a.amax_idx[blockIdx.x] = best_i;
        }
    }
}

// Returns KF_OK and sets *used when the table kernel took the launch; *used = false means "not applicable, use the arithmetic kernel".
int gemv_q4lut_launch(hipStream_t st, GemvLaunch& L, bool* used) {
    *used = false;
    static int knob = -99;
    if (knob == -99) {
        const char* e = getenv("KF_Q4_LUT");
        knob = e ? atoi(e) : 0; /* 0 (default) never, 1 whenever applicable, -1 by size.  Measured on 25600x5120: 33 us vs 30 us for the
                                    arithmetic kernel -- 3.7 instead of 7.7 VALU per weight, but the lookups leave the waves latency-bound
                                    (VALU 52 % / LDS 43 % busy, SQ_WAIT_ANY 53 %), so it stays an opt-in experiment. */
    }
    GemvArgs& a = L.args;
    const kf_weight* w = L.w[0];
    if (knob == 0 || L.n != 1 || L.mode == GEMV_PAIRED || w->type != KF_Q4 || w->qzeros || w->lGroup != 128) return KF_OK;
    const int K = w->ne1, M = w->ne0;
    if (K % 1024 || !w->gama) return KF_OK; /* 8 lanes x 128 weights per row step */
    if (knob < 0 && K < 4096) return KF_OK; /* short rows: too few iterations to split; the arithmetic kernel is latency-optimal there */
    if (((uintptr_t)w->data & 15) != 0) return KF_BLAS_UNALIGN;
    const int nGrp = K / 128;
    if ((unsigned long long)M * (unsigned long long)nGrp >= (1ull << 30)) return KF_OK;
    a.K = K, a.nBlk = nGrp, a.lpr_log2 = 3, a.iters = nGrp / 8;
    a.inv_dim = 1.0f / (float)K;
    a.njobs = 1;
    a.job[0].w = w->data;
    a.job[0].zero = w->gama + w->ne0 + w->ne1;
    a.job[0].step = a.job[0].zero + (size_t)M * K / 128;
    a.job[0].M = M, a.job[0].qBias = w->qBias, a.job[0].slot0 = 0;
    a.lGroup = 128, a.gshift = 0;
    const long slots = (M + 7) / 8;
    if (knob == 3 || knob == -3) { /* wave-independent register-table kernel */
        long tw = 16384;
        if (const char* e = getenv("KF_GEMV_WAVES")) tw = atol(e);
        long spw = (slots + tw - 1) / tw;
        if (spw < 1) spw = 1;
        a.spw = (int)spw, a.total_slots = (int)slots;
        const long waves = (slots + spw - 1) / spw;
        const int blocks = (int)((waves + 3) / 4);
        if (L.mode == GEMV_ARGMAX && blocks > KF_MAX_ARGMAX_PARTIALS) return KF_OK;
        const size_t smem = (size_t)K * 2 + 256;
        if (smem > 160 * 1024) return KF_OK;
        if (L.mode == GEMV_ARGMAX)
            hipLaunchKernelGGL((gemv_q4perm_kernel<GEMV_ARGMAX>), dim3(blocks), dim3(256), smem, st, a);
        else
            hipLaunchKernelGGL((gemv_q4perm_kernel<GEMV_PLAIN>), dim3(blocks), dim3(256), smem, st, a);
        L.blocks = blocks;
        *used = true;
        return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
    }
    const int NW = (slots < 2048 && a.iters >= 8) ? 8 : 4;
    long target_blocks = 4096;
    if (const char* e = getenv("KF_LUT_BLOCKS")) target_blocks = atol(e);
    long spb = (slots + target_blocks - 1) / target_blocks;
    if (spb < 1) spb = 1;
    a.spw = (int)spb, a.total_slots = (int)slots;
    const int blocks = (int)((slots + spb - 1) / spb);
    if (L.mode == GEMV_ARGMAX && blocks > KF_MAX_ARGMAX_PARTIALS) return KF_OK;
    const bool perm = knob == 2 || knob == -2;
    const size_t smem = (perm ? 0 : (size_t)NW * 4096) + (size_t)K * 2 + 256;
    if (smem > 160 * 1024) return KF_OK;
#define KF_LUT_LAUNCH(NWV, MODEV)                                                                                                 \
    do {                                                                                                                          \
        if (perm)                                                                                                                 \
            hipLaunchKernelGGL((gemv_q4lut_kernel<NWV, MODEV, true>), dim3(blocks), dim3(NWV * 64), smem, st, a);                 \
        else                                                                                                                      \
            hipLaunchKernelGGL((gemv_q4lut_kernel<NWV, MODEV, false>), dim3(blocks), dim3(NWV * 64), smem, st, a);                \
    } while (0)
    if (NW == 8) {
        if (L.mode == GEMV_ARGMAX)
            KF_LUT_LAUNCH(8, GEMV_ARGMAX);
        else
            KF_LUT_LAUNCH(8, GEMV_PLAIN);
    } else {
        if (L.mode == GEMV_ARGMAX)
            KF_LUT_LAUNCH(4, GEMV_ARGMAX);
        else
            KF_LUT_LAUNCH(4, GEMV_PLAIN);
    }
#undef KF_LUT_LAUNCH
    L.blocks = blocks;
    *used = true;
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
