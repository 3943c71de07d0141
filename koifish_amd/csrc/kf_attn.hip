// kf_attn.hip -- decode attention for one new token (GQA, full causal length), gfx950 / wave64.
//
// Replaces the reference's three launches attention_qk_kernel / CU_softmax_multihead / attention_v_kernel
// (QKV.cu:669-673; operator.cuh:572-632, 251-277, 649-668 -- one thread per key capped at 1024 keys, one
// THREAD per head for the softmax) with a split-KV pass (grid = splits x kv-heads, every workgroup streams its
// slice of K and V once with 16-byte loads and keeps an online softmax per query head of the GQA group) and a
// merge.  The q/k RMSNorm + RoPE of ROPE::cuInfer (rope.cu:645-672 -> CU_rms_forward_v2 layernorm.cuh:129-167,
// CU_rope2_v0 operator.cuh:734-772) is folded into the prologue; the workgroup whose slice holds the new
// position writes the normed+roped key into the cache row.  HBM-bound: bytes = 2 * (pos+1) * kv_dim * 2.
//
// Arithmetic (matches oracle/kf_oracle.c kfo_attn_decode mode FUSED): score = bf16(dot / sqrtf(hd)) -- the same
// bf16 store the reference makes (qk_v is floatX) -- then fp32 softmax with the fixed kf_expf and a single bf16
// store of out = (sum e_t v_t) * (1 / sum e_t).
#include "kf_kernels.h"

namespace kf {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 2^lg (<= 16) lanes of each aligned lane group, DPP only
__device__ __forceinline__ float group_sum16(float v, int lg) {
    if (lg >= 1) v += dpp_f<0xB1>(v);
    if (lg >= 2) v += dpp_f<0x4E>(v);
    if (lg >= 3) v += dpp_f<0x141>(v);
    if (lg >= 4) v += dpp_f<0x140>(v);
    return v;
}

// Prepare one head: optional per-head RMSNorm (s rounded to bf16 first, then (a*s)*w, RN store) and rotate-half
// RoPE from the host-built (cos,sin) table.  One wave per head; lane j handles the pair (j, j + hd/2).
// Result: bf16-rounded values as floats in dst[hd].
__device__ __forceinline__ void prep_head(const uint16_t* __restrict__ src, const uint16_t* __restrict__ wn, const float* __restrict__ tab_pos,
                                          int hd, float eps, float* dst) {
    const int lane = threadIdx.x & 63, half = hd >> 1;
    for (int j0 = 0; j0 < half; j0 += 64) { /* hd <= 128: one trip */
        const int j = j0 + lane;
        const bool act = j < half;
        float x0 = act ? bf2f(src[j]) : 0.f, x1 = act ? bf2f(src[j + half]) : 0.f;
        if (wn) {
            // hd <= 128 means the whole head sits in this one trip, so the wave sum is the head's sum
            double ss = wave_sum_f64(fma((double)x0, (double)x0, (double)x1 * (double)x1));
            const float s0 = 1.0f / sqrtf((float)ss / (float)hd + eps);
            const float s = round_bf16(s0);
            if (act) {
                x0 = round_bf16(x0 * s * bf2f(wn[j]));
                x1 = round_bf16(x1 * s * bf2f(wn[j + half]));
            }
        }
        if (tab_pos && act) {
            const float c = tab_pos[2 * j], sn = tab_pos[2 * j + 1];
            const float a = x0 * c, b = x1 * sn, cc = x0 * sn, d = x1 * c;
            x0 = round_bf16(a - b);
            x1 = round_bf16(cc + d);
        }
        if (act) dst[j] = x0, dst[j + half] = x1;
    }
}

constexpr int ATTN_U = 4; /* key tiles kept in flight per wave */

template <int GQ>
__global__ void __launch_bounds__(256) attn_partial_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int hd = a.hd, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* qf = reinterpret_cast<float*>(smem_raw);  // [GQ][hd]
    float* knew = qf + GQ * hd;                      // [hd]
    float* comb = knew + hd;                         // [4*KPW][GQ][PS]

    const int pos = a.d_pos ? *a.d_pos : a.pos;
    const int len = pos + 1;
    const int split = blockIdx.x, kvh = blockIdx.y, nsp = a.n_splits;
    const int chunk = (len + nsp - 1) / nsp;
    const int t0 = split * chunk;
    int t1 = t0 + chunk;
    if (t1 > len) t1 = len;
    const int h0 = kvh * GQ;
    const int PS = hd + 4; /* {acc[hd], m, l, pad, pad}: keeps every partial 16-byte aligned */

    if (t0 >= len) { /* empty slice: neutral partial */
        for (int i = tid; i < GQ * PS; i += blockDim.x) {
            const int hq = i / PS, d = i - hq * PS;
            a.part[((size_t)(h0 + hq) * nsp + split) * PS + d] = (d == hd) ? -__builtin_inff() : 0.f;
        }
        return;
    }

    // LPK lanes per key (8 dims each), KPW keys per wave step, 4 waves interleaved over the slice
    const int LPK = hd >> 3, KPW = 64 / LPK, lpk_log2 = __builtin_ctz(LPK);
    const int grp = lane / LPK, d0 = (lane - grp * LPK) * 8;
    const bool has_new = a.k_raw != nullptr;
    const int tstart = t0 + wave * KPW + grp, tstride = 4 * KPW;

    // ---- issue the first K/V tiles before anything that depends on the previous kernel's q
    u32x4 kk[ATTN_U], vv[ATTN_U];
    auto issue = [&](int tb) {
#pragma unroll
        for (int u = 0; u < ATTN_U; u++) {
            const int t = tb + u * tstride;
            kk[u] = u32x4{0, 0, 0, 0}, vv[u] = u32x4{0, 0, 0, 0};
            if (t < t1) {
                const size_t off = (size_t)t * a.kv_stride + (size_t)kvh * hd + d0;
                vv[u] = *reinterpret_cast<const u32x4*>(a.vcache + off);
                if (!(has_new && t == pos)) kk[u] = *reinterpret_cast<const u32x4*>(a.kcache + off);
            }
        }
    };
    issue(tstart);

    // ---- prologue: q heads of this group, and the new key when it lies in this slice
    const float* tab_pos = a.rope_table ? a.rope_table + (size_t)pos * hd : nullptr;
    for (int hq = wave; hq < GQ; hq += 4) prep_head(a.q + (size_t)(h0 + hq) * hd, a.rope_table ? a.wq_norm : nullptr, tab_pos, hd, a.eps, qf + hq * hd);
    const bool own_new = has_new && (pos >= t0) && (pos < t1);
    if (own_new && wave == (GQ & 3)) prep_head(a.k_raw + (size_t)kvh * hd, a.wk_norm, tab_pos, hd, a.eps, knew);
    __syncthreads();
    if (own_new) {
        uint16_t* krow = a.kcache + (size_t)pos * a.kv_stride + (size_t)kvh * hd;
        for (int i = tid; i < hd; i += blockDim.x) krow[i] = f2bf(knew[i]);
    }

    float qreg[GQ][8];
#pragma unroll
    for (int hq = 0; hq < GQ; hq++)
#pragma unroll
        for (int i = 0; i < 8; i++) qreg[hq][i] = qf[hq * hd + d0 + i];

    float m[GQ], l[GQ], acc[GQ][8];
#pragma unroll
    for (int hq = 0; hq < GQ; hq++) {
        m[hq] = -__builtin_inff(), l[hq] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) acc[hq][i] = 0.f;
    }
    const float den = a.inv_sqrt_hd_den; /* sqrtf(hd): score /= sqrtf(head_dim) (operator.cuh:630) */
    for (int tb = tstart; tb - grp < t1; tb += ATTN_U * tstride) {
        u32x4 ck[ATTN_U], cv[ATTN_U];
#pragma unroll
        for (int u = 0; u < ATTN_U; u++) ck[u] = kk[u], cv[u] = vv[u];
        if (tb - grp + ATTN_U * tstride < t1) issue(tb + ATTN_U * tstride);
#pragma unroll
        for (int u = 0; u < ATTN_U; u++) {
            const int t = tb + u * tstride;
            const bool valid = t < t1;
            float kf_[8], vf_[8];
            const uint32_t vw[4] = {cv[u].x, cv[u].y, cv[u].z, cv[u].w}, kw[4] = {ck[u].x, ck[u].y, ck[u].z, ck[u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                vf_[2 * i] = bf_lo(vw[i]), vf_[2 * i + 1] = bf_hi(vw[i]);
                kf_[2 * i] = bf_lo(kw[i]), kf_[2 * i + 1] = bf_hi(kw[i]);
            }
            if (has_new && valid && t == pos) {
#pragma unroll
                for (int i = 0; i < 8; i++) kf_[i] = knew[d0 + i];
            }
            if (tb - grp + u * tstride >= t1) continue; /* wave-uniform: the whole tile is past the slice */
#pragma unroll
            for (int hq = 0; hq < GQ; hq++) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; i++) s = fmaf(qreg[hq][i], kf_[i], s);
                s = group_sum16(s, lpk_log2);
                s = round_bf16(s / den);
                if (valid) {
                    if (s > m[hq]) {
                        const float sc = kf_expf(m[hq] - s);
                        l[hq] *= sc;
#pragma unroll
                        for (int i = 0; i < 8; i++) acc[hq][i] *= sc;
                        m[hq] = s;
                    }
                    const float p = kf_expf(s - m[hq]);
                    l[hq] += p;
#pragma unroll
                    for (int i = 0; i < 8; i++) acc[hq][i] = fmaf(p, vf_[i], acc[hq][i]);
                }
            }
        }
    }

    // ---- combine the 4*KPW key groups of this workgroup through LDS
    const int slot = wave * KPW + grp, nslot = 4 * KPW;
#pragma unroll
    for (int hq = 0; hq < GQ; hq++) {
        float* c = comb + ((size_t)slot * GQ + hq) * PS;
#pragma unroll
        for (int i = 0; i < 8; i++) c[d0 + i] = acc[hq][i];
        if (d0 == 0) c[hd] = m[hq], c[hd + 1] = l[hq];
    }
    __syncthreads();
    for (int i = tid; i < GQ * hd; i += blockDim.x) {
        const int hq = i / hd, d = i - hq * hd;
        float M = -__builtin_inff();
        for (int s = 0; s < nslot; s++) M = fmaxf(M, comb[((size_t)s * GQ + hq) * PS + hd]);
        float o = 0.f, L = 0.f;
        for (int s = 0; s < nslot; s++) {
            const float* c = comb + ((size_t)s * GQ + hq) * PS;
            const float ms = c[hd];
            if (ms == -__builtin_inff()) continue;
            const float sc = kf_expf(ms - M);
            o = fmaf(c[d], sc, o);
            L = fmaf(c[hd + 1], sc, L);
        }
        float* dst = a.part + ((size_t)(h0 + hq) * nsp + split) * PS;
        dst[d] = o;
        if (d == 0) dst[hd] = M, dst[hd + 1] = L;
    }
}

// merge the per-slice partials: grid = n_head, block = hd.  (The decode step folds this into the o_proj mat-vec's
// prologue -- kf_gemv.hip -- with the same operation order; this kernel serves the stand-alone kf_attn_decode.)
__global__ void attn_merge_kernel(const float* __restrict__ part, uint16_t* __restrict__ out, int hd, int nsp) {
    __shared__ float ms[KF_ATTN_MAX_SPLITS], sc[KF_ATTN_MAX_SPLITS], ls[KF_ATTN_MAX_SPLITS];
    const int h = blockIdx.x, d = threadIdx.x, PS = hd + 4;
    const float* p = part + (size_t)h * nsp * PS;
    for (int s = d; s < nsp; s += blockDim.x) ms[s] = p[(size_t)s * PS + hd], ls[s] = p[(size_t)s * PS + hd + 1];
    __syncthreads();
    float M = -__builtin_inff();
    for (int s = 0; s < nsp; s++) M = fmaxf(M, ms[s]);
    for (int s = d; s < nsp; s += blockDim.x) sc[s] = (ms[s] == -__builtin_inff()) ? 0.f : kf_expf(ms[s] - M);
    __syncthreads();
    float o = 0.f, L = 0.f;
#pragma unroll 4
    for (int s = 0; s < nsp; s++) {
        if (ms[s] == -__builtin_inff()) continue;
        o = fmaf(p[(size_t)s * PS + d], sc[s], o);
        L = fmaf(ls[s], sc[s], L);
    }
    const float inv = 1.0f / L;
    out[(size_t)h * hd + d] = f2bf(o * inv);
}

// standalone ROPE::cuInfer: grid = n_head + n_kv, one wave each
__global__ void __launch_bounds__(64) qknorm_rope_kernel(uint16_t* q, uint16_t* k, const uint16_t* wq, const uint16_t* wk, const float* table, int pos_,
                                                         const int* d_pos, int n_head, int n_kv, int hd, float eps) {
    __shared__ float buf[256];
    const int pos = d_pos ? *d_pos : pos_;
    const int b = blockIdx.x;
    uint16_t* src = b < n_head ? q + (size_t)b * hd : k + (size_t)(b - n_head) * hd;
    const uint16_t* wn = b < n_head ? wq : wk;
    prep_head(src, wn, table ? table + (size_t)pos * hd : nullptr, hd, eps, buf);
    __syncthreads();
    for (int i = threadIdx.x; i < hd; i += 64) src[i] = f2bf(buf[i]);
}

// slices: ~64 keys each, enough workgroups to cover the chip, bounded by the scratch layout
int attn_splits(int pos_bound, int n_kv) {
    int nsp = (pos_bound + 1 + 63) / 64;
    int cap = 512 / n_kv;
    if (cap < 1) cap = 1;
    if (nsp > cap) nsp = cap;
    if (nsp > KF_ATTN_MAX_SPLITS) nsp = KF_ATTN_MAX_SPLITS;
    if (nsp < 1) nsp = 1;
    return nsp;
}

int attn_launch(hipStream_t st, AttnArgs& a, bool merge) {
    const int hd = a.hd;
    if (hd < 64 || hd > 128 || (hd & (hd - 1)) != 0) return KF_INVALID_ARGS; /* 8 dims per lane, one RoPE trip per wave */
    if (a.n_kv <= 0 || a.n_head % a.n_kv != 0) return KF_INVALID_ARGS;
    const int GQ = a.n_head / a.n_kv;
    const int nsp = attn_splits(a.pos, a.n_kv);
    a.n_splits = nsp;
    a.inv_sqrt_hd_den = sqrtf((float)hd);
    const int KPW = 64 / (hd >> 3);
    const size_t smem = sizeof(float) * ((size_t)GQ * hd + hd + (size_t)4 * KPW * GQ * (hd + 4));
    dim3 grid(nsp, a.n_kv);
    switch (GQ) {
        case 1: hipLaunchKernelGGL((attn_partial_kernel<1>), grid, dim3(256), smem, st, a); break;
        case 2: hipLaunchKernelGGL((attn_partial_kernel<2>), grid, dim3(256), smem, st, a); break;
        case 4: hipLaunchKernelGGL((attn_partial_kernel<4>), grid, dim3(256), smem, st, a); break;
        case 8: hipLaunchKernelGGL((attn_partial_kernel<8>), grid, dim3(256), smem, st, a); break;
        default: return KF_INVALID_ARGS;
    }
    if (merge) hipLaunchKernelGGL(attn_merge_kernel, dim3(a.n_head), dim3(hd), 0, st, a.part, a.out, hd, nsp);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

int qknorm_rope_launch(hipStream_t st, uint16_t* q, uint16_t* k, const uint16_t* wq, const uint16_t* wk, const float* table, int pos,
                       const int* d_pos, int n_head, int n_kv, int hd, float eps) {
    if (hd < 64 || hd > 128 || (hd & (hd - 1)) != 0) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(qknorm_rope_kernel, dim3(n_head + (k ? n_kv : 0)), dim3(64), 0, st, q, k, wq, wk, table, pos, d_pos, n_head, n_kv, hd, eps);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
