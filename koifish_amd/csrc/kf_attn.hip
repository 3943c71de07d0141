// kf_attn.hip -- decode attention for one new token (GQA, full causal length), gfx950 / wave64.
//
// Replaces the reference's three launches attention_qk_kernel / CU_softmax_multihead / attention_v_kernel
// (QKV.cu:669-673; operator.cuh:572-632, 251-277, 649-668 -- one thread per key capped at 1024 keys, one
// THREAD per head for the softmax) by ONE launch: grid = slices x kv-heads; every workgroup streams its slice
// of K and V once (16-byte loads issued before anything else), keeps a softmax for the query heads of its GQA
// group against a workgroup-wide running maximum, and publishes {acc[hd], m, l}; the workgroup that arrives
// last for a kv-head merges the slices and writes the bf16 output (no separate merge launch).
// The q/k RMSNorm + RoPE of ROPE::cuInfer (rope.cu:645-672 -> CU_rms_forward_v2 layernorm.cuh:129-167,
// CU_rope2_v0 operator.cuh:734-772) is folded into the prologue; the workgroup whose slice holds the new
// position writes the normed+roped key into the cache row.  HBM-bound: bytes = 2 * (pos+1) * kv_dim * 2.
//
// Arithmetic (= oracle/kf_oracle.c kfo_attn_decode mode CANON, bit for bit; kf_attn_common.h states it): score = bf16(dot * (1 / sqrtf(hd))) -- the same
// bf16 store the reference makes (qk_v is floatX) -- with the dot as per-lane chains of v_fma_f32 + a tree; softmax weights as f * 2^(n - m) from a fixed
// polynomial with exact power-of-two rescales between lanes, waves and slices; fp64 sums; one bf16 store of out = (float)(O / L).
//
// Cross-workgroup hand-off (MI355X_MICROARCH.md "Valid forms", table row 1): partials are written with
// agent-scope relaxed atomic stores (write-through `sc1`), every storing wave drains vmcnt, the workgroup
// barriers, ONE lane adds to the kv-head's arrival counter (agent-scope atomic); the workgroup whose add returned
// nsp-1 reads every partial with agent-scope relaxed atomic loads (`sc1`) after a barrier, and re-zeroes the counter.
#include <stdlib.h>

#include "kf_attn_common.h"

namespace kf {

template <int GQ, int NW, int HD>
__global__ void __launch_bounds__(NW * 64) attn_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int hd = HD, hd_log2 = HD == 128 ? 7 : 6; /* head_dim 64 or 128: compile-time, so that the lane-group reductions are straight-line code */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PS = hd + 2; /* {O[hd], L, m} as doubles */
    double* comb = reinterpret_cast<double*>(smem_raw);        // [NW][GQ][PS]
    uint16_t* qb = reinterpret_cast<uint16_t*>(comb + NW * GQ * PS);  // [GQ][hd] prepared q, bf16 bits
    uint16_t* knew = qb + GQ * hd;                         // [hd]
    int* flag = reinterpret_cast<int*>(knew + hd);

    const int split = blockIdx.x, kvh = (int)blockIdx.y / a.gq_split, gpart = (int)blockIdx.y - kvh * a.gq_split, nsp = a.n_splits;
    // token batch (prefill): blockIdx.z = token, one slice per kv-head, position pos0 + token, q / out rows q_stride apart
    const int pos_l = a.pos + (int)blockIdx.z;
    const uint16_t* const qsrc = a.q + (size_t)blockIdx.z * a.q_stride;
    uint16_t* const odst = a.out + (size_t)blockIdx.z * a.q_stride;
    const int chunk = a.chunk; /* keys per slice, fixed by the launch bound so that the K/V stream can start before pos is known */
    const int t0 = split * chunk;
    const int h0 = (kvh * a.gq_split + gpart) * GQ; /* GQ = the heads of THIS workgroup */

    // LPK lanes per key (8 dims each), KPW keys per wave step, the waves interleaved over the slice
    constexpr int LPK = hd >> 3, KPW = 64 / LPK, lpk_log2 = hd_log2 - 3;
    const int grp = lane >> lpk_log2, d0 = (lane & (LPK - 1)) * 8;
    const bool has_new = a.k_raw != nullptr;
    const int tstart = t0 + wave * KPW + grp, tstride = NW * KPW;

    // ---- issue the first K/V tiles before anything else: they depend neither on the position (read from device memory in
    // graph replay) nor on the previous kernel's q.  Rows up to the launch bound exist in the cache; rows past the real position
    // are masked later, and the row AT the position is replaced by the freshly normed+roped key.
    u32x4 kk[ATTN_U], vv[ATTN_U];
    auto issue = [&](int tb, int tend) {
#pragma unroll
        for (int u = 0; u < ATTN_U; u++) {
            const int t = tb + u * tstride;
            kk[u] = u32x4{0, 0, 0, 0}, vv[u] = u32x4{0, 0, 0, 0};
            if (t < tend) {
                const size_t off = (size_t)t * a.kv_stride + (size_t)kvh * hd + d0;
                kk[u] = *reinterpret_cast<const u32x4*>(a.kcache + off);
                vv[u] = *reinterpret_cast<const u32x4*>(a.vcache + off);
            }
        }
    };
    {
        int tb_end = t0 + chunk;
        if (tb_end > pos_l + 1) tb_end = pos_l + 1; /* a.pos is the launch bound here */
        issue(tstart, tb_end);
    }
    // ... and the q heads (+ the raw new key) this wave will prepare: they do not depend on the position either
    constexpr int NQ = (GQ + NW - 1) / NW;
    const bool qnorm = a.rope_table && a.wq_norm;
    HeadRaw qraw[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int hq = wave + i * NW;
        qraw[i] = load_head(qsrc + (size_t)(h0 + (hq < GQ ? hq : 0)) * hd, qnorm ? a.wq_norm : nullptr, hd);
    }
    const HeadRaw kraw = load_head(has_new ? a.k_raw + (size_t)kvh * hd : qsrc, a.wk_norm, hd);

    const int pos = a.d_pos ? *a.d_pos : pos_l;
    const int len = pos + 1;
    int t1 = t0 + chunk;
    if (t1 > len) t1 = len;
    const bool empty = t0 >= len;
    double* const part = reinterpret_cast<double*>(a.part); /* [n_head][n_splits][PS] doubles */

    if (!empty) {
        // ---- prologue: q heads of this group, and the new key when it lies in this slice
        const float* tab_pos = a.rope_table ? a.rope_table + (size_t)pos * hd : nullptr;
#pragma unroll
        for (int i = 0; i < NQ; i++)
            if (wave + i * NW < GQ) prep_head(qraw[i], qnorm, tab_pos, hd, a.eps, qb + (wave + i * NW) * hd);
        const bool own_new = has_new && (pos >= t0) && (pos < t1);
        if (own_new && wave == (GQ % NW)) prep_head(kraw, a.wk_norm != nullptr, tab_pos, hd, a.eps, knew);
        __syncthreads();
        if (own_new && gpart == 0) {
            uint16_t* krow = a.kcache + (size_t)pos * a.kv_stride + (size_t)kvh * hd;
            for (int i = tid; i < hd; i += blockDim.x) krow[i] = knew[i];
        }
        float qf[GQ][8]; /* this lane's 8 dims of every q head */
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            const u32x4 qv = *reinterpret_cast<const u32x4*>(qb + hq * hd + d0);
            const uint32_t q4[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int i = 0; i < 4; i++) qf[hq][2 * i] = bf_lo(q4[i]), qf[hq][2 * i + 1] = bf_hi(q4[i]);
        }
        CanonAcc<GQ> A;
        A.init();
        const float rden = 1.0f / a.inv_sqrt_hd_den; /* score /= sqrtf(head_dim) (operator.cuh:630), as a multiply by the rounded reciprocal */
        for (int tb = tstart; tb - grp - wave * KPW < t1; tb += ATTN_U * tstride) { /* workgroup-uniform trip count */
            u32x4 ck[ATTN_U], cv[ATTN_U];
            bool valid[ATTN_U];
#pragma unroll
            for (int u = 0; u < ATTN_U; u++) {
                const int t = tb + u * tstride;
                ck[u] = kk[u], cv[u] = vv[u], valid[u] = t < t1;
                if (has_new && valid[u] && t == pos) ck[u] = *reinterpret_cast<const u32x4*>(knew + d0);
            }
            if (tb - grp - wave * KPW + ATTN_U * tstride < t1) issue(tb + ATTN_U * tstride, t1);
            canon_batch<GQ, LPK>(A, qf, ck, cv, valid, lpk_log2, rden);
        }
        // ---- the wave's key groups summed (reduce-scatter by row swaps), the waves through LDS: exact rescales to the slice's maximum exponent
        canon_wave_to_lds<GQ, LPK>(A, comb + (size_t)wave * GQ * PS, hd, lane, d0);
        __syncthreads();
        for (int i = tid; i < GQ * hd; i += blockDim.x) {
            const int hq = i >> hd_log2, d = i & (hd - 1);
            float ms = -__builtin_inff();
#pragma unroll
            for (int sl = 0; sl < NW; sl++) ms = fmaxf(ms, (float)comb[((size_t)sl * GQ + hq) * PS + hd + 1]);
            double o = 0.0, L = 0.0;
#pragma unroll
            for (int sl = 0; sl < NW; sl++) {
                const double* c = comb + ((size_t)sl * GQ + hq) * PS;
                const int e = canon_shift((float)c[hd + 1] - ms);
                o += ldexp_d(c[d], e);
                L += ldexp_d(c[hd], e);
            }
            if (nsp == 1) {
                odst[(size_t)(h0 + hq) * hd + d] = f2bf((float)(o / L));
            } else {
                double* dst = part + ((size_t)(h0 + hq) * nsp + split) * PS;
                __hip_atomic_store(dst + d, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (d == 0) __hip_atomic_store(dst + hd, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_store(dst + hd + 1, (double)ms, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (nsp > 1) { /* empty slice: neutral partial, but it still arrives */
        for (int i = tid; i < GQ * hd; i += blockDim.x) {
            const int hq = i >> hd_log2, d = i & (hd - 1);
            double* dst = part + ((size_t)(h0 + hq) * nsp + split) * PS;
            __hip_atomic_store(dst + d, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == 0) __hip_atomic_store(dst + hd, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_store(dst + hd + 1, -(double)__builtin_inff(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (nsp == 1) return;

    // ---- arrival; the last workgroup of this kv-head merges
    int* const counter = a.counters + (int)blockIdx.y * a.cnt_stride; /* per (kv-head, part): its slices' workgroups meet here */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = (old == nsp - 1);
    }
    __syncthreads();
    if (!flag[0]) return;

    // Every thread merges its output elements over the slices: the head is the same for the whole wave (64 | hd), lane sp of the wave fetches (m, L) of slice sp,
    // the maximum exponent and the rescaled L are formed in registers; the element's own chain walks the slices in order (fp64: the order does not matter).
    constexpr int NT = NW * 64;
    constexpr int NV = (GQ * 128 + NT - 1) / NT; /* output elements per thread (hd <= 128) */
#pragma unroll
    for (int e = 0; e < NV; e++) {
        const int i = tid + e * NT;
        const bool in = i < GQ * hd;
        const int hq = in ? (i >> hd_log2) : 0, d = i & (hd - 1);
        const double* p = part + (size_t)(h0 + hq) * nsp * PS;
        const bool mine = in && lane < nsp;
        const float ms = mine ? (float)__hip_atomic_load(p + (size_t)lane * PS + hd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -__builtin_inff();
        const double ls = mine ? __hip_atomic_load(p + (size_t)lane * PS + hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        const float Mx = wave_max(ms);
        const int sh = canon_shift(ms - Mx);
        const double L = wave_sum_f64_fast(ldexp_d(ls, sh));
        double o = 0.0;
        for (int sp = 0; sp < nsp; sp++) {
            const double v = in ? __hip_atomic_load(p + (size_t)sp * PS + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            o += ldexp_d(v, __builtin_amdgcn_readlane(sh, sp));
        }
        if (in) odst[(size_t)(h0 + hq) * hd + d] = f2bf((float)(o / L));
    }
    if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the fp32 form of rounds 1-2 (kf_set_canonical(ctx, 0), the default): scores by v_dot2c_f32_bf16, exp by v_exp_f32(x * log2 e), a workgroup-wide running
// maximum, fp32 sums, slices merged with fp32 exponentials.  <= 1 bf16 ulp / 2^-10 of the scale from the oracle's FUSED mode; about half the vector work of the
// canonical form above (no fp64).  Partials: [n_head][n_splits][hd + 4] floats in the same scratch.
template <int GQ, int NW, int HD>
__global__ void __launch_bounds__(NW * 64) attn_fast_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int hd = HD, hd_log2 = HD == 128 ? 7 : 6; /* head_dim 64 or 128: compile-time, so that the lane-group reductions are straight-line code */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PS = hd + 4; /* {acc[hd], m, l, pad, pad} */
    uint16_t* qb = reinterpret_cast<uint16_t*>(smem_raw);  // [GQ][hd] prepared q, bf16 bits
    uint16_t* knew = qb + GQ * hd;                         // [hd]
    float* wmax = reinterpret_cast<float*>(knew + hd);     // [NW][GQ]
    int* flag = reinterpret_cast<int*>(wmax + NW * GQ);
    float* comb = reinterpret_cast<float*>(flag + 4);  // [NW][GQ][PS]

    const int split = blockIdx.x, kvh = (int)blockIdx.y / a.gq_split, gpart = (int)blockIdx.y - kvh * a.gq_split, nsp = a.n_splits;
    // token batch (prefill): blockIdx.z = token, one slice per kv-head, position pos0 + token, q / out rows q_stride apart
    const int pos_l = a.pos + (int)blockIdx.z;
    const uint16_t* const qsrc = a.q + (size_t)blockIdx.z * a.q_stride;
    uint16_t* const odst = a.out + (size_t)blockIdx.z * a.q_stride;
    const int chunk = a.chunk; /* keys per slice, fixed by the launch bound so that the K/V stream can start before pos is known */
    const int t0 = split * chunk;
    const int h0 = (kvh * a.gq_split + gpart) * GQ; /* GQ = the heads of THIS workgroup */

    // LPK lanes per key (8 dims each), KPW keys per wave step, the waves interleaved over the slice
    constexpr int LPK = hd >> 3, KPW = 64 / LPK, lpk_log2 = hd_log2 - 3;
    const int grp = lane >> lpk_log2, d0 = (lane & (LPK - 1)) * 8;
    const bool has_new = a.k_raw != nullptr;
    const int tstart = t0 + wave * KPW + grp, tstride = NW * KPW;

    // ---- issue the first K/V tiles before anything else: they depend neither on the position (read from device memory in
    // graph replay) nor on the previous kernel's q.  Rows up to the launch bound exist in the cache; rows past the real position
    // are masked later, and the row AT the position is replaced by the freshly normed+roped key.
    u32x4 kk[ATTN_U], vv[ATTN_U];
    auto issue = [&](int tb, int tend) {
#pragma unroll
        for (int u = 0; u < ATTN_U; u++) {
            const int t = tb + u * tstride;
            kk[u] = u32x4{0, 0, 0, 0}, vv[u] = u32x4{0, 0, 0, 0};
            if (t < tend) {
                const size_t off = (size_t)t * a.kv_stride + (size_t)kvh * hd + d0;
                kk[u] = *reinterpret_cast<const u32x4*>(a.kcache + off);
                vv[u] = *reinterpret_cast<const u32x4*>(a.vcache + off);
            }
        }
    };
    {
        int tb_end = t0 + chunk;
        if (tb_end > pos_l + 1) tb_end = pos_l + 1; /* a.pos is the launch bound here */
        issue(tstart, tb_end);
    }
    // ... and the q heads (+ the raw new key) this wave will prepare: they do not depend on the position either
    constexpr int NQ = (GQ + NW - 1) / NW;
    const bool qnorm = a.rope_table && a.wq_norm;
    HeadRaw qraw[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int hq = wave + i * NW;
        qraw[i] = load_head(qsrc + (size_t)(h0 + (hq < GQ ? hq : 0)) * hd, qnorm ? a.wq_norm : nullptr, hd);
    }
    const HeadRaw kraw = load_head(has_new ? a.k_raw + (size_t)kvh * hd : qsrc, a.wk_norm, hd);

    const int pos = a.d_pos ? *a.d_pos : pos_l;
    const int len = pos + 1;
    int t1 = t0 + chunk;
    if (t1 > len) t1 = len;
    const bool empty = t0 >= len;

    if (!empty) {
        // ---- prologue: q heads of this group, and the new key when it lies in this slice
        const float* tab_pos = a.rope_table ? a.rope_table + (size_t)pos * hd : nullptr;
#pragma unroll
        for (int i = 0; i < NQ; i++)
            if (wave + i * NW < GQ) prep_head(qraw[i], qnorm, tab_pos, hd, a.eps, qb + (wave + i * NW) * hd);
        const bool own_new = has_new && (pos >= t0) && (pos < t1);
        if (own_new && wave == (GQ % NW)) prep_head(kraw, a.wk_norm != nullptr, tab_pos, hd, a.eps, knew);
        __syncthreads();
        if (own_new && gpart == 0) {
            uint16_t* krow = a.kcache + (size_t)pos * a.kv_stride + (size_t)kvh * hd;
            for (int i = tid; i < hd; i += blockDim.x) krow[i] = knew[i];
        }
        u32x4 qreg[GQ]; /* this lane's 8 dims of every q head, packed bf16 */
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) qreg[hq] = *reinterpret_cast<const u32x4*>(qb + hq * hd + d0);

        float M[GQ], l[GQ], acc[GQ][8];
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            M[hq] = -__builtin_inff(), l[hq] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) acc[hq][i] = 0.f;
        }
        const float rden = 1.0f / a.inv_sqrt_hd_den; /* score /= sqrtf(head_dim) (operator.cuh:630), as a multiply by the rounded reciprocal */
        for (int tb = tstart; tb - grp - wave * KPW < t1; tb += ATTN_U * tstride) { /* workgroup-uniform trip count */
            u32x4 ck[ATTN_U], cv[ATTN_U];
#pragma unroll
            for (int u = 0; u < ATTN_U; u++) ck[u] = kk[u], cv[u] = vv[u];
            if (tb - grp - wave * KPW + ATTN_U * tstride < t1) issue(tb + ATTN_U * tstride, t1);
            // scores of this batch
            float s[ATTN_U][GQ], bm[GQ];
#pragma unroll
            for (int hq = 0; hq < GQ; hq++) bm[hq] = -__builtin_inff();
#pragma unroll
            for (int u = 0; u < ATTN_U; u++) {
                const int t = tb + u * tstride;
                const bool valid = t < t1;
                u32x4 kw = ck[u];
                if (has_new && valid && t == pos) kw = *reinterpret_cast<const u32x4*>(knew + d0);
#pragma unroll
                for (int hq = 0; hq < GQ; hq++) {
                    float d = dot2_bf16(qreg[hq].x, kw.x, 0.f);
                    d = dot2_bf16(qreg[hq].y, kw.y, d);
                    d = dot2_bf16(qreg[hq].z, kw.z, d);
                    d = dot2_bf16(qreg[hq].w, kw.w, d);
                    d = group_sum16(d, lpk_log2);
                    d = round_bf16(d * rden);
                    s[u][hq] = valid ? d : -__builtin_inff();
                    bm[hq] = fmaxf(bm[hq], s[u][hq]);
                }
            }
            // workgroup-wide maximum of the batch -> one running maximum shared by every lane
#pragma unroll
            for (int hq = 0; hq < GQ; hq++) {
                bm[hq] = xmax32(xmax16(bm[hq]));
                if (LPK < 16) bm[hq] = fmaxf(bm[hq], dpp_f<0x128>(bm[hq]));
            }
            __syncthreads(); /* previous batch's readers of wmax are done */
            if (lane == 0) {
#pragma unroll
                for (int hq = 0; hq < GQ; hq++) wmax[wave * GQ + hq] = bm[hq];
            }
            __syncthreads();
#pragma unroll
            for (int hq = 0; hq < GQ; hq++) {
                float Mb = wmax[hq];
#pragma unroll
                for (int w2 = 1; w2 < NW; w2++) Mb = fmaxf(Mb, wmax[w2 * GQ + hq]);
                if (Mb > M[hq]) {
                    const float sc = fast_exp(M[hq] - Mb);
                    l[hq] *= sc;
#pragma unroll
                    for (int i = 0; i < 8; i++) acc[hq][i] *= sc;
                    M[hq] = Mb;
                }
            }
#pragma unroll
            for (int u = 0; u < ATTN_U; u++) {
                float vf_[8];
                const uint32_t vw[4] = {cv[u].x, cv[u].y, cv[u].z, cv[u].w};
#pragma unroll
                for (int i = 0; i < 4; i++) vf_[2 * i] = bf_lo(vw[i]), vf_[2 * i + 1] = bf_hi(vw[i]);
#pragma unroll
                for (int hq = 0; hq < GQ; hq++) {
                    const float p = fast_exp(s[u][hq] - M[hq]); /* -inf (masked key) -> 0 */
                    l[hq] += p;
#pragma unroll
                    for (int i = 0; i < 8; i++) acc[hq][i] = fmaf(p, vf_[i], acc[hq][i]);
                }
            }
        }

        // ---- sum the key groups (same reference maximum everywhere: plain sums).  Inside the wave a reduce-scatter by row swaps:
        // swapping the halves of (acc[i], acc[i+4]) and adding leaves dims i in lanes 0-31 and i+4 in lanes 32-63, the same on 16-lane
        // rows leaves row R with the totals of dims 2R and 2R+1 -- 6 swaps + 6 adds per head instead of 8 butterflies
        const int row = lane >> 4;
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            float s1[4], r2[2];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[hq][i]), __float_as_uint(acc[hq][i + 4]), false, false);
                s1[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s1[i]), __float_as_uint(s1[i + 2]), false, false);
                r2[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                if (LPK < 16) r2[i] += dpp_f<0x128>(r2[i]); /* two key groups per row: row_ror:8 */
            }
            float lt = xsum16(xsum32(l[hq]));
            if (LPK < 16) lt += dpp_f<0x128>(lt);
            float* c = comb + ((size_t)wave * GQ + hq) * PS;
            if (LPK == 16 || (lane & 8) == 0) *reinterpret_cast<float2*>(c + d0 + 2 * row) = float2{r2[0], r2[1]};
            if (lane == 0) c[hd] = lt;
        }
        __syncthreads();
        for (int i = tid; i < GQ * hd; i += blockDim.x) {
            const int hq = i >> hd_log2, d = i & (hd - 1);
            float o = 0.f, L = 0.f;
#pragma unroll
            for (int sl = 0; sl < NW; sl++) {
                const float* c = comb + ((size_t)sl * GQ + hq) * PS;
                o += c[d];
                L += c[hd];
            }
            if (nsp == 1) {
                odst[(size_t)(h0 + hq) * hd + d] = f2bf(o * (1.0f / L));
            } else {
                float Mh = M[0];
#pragma unroll
                for (int q2 = 1; q2 < GQ; q2++) Mh = (hq == q2) ? M[q2] : Mh;
                float* dst = a.part + ((size_t)(h0 + hq) * nsp + split) * PS;
                st_sc1(dst + d, o);
                if (d == 0) st_sc1(dst + hd, Mh), st_sc1(dst + hd + 1, L);
            }
        }
    } else if (nsp > 1) { /* empty slice: neutral partial, but it still arrives */
        for (int i = tid; i < GQ * hd; i += blockDim.x) {
            const int hq = i >> hd_log2, d = i & (hd - 1);
            float* dst = a.part + ((size_t)(h0 + hq) * nsp + split) * PS;
            st_sc1(dst + d, 0.f);
            if (d == 0) st_sc1(dst + hd, -__builtin_inff()), st_sc1(dst + hd + 1, 0.f);
        }
    }
    if (nsp == 1) return;

    // ---- arrival; the last workgroup of this kv-head merges
    int* const counter = a.counters + (int)blockIdx.y * a.cnt_stride; /* per (kv-head, part): its slices' workgroups meet here */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = (old == nsp - 1);
    }
    __syncthreads();
    if (!flag[0]) return;

    // Every thread fetches its output element from all slices in one round of loads; the head is the same for the whole wave
    // (64 | hd), so lane sp of the wave fetches (m, l) of slice sp and the scales are made in registers: no LDS, no barrier.
    constexpr int NT = NW * 64;
    constexpr int NV = (GQ * 128 + NT - 1) / NT; /* output elements per thread (hd <= 128) */
    float v[NV][KF_ATTN_MAX_SPLITS], ms[NV], ls[NV];
#pragma unroll
    for (int e = 0; e < NV; e++) {
        const int i = tid + e * NT;
        const bool in = i < GQ * hd;
        const int hq = in ? (i >> hd_log2) : 0, d = i & (hd - 1);
        const float* p = a.part + (size_t)(h0 + hq) * nsp * PS;
        const bool mine = in && lane < nsp;
        ms[e] = mine ? ld_sc1(p + (size_t)lane * PS + hd) : -__builtin_inff();
        ls[e] = mine ? ld_sc1(p + (size_t)lane * PS + hd + 1) : 0.f;
#pragma unroll
        for (int sp = 0; sp < KF_ATTN_MAX_SPLITS; sp++) v[e][sp] = (in && sp < nsp) ? ld_sc1(p + (size_t)sp * PS + d) : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NV; e++) {
        const int i = tid + e * NT;
        const float Mx = wave_max(ms[e]);
        const float sc = (ms[e] == -__builtin_inff()) ? 0.f : fast_exp(ms[e] - Mx);
        const float L = wave_sum(ls[e] * sc);
        float o = 0.f;
#pragma unroll
        for (int sp = 0; sp < KF_ATTN_MAX_SPLITS; sp++)
            o = fmaf(v[e][sp], __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sc), sp)), o);
        if (i < GQ * hd) odst[(size_t)(h0 + (i >> hd_log2)) * hd + (i & (hd - 1))] = f2bf(o * (1.0f / L));
    }
    if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// standalone ROPE::cuInfer: grid = n_head + n_kv, one wave each
__global__ void __launch_bounds__(64) qknorm_rope_kernel(uint16_t* q, uint16_t* k, const uint16_t* wq, const uint16_t* wk, const float* table, int pos_,
                                                         const int* d_pos, int n_head, int n_kv, int hd, float eps, long long q_stride, long long k_stride, int seq_len, float* rstd_q,
                                                         float* rstd_k) {
    __shared__ uint16_t buf[256];
    /* blockIdx.y = token of a batch; seq_len > 0: sequences of seq_len tokens back to back, positions restart at pos_ in each */
    const int pos = (d_pos ? *d_pos : pos_) + (seq_len > 0 ? (int)blockIdx.y % seq_len : (int)blockIdx.y);
    const int b = blockIdx.x;
    uint16_t* src = b < n_head ? q + (size_t)blockIdx.y * q_stride + (size_t)b * hd : k + (size_t)blockIdx.y * k_stride + (size_t)(b - n_head) * hd;
    const uint16_t* wn = b < n_head ? wq : wk;
    float* ro = b < n_head ? (rstd_q ? rstd_q + (size_t)blockIdx.y * n_head + b : nullptr) : (rstd_k ? rstd_k + (size_t)blockIdx.y * n_kv + (b - n_head) : nullptr);
    prep_head(load_head(src, wn, hd), wn != nullptr, table ? table + (size_t)pos * hd : nullptr, hd, eps, buf, ro);
    __syncthreads();
    for (int i = threadIdx.x; i < hd; i += 64) src[i] = buf[i];
}

// slices: ~64 keys each, enough workgroups to cover the chip, bounded by the scratch layout
int attn_splits(int pos_bound, int n_kv) {
    // keys per slice: one workgroup streams a slice in batches of 64 keys (hd 128, 4 waves); short contexts stay in ONE slice per
    // kv-head (no cross-workgroup hand-off at all), long ones are cut so that the chip is covered.
    constexpr int slice = 64, single = 192; /* keys per slice / longest context served by one slice (settled by the round-1 sweeps) */
    const int len = pos_bound + 1;
    if (len <= single) return 1;
    int nsp = (len + slice - 1) / slice;
    int cap = 512 / n_kv;
    if (cap < 1) cap = 1;
    if (nsp > cap) nsp = cap;
    if (nsp > KF_ATTN_MAX_SPLITS) nsp = KF_ATTN_MAX_SPLITS;
    if (nsp < 1) nsp = 1;
    return nsp;
}

int attn_launch(hipStream_t st, AttnArgs& a) {
    const int hd = a.hd;
    if (hd < 64 || hd > 128 || (hd & (hd - 1)) != 0) return KF_INVALID_ARGS; /* 8 dims per lane, one RoPE trip per wave */
    if (a.n_kv <= 0 || a.n_head % a.n_kv != 0) return KF_INVALID_ARGS;
    const int GQ = a.n_head / a.n_kv;
    const bool batch = a.one_slice != 0;
    if (a.n_tok < 1) a.n_tok = 1;
    if (!batch) a.n_tok = 1, a.q_stride = 0;
    const int pos_max = a.pos + a.n_tok - 1;
    const int nsp = batch ? 1 : attn_splits(a.pos, a.n_kv);
    a.n_splits = nsp;
    a.chunk = (pos_max + 1 + nsp - 1) / nsp;
    a.inv_sqrt_hd_den = sqrtf((float)hd);
    a.gq_split = (a.canon && GQ == 8) ? (g_knobs.attn_gq_split == 8 ? 8 : (g_knobs.attn_gq_split == 4 ? 4 : 2)) : ((a.canon && GQ == 4 && g_knobs.attn_gq_split >= 4) ? 2 : 1);
    a.cnt_stride = KF_ATTN_CNT_BYTES / 4 / (a.n_kv * a.gq_split); /* arrival counters of different kv-heads in different cache lines: atomics on one line serialise */
    if (a.cnt_stride > 64) a.cnt_stride = 64;
    if (a.cnt_stride < 1) return KF_INVALID_ARGS;
    // one 64-key batch per 4-wave workgroup (one wave per SIMD: the kernel is bound by VALU issue inside a latency chain, so
    // spreading the keys over more CUs beats more waves per CU); 8 waves once the slices have to grow past 128 keys
    int NW = (GQ <= 2 && a.chunk > 128) ? 8 : 4;
    const int gq_wg = GQ / a.gq_split; /* query heads per workgroup */
    size_t smem = sizeof(double) * ((size_t)NW * gq_wg * (hd + 2)) + sizeof(uint16_t) * ((size_t)gq_wg * hd + hd) + 16;
    const size_t smem_fast = sizeof(uint16_t) * ((size_t)GQ * hd + hd) + sizeof(float) * (NW * GQ + 4 + (size_t)NW * GQ * (hd + 4));
    if (!a.canon) smem = smem_fast;
    dim3 grid(nsp, a.n_kv * a.gq_split, a.n_tok);
#define KF_ATTN_GO(gq, nw)                                                                                       \
    do {                                                                                                        \
        if (a.canon) {                                                                                          \
            if (hd == 128) hipLaunchKernelGGL((attn_kernel<gq, nw, 128>), grid, dim3(nw * 64), smem, st, a);      \
            else hipLaunchKernelGGL((attn_kernel<gq, nw, 64>), grid, dim3(nw * 64), smem, st, a);                 \
        } else {                                                                                                \
            if (hd == 128) hipLaunchKernelGGL((attn_fast_kernel<gq, nw, 128>), grid, dim3(nw * 64), smem, st, a); \
            else hipLaunchKernelGGL((attn_fast_kernel<gq, nw, 64>), grid, dim3(nw * 64), smem, st, a);            \
        }                                                                                                       \
    } while (0)
    switch (GQ) {
        case 1: if (NW == 8) KF_ATTN_GO(1, 8); else KF_ATTN_GO(1, 4); break;
        case 2: if (NW == 8) KF_ATTN_GO(2, 8); else KF_ATTN_GO(2, 4); break;
        case 4:
            if (a.gq_split == 2) KF_ATTN_GO(2, 4); /* canonical order: two workgroups of two heads each, as for GQA-8 */
            else KF_ATTN_GO(4, 4);
            break;
        case 8:
            if (a.gq_split == 8) KF_ATTN_GO(1, 4); /* one head per workgroup */
            else if (a.gq_split == 4) KF_ATTN_GO(2, 4); /* four workgroups of two heads each */
            else if (a.gq_split == 2) KF_ATTN_GO(4, 4); /* two workgroups of four heads each (canonical order) */
            else KF_ATTN_GO(8, 4);
            break;
        default: return KF_INVALID_ARGS;
    }
#undef KF_ATTN_GO
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

int qknorm_rope_launch(hipStream_t st, uint16_t* q, uint16_t* k, const uint16_t* wq, const uint16_t* wk, const float* table, int pos,
                       const int* d_pos, int n_head, int n_kv, int hd, float eps, int n_tok, long long q_stride, long long k_stride, int seq_len, float* rstd_q, float* rstd_k) {
    if (hd < 64 || hd > 128 || (hd & (hd - 1)) != 0 || n_tok < 1) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(qknorm_rope_kernel, dim3(n_head + (k ? n_kv : 0), n_tok), dim3(64), 0, st, q, k, wq, wk, table, pos, d_pos, n_head, n_kv, hd, eps,
                       q_stride, k_stride, seq_len, rstd_q, rstd_k);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
