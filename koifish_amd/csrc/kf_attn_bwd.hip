// kf_attn_bwd.hip -- causal multi-head attention backward for a token batch (training forward/backward of BASELINE config 3).
//
// The reference calls cuDNN's fused SDPA backward through cudnn-frontend (src/Device/CUDA/QKV.cu:130-315, 427-447): closed-source arithmetic, so
// what is restated is the mathematics of softmax attention (scale = 1/sqrt(hd), causal mask, fp32 softmax, bf16 tensors):
//   S = scale Q K^T,  P = softmax_causal(S),  O = P V
//   D_i = sum_d dO_i[d] O_i[d];  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - D);  dQ = scale dS K;  dK = scale dS^T Q.
// This file: the first version, VALU arithmetic on LDS tiles (KF_ATTN_BWD=valu; the default is the MFMA flash form of kf_attn_bwd_mfma.hip,
// which attn_backward_launch below tries first), two launches per sequence:
//   attn_bwd_dq_kernel : one workgroup per (64-query tile, head); thread = (query row, quarter of the key tile).  Pass A over the key tiles gives the
//                        row's log-sum-exp L, pass B recomputes P and accumulates dQ; L and D go to scratch for the second launch.
//   attn_bwd_dkv_kernel: one workgroup per (64-key tile, head); thread = (key row, half of head_dim, half of the query tile); walks the query tiles
//                        from the diagonal down, recomputes P from L, accumulates dK and dV.
// Scores are recomputed in fp32 (no bf16 rounding of S), exp through v_exp_f32; outputs are single bf16 stores.  head_dim 64.
#include <stdlib.h>

#include "kf_kernels.h"

namespace kf {

__device__ __forceinline__ float ab_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f); }

template <int HD>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, long long ld_qkv,
                                                          const uint16_t* __restrict__ o, const uint16_t* __restrict__ dO, long long ld_o, uint16_t* __restrict__ dq,
                                                          long long ld_d, float* __restrict__ Lbuf, float* __restrict__ Dbuf, int T, float scale) {
    static_assert(HD == 64, "head_dim 64");
    __shared__ __attribute__((aligned(16))) uint16_t Ks[64][HD], Vs[64][HD];
    const int tid = threadIdx.x, r = tid >> 2, p = tid & 3, qt = blockIdx.x, h = blockIdx.y;
    const int qi = qt * 64 + r;
    const bool row_ok = qi < T;
    const size_t hoff = (size_t)h * HD;
    float qf[HD], dof[HD], dqa[HD];
    {
        const int qr = row_ok ? qi : T - 1;
        const uint16_t* qp = q + (size_t)qr * ld_qkv + hoff;
        const uint16_t* dp_ = dO + (size_t)qr * ld_o + hoff;
#pragma unroll
        for (int d = 0; d < HD; d += 8) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(qp + d), b = *reinterpret_cast<const u32x4*>(dp_ + d);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 4; e++) qf[d + 2 * e] = bf_lo(aw[e]), qf[d + 2 * e + 1] = bf_hi(aw[e]), dof[d + 2 * e] = bf_lo(bw[e]), dof[d + 2 * e + 1] = bf_hi(bw[e]);
        }
    }
    float Dr = 0.f;
    {
        const uint16_t* op = o + (size_t)(row_ok ? qi : T - 1) * ld_o + hoff;
#pragma unroll
        for (int d = 0; d < HD; d += 8) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(op + d);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; e++) Dr = fmaf(dof[d + 2 * e], bf_lo(aw[e]), Dr), Dr = fmaf(dof[d + 2 * e + 1], bf_hi(aw[e]), Dr);
        }
    }
    auto load_tile = [&](int kt, bool with_v) {
        // 64 rows x 128 bytes = 512 16-byte vectors per tensor: two per thread
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int vi = tid + 256 * i, row = vi >> 3, col = (vi & 7) * 8;
            int kr = kt * 64 + row;
            kr = kr < T ? kr : T - 1;
            *reinterpret_cast<u32x4*>(&Ks[row][col]) = *reinterpret_cast<const u32x4*>(k + (size_t)kr * ld_qkv + hoff + col);
            if (with_v) *reinterpret_cast<u32x4*>(&Vs[row][col]) = *reinterpret_cast<const u32x4*>(v + (size_t)kr * ld_qkv + hoff + col);
        }
    };
    auto dot_row = [&](const float* a, const uint16_t* row) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 8) {
            const u32x4 x = *reinterpret_cast<const u32x4*>(row + d);
            const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; e++) s = fmaf(a[d + 2 * e], bf_lo(xw[e]), s), s = fmaf(a[d + 2 * e + 1], bf_hi(xw[e]), s);
        }
        return s;
    };
    // ---- pass A: log-sum-exp of the row
    float m = -__builtin_inff(), l = 0.f;
    for (int kt = 0; kt <= qt; kt++) {
        __syncthreads();
        load_tile(kt, false);
        __syncthreads();
        for (int j = p * 16; j < p * 16 + 16; j++) {
            const int kj = kt * 64 + j;
            if (kj > qi || kj >= T) break; /* keys ascend inside the quarter */
            const float s = dot_row(qf, Ks[j]) * scale;
            if (s > m) l *= ab_exp(m - s), m = s;
            l += ab_exp(s - m);
        }
    }
#pragma unroll
    for (int x = 1; x <= 2; x <<= 1) { /* the four quarters of the row sit in one lane quad */
        const float mo = __shfl_xor(m, x, 64), lo = __shfl_xor(l, x, 64);
        const float mn = fmaxf(m, mo);
        l = (m == -__builtin_inff() ? 0.f : l * ab_exp(m - mn)) + (mo == -__builtin_inff() ? 0.f : lo * ab_exp(mo - mn));
        m = mn;
    }
    const float L = m + __logf(l);
    if (row_ok && p == 0) Lbuf[(size_t)h * T + qi] = L, Dbuf[(size_t)h * T + qi] = Dr;
    // ---- pass B: dQ
#pragma unroll
    for (int d = 0; d < HD; d++) dqa[d] = 0.f;
    for (int kt = 0; kt <= qt; kt++) {
        __syncthreads();
        load_tile(kt, true);
        __syncthreads();
        for (int j = p * 16; j < p * 16 + 16; j++) {
            const int kj = kt * 64 + j;
            if (kj > qi || kj >= T) break;
            const float s = dot_row(qf, Ks[j]) * scale;
            const float pr = ab_exp(s - L);
            const float dp = dot_row(dof, Vs[j]);
            const float ds = pr * (dp - Dr) * scale;
#pragma unroll
            for (int d = 0; d < HD; d += 8) {
                const u32x4 x = *reinterpret_cast<const u32x4*>(&Ks[j][d]);
                const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int e = 0; e < 4; e++) dqa[d + 2 * e] = fmaf(ds, bf_lo(xw[e]), dqa[d + 2 * e]), dqa[d + 2 * e + 1] = fmaf(ds, bf_hi(xw[e]), dqa[d + 2 * e + 1]);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < HD; d++) {
        float t = dqa[d];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        dqa[d] = t;
    }
    if (row_ok) { /* quarter p stores dims 16 p .. 16 p + 15 */
        uint16_t* dst = dq + (size_t)qi * ld_d + hoff;
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float lo = 0.f, hi = 0.f;
#pragma unroll
            for (int pp = 0; pp < 4; pp++) /* static register indices: select, do not index */
                if (p == pp) lo = dqa[16 * pp + 2 * e], hi = dqa[16 * pp + 2 * e + 1];
            w[e] = pack_bf16x2(lo, hi);
        }
        *reinterpret_cast<u32x4*>(dst + 16 * p) = u32x4{w[0], w[1], w[2], w[3]};
        *reinterpret_cast<u32x4*>(dst + 16 * p + 8) = u32x4{w[4], w[5], w[6], w[7]};
    }
}

template <int HD>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, long long ld_qkv,
                                                           const uint16_t* __restrict__ dO, long long ld_o, uint16_t* __restrict__ dk, uint16_t* __restrict__ dv, long long ld_d,
                                                           const float* __restrict__ Lbuf, const float* __restrict__ Dbuf, int T, float scale) {
    static_assert(HD == 64, "head_dim 64");
    constexpr int HH = HD / 2;
    __shared__ __attribute__((aligned(16))) uint16_t Qs[64][HD], Os[64][HD];
    __shared__ float Ls[64], Ds[64];
    const int tid = threadIdx.x, c = tid >> 2, dh = (tid >> 1) & 1, qh = tid & 1, kt = blockIdx.x, h = blockIdx.y;
    const int kc = kt * 64 + c;
    const bool key_ok = kc < T;
    const size_t hoff = (size_t)h * HD + (size_t)dh * HH;
    float kf_[HH], vf[HH], dka[HH], dva[HH];
    {
        const int kr = key_ok ? kc : T - 1;
#pragma unroll
        for (int d = 0; d < HH; d += 8) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(k + (size_t)kr * ld_qkv + hoff + d), b = *reinterpret_cast<const u32x4*>(v + (size_t)kr * ld_qkv + hoff + d);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 4; e++) kf_[d + 2 * e] = bf_lo(aw[e]), kf_[d + 2 * e + 1] = bf_hi(aw[e]), vf[d + 2 * e] = bf_lo(bw[e]), vf[d + 2 * e + 1] = bf_hi(bw[e]);
        }
#pragma unroll
        for (int d = 0; d < HH; d++) dka[d] = 0.f, dva[d] = 0.f;
    }
    const int nqt = (T + 63) / 64;
    for (int qt = kt; qt < nqt; qt++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int vi = tid + 256 * i, row = vi >> 3, col = (vi & 7) * 8;
            int qr = qt * 64 + row;
            qr = qr < T ? qr : T - 1;
            *reinterpret_cast<u32x4*>(&Qs[row][col]) = *reinterpret_cast<const u32x4*>(q + (size_t)qr * ld_qkv + (size_t)h * HD + col);
            *reinterpret_cast<u32x4*>(&Os[row][col]) = *reinterpret_cast<const u32x4*>(dO + (size_t)qr * ld_o + (size_t)h * HD + col);
        }
        if (tid < 64) {
            const int qr = qt * 64 + tid;
            Ls[tid] = qr < T ? Lbuf[(size_t)h * T + qr] : 0.f, Ds[tid] = qr < T ? Dbuf[(size_t)h * T + qr] : 0.f;
        }
        __syncthreads();
        for (int ii = qh * 32; ii < qh * 32 + 32; ii++) {
            const int qi = qt * 64 + ii;
            float sp = 0.f, dpp = 0.f;
            float qv[HH], ov[HH];
#pragma unroll
            for (int d = 0; d < HH; d += 8) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(&Qs[ii][dh * HH + d]), b = *reinterpret_cast<const u32x4*>(&Os[ii][dh * HH + d]);
                const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    qv[d + 2 * e] = bf_lo(aw[e]), qv[d + 2 * e + 1] = bf_hi(aw[e]), ov[d + 2 * e] = bf_lo(bw[e]), ov[d + 2 * e + 1] = bf_hi(bw[e]);
                }
            }
#pragma unroll
            for (int d = 0; d < HH; d++) sp = fmaf(qv[d], kf_[d], sp), dpp = fmaf(ov[d], vf[d], dpp);
            const float s = (sp + __shfl_xor(sp, 2, 64)) * scale; /* the other half of head_dim: lane ^ 2 */
            const float dp = dpp + __shfl_xor(dpp, 2, 64);
            const bool valid = key_ok && qi < T && qi >= kc;
            const float pr = valid ? ab_exp(s - Ls[ii]) : 0.f;
            const float ds = pr * (dp - Ds[ii]) * scale;
#pragma unroll
            for (int d = 0; d < HH; d++) dva[d] = fmaf(pr, ov[d], dva[d]), dka[d] = fmaf(ds, qv[d], dka[d]);
        }
    }
    // the two query halves of a key row: lane ^ 1
    uint32_t wk[HH / 2], wv[HH / 2];
#pragma unroll
    for (int d = 0; d < HH; d += 2) {
        const float k0 = dka[d] + __shfl_xor(dka[d], 1, 64), k1 = dka[d + 1] + __shfl_xor(dka[d + 1], 1, 64);
        const float v0 = dva[d] + __shfl_xor(dva[d], 1, 64), v1 = dva[d + 1] + __shfl_xor(dva[d + 1], 1, 64);
        wk[d / 2] = pack_bf16x2(k0, k1), wv[d / 2] = pack_bf16x2(v0, v1);
    }
    if (key_ok && qh == 0) {
        uint16_t* dkp = dk + (size_t)kc * ld_d + hoff;
        uint16_t* dvp = dv + (size_t)kc * ld_d + hoff;
#pragma unroll
        for (int d = 0; d < HH / 2; d += 4) {
            *reinterpret_cast<u32x4*>(dkp + 2 * d) = u32x4{wk[d], wk[d + 1], wk[d + 2], wk[d + 3]};
            *reinterpret_cast<u32x4*>(dvp + 2 * d) = u32x4{wv[d], wv[d + 1], wv[d + 2], wv[d + 3]};
        }
    }
}

int attn_backward_launch(hipStream_t st, const uint16_t* q, const uint16_t* k, const uint16_t* v, long long ld_qkv, const uint16_t* o, const uint16_t* dO, long long ld_o,
                         uint16_t* dq, uint16_t* dk, uint16_t* dv, long long ld_d, int T, int n_head, int n_kv, int hd, int n_seq, float* scratch) {
    if (T < 1 || n_head < 1 || n_seq < 1 || n_kv < 1 || n_head % n_kv != 0) return KF_INVALID_ARGS;
    static int form = -1; /* KF_ATTN_BWD=valu keeps this file's first version (head_dim 64 only); default: the MFMA form of kf_attn_bwd_mfma.hip */
    if (form < 0) {
        const char* e = getenv("KF_ATTN_BWD");
        form = (e && e[0] == 'v') ? 0 : 1;
    }
    if (form == 1) {
        const int rc = attn_backward_mfma_launch(st, q, k, v, ld_qkv, o, dO, ld_o, dq, dk, dv, ld_d, T, n_head, hd, n_seq, scratch, n_kv, ld_qkv, ld_d);
        if (rc != 1) return rc;
    }
    if (hd != 64 || n_kv != n_head) return KF_UNSUPPORTED_DATATYPE; /* the first version: head_dim 64, no GQA */
    const float scale = 1.0f / sqrtf((float)hd);
    const dim3 grid((T + 63) / 64, n_head);
    for (int b = 0; b < n_seq; b++) { /* one sequence per launch pair in this form */
        const size_t ro = (size_t)b * T;
        float* Lb = scratch + (size_t)b * 2 * n_head * T;
        float* Db = Lb + (size_t)n_head * T;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), grid, dim3(256), 0, st, q + ro * ld_qkv, k + ro * ld_qkv, v + ro * ld_qkv, ld_qkv, o + ro * ld_o, dO + ro * ld_o, ld_o,
                           dq + ro * ld_d, ld_d, Lb, Db, T, scale);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<64>), grid, dim3(256), 0, st, q + ro * ld_qkv, k + ro * ld_qkv, v + ro * ld_qkv, ld_qkv, dO + ro * ld_o, ld_o, dk + ro * ld_d,
                           dv + ro * ld_d, ld_d, Lb, Db, T, scale);
    }
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
