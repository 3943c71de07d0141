// kf_loss.hip -- fused classifier: softmax cross-entropy loss of every token row + the logit gradient written over the logits.
//
// Replaces fused_classifier (src/Device/CUDA/kernel/fused_classifier.cuh:68-140 with prepare_softmax_blockwide3 :21-62), launched by
// Head4Token as fused_classifier<<<dB*T, 1024>>>(logits, losses, nullptr, rLoss, targets, dB, T, V, Vp, devMask, write_dlogits)
// (NeuronFuse.cu:923).  One workgroup of 1024 threads per row, the same decomposition as the reference so that the fp32 results follow the
// same order of operations:
//   * thread t visits the 8-element vectors i = ceil(V/8) + t - 1024, i - 1024, ... >= 0 (highest first; the ragged last vector is
//     bounds-checked) keeping a running (max, sum): on a new maximum  sum *= exp(old - new), then  sum += exp(v - max)
//     (the reference multiplies by exp(0) = 1 when the maximum does not move; skipping that product is bit-identical);
//   * block maximum; every thread rescales  sum *= exp(max_t - max_block); block sum.  Both block reductions are the reference's
//     blockReduce_v0 (utils.cuh:235-270): an xor-butterfly (offsets 16, 8, 4, 2, 1) inside each 32-thread group, the 32 group results
//     through shared memory, the same butterfly again -- restated here on half-waves;
//   * losses[row] -= log(exp(logit[target] - max) * (1 / sum))        (accumulates: the caller zeroes the buffer);
//   * every element: prob = exp(logit - max) * (1 / sum); dlogit = bf16((prob - [element == target]) * dloss) over the logit,
//     probs (optional) = bf16(prob).
// exp / log are the fixed recipes kf_expf / kf_logf (the reference calls CUDA expf / logf, 2-ulp routines of their own), so the HIP kernel and
// oracle/kf_oracle.c kfo_fused_classifier agree bit for bit.  Rows whose mask word has F_IGNORE_LOSS (0x10000, DataLoader.hpp:78) are left
// untouched.  The reference's tail loop for V % 8 != 0 advances by one element per thread (`i++`, :124), so with V % 8 >= 2 its threads
// rewrite each other's elements while reading them; here every tail element is handled once (identical for V % 8 <= 1, e.g. GPT-2's 50257).
// HBM-bound: V*2 B read + V*2 B written per row (the second read of the row hits L2).
#include "kf_kernels.h"

namespace kf {

template <bool IS_MAX>
__device__ __forceinline__ float half_wave_butterfly(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = IS_MAX ? fmaxf(v, o) : v + o;
    }
    return v;
}
// blockReduce_v0 for 1024 threads = 32 groups of 32
template <bool IS_MAX>
__device__ __forceinline__ float block_reduce_1024(float v, float* sh) {
    const int grp = threadIdx.x >> 5, l32 = threadIdx.x & 31;
    __syncthreads(); /* sh may still be read by the previous reduction */
    v = half_wave_butterfly<IS_MAX>(v);
    if (l32 == 0) sh[grp] = v;
    __syncthreads();
    return half_wave_butterfly<IS_MAX>(sh[l32]);
}

__global__ void __launch_bounds__(1024) fused_classifier_kernel(uint16_t* __restrict__ logits, float* __restrict__ losses, uint16_t* __restrict__ probs,
                                                                float dloss, const int* __restrict__ targets, int V, int P, const int* __restrict__ mask,
                                                                int write_dlogits) {
    __shared__ float sh[32];
    const long idx = (long)gridDim.x - ((long)blockIdx.x + 1); /* reverse order: the last rows of the LM-head GEMM are the warmest in L2 */
    if (mask && (mask[idx] & 0x10000)) return;
    const int ix = targets[idx];
    uint16_t* const row = logits + idx * (long)P;
    const int tid = threadIdx.x;

    float tmax = -__builtin_inff(), tsum = 0.0f;
    auto visit = [&](float v) {
        if (v > tmax) {
            if (tmax != -__builtin_inff()) tsum *= kf_expf(tmax - v);
            tmax = v;
        }
        tsum += kf_expf(v - tmax);
    };
    int i = (V + 7) / 8 + tid - 1024;
    while (i >= 0 && (i + 1) * 8 > V) { /* the ragged last vector */
        for (int k = 0; k < 8 && i * 8 + k < V; k++) visit(bf2f(row[i * 8 + k]));
        i -= 1024;
    }
    for (; i >= 0; i -= 1024) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(row + (size_t)i * 8);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) visit(bf_lo(w[k])), visit(bf_hi(w[k]));
    }
    const float bmax = block_reduce_1024<true>(tmax, sh);
    tsum *= kf_expf(tmax - bmax); /* a thread without elements: 0 * exp(-inf) = 0 */
    const float bsum = block_reduce_1024<false>(tsum, sh);
    const float scale = 1.0f / bsum;

    if (tid == 0) {
        const float prob = kf_expf(bf2f(row[ix]) - bmax) * scale;
        losses[idx] -= kf_logf(prob);
    }
    __syncthreads(); /* the target logit is read before anyone overwrites the row */

    uint16_t* const prow = probs ? probs + idx * (long)P : nullptr;
    const int nvec = V / 8;
    for (int v8 = tid; v8 < nvec; v8 += 1024) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(row + (size_t)v8 * 8);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        uint32_t g[4], pr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float p0 = kf_expf(bf_lo(w[k]) - bmax) * scale, p1 = kf_expf(bf_hi(w[k]) - bmax) * scale;
            const int e0 = v8 * 8 + 2 * k;
            g[k] = pack_bf16x2((p0 - (e0 == ix ? 1.0f : 0.0f)) * dloss, (p1 - (e0 + 1 == ix ? 1.0f : 0.0f)) * dloss);
            pr[k] = pack_bf16x2(p0, p1);
        }
        if (write_dlogits) *reinterpret_cast<u32x4*>(row + (size_t)v8 * 8) = u32x4{g[0], g[1], g[2], g[3]};
        if (prow) *reinterpret_cast<u32x4*>(prow + (size_t)v8 * 8) = u32x4{pr[0], pr[1], pr[2], pr[3]};
    }
    for (int e = nvec * 8 + tid; e < V; e += 1024) {
        const float p = kf_expf(bf2f(row[e]) - bmax) * scale;
        if (write_dlogits) row[e] = f2bf((p - (e == ix ? 1.0f : 0.0f)) * dloss);
        if (prow) prow[e] = f2bf(p);
    }
}

int fused_classifier_launch(hipStream_t st, uint16_t* logits, float* losses, uint16_t* probs, float dloss, const int* targets, long rows, int V, int P,
                            const int* mask, int write_dlogits) {
    if (rows < 1 || rows > 0x7fffffffL || V < 1 || P < V || (P % 8) != 0) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(fused_classifier_kernel, dim3((unsigned)rows), dim3(1024), 0, st, logits, losses, probs, dloss, targets, V, P, mask, write_dlogits);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
