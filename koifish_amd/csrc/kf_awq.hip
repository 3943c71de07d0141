// kf_awq.hip -- vendor AutoAWQ GEMM-format 4-bit weights (Qwen3-*-AWQ checkpoints), gfx950.
//
// Reference: CU_Q42X_awq + CU_I2Q4_unpack (src/Device/CUDA/kernel/quantizer.cu:131-156, kernel/packedN.cuh:109-116) dequantise
// the whole tensor to a bf16 [in, out] matrix (TransA = 0, GeQuant.cpp:989-992) and cuBLASLt multiplies it.  Here the mat-vec
// reads qweight int32 [in, out/8] once: a lane owns 8 output columns (one word per input row, nibble k at bits 4*{0,4,1,5,2,6,3,7}[k]),
// waves split the input rows in chunks of 32, W^T[i,o] = bf16((q - z) * float(scale_fp16)) exactly as the reference forms it,
// fp32 accumulation, slices combined in a fixed order by a second tiny launch (deterministic, no atomics).
#include "kf_kernels.h"

namespace kf {

constexpr int AWQ_ROWS = 32; /* input rows per wave step (a quarter of the 128-row quantisation group) */

__device__ __forceinline__ void awq_unpack8(uint32_t w, float* q) {
    // element k <- nibble ORDER[k], ORDER = {0,4,1,5,2,6,3,7}: low nibbles of bytes 0..3 are elements 0,2,4,6... spelled out:
    q[0] = (float)(w & 0xFu), q[1] = (float)((w >> 16) & 0xFu), q[2] = (float)((w >> 4) & 0xFu), q[3] = (float)((w >> 20) & 0xFu);
    q[4] = (float)((w >> 8) & 0xFu), q[5] = (float)((w >> 24) & 0xFu), q[6] = (float)((w >> 12) & 0xFu), q[7] = (float)((w >> 28) & 0xFu);
}

// grid (ceil(out/512), nslice), block 256.  partial [nslice][out] fp32.
__global__ void __launch_bounds__(256) awq_gemv_kernel(const uint32_t* __restrict__ qweight, const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales,
                                                       const uint16_t* __restrict__ x, float* __restrict__ partial, int n_in, int n_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);      // [n_in]
    float* red = xs + n_in;                               // [4][64][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n_in; i += 256) xs[i] = bf2f(x[i]);
    __syncthreads();
    const int w8 = n_out >> 3;
    const int wcol = blockIdx.x * 64 + lane; /* word column = 8 outputs */
    const bool act = wcol < w8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = 0.f;
    const int nunit = n_in / AWQ_ROWS;
    for (int u = blockIdx.y * 4 + wave; u < nunit; u += gridDim.y * 4) {
        const int i0 = u * AWQ_ROWS, g = i0 >> 7;
        float zf[8], sf[8];
        if (act) {
            awq_unpack8(qzeros[(size_t)g * w8 + wcol], zf);
            const u32x4 sv = *reinterpret_cast<const u32x4*>(scales + (size_t)g * n_out + (size_t)wcol * 8);
            const uint32_t sw[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
            for (int k = 0; k < 4; k++) sf[2 * k] = half_bits_to_f32(sw[k] & 0xffffu), sf[2 * k + 1] = half_bits_to_f32(sw[k] >> 16);
            uint32_t wv[AWQ_ROWS];
#pragma unroll
            for (int r = 0; r < AWQ_ROWS; r++) wv[r] = __builtin_nontemporal_load(qweight + (size_t)(i0 + r) * w8 + wcol);
#pragma unroll
            for (int r = 0; r < AWQ_ROWS; r++) {
                float q[8];
                awq_unpack8(wv[r], q);
                const float xi = xs[i0 + r];
#pragma unroll
                for (int k = 0; k < 8; k++) acc[k] = fmaf(round_bf16((q[k] - zf[k]) * sf[k]), xi, acc[k]);
            }
        }
    }
    // fixed-order sum of the 4 waves, then one partial row per workgroup slice
#pragma unroll
    for (int k = 0; k < 8; k++) red[(wave * 64 + lane) * 8 + k] = acc[k];
    __syncthreads();
    if (wave == 0 && act) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float v = ((red[(0 * 64 + lane) * 8 + k] + red[(1 * 64 + lane) * 8 + k]) + red[(2 * 64 + lane) * 8 + k]) + red[(3 * 64 + lane) * 8 + k];
            partial[(size_t)blockIdx.y * n_out + (size_t)wcol * 8 + k] = v;
        }
    }
}

__global__ void awq_finish_kernel(const float* __restrict__ partial, int nslice, int n_out, uint16_t* __restrict__ y, const uint16_t* __restrict__ bias, float alpha,
                                  float beta, const uint16_t* __restrict__ residual) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    float v = partial[o];
    for (int s = 1; s < nslice; s++) v = v + partial[(size_t)s * n_out + o];
    if (alpha != 1.0f) v = alpha * v;
    if (beta != 0.0f) v = v + beta * bf2f(y[o]);
    if (bias) v = v + bf2f(bias[o]);
    uint16_t r = f2bf(v);
    if (residual) r = f2bf(bf2f(residual[o]) + bf2f(r));
    y[o] = r;
}

// mat0 [in, out] bf16, what GetDataX leaves in tmpTernary for an AWQ tensor
__global__ void awq_dequant_kernel(const uint32_t* __restrict__ qweight, const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales, int n_in, int n_out,
                                   uint16_t* __restrict__ out) {
    const int w8 = n_out >> 3;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_in * w8) return;
    const int i = (int)(idx / w8), wcol = (int)(idx - (size_t)i * w8), g = i >> 7;
    float q[8], z[8];
    awq_unpack8(qweight[idx], q);
    awq_unpack8(qzeros[(size_t)g * w8 + wcol], z);
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float s0 = half_bits_to_f32(scales[(size_t)g * n_out + (size_t)wcol * 8 + 2 * k]), s1 = half_bits_to_f32(scales[(size_t)g * n_out + (size_t)wcol * 8 + 2 * k + 1]);
        o[k] = pack_bf16x2((q[2 * k] - z[2 * k]) * s0, (q[2 * k + 1] - z[2 * k + 1]) * s1);
    }
    *reinterpret_cast<u32x4*>(out + (size_t)i * n_out + (size_t)wcol * 8) = u32x4{o[0], o[1], o[2], o[3]};
}

static int awq_check(const kf_weight* w) {
    if (!w->qzeros || !w->qscales || w->type != KF_Q4) return KF_QUANT_ERR;
    if (w->ne0 % 8 || w->ne1 % 128 || w->lGroup != 128) return KF_QUANT_ERR;
    return KF_OK;
}

size_t awq_scratch_bytes(const kf_weight* w) { return sizeof(float) * (size_t)(w->ne1 / AWQ_ROWS / 4 + 1) * w->ne0; }

int awq_linear_launch(hipStream_t st, const kf_weight* w, const uint16_t* x, uint16_t* y, const uint16_t* bias, float alpha, float beta, const uint16_t* residual,
                      float* scratch) {
    int r = awq_check(w);
    if (r) return r;
    const int n_out = w->ne0, n_in = w->ne1, w8 = n_out / 8;
    const int nunit = n_in / AWQ_ROWS;
    int nslice = (nunit + 3) / 4;
    if (nslice < 1) nslice = 1;
    const size_t smem = sizeof(float) * ((size_t)n_in + 4 * 64 * 8);
    if (smem > 160 * 1024) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(awq_gemv_kernel, dim3((w8 + 63) / 64, nslice), dim3(256), smem, st, (const uint32_t*)w->data, (const uint32_t*)w->qzeros,
                       (const uint16_t*)w->qscales, x, scratch, n_in, n_out);
    hipLaunchKernelGGL(awq_finish_kernel, dim3((n_out + 255) / 256), dim3(256), 0, st, scratch, nslice, n_out, y, bias, alpha, beta, residual);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

int awq_dequant_launch(hipStream_t st, const kf_weight* w, uint16_t* out) {
    int r = awq_check(w);
    if (r) return r;
    const size_t n = (size_t)w->ne1 * (w->ne0 / 8);
    hipLaunchKernelGGL(awq_dequant_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)w->data, (const uint32_t*)w->qzeros,
                       (const uint16_t*)w->qscales, w->ne1, w->ne0, out);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
