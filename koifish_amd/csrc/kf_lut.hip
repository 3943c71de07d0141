// kf_lut.hip -- the row-codebook 4-bit storage (KF_QUANT_ROW_LUT): quantiser, dequant, embedding rows.  gfx950 / wave64.
//
// Reference: GeQuant::RT_NormalF / _row_lut (GeQuant.cpp:696-755) with Distri_PIPE::Prepare / X2NormalF (GeQuant.cpp:641-694) build it on the CPU
// at load time; CU_Q42X_NF4 / CU_Q42X_lut (quantizer.cu:583-652) unpack it inside GTensor::GetDataX; CU_embed_forw_q4 / _nf4 (embed.cuh:54-121)
// read one row of it for TokenEmbed::cuInfer (NeuronFuse.cu:176-207).  The mat-vec over this storage is gemv_kernel<FMT_Q4R> (kf_gemv.hip).
//
// Layout: data = ne0 * ne1 / 2 bytes, element i of the row-major matrix in byte i / 2, even i in the high nibble (BIT_SET_k, CLI_params.cpp:2177-2190);
// gama (bf16) = [R_SCALE ne0][C_SCALE ne1][LUT ne0 x 16].  weight(r, c) = lut[r][nibble(r, c)] (row / column scales are 1: LowBit_worker sweeps
// NORMAL_MODE::NO_NORMAL only, GeQuant.cpp:844, so rc_normal = 0 and sR = 1).  All of it is HBM-bound byte work: 0.5 B read + 2 B written per weight
// for the dequant, 2.5 B read + 0.5 B written for the quantiser.
#include "kf_kernels.h"

namespace kf {

// NF4_LUT::table (g_float.hpp:543-558)
__constant__ float kNF4[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f, -0.28444138169288635f, -0.18477343022823334f,
                               -0.09105003625154495f, 0.0f, 0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
                               0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

// ---------------------------------------------------------------- quantiser: one wave per row
// Distri_PIPE::Next over the row (vmin, vmax of the fp32 values of the bf16 weights), Prepare(16) in its default symmetric mode:
//   abs_max = max(|vmin|, |vmax|); scale = abs_max > 0 ? float(1.0f / (double)abs_max) : 1;  codebook[i] = table[i] / scale  (fp32 division)
// the stored table is bf16(codebook[i]); an element takes the FIRST index of minimal |w - codebook[i]| over the fp32 codebook (X2NormalF).
__global__ void __launch_bounds__(256) lut_quantize_kernel(const uint16_t* __restrict__ src, unsigned char* __restrict__ packed, uint16_t* __restrict__ lut, int nRow,
                                                           int nCol) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nRow) return;
    const uint16_t* dat = src + (size_t)row * nCol;
    float vmax = -3.402823466e+38f, vmin = 3.402823466e+38f;
    for (int i = lane; i < nCol; i += 64) {
        const float a = bf2f(dat[i]);
        vmax = fmaxf(vmax, a), vmin = fminf(vmin, a);
    }
    vmax = wave_max(vmax), vmin = -wave_max(-vmin);
    const float abs_max = fmaxf(fabsf(vmin), fabsf(vmax));
    const float scale = abs_max > 0.0f ? (float)(1.0 / (double)abs_max) : 1.0f;
    float cb[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cb[i] = kNF4[i] / scale;
    if (lane < 16) lut[(size_t)row * 16 + lane] = f2bf(kNF4[lane] / scale);
    unsigned char* dst = packed + (size_t)row * (nCol / 2);
    for (int b = lane; b < nCol / 2; b += 64) {
        unsigned int byte = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float w = bf2f(dat[2 * b + h]);
            float best = 3.402823466e+38f;
            unsigned int id = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float d = fabsf(w - cb[i]);
                if (d < best) best = d, id = i;
            }
            byte = (byte << 4) | id;
        }
        dst[b] = (unsigned char)byte;
    }
}

// NF3_LUT::table (g_float.hpp:566)
__constant__ float kNF3[8] = {-1.0f, -0.5350227355957031f, -0.2469314038753510f, 0.0f, 0.1833375245332718f, 0.3819939494132996f, 0.6229856610298157f, 1.0f};

// the 3-bit form of the same quantiser (RT_NormalF with bits == 3): a lane packs 8 weights into 3 bytes, most significant bit first (BIT_SET_k)
__global__ void __launch_bounds__(256) lut_quantize3_kernel(const uint16_t* __restrict__ src, unsigned char* __restrict__ packed, uint16_t* __restrict__ lut, int nRow,
                                                            int nCol) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nRow) return;
    const uint16_t* dat = src + (size_t)row * nCol;
    float vmax = -3.402823466e+38f, vmin = 3.402823466e+38f;
    for (int i = lane; i < nCol; i += 64) {
        const float a = bf2f(dat[i]);
        vmax = fmaxf(vmax, a), vmin = fminf(vmin, a);
    }
    vmax = wave_max(vmax), vmin = -wave_max(-vmin);
    const float abs_max = fmaxf(fabsf(vmin), fabsf(vmax));
    const float scale = abs_max > 0.0f ? (float)(1.0 / (double)abs_max) : 1.0f;
    float cb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cb[i] = kNF3[i] / scale;
    if (lane < 8) lut[(size_t)row * 8 + lane] = f2bf(kNF3[lane] / scale);
    unsigned char* dst = packed + (size_t)row * (nCol / 8) * 3;
    for (int u = lane; u < nCol / 8; u += 64) {
        unsigned int v = 0;
#pragma unroll
        for (int h = 0; h < 8; h++) {
            const float w = bf2f(dat[8 * u + h]);
            float best = 3.402823466e+38f;
            unsigned int id = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float d = fabsf(w - cb[i]);
                if (d < best) best = d, id = i;
            }
            v = (v << 3) | id;
        }
        dst[3 * u] = (unsigned char)(v >> 16), dst[3 * u + 1] = (unsigned char)(v >> 8), dst[3 * u + 2] = (unsigned char)v;
    }
}

int lut_quantize_launch(hipStream_t st, const kf_weight* w, const uint16_t* src) {
    if (w->quant == KF_QUANT_ROW_LUT && w->type == KF_Q3 && w->gama) {
        if (w->ne1 % 8) return KF_INVALID_ARGS;
        hipLaunchKernelGGL(lut_quantize3_kernel, dim3((unsigned)((w->ne0 + 3) / 4)), dim3(256), 0, st, src, (unsigned char*)const_cast<void*>(w->data),
                           const_cast<uint16_t*>(w->gama) + w->ne0 + w->ne1, w->ne0, w->ne1);
        return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
    }
    if (w->quant != KF_QUANT_ROW_LUT || w->type != KF_Q4 || !w->gama) return KF_QUANT_ERR; /* RT_NormalF: bits == 4 || 3 */
    if (w->ne1 % 2) return KF_INVALID_ARGS;
    uint16_t* lut = const_cast<uint16_t*>(w->gama) + w->ne0 + w->ne1;
    hipLaunchKernelGGL(lut_quantize_kernel, dim3((unsigned)((w->ne0 + 3) / 4)), dim3(256), 0, st, src, (unsigned char*)const_cast<void*>(w->data), lut, w->ne0, w->ne1);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---------------------------------------------------------------- dequant / embedding rows
// one thread per 4 bytes of the stream (8 weights -> one 16-byte store); the row's table comes through the cache (32 B per row)
__device__ __forceinline__ u32x4 lut_unpack8(uint32_t d, const uint16_t* __restrict__ t) {
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t byte = (d >> (8 * k)) & 0xffu;
        o[k] = (uint32_t)t[byte >> 4] | ((uint32_t)t[byte & 15u] << 16);
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}
__global__ void __launch_bounds__(256) lut_dequant_kernel(const uint32_t* __restrict__ data, const uint16_t* __restrict__ lut, int words_per_row, size_t nwords,
                                                          uint16_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    const size_t row = i / (size_t)words_per_row;
    *reinterpret_cast<u32x4*>(out + i * 8) = lut_unpack8(data[i], lut + row * 16);
}
// 3- and 2-bit row forms (CU_Q32X_NF3 / CU_Q32X_ / CU_Q22X_ / CU_Q22X_RTN): one thread per 8 weights = BITS bytes of the stream, most significant bit first
template <int BITS, bool RTN>
__global__ void __launch_bounds__(256) row_dequant_kernel(const unsigned char* __restrict__ data, const uint16_t* __restrict__ tab, int units_per_row, size_t nunits,
                                                          uint16_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nunits) return;
    const size_t row = i / (size_t)units_per_row;
    const unsigned char* q = data + i * BITS;
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < BITS; b++) v = (v << 8) | q[b];
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint16_t e[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t id = (v >> (BITS * (7 - (2 * k + h)))) & ((1u << BITS) - 1u);
            if (RTN) { /* (zero + step * (floatGama)id) * sR, every operator a bf16 operator, sR = 1 */
                const float zero = bf2f(tab[row * 2]), step = bf2f(tab[row * 2 + 1]);
                e[h] = f2bf(zero + round_bf16(step * (float)id));
            } else {
                e[h] = tab[row * (1u << BITS) + id];
            }
        }
        o[k] = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
    }
    *reinterpret_cast<u32x4*>(out + i * 8) = u32x4{o[0], o[1], o[2], o[3]};
}

int lut_dequant_launch(hipStream_t st, const kf_weight* w, uint16_t* out) {
    if (!w->gama) return KF_QUANT_ERR;
    if (w->type != KF_Q4 || w->quant != KF_QUANT_ROW_LUT) {
        if (w->ne1 % 8) return KF_INVALID_ARGS;
        const size_t nunits = (size_t)w->ne0 * w->ne1 / 8;
        const dim3 grid((unsigned)((nunits + 255) / 256));
        const unsigned char* d = (const unsigned char*)w->data;
        const uint16_t* tab = w->gama + w->ne0 + w->ne1;
        if (w->quant == KF_QUANT_ROW_LUT && w->type == KF_Q3)
            hipLaunchKernelGGL((row_dequant_kernel<3, false>), grid, dim3(256), 0, st, d, tab, w->ne1 / 8, nunits, out);
        else if (w->quant == KF_QUANT_ROW_LUT && w->type == KF_Q2)
            hipLaunchKernelGGL((row_dequant_kernel<2, false>), grid, dim3(256), 0, st, d, tab, w->ne1 / 8, nunits, out);
        else if (w->quant == KF_QUANT_ROW_RTN && w->type == KF_Q2)
            hipLaunchKernelGGL((row_dequant_kernel<2, true>), grid, dim3(256), 0, st, d, tab, w->ne1 / 8, nunits, out);
        else
            return KF_UNSUPPORTED_DATATYPE;
        return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
    }
    if (w->ne1 % 8) return KF_INVALID_ARGS;
    const size_t nwords = (size_t)w->ne0 * w->ne1 / 8;
    hipLaunchKernelGGL(lut_dequant_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, (const uint32_t*)w->data, w->gama + w->ne0 + w->ne1, w->ne1 / 8, nwords,
                       out);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// token selection as embed_kernel (kf_ops.hip): by value, from d_token[blockIdx.y], or from the decode state (+ teacher forcing)
__global__ void lut_embed_kernel(const uint32_t* __restrict__ data, const uint16_t* __restrict__ lut, int words_per_row, int token_, const int32_t* d_token,
                                 const int32_t* d_state, const int32_t* d_forced, uint16_t* __restrict__ out, int n_rows) {
    int token = token_;
    if (d_token) token = d_token[blockIdx.y];
    if (d_state) {
        token = d_state[0];
        if (d_forced) {
            const int f = d_forced[d_state[1]];
            if (f >= 0) token = f;
        }
    }
    if (token < 0 || token >= n_rows) token = 0; /* ids come from device memory: never index outside the table */
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words_per_row) return;
    out += (size_t)blockIdx.y * words_per_row * 8;
    *reinterpret_cast<u32x4*>(out + (size_t)i * 8) = lut_unpack8(data[(size_t)token * words_per_row + i], lut + (size_t)token * 16);
}
int lut_embed_launch(hipStream_t st, const kf_weight* w, int token, const int32_t* d_token, const int32_t* d_state, const int32_t* d_forced, uint16_t* out, int n_tok) {
    if (w->type != KF_Q4 || w->quant != KF_QUANT_ROW_LUT) return KF_UNSUPPORTED_DATATYPE; /* TokenEmbed::cuInfer: Q4 only (Q3: assert(0), NeuronFuse.cu:187-192) */
    if (!w->gama) return KF_QUANT_ERR;
    if (w->ne1 % 8) return KF_INVALID_ARGS;
    if (!d_token && !d_state && (token < 0 || token >= w->ne0)) return KF_INVALID_ARGS;
    if (n_tok < 1 || (n_tok > 1 && !d_token)) return KF_INVALID_ARGS;
    const int wpr = w->ne1 / 8;
    hipLaunchKernelGGL(lut_embed_kernel, dim3((wpr + 63) / 64, n_tok), dim3(64), 0, st, (const uint32_t*)w->data, w->gama + w->ne0 + w->ne1, wpr, token, d_token, d_state,
                       d_forced, out, w->ne0);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
