// kf_ops.hip -- the small operators of the decode step and the (load-time) dequant / quantise kernels.
#include "kf_gemm_common.h"

namespace kf {

// ---------------------------------------------------------------- token-batch norms: one WAVE per row
// RMSNorm / LayerNorm of many rows (prefill chunks, training batches): a workgroup of 4 waves takes 4 rows, lane l holds the 8-element vectors
// l, l + 64, ... of its row in registers (dim a multiple of 8, <= 512 * NV), the fp64 row sums need only the wave's own reduction -- no LDS, no
// barrier, one pass over the row.  Same arithmetic as the one-workgroup-per-row kernels below (fp64 sums are order-free: identical bits).
template <bool IS_LN, int NV>
__global__ void __launch_bounds__(256) norm_rows_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ b,
                                                        uint16_t* __restrict__ y, int rows, int dim, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = dim >> 3;
    const uint16_t* xr = x + (size_t)row * dim;
    u32x4 xv[NV];
    bool has[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        has[j] = lane + 64 * j < nvec;
        xv[j] = *reinterpret_cast<const u32x4*>(xr + (size_t)(has[j] ? lane + 64 * j : 0) * 8);
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        if (!has[j]) continue;
        const uint32_t q[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double lo = (double)bf_lo(q[k]), hi = (double)bf_hi(q[k]);
            if (IS_LN) acc += lo + hi;
            else acc = fma(lo, lo, acc), acc = fma(hi, hi, acc);
        }
    }
    acc = wave_sum_f64_fast(acc);
    float m = 0.0f, s;
    if (IS_LN) {
        m = (float)acc / (float)dim;
        double sq = 0.0;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            if (!has[j]) continue;
            const uint32_t q[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float d0 = bf_lo(q[k]) - m, d1 = bf_hi(q[k]) - m;
                sq = fma((double)d0, (double)d0, sq), sq = fma((double)d1, (double)d1, sq);
            }
        }
        sq = wave_sum_f64_fast(sq);
        s = 1.0f / sqrtf((float)sq / (float)dim + eps);
    } else {
        s = 1.0f / sqrtf(fmaf((float)acc, 1.0f / (float)dim, eps));
    }
    uint16_t* yr = y + (size_t)row * dim;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        if (!has[j]) continue;
        const size_t o = (size_t)(lane + 64 * j) * 8;
        const u32x4 wv = *reinterpret_cast<const u32x4*>(w + o);
        u32x4 bv = u32x4{0, 0, 0, 0};
        if (IS_LN && b) bv = *reinterpret_cast<const u32x4*>(b + o);
        const uint32_t q[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, ww[4] = {wv.x, wv.y, wv.z, wv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float o0, o1;
            if (IS_LN) {
                const float n0 = s * (bf_lo(q[k]) - m), n1 = s * (bf_hi(q[k]) - m);
                o0 = n0 * bf_lo(ww[k]) + bf_lo(bb[k]), o1 = n1 * bf_hi(ww[k]) + bf_hi(bb[k]);
            } else {
                o0 = (bf_lo(q[k]) * s) * bf_lo(ww[k]), o1 = (bf_hi(q[k]) * s) * bf_hi(ww[k]);
            }
            r[k] = pack_bf16x2(o0, o1);
        }
        *reinterpret_cast<u32x4*>(yr + o) = u32x4{r[0], r[1], r[2], r[3]};
    }
    if (lane == 0) {
        if (IS_LN && mean) mean[row] = m;
        if (rstd) rstd[row] = s;
    }
}
template <bool IS_LN>
static bool norm_rows_launch(hipStream_t st, const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* y, int rows, int dim, float eps, float* mean, float* rstd) {
    if (rows < 16 || (dim % 8) != 0 || dim > 4096 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(b)) & 15))
        return false;
    const dim3 grid((rows + 3) / 4);
    const int nv = (dim / 8 + 63) / 64;
    if (nv <= 2) hipLaunchKernelGGL((norm_rows_kernel<IS_LN, 2>), grid, dim3(256), 0, st, x, w, b, y, rows, dim, eps, mean, rstd);
    else if (nv <= 4) hipLaunchKernelGGL((norm_rows_kernel<IS_LN, 4>), grid, dim3(256), 0, st, x, w, b, y, rows, dim, eps, mean, rstd);
    else hipLaunchKernelGGL((norm_rows_kernel<IS_LN, 8>), grid, dim3(256), 0, st, x, w, b, y, rows, dim, eps, mean, rstd);
    return true;
}

// ---------------------------------------------------------------- RMSNorm: rms_norm_kernel (layernorm.cuh:800-847)
__global__ void __launch_bounds__(256) rmsnorm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, uint16_t* __restrict__ y, int dim,
                                                      float eps, float inv_dim, float* rstd) {
    __shared__ double red[16];
    const uint16_t* xr = x + (size_t)blockIdx.x * dim;
    uint16_t* yr = y + (size_t)blockIdx.x * dim;
    const double ss = block_sumsq_bf16(xr, dim, red);
    const float mul = 1.0f / sqrtf(fmaf((float)ss, inv_dim, eps));
    for (int i = threadIdx.x; i < dim; i += blockDim.x) yr[i] = f2bf((bf2f(xr[i]) * mul) * bf2f(w[i]));
    if (rstd && threadIdx.x == 0) rstd[blockIdx.x] = mul;
}
int rmsnorm_launch(hipStream_t st, const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int dim, float eps, float* rstd) {
    if (dim % 2 != 0 || rows <= 0) return KF_RMS_PARAMS; /* CU_rms_infer refuses odd dims (layernorm.cuh:851-854) */
    if (norm_rows_launch<false>(st, x, w, nullptr, y, rows, dim, eps, nullptr, rstd)) return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 0, st, x, w, y, dim, eps, 1.0f / (float)dim, rstd);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---------------------------------------------------------------- LayerNorm forward (CU_lm_forward, layernorm.cuh:226-300; GPT-2 family) and GELU
// One workgroup per row; the two sums in fp64 (order-independent, like RMSNorm), s = 1/sqrtf(v + eps) with IEEE ops, out = bf16(s*(x-m)*w + b).
__device__ __forceinline__ double block_sum_f64(double acc, double* red) {
    acc = wave_sum_f64(acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < nw; w++) tot += red[w];
    __syncthreads();
    return tot;
}
__global__ void __launch_bounds__(256) layernorm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ b,
                                                        uint16_t* __restrict__ y, int dim, float eps, float* mean, float* rstd) {
    __shared__ double red[16];
    const uint16_t* xr = x + (size_t)blockIdx.x * dim;
    uint16_t* yr = y + (size_t)blockIdx.x * dim;
    double acc = 0.0;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) acc += (double)bf2f(xr[i]);
    const float m = (float)block_sum_f64(acc, red) / (float)dim;
    acc = 0.0;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        const float d = bf2f(xr[i]) - m;
        acc = fma((double)d, (double)d, acc);
    }
    const float v = (float)block_sum_f64(acc, red) / (float)dim;
    const float s = 1.0f / sqrtf(v + eps);
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        const float n = s * (bf2f(xr[i]) - m);
        const float o = n * bf2f(w[i]) + (b ? bf2f(b[i]) : 0.0f);
        yr[i] = f2bf(o);
    }
    if (threadIdx.x == 0) {
        if (mean) mean[blockIdx.x] = m;
        if (rstd) rstd[blockIdx.x] = s;
    }
}
int layernorm_launch(hipStream_t st, const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* y, int rows, int dim, float eps, float* mean, float* rstd) {
    if (rows <= 0 || dim <= 0) return KF_INVALID_ARGS;
    if (norm_rows_launch<true>(st, x, w, b, y, rows, dim, eps, mean, rstd)) return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, st, x, w, b, y, dim, eps, mean, rstd);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}
// GELU (tanh form), gelu_forward_kernel2 (Activation.cu:23-40); tanh from the fixed exp recipe, as oracle/kf_oracle.c kfo_gelu
__device__ __forceinline__ float kf_tanhf(float z) {
    if (z > 10.0f) return 1.0f;
    if (z < -10.0f) return -1.0f;
    const float e = kf_expf(2.0f * z);
    return (e - 1.0f) / (e + 1.0f);
}
__device__ __forceinline__ float gelu_f(float xi) {
    const float cube = 0.044715f * xi * xi * xi;
    return 0.5f * xi * (1.0f + kf_tanhf(0.797884583473205566406250f * (xi + cube)));
}
// 8 elements per thread (16-byte loads and stores) when both pointers are 16-byte aligned, one element per thread otherwise
__global__ void gelu_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, size_t n, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nvec) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + i * 8);
        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = pack_bf16x2(gelu_f(bf_lo(xw[k])), gelu_f(bf_hi(xw[k])));
        *reinterpret_cast<u32x4*>(y + i * 8) = u32x4{o[0], o[1], o[2], o[3]};
    }
    const size_t t = nvec * 8 + i;
    if (t < n && (nvec == 0 || i < 8)) y[t] = f2bf(gelu_f(bf2f(x[t])));
}
int gelu_launch(hipStream_t st, const uint16_t* x, uint16_t* y, size_t n) {
    const size_t nvec = (((uintptr_t)x | (uintptr_t)y) & 15) ? 0 : n / 8;
    const size_t threads = nvec ? (nvec > 8 ? nvec : 8) : n;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, x, y, n, nvec);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// GELU backward in place (gelu_backward_inplace_kernel, Activation.cu:42-60): d = bf16(local_grad(x) * d),
// local_grad = 0.5 (1 + tanh z) + x * 0.5 * sech^2 z * sqrt(2/pi) * (1 + 3 * 0.044715 x^2), z = sqrt(2/pi) (x + 0.044715 x^3).
// tanh and sech^2 = 1 / cosh^2 both from ONE e = kf_expf(2z): tanh = (e-1)/(e+1), sech^2 = 4e / ((e+1)(e+1)); saturated beyond |z| = 10
// exactly like the forward's kf_tanhf (the reference calls tanhf and coshf of the CUDA libm).
__device__ __forceinline__ float gelu_grad(float xi, float d) {
    const float cube = 0.044715f * xi * xi * xi;
    const float z = 0.797884583473205566406250f * (xi + cube);
    float th, sech2;
    if (z > 10.0f) th = 1.0f, sech2 = 0.0f;
    else if (z < -10.0f) th = -1.0f, sech2 = 0.0f;
    else {
        const float e = kf_expf(2.0f * z), e1 = e + 1.0f;
        th = (e - 1.0f) / e1;
        sech2 = (4.0f * e) / (e1 * e1);
    }
    const float local_grad = 0.5f * (1.0f + th) + xi * 0.5f * sech2 * 0.797884583473205566406250f * (1.0f + 3.0f * 0.044715f * xi * xi);
    return local_grad * d;
}
// 8 elements per thread (16-byte loads) where the pointers allow it; a scalar tail
__global__ void gelu_backward_kernel(uint16_t* __restrict__ d_in_out, const uint16_t* __restrict__ x, size_t n, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nvec) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + i * 8), dv = *reinterpret_cast<const u32x4*>(d_in_out + i * 8);
        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = pack_bf16x2(gelu_grad(bf_lo(xw[k]), bf_lo(dw[k])), gelu_grad(bf_hi(xw[k]), bf_hi(dw[k])));
        *reinterpret_cast<u32x4*>(d_in_out + i * 8) = u32x4{o[0], o[1], o[2], o[3]};
    }
    const size_t t = nvec * 8 + i;
    if (t < n && (nvec == 0 || i < 8)) d_in_out[t] = f2bf(gelu_grad(bf2f(x[t]), bf2f(d_in_out[t])));
}
int gelu_backward_launch(hipStream_t st, uint16_t* d_in_out, const uint16_t* x, size_t n) {
    const size_t nvec = (((uintptr_t)d_in_out | (uintptr_t)x) & 15) ? 0 : n / 8; /* unaligned: one element per thread */
    const size_t threads = nvec ? (nvec > 8 ? nvec : 8) : n;
    hipLaunchKernelGGL(gelu_backward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_in_out, x, n, nvec);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}
// SwiGLU backward (CU_swiglu_back_v0, Activation.cu:245-260): sig = 1 / (1 + exp(-gate));
// delta_gate = bf16(delta * up * sig * (1 + gate * (1 - sig))); delta_in_out = bf16(delta * gate * sig)   (the gradient of the up projection).
// The reference rounds both stores stochastically (seed 42); round-to-nearest here, as everywhere.
__device__ __forceinline__ void swiglu_grad(float xiW, float xiV, float delta, float& d_gate, float& d_up) {
    const float sigW = 1.0f / (1.0f + kf_expf(-xiW));
    d_gate = delta * xiV * sigW * (1.0f + xiW * (1.0f - sigW));
    d_up = delta * xiW * sigW;
}
__global__ void swiglu_backward_kernel(uint16_t* __restrict__ delta_in_out, uint16_t* __restrict__ delta_gate, const uint16_t* __restrict__ gate,
                                       const uint16_t* __restrict__ up, size_t n, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nvec) {
        const u32x4 gv = *reinterpret_cast<const u32x4*>(gate + i * 8), uv = *reinterpret_cast<const u32x4*>(up + i * 8),
                    dv = *reinterpret_cast<const u32x4*>(delta_in_out + i * 8);
        const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, uw[4] = {uv.x, uv.y, uv.z, uv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
        uint32_t og[4], ou[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float g0, u0, g1, u1;
            swiglu_grad(bf_lo(gw[k]), bf_lo(uw[k]), bf_lo(dw[k]), g0, u0);
            swiglu_grad(bf_hi(gw[k]), bf_hi(uw[k]), bf_hi(dw[k]), g1, u1);
            og[k] = pack_bf16x2(g0, g1), ou[k] = pack_bf16x2(u0, u1);
        }
        *reinterpret_cast<u32x4*>(delta_gate + i * 8) = u32x4{og[0], og[1], og[2], og[3]};
        *reinterpret_cast<u32x4*>(delta_in_out + i * 8) = u32x4{ou[0], ou[1], ou[2], ou[3]};
    }
    const size_t t = nvec * 8 + i;
    if (t < n && (nvec == 0 || i < 8)) {
        float g0, u0;
        swiglu_grad(bf2f(gate[t]), bf2f(up[t]), bf2f(delta_in_out[t]), g0, u0);
        delta_gate[t] = f2bf(g0), delta_in_out[t] = f2bf(u0);
    }
}
int swiglu_backward_launch(hipStream_t st, uint16_t* delta_in_out, uint16_t* delta_gate, const uint16_t* gate, const uint16_t* up, size_t n) {
    const size_t nvec = (((uintptr_t)delta_in_out | (uintptr_t)delta_gate | (uintptr_t)gate | (uintptr_t)up) & 15) ? 0 : n / 8;
    const size_t threads = nvec ? (nvec > 8 ? nvec : 8) : n;
    hipLaunchKernelGGL(swiglu_backward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, delta_in_out, delta_gate, gate, up, n, nvec);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// RoPE backward, rotate-half form (the transpose of CU_rope2_v0, operator.cuh:734-772), in place on a gradient with rows of n_head * hd:
// (g_j, g_{j+hd/2}) -> (g_j c + g_{j+hd/2} s, g_{j+hd/2} c - g_j s) with (c, s) = table[pos][j]; position of row t = pos0 + t % seq_len.
__global__ void rope_backward_kernel(uint16_t* __restrict__ d, const float* __restrict__ table, int pos0, int seq_len, long long stride, int n_head, int hd) {
    const int half = hd >> 1, t = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x; /* (head, pair) */
    if (i >= n_head * half) return;
    const int h = i / half, j = i - h * half;
    const float* cs = table + ((size_t)(pos0 + t % seq_len) * half + j) * 2;
    uint16_t* p = d + (size_t)t * stride + (size_t)h * hd + j;
    const float g0 = bf2f(p[0]), g1 = bf2f(p[half]), c = cs[0], sn = cs[1];
    const float a = g0 * c, b = g1 * sn, cc = g1 * c, dd = g0 * sn;
    p[0] = f2bf(a + b);
    p[half] = f2bf(cc - dd);
}
int rope_backward_launch(hipStream_t st, uint16_t* d, const float* table, int pos0, int n_tok, int seq_len, long long stride, int n_head, int hd) {
    if (n_tok < 1 || seq_len < 1 || n_head < 1 || hd < 2 || (hd & 1)) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(rope_backward_kernel, dim3((n_head * (hd / 2) + 255) / 256, n_tok), dim3(256), 0, st, d, table, pos0, seq_len, stride, n_head, hd);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---------------------------------------------------------------- SwiGLU / add
__global__ void swiglu_kernel(const uint16_t* __restrict__ gate, const uint16_t* __restrict__ up, uint16_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float g = bf2f(gate[i]), u = bf2f(up[i]);
        out[i] = f2bf((g * u) / (1.0f + kf_expf(-g)));
    }
}
int swiglu_launch(hipStream_t st, const uint16_t* gate, const uint16_t* up, uint16_t* out, int n) {
    hipLaunchKernelGGL(swiglu_kernel, dim3((n + 255) / 256), dim3(256), 0, st, gate, up, out, n);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}
__global__ void add_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}
int add_launch(hipStream_t st, const uint16_t* a, const uint16_t* b, uint16_t* out, int n) {
    hipLaunchKernelGGL(add_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, out, n);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// epilogue of the library GEMM path (kf_abi.hip lib_gemm): y[t][m] = bf16(y + bias[m]), then bf16(residual + y) -- the order of the fused
// epilogue (bias, store, residual; kf_gemm_common.h), applied to the GEMM's bf16 result
__global__ void bias_residual_kernel(uint16_t* __restrict__ y, const uint16_t* __restrict__ bias, const uint16_t* __restrict__ residual, size_t nvec, int M) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    const int m0 = (int)((i * 8) % (size_t)M);
    const u32x4 yv = *reinterpret_cast<const u32x4*>(y + i * 8);
    uint32_t w[4] = {yv.x, yv.y, yv.z, yv.w};
    if (bias) {
        const u32x4 bv = *reinterpret_cast<const u32x4*>(bias + m0);
        const uint32_t b[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = pack_bf16x2(bf_lo(w[k]) + bf_lo(b[k]), bf_hi(w[k]) + bf_hi(b[k]));
    }
    if (residual) {
        const u32x4 rv = *reinterpret_cast<const u32x4*>(residual + i * 8);
        const uint32_t r[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = pack_bf16x2(bf_lo(r[k]) + bf_lo(w[k]), bf_hi(r[k]) + bf_hi(w[k]));
    }
    *reinterpret_cast<u32x4*>(y + i * 8) = u32x4{w[0], w[1], w[2], w[3]};
}
int bias_residual_launch(hipStream_t st, uint16_t* y, const uint16_t* bias, const uint16_t* residual, size_t n, int M) {
    if ((M % 8) != 0) return KF_INVALID_ARGS;
    const size_t nvec = n * (size_t)M / 8;
    hipLaunchKernelGGL(bias_residual_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, y, bias, residual, nvec, M);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---------------------------------------------------------------- block dequant (CU_Q128toX_, T.cu:245-294; CU_F82Float)
__device__ __forceinline__ float dq(float step, float zero, float qm) { return round_bf16(round_bf16(step * qm) - zero); }

// one thread per 16-byte block; out points at the first element of the block
__device__ __forceinline__ void dequant_block(int fmt, u32x4 w, float step, float zero, int qBias, uint16_t* out) {
    const uint32_t d[4] = {w.w, w.z, w.y, w.x}; /* d[0] = first elements of a Packed128 */
    if (fmt == FMT_BF16) {
        *reinterpret_cast<u32x4*>(out) = w;
    } else if (fmt == FMT_F8) { /* 16 bytes -> two 16-byte stores (the tile kernels' conversion: v_cvt_pk_f32_bf8) */
        u32x4* o = reinterpret_cast<u32x4*>(out);
        o[0] = frag_f8(w.x, w.y), o[1] = frag_f8(w.z, w.w);
    } else if (fmt == FMT_Q4) { /* 32 weights -> four 16-byte stores; frag_q4 forms step * (q - qBias) with one exact fma, then the two bf16 roundings of dq() */
        u32x4* o = reinterpret_cast<u32x4*>(out);
        const float nb = -((float)qBias * step), s16 = step * 0.0625f;
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = frag_q4(d[k], step, s16, nb, zero);
    } else if (fmt == FMT_Q2) {
        for (int k = 0; k < 64; k++) out[k] = f2bf(dq(step, zero, (float)((int)((d[k >> 4] >> (30 - 2 * (k & 15))) & 0x3u) - qBias)));
    } else {
        for (int k = 0; k < 128; k++) out[k] = f2bf(dq(step, zero, (float)((int)((d[k >> 5] >> (31 - (k & 31))) & 0x1u) - qBias)));
    }
}
static int fmt_of(int type, int* epb) {
    switch (type) {
        case KF_BF16: *epb = 8; return FMT_BF16;
        case KF_F8E5M2: *epb = 16; return FMT_F8;
        case KF_Q4: *epb = 32; return FMT_Q4;
        case KF_T_SIGN: *epb = 64; return FMT_Q2;
        case KF_BOOL1: case KF_T_BINARY: *epb = 128; return FMT_Q1;
        default: *epb = 0; return -1;
    }
}
// ilv_n > 1: the rows land INTERLEAVED with those of ilv_n - 1 other matrices of the same shape, in blocks of 16 rows: row r -> row (r / 16) * 16 * ilv_n + 16 * ilv_i + r % 16
// of the stacked copy (gate | up side by side in every 32-row block of the tile GEMM's operand: its SwiGLU epilogue finds both projections of an FFN row in one lane)
__global__ void dequant_kernel(int fmt, int epb, const u32x4* __restrict__ data, const uint16_t* __restrict__ zero, const uint16_t* __restrict__ step, int lGroup,
                               int qBias, size_t nblocks, size_t block0, uint16_t* __restrict__ out, int bpr, int ilv_n, int ilv_i) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const size_t gb = block0 + b;
    float st = 0.f, ze = 0.f;
    if (zero) {
        const size_t gi = gb * epb / lGroup;
        st = bf2f(step[gi]), ze = bf2f(zero[gi]);
    }
    size_t ob = b;
    if (ilv_n > 1) {
        const size_t r = b / bpr, c = b - r * bpr;
        ob = ((r >> 4) * 16 * ilv_n + 16 * ilv_i + (r & 15)) * bpr + c;
    }
    dequant_block(fmt, data[gb], st, ze, qBias, out + ob * epb);
}
int dequant_launch(hipStream_t st, const kf_weight* w, uint16_t* out, int ilv_n, int ilv_i) {
    if (is_row_lut(w)) return ilv_n > 1 ? KF_UNSUPPORTED_DATATYPE : lut_dequant_launch(st, w, out);
    int epb;
    const int fmt = fmt_of(w->type, &epb);
    if (fmt < 0) return KF_UNSUPPORTED_DATATYPE;
    const size_t n = (size_t)w->ne0 * w->ne1;
    if (n % epb) return KF_INVALID_ARGS;
    const uint16_t *zero = nullptr, *step = nullptr;
    if (fmt >= FMT_Q4) {
        if (!w->gama || w->lGroup <= 0) return KF_QUANT_ERR;
        zero = w->gama + w->ne0 + w->ne1;
        step = zero + n / w->lGroup;
    }
    const size_t nb = n / epb;
    if (ilv_n > 1 && (w->ne1 % epb != 0 || w->ne0 % 16 != 0)) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(dequant_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, fmt, epb, (const u32x4*)w->data, zero, step, w->lGroup, w->qBias, nb,
                       (size_t)0, out, w->ne1 / epb, ilv_n, ilv_i);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---------------------------------------------------------------- embedding row (CU_embed_forw_1/_q4, embed.cuh:54-132)
__global__ void embed_kernel(int fmt, int epb, const u32x4* __restrict__ data, const uint16_t* __restrict__ zero, const uint16_t* __restrict__ step, int lGroup,
                             int qBias, int nBlk, int token_, const int32_t* d_token, const int32_t* d_state, const int32_t* d_forced,
                             uint16_t* __restrict__ out, int n_rows) {
    int token = token_;
    if (d_token) token = d_token[blockIdx.y]; /* blockIdx.y = row of a token batch (0 for a single token) */
    if (token < 0 || token >= n_rows) token = 0; /* ids come from device memory: never index outside the table */
    out += (size_t)blockIdx.y * nBlk * epb;
    if (d_state) {
        token = d_state[0];
        if (d_forced) {
            const int f = d_forced[d_state[1]];
            if (f >= 0) token = f;
        }
        if (token < 0 || token >= n_rows) token = 0;
    }
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlk) return;
    const size_t gb = (size_t)token * nBlk + b;
    float st = 0.f, ze = 0.f;
    if (zero) {
        const size_t gi = gb * epb / lGroup;
        st = bf2f(step[gi]), ze = bf2f(zero[gi]);
    }
    dequant_block(fmt, data[gb], st, ze, qBias, out + (size_t)b * epb);
}
int embed_launch(hipStream_t st, const kf_weight* w, int token, const int32_t* d_token, const int32_t* d_state, const int32_t* d_forced, uint16_t* out,
                 int n_tok) {
    if (is_row_lut(w)) return lut_embed_launch(st, w, token, d_token, d_state, d_forced, out, n_tok);
    int epb;
    const int fmt = fmt_of(w->type, &epb);
    if (fmt < 0) return KF_UNSUPPORTED_DATATYPE;
    if (w->ne1 % epb) return KF_INVALID_ARGS;
    if (!d_token && !d_state && (token < 0 || token >= w->ne0)) return KF_INVALID_ARGS;
    const uint16_t *zero = nullptr, *step = nullptr;
    if (fmt >= FMT_Q4) {
        if (!w->gama || w->lGroup <= 0) return KF_QUANT_ERR;
        zero = w->gama + w->ne0 + w->ne1;
        step = zero + (size_t)w->ne0 * w->ne1 / w->lGroup;
    }
    const int nBlk = w->ne1 / epb;
    if (n_tok < 1 || (n_tok > 1 && !d_token)) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(embed_kernel, dim3((nBlk + 63) / 64, n_tok), dim3(64), 0, st, fmt, epb, (const u32x4*)w->data, zero, step, w->lGroup, w->qBias, nBlk, token,
                       d_token, d_state, d_forced, out, w->ne0);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---------------------------------------------------------------- GeneratOnPrompt::Sample, non-greedy branch (GoPT.cpp:614-630), on the device
// One workgroup.  Candidate set exactly as TOPK_heap::Select builds it (GoPT.cpp:666-704: the std::priority_queue<int> is ordered by
// token index, so indices 0..k-2 stay and only the newest entry is ever replaced): {0..k-2} + first maximum over i >= k-1; stable
// descending order by logit; p = expf((a - max)/T) / sum with the sequential fp32 sum of UpdateLogits (GoPT.cpp:754-769); TopP cut
// (GoPT.cpp:729-751; top_p >= 1 keeps all k); coin from xorshift64* (GoPT.cpp:594-600) walked along the CDF (GoPT.cpp:771-790).
// The rng state lives in device memory and advances once per call, so a captured graph replays a fresh coin every step.
// Same operation sequence as oracle/kf_oracle.c kfo_sample: token ids agree bit for bit.
constexpr int KF_SAMPLE_MAX_K = 1024;
// TRUE_TOPK (kf_sample_topk): the k LARGEST logits instead -- what the reference's TopK evidently means to keep -- found with a two-level
// radix select on the order-preserving 16-bit key of a bf16 (high byte, then low byte: two 256-bin histograms), ties at the threshold
// resolved towards the lower token index; everything after the candidate set is the same code.
__device__ __forceinline__ unsigned int bf16_key(uint16_t u) { return (u & 0x8000u) ? (~(unsigned int)u & 0xFFFFu) : ((unsigned int)u | 0x8000u); }

template <bool TRUE_TOPK>
__global__ void __launch_bounds__(1024) sample_kernel(const uint16_t* __restrict__ logits, int n, int k, float temperature, float top_p,
                                                      unsigned long long* rng, int32_t* d_token, int32_t* d_state, int32_t* d_tokens_out,
                                                      const int32_t* d_forced, int n_forced) {
    __shared__ float sv[KF_SAMPLE_MAX_K];
    __shared__ int si[KF_SAMPLE_MAX_K];
    __shared__ float pv[KF_SAMPLE_MAX_K];
    __shared__ int picks[KF_SAMPLE_MAX_K];
    __shared__ float s_sum;
    const int tid = threadIdx.x;
    // teacher-forced next token (prompt prefill through the decode path): the reference's prefill loop never samples (GoPT.cpp:1139-1146),
    // so no coin is drawn; only the position advances
    if (d_state && d_forced) {
        const int p = d_state[1];
        if (p + 1 < n_forced && d_forced[p + 1] >= 0) {
            __syncthreads();
            if (tid == 0) {
                if (d_tokens_out) d_tokens_out[p] = d_forced[p + 1];
                d_state[0] = d_forced[p + 1];
                d_state[1] = p + 1;
            }
            return;
        }
    }
    if constexpr (!TRUE_TOPK) {
        // first maximum over i >= k-1
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int i = k - 1 + tid; i < n; i += blockDim.x) {
            const float v = bf2f(logits[i]);
            if (v > bv) bv = v, bi = i;
        }
        sv[tid] = bv, si[tid] = bi;
        __syncthreads();
        for (int m = blockDim.x >> 1; m > 0; m >>= 1) {
            if (tid < m) {
                const float ov = sv[tid + m];
                const int oi = si[tid + m];
                if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) sv[tid] = ov, si[tid] = oi;
            }
            __syncthreads();
        }
        const int last = si[0];
        __syncthreads();
        // extraction order: last, k-2, .., 0; stable rank sort, descending by logit
        int my = 0;
        float mv = 0.f;
        if (tid < k) {
            my = tid == 0 ? last : k - 1 - tid;
            mv = bf2f(logits[my]);
            sv[tid] = mv;
        }
        __syncthreads();
        if (tid < k) {
            int rank = 0;
            for (int m = 0; m < k; m++) {
                const float o = sv[m];
                rank += (o > mv || (o == mv && m < tid)) ? 1 : 0;
            }
            picks[rank] = my;
            pv[rank] = mv;
        }
        __syncthreads();
    } else {
        __shared__ int hist[256];
        __shared__ int scan[1024];
        __shared__ int sel[4]; /* {bin, count above, appended so far, -} */
        // ---- level 1: high byte of the key
        auto select_bin = [&](int want) { /* hist filled; thread 0 finds the bin holding the want-th largest, sel = {bin, count above it} */
            __syncthreads();
            if (tid == 0) {
                int cum = 0, b = 255;
                for (; b > 0; b--) {
                    if (cum + hist[b] >= want) break;
                    cum += hist[b];
                }
                sel[0] = b, sel[1] = cum;
            }
            __syncthreads();
        };
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) atomicAdd(&hist[bf16_key(logits[i]) >> 8], 1);
        select_bin(k);
        const int b1 = sel[0], above1 = sel[1];
        __syncthreads();
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            const unsigned int key = bf16_key(logits[i]);
            if ((int)(key >> 8) == b1) atomicAdd(&hist[key & 0xFF], 1);
        }
        select_bin(k - above1);
        const unsigned int T = ((unsigned int)b1 << 8) | (unsigned int)sel[0];
        const int n_gt = above1 + sel[1], r_eq = k - n_gt; /* keys > T: all n_gt of them; keys == T: the r_eq with the lowest indices */
        __syncthreads();
        if (tid == 0) sel[2] = 0;
        // every thread owns a contiguous index range, so the running count of == T keys gives their order by index
        const int chunk = (n + (int)blockDim.x - 1) / (int)blockDim.x, i0 = tid * chunk, i1 = min(n, i0 + chunk);
        int eq = 0;
        for (int i = i0; i < i1; i++) eq += bf16_key(logits[i]) == T ? 1 : 0;
        scan[tid] = eq;
        __syncthreads();
        for (int off = 1; off < (int)blockDim.x; off <<= 1) { /* inclusive Hillis-Steele scan */
            const int v = tid >= off ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        int eq_rank = scan[tid] - eq; /* == T keys before this thread's range */
        for (int i = i0; i < i1; i++) {
            const unsigned int key = bf16_key(logits[i]);
            int slot = -1;
            if (key > T)
                slot = atomicAdd(&sel[2], 1); /* any order: sorted below */
            else if (key == T && eq_rank++ < r_eq)
                slot = n_gt + eq_rank - 1;
            if (slot >= 0) si[slot] = i, sv[slot] = bf2f(logits[i]);
        }
        __syncthreads();
        // descending by logit, equal logits by ascending token index
        if (tid < k) {
            const float mv = sv[tid];
            const int my = si[tid];
            int rank = 0;
            for (int m = 0; m < k; m++) {
                const float o = sv[m];
                rank += (o > mv || (o == mv && si[m] < my)) ? 1 : 0;
            }
            picks[rank] = my;
            pv[rank] = mv;
        }
        __syncthreads();
    }
    const float maxLogit = pv[0];
    float e = 0.f;
    if (tid < k) e = kf_expf((pv[tid] - maxLogit) / temperature);
    __syncthreads();
    if (tid < k) pv[tid] = e;
    __syncthreads();
    if (tid == 0) {
        float sum = 0.f;
        for (int j = 0; j < k; j++) sum += pv[j];
        s_sum = sum;
    }
    __syncthreads();
    if (tid < k) pv[tid] = pv[tid] / s_sum;
    __syncthreads();
    if (tid == 0) {
        int nPick = k;
        if (top_p < 1.0f) {
            float cum = 0.f;
            int last_idx = k - 1;
            for (int j = 0; j < k; j++) {
                cum += pv[j];
                if (cum > top_p) {
                    last_idx = j;
                    break;
                }
            }
            nPick = last_idx + 1;
        }
        float ps = 0.f;
        for (int j = 0; j < nPick; j++) ps += pv[j];
        unsigned long long st = *rng;
        st ^= st >> 12;
        st ^= st << 25;
        st ^= st >> 27;
        *rng = st;
        const unsigned int u = (unsigned int)((st * 0x2545F4914F6CDD1Dull) >> 32);
        const float coin = ((float)(u >> 8) / 16777216.0f) * ps;
        float cdf = 0.f;
        int qu = picks[nPick - 1];
        for (int j = 0; j < nPick; j++) {
            cdf += pv[j];
            if (coin < cdf) {
                qu = picks[j];
                break;
            }
        }
        if (d_token) *d_token = qu;
        if (d_state) {
            const int p = d_state[1];
            if (d_tokens_out) d_tokens_out[p] = qu;
            d_state[0] = qu;
            d_state[1] = p + 1;
        }
    }
}
int sample_launch(hipStream_t st, const uint16_t* logits, int n, int top_k, float temperature, float top_p, unsigned long long* rng, int32_t* d_token,
                  int32_t* d_state, int32_t* d_tokens_out, const int32_t* d_forced, int n_forced, int true_topk) {
    const int k = top_k < n ? top_k : n;
    if (k < 2 || k >= n / 2 || k > KF_SAMPLE_MAX_K || !(temperature > 0.0f) || !(top_p > 0.0f)) return KF_INVALID_ARGS;
    if (true_topk)
        hipLaunchKernelGGL(sample_kernel<true>, dim3(1), dim3(1024), 0, st, logits, n, k, temperature, top_p, rng, d_token, d_state, d_tokens_out, d_forced,
                           n_forced);
    else
        hipLaunchKernelGGL(sample_kernel<false>, dim3(1), dim3(1024), 0, st, logits, n, k, temperature, top_p, rng, d_token, d_state, d_tokens_out, d_forced,
                           n_forced);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---------------------------------------------------------------- quantiser: GeQuant::RTN_x / YinYang (GeQuant.cpp:428-628)
// One wave per group of lGroup consecutive elements (any multiple of the 128 / bits elements of a Packed128 block).
// mode 0: RTN asymmetric, 1: RTN symmetric, 2: YinYang (step = max(1e-5, sqrt(mean(relu(a)^2))), zero = 0)
__global__ void __launch_bounds__(256) quantize_kernel(const uint16_t* __restrict__ src, unsigned char* __restrict__ packed, uint16_t* __restrict__ zero_out,
                                                       uint16_t* __restrict__ step_out, size_t nGroup, int lGroup, int bits, int mode, int qMin, int qMax,
                                                       int qBias) {
    const int lane = threadIdx.x & 63;
    const size_t g = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= nGroup) return;
    const uint16_t* dat = src + g * lGroup;
    float vmax = -3.402823466e+38f, vmin = 3.402823466e+38f;
    double vsum = 0.0;
    for (int i = lane; i < lGroup; i += 64) {
        const float a = bf2f(dat[i]);
        vmax = fmaxf(vmax, a), vmin = fminf(vmin, a);
        vsum += (a < 0.0f) ? 0.0 : (double)(a * a);
    }
    for (int m = 32; m > 0; m >>= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, m, 64));
        vmin = fminf(vmin, __shfl_xor(vmin, m, 64));
    }
    float step, zero;
    if (mode == 2) {
        vsum = wave_sum_f64(vsum);
        step = fmaxf(1e-5f, (float)sqrt(vsum / (double)lGroup));
        zero = 0.f;
    } else if (mode == 1) {
        step = fmaxf(fabsf(vmax), fabsf(vmin)) / (float)qMax;
        zero = 0.f;
    } else {
        step = (vmax - vmin) / (float)(qMax - qMin);
        zero = -vmin;
    }
    if (lane == 0) zero_out[g] = f2bf(zero), step_out[g] = f2bf(step);
    // pack (Packed128 blocks, MSB-first within high then low; PackedQ.hpp:99-239): every lane quantises elements lane, lane + 64, ...; the 64 / bits lanes whose
    // codes share a 64-bit word OR them together with xor-shuffles and the first of them stores the word (even words are `high` = bytes 8..15 of the block)
    const int epw = 64 / bits; /* elements per 64-bit word: 16 / 32 / 64 lanes side by side */
    unsigned char* dst = packed + g * ((size_t)lGroup * bits / 8);
    for (int e0 = 0; e0 < lGroup; e0 += 64) {
        const int e = e0 + lane;
        unsigned long long word = 0; /* lanes past a group that is not a multiple of 64 long (32, 96 at 4 bits) take part in the shuffles with nothing */
        if (e < lGroup) {
            const float a = bf2f(dat[e]);
            int q = (int)roundf((a + zero) / step);
            q = q < qMin ? qMin : (q > qMax ? qMax : q);
            word = (unsigned long long)((q + qBias) & ((1 << bits) - 1)) << (64 - bits * ((e % epw) + 1));
        }
        for (int m = 1; m < epw; m <<= 1) word |= __shfl_xor(word, m, 64);
        if (e < lGroup && (lane & (epw - 1)) == 0) {
            const int wi = e / epw; /* word wi: block wi / 2, high first */
            reinterpret_cast<unsigned long long*>(dst + 16 * (wi >> 1))[(wi & 1) ^ 1] = word;
        }
    }
}
// The same for the storage every 4-bit model here uses (RTN, groups of 128, asymmetric or symmetric), HBM-shaped: 16 lanes per group, a lane takes 8 consecutive elements
// as ONE 16-byte load, min / max over its 8 then four xor-shuffles inside the 16 lanes; its 8 codes are one dword of the Packed128 block (element i < 16 in `high`
// = bytes 8..15, MSB first: elements 8 j .. 8 j + 7 of a block are the dword at byte 4 (3 - j)), stored as such -- a wave writes 256 contiguous bytes.  Same arithmetic
// as quantize_kernel (fp32 min / max, IEEE division, roundf), same bits; the re-quantisation of a training step's matrices ran at 0.6 TB/s through the general kernel.
__global__ void __launch_bounds__(256) quantize4_g128_kernel(const uint16_t* __restrict__ src, unsigned char* __restrict__ packed, uint16_t* __restrict__ zero_out,
                                                             uint16_t* __restrict__ step_out, size_t nGroup, int mode, int qMin, int qMax, int qBias) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const size_t g = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = g < nGroup;
    u32x4 raw = u32x4{0, 0, 0, 0};
    if (live) raw = *reinterpret_cast<const u32x4*>(src + g * 128 + sub * 8);
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    float a[8];
#pragma unroll
    for (int k = 0; k < 4; k++) a[2 * k] = bf_lo(w[k]), a[2 * k + 1] = bf_hi(w[k]);
    float vmax = -3.402823466e+38f, vmin = 3.402823466e+38f;
#pragma unroll
    for (int k = 0; k < 8; k++) vmax = fmaxf(vmax, a[k]), vmin = fminf(vmin, a[k]);
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, m, 64));
        vmin = fminf(vmin, __shfl_xor(vmin, m, 64));
    }
    float step, zero;
    if (mode == 1)
        step = fmaxf(fabsf(vmax), fabsf(vmin)) / (float)qMax, zero = 0.f;
    else
        step = (vmax - vmin) / (float)(qMax - qMin), zero = -vmin;
    if (!live) return;
    if (sub == 0) zero_out[g] = f2bf(zero), step_out[g] = f2bf(step);
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int q = (int)roundf((a[k] + zero) / step);
        q = q < qMin ? qMin : (q > qMax ? qMax : q);
        word |= (uint32_t)((q + qBias) & 15) << (28 - 4 * k);
    }
    // element e = 8 sub + k of the group: block e / 32 = sub >> 2, dword j = sub & 3 of the block -> byte 4 (3 - j)
    *reinterpret_cast<uint32_t*>(packed + g * 64 + (size_t)(sub >> 2) * 16 + 4 * (3 - (sub & 3))) = word;
}
// greedy pick over each of n_rows rows of bf16 logits (sample_argmax, GoPT.cpp:602-612: the FIRST maximum over float(logits)), one workgroup per row, followed by the
// decode-state update of kf_norm_lm_head for the sequence d_seq[row]: states [seq][4] = {token, pos, ..}: tokens_out[seq][pos] = id, state = {id, pos + 1}
__global__ void __launch_bounds__(1024) argmax_rows_state_kernel(const uint16_t* __restrict__ logits, long long ld, int n, const int* __restrict__ d_seq, int32_t* __restrict__ states,
                                                                 int32_t* __restrict__ tokens_out, int tokens_stride) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const uint16_t* row = logits + (size_t)blockIdx.x * ld;
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) { /* ascending i per thread: a strict > keeps the thread's first maximum */
        const float v = bf2f(row[i]);
        if (v > best || bi == 0x7fffffff) best = v, bi = i;
    }
    auto better = [](float v, int i, float bv, int bi2) { return v > bv || (v == bv && i < bi2); };
    for (int m = 32; m > 0; m >>= 1) {
        const float ov = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (better(ov, oi, best, bi)) best = ov, bi = oi;
    }
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = best, si[threadIdx.x >> 6] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++)
            if (better(sv[w], si[w], best, bi)) best = sv[w], bi = si[w];
        const int seq = d_seq[blockIdx.x];
        int32_t* st = states + 4 * (size_t)seq;
        const int p = st[1];
        if (tokens_out) tokens_out[(size_t)seq * tokens_stride + p] = bi;
        st[0] = bi, st[1] = p + 1;
    }
}
int argmax_rows_state_launch(hipStream_t st, const uint16_t* logits, long long ld, int n, int n_rows, const int* d_seq, int32_t* states, int32_t* tokens_out, int tokens_stride) {
    if (n < 1 || n_rows < 1 || ld < n) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(argmax_rows_state_kernel, dim3(n_rows), dim3(1024), 0, st, logits, ld, n, d_seq, states, tokens_out, tokens_stride);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// block b of src (blocks src_stride bytes apart) -> dst_table[b] + dst_offset: the K / V rows of a batch of prompts into the prompts' own caches (16-byte units)
__global__ void __launch_bounds__(256) copy_blocks_kernel(void* const* __restrict__ dst_table, size_t dst_offset, const unsigned char* __restrict__ src, size_t src_stride,
                                                          size_t block_bytes) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i >= block_bytes) return;
    unsigned char* d = reinterpret_cast<unsigned char*>(dst_table[blockIdx.y]) + dst_offset;
    *reinterpret_cast<u32x4*>(d + i) = *reinterpret_cast<const u32x4*>(src + (size_t)blockIdx.y * src_stride + i);
}
int copy_blocks_launch(hipStream_t st, void* const* dst_table, size_t dst_offset, const void* src, size_t src_stride, size_t block_bytes, int n_blocks) {
    if (n_blocks < 1 || n_blocks > 65535 || (block_bytes & 15) || (src_stride & 15) || (dst_offset & 15)) return KF_INVALID_ARGS;
    if (!block_bytes) return KF_OK;
    hipLaunchKernelGGL(copy_blocks_kernel, dim3((unsigned)((block_bytes / 16 + 255) / 256), n_blocks), dim3(256), 0, st, dst_table, dst_offset, (const unsigned char*)src, src_stride,
                       block_bytes);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// bf16 -> f8e5m2 storage (Float2T<f8e5>, g_float.hpp:433-443; ToF8Ex huTensor.cu:821): float -> half round-to-nearest-even, keep the high byte
__global__ void to_f8e5m2_kernel(const uint16_t* __restrict__ src, unsigned char* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (unsigned char)(__builtin_bit_cast(unsigned short, (_Float16)bf2f(src[i])) >> 8);
}

int quantize_launch(hipStream_t st, const kf_weight* w, const uint16_t* src, int symmetric) {
    if (is_row_lut(w)) return lut_quantize_launch(st, w, src);
    if (w->type == KF_F8E5M2) {
        const size_t n = (size_t)w->ne0 * w->ne1;
        hipLaunchKernelGGL(to_f8e5m2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, (unsigned char*)const_cast<void*>(w->data), n);
        return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
    }
    int bits, mode = symmetric ? 1 : 0;
    switch (w->type) {
        case KF_Q4: bits = 4; break;
        case KF_T_SIGN: bits = 2, mode = 2; break;
        case KF_BOOL1: case KF_T_BINARY: bits = 1, mode = 2; break;
        default: return KF_UNSUPPORTED_DATATYPE;
    }
    const size_t n = (size_t)w->ne0 * w->ne1;
    if (!w->gama || w->lGroup <= 0 || n % w->lGroup || w->lGroup % (128 / bits)) return KF_QUANT_ERR;
    const size_t nGroup = n / w->lGroup;
    uint16_t* zero = const_cast<uint16_t*>(w->gama) + w->ne0 + w->ne1;
    uint16_t* step = zero + nGroup;
    if (bits == 4 && w->lGroup == 128 && mode != 2 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)w->data & 3) == 0) {
        hipLaunchKernelGGL(quantize4_g128_kernel, dim3((unsigned)((nGroup + 15) / 16)), dim3(256), 0, st, src, (unsigned char*)const_cast<void*>(w->data), zero, step, nGroup, mode,
                           w->qMin, w->qMax, w->qBias);
        return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
    }
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((nGroup + 3) / 4)), dim3(256), 0, st, src, (unsigned char*)const_cast<void*>(w->data), zero, step, nGroup,
                       w->lGroup, bits, mode, w->qMin, w->qMax, w->qBias);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---------------------------------------------------------------- AdamW update (CU_adamw_p, Optimizer.cu:393-442; training kernel path)
// Launched exactly as the reference's TASKA_1p1 (packedN.cuh:612-643): blocks of 512 threads, 8 bf16 parameters per thread (one 16-byte
// load / store per tensor per thread), because the stochastic rounding of the stores (CU_Float2T<bf16>, packedN.cuh:62-72) draws ONE
// 16-bit threshold per thread from SquirrelNoise5(threadIdx.x + PRIME * (blockIdx.x * blockDim.x), seed) (utils.cuh:296-326): with the
// same geometry the update is bit-identical to the oracle's restatement.  HBM-bound: 16 B per parameter with bf16 moments
// (p, g, m, v read and written), 24 B with fp32 moments.
__device__ __forceinline__ unsigned int squirrel5(unsigned int pos, unsigned int seed) {
    unsigned int b = pos;
    b *= 0xd2a80a3fu;
    b += seed;
    b ^= (b >> 9);
    b += 0xa884f197u;
    b ^= (b >> 11);
    b *= 0x6C736F4Bu;
    b ^= (b >> 13);
    b += 0xB79F3ABBu;
    b ^= (b >> 15);
    b *= 0x1b56c4f5u;
    b ^= (b >> 17);
    return b;
}
__device__ __forceinline__ uint16_t stochastic_bf16(float a, unsigned int threshold) {
    unsigned int u = __float_as_uint(a);
    u = ((u & 0xFFFFu) > threshold) ? (u | 0xFFFFu) : (u & ~0xFFFFu);
    return f2bf(__uint_as_float(u));
}
template <bool MV_BF16>
__global__ void __launch_bounds__(512) adamw_kernel(uint16_t* __restrict__ params, uint16_t* __restrict__ grads, void* __restrict__ gm_, void* __restrict__ gv_, size_t n,
                                                    float lr, float beta1, float beta2, float b1c, float b2c, float eps, float wd, float grad_scale, unsigned int seed,
                                                    int* __restrict__ status) {
    const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (idx >= n) return;
    const unsigned int thr = squirrel5(threadIdx.x + 198491317u * (blockIdx.x * blockDim.x), seed) & 0xFFFFu;
    const u32x4 P = *reinterpret_cast<const u32x4*>(params + idx), G = *reinterpret_cast<const u32x4*>(grads + idx);
    const uint32_t pw[4] = {P.x, P.y, P.z, P.w}, gw[4] = {G.x, G.y, G.z, G.w};
    float m[8], v[8], p[8];
    if (MV_BF16) {
        const u32x4 M = *reinterpret_cast<const u32x4*>(reinterpret_cast<uint16_t*>(gm_) + idx), V = *reinterpret_cast<const u32x4*>(reinterpret_cast<uint16_t*>(gv_) + idx);
        const uint32_t mw[4] = {M.x, M.y, M.z, M.w}, vw[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
        for (int i = 0; i < 4; i++) m[2 * i] = bf_lo(mw[i]), m[2 * i + 1] = bf_hi(mw[i]), v[2 * i] = bf_lo(vw[i]), v[2 * i + 1] = bf_hi(vw[i]);
    } else {
        const float* fm = reinterpret_cast<const float*>(gm_) + idx;
        const float* fv = reinterpret_cast<const float*>(gv_) + idx;
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = fm[i], v[i] = fv[i];
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float g = grad_scale * ((i & 1) ? bf_hi(gw[i >> 1]) : bf_lo(gw[i >> 1]));
        m[i] = fmaf(beta1, m[i], fmaf(-beta1, g, g));
        const float g2 = g * g;
        v[i] = fmaf(beta2, v[i], fmaf(-beta2, g2, g2));
        const float mh = m[i] / b1c, vh = v[i] / b2c; /* `/` and sqrtf are correctly rounded under hipcc's defaults (checked against the host over
                                                         2^24 random operands); __fsqrt_rn is NOT (1 ulp low on 0x397d20ce) */
        const float step = mh / (sqrtf(vh) + eps);
        const float old = (i & 1) ? bf_hi(pw[i >> 1]) : bf_lo(pw[i >> 1]);
        bad = bad || !isfinite(old) || !isfinite(step);
        p[i] = old - lr * wd * old - lr * step;
    }
    if (bad) { /* the reference thread returns before any store and raises KOIFISH_ADAMW_MV through its prober */
        if (status) *status = KF_ADAMW_MV;
        return;
    }
    uint32_t po[4];
#pragma unroll
    for (int i = 0; i < 4; i++) po[i] = (uint32_t)stochastic_bf16(p[2 * i], thr) | ((uint32_t)stochastic_bf16(p[2 * i + 1], thr) << 16);
    *reinterpret_cast<u32x4*>(params + idx) = u32x4{po[0], po[1], po[2], po[3]};
    *reinterpret_cast<u32x4*>(grads + idx) = u32x4{0, 0, 0, 0};
    if (MV_BF16) {
        uint32_t mo[4], vo[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            mo[i] = (uint32_t)stochastic_bf16(m[2 * i], thr) | ((uint32_t)stochastic_bf16(m[2 * i + 1], thr) << 16);
            vo[i] = (uint32_t)stochastic_bf16(v[2 * i], thr) | ((uint32_t)stochastic_bf16(v[2 * i + 1], thr) << 16);
        }
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(gm_) + idx) = u32x4{mo[0], mo[1], mo[2], mo[3]};
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(gv_) + idx) = u32x4{vo[0], vo[1], vo[2], vo[3]};
    } else {
        float* fm = reinterpret_cast<float*>(gm_) + idx;
        float* fv = reinterpret_cast<float*>(gv_) + idx;
#pragma unroll
        for (int i = 0; i < 8; i++) fm[i] = m[i], fv[i] = v[i];
    }
}
int adamw_launch(hipStream_t st, uint16_t* params, uint16_t* grads, void* gm, void* gv, size_t n, int mv_bf16, float lr, float beta1, float beta2, float b1c,
                 float b2c, float eps, float wd, float grad_scale, unsigned int seed, int* status) {
    if (n == 0 || n % 8) return KF_INVALID_ARGS;
    const size_t nthread = n / 8;
    const unsigned blocks = (unsigned)((nthread + 511) / 512);
    if (mv_bf16)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3(blocks), dim3(512), 0, st, params, grads, gm, gv, n, lr, beta1, beta2, b1c, b2c, eps, wd, grad_scale, seed, status);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3(blocks), dim3(512), 0, st, params, grads, gm, gv, n, lr, beta1, beta2, b1c, b2c, eps, wd, grad_scale, seed, status);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// Tensor-parallel combine: out = bf16(residual + bf16(sum_r partial[r])) with the partial sums added in rank order 0..R-1
// (fixed order: every rank computes the same bits).  residual may be NULL.
__global__ void tp_reduce_kernel(const float* __restrict__ partials, int R, int n, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float tot = partials[i];
    for (int r = 1; r < R; r++) tot = tot + partials[(size_t)r * n + i];
    uint16_t o = f2bf(tot);
    if (residual) o = f2bf(bf2f(residual[i]) + bf2f(o));
    out[i] = o;
}
int tp_reduce_launch(hipStream_t st, const float* partials, int R, int n, const uint16_t* residual, uint16_t* out) {
    hipLaunchKernelGGL(tp_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, partials, R, n, residual, out);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

__global__ void set_state_kernel(int32_t* st, int token, int pos) { st[0] = token, st[1] = pos; }
int set_state_launch(hipStream_t st, int32_t* d_state, int token, int pos) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, d_state, token, pos);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// ---- hot-row lists for the sparse forward (CS_Picker's `hot` array, SparseNeuron.cpp:20-29; D_matmul_sparse, GST_float.cpp:306-318)
// rows[0 .. count) = the indices i with hot[i] == 1, ascending; one workgroup walks the mask in chunks of 1024 with a ballot-ordered compaction
__global__ void __launch_bounds__(1024) hot_rows_kernel(const int32_t* hot, int n, int32_t* rows, int32_t* count) {
    __shared__ int wsum[16], base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + tid;
        const bool h = i < n && hot[i] == 1;
        const unsigned long long b = __ballot(h);
        if (lane == 0) wsum[wave] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++) off += wsum[w];
        if (h) rows[off + __popcll(b & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; w++) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) *count = base;
}
int hot_rows_launch(hipStream_t st, const int32_t* hot, int n, int32_t* rows, int32_t* count) {
    hipLaunchKernelGGL(hot_rows_kernel, dim3(1), dim3(1024), 0, st, hot, n, rows, count);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}
// cold rows of a masked product: y[i] = bf16(0 + bias[i]) (val = 0, then the bias: D_matmul_sparse)
__global__ void cold_fill_kernel(uint16_t* y, const uint16_t* bias, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = bias ? f2bf(0.0f + bf2f(bias[i])) : (uint16_t)0;
}
int cold_fill_launch(hipStream_t st, uint16_t* y, const uint16_t* bias, int n) {
    hipLaunchKernelGGL(cold_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, y, bias, n);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
