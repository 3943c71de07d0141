// kf_linear_bwd.hip -- helpers of the linear layer's backward pass (SLP::Back, src/Device/CUDA/NeuronFuse.cu:495-547):
//   delta  [n, IC] (=|+=) deltaIn [n, OC] . W [OC, IC]          (TASKA_AxB(nIn, B*T, nOut).blasLt(delta, wX, deltaIn), wX = w->GetDataX())
//   gW     [OC, IC] +=     deltaIn^T [OC, n] . inp [n, IC]       (TASKA_AxB(nIn, nOut, B*T, ..., beta 1).blasLt(ToG(w), inp, deltaIn))
//   dbias  [OC]     +=     column sums of deltaIn                (matmul_backward_bias_kernel9 + reduce_add_sum_kernel)
// Like the reference, the weight is first dequantised to bf16 (GetDataX); the three contractions then run on the token-batch MFMA kernels of
// kf_gemm.hip / kf_gemm2.hip, which contract over the contiguous index of both operands -- so the operands are brought into that form by the
// tiled bf16 transpose below: delta = gemm(W^T as [IC, OC] rows, deltaIn), gW = gemm(inp^T as [IC, n] rows, deltaIn^T [OC, n]) with beta = 1.
// This file: the transpose, and the bias column sums (fp64 per 256-row slab, slabs in index order: oracle/kf_oracle.c kfo_colsum_add).
#include "kf_kernels.h"

namespace kf {

// out[c][r] = in[r][c] for bf16 [R, C] -> [C, R]; 64 x 64 tiles through LDS (rows padded to 66 halfwords: conflict-free column reads)
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int C) {
    __shared__ uint16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        if (r < R && c < C) tile[ty * 16 + i][tx] = in[(size_t)r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (r < R && c < C) out[(size_t)c * R + r] = tile[tx][ty * 16 + i];
    }
}
int transpose_bf16_launch(hipStream_t st, const uint16_t* in, uint16_t* out, int R, int C) {
    if (R < 1 || C < 1) return KF_INVALID_ARGS;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, st, in, out, R, C);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// column sums of x [n, C]: slab s = rows 256 s .. 256 s + 255 summed in row order in fp64 -> part[s][C]; then the slabs in index order and
// bf16(fp32 sum + old) in a second launch
// thread = (8-column vector cv, row group g): rows 32 g .. 32 g + 31 of the slab with 16-byte loads, then the 8 row groups in LDS.  fp64 sums of
// bf16 values are exact (8 significant bits each, far fewer than 2^44 of them), so the grouping changes no bit of the slab's sum.
__global__ void __launch_bounds__(256) colsum_part_kernel(const uint16_t* __restrict__ x, double* __restrict__ part, int n, int C) {
    __shared__ double red[8][256];
    const int cv = threadIdx.x & 31, g = threadIdx.x >> 5, s = blockIdx.y;
    const int c0 = blockIdx.x * 256 + cv * 8;
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (c0 < C) { /* C is a multiple of 8: the vector is inside the row */
        const int r0 = s * 256 + g * 32;
        int r1 = r0 + 32;
        r1 = r1 < n ? r1 : n;
        for (int r = r0; r < r1; r++) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(x + (size_t)r * C + c0);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) acc[2 * k] += (double)bf_lo(w[k]), acc[2 * k + 1] += (double)bf_hi(w[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) red[g][cv * 8 + k] = acc[k];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < 8; gg++) t += red[gg][threadIdx.x];
        part[(size_t)s * C + c] = t;
    }
}
// any C (scalar loads): used when C is not a multiple of 8 or x is not 16-byte aligned
__global__ void __launch_bounds__(256) colsum_part_scalar_kernel(const uint16_t* __restrict__ x, double* __restrict__ part, int n, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (c >= C) return;
    const int r1 = (s + 1) * 256 < n ? (s + 1) * 256 : n;
    double acc = 0.0;
    for (int r = s * 256; r < r1; r++) acc += (double)bf2f(x[(size_t)r * C + c]);
    part[(size_t)s * C + c] = acc;
}
__global__ void __launch_bounds__(256) colsum_finish_kernel(uint16_t* __restrict__ dst, const double* __restrict__ part, int nslab, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double acc = 0.0;
    int s = 0;
    for (; s + 8 <= nslab; s += 8) { /* slab order, eight loads in flight (one dependent load at a time was 9 us for 32 slabs) */
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = part[(size_t)(s + u) * C + c];
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    for (; s < nslab; s++) acc += part[(size_t)s * C + c];
    dst[c] = f2bf((float)acc + bf2f(dst[c]));
}
int colsum_add_launch(hipStream_t st, const uint16_t* x, uint16_t* dst, int n, int C, double* scratch) {
    if (n < 1 || C < 1) return KF_INVALID_ARGS;
    const int nslab = (n + 255) / 256;
    if ((C % 8) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL(colsum_part_kernel, dim3((C + 255) / 256, nslab), dim3(256), 0, st, x, scratch, n, C);
    else
        hipLaunchKernelGGL(colsum_part_scalar_kernel, dim3((C + 255) / 256, nslab), dim3(256), 0, st, x, scratch, n, C);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, dst, scratch, nslab, C);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
