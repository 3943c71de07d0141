// kf_embed_bwd.hip -- gradient of the token / position embedding (encoder_backward, src/Device/CUDA/kernel/embed.cuh:380-470:
// wte_backward_kernel :257-331 over host-built buckets, wpe_backward_kernel :333-366).
//   dwpe[t][c] = bf16(sum_b dout[b][t][c] + dwpe[t][c])                      (b ascending, fp32: exactly the reference's loop)
//   dwte[v][c] = bf16(sum_{bt : tokens[bt] == v} dout[bt][c] + dwte[v][c])    (bt ascending, fp32)
// The reference sorts the (token, position) pairs into buckets on the host for every batch; here everything stays on the device and
// deterministic: one workgroup per position bt; it is the *leader* of its token when no earlier position holds the same token (a cooperative
// scan of tokens[0..bt)), and a leader walks the later positions in chunks of 256 -- one candidate per thread, matches compacted in order with
// wave ballots -- adding the matching rows in ascending order.  HBM traffic: dout once for wte, once for wpe.  Stores are round-to-nearest
// (the reference rounds stochastically, seed + element index).
#include "kf_kernels.h"

namespace kf {

// Forward of the same pair (encoder_forward_kernel3, embed.cuh:20-45: out[bt] = wte[tokens[bt]] + wpe[t], fp32 sum, round to nearest): one 8-column vector per thread.
// A token id outside [0, V) reads row 0 (the ABI has checked nothing on the device ids; the loss of such a row is the caller's problem, never a fault).
__global__ void __launch_bounds__(256) embed_pos_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ wte, long long ldw, const uint16_t* __restrict__ wpe,
                                                        const int* __restrict__ tokens, int N, int T, int C, int V) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int cv = C / 8;
    if (v >= (size_t)N * cv) return;
    const int bt = (int)(v / cv), c8 = (int)(v % cv) * 8;
    int id = tokens[bt];
    id = (id < 0 || id >= V) ? 0 : id;
    const u32x4 a = *reinterpret_cast<const u32x4*>(wte + (size_t)id * ldw + c8);
    const u32x4 b = *reinterpret_cast<const u32x4*>(wpe + (size_t)(bt % T) * C + c8);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = pack_bf16x2(bf_lo(aw[k]) + bf_lo(bw[k]), bf_hi(aw[k]) + bf_hi(bw[k]));
    *reinterpret_cast<u32x4*>(out + (size_t)bt * C + c8) = u32x4{r[0], r[1], r[2], r[3]};
}

int embed_pos_launch(hipStream_t st, const uint16_t* wte, long long ldw, const uint16_t* wpe, const int* tokens, int B, int T, int C, int V, uint16_t* out) {
    if (B < 1 || T < 1 || V < 1 || C < 8 || (C & 7) || ldw < C || (ldw & 7)) return KF_INVALID_ARGS;
    const size_t nv = (size_t)B * T * (C / 8);
    hipLaunchKernelGGL(embed_pos_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, out, wte, ldw, wpe, tokens, B * T, T, C, V);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

constexpr int EB_MAXV = 4; /* 8-column vectors per thread: C <= 8192 */

__global__ void __launch_bounds__(256) wpe_backward_kernel(uint16_t* __restrict__ dwpe, const uint16_t* __restrict__ dout, int B, int T, int C) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; /* 8-element vector index inside [T, C] */
    if (v * 8 >= (size_t)T * C) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; b++) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(dout + (size_t)b * T * C + v * 8);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) acc[2 * k] += bf_lo(w[k]), acc[2 * k + 1] += bf_hi(w[k]);
    }
    const u32x4 o = *reinterpret_cast<const u32x4*>(dwpe + v * 8);
    const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = pack_bf16x2(acc[2 * k] + bf_lo(ow[k]), acc[2 * k + 1] + bf_hi(ow[k]));
    *reinterpret_cast<u32x4*>(dwpe + v * 8) = u32x4{r[0], r[1], r[2], r[3]};
}

template <int NV>
__global__ void __launch_bounds__(256) wte_backward_kernel(uint16_t* __restrict__ dwte, long long ldw, const uint16_t* __restrict__ dout, const int* __restrict__ tokens,
                                                           int N, int C, int V) {
    __shared__ int found;
    __shared__ int match[256];
    __shared__ int wave_cnt[4];
    const int bt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = tokens[bt];
    if (tok < 0 || tok >= V) return; /* masked / padding position */
    if (tid == 0) found = 0;
    __syncthreads();
    int seen = 0;
    for (int i = tid; i < bt; i += 256) seen |= (tokens[i] == tok);
    if (seen) found = 1; /* benign race: every writer stores 1 */
    __syncthreads();
    if (found) return; /* an earlier position leads this token */

    const int nvec = C >> 3;
    bool has[NV];
    float acc[NV][8];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        has[j] = tid + 256 * j < nvec;
#pragma unroll
        for (int k = 0; k < 8; k++) acc[j][k] = 0.f;
    }
    for (int base = bt; base < N; base += 256) {
        const int i = base + tid;
        const bool m = i < N && tokens[i] == tok;
        const unsigned long long bal = __ballot(m);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int w2 = 0; w2 < wave; w2++) off += wave_cnt[w2];
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (m) match[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        for (int q = 0; q < total; q++) { /* ascending positions */
            const size_t row = (size_t)match[q] * C;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                if (!has[j]) continue;
                const u32x4 d = *reinterpret_cast<const u32x4*>(dout + row + (size_t)(tid + 256 * j) * 8);
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int k = 0; k < 4; k++) acc[j][2 * k] += bf_lo(w[k]), acc[j][2 * k + 1] += bf_hi(w[k]);
            }
        }
        __syncthreads(); /* match[] and wave_cnt[] are rewritten by the next chunk */
    }
    uint16_t* dst = dwte + (size_t)tok * ldw;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        if (!has[j]) continue;
        const u32x4 o = *reinterpret_cast<const u32x4*>(dst + (size_t)(tid + 256 * j) * 8);
        const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = pack_bf16x2(acc[j][2 * k] + bf_lo(ow[k]), acc[j][2 * k + 1] + bf_hi(ow[k]));
        *reinterpret_cast<u32x4*>(dst + (size_t)(tid + 256 * j) * 8) = u32x4{r[0], r[1], r[2], r[3]};
    }
}

int embed_backward_launch(hipStream_t st, uint16_t* dwte, long long ldw, uint16_t* dwpe, const uint16_t* dout, const int* tokens, int B, int T, int C, int V) {
    if (B < 1 || T < 1 || C < 8 || (C % 8) != 0 || C > EB_MAXV * 2048 || V < 1 || ldw < C || (ldw % 8) != 0) return KF_INVALID_ARGS;
    const int N = B * T;
    if (dwpe) hipLaunchKernelGGL(wpe_backward_kernel, dim3((unsigned)(((size_t)T * C / 8 + 255) / 256)), dim3(256), 0, st, dwpe, dout, B, T, C);
    if (dwte) {
        switch ((C / 8 + 255) / 256) {
            case 1: hipLaunchKernelGGL((wte_backward_kernel<1>), dim3(N), dim3(256), 0, st, dwte, ldw, dout, tokens, N, C, V); break;
            case 2: hipLaunchKernelGGL((wte_backward_kernel<2>), dim3(N), dim3(256), 0, st, dwte, ldw, dout, tokens, N, C, V); break;
            case 3: hipLaunchKernelGGL((wte_backward_kernel<3>), dim3(N), dim3(256), 0, st, dwte, ldw, dout, tokens, N, C, V); break;
            default: hipLaunchKernelGGL((wte_backward_kernel<4>), dim3(N), dim3(256), 0, st, dwte, ldw, dout, tokens, N, C, V); break;
        }
    }
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
