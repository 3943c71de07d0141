// kf_tp.hip -- the exchange step of tensor-parallel decode (Qwen3-32B over 8 MI355X): one-shot writes into every peer's
// receive area over xGMI peer-to-peer mappings, then a local rank-ordered sum.  No collective library call, no host involvement: the whole TP
// step is a chain of kernels that a hipGraph replays.
//
// Protocol ("the data is the flag"): a receive area holds 8-byte granules {fp32 value | 32-bit tag}, one per (exchange buffer, source rank, row).
// A producer writes its row into its own slot of EVERY rank's area with ONE naturally aligned 8-byte system-scope store per granule; a consumer
// re-reads a granule (system-scope load) until the tag equals the tag of the exchange it is waiting for.  Tags never repeat:
//     tag = generation * per_step + index + 1,    generation = the device word kf_tp_pick advances once per token,
// so no buffer is ever reset, and two vector buffers suffice (attention exchanges use buffer 0, FFN exchanges buffer 1: a rank can run at most
// one exchange ahead of the slowest one, because finishing exchange k needs every rank's push of k, which a rank issues only after it has read
// exchange k - 1).  A poll that runs out of spins sets the error word instead of hanging the queue.
#include "kf_kernels.h"

namespace kf {

static constexpr int TP_SPINS = 1 << 24;

__device__ __forceinline__ unsigned long long tp_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// out[i] = bf16(residual[i] + bf16(sum over ranks 0..R-1 of the fp32 partials of row i)) -- the arithmetic of tp_reduce_kernel (kf_ops.hip)
__global__ void __launch_bounds__(256) tp_reduce_recv_kernel(const unsigned long long* __restrict__ slots, int R, int n_max, int n, const unsigned* __restrict__ d_step,
                                                             unsigned per_step, unsigned index, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out,
                                                             int* __restrict__ d_err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned tag = *d_step * per_step + index + 1u;
    float tot = 0.f;
    for (int r = 0; r < R; r++) {
        const unsigned long long* g = slots + (size_t)r * n_max + i;
        unsigned long long v = tp_load(g);
        for (int spin = 0; (unsigned)(v >> 32) != tag; spin++) {
            if (spin > TP_SPINS) {
                atomicExch(d_err, 1 + r);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            v = tp_load(g);
        }
        const float p = __uint_as_float((unsigned)v);
        tot = r == 0 ? p : tot + p;
    }
    uint16_t o = f2bf(tot);
    if (residual) o = f2bf(bf2f(residual[i]) + bf2f(o));
    out[i] = o;
}

int tp_reduce_recv_launch(hipStream_t st, const unsigned long long* slots, int R, int n_max, int n, const unsigned* d_step, unsigned per_step, unsigned index,
                          const uint16_t* residual, uint16_t* out, int* d_err) {
    hipLaunchKernelGGL(tp_reduce_recv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, slots, R, n_max, n, d_step, per_step, index, residual, out, d_err);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// After the local LM-head shard: first maximum over the workgroup partials (value, LOCAL row), made global by row0, pushed to every rank as two
// granules {value}, {index}.
struct TpPeers {
    unsigned long long* p[8];
};
__global__ void __launch_bounds__(256) tp_argmax_push_kernel(const float* __restrict__ val, const int* __restrict__ idx, int n, int row0, TpPeers peers, int R,
                                                             const unsigned* __restrict__ d_step, unsigned per_step, unsigned index) {
    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float w = val[i];
        const int jx = idx[i];
        if (w > bv || (w == bv && jx < bi)) bv = w, bi = jx;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const float ov = __shfl_xor(bv, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = bv, si[threadIdx.x >> 6] = bi;
    __syncthreads();
    if (threadIdx.x < R) {
        for (int w = 0; w < 4; w++)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) bv = sv[w], bi = si[w];
        const unsigned long long tag = (unsigned long long)(*d_step * per_step + index + 1u) << 32;
        unsigned long long* dst = peers.p[threadIdx.x];
        __hip_atomic_store(dst, tag | __float_as_uint(bv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, tag | (unsigned)(bi + row0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int tp_argmax_push_launch(hipStream_t st, const float* val, const int* idx, int n, int row0, unsigned long long* const* peers, int R, const unsigned* d_step,
                          unsigned per_step, unsigned index) {
    TpPeers P;
    for (int r = 0; r < 8; r++) P.p[r] = r < R ? peers[r] : nullptr;
    hipLaunchKernelGGL(tp_argmax_push_kernel, dim3(1), dim3(256), 0, st, val, idx, n, row0, P, R, d_step, per_step, index);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// (max, index) pairs of all ranks -> first maximum (lowest index among equals; sample_argmax, GoPT.cpp:602-612), the decode-state update of
// argmax_finish_kernel, and the generation word moves on.
__global__ void tp_pick_kernel(const unsigned long long* __restrict__ pairs, int R, unsigned* __restrict__ d_step, unsigned per_step, unsigned index,
                               int32_t* __restrict__ d_state, int32_t* __restrict__ d_tokens_out, int* __restrict__ d_err, int vocab) {
    if (threadIdx.x != 0) return;
    const unsigned tag = *d_step * per_step + index + 1u;
    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int r = 0; r < R; r++) {
        unsigned long long g[2];
        for (int k = 0; k < 2; k++) {
            g[k] = tp_load(pairs + 2 * r + k);
            for (int spin = 0; (unsigned)(g[k] >> 32) != tag; spin++) {
                if (spin > TP_SPINS) {
                    atomicExch(d_err, 101 + r);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                g[k] = tp_load(pairs + 2 * r + k);
            }
        }
        const float v = __uint_as_float((unsigned)g[0]);
        const int i = (int)(unsigned)g[1];
        if (v > bv || (v == bv && i < bi)) bv = v, bi = i;
    }
    // A timed-out exchange (this poll or an earlier one of the step: d_err is set) leaves stale pairs: the decode state is NOT advanced with them -- the id would
    // feed the next step's embedding lookup -- and an index outside the vocabulary is refused the same way.  The generation word still advances, so that every
    // rank keeps forming the same tags; the host sees the error word at its next check (kfh_tp_check) and the ids behind it are void.
    const int err = __hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (err == 0 && bi >= 0 && bi < vocab) {
        const int p = d_state[1];
        if (d_tokens_out) d_tokens_out[p] = bi;
        d_state[0] = bi;
        d_state[1] = p + 1;
        d_state[2] = bi;
    } else if (err == 0) {
        atomicExch(d_err, 201);
    }
    *d_step = *d_step + 1u;
}
int tp_pick_launch(hipStream_t st, const unsigned long long* pairs, int R, unsigned* d_step, unsigned per_step, unsigned index, int32_t* d_state, int32_t* d_tokens_out,
                   int* d_err, int vocab) {
    hipLaunchKernelGGL(tp_pick_kernel, dim3(1), dim3(64), 0, st, pairs, R, d_step, per_step, index, d_state, d_tokens_out, d_err, vocab);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
