// kf_gemm3.hip -- large-batch bf16 GEMM on 256 x 256 x 64 tiles: y[n, M] (+)= x[n, K] . W[M, K]^T with both operands K-contiguous (SLP::Forw's product
// for a dequantised or bf16 weight, and -- with the transposed copies kf_linear_backward already makes -- both GEMMs of SLP::Back).  This is the tile
// the LDS accounting of DESIGN.md section 8 asked for, built MI355X-first:
//   * both operand tiles go global -> LDS with global_load_lds_dwordx4 (16 bytes per lane, no VGPR round trip).  The LDS image of such a load is
//     lane-linear, so the bank-conflict-free layout is produced on the SOURCE side: a tile row is 128 bytes = 8 chunks of 16 bytes, chunk c of row r
//     is stored at position c ^ ((r >> 1) & 7); a fragment read (16 rows x one chunk, ds_read_b128) then touches all 64 banks exactly once;
//   * two LDS buffers (4 x 32 KiB): the loads of k-tile t+1 are issued before k-tile t is multiplied and drained with a COUNTED s_waitcnt vmcnt(8)
//     and raw s_barrier (a __syncthreads() would drain the loads in flight);
//   * 8 waves as 2 (rows) x 4 (tokens), 128 x 64 outputs per wave = 32 accumulator tiles of v_mfma_f32_16x16x32_bf16, 64 MFMAs per wave and k-tile;
//   * workgroup order remapped so that the blocks an XCD runs back to back share their W row-tile in that XCD's L2.
// Measured on random operands (scratch/ub_gemm3.hip): 4096^3 997 TFLOP/s, 8192^3 1029, GPT2-1558M shapes (8192 tokens) 726-843, head 50304 x 1600 831.
// The epilogue is gemm_epilogue's (alpha, beta, bias, one bf16 store, residual added to the rounded value).
#include "kf_gemm_common.h"

namespace kf {

constexpr int G3_BM = 256, G3_BN = 256, G3_BK = 64;
constexpr int G3_TILE = G3_BM * G3_BK * 2; /* 32 KiB per operand tile */

// one operand tile (256 rows x 128 B) = 32 wave instructions of 1 KiB (8 rows each); wave `wid` issues instructions 4 wid .. 4 wid + 3
__device__ __forceinline__ void g3_stage(const uint16_t* __restrict__ src, long long ld, int row0, int nrows, int k0, unsigned char* lds_tile, int wid, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = wid * 4 + i;
        const int r = j * 8 + (lane >> 3), p = lane & 7, c = p ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1; /* rows past the end re-read the last row: their outputs are not stored */
        const uint16_t* g = src + (size_t)gr * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(lds_tile + j * 1024), 16, 0, 0);
    }
}

__global__ void __launch_bounds__(512) gemm3_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int nbx = (a.M + G3_BM - 1) / G3_BM, nby = (a.n + G3_BN - 1) / G3_BN, nwg = nbx * nby;
    // bijective XCD remap: the blocks with equal blockIdx % 8 (one XCD under round-robin placement) get consecutive tiles
    const int orig = blockIdx.x, q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    const int bx = wg % nbx, by = wg / nbx;
    const int m0 = bx * G3_BM, t0 = by * G3_BN;
    const int nkt = a.K / G3_BK;
    const uint16_t* const W = reinterpret_cast<const uint16_t*>(a.w);
    auto bufA = [&](int b) { return smem_raw + (size_t)b * 2 * G3_TILE; };
    auto bufB = [&](int b) { return smem_raw + (size_t)b * 2 * G3_TILE + G3_TILE; };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    g3_stage(W, a.K, m0, a.M, 0, bufA(0), wid, lane);
    g3_stage(a.x, a.ldx, t0, a.n, 0, bufB(0), wid, lane);
    const int r16 = lane & 15, q4 = lane >> 4;
    for (int kt = 0; kt < nkt; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
            g3_stage(W, a.K, m0, a.M, (kt + 1) * G3_BK, bufA(cur ^ 1), wid, lane);
            g3_stage(a.x, a.ldx, t0, a.n, (kt + 1) * G3_BK, bufB(cur ^ 1), wid, lane);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); /* k-tile kt has landed; the 8 loads of k-tile kt+1 stay in flight */
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const int row = wn * 64 + nt * 16 + r16, c = kk * 4 + q4;
                bfr[nt] = *reinterpret_cast<const bf16x8*>(bufB(cur) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < 8; mt++) {
                const int row = wm * 128 + mt * 16 + r16, c = kk * 4 + q4;
                af[mt] = *reinterpret_cast<const bf16x8*>(bufA(cur) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < 8; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); /* every wave is done reading buffer `cur`: the next iteration's loads may overwrite it */
    }
    // epilogue (gemm_epilogue's order): lane holds rows m .. m+3 of a 16 x 16 tile for token column r16
    const bool vec_ok = ((a.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 7) == 0);
#pragma unroll
    for (int mt = 0; mt < 8; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int tok = t0 + wn * 64 + nt * 16 + r16, m = m0 + wm * 128 + mt * 16 + 4 * q4;
            if (tok >= a.n || m >= a.M) continue;
            uint16_t* yp = a.y + (size_t)tok * a.ldy + m;
            const float vv[4] = {acc[mt][nt].x, acc[mt][nt].y, acc[mt][nt].z, acc[mt][nt].w};
            uint16_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = vv[j];
                o[j] = 0;
                if (m + j < a.M) {
                    if (a.alpha != 1.0f) v = a.alpha * v;
                    if (a.beta != 0.0f) v = v + a.beta * bf2f(yp[j]);
                    if (a.bias) v = v + bf2f(a.bias[m + j]);
                    uint16_t qv = f2bf(v);
                    if (a.residual) qv = f2bf(bf2f(a.residual[(size_t)tok * a.ldr + m + j]) + bf2f(qv));
                    o[j] = qv;
                }
            }
            if (vec_ok && m + 3 < a.M) {
                *reinterpret_cast<u32x2*>(yp) = u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (m + j < a.M) yp[j] = o[j];
            }
        }
}

// KF_OK launched, 1 = not for this kernel (the caller's other tile kernels take the shape), < 0 error.  bf16 "weights" only: quantised ones are dequantised first.
int gemm3_launch(hipStream_t st, int fmt, const GemmArgs& a) {
    if (fmt != FMT_BF16 || a.K % G3_BK != 0 || a.K < G3_BK || a.n < G3_BN || a.M < G3_BM) return 1;
    if ((a.ldx & 7) != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15) != 0 || (reinterpret_cast<uintptr_t>(a.w) & 15) != 0 || (a.K & 7) != 0) return 1;
    const long nwg = (long)((a.M + G3_BM - 1) / G3_BM) * ((a.n + G3_BN - 1) / G3_BN);
    if (nwg < 128) return 1; /* fewer tiles than half the CUs: the 128-row tile kernels fill the chip better */
    static int attr_set = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G3_TILE) != hipSuccess) return KF_HIP_CHECK;
        attr_set = 1;
    }
    hipLaunchKernelGGL(gemm3_kernel, dim3((unsigned)nwg), dim3(512), 4 * G3_TILE, st, a);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
