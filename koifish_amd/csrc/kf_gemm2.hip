// kf_gemm2.hip -- token-batch GEMM for LARGE batches: LDS-staged dequant tiles, producer / consumer waves, gfx950 / wave64.
//
// Same contract as kf_gemm.hip (y[n, M] = x[n, K] . W[M, K]^T, fp32 accumulate in the MFMA, one bf16 store, the mat-vec's epilogue
// order); used when a batch has at least 256 token rows and the grid fills the chip.  What the PMC passes on the first tile kernel
// showed (58 % of the wave cycles parked on waits, MFMA busy 18 %, one wave per SIMD) is answered structurally:
//
//   * 8 waves per workgroup = 2 per SIMD, specialised.  Waves 0..3 PRODUCE: they stream the packed weight tile (coalesced: consecutive
//     lanes read consecutive 16-byte blocks of a row) and the x tile, unpack every weight ONCE per workgroup with the reference's
//     bf16-stepwise arithmetic (T.cu:274) and lay both operands down in LDS as plain bf16 rows.  Waves 4..7 CONSUME: nothing but
//     ds_read_b128 fragments and v_mfma_f32_32x32x16_bf16, 64 rows x 128 tokens each.  A producer and a consumer share each SIMD, so the
//     unpack VALU work and the matrix pipe run side by side instead of taking turns inside one wave.
//   * Tile 128 rows x 256 tokens x 64 k, two LDS stages (108 KiB), one workgroup barrier per k tile.  Per k tile a consumer issues
//     32 MFMAs (1024 cycles of matrix pipe) against 24 fragment reads; a producer unpacks 32 weights (~180 VALU instructions) and moves
//     8 x-chunks: the matrix pipe is the longer side.
//   * LDS rows of 64 bf16 + 8 pad (144 B): the 32 lanes of a fragment read hit 64 distinct banks.
// bf16 / f8e5m2 / 4-bit weights (2-bit and 1-bit blocks are longer than a 64-element k tile: they stay on kf_gemm.hip).
#include <stdlib.h>

#include "kf_gemm_common.h"

namespace kf {

constexpr int G2_BM = 128, G2_BN = 256, G2_BK = 64;
constexpr int G2_LS = G2_BK + 8; /* LDS row, bf16 elements */
constexpr size_t G2_STAGE = (size_t)(G2_BM + G2_BN) * G2_LS * sizeof(uint16_t);

template <int FMT>
__global__ void __launch_bounds__(512) gemm2_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * G2_BM, tok0 = blockIdx.y * G2_BN;
    const int nkt = a.K / G2_BK;
    auto stageA = [&](int st) { return reinterpret_cast<uint16_t*>(smem_raw + (size_t)st * G2_STAGE); };
    auto stageX = [&](int st) { return reinterpret_cast<uint16_t*>(smem_raw + (size_t)st * G2_STAGE) + G2_BM * G2_LS; };

    if (wave < 4) {
        // ------------------------------------------------------------------ producers (256 threads)
        const int p = tid; /* 0..255 */
        // weights: thread p owns row (p >> 1), half (p & 1) of the 64-element k tile = 32 consecutive elements
        int wrow = row0 + (p >> 1);
        if (wrow >= a.M) wrow = a.M - 1;
        const int whalf = p & 1;
        // x: chunk ids p + 256 i, i < 8: token (id >> 3), 16-byte chunk (id & 7)
        // PF tiles in flight per thread: a tile's loads are issued PF k-tiles (~PF x 0.5 us of consumer work) before they are unpacked --
        // one tile ahead left every k tile waiting out a full memory latency (measured: 2.1 us per k tile instead of ~0.5)
        constexpr int PF = 4;
        constexpr int NWR = FMT == FMT_BF16 ? 4 : (FMT == FMT_F8 ? 2 : 1);
        u32x4 wreg[PF][NWR];
        uint16_t wst[PF], wze[PF]; /* raw bf16 bits: converting here would make the thread wait for the loads it has just issued */
        u32x4 xreg[PF][8];
        auto issue = [&](int kt, int q) { /* tile kt (clamped to the last one) -> register set q */
            if (kt >= nkt) kt = nkt - 1;
            const int e0 = kt * G2_BK + whalf * 32; /* first element of this thread's 32 */
            if constexpr (FMT == FMT_Q4) {
                const uint32_t bidx = (uint32_t)wrow * (uint32_t)a.nBlk + (uint32_t)(e0 >> 5);
                wreg[q][0] = ld_nt(reinterpret_cast<const u32x4*>(a.w) + bidx);
                const uint32_t gi = bidx >> a.gshift;
                wst[q] = a.step[gi], wze[q] = a.zero[gi];
            } else if constexpr (FMT == FMT_BF16) {
                const u32x4* src = reinterpret_cast<const u32x4*>(a.w + ((size_t)wrow * a.K + e0) * 2);
#pragma unroll
                for (int c = 0; c < 4; c++) wreg[q][c] = ld_nt(src + c);
            } else { /* f8e5m2: 32 bytes */
                const u32x4* src = reinterpret_cast<const u32x4*>(a.w + (size_t)wrow * a.K + e0);
                wreg[q][0] = ld_nt(src), wreg[q][1] = ld_nt(src + 1);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int id = p + 256 * i;
                int tok = tok0 + (id >> 3);
                if (tok >= a.n) tok = a.n - 1;
                xreg[q][i] = *reinterpret_cast<const u32x4*>(a.x + (size_t)tok * a.ldx + (size_t)kt * G2_BK + (id & 7) * 8);
            }
        };
        auto lay = [&](int st, int q) {
            uint16_t* A = stageA(st) + (p >> 1) * G2_LS + whalf * 32;
            if constexpr (FMT == FMT_Q4) {
                const float s = bf2f(wst[q]), s16 = s * 0.0625f, nb = -a.qBias * s, z = bf2f(wze[q]);
                *reinterpret_cast<u32x4*>(A + 0) = frag_q4(wreg[q][0].w, s, s16, nb, z);
                *reinterpret_cast<u32x4*>(A + 8) = frag_q4(wreg[q][0].z, s, s16, nb, z);
                *reinterpret_cast<u32x4*>(A + 16) = frag_q4(wreg[q][0].y, s, s16, nb, z);
                *reinterpret_cast<u32x4*>(A + 24) = frag_q4(wreg[q][0].x, s, s16, nb, z);
            } else if constexpr (FMT == FMT_BF16) {
#pragma unroll
                for (int c = 0; c < 4; c++) *reinterpret_cast<u32x4*>(A + 8 * c) = wreg[q][c];
            } else {
                const u32x4 v0 = wreg[q][0], v1 = wreg[q][1];
                *reinterpret_cast<u32x4*>(A + 0) = frag_f8(v0.x, v0.y);
                *reinterpret_cast<u32x4*>(A + 8) = frag_f8(v0.z, v0.w);
                *reinterpret_cast<u32x4*>(A + 16) = frag_f8(v1.x, v1.y);
                *reinterpret_cast<u32x4*>(A + 24) = frag_f8(v1.z, v1.w);
            }
            uint16_t* X = stageX(st);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int id = p + 256 * i;
                *reinterpret_cast<u32x4*>(X + (id >> 3) * G2_LS + (id & 7) * 8) = xreg[q][i];
            }
        };
#pragma unroll
        for (int q = 0; q < PF; q++) issue(q, q);
        lay(0, 0);
        issue(PF, 0);
        __syncthreads();
        // iteration kt lays tile kt+1 (register set (kt+1) % PF) into the stage the consumers have just left and refills the set with tile kt+1+PF;
        // unrolled by PF so that the set index is a compile-time constant
        // (hipcc's schedule of this loop is fragile: the same body without the two guards, or with sched_barrier pins between lay and issue,
        //  measured 720 and 509 us against 476 us for this form on 6400 x 5120 x 4096 tokens)
        for (int kb = 0; kb < nkt; kb += PF) {
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const int kt = kb + j;
                if (kt < nkt) {
                    if (kt + 1 < nkt) {
                        lay((kt + 1) & 1, (j + 1) % PF);
                        issue(kt + 1 + PF, (j + 1) % PF);
                    }
                    __syncthreads();
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers (waves 4..7)
    const int c = wave - 4, rh = c & 1, th = c >> 1;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int tb = 0; tb < 4; tb++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[rb][tb][i] = 0.f;
    __syncthreads(); /* stage 0 laid down */
    for (int kt = 0; kt < nkt; kt++) {
        const uint16_t* A = stageA(kt & 1) + (rh * 64 + r) * G2_LS + 8 * h;
        const uint16_t* X = stageX(kt & 1) + (th * 128 + r) * G2_LS + 8 * h;
#pragma unroll
        for (int s = 0; s < G2_BK / 16; s++) {
            u32x4 af[2], bf[4];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) af[rb] = *reinterpret_cast<const u32x4*>(A + rb * 32 * G2_LS + 16 * s);
#pragma unroll
            for (int tb = 0; tb < 4; tb++) bf[tb] = *reinterpret_cast<const u32x4*>(X + tb * 32 * G2_LS + 16 * s);
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int tb = 0; tb < 4; tb++)
                    acc[rb][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[rb]), __builtin_bit_cast(bf16x8, bf[tb]), acc[rb][tb], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rb = 0; rb < 2; rb++) gemm_epilogue<4>(acc[rb], a, tok0 + th * 128, row0 + rh * 64 + rb * 32, r, h);
}

// KF_OK launched, 1 = not for this kernel
int gemm2_launch(hipStream_t st, int fmt, const GemmArgs& a) {
    constexpr int min_wg = 128; /* fewer workgroups than this: the staged / direct tiles of kf_gemm.hip fill the chip better */
    if ((fmt != FMT_Q4 && fmt != FMT_BF16 && fmt != FMT_F8) || a.K % G2_BK != 0 || a.n < G2_BN) return 1;
    const dim3 grid((a.M + G2_BM - 1) / G2_BM, (a.n + G2_BN - 1) / G2_BN);
    if ((long)grid.x * grid.y < min_wg) return 1;
    const size_t smem = 2 * G2_STAGE;
    static bool raised[3] = {false, false, false};
    switch (fmt) {
        case FMT_Q4:
            if (!raised[0]) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<FMT_Q4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), raised[0] = true;
            hipLaunchKernelGGL(gemm2_kernel<FMT_Q4>, grid, dim3(512), smem, st, a);
            break;
        case FMT_BF16:
            if (!raised[1]) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<FMT_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), raised[1] = true;
            hipLaunchKernelGGL(gemm2_kernel<FMT_BF16>, grid, dim3(512), smem, st, a);
            break;
        default:
            if (!raised[2]) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<FMT_F8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), raised[2] = true;
            hipLaunchKernelGGL(gemm2_kernel<FMT_F8>, grid, dim3(512), smem, st, a);
            break;
    }
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
