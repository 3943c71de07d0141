// kf_gemm.hip -- fused PackedQ unpack + MFMA GEMM for a batch of tokens (prefill, SLP::Forw with nToken > 1), gfx950 / wave64.
//
// Reference: SLP::Forw (NeuronFuse.cu:305-381) -> GTensor::GetDataX dequantises the WHOLE weight to bf16 in gBUFF->tmpTernary
// (quantizer.cu:249-392), then cuBLASLt multiplies (gemm.cu:93-214); Fish::Chat feeds the prompt one token at a time through it
// (GoPT.cpp:1139-1146).  Here y[n, M] = x[n, K] . W[M, K]^T reads the packed stream once per 128-token tile, dequantises 8 weights
// per lane in registers with the reference's bf16-stepwise arithmetic (T.cu:274) straight into an MFMA A-fragment, and contracts on
// v_mfma_f32_32x32x16_bf16 against x fragments staged through LDS.
//
// Fragment mapping (cdna_hip_programming.md section 3): for the 32x32x16 bf16 MFMA lane l (r = l & 31, h = l >> 5) holds
// A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7; D has col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
// A = weights (rows = output features), B = x^T (cols = tokens).  The contraction index may be visited in any order as long as A and B
// agree, so the two lane halves do not take k and k+8 of one 16-wide step: over a staged tile of 128 k, half h owns whole 16-byte
// packed blocks (4-bit: blocks 2p + h; 2-bit: block h; 1-bit: the 8-byte half h of the block; bf16 / f8: the same element ranges) and
// walks them 8 elements per MFMA step.  Every lane therefore issues plain 16-byte (8-byte for 1-bit) loads of the packed stream and
// never exchanges data with another lane; the x fragment of a step is read from LDS at the matching element offset.
//
// Workgroup = 4 waves = RBW row blocks of 32 output rows x KS halves of the staged k tile (KS = 1: 128 rows; KS = 2: 64 rows, each k
// half on its own wave, partial sums combined through LDS in a fixed order).  Grid = (row tiles, 128-token tiles).
// fp32 accumulation inside the MFMA, one bf16 round-to-nearest store; epilogue order as the mat-vec (alpha, beta*y, bias, store,
// residual add + store).  Differs from the token-serial mat-vec only in fp32 summation order.
#include <stdlib.h>
#include <string.h>

#include "kf_gemm_common.h"

namespace kf {

// ---- the weights one lane needs for one staged tile: NS = 8 / KS MFMA steps
// step sl of wave-half ks: elements [koff, koff + 8) of the 128-wide tile
template <int FMT, int KS>
struct WTile;

// 4-bit, bf16, f8 share the element ranges: pair P = ks * NP + p (NP = 2 / KS), lane half h owns elements (2P + h) * 32 .. + 32
template <int KS>
struct WTile<FMT_Q4, KS> {
    static constexpr int NP = 2 / KS;
    u32x4 b[NP];
    uint16_t st[NP], ze[NP]; /* raw bf16 bits: converted at use, so that nothing waits on the loads right behind their issue */
    bool in[NP];
    __device__ __forceinline__ void load(const GemmArgs& a, int row, int it, int h, int ks) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            // a row's last staged tile may be half empty (K a multiple of 64 only): blocks past the row end are read from the last valid
            // block (no branch around the load) and given step = zero = 0, which unpacks to weights of 0
            int bi = it * 4 + 2 * (ks * NP + p) + h;
            const bool in_row = bi < a.nBlk;
            if (!in_row) bi = a.nBlk - 1;
            const uint32_t bidx = (uint32_t)row * (uint32_t)a.nBlk + (uint32_t)bi;
            b[p] = ld_nt(reinterpret_cast<const u32x4*>(a.w) + bidx);
            const uint32_t gi = bidx >> a.gshift;
            st[p] = a.step[gi], ze[p] = a.zero[gi], in[p] = in_row;
        }
    }
    static __device__ __forceinline__ int koff(int sl, int h, int ks) { return (2 * (ks * NP + (sl >> 2)) + h) * 32 + 8 * (sl & 3); }
    __device__ __forceinline__ u32x4 frag(int sl, const GemmArgs& a) const {
        const int p = sl >> 2, c = sl & 3;
        const uint32_t D = c == 0 ? b[p].w : (c == 1 ? b[p].z : (c == 2 ? b[p].y : b[p].x));
        const float s = in[p] ? bf2f(st[p]) : 0.f, z = in[p] ? bf2f(ze[p]) : 0.f;
        return frag_q4(D, s, s * 0.0625f, -a.qBias * s, z);
    }
};
// 4-bit row codebook: the same block ranges as the Packed128 form; the row's 16-entry table rides along (two 16-byte loads through the cache)
template <int KS>
struct WTile<FMT_Q4R, KS> {
    static constexpr int NP = 2 / KS;
    u32x4 b[NP];
    u32x4 ta, tb;
    bool in[NP];
    __device__ __forceinline__ void load(const GemmArgs& a, int row, int it, int h, int ks) {
        const u32x4* lt = reinterpret_cast<const u32x4*>(a.zero) + 2 * (size_t)row; /* a.zero carries the table base */
        ta = lt[0], tb = lt[1];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            int bi = it * 4 + 2 * (ks * NP + p) + h;
            const bool in_row = bi < a.nBlk;
            if (!in_row) bi = a.nBlk - 1;
            b[p] = ld_nt(reinterpret_cast<const u32x4*>(a.w) + (uint32_t)row * (uint32_t)a.nBlk + (uint32_t)bi);
            in[p] = in_row;
        }
    }
    static __device__ __forceinline__ int koff(int sl, int h, int ks) { return (2 * (ks * NP + (sl >> 2)) + h) * 32 + 8 * (sl & 3); }
    __device__ __forceinline__ u32x4 frag(int sl, const GemmArgs&) const {
        const int p = sl >> 2, c = sl & 3;
        const uint32_t D = c == 0 ? b[p].x : (c == 1 ? b[p].y : (c == 2 ? b[p].z : b[p].w)); /* stream order: dword 0 holds elements 0..7 */
        const u32x4 o = frag_q4r(D, ta, tb);
        const uint32_t keep = in[p] ? 0xffffffffu : 0u;
        return u32x4{o.x & keep, o.y & keep, o.z & keep, o.w & keep};
    }
};
template <int KS>
struct WTile<FMT_BF16, KS> {
    static constexpr int NP = 2 / KS;
    u32x4 b[NP][4];
    __device__ __forceinline__ void load(const GemmArgs& a, int row, int it, int h, int ks) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            int e = it * GM_KT + (2 * (ks * NP + p) + h) * 32; /* first of this lane's 32 elements; past the row end: zeros (address clamped) */
            const uint32_t keep = e < a.K ? 0xffffffffu : 0u;
            if (e >= a.K) e = a.K - 32;
            const u32x4* src = reinterpret_cast<const u32x4*>(a.w + ((size_t)row * a.K + (size_t)e) * 2);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const u32x4 v = ld_nt(src + c);
                b[p][c] = u32x4{v.x & keep, v.y & keep, v.z & keep, v.w & keep};
            }
        }
    }
    static __device__ __forceinline__ int koff(int sl, int h, int ks) { return (2 * (ks * NP + (sl >> 2)) + h) * 32 + 8 * (sl & 3); }
    __device__ __forceinline__ u32x4 frag(int sl, const GemmArgs&) const { return b[sl >> 2][sl & 3]; }
};
template <int KS>
struct WTile<FMT_F8, KS> {
    static constexpr int NP = 2 / KS;
    u32x4 b[NP][2];
    __device__ __forceinline__ void load(const GemmArgs& a, int row, int it, int h, int ks) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            int e = it * GM_KT + (2 * (ks * NP + p) + h) * 32;
            const uint32_t keep = e < a.K ? 0xffffffffu : 0u;
            if (e >= a.K) e = a.K - 32;
            const u32x4* src = reinterpret_cast<const u32x4*>(a.w + (size_t)row * a.K + (size_t)e);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const u32x4 v = ld_nt(src + c);
                b[p][c] = u32x4{v.x & keep, v.y & keep, v.z & keep, v.w & keep};
            }
        }
    }
    static __device__ __forceinline__ int koff(int sl, int h, int ks) { return (2 * (ks * NP + (sl >> 2)) + h) * 32 + 8 * (sl & 3); }
    __device__ __forceinline__ u32x4 frag(int sl, const GemmArgs&) const {
        const int p = sl >> 2, c = sl & 3;
        const u32x4 v = b[p][c >> 1];
        return (c & 1) ? frag_f8(v.z, v.w) : frag_f8(v.x, v.y);
    }
};
// 2-bit: the staged tile is two 64-element blocks, lane half h owns block h; global step s = ks * NS + sl covers elements 8s..8s+7
template <int KS>
struct WTile<FMT_Q2, KS> {
    static constexpr int NS = 8 / KS;
    u32x4 b;
    uint16_t st, ze;
    __device__ __forceinline__ void load(const GemmArgs& a, int row, int it, int h, int) {
        const uint32_t bidx = (uint32_t)row * (uint32_t)a.nBlk + (uint32_t)(it * 2 + h);
        b = ld_nt(reinterpret_cast<const u32x4*>(a.w) + bidx);
        const uint32_t gi = bidx >> a.gshift;
        st = a.step[gi], ze = a.zero[gi];
    }
    static __device__ __forceinline__ int koff(int sl, int h, int ks) { return 64 * h + 8 * (ks * NS + sl); }
    __device__ __forceinline__ u32x4 frag(int sl, const GemmArgs& a, int ks) const {
        const uint32_t dw[4] = {b.w, b.z, b.y, b.x}; /* dword3 holds elements 0..15 */
        uint32_t D = dw[sl >> 1];
        if (KS == 2) D = ks ? dw[2 + (sl >> 1)] : D;
        const uint32_t v = (sl & 1) ? (D & 0xffffu) : (D >> 16);
        const float s = bf2f(st);
        return frag_q2(v, s, -a.qBias * s, bf2f(ze));
    }
};
// 1-bit: the staged tile is one 128-element block, lane half h owns its 8-byte half (elements 64h .. 64h+63)
template <int KS>
struct WTile<FMT_Q1, KS> {
    static constexpr int NS = 8 / KS;
    u32x2 b; /* b.y: first 32 elements of the half, b.x: next 32 */
    uint16_t st, ze;
    __device__ __forceinline__ void load(const GemmArgs& a, int row, int it, int h, int) {
        const uint32_t bidx = (uint32_t)row * (uint32_t)a.nBlk + (uint32_t)it;
        b = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(a.w) + (size_t)bidx * 2 + (1 - h));
        const uint32_t gi = bidx >> a.gshift;
        st = a.step[gi], ze = a.zero[gi];
    }
    static __device__ __forceinline__ int koff(int sl, int h, int ks) { return 64 * h + 8 * (ks * NS + sl); }
    __device__ __forceinline__ u32x4 frag(int sl, const GemmArgs& a, int ks) const {
        const uint32_t D = (KS == 2) ? (ks ? b.x : b.y) : ((sl >> 2) ? b.x : b.y);
        const uint32_t byte = (D >> (24 - 8 * (sl & 3))) & 0xffu;
        const float s = bf2f(st), z = bf2f(ze), nb = -a.qBias * s; /* dequant(0), dequant(1): loop-invariant, hoisted by the compiler */
        const uint32_t r = pack_bf16x2(fmaf(0.0f, s, nb), fmaf(1.0f, s, nb));
        const uint32_t ww = pack_bf16x2(bf_lo(r) - z, bf_hi(r) - z);
        return frag_q1(byte, ww & 0xffffu, ww >> 16);
    }
};

template <int FMT, int KS>
__device__ __forceinline__ u32x4 get_frag(const WTile<FMT, KS>& t, int sl, const GemmArgs& a, int ks) {
    if constexpr (FMT == FMT_Q2 || FMT == FMT_Q1)
        return t.frag(sl, a, ks);
    else
        return t.frag(sl, a);
}

template <int FMT, int KS>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem_raw); /* 2 x [GM_TOK][GM_XS] bf16; reused as the k-half combine buffer at the end */
    constexpr int RBW = 4 / KS, NS = 8 / KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rb = wave % RBW, ks = wave / RBW;
    const int row_base = (blockIdx.x * RBW + rb) * 32;
    int row = row_base + r;
    if (row >= a.M) row = a.M - 1; /* rows past the end recompute the last row; their results are not stored */
    const int tok0 = blockIdx.y * GM_TOK;
    const int nit = (a.K + GM_KT - 1) / GM_KT;

    f32x16 acc[4];
#pragma unroll
    for (int tb = 0; tb < 4; tb++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[tb][i] = 0.f;

    // x staging: 16 threads cover one token row of 256 B, 16 rows per pass, 8 passes
    const int trow = tid >> 4, seg = tid & 15;
    u32x4 xr[8];
    auto xload = [&](int it) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            int tok = tok0 + p * 16 + trow;
            if (tok >= a.n) tok = a.n - 1; /* rows past the batch repeat the last token: loaded unconditionally, never stored */
            int col = it * GM_KT + seg * 8;
            if (col >= a.K) col = a.K - 8; /* half-empty last tile: the weights there are 0, any finite x will do */
            xr[p] = *reinterpret_cast<const u32x4*>(a.x + (size_t)tok * a.ldx + col);
        }
    };
    auto xstore = [&](int buf) {
        uint16_t* dst = xs + (size_t)buf * GM_TOK * GM_XS;
#pragma unroll
        for (int p = 0; p < 8; p++) *reinterpret_cast<u32x4*>(dst + (p * 16 + trow) * GM_XS + seg * 8) = xr[p];
    };

    // Two tiles in flight: while tile `it` is multiplied out of LDS, tile it+1 waits in registers (loaded during the previous iteration,
    // written to the other LDS buffer at the end of this one) and tile it+2's loads are issued -- a load has a whole iteration of
    // arithmetic plus a barrier to land before anything waits for it (one tile ahead left ~2 us of latency exposed per iteration).
    WTile<FMT, KS> wc, w1, w2;
    u32x4 xr2[8];
    auto xload2 = [&](int it) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            int tok = tok0 + p * 16 + trow;
            if (tok >= a.n) tok = a.n - 1;
            int col = it * GM_KT + seg * 8;
            if (col >= a.K) col = a.K - 8;
            xr2[p] = *reinterpret_cast<const u32x4*>(a.x + (size_t)tok * a.ldx + col);
        }
    };
    wc.load(a, row, 0, h, ks);
    xload(0);
    xstore(0);
    // every load below is unconditional (past the last tile the last one is simply fetched again): with branches around them hipcc
    // cannot count the loads in flight and falls back to s_waitcnt vmcnt(0), which waits for the NEWEST tile as well
    {
        const int i1 = nit > 1 ? 1 : 0;
        w1.load(a, row, i1, h, ks);
        xload(i1);
    }
    __syncthreads();
    for (int it = 0; it < nit; it++) {
        const bool more = it + 1 < nit, more2 = true;
        {
            const int i2 = it + 2 < nit ? it + 2 : nit - 1;
            w2.load(a, row, i2, h, ks);
            xload2(i2);
        }
        const uint16_t* xb = xs + (size_t)(it & 1) * GM_TOK * GM_XS;
        // x fragments of step sl+1 are read from LDS while the weights of step sl are unpacked and multiplied (read just before
        // each MFMA, every ds_read_b128 round trip sat in front of its MFMA)
        u32x4 Bc[4], Bn[4];
        {
            const int ko = WTile<FMT, KS>::koff(0, h, ks);
#pragma unroll
            for (int tb = 0; tb < 4; tb++) Bc[tb] = *reinterpret_cast<const u32x4*>(xb + (tb * 32 + r) * GM_XS + ko);
        }
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
            if (sl + 1 < NS) {
                const int ko = WTile<FMT, KS>::koff(sl + 1, h, ks);
#pragma unroll
                for (int tb = 0; tb < 4; tb++) Bn[tb] = *reinterpret_cast<const u32x4*>(xb + (tb * 32 + r) * GM_XS + ko);
            }
            __builtin_amdgcn_sched_barrier(0); /* keep the reads up here: left alone, the scheduler sinks each one next to its MFMA */
            const bf16x8 A = __builtin_bit_cast(bf16x8, get_frag<FMT, KS>(wc, sl, a, ks));
#pragma unroll
            for (int tb = 0; tb < 4; tb++) acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(bf16x8, Bc[tb]), acc[tb], 0, 0, 0);
#pragma unroll
            for (int tb = 0; tb < 4; tb++) Bc[tb] = Bn[tb];
        }
        if (more) {
            xstore((it + 1) & 1);
            wc = w1;
        }
        if (more2) {
            w1 = w2;
#pragma unroll
            for (int p = 0; p < 8; p++) xr[p] = xr2[p];
        }
        __syncthreads();
    }

    if (KS == 2) { /* the k-halves of a row block: upper half hands its sums over, fixed order (lower + upper) */
        float* red = reinterpret_cast<float*>(smem_raw);
        if (ks == 1) {
#pragma unroll
            for (int tb = 0; tb < 4; tb++)
#pragma unroll
                for (int i = 0; i < 16; i++) red[((rb * 4 + tb) * 16 + i) * 64 + lane] = acc[tb][i];
        }
        __syncthreads();
        if (ks == 1) return;
#pragma unroll
        for (int tb = 0; tb < 4; tb++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[tb][i] = acc[tb][i] + red[((rb * 4 + tb) * 16 + i) * 64 + lane];
    }

    gemm_epilogue<4>(acc, a, tok0, row_base, r, h);
}

// ---- small token counts: every wave is independent.  Workgroup = one block of 32 output rows x TB*32 tokens; its NW waves split k in
// groups of UPG consecutive 64-element units (for 2-bit / 1-bit: halves of a 128-element unit): wave w takes groups w, w + NW, ...,
// reads its x fragments straight from global memory (x is n*K*2 bytes: L2-resident) and meets the others once, at the end, where
// wave 0 adds the sums in wave order.  All loads of a group are issued before any of its arithmetic and the next group's loads before
// the current group's arithmetic: the prompt-sized GEMMs of a 0.6B model are latency-bound (a 1024 x 1024 4-bit matrix is 0.5 MB,
// far fewer waves than SIMDs), so the k loop is kept as short as the register file allows (K = 1024: one group per wave, no loop).
// PAIRED: the workgroup multiplies the same x fragments against TWO matrices (gate, up) and stores silu(gate) * up: the bf16 stores of
// gate and up and the SwiGLU expression are the ones of the separate launches (kf_linear x 2 + kf_swiglu), so the result is bit-identical.
// Otherwise a launch may carry up to three matrices that share x (Q, K, V): blockIdx.x walks their 32-row blocks one job after the other.
template <int FMT, int TB, int UPG, int NW, bool PAIRED>
__global__ void __launch_bounds__(NW * 64) gemm_direct_kernel(const GemmArgs a0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw); /* [PAIRED ? 2 : 1][NW][TB][16][64] */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    GemmArgs a = a0;
    int rb = blockIdx.x;
    if (!PAIRED && a0.njobs > 1 && rb >= a0.rb_end[0]) { /* workgroup-uniform: the fields below stay scalar */
        const bool j2 = a0.njobs > 2 && rb >= a0.rb_end[1];
        rb -= j2 ? a0.rb_end[1] : a0.rb_end[0];
        a.w = j2 ? a0.xw[1] : a0.xw[0];
        a.zero = j2 ? a0.xzero[1] : a0.xzero[0];
        a.step = j2 ? a0.xstep[1] : a0.xstep[0];
        a.qBias = j2 ? a0.xqBias[1] : a0.xqBias[0];
        a.M = j2 ? a0.xM[1] : a0.xM[0];
        a.y = j2 ? a0.xy[1] : a0.xy[0];
        a.ldy = j2 ? a0.xldy[1] : a0.xldy[0];
    }
    GemmArgs b = a0; /* PAIRED: the second matrix */
    if (PAIRED) b.w = a0.xw[0], b.zero = a0.xzero[0], b.step = a0.xstep[0], b.qBias = a0.xqBias[0];
    const int row_base = rb * 32;
    int row = row_base + r;
    if (row >= a.M) row = a.M - 1;
    const int tok0 = blockIdx.y * (TB * 32);
    const int nunit = a.K / 64;

    f32x16 acc[TB], acc2[PAIRED ? TB : 1];
#pragma unroll
    for (int tb = 0; tb < TB; tb++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            acc[tb][i] = 0.f;
            if (PAIRED) acc2[tb][i] = 0.f;
        }

    // x rows of this lane's tokens (rows past n read row n-1: their results are not stored)
    const uint16_t* xrow[TB];
#pragma unroll
    for (int tb = 0; tb < TB; tb++) {
        int tok = tok0 + tb * 32 + r;
        if (tok >= a.n) tok = a.n - 1;
        xrow[tb] = a.x + (size_t)tok * a.ldx;
    }
    struct Group {
        WTile<FMT, 2> w[UPG];
        WTile<FMT, 2> w2[PAIRED ? UPG : 1];
        u32x4 x[UPG][4][TB];
    };
    auto gload = [&](int u0, Group& g) { /* units u0 .. u0+UPG-1; units past the end repeat the last one and are skipped in the arithmetic */
#pragma unroll
        for (int j = 0; j < UPG; j++) {
            int u = u0 + j;
            if (u >= nunit) u = nunit - 1;
            const int it = u >> 1, ks = u & 1;
            g.w[j].load(a, row, it, h, ks);
            if (PAIRED) g.w2[j].load(b, row, it, h, ks);
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                const int ko = it * GM_KT + WTile<FMT, 2>::koff(sl, h, ks);
#pragma unroll
                for (int tb = 0; tb < TB; tb++) g.x[j][sl][tb] = *reinterpret_cast<const u32x4*>(xrow[tb] + ko);
            }
        }
    };
    Group gc, gn;
    int u0 = wave * UPG;
    if (u0 < nunit) gload(u0, gc);
    for (; u0 < nunit; u0 += NW * UPG) {
        const bool more = u0 + NW * UPG < nunit;
        if (more) gload(u0 + NW * UPG, gn);
#pragma unroll
        for (int j = 0; j < UPG; j++) {
            if (u0 + j < nunit) {
#pragma unroll
                for (int sl = 0; sl < 4; sl++) {
                    const bf16x8 A = __builtin_bit_cast(bf16x8, get_frag<FMT, 2>(gc.w[j], sl, a, (u0 + j) & 1));
#pragma unroll
                    for (int tb = 0; tb < TB; tb++)
                        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(bf16x8, gc.x[j][sl][tb]), acc[tb], 0, 0, 0);
                    if (PAIRED) {
                        const bf16x8 A2 = __builtin_bit_cast(bf16x8, get_frag<FMT, 2>(gc.w2[j], sl, b, (u0 + j) & 1));
#pragma unroll
                        for (int tb = 0; tb < TB; tb++)
                            acc2[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, __builtin_bit_cast(bf16x8, gc.x[j][sl][tb]), acc2[tb], 0, 0, 0);
                    }
                }
            }
        }
        if (more) gc = gn;
    }
    // fixed-order sum over the k-slices, ((w0 + w1) + w2) + ..., and the epilogue, spread over the waves: every wave leaves its 16 sums per token block in LDS and
    // wave w adds and stores accumulator registers 16 w / NW .. (two adjacent output rows per lane with 8 waves) -- the same additions in the same order as one wave
    // doing all of it, at an eighth of the serial work behind the barrier (measured on the 0.6B shapes at 128 tokens: reduce 0.65 us + epilogue 1.3-1.7 us by wave 0)
    constexpr int RPW = 16 / NW; /* registers per wave */
    static_assert(NW == 8 || NW == 4, "registers per wave: pairs of adjacent rows");
#pragma unroll
    for (int tb = 0; tb < TB; tb++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            red[((wave * TB + tb) * 16 + i) * 64 + lane] = acc[tb][i];
            if (PAIRED) red[(((NW + wave) * TB + tb) * 16 + i) * 64 + lane] = acc2[tb][i];
        }
    __syncthreads();
#pragma unroll
    for (int tb = 0; tb < TB; tb++) {
        const int tok = tok0 + tb * 32 + r;
#pragma unroll
        for (int q = 0; q < RPW; q += 2) {
            const int i0 = wave * RPW + q; /* registers i0, i0 + 1: rows rg, rg + 1 */
            float v[2] = {0.f, 0.f}, v2[2] = {0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; w++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float p = red[((w * TB + tb) * 16 + i0 + j) * 64 + lane];
                    v[j] = w == 0 ? p : v[j] + p;
                    if (PAIRED) {
                        const float p2 = red[(((NW + w) * TB + tb) * 16 + i0 + j) * 64 + lane];
                        v2[j] = w == 0 ? p2 : v2[j] + p2;
                    }
                }
            const int rg = row_base + 8 * (i0 >> 2) + 4 * h + (i0 & 3);
            if (tok >= a.n || rg >= a.M) continue;
            uint16_t o[2] = {0, 0};
            uint16_t* yp = a.y + (size_t)tok * a.ldy + rg;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (rg + j >= a.M) continue;
                if (!PAIRED) { /* gemm_epilogue's expression */
                    float x = v[j];
                    if (a.alpha != 1.0f) x = a.alpha * x;
                    if (a.beta != 0.0f) x = x + a.beta * bf2f(yp[j]);
                    if (a.bias) x = x + bf2f(a.bias[rg + j]);
                    uint16_t qv = f2bf(x);
                    if (a.residual) qv = f2bf(bf2f(a.residual[(size_t)tok * a.ldr + rg + j]) + bf2f(qv));
                    o[j] = qv;
                } else { /* act = silu(gate) * up on the bf16-rounded gate / up (Relu::Forw SWIG, Activation.cu:85-93; the expression of kf_swiglu) */
                    const float g = round_bf16(v[j]), u = round_bf16(v2[j]);
                    o[j] = f2bf((g * u) / (1.0f + kf_expf(-g)));
                }
            }
            if (((a.ldy & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 3) == 0) && rg + 1 < a.M) {
                *reinterpret_cast<uint32_t*>(yp) = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            } else {
                yp[0] = o[0];
                if (rg + 1 < a.M) yp[1] = o[1];
            }
        }
    }
}

static int gm_fmt_of(int type) {
    switch (type) {
        case KF_BF16: return FMT_BF16;
        case KF_F8E5M2: return FMT_F8;
        case KF_Q4: return FMT_Q4;
        case KF_T_SIGN: return FMT_Q2;
        case KF_BOOL1: case KF_T_BINARY: return FMT_Q1;
        default: return -1;
    }
}

constexpr int GD_NW = 8, GD_UPG = 2; /* direct kernel: 8 waves, groups of two 64-element units, 32-token tiles */

template <int FMT>
static void gm_launch(const GemmArgs& a, int KS, dim3 grid, size_t smem, hipStream_t st) {
    if (KS == 0)
        hipLaunchKernelGGL((gemm_direct_kernel<FMT, 1, GD_UPG, GD_NW, false>), grid, dim3(GD_NW * 64), (size_t)GD_NW * 16 * 64 * 4, st, a);
    else if (KS == -2) /* 64-token tiles: every weight block is unpacked for two token blocks */
        hipLaunchKernelGGL((gemm_direct_kernel<FMT, 2, GD_UPG, GD_NW, false>), grid, dim3(GD_NW * 64), (size_t)GD_NW * 2 * 16 * 64 * 4, st, a);
    else if (KS == -1) /* paired SwiGLU */
        hipLaunchKernelGGL((gemm_direct_kernel<FMT, 1, GD_UPG, GD_NW, true>), grid, dim3(GD_NW * 64), (size_t)GD_NW * 2 * 16 * 64 * 4, st, a);
    else if (KS == 2)
        hipLaunchKernelGGL((gemm_kernel<FMT, 2>), grid, dim3(256), smem, st, a);
    else
        hipLaunchKernelGGL((gemm_kernel<FMT, 1>), grid, dim3(256), smem, st, a);
}
static void gm_dispatch(int fmt, const GemmArgs& a, int KS, dim3 grid, size_t smem, hipStream_t st) {
    switch (fmt) {
        case FMT_BF16: gm_launch<FMT_BF16>(a, KS, grid, smem, st); break;
        case FMT_F8: gm_launch<FMT_F8>(a, KS, grid, smem, st); break;
        case FMT_Q4: gm_launch<FMT_Q4>(a, KS, grid, smem, st); break;
        case FMT_Q2: gm_launch<FMT_Q2>(a, KS, grid, smem, st); break;
        case FMT_Q4R: gm_launch<FMT_Q4R>(a, KS, grid, smem, st); break;
        default: gm_launch<FMT_Q1>(a, KS, grid, smem, st); break;
    }
}

static const int gm_epb[7] = {8, 16, 32, 64, 128, 32, 32};
struct GmWeight {
    const unsigned char* w;
    const uint16_t *zero, *step;
    float qBias;
    int M, K, fmt, gshift;
};
// 0 ok, 1 not eligible for the tile kernels, < 0 error
static int gm_weight(const kf_weight* w, GmWeight& o) {
    o.fmt = gm_fmt_of(w->type);
    if (is_row_lut(w)) { /* 4-bit row codebooks unpack in registers like the Packed128 form; the 3- / 2-bit row forms are dequantised by the caller */
        if (w->type != KF_Q4 || w->quant != KF_QUANT_ROW_LUT) return 1;
        o.fmt = FMT_Q4R;
    }
    if (o.fmt < 0 || w->qzeros || w->qscales) return 1;
    o.M = w->ne0, o.K = w->ne1;
    // K a multiple of 128 for every kernel; a multiple of 64 is enough for the direct kernel on the formats whose 64-element unit is made of
    // whole blocks (bf16, f8, 4-bit) -- GPT-2's n_embd = 1600
    if (o.K % 64 != 0 || o.K < GM_KT || o.M < 1 || (reinterpret_cast<uintptr_t>(w->data) & 15) != 0) return 1;
    if (o.K % GM_KT != 0 && o.fmt > FMT_Q4 && o.fmt != FMT_Q4R) return 1;
    if ((unsigned long long)o.M * (unsigned long long)(o.K / gm_epb[o.fmt]) >= (1ull << 32)) return 1;
    o.w = reinterpret_cast<const unsigned char*>(w->data);
    o.zero = o.step = nullptr, o.qBias = (float)w->qBias, o.gshift = 0;
    if (o.fmt == FMT_Q4R) {
        if (!w->gama) return KF_QUANT_ERR;
        o.zero = w->gama + w->ne0 + w->ne1; /* the rows' tables */
        if ((reinterpret_cast<uintptr_t>(o.zero) & 15) != 0) return 1;
    } else if (o.fmt >= FMT_Q4) {
        if (!w->gama || w->lGroup <= 0 || (w->lGroup % gm_epb[o.fmt]) != 0 || ((long)o.M * o.K) % w->lGroup != 0) return KF_QUANT_ERR;
        const int bpg = w->lGroup / gm_epb[o.fmt];
        if (bpg < 1 || (bpg & (bpg - 1)) != 0) return KF_QUANT_ERR;
        o.gshift = __builtin_ctz(bpg);
        o.zero = w->gama + w->ne0 + w->ne1;
        o.step = o.zero + (size_t)o.M * o.K / w->lGroup;
    }
    return 0;
}
static void gm_base(GemmArgs& a, const GmWeight& g, const uint16_t* x, long long ldx, int n, uint16_t* y, long long ldy) {
    memset(&a, 0, sizeof(a));
    a.w = g.w, a.zero = g.zero, a.step = g.step, a.qBias = g.qBias;
    a.M = g.M, a.K = g.K, a.nBlk = g.K / gm_epb[g.fmt], a.gshift = g.gshift;
    a.x = x, a.ldx = ldx, a.n = n, a.y = y, a.ldy = ldy, a.alpha = 1.0f, a.njobs = 1;
    a.rb_end[0] = a.rb_end[1] = a.rb_end[2] = (g.M + 31) / 32;
}
static bool gm_x_ok(const uint16_t* x, long long ldx) { return (ldx & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0; }
constexpr long GD_FUSED_MAX = 2048; /* workgroups up to which the fused direct launches beat separate (staged) ones */

// up to three matrices sharing x (Q, K, V) in ONE direct launch: KF_OK, 1 = not eligible (the caller launches them one by one), < 0 error
int gemm_multi_launch(hipStream_t st, int n_w, const kf_weight* const* w, const uint16_t* x, long long ldx, int n, uint16_t* const* y) {
    if (n_w < 2 || n_w > 3 || !gm_x_ok(x, ldx)) return 1;
    GmWeight g[3];
    long rbs = 0;
    for (int i = 0; i < n_w; i++) {
        const int rc = gm_weight(w[i], g[i]);
        if (rc) return rc;
        if (g[i].fmt != g[0].fmt || g[i].K != g[0].K || g[i].gshift != g[0].gshift) return 1;
        rbs += (g[i].M + 31) / 32;
    }
    if (rbs * ((n + 31) / 32) > GD_FUSED_MAX) return 1;
    GemmArgs a;
    gm_base(a, g[0], x, ldx, n, y[0], g[0].M);
    a.njobs = n_w;
    int end = 0;
    for (int i = 0; i < 3; i++) {
        if (i < n_w) end += (g[i].M + 31) / 32;
        a.rb_end[i] = end;
        if (i >= 1 && i < n_w)
            a.xw[i - 1] = g[i].w, a.xzero[i - 1] = g[i].zero, a.xstep[i - 1] = g[i].step, a.xqBias[i - 1] = g[i].qBias, a.xM[i - 1] = g[i].M, a.xy[i - 1] = y[i],
                     a.xldy[i - 1] = g[i].M;
    }
    if (n >= 64 && (long)end * ((n + 63) / 64) >= 256)
        gm_dispatch(g[0].fmt, a, -2, dim3(end, (n + 63) / 64), 0, st);
    else
        gm_dispatch(g[0].fmt, a, 0, dim3(end, (n + 31) / 32), 0, st);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// act[n, M] = silu(x . gate^T) * (x . up^T) in one direct launch: KF_OK, 1 = not eligible, < 0 error
int gemm_paired_launch(hipStream_t st, const kf_weight* gate, const kf_weight* up, const uint16_t* x, long long ldx, int n, uint16_t* act) {
    if (!gm_x_ok(x, ldx)) return 1;
    GmWeight g, u;
    int rc = gm_weight(gate, g);
    if (rc) return rc;
    rc = gm_weight(up, u);
    if (rc) return rc;
    if (g.fmt != u.fmt || g.K != u.K || g.M != u.M || g.gshift != u.gshift) return 1;
    if ((long)((g.M + 31) / 32) * ((n + 31) / 32) > GD_FUSED_MAX) return 1;
    GemmArgs a;
    gm_base(a, g, x, ldx, n, act, g.M);
    a.xw[0] = u.w, a.xzero[0] = u.zero, a.xstep[0] = u.step, a.xqBias[0] = u.qBias;
    gm_dispatch(g.fmt, a, -1, dim3((g.M + 31) / 32, (n + 31) / 32), 0, st);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

// Returns KF_OK when launched, 1 when the shape is not eligible (the caller then loops the mat-vec), < 0 on error.
int gemm_launch(hipStream_t st, const kf_weight* w, const uint16_t* x, long long ldx, int n, uint16_t* y, long long ldy, const uint16_t* bias, float alpha,
                float beta, const uint16_t* residual, long long ldr, void* ws, size_t ws_bytes) {
    GmWeight g;
    const int rc = gm_weight(w, g);
    if (rc) return rc;
    if (!gm_x_ok(x, ldx)) return 1;
    const int M = g.M;
    GemmArgs a;
    gm_base(a, g, x, ldx, n, y, ldy);
    a.bias = bias, a.residual = residual, a.ldr = ldr, a.alpha = alpha, a.beta = beta;
    const int ttiles = (n + GM_TOK - 1) / GM_TOK;
    // prompt-sized batches on small matrices: the wave-independent kernel while its 32 x 32 workgroups fit ~2-3 rounds of the chip
    // (measured crossover against the staged tiles, scratch/ub_gemm.py: 1024 rows up to n ~ 1024, 2048 up to ~ 600, 3072 up to ~ 400)
    constexpr int direct_max = 1280;
    int KS = ((long)((M + 127) / 128) * ttiles < 512) ? 2 : 1;
    dim3 grid;
    // bf16 operands (weights stored as bf16, or the resident dequantised copies of a long prompt): the global_load_lds tile kernels first from g3_first token rows --
    // the crossover above was measured on the 4-bit in-register unpack; on bf16 the direct kernel took 64 us for 1024 x 2048 at 1024 rows
    if (g.fmt == FMT_BF16 && n >= g_knobs.g3_first) {
        const int rc3 = gemm3_launch(st, g.fmt, a, ws, ws_bytes);
        if (rc3 != 1) return rc3;
    }
    if ((long)((M + 31) / 32) * ((n + 31) / 32) <= direct_max) {
        KS = 0;
        grid = dim3((M + 31) / 32, (n + 31) / 32);
    } else {
        const int rc3 = gemm3_launch(st, g.fmt, a); /* bf16 operands, >= 128 tiles of 256 x 256: the global_load_lds tile kernel (kf_gemm3.hip) */
        if (rc3 != 1) return rc3;
        const int rc2 = gemm2_launch(st, g.fmt, a); /* large batches: the producer / consumer tile kernel when it applies */
        if (rc2 != 1) return rc2;
        const int rows_per_wg = 32 * (4 / KS);
        grid = dim3((M + rows_per_wg - 1) / rows_per_wg, ttiles);
    }
    gm_dispatch(g.fmt, a, KS, grid, (size_t)2 * GM_TOK * GM_XS * sizeof(uint16_t), st);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
