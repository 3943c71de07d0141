// kf_gemm3.hip -- large-batch bf16 GEMM on 256 x 256 x 64 (or 128 x 128 x 64) tiles: y[n, M] (+)= x[n, K] . W[M, K]^T (SLP::Forw's product for a dequantised or bf16
// weight) and, with either operand stored K-MAJOR, both GEMMs of SLP::Back without a transpose of anything.  The tile the LDS accounting of DESIGN.md section 8 asked
// for, built MI355X-first:
//   * both operand tiles go global -> LDS with global_load_lds_dwordx4 (16 bytes per lane, no VGPR round trip).  The LDS image of such a load is
//     lane-linear, so the bank-conflict-free layout is produced on the SOURCE side: a tile row is 128 bytes = 8 chunks of 16 bytes, chunk c of row r
//     is stored at position c ^ ((r >> 1) & 7); a fragment read (16 rows x one chunk, ds_read_b128) then touches all 64 banks exactly once;
//     a k-major operand's image is [64 k][rows], two-term swizzle, fragments by the transposing ds_read_b64_tr_b16 (g3_src_km / g3_frag_km);
//   * two LDS buffers: the loads of k-step t+1 are issued while k-step t is multiplied and drained with a COUNTED s_waitcnt vmcnt and ONE raw s_barrier per
//     step (a __syncthreads() would drain the loads in flight); per-lane source pointers set up once, LDS buffer indices compile-time;
//   * 256 x 256: 8 waves as 2 (rows) x 4 (tokens), 128 x 64 outputs per wave = 32 accumulator tiles of v_mfma_f32_16x16x32_bf16, one workgroup per CU;
//     128 x 128: 4 waves as 2 x 2, 64 x 64 outputs per wave, two workgroups per CU -- for products with too few big tiles to fill the chip;
//   * workgroup order remapped so that the blocks an XCD runs back to back share their W row-tile in that XCD's L2;
//   * split-K forms (gemm3_sk_kernel) with an ordered in-kernel fix-up for launches with fewer tiles than resident workgroups: deterministic, no atomics on data;
//   * up to three matrices stacked along M (Q | K | V, gate | up) in one launch, each tile's rows routed to its matrix's output (gemm3_multi_launch).
// Measured (MI355X): 8192 x 6400 x 5120 993-1009 TFLOP/s; M 1024 x K 3072 at 8192 rows 890 on the small tile; GPT2-1558M weight gradients 540-740; per-shape numbers
// and the measured dead ends (stream-K ranges, BK = 32 / 4 buffers, mid-step barrier, 4 waves x 128 x 128) are in DESIGN.md section 0 (3).
// The epilogue is gemm_epilogue's (alpha, beta, bias, one bf16 store, residual added to the rounded value).
#include <string.h>

#include <type_traits>

#include <mutex>

#include "kf_gemm_common.h"

namespace kf {

constexpr int G3_BM = 256, G3_BN = 256, G3_BK = 64;
// the workgroup tile: BM rows x BN tokens, NW waves as 2 (rows) x NW / 2 (tokens).  Big = the 256 x 256 tile of the header (8 waves, 128 x 64 outputs each, 128 KiB of
// LDS: one workgroup per CU); Small = 128 x 128 on 4 waves (64 x 64 outputs each, 64 KiB: two workgroups per CU) for products with too few 256 x 256 tiles to fill the
// chip -- the o_proj / down projections of a 1-2 k token prompt (M = 1024: 32 big tiles), with the split-K form below on top.  K-contiguous operands only.
template <int BM_, int BN_, int NW_>
struct G3Cfg {
    static constexpr int BM = BM_, BN = BN_, NW = NW_, WN = NW_ / 2, MT = BM_ / 32, NT = BN_ / (16 * (NW_ / 2));
    static constexpr int NIA = BM_ / (8 * NW_), NIB = BN_ / (8 * NW_); /* global_load_lds instructions per wave, operand and k-step (8 rows of 128 bytes each) */
    static constexpr int STAGE = (BM_ + BN_) * G3_BK * 2, NTH = 64 * NW_, WGS_PER_CU = (160 * 1024) / (2 * STAGE);
};
using G3Big = G3Cfg<256, 256, 8>;
using G3Small = G3Cfg<128, 128, 4>;
// round 4: products whose 128 x 128 tiles do not fill the chip either (M = 1024 at 1-2 k tokens: 64-128 of them; split-K S = 4 measured 37.8 us against 27 plain: three
// 64 KiB partials per tile through memory) get MORE, SMALLER tiles instead of k-pieces: 64 x 128 (48 KiB of LDS: three workgroups per CU) and 64 x 64 (32 KiB: five)
using G3Mid = G3Cfg<64, 128, 4>;
using G3Tiny = G3Cfg<64, 64, 4>;
// 192 x 256: a product whose 256 x 256 tiles number 160-255 (gate | up of a 0.6B model at 2047 tokens: 24 x 8 = 192 tiles on 256 CUs) becomes 32 x 8 = 256 tiles of 3/4 the work
using G3Wide = G3Cfg<192, 256, 8>;

// one operand tile (256 rows x 2 BK bytes) = BK / 2 wave instructions of 1 KiB; wave `wid` issues BK / 16 of them.  BK = 64: 8 rows per instruction, chunk c of row r
// at position c ^ ((r >> 1) & 7); BK = 32: 16 rows per instruction (64-byte rows), chunk c at position c ^ ((r >> 2) & 3) -- either way a fragment read
// (16 rows x one 16-byte chunk) touches all 64 banks once.
template <int BK, int NI>
__device__ __forceinline__ const uint16_t* g3_src(const uint16_t* __restrict__ src, long long ld, int row0, int nrows, int i, int wid, int lane) {
    constexpr int CPR = BK / 8, RPI = 64 / CPR; /* chunks per row, rows per instruction */
    const int j = wid * NI + i;
    const int r = j * RPI + lane / CPR, p = lane % CPR, c = BK == 64 ? p ^ ((r >> 1) & 7) : p ^ ((r >> 2) & 3);
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1; /* rows past the end re-read the last row: their outputs are not stored */
    return src + (size_t)gr * ld + c * 8; /* + k0 per step */
}
template <int BK>
__device__ __forceinline__ bf16x8 g3_frag(const unsigned char* tile, int row, int c) {
    const int p = BK == 64 ? c ^ ((row >> 1) & 7) : c ^ ((row >> 2) & 3);
    return *reinterpret_cast<const bf16x8*>(tile + row * (2 * BK) + (p << 4));
}

// the same tile from a K-MAJOR operand (src[k][row], `row` contiguous: an activation or a weight as it lies in memory when the contraction runs over its ROWS --
// both GEMMs of SLP::Back): the LDS image is [BK k][ROWS rows] (512- or 256-byte k-rows), one wave instruction = two or four k-rows; its 32-byte blocks are XOR-swizzled by k & 3 and
// its 128-byte quarters by bit 3 of k, so that the transposing fragment read below (per 32 lanes: k-rows k .. k+3 and k+8 .. k+11, 32 bytes of each) touches all 64
// banks once (without the second term the two 16-lane halves meet in the same 32 banks: SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles, measured).
// ROWS = the tile's rows (256 or 128): a k-row of the image is 2 ROWS bytes = CPK chunks of 16 bytes, one wave instruction covers 64 / CPK k-rows, wave `wid` issues
// instructions NI wid .. NI wid + NI - 1 of the ROWS / 8
template <int BK, int ROWS, int NI>
__device__ __forceinline__ const uint16_t* g3_src_km(const uint16_t* __restrict__ src, long long ld, int row0, int nrows, int i, int wid, int lane) {
    constexpr int CPK = ROWS / 8, KPI = 64 / CPK;
    const int j = wid * NI + i;
    const int kr = j * KPI + lane / CPK, p = lane % CPK, c = p ^ ((kr & 3) << 1) ^ (((kr >> 3) & 1) << 3);
    int col = row0 + c * 8;
    col = col + 8 <= nrows ? col : (nrows - 8 > 0 ? nrows - 8 : 0); /* chunks past the end re-read the last whole chunk: their outputs are not stored (rows % 8 == 0) */
    return src + (size_t)kr * ld + col; /* + k0 * ld per step */
}
// fragment (8 consecutive k of row `row`, k = kbase .. kbase + 7) of a k-major tile: two ds_read_b64_tr_b16, each a 4 (k) x 16 (rows) block transposed across 16 lanes
// (lane i of the 16 receives D[(i >> 2) + 4 j][i & 3], scratch/dbg/ds_read_tr_probe.hip): lane l16 reads k-row kbase + (l16 >> 2), 8-byte piece l16 & 3 of the 16-row block
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <int ROWS>
__device__ __forceinline__ bf16x8 g3_frag_km(const unsigned char* tile, int rowblk16, int kbase, int l16) {
    bf16x4 h[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int k = kbase + 4 * t + (l16 >> 2);
        const int byte_in_row = (rowblk16 * 16 + 4 * (l16 & 3)) * 2, c = byte_in_row >> 4, p = c ^ ((k & 3) << 1) ^ (((k >> 3) & 1) << 3);
        h[t] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + k * (2 * ROWS) + p * 16 + (byte_in_row & 8)));
    }
    return bf16x8{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
}

// k-steps kt0 .. kt1 - 1 (of BK) of the output tile at (m0, t0) accumulated into acc (the caller zeroes it).  NST = 128 KiB / stage LDS buffers: the loads of step
// kt + NST - 1 are issued while step kt is multiplied and drained with a COUNTED s_waitcnt vmcnt + one raw s_barrier per step (a __syncthreads() would drain the loads
// in flight): BK = 64 -> 2 buffers; BK = 32 -> 4 buffers, three steps in flight (measured slower, see g3_go).  Also measured and dropped: the barrier moved into
// the middle of the step's MFMA stream (the last 16 MFMAs of a step held back behind it, the first fragments of the next step read under them): 979 TFLOP/s against
// 1000 on the forward shapes at 246 VGPRs, and the k-major forms spill; four waves of 128 x 128 outputs each (256 accumulator AGPRs, a third less LDS traffic per MFMA,
// the vendor library's shape): the compiler fills all 512 registers and still spills inside the loop, 228 TFLOP/s -- that form needs a hand-scheduled loop.
template <bool AKM, bool BKM, int BK, class C>
__device__ __forceinline__ void g3_mainloop(const GemmArgs& a, int m0, int t0, int kt0, int kt1, f32x4 (&acc)[C::MT][C::NT], unsigned char* smem_raw, int wid, int lane) {
    static_assert(BK == G3_BK, "two k-halves of 32 per step");
    constexpr int NST = 2, LPW = C::NIA + C::NIB; /* loads per wave and step */
    const int wm = wid / C::WN, wn = wid % C::WN;
    const uint16_t* const W = reinterpret_cast<const uint16_t*>(a.w);
    auto bufA = [&](int b) { return smem_raw + (size_t)b * C::STAGE; };
    auto bufB = [&](int b) { return smem_raw + (size_t)b * C::STAGE + C::BM * BK * 2; };
    // AKM: W is [K][M] with row stride a.ldr (re-used field: the residual is not served by the k-major forms); BKM: x is [K][n] with row stride a.ldx.
    // Per-lane source pointers are set up once; a step adds a uniform offset (a 64-bit multiply per load and step was a tenth of the loop's issue slots).
    const uint16_t *pa[C::NIA], *pb[C::NIB];
#pragma unroll
    for (int i = 0; i < C::NIA; i++) pa[i] = AKM ? g3_src_km<BK, C::BM, C::NIA>(W, a.ldr, m0, a.M, i, wid, lane) : g3_src<BK, C::NIA>(W, a.K, m0, a.M, i, wid, lane);
#pragma unroll
    for (int i = 0; i < C::NIB; i++) pb[i] = BKM ? g3_src_km<BK, C::BN, C::NIB>(a.x, a.ldx, t0, a.n, i, wid, lane) : g3_src<BK, C::NIB>(a.x, a.ldx, t0, a.n, i, wid, lane);
    const long long stepA = AKM ? (long long)BK * a.ldr : BK, stepB = BKM ? (long long)BK * a.ldx : BK;
    auto stage = [&](int kt, int b) {
        const long long oa = stepA * kt, ob = stepB * kt;
#pragma unroll
        for (int i = 0; i < C::NIA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[i] + oa),
                                             (__attribute__((address_space(3))) void*)(bufA(b) + (wid * C::NIA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < C::NIB; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[i] + ob),
                                             (__attribute__((address_space(3))) void*)(bufB(b) + (wid * C::NIB + i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < NST - 1; st++)
        if (kt0 + st < kt1) stage(kt0 + st, st);
    const int r16 = lane & 15, q4 = lane >> 4;
    // the steps are unrolled by the NST buffers so that every LDS address of a step is base + an immediate
    auto step = [&](int kt, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        const int left = kt1 - 1 - kt; /* steps issued after kt that may stay in flight */
        if (NST >= 4 && left >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
        else if (NST >= 3 && left >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); /* step kt has landed for every wave, and every wave is done reading the buffer of step kt - 1: the next loads overwrite that one */
        if (kt + NST - 1 < kt1) stage(kt + NST - 1, (cur + NST - 1) % NST);
#pragma unroll
        for (int kk = 0; kk < BK / 32; kk++) {
            bf16x8 af[C::MT], bfr[C::NT];
#pragma unroll
            for (int nt = 0; nt < C::NT; nt++) {
                if constexpr (BKM) bfr[nt] = g3_frag_km<C::BN>(bufB(cur), wn * C::NT + nt, kk * 32 + 8 * q4, r16);
                else bfr[nt] = g3_frag<BK>(bufB(cur), wn * (16 * C::NT) + nt * 16 + r16, kk * 4 + q4);
            }
#pragma unroll
            for (int mt = 0; mt < C::MT; mt++) {
                if constexpr (AKM) af[mt] = g3_frag_km<C::BM>(bufA(cur), wm * C::MT + mt, kk * 32 + 8 * q4, r16);
                else af[mt] = g3_frag<BK>(bufA(cur), wm * (16 * C::MT) + mt * 16 + r16, kk * 4 + q4);
            }
#pragma unroll
            for (int mt = 0; mt < C::MT; mt++)
#pragma unroll
                for (int nt = 0; nt < C::NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int kt = kt0; kt < kt1; kt += NST) {
        step(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < kt1) step(kt + 1, std::integral_constant<int, 1>{});
    }
    __builtin_amdgcn_s_barrier(); /* a following segment's first loads must not overtake the last reads */
}
// epilogue (gemm_epilogue's order): lane holds rows m .. m+3 of a 16 x 16 tile for token column r16
// ROPE::cuInfer on a 128-row tile that is one head (hd 128) of Q or K, in the epilogue of the stacked Q | K | V launch (the arithmetic of prep_head_cs, kf_attn_common.h: the
// projection rounded to bf16 as the plain epilogue stores it, fp64 sum of squares, s rounded to bf16, (x s) w rounded, rotate-half pairs (j, j + 64) with every product and the
// sum rounded once).  Wave (wm, wn) holds rows 64 wm .. + 63 x tokens 64 wn .. + 63: a token's 128 squares are summed over the 4 lanes of a column (row swaps) and the two wm
// waves (LDS), and a rotation pair's other element sits in the other wm wave at the SAME lane and register: one exchange through the (now free) stage buffers.
template <class C>
__device__ __forceinline__ void g3_epilogue_qkrope(const GemmArgs& a, int m0, int t0, const f32x4 (&acc)[C::MT][C::NT], int wid, int lane, unsigned char* smem_raw, uint16_t* ybase, long long ldy,
                                                   int mshift, const uint16_t* nw) {
    static_assert(C::BM == 128 && C::MT == 4 && C::WN == 2, "one head per tile: 2 x 2 waves of 64 rows x 64 tokens");
    const int wm = wid / C::WN, wn = wid % C::WN, r16 = lane & 15, q4 = lane >> 4;
    float x[C::MT][C::NT][4];
    double ss[C::NT];
#pragma unroll
    for (int nt = 0; nt < C::NT; nt++) {
        ss[nt] = 0.0;
#pragma unroll
        for (int mt = 0; mt < C::MT; mt++) {
            const float vv[4] = {acc[mt][nt].x, acc[mt][nt].y, acc[mt][nt].z, acc[mt][nt].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                x[mt][nt][j] = round_bf16(vv[j]); /* the value the plain epilogue stores (alpha 1, no bias) */
                ss[nt] = fma((double)x[mt][nt][j], (double)x[mt][nt][j], ss[nt]);
            }
        }
        ss[nt] = xsum32_d(xsum16_d(ss[nt])); /* the column's 4 lanes (l ^ 16, l ^ 32): this wave's 64 rows */
    }
    double* red = reinterpret_cast<double*>(smem_raw);                                  /* [2 wm][128 tokens] */
    uint32_t* xch = reinterpret_cast<uint32_t*>(smem_raw + 2 * 128 * sizeof(double));   /* [4 waves][32 bf16 pairs][64 lanes] */
    if (q4 == 0) {
#pragma unroll
        for (int nt = 0; nt < C::NT; nt++) red[wm * 128 + wn * 64 + nt * 16 + r16] = ss[nt];
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < C::NT; nt++) {
        const int tl = wn * 64 + nt * 16 + r16;
        float s = 1.0f;
        if (nw) s = round_bf16(1.0f / sqrtf((float)(red[tl] + red[128 + tl]) / 128.0f + a.qk_eps));
#pragma unroll
        for (int mt = 0; mt < C::MT; mt++) {
            if (nw) {
                const u32x2 w4 = *reinterpret_cast<const u32x2*>(nw + wm * 64 + mt * 16 + 4 * q4); /* this lane's four norm weights */
                x[mt][nt][0] = round_bf16(x[mt][nt][0] * s * bf_lo(w4.x)), x[mt][nt][1] = round_bf16(x[mt][nt][1] * s * bf_hi(w4.x));
                x[mt][nt][2] = round_bf16(x[mt][nt][2] * s * bf_lo(w4.y)), x[mt][nt][3] = round_bf16(x[mt][nt][3] * s * bf_hi(w4.y));
            }
            xch[((size_t)wid * 32 + (mt * C::NT + nt) * 2) * 64 + lane] = pack_bf16x2(x[mt][nt][0], x[mt][nt][1]);
            xch[((size_t)wid * 32 + (mt * C::NT + nt) * 2 + 1) * 64 + lane] = pack_bf16x2(x[mt][nt][2], x[mt][nt][3]);
        }
    }
    __syncthreads();
    const int pw = (1 - wm) * C::WN + wn; /* the wave that holds the other element of every pair */
#pragma unroll
    for (int mt = 0; mt < C::MT; mt++)
#pragma unroll
        for (int nt = 0; nt < C::NT; nt++) {
            const int tok = t0 + wn * 64 + nt * 16 + r16, jj = mt * 16 + 4 * q4; /* pair index 0 .. 63 of this lane's first row */
            if (tok >= a.n) continue;
            uint16_t o[4];
            const uint32_t op0 = xch[((size_t)pw * 32 + (mt * C::NT + nt) * 2) * 64 + lane], op1 = xch[((size_t)pw * 32 + (mt * C::NT + nt) * 2 + 1) * 64 + lane];
            const float oth[4] = {bf_lo(op0), bf_hi(op0), bf_lo(op1), bf_hi(op1)};
            f32x4 cs01 = f32x4{1.f, 0.f, 1.f, 0.f}, cs23 = cs01; /* (cos, sin) of the lane's four pairs jj .. jj + 3: 32 contiguous bytes of the table */
            if (a.rope_table) {
                const f32x4* tp = reinterpret_cast<const f32x4*>(a.rope_table + (size_t)(a.rope_pos0 + (a.rope_seq > 0 ? tok % a.rope_seq : tok)) * 128 + 2 * jj);
                cs01 = tp[0], cs23 = tp[1];
            }
            const float csv[8] = {cs01.x, cs01.y, cs01.z, cs01.w, cs23.x, cs23.y, cs23.z, cs23.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float other = oth[j], own = x[mt][nt][j];
                float v = own;
                if (a.rope_table) {
                    const float2 cs = float2{csv[2 * j], csv[2 * j + 1]};
                    if (wm == 0) { /* x_j c - x_{j+64} s */
                        const float p0 = own * cs.x, p1 = other * cs.y;
                        v = p0 - p1;
                    } else { /* x_j s + x_{j+64} c */
                        const float p0 = other * cs.y, p1 = own * cs.x;
                        v = p0 + p1;
                    }
                }
                o[j] = f2bf(v);
            }
            uint16_t* yp = ybase + (size_t)tok * ldy + (m0 - mshift) + wm * 64 + jj;
            *reinterpret_cast<u32x2*>(yp) = u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
        }
}

template <bool AKM, class C>
__device__ __forceinline__ void g3_epilogue(const GemmArgs& a, int m0, int t0, const f32x4 (&acc)[C::MT][C::NT], int wid, int lane, unsigned char* smem_raw = nullptr) {
    const int wm = wid / C::WN, wn = wid % C::WN, r16 = lane & 15, q4 = lane >> 4;
    if (a.swiglu) { /* Relu::Forw (CU_swiglu_v0, Activation.cu:85-93) on the two bf16-rounded projections, as swiglu_kernel: accumulator block 2i holds gate rows, block 2i + 1 the up rows of
                       the same 16 FFN rows (the operand was dequantised interleaved), same lane, same register */
        const int F = a.M >> 1;
#pragma unroll
        for (int mt = 0; mt < C::MT; mt += 2)
#pragma unroll
            for (int nt = 0; nt < C::NT; nt++) {
                const int tok = t0 + wn * (16 * C::NT) + nt * 16 + r16, f = ((m0 + wm * (16 * C::MT)) >> 1) + (mt >> 1) * 16 + 4 * q4;
                if (tok >= a.n || f >= F) continue;
                const float gv[4] = {acc[mt][nt].x, acc[mt][nt].y, acc[mt][nt].z, acc[mt][nt].w}, uv[4] = {acc[mt + 1][nt].x, acc[mt + 1][nt].y, acc[mt + 1][nt].z, acc[mt + 1][nt].w};
                uint16_t o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float g = round_bf16(gv[j]), u = round_bf16(uv[j]);
                    o[j] = f2bf((g * u) / (1.0f + kf_expf(-g)));
                }
                uint16_t* yp = a.y + (size_t)tok * a.ldy + f;
                if (f + 3 < F && ((a.ldy & 3) == 0)) {
                    *reinterpret_cast<u32x2*>(yp) = u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (f + j < F) yp[j] = o[j];
                }
            }
        return;
    }
    // up to three matrices stacked along M (Q | K | V or gate | up, each a multiple of the 256-row tile: a.rb_end[j] = cumulative rows / 32): the tile's rows go to
    // that matrix's own output (the K and V rows of a prompt land in the cache, Q in its buffer)
    uint16_t* ybase = a.y;
    long long ldy = a.ldy;
    int mshift = 0, Mlim = a.M;
    if (a.njobs > 1) {
        const int e0 = a.rb_end[0] * 32, e1 = a.rb_end[1] * 32;
        if (m0 >= e1 && a.njobs > 2) ybase = a.xy[1], ldy = a.xldy[1], mshift = e1, Mlim = a.rb_end[2] * 32;
        else if (m0 >= e0) ybase = a.xy[0], ldy = a.xldy[0], mshift = e0, Mlim = e1;
        else Mlim = e0;
    }
    if constexpr (!AKM && C::BM == 128) {
        if (a.qkrope && smem_raw) { /* workgroup-uniform: a tile lies inside one job */
            const int job = a.njobs > 1 ? (m0 >= a.rb_end[1] * 32 ? 2 : (m0 >= a.rb_end[0] * 32 ? 1 : 0)) : 0;
            if (job < 2) {
                g3_epilogue_qkrope<C>(a, m0, t0, acc, wid, lane, smem_raw, ybase, ldy, mshift, a.qk_norm[job]);
                return;
            }
        }
    }
    const bool vec_ok = ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(ybase) & 7) == 0);
#pragma unroll
    for (int mt = 0; mt < C::MT; mt++)
#pragma unroll
        for (int nt = 0; nt < C::NT; nt++) {
            const int tok = t0 + wn * (16 * C::NT) + nt * 16 + r16, m = m0 + wm * (16 * C::MT) + mt * 16 + 4 * q4;
            if (tok >= a.n || m >= Mlim) continue;
            uint16_t* yp = ybase + (size_t)tok * ldy + (m - mshift);
            const float vv[4] = {acc[mt][nt].x, acc[mt][nt].y, acc[mt][nt].z, acc[mt][nt].w};
            uint16_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = vv[j];
                o[j] = 0;
                if (m + j < Mlim) {
                    if (a.alpha != 1.0f) v = a.alpha * v;
                    if (a.beta != 0.0f) v = v + a.beta * bf2f(yp[j]);
                    if (a.bias) v = v + bf2f(a.bias[m + j]);
                    uint16_t qv = f2bf(v);
                    if (!AKM && a.residual) qv = f2bf(bf2f(a.residual[(size_t)tok * a.ldr + m + j]) + bf2f(qv));
                    o[j] = qv;
                }
            }
            if (vec_ok && m + 3 < Mlim) {
                *reinterpret_cast<u32x2*>(yp) = u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (m + j < Mlim) yp[j] = o[j];
            }
        }
}
// bijective XCD remap: the blocks with equal blockIdx % 8 (one XCD under round-robin placement) get consecutive logical indices
__device__ __forceinline__ int g3_remap(int orig, int nwg) {
    const int q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
}

template <bool AKM, bool BKM, int BK, class C>
__global__ void __launch_bounds__(C::NTH, C::WGS_PER_CU * C::NW / 4) gemm3_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nbx = (a.M + C::BM - 1) / C::BM, nby = (a.n + C::BN - 1) / C::BN, nwg = nbx * nby;
    const int wg = g3_remap(blockIdx.x, nwg);
    // (round 4: the other orientation -- row tiles slow, so that an XCD's L2 fetches all of x and an eighth of W when W is the larger operand -- measured on the 2047-token
    // prompt: gate | up 41.5 us either way, Q | K | V 34.1 vs 32.5: not the fabric traffic that bounds these launches)
    const int bx = wg % nbx, by = wg / nbx;
    const int m0 = bx * C::BM, t0 = by * C::BN;
    f32x4 acc[C::MT][C::NT];
#pragma unroll
    for (int i = 0; i < C::MT; i++)
#pragma unroll
        for (int j = 0; j < C::NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    g3_mainloop<AKM, BKM, BK, C>(a, m0, t0, 0, a.K / BK, acc, smem_raw, wid, lane);
    g3_epilogue<AKM, C>(a, m0, t0, acc, wid, lane, smem_raw);
}

// SPLIT-K forms for launches with fewer tiles than CUs (a weight gradient [OC, IC] is 49 .. 175 tiles, its contraction 8192 token rows long).  The pieces are cut so
// that the workgroups an XCD runs side by side stay at (nearly) the SAME k: they share operand strips in that XCD's L2.  (Ranges of a plain stream-K cut drift apart
// in k and every workgroup then streams its own strips through the fabric -- measured: 496 TFLOP/s.)
//   S >= 2 (2 P <= 256):  every tile's k-steps are cut in S equal pieces, P S workgroups; piece 0 owns the tile;
//   S == 1 (P < 256 < 2 P): workgroup t < P owns tile t and multiplies k-steps [0, kp), kp = nkt P / 256; the 256 - P helpers cut the tails [kp, nkt) of all tiles,
//                          laid end to end, into equal ranges (a helper's range covers pieces of 2 .. 5 tiles; neighbours start a few steps apart).  Owners and
//                          helpers finish together.
// A non-owner leaves each fp32 partial in a slot of `ws` with write-through stores and raises the slot's flag; the owner adds its tile's partials in slot order -- a
// fixed order: the result does not depend on timing -- and runs the epilogue.  a raised flag holds the launch's own value (g3_sk_epoch below: no clearing between launches); only owners wait, for pieces that
// were dispatched before them and wait for nobody.
struct G3SkArgs {
    float* ws;       /* [<= 256][256 * 256] */
    uint32_t* flags; /* [<= 256] */
    int P, S, kp, R;
    uint32_t epoch; /* the value a raised flag holds in THIS launch (g3_sk_epoch): no memset between launches */
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t g3_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
template <class C>
__device__ __forceinline__ void g3_publish(const G3SkArgs& s, int slot, const f32x4 (&acc)[C::MT][C::NT], int tid) {
    const __amdgpu_buffer_rsrc_t rs = g3_rsrc(s.ws + (size_t)slot * (C::BM * C::BN), C::BM * C::BN * 4);
#pragma unroll
    for (int i = 0; i < C::MT * C::NT; i++)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i / C::NT][i % C::NT]), rs, (i * C::NTH + tid) * 16, 0, 16 /* sc1 */);
    // the partial went out with write-through (sc1) stores and every lane has their acknowledgements (vmcnt 0) before the flag is raised -- itself a write-through
    // store.  A release fence here (buffer_wbl2) would write back every dirty line of this XCD's L2, the finished output tiles of its other workgroups included:
    // measured 37 vs 27 us on a 128-tile product.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32(s.epoch, g3_rsrc(s.flags + slot, 4), 0, 0, 16 /* sc1 */);
}
template <class C>
__device__ __forceinline__ void g3_collect(const G3SkArgs& s, int slot, f32x4 (&acc)[C::MT][C::NT], int tid) {
    if (tid == 0) {
        for (int spins = 0; spins < (1 << 24); spins++) {
            if (__builtin_amdgcn_raw_buffer_load_b32(g3_rsrc(s.flags + slot, 4), 0, 0, 16 /* sc1 */) == s.epoch) break; /* the partial is read with sc1 loads too: nothing cached to invalidate */
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = g3_rsrc(s.ws + (size_t)slot * (C::BM * C::BN), C::BM * C::BN * 4);
#pragma unroll
    for (int g = 0; g < C::MT * C::NT / 8; g++) { /* 8 loads in flight at a time: the accumulators hold half the register file */
        f32x4 pv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((g * 8 + i) * C::NTH + tid) * 16, 0, 16 /* sc1 */));
#pragma unroll
        for (int i = 0; i < 8; i++) acc[(g * 8 + i) / C::NT][(g * 8 + i) % C::NT] += pv[i];
        asm volatile("" ::: "memory");
    }
}
template <bool AKM, bool BKM, int BK, class C>
__global__ void __launch_bounds__(C::NTH, C::WGS_PER_CU * C::NW / 4) gemm3_sk_kernel(const GemmArgs a, const G3SkArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nbx = (a.M + C::BM - 1) / C::BM, nkt = a.K / BK;
    // Roles by DISPATCH order: every non-owner piece of a tile has a smaller blockIdx than the tile's owner, and non-owners wait for nobody -- so when an owner runs, the
    // pieces it waits for have at least started, whatever else occupies the chip (a collective on another stream): the waits end, no residency assumption.
    // S >= 2: blockIdx group g = blockIdx / P holds piece S - 1 - g of every tile (the owners, piece 0, are the last group); S == 1: the helpers first, then the owners.
    // Inside a group the XCD remap keeps neighbouring tiles on one XCD (the group offset only rotates the XCD labels).
    const int nhelp = (int)gridDim.x - s.P; /* S == 1 */
    const bool helper = s.S < 2 && (int)blockIdx.x < nhelp;
    const int w = s.S >= 2 ? (s.S - 1 - (int)blockIdx.x / s.P) * s.P + g3_remap((int)blockIdx.x % s.P, s.P)
                           : (helper ? s.P + g3_remap((int)blockIdx.x, nhelp) : g3_remap((int)blockIdx.x - nhelp, s.P));
    // tail mode: tail length L, T = P L tail steps in all, cut into ranges of R = s.R steps (the host's ceil(T / helpers)); 32-bit: the host checks T < 2^31
    const int L = nkt - s.kp, T = s.P * L, R = s.R;
    int hb = 0, he = 1; /* a helper's range of tail steps; the other roles make one pass */
    if (helper) hb = (w - s.P) * R, he = hb + R < T ? hb + R : T;
    for (int it = hb; it < he;) {
        int tile, k0, k1, slot = 0, c0 = 0, c1 = 0; /* c0 .. c1 - 1: the slots an owner collects */
        bool owner;
        if (s.S >= 2) {
            const int sp = w / s.P;
            tile = w % s.P, k0 = (int)((long long)nkt * sp / s.S), k1 = (int)((long long)nkt * (sp + 1) / s.S);
            owner = sp == 0, slot = tile * (s.S - 1) + sp - 1, c0 = tile * (s.S - 1), c1 = c0 + s.S - 1;
        } else if (!helper) {
            tile = w, k0 = 0, k1 = s.kp, owner = true;
            const int lo = tile * L; /* helpers lo / R .. (lo + L - 1) / R meet this tile's tail; the piece of helper h in tile t has slot h + t */
            c0 = lo / R + tile, c1 = (lo + L - 1) / R + tile + 1;
        } else {
            tile = it / L;
            const int o = it - tile * L, len = L - o < he - it ? L - o : he - it;
            k0 = s.kp + o, k1 = k0 + len, owner = false, slot = (w - s.P) + tile; /* pieces in range order: one slot each, < P + helpers */
        }
        c0 = __builtin_amdgcn_readfirstlane(c0), c1 = __builtin_amdgcn_readfirstlane(c1), slot = __builtin_amdgcn_readfirstlane(slot);
        const int bx = tile % nbx, by = tile / nbx, m0 = bx * C::BM, t0 = by * C::BN;
        f32x4 acc[C::MT][C::NT];
#pragma unroll
        for (int i = 0; i < C::MT; i++)
#pragma unroll
            for (int j = 0; j < C::NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        g3_mainloop<AKM, BKM, BK, C>(a, m0, t0, k0, k1, acc, smem_raw, wid, lane);
        if (!owner) {
            g3_publish<C>(s, slot, acc, tid);
        } else {
            for (int c = c0; c < c1; c++) g3_collect<C>(s, c, acc, tid);
            g3_epilogue<AKM, C>(a, m0, t0, acc, wid, lane, smem_raw);
        }
        it += helper ? k1 - k0 : 1;
    }
}

// The flag value of one split-K launch.  Launches outside a capture take a fresh value each (a process-wide counter from 2 up), so the flags of the launch before need
// no clearing -- the memset was a dependent 2-3 us node in front of every such product; a flag area seen for the first time is cleared once.  A launch that is being
// CAPTURED is replayed with the arguments it was recorded with: it keeps the clearing node and the value 1.
constexpr size_t G3_SK_FLAG_BYTES = 8192; /* up to 2048 resident workgroups (64 x 64 tiles, eight per CU) */
static std::mutex g_sk_mu;
static uint32_t g_sk_epoch = 1;
static const void* g_sk_seen[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
static int g3_sk_epoch(hipStream_t st, uint32_t* flags, int n, uint32_t* epoch) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) return KF_HIP_CHECK;
    std::lock_guard<std::mutex> lk(g_sk_mu);
    bool seen = false;
    for (int i = 0; i < 8; i++) seen = seen || g_sk_seen[i] == flags;
    if (cs != hipStreamCaptureStatusNone || !seen) {
        if (hipMemsetAsync(flags, 0, n * sizeof(uint32_t), st) != hipSuccess) return KF_HIP_CHECK;
    }
    if (cs != hipStreamCaptureStatusNone) {
        *epoch = 1;
        return KF_OK;
    }
    if (!seen) {
        static int next = 0;
        g_sk_seen[next] = flags, next = (next + 1) & 7;
    }
    if (++g_sk_epoch < 2) g_sk_epoch = 2;
    *epoch = g_sk_epoch;
    return KF_OK;
}

// KF_OK launched, 1 = not for this kernel (the caller's other tile kernels take the shape), < 0 error.  bf16 "weights" only: quantised ones are dequantised first.
// BK = 32 (4 LDS buffers, three steps in flight) was measured 8-10 % slower than BK = 64 (2 buffers) on every shape, forward and backward: the per-step costs
// (barrier, counted wait, loop) double, and the loop is not waiting for memory.  Only BK = 64 is instantiated.
template <bool AKM, bool BKM, class C>
static int g3_go_c(hipStream_t st, const GemmArgs& a, long nwg, void* ws, size_t ws_bytes, long min_plain) {
    constexpr int BK = G3_BK, SMEM = 2 * C::STAGE;
    static int attr_set = 0;
    if (!attr_set && SMEM > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)gemm3_kernel<AKM, BKM, BK, C>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return KF_HIP_CHECK;
        if (hipFuncSetAttribute((const void*)gemm3_sk_kernel<AKM, BKM, BK, C>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return KF_HIP_CHECK;
        attr_set = 1;
    }
    // split-K when the tiles are fewer than 4/5 of the resident workgroups, the caller lent the workspace, and the pieces stay >= 512 deep in k
    const int G = 256 * C::WGS_PER_CU, nkt = a.K / BK, min_steps = 512 / BK;
    if (ws && ws_bytes >= gemm3_sk_ws_bytes() && 5 * nwg < 4 * G) {
        G3SkArgs s;
        // the flags live in the LAST 8 KiB of the lent workspace whatever the tile configuration (behind the partial tiles of THIS configuration they moved with it,
        // and another configuration's fp32 partials could overwrite words already marked "seen" with values that pass for an epoch)
        static_assert((size_t)256 * C::WGS_PER_CU * C::BM * C::BN * 4 <= (size_t)256 * G3_BM * G3_BN * 4 && 256 * C::WGS_PER_CU * 4 <= G3_SK_FLAG_BYTES, "partials and flags fit the lent workspace");
        s.ws = (float*)ws, s.flags = (uint32_t*)((char*)ws + gemm3_sk_ws_bytes() - G3_SK_FLAG_BYTES);
        s.P = (int)nwg, s.S = G / s.P, s.kp = 0, s.R = 1;
        int nlaunch;
        if (s.S >= 2) {
            while (s.S > 1 && nkt / s.S < min_steps) s.S--;
            nlaunch = s.P * s.S;
        } else {
            s.kp = (int)(((long long)nkt * s.P + G / 2) / G);
            nlaunch = G;
            const long long T = (long long)s.P * (nkt - s.kp);
            s.R = (int)((T + (G - s.P) - 1) / (G - s.P));
            if (nkt - s.kp < 1 || nkt < 4 * min_steps || T * 2 >= (1LL << 31)) s.S = 0;
        }
        if (s.S >= 1 && (s.S >= 2 || s.kp > 0)) {
            if (g3_sk_epoch(st, s.flags, G, &s.epoch) != KF_OK) return KF_HIP_CHECK;
            hipLaunchKernelGGL((gemm3_sk_kernel<AKM, BKM, BK, C>), dim3(nlaunch), dim3(C::NTH), SMEM, st, a, s);
            return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
        }
    }
    if (nwg < min_plain) return 1;
    hipLaunchKernelGGL((gemm3_kernel<AKM, BKM, BK, C>), dim3((unsigned)nwg), dim3(C::NTH), SMEM, st, a);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}
template <bool AKM, bool BKM>
static int g3_go(hipStream_t st, const GemmArgs& a, long nwg, void* ws, size_t ws_bytes) {
    return g3_go_c<AKM, BKM, G3Big>(st, a, nwg, ws, ws_bytes, 64);
}
size_t gemm3_sk_ws_bytes() { return (size_t)256 * G3_BM * G3_BN * 4 + G3_SK_FLAG_BYTES; }
int gemm3_launch(hipStream_t st, int fmt, const GemmArgs& a, void* ws, size_t ws_bytes) {
    if (fmt != FMT_BF16 || a.K % G3_BK != 0 || a.K < G3_BK) return 1;
    if ((a.ldx & 7) != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15) != 0 || (reinterpret_cast<uintptr_t>(a.w) & 15) != 0 || (a.K & 7) != 0) return 1;
    if (a.n >= G3_BN && a.M >= G3_BM) {
        const long nwg = (long)((a.M + G3_BM - 1) / G3_BM) * ((a.n + G3_BN - 1) / G3_BN);
        // (M 1024 x K 3072 at 8192 rows, 128 big tiles: 83 -> 57 us = 890 TFLOP/s on the small ones; from 400 big tiles up the big tile is 8-12 % faster)
        if (nwg >= 160) return g3_go<false, false>(st, a, nwg, nullptr, 0); /* fewer big tiles than that: four times as many small ones fill the chip better */
    }
    // 128 x 128 tiles, two workgroups per CU: bf16 products of 1-4 k rows whose M is 1024-3072 (M 1024 x K 2048 at 2048 rows: 36.9 -> 27.4 us, 3072 x 1024: 35.6 -> 23.9;
    // 4096 rows: 583-656 TFLOP/s).  For QUANTISED weights of that size dequantise + this kernel only ties with the in-register-unpack kernels (the 5 us dequantise pass
    // and, with split-K, the flag memset eat the gain): kf_linear keeps those on kf_gemm.hip.
    if (a.n < 128 || a.M < 128) return 1;
    const long nwg = (long)((a.M + 127) / 128) * ((a.n + 127) / 128);
    if (nwg < 32 && (g_knobs.g3_tiles < 3 || (long)((a.M + 63) / 64) * ((a.n + 63) / 64) < 64)) return 1;
    if (nwg < 256 && g_knobs.g3_tiles >= 1 && !a.swiglu && !a.qkrope) { /* fewer 128 x 128 tiles than CUs: twice or four times as many smaller ones, no k-pieces */
        // with the caller's workspace these go in k-pieces when even the small tiles leave CUs idle (1024 rows x K 2048-3072 at 512 tokens: 128 tiles of 64 x 64, each a
        // 32-48 step k-loop of ~0.7 us per step on its own CU: 31 us; a 16 KiB partial per piece is cheap where the 64 KiB ones of the 128 x 128 tile were not)
        const long nmid = (long)((a.M + 63) / 64) * ((a.n + 127) / 128);
        if (nmid >= g_knobs.g3_mid_min || g_knobs.g3_tiles < 3) return nmid >= 64 ? g3_go_c<false, false, G3Mid>(st, a, nmid, ws, ws_bytes, 64) : 1;
        return g3_go_c<false, false, G3Tiny>(st, a, (long)((a.M + 63) / 64) * ((a.n + 63) / 64), ws, ws_bytes, 64);
    }
    return g3_go_c<false, false, G3Small>(st, a, nwg, nullptr, 0, 128);
}
// y[n, M] = alpha * sum_k B(k, tok) A(k, m) + beta * y (+ bias) with either operand stored K-MAJOR: akm: A = w[K][lda] (element (k, m) at k * lda + m), else w[M][K];
// bkm: B = x[K][ldb], else x[n][ldb].  The two GEMMs of SLP::Back without a transpose of anything:  delta[n, IC] = deltaIn[n, OC] . W[OC, IC]  (A = W k-major),
// gW[OC, IC] += deltaIn^T . inp  (A = inp[n][IC] k-major, B = deltaIn[n][OC] k-major, contraction over the n token rows).  1 = shape not served.
// ws (gemm3_sk_ws_bytes() bytes, 16-byte aligned) lends the stream-K form its partial-tile slots; NULL = one workgroup per tile only.
int gemm3_km_launch(hipStream_t st, const uint16_t* A, long long lda, bool akm, const uint16_t* B, long long ldb, bool bkm, int n, int M, int K, uint16_t* y, long long ldy,
                    const uint16_t* bias, float alpha, float beta, void* ws, size_t ws_bytes) {
    if (K % G3_BK != 0 || K < G3_BK || n < G3_BN || M < G3_BM || (M & 7) || (n & 7) || (lda & 7) || (ldb & 7)) return 1;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return 1;
    const long nwg = (long)((M + G3_BM - 1) / G3_BM) * ((n + G3_BN - 1) / G3_BN);
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.w = reinterpret_cast<const unsigned char*>(A), a.M = M, a.K = K, a.x = B, a.ldx = ldb, a.n = n, a.y = y, a.ldy = ldy, a.bias = bias, a.alpha = alpha, a.beta = beta;
    a.ldr = lda; /* the k-major A operand's row stride */
    if (!akm && lda != K) return 1; /* the row-major A form reads rows of K */
    // big tiles that fill less than 4/5 of the CUs: 128 x 128 tiles instead when they make (nearly) whole rounds of the 512 resident workgroups, or -- with the
    // workspace -- their split-K form (partials of 64 KiB instead of 256 KiB, twice the workgroups)
    // (measured: Qwen3-0.6B training step 99.3 -> 95.7 ms with the first rule alone, -> 87.2 ms with both; GPT2-1558M 166.0 -> 163.7 -> 161.2 ms)
    const long nws = (long)((M + 127) / 128) * ((n + 127) / 128), rounds = (nws + 511) / 512;
    const bool small = 5 * nwg < 4 * 256 && akm && (20 * nws >= 17 * rounds * 512 || (ws && ws_bytes >= gemm3_sk_ws_bytes() && 5 * nws < 4 * 512));
    if (small) return bkm ? g3_go_c<true, true, G3Small>(st, a, nws, ws, ws_bytes, 1) : g3_go_c<true, false, G3Small>(st, a, nws, ws, ws_bytes, 1);
    if (akm && bkm) return g3_go<true, true>(st, a, nwg, ws, ws_bytes);
    if (akm) return g3_go<true, false>(st, a, nwg, ws, ws_bytes);
    if (bkm) return g3_go<false, true>(st, a, nwg, ws, ws_bytes);
    return g3_go<false, false>(st, a, nwg, ws, ws_bytes);
}

// up to three bf16 matrices stacked along M in ONE contiguous buffer (the caller dequantised them back to back: Q | K | V, or gate | up), each a multiple of 256
// rows, multiplied by the same x in one launch; matrix j's rows go to y[j] (row stride M[j]).  1 = shape not served.
// rope != NULL (n_w == 3: Q | K | V, head_dim 128): q/k-norm + RoPE in the epilogue (GemmArgs::qkrope); served on the 128 x 128 tile only -- 1 otherwise, the caller then
// runs the plain stacked launch and qknorm_rope_launch
int gemm3_multi_launch(hipStream_t st, int n_w, const uint16_t* Wcat, const int* M, int K, const uint16_t* x, long long ldx, int n, uint16_t* const* y, const G3Rope* rope) {
    if (n_w < 2 || n_w > 3 || K % G3_BK != 0 || K < G3_BK || n < G3_BN || (ldx & 7) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(Wcat) & 15)) return 1;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    int tot = 0;
    for (int j = 0; j < 3; j++) {
        if (j < n_w) {
            if (M[j] < G3_BM || M[j] % G3_BM != 0) return 1;
            tot += M[j];
        }
        a.rb_end[j] = tot / 32;
        if (j >= 1 && j < n_w) a.xy[j - 1] = y[j], a.xldy[j - 1] = M[j];
    }
    a.njobs = n_w;
    a.w = reinterpret_cast<const unsigned char*>(Wcat), a.M = tot, a.K = K, a.x = x, a.ldx = ldx, a.n = n, a.y = y[0], a.ldy = M[0], a.alpha = 1.0f, a.beta = 0.0f;
    const long nwg = (long)(tot / G3_BM) * ((n + G3_BN - 1) / G3_BN);
    if (nwg < 64 && (long)(tot / 128) * ((n + 127) / 128) < 64) return 1;
    // fewer than 160 big tiles (Q | K | V of a 0.6B model at 2047 tokens: 128): four times as many 128 x 128 tiles fill the chip better (2047-token prompt 8.20 -> 7.84 ms;
    // gate | up, 192 big tiles, is better left on them: 7.93 ms with both small)
    if (rope) {
        if (n_w != 3 || M[0] % 128 != 0 || M[1] % 128 != 0 || ((M[0] | M[1] | M[2]) & 3) || (reinterpret_cast<uintptr_t>(y[0]) & 7) || (reinterpret_cast<uintptr_t>(y[1]) & 7)) return 1;
        a.qkrope = 1, a.qk_norm[0] = rope->wq, a.qk_norm[1] = rope->wk, a.rope_table = rope->table, a.rope_pos0 = rope->pos0, a.rope_seq = rope->seq_len, a.qk_eps = rope->eps;
        return g3_go_c<false, false, G3Small>(st, a, (long)(tot / 128) * ((n + 127) / 128), nullptr, 0, 1); /* always the head-sized tile */
    }
    if (nwg < 160) return g3_go_c<false, false, G3Small>(st, a, (long)(tot / 128) * ((n + 127) / 128), nullptr, 0, 1);
    return g3_go<false, false>(st, a, nwg, nullptr, 0);
}

// gate | up dequantised INTERLEAVED in blocks of 16 rows (dequant_launch ilv_n = 2) x the same x, SwiGLU in the epilogue: act[n, ffn] in one launch (the stacked route
// wrote both projections and ran swiglu_kernel over them: 12 us and 38 MB of traffic per layer of a 2047-token prompt).  1 = shape not served.
int gemm3_swiglu_launch(hipStream_t st, const uint16_t* Wilv, int ffn, int K, const uint16_t* x, long long ldx, int n, uint16_t* act) {
    const int M = 2 * ffn;
    if (ffn % 16 != 0 || M % G3_BM != 0 || K % G3_BK != 0 || K < G3_BK || n < G3_BN || (ldx & 7) || (ffn & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(Wilv) & 15) ||
        (reinterpret_cast<uintptr_t>(act) & 7))
        return 1;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.njobs = 1, a.swiglu = 1;
    a.w = reinterpret_cast<const unsigned char*>(Wilv), a.M = M, a.K = K, a.x = x, a.ldx = ldx, a.n = n, a.y = act, a.ldy = ffn, a.alpha = 1.0f, a.beta = 0.0f;
    const long nwg = (long)(M / G3_BM) * ((n + G3_BN - 1) / G3_BN);
    if (nwg < 64 && (long)(M / 128) * ((n + 127) / 128) < 64) return 1;
    const long nws = (long)(M / 128) * ((n + 127) / 128), nww = (long)(M / 192) * ((n + G3_BN - 1) / G3_BN);
    const bool wide_ok = g_knobs.g3_wide && M % 192 == 0 && nww <= 256;
    // fewer than 160 big tiles: four times as many 128 x 128 ones -- unless those spill into a second round of the 512 resident workgroups and the 192 x 256 ones make one round
    if (nwg < 160 && !(wide_ok && nws > 512 && nww >= 160)) return g3_go_c<false, false, G3Small>(st, a, nws, nullptr, 0, 1);
    if (wide_ok && nwg < 256) return g3_go_c<false, false, G3Wide>(st, a, nww, nullptr, 0, 1);
    return g3_go<false, false>(st, a, nwg, nullptr, 0);
}

}  // namespace kf
