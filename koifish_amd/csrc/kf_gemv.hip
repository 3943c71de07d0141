// kf_gemv.hip -- fused PackedQ unpack + mat-vec for decode (nTok = 1), gfx950 / wave64.
//
// Replaces GTensor::GetDataX (dequantise the whole weight to bf16 in gBUFF->tmpTernary, quantizer.cu:249-392)
// followed by cuBLASLt (gemm.cu:93-214): the packed stream is read ONCE, 16 bytes per lane, fully coalesced,
// dequantised in registers with the reference's bf16-stepwise arithmetic (T.cu:274) and contracted against the
// activation held in LDS.  HBM-bound: algorithmic bytes = packed data + zero/step (+ x, y).
//
// Data mapping.  W[M,K] row-major flattened is a stream of 16-byte blocks (one Packed128 for 4/2/1-bit; 8 bf16;
// 16 f8).  EPB = elements per block; a row has nBlk = K/EPB blocks.  LPR lanes (a power of two <= 64) walk one
// row, RPS = 64/LPR rows are processed side by side by one wave ("slot"), ITERS = ceil(nBlk/LPR) loads per row.
// Each wave owns SPW consecutive slots and keeps G of them in flight.  x sits in LDS as 16-byte chunks laid
// out [chunk j of block][block column] so that consecutive lanes read consecutive 16-byte words (no bank
// conflicts for ds_read_b128).
#include <stdlib.h>

#include "kf_gemv_blocks.h"

// This file is compiled twice: as it stands (the v_dot2c_f32_bf16 forms) and through kf_gemv_canon.hip with KF_GEMV_CANON = 1 (the canonical order of
// oracle/kf_oracle.c section 4c: two v_fma_f32 per weight pair, every output bit reproducible with fmaf on the host).  Same kernels, same geometry, same
// launcher; gemv_launch picks by GemvLaunch::canon (the context's flag, kf_set_canonical).
#ifndef KF_GEMV_CANON
#define KF_GEMV_CANON 0
#endif

namespace kf {

constexpr bool GEMV_CANON = KF_GEMV_CANON != 0;
#if !KF_GEMV_CANON
Knobs g_knobs;
#endif

template <int G, bool PAIRED, bool LUT>
struct Batch {
    u32x4 w[G];
    u32x4 w2[PAIRED ? G : 1];
    u32x4 ta[LUT ? G : 1], tb[LUT ? G : 1];                       /* row codebook (FMT_Q4R) */
    u32x4 ta2[LUT && PAIRED ? G : 1], tb2[LUT && PAIRED ? G : 1];
    // zero / step stay raw bf16 bits until the block is multiplied: converted when loaded, the shift makes the wave wait for the loads it has
    // just issued (s_waitcnt vmcnt right behind the prefetch) instead of overlapping them with the current batch's arithmetic
    uint16_t st[G], ze[G];
    uint16_t st2[PAIRED ? G : 1], ze2[PAIRED ? G : 1];
};

// LDS: x as u32x4 chunks [XCH][nBlk] (K*2 bytes) | 256 B reduction scratch.
// Each wave keeps two batches of G blocks in flight: the first batch is issued BEFORE the x prologue so that the
// weight stream's HBM latency overlaps the (dependent) activation load + norm.
// ONEJOB: a launch with a single matrix (o_proj, down_proj, LM head, sparse rows) never reads the descriptors of jobs 1 and 2: kernel arguments are fetched ahead of the
// first load, and every one a launch touches is on its critical path
// XF (canonical 4-bit forms only, chosen by the launcher when K * 4 bytes of LDS leave the occupancy alone): x is staged as fp32 chunks [8][nBlk] and multiplied through
// BlockDotF (the engine's form): a product is two conversions of the weight pair + one v_pk_fma_f32 instead of four conversions + one -- same chains, same bits
// XF2 (XF of a row too long for that: the 25600-wide down_proj of Qwen3-32B, one row slot per wave): the fp32 chunks of HALF the block columns at a time -- the iterations
// of the first half run against the first window, then the workgroup restages and the same chains go on over the second (weights stay in flight across the two barriers)
template <int FMT, int G, int MODE, bool SPARSE, bool ONEJOB, bool CANON, bool XF_ = false, bool XF2_ = false>
__global__ void __launch_bounds__(256) gemv_kernel(const GemvArgs a) {
    using BD = BlockDot<FMT, CANON>;
    constexpr bool PAIRED = (MODE == GEMV_PAIRED), LUT = (FMT == FMT_Q4R), XF = XF_ && CANON && (FMT == FMT_Q4 || FMT == FMT_Q4P);
    constexpr bool XF2 = XF2_ && XF && G == 1 && !PAIRED && !SPARSE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);
    constexpr int RD = 4; /* XF2: a ring of four steps in flight per wave (25 steps per row behind ONE step of prefetch left the launch latency-bound: 1.7 TB/s) */
    const int it_half = XF2 ? (a.iters >> 1) / RD * RD : a.iters;        /* iterations against the first window: whole ring rounds */
    const int wcols = XF2 ? (a.iters - it_half) << a.lpr_log2 : a.nBlk;   /* block columns of the (larger, second) window = the chunk stride of the staged activations */
    const int wcol0 = it_half << a.lpr_log2;                              /* first block column of the second window */
    double* red = reinterpret_cast<double*>(smem_raw + (XF2 ? (size_t)wcols * 128 : (size_t)a.K * (XF ? 4 : 2)));

    const int tid = threadIdx.x, lane = tid & 63, wave_in_blk = tid >> 6;
    const int nBlk = a.nBlk, iters = a.iters;
    if constexpr (FMT == FMT_Q2T) { /* selector table: entry B, dword p = bytes {2q, 2q+1, 2q', 2q'+1}, q / q' = the levels of elements 2p, 2p+1 of byte B */
        uint32_t e[2];
#pragma unroll
        for (int p = 0; p < 2; p++) e[p] = 0x01000100u + 0x0202u * ((tid >> (6 - 4 * p)) & 3u) + 0x02020000u * ((tid >> (4 - 4 * p)) & 3u);
        reinterpret_cast<u32x2*>(xs + nBlk * 8 + 16)[tid] = u32x2{e[0], e[1]};
    }
    if constexpr (FMT == FMT_Q1T) { /* selector table: entry B, dword p = bytes {2a, 2a+1, 2b, 2b+1}, a / b = bits 7-2p / 6-2p of B (elements 2p, 2p+1) */
        uint32_t e[4];
#pragma unroll
        for (int p = 0; p < 4; p++) e[p] = 0x01000100u + 0x0202u * ((tid >> (7 - 2 * p)) & 1u) + 0x02020000u * ((tid >> (6 - 2 * p)) & 1u);
        (xs + nBlk * 16 + 16)[tid] = u32x4{e[0], e[1], e[2], e[3]}; /* 256 threads, 256 entries; visible after the prologue's barrier */
    }
    const int LPR = 1 << a.lpr_log2, RPS = 64 >> a.lpr_log2;
    const int sub = lane >> a.lpr_log2, ll = lane & (LPR - 1);
    const long gwave = (long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(wave_in_blk); /* wave-uniform: the slot range and step count stay scalar */
    const long s_begin = gwave * a.spw;
    long s_end = s_begin + a.spw;
    if (s_end > a.total_slots) s_end = a.total_slots;
    const int nbatch = s_end > s_begin ? (int)((s_end - s_begin + G - 1) / G) : 0;
    const int nsteps = nbatch * iters;

    // Every wave works inside ONE job (the launcher pads each job's slot range to a multiple of spw), so the job's fields are
    // selected once, with constant indices into the kernel arguments (SGPRs), and stay scalar for the whole loop.  Indexing
    // a.job[] with a run-time value inside the loop would turn each access into a load from the kernarg segment whose wait
    // serialises the weight stream.
    int jx = 0;
    if constexpr (!ONEJOB) {
        if (a.njobs > 1 && s_begin >= a.job[1].slot0) jx = 1;
        if (a.njobs > 2 && s_begin >= a.job[2].slot0) jx = 2;
        jx = __builtin_amdgcn_readfirstlane(jx);
    }
#define JF(f) (ONEJOB ? a.job[0].f : (jx == 0 ? a.job[0].f : (jx == 1 ? a.job[1].f : a.job[2].f)))
    const u32x4* const jw = reinterpret_cast<const u32x4*>(JF(w));
    const u32x4* const jw2 = reinterpret_cast<const u32x4*>(a.job[1].w);
    const uint16_t* const jstep = JF(step);
    const uint16_t* const jzero = JF(zero);
    uint16_t* const jy = JF(y);
    const long long jystride = JF(y_pos_stride);
    const int jM = JF(M), jslot0 = JF(slot0);
    const float jqb = (float)JF(qBias), jqb2 = (float)a.job[1].qBias;
#undef JF
    const int gshift = a.gshift;
    auto slot = [&](long s, int& row) -> bool {
        row = (int)(s - jslot0) * RPS + sub;
        return (s < s_end) && (row < jM);
    };
    // Two load policies.  LAT (one slot per wave: the short, latency-bound launches of decode): unconditional loads from clamped
    // (row, column) -- a lane outside the matrix re-reads a valid block and is masked when the block is multiplied -- because loads under
    // a lane condition make the number of loads in flight path-dependent and every wait behind them a drain (vmcnt(0)), including the
    // wait for x, which is requested FIRST so that its staging overlaps the weights' HBM latency.  Long launches (G > 1) are bound by
    // the dequant arithmetic and hide latency with resident waves: they keep the masked loads (no per-step mask arithmetic).
    constexpr bool LAT = (G == 1);
    auto load = [&](int bi, int it, Batch<G, PAIRED, LUT>& b) {
        const long s0 = s_begin + (long)bi * G;
        int col = it * LPR + ll;
        const bool col_ok = col < nBlk;
        col = col_ok ? col : nBlk - 1;
#pragma unroll
        for (int g = 0; g < G; g++) {
            int row;
            const bool ok = slot(s0 + g, row) && col_ok;
            if constexpr (LAT) {
                row = row < jM ? row : jM - 1;
                row = row > 0 ? row : 0;
            } else {
                b.w[g] = u32x4{0, 0, 0, 0};
                b.st[g] = b.ze[g] = 0;
                if (PAIRED) b.w2[g] = u32x4{0, 0, 0, 0}, b.st2[g] = b.ze2[g] = 0;
                if constexpr (LUT) {
                    b.ta[g] = b.tb[g] = u32x4{0, 0, 0, 0};
                    if constexpr (PAIRED) b.ta2[g] = b.tb2[g] = u32x4{0, 0, 0, 0};
                }
            }
            if (LAT || ok) {
                if constexpr (SPARSE) row = a.row_map[row]; /* sparse forward: the slot's row is the row-th hot row (one dependent, wave-uniform-per-group load).  A template
                                                              parameter: as a run-time branch its join carried an s_waitcnt vmcnt(0) that drained the weight stream of every launch */
                const uint32_t bidx = (uint32_t)row * (uint32_t)nBlk + (uint32_t)col; /* < 2^32 blocks = 64 GiB per tensor */
                b.w[g] = ld_nt(jw + bidx);
                if (PAIRED) b.w2[g] = ld_nt(jw2 + bidx);
                if (BD::HAS_GAMA) {
                    const uint32_t gi = bidx >> gshift; /* group = element / lGroup, lGroup / EPB a power of two */
                    b.st[g] = jstep[gi], b.ze[g] = jzero[gi];
                    if (PAIRED) b.st2[g] = a.job[1].step[gi], b.ze2[g] = a.job[1].zero[gi];
                }
                if constexpr (LUT) { /* job.zero carries the table base: 16 bf16 per row */
                    const u32x4* lt = reinterpret_cast<const u32x4*>(jzero) + 2 * (size_t)row;
                    b.ta[g] = lt[0], b.tb[g] = lt[1];
                    if constexpr (PAIRED) {
                        const u32x4* lt2 = reinterpret_cast<const u32x4*>(a.job[1].zero) + 2 * (size_t)row;
                        b.ta2[g] = lt2[0], b.tb2[g] = lt2[1];
                    }
                }
            }
        }
    };

    // STREAM (long dense launches, G > 1): the same blocks through buffer loads.  The G rows of a batch lie a constant number of bytes apart, so
    // ONE lane offset serves all of them (the row stride rides in the instruction's scalar offset), rows past the matrix and columns past the row
    // fall outside the buffer and read as zero (no lane branches, no zero fill), and the zero / step words come the same way.  The launcher
    // sets stream_ok when every offset fits 31 bits and a group never straddles two rows.
    constexpr bool STREAM = !LAT && !SPARSE && !LUT;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_w, rs_w2, rs_st, rs_ze, rs_st2, rs_ze2;
    [[maybe_unused]] uint32_t wbytes = 0, gbytes = 0, gstride_w = 0, gstride_g = 0;
    if constexpr (STREAM) {
        wbytes = (uint32_t)jM * (uint32_t)nBlk * 16u, gbytes = ((uint32_t)jM * (uint32_t)nBlk >> gshift) * 2u;
        gstride_w = (uint32_t)RPS * (uint32_t)nBlk * 16u, gstride_g = ((uint32_t)RPS * (uint32_t)nBlk >> gshift) * 2u;
        rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(jw), 0, (int)wbytes, 0x00020000);
        if constexpr (PAIRED) rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(jw2), 0, (int)wbytes, 0x00020000);
        if constexpr (BD::HAS_GAMA) {
            rs_st = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(jstep), 0, (int)gbytes, 0x00020000);
            rs_ze = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(jzero), 0, (int)gbytes, 0x00020000);
            if constexpr (PAIRED) {
                rs_st2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.job[1].step), 0, (int)gbytes, 0x00020000);
                rs_ze2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.job[1].zero), 0, (int)gbytes, 0x00020000);
            }
        }
    }
    auto sload = [&](int bi, int it, Batch<G, PAIRED, LUT>& b) {
        if constexpr (STREAM) {
            const int col = it * LPR + ll;
            const uint32_t row0 = (uint32_t)((int)(s_begin - jslot0) + bi * G) * (uint32_t)RPS + (uint32_t)sub;
            const uint32_t bidx = row0 * (uint32_t)nBlk + (uint32_t)col;
            const bool col_ok = col < nBlk;
            const uint32_t vo = col_ok ? bidx * 16u : wbytes, go = col_ok ? (bidx >> gshift) * 2u : gbytes;
#pragma unroll
            for (int g = 0; g < G; g++) {
                b.w[g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo, g * gstride_w, 2 /* nt */));
                if constexpr (PAIRED) b.w2[g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, vo, g * gstride_w, 2));
                if constexpr (BD::HAS_GAMA) {
                    b.st[g] = __builtin_amdgcn_raw_buffer_load_b16(rs_st, go, g * gstride_g, 0), b.ze[g] = __builtin_amdgcn_raw_buffer_load_b16(rs_ze, go, g * gstride_g, 0);
                    if constexpr (PAIRED)
                        b.st2[g] = __builtin_amdgcn_raw_buffer_load_b16(rs_st2, go, g * gstride_g, 0), b.ze2[g] = __builtin_amdgcn_raw_buffer_load_b16(rs_ze2, go, g * gstride_g, 0);
                }
            }
        }
    };
    const bool stream = STREAM && a.stream_ok;

    // x (and the norm weight) of vectors up to 4096 elements: <= 2 chunks of 8 per thread, requested before the weights
    const int nch = a.K >> 3;
    const bool xreg = LAT && nch <= 512, has_norm = a.norm_w != nullptr;
    const bool h0 = tid < nch, h1 = tid + 256 < nch;
    u32x4 r0 = u32x4{0, 0, 0, 0}, r1 = r0, n0 = r0, n1 = r0;
    if (xreg) {
        const size_t c0 = h0 ? tid : 0, c1 = h1 ? tid + 256 : 0;
        r0 = *reinterpret_cast<const u32x4*>(a.x + c0 * 8), r1 = *reinterpret_cast<const u32x4*>(a.x + c1 * 8);
        if (has_norm) n0 = *reinterpret_cast<const u32x4*>(a.norm_w + c0 * 8), n1 = *reinterpret_cast<const u32x4*>(a.norm_w + c1 * 8);
    }

    Batch<G, PAIRED, LUT> cur, nxt;
    [[maybe_unused]] Batch<G, PAIRED, LUT> ring[XF2 ? RD : 1];
    if constexpr (XF2) {
#pragma unroll
        for (int d = 0; d < RD; d++) load(0, d < iters ? d : iters - 1, ring[d]);
    } else if (stream) {
        if (nsteps > 0) sload(0, 0, cur);
    } else if (LAT || nsteps > 0) {
        load(0, 0, cur); /* LAT: waves without work re-read row 0 */
    }
    const int pos = a.d_pos ? *a.d_pos : a.pos;

    // ---- prologue: stage x into LDS as packed bf16 chunks (XF: the same elements widened to fp32, two chunks of four)
    {
        constexpr int XCH = BD::XCH;
        auto put = [&](int c, int j, u32x4 o) {
            if constexpr (XF) {
                xs[(2 * j) * wcols + c] = u32x4{o.x << 16, o.x & 0xffff0000u, o.y << 16, o.y & 0xffff0000u};
                xs[(2 * j + 1) * wcols + c] = u32x4{o.z << 16, o.z & 0xffff0000u, o.w << 16, o.w & 0xffff0000u};
            } else {
                xs[j * nBlk + c] = o;
            }
        };
        if (xreg) {
            // RMSNorm prologue, one pass (rms_norm_kernel, layernorm.cuh:800-847): fp64 sum of squares over the workgroup, then the
            // normalised chunks go to LDS; without a norm weight the chunks go to LDS as they are.
            const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            uint32_t ow[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            if (has_norm) {
                const uint32_t ww[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
                double ss0 = 0.0, ss1 = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double lo = (double)bf_lo(rw[k]), hi = (double)bf_hi(rw[k]), lo1 = (double)bf_lo(rw[4 + k]), hi1 = (double)bf_hi(rw[4 + k]);
                    ss0 = fma(lo, lo, ss0), ss0 = fma(hi, hi, ss0);
                    ss1 = fma(lo1, lo1, ss1), ss1 = fma(hi1, hi1, ss1);
                }
                double ss = (h0 ? ss0 : 0.0) + (h1 ? ss1 : 0.0); /* clamped lanes hold a copy of chunk 0 */
                ss = wave_sum_f64_fast(ss);
                if (lane == 0) red[wave_in_blk] = ss;
                __syncthreads();
                const double tot = (red[0] + red[1]) + (red[2] + red[3]);
                const float mul = 1.0f / sqrtf(fmaf((float)tot, a.inv_dim, a.eps));
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float v0 = (bf_lo(rw[k]) * mul) * bf_lo(ww[k]), v1 = (bf_hi(rw[k]) * mul) * bf_hi(ww[k]);
                    ow[k] = pack_bf16x2(v0, v1);
                }
            }
            if (h0) {
                const int c = tid / XCH, j = tid - c * XCH;
                put(c, j, u32x4{ow[0], ow[1], ow[2], ow[3]});
            }
            if (h1) {
                const int e8 = tid + 256, c = e8 / XCH, j = e8 - c * XCH;
                put(c, j, u32x4{ow[4], ow[5], ow[6], ow[7]});
            }
        } else {
            float mul = 1.0f;
            if (a.norm_w) { /* large K: two passes */
                double ss = block_sumsq_bf16(a.x, a.K, red);
                float val = fmaf((float)ss, a.inv_dim, a.eps);
                mul = 1.0f / sqrtf(val);
            }
            // element e of block column c, chunk j (e = c*EPB + j*8 + i)  ->  LDS chunk (j*nBlk + c)
            const int nch_w = XF2 ? wcol0 * XCH : nch; /* XF2: the first window (the launcher takes this form only without a norm) */
            for (int e8 = tid; e8 < nch_w; e8 += blockDim.x) {
                const int c = e8 / XCH, j = e8 - c * XCH;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(a.x + (size_t)e8 * 8);
                u32x4 o = raw;
                if (a.norm_w) {
                    const u32x4 nw = *reinterpret_cast<const u32x4*>(a.norm_w + (size_t)e8 * 8);
                    const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w}, ww[4] = {nw.x, nw.y, nw.z, nw.w};
                    uint32_t ow[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float v0 = (bf_lo(rw[k]) * mul) * bf_lo(ww[k]), v1 = (bf_hi(rw[k]) * mul) * bf_hi(ww[k]);
                        ow[k] = pack_bf16x2(v0, v1);
                    }
                    o.x = ow[0], o.y = ow[1], o.z = ow[2], o.w = ow[3];
                }
                put(c, j, o);
            }
        }
        __syncthreads();
    }

    // ---- main: pipelined over (batch, iteration) steps
    float best_v = -__builtin_inff();
    int best_i = 0x7fffffff;
    using Acc = typename BD::Acc;
    Acc acc[G], acc2[PAIRED ? G : 1]; /* per-lane chains (canonical: an even and an odd one; four such pairs for 1-bit blocks) */
    float sum[G], sum2[PAIRED ? G : 1];
    int bi = 0, it = 0;       // the step being computed
    int nbi = 0, nit = 0;     // the step being loaded
    auto compute = [&](const Batch<G, PAIRED, LUT>& bt) {
        if (it == 0) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                acc[g] = Acc{};
                if (PAIRED) acc2[g] = Acc{};
            }
        }
        {
            int col = it * LPR + ll;
            const bool col_ok = col < nBlk;
            if (!col_ok) col = nBlk - 1; /* keep the LDS reads in range */
#pragma unroll
            for (int g = 0; g < G; g++) {
                int row;
                const bool ok = !LAT || (slot(s_begin + (long)bi * G + g, row) && col_ok); /* masked loads carry zero weights */
                if constexpr (LUT) {
                    const Acc r = BD::run_lut(bt.w[g], xs, col, nBlk, bt.ta[g], bt.tb[g], acc[g]);
                    acc[g] = acc_pick(ok, r, acc[g]);
                    if constexpr (PAIRED) {
                        const Acc r2 = BD::run_lut(bt.w2[g], xs, col, nBlk, bt.ta2[g], bt.tb2[g], acc2[g]);
                        acc2[g] = acc_pick(ok, r2, acc2[g]);
                    }
                } else {
                    const float st = bf2f(bt.st[g]);
                    Acc r;
                    if constexpr (XF2) r = BlockDotF<FMT>::run(bt.w[g], reinterpret_cast<const f32x4*>(xs), it >= it_half ? col - wcol0 : col, wcols, st, bf2f(bt.ze[g]), -(jqb * st), acc[g]);
                    else if constexpr (XF) r = BlockDotF<FMT>::run(bt.w[g], reinterpret_cast<const f32x4*>(xs), col, nBlk, st, bf2f(bt.ze[g]), -(jqb * st), acc[g]);
                    else r = BD::run(bt.w[g], xs, col, nBlk, st, bf2f(bt.ze[g]), -(jqb * st), acc[g]);
                    acc[g] = acc_pick(ok, r, acc[g]);
                    if (PAIRED) {
                        const float st2 = bf2f(bt.st2[g]);
                        Acc r2;
                        if constexpr (XF) r2 = BlockDotF<FMT>::run(bt.w2[g], reinterpret_cast<const f32x4*>(xs), col, nBlk, st2, bf2f(bt.ze2[g]), -(jqb2 * st2), acc2[g]);
                        else r2 = BD::run(bt.w2[g], xs, col, nBlk, st2, bf2f(bt.ze2[g]), -(jqb2 * st2), acc2[g]);
                        acc2[g] = acc_pick(ok, r2, acc2[g]);
                    }
                }
            }
        }
        if (it == iters - 1) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                sum[g] = group_sum(acc_join(acc[g]), a.lpr_log2);
                if (PAIRED) sum2[g] = group_sum(acc_join(acc2[g]), a.lpr_log2);
            }
            if (ll == 0) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    int r;
                    if (!slot(s_begin + (long)bi * G + g, r)) continue;
                    if constexpr (SPARSE) r = a.row_map[r];
                    uint16_t* y = jy + (size_t)pos * jystride;
                    float v = sum[g];
                    if (PAIRED) {
                        // SwiGLU of the two bf16-rounded projections (CU_swiglu_v0, Activation.cu:85-93)
                        const float gt = round_bf16(v), up = round_bf16(sum2[g]);
                        y[r] = f2bf((gt * up) / (1.0f + kf_expf(-gt)));
                        continue;
                    }
                    if (a.tp) { /* tensor-parallel push: one 8-byte {value | tag} granule into this rank's slot of every rank's receive area (kf_tp.hip) */
                        const TpPushDev& t = *a.tp;
                        const unsigned long long gr = ((unsigned long long)(*t.step * t.per_step + t.index + 1u) << 32) | __float_as_uint(v);
                        for (int p = 0; p < t.world; p++) __hip_atomic_store(t.peer[p] + r, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        continue;
                    }
                    if (a.yf) { /* un-rounded fp32 row dots: tensor-parallel partial sums (column-split o_proj / down_proj) */
                        a.yf[r] = v;
                        continue;
                    }
                    if (a.alpha != 1.0f) v = a.alpha * v;
                    if (a.beta != 0.0f) v = v + a.beta * bf2f(y[r]);
                    if (a.bias) v = v + bf2f(a.bias[r]);
                    uint16_t o = f2bf(v);
                    if (a.residual) o = f2bf(bf2f(a.residual[r]) + bf2f(o)); /* CU_add3: bf16(x + bf16(W.x)) */
                    y[r] = o;
                    if (MODE == GEMV_ARGMAX) {
                        const float fv = bf2f(o);
                        if (fv > best_v || (fv == best_v && r < best_i)) best_v = fv, best_i = r;
                    }
                }
            }
        }
        if (++it == iters) it = 0, bi++;
    };
    if (stream) { /* two batches ping-pong: no register copies between steps */
        for (int k = 0; k < nsteps; k += 2) {
            if (k + 1 < nsteps) {
                if (++nit == iters) nit = 0, nbi++;
                sload(nbi, nit, nxt);
            }
            compute(cur);
            if (k + 1 >= nsteps) break;
            if (k + 2 < nsteps) {
                if (++nit == iters) nit = 0, nbi++;
                sload(nbi, nit, cur);
            }
            compute(nxt);
        }
    } else if constexpr (XF2) { /* one row slot per wave (nsteps = iters, or 0 for a wave past the last slot: it still meets the two barriers) */
        auto run = [&](int k0, int k1) { /* k0: a multiple of RD */
            for (int k = k0; k < k1; k += RD) {
#pragma unroll
                for (int d = 0; d < RD; d++) {
                    if (k + d < k1) compute(ring[d]);
                    load(0, k + d + RD < iters ? k + d + RD : iters - 1, ring[d]); /* unconditional (a request behind a branch turns the waits into drains); past the row: its last step again */
                }
            }
        };
        run(0, nsteps < it_half ? nsteps : it_half);
        __syncthreads(); /* every wave has read the first window for the last time */
        {
            constexpr int XCH = BD::XCH;
            for (int e8 = wcol0 * XCH + tid; e8 < nch; e8 += blockDim.x) {
                const int c = e8 / XCH - wcol0, j = e8 % XCH;
                const u32x4 o = *reinterpret_cast<const u32x4*>(a.x + (size_t)e8 * 8);
                xs[(2 * j) * wcols + c] = u32x4{o.x << 16, o.x & 0xffff0000u, o.y << 16, o.y & 0xffff0000u};
                xs[(2 * j + 1) * wcols + c] = u32x4{o.z << 16, o.z & 0xffff0000u, o.w << 16, o.w & 0xffff0000u};
            }
        }
        __syncthreads();
        run(it_half, nsteps);
    } else {
        for (int k = 0; k < nsteps; k++) {
            if (k + 1 < nsteps) {
                if (++nit == iters) nit = 0, nbi++;
                load(nbi, nit, nxt);
            }
            compute(cur);
            cur = nxt;
        }
    }

    if (MODE == GEMV_ARGMAX) {
        // first-maximum over this workgroup's rows (sample_argmax, GoPT.cpp:602-612)
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            float ov = __shfl_xor(best_v, m, 64);
            int oi = __shfl_xor(best_i, m, 64);
            if (ov > best_v || (ov == best_v && oi < best_i)) best_v = ov, best_i = oi;
        }
        float* rv = reinterpret_cast<float*>(red);
        int* ri = reinterpret_cast<int*>(rv + 16);
        __syncthreads();
        if (lane == 0) rv[wave_in_blk] = best_v, ri[wave_in_blk] = best_i;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < (int)(blockDim.x >> 6); w++)
                if (rv[w] > best_v || (rv[w] == best_v && ri[w] < best_i)) best_v = rv[w], best_i = ri[w];
            a.amax_val[blockIdx.x] = best_v;
            a.amax_idx[blockIdx.x] = best_i;
        }
    }
}

#if !KF_GEMV_CANON
// Final pick over the per-workgroup partial maxima, then the decode-state update for graph replay.
__global__ void __launch_bounds__(256) argmax_finish_kernel(const float* val, const int* idx, int n, int32_t* d_argmax, int32_t* d_state,
                                                            int32_t* d_tokens_out) {
    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    // all of a thread's partials are requested before the first compare (n <= KF_MAX_ARGMAX_PARTIALS = 16 per thread): one memory latency instead of 16
    constexpr int PER = KF_MAX_ARGMAX_PARTIALS / 256;
    float v[PER];
    int ix[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = threadIdx.x + k * 256;
        const int ic = i < n ? i : 0;
        v[k] = val[ic], ix[k] = idx[ic];
        if (i >= n) v[k] = -__builtin_inff(), ix[k] = 0x7fffffff;
    }
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (v[k] > bv || (v[k] == bv && ix[k] < bi)) bv = v[k], bi = ix[k];
    for (int i = threadIdx.x + PER * 256; i < n; i += blockDim.x) { /* never taken: n <= KF_MAX_ARGMAX_PARTIALS */
        const float w = val[i];
        const int jx = idx[i];
        if (w > bv || (w == bv && jx < bi)) bv = w, bi = jx;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        float ov = __shfl_xor(bv, m, 64);
        int oi = __shfl_xor(bi, m, 64);
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = bv, si[threadIdx.x >> 6] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) bv = sv[w], bi = si[w];
        if (d_argmax) *d_argmax = bi;
        if (d_state) {
            const int p = d_state[1];
            if (d_tokens_out) d_tokens_out[p] = bi;
            d_state[0] = bi;
            d_state[1] = p + 1;
        }
    }
}

#endif
// ------------------------------------------------------------------------------------------------ launcher
static int fmt_of(int type) {
    switch (type) {
        case KF_BF16: return FMT_BF16;
        case KF_F8E5M2: return FMT_F8;
        case KF_Q4: return FMT_Q4;
        case KF_T_SIGN: return FMT_Q2;
        case KF_BOOL1: case KF_T_BINARY: return FMT_Q1;
        default: return -1;
    }
}
static int fmt_of_w(const kf_weight* w) {
    if (is_row_lut(w)) return (w->type == KF_Q4 && w->quant == KF_QUANT_ROW_LUT) ? FMT_Q4R : -1;
    return fmt_of(w->type);
}
#if !KF_GEMV_CANON
int gemv_fmt_of(const kf_weight* w) { return fmt_of_w(w); }
#endif
static int epb_of(int fmt) {
    switch (fmt) {
        case FMT_BF16: return 8;
        case FMT_F8: return 16;
        case FMT_Q4: case FMT_Q4R: return 32;
        case FMT_Q2: return 64;
        default: return 128;
    }
}

#if !KF_GEMV_CANON
// Lanes per row of a launch (shared with the persistent decode engine, kf_engine.hip, so that both sum a row in the same order).
// LPR: the largest power of two <= 64 dividing nBlk when that is >= 16 (no idle lanes), else the largest power of two <= min(64, nBlk)
// with the row tail masked.  Short launches (fewer waves than the chip has SIMDs) are bound by the per-step dequant arithmetic of the
// longest wave, not by lanes: a row gets more lanes, even with the tail of the last step masked, while that shortens the step count
// (down_proj of the 0.6B model, 1024 x 3072: 512 waves x 3 steps -> 1024 waves x 2 steps).  `rows` = the rows of every job of the launch.
// the same per storage: 1-bit rows take the rule's figure for K / 32 "virtual" blocks -- the four dwords of a 128-element block are what four neighbouring lanes of the persistent
// engine multiply side by side (canonical order: a chain pair per dword position, oracle/kf_oracle.c section 4c) -- divided by four: logical lanes, one whole block each, here
int gemv_lpr_log2_fmt(int fmt, int K, long rows) {
    if (fmt == FMT_Q1 || fmt == FMT_Q1T || fmt == FMT_Q2 || fmt == FMT_Q2T) { /* 4 / 2 sub-blocks of 32 per block */
        const int l = gemv_lpr_log2(K / 32, rows) - ((fmt == FMT_Q1 || fmt == FMT_Q1T) ? 2 : 1);
        return l > 0 ? l : 0;
    }
    return gemv_lpr_log2(K / epb_of(fmt), rows);
}
int gemv_lpr_log2(int nBlk, long rows) {
    int lpr_log2 = 6;
    while (lpr_log2 > 0 && (nBlk % (1 << lpr_log2)) != 0) lpr_log2--;
    if ((1 << lpr_log2) < 16) {
        lpr_log2 = 6;
        while ((1 << lpr_log2) > nBlk) lpr_log2--;
    }
    while (lpr_log2 < 6 && nBlk > (1 << lpr_log2) && (rows << lpr_log2) / 64 < 1024 &&
           (nBlk + (2 << lpr_log2) - 1) / (2 << lpr_log2) < (nBlk + (1 << lpr_log2) - 1) / (1 << lpr_log2))
        lpr_log2++;
    return lpr_log2;
}
#endif

template <int FMT, int MODE, bool SPARSE, bool ONEJOB, bool XF>
static void launch_x(const GemvArgs& a, int G, dim3 grid, size_t smem, hipStream_t st) {
    if (G >= 4)
        hipLaunchKernelGGL((gemv_kernel<FMT, 4, MODE, SPARSE, ONEJOB, GEMV_CANON, XF>), grid, dim3(256), smem, st, a);
    else if (G == 2)
        hipLaunchKernelGGL((gemv_kernel<FMT, 2, MODE, SPARSE, ONEJOB, GEMV_CANON, XF>), grid, dim3(256), smem, st, a);
    else
        hipLaunchKernelGGL((gemv_kernel<FMT, 1, MODE, SPARSE, ONEJOB, GEMV_CANON, XF>), grid, dim3(256), smem, st, a);
}
// the canonical 4-bit forms take x as fp32 in LDS while 4 K + 256 bytes stay inside 48 KiB (K <= 12224: every matrix of the 0.6B model, Q | K | V, o_proj and gate | up of
// Qwen3-32B and all of its TP = 8 shards; not its 25600-wide down_proj, where 100 KiB per workgroup would halve the resident waves)
constexpr size_t GEMV_XF_MAX_SMEM = 48 * 1024;
template <int FMT, int MODE, bool SPARSE, bool ONEJOB>
static void launch_j(const GemvArgs& a, int G, dim3 grid, size_t smem, hipStream_t st) {
    if constexpr (GEMV_CANON && (FMT == FMT_Q4 || FMT == FMT_Q4P)) {
        const size_t smem_f = smem + (size_t)a.K * 2;
        if (smem_f <= GEMV_XF_MAX_SMEM) {
            launch_x<FMT, MODE, SPARSE, ONEJOB, true>(a, G, grid, smem_f, st);
            return;
        }
        if constexpr (MODE == GEMV_PLAIN && !SPARSE && ONEJOB) { /* longer rows, one row slot per wave, no norm in front: half the block columns at a time (XF2) */
            const int it_half = (a.iters >> 1) / 4 * 4; /* = the kernel's: whole rounds of its four-step ring against the first window */
            const size_t smem_2 = (size_t)((a.iters - it_half) << a.lpr_log2) * 128 + 256;
            if (G == 1 && a.spw == 1 && !a.norm_w && it_half >= 4 && smem_2 <= 54 * 1024 && g_knobs.gemv_xf2 != 0) { /* 54 KiB: three workgroups per CU */
                hipLaunchKernelGGL((gemv_kernel<FMT, 1, MODE, SPARSE, ONEJOB, GEMV_CANON, true, true>), grid, dim3(256), smem_2, st, a);
                return;
            }
        }
    }
    launch_x<FMT, MODE, SPARSE, ONEJOB, false>(a, G, grid, smem, st);
}
template <int FMT, int MODE, bool SPARSE>
static void launch_g(const GemvArgs& a, int G, dim3 grid, size_t smem, hipStream_t st) {
    // paired launches read job 1 by name; the arg-max head and the sparse forms are single-matrix by construction
    if constexpr (MODE == GEMV_PAIRED) {
        launch_j<FMT, MODE, SPARSE, false>(a, G, grid, smem, st);
    } else if constexpr (MODE == GEMV_ARGMAX || SPARSE) {
        launch_j<FMT, MODE, SPARSE, true>(a, G, grid, smem, st);
    } else {
        if (a.njobs == 1)
            launch_j<FMT, MODE, SPARSE, true>(a, G, grid, smem, st);
        else
            launch_j<FMT, MODE, SPARSE, false>(a, G, grid, smem, st);
    }
}
template <int FMT>
static void launch_m(const GemvArgs& a, int mode, int G, dim3 grid, size_t smem, hipStream_t st) {
    if (a.row_map) { /* the sparse forward: kf_linear_masked (plain) and kf_norm_gateup_swiglu_masked (paired) */
        if (mode == GEMV_PAIRED)
            launch_g<FMT, GEMV_PAIRED, true>(a, G, grid, smem, st);
        else
            launch_g<FMT, GEMV_PLAIN, true>(a, G, grid, smem, st);
        return;
    }
    if (mode == GEMV_PAIRED)
        launch_g<FMT, GEMV_PAIRED, false>(a, G, grid, smem, st);
    else if (mode == GEMV_ARGMAX)
        launch_g<FMT, GEMV_ARGMAX, false>(a, G, grid, smem, st);
    else
        launch_g<FMT, GEMV_PLAIN, false>(a, G, grid, smem, st);
}

#if KF_GEMV_CANON
int gemv_launch_canon(hipStream_t st, GemvLaunch& L) {
#else
int gemv_launch(hipStream_t st, GemvLaunch& L) { return L.canon ? gemv_launch_canon(st, L) : gemv_launch_dot2(st, L); }
int gemv_launch_dot2(hipStream_t st, GemvLaunch& L) {
#endif
    GemvArgs& a = L.args;
    const kf_weight* w0 = L.w[0];
    const int fmt = fmt_of_w(w0);
    if (fmt < 0) return KF_UNSUPPORTED_DATATYPE;
    const int K = w0->ne1, epb = epb_of(fmt);
    if (K % epb != 0 || K % 8 != 0) return KF_INVALID_ARGS;
    const int nBlk = K / epb;
    long rows_all = 0;
    for (int j = 0; j < L.n; j++)
        if (!(L.mode == GEMV_PAIRED && j == 1)) rows_all += L.w[j]->ne0;
    const int lpr_log2 = gemv_lpr_log2_fmt(fmt, K, rows_all);
    const int LPR = 1 << lpr_log2, RPS = 64 / LPR;
    a.K = K, a.nBlk = nBlk, a.lpr_log2 = lpr_log2, a.iters = (nBlk + LPR - 1) / LPR;
    a.inv_dim = 1.0f / (float)K;
    a.njobs = L.n;
    long rows_slots[3] = {0, 0, 0}, raw_slots = 0;
    for (int j = 0; j < L.n; j++) {
        const kf_weight* w = L.w[j];
        if (fmt_of_w(w) != fmt || w->ne1 != K) return KF_INVALID_ARGS;
        if (w->qzeros || w->qscales) return KF_UNSUPPORTED_DATATYPE; /* AutoAWQ layout: kf_linear only */
        if (((uintptr_t)w->data & 15) != 0) return KF_BLAS_UNALIGN;
        if ((unsigned long long)w->ne0 * (unsigned long long)nBlk >= (1ull << 32)) return KF_INVALID_ARGS;
        GemvJob& jb = a.job[j];
        jb.w = w->data;
        jb.zero = jb.step = nullptr;
        if (fmt == FMT_Q4R) {
            if (!w->gama) return KF_QUANT_ERR;
            jb.zero = w->gama + w->ne0 + w->ne1; /* gama_T(LUT): 16 entries per row behind the row / column scales */
            if (((uintptr_t)jb.zero & 15) != 0) return KF_BLAS_UNALIGN;
        } else if (fmt >= FMT_Q4) {
            if (!w->gama || w->lGroup <= 0 || (w->lGroup % epb) != 0 || ((long)w->ne0 * w->ne1) % w->lGroup != 0) return KF_QUANT_ERR;
            jb.zero = w->gama + w->ne0 + w->ne1; /* gama_T(ZERO), GTensor.cpp:456-510 */
            jb.step = jb.zero + (size_t)w->ne0 * w->ne1 / w->lGroup;
            a.lGroup = w->lGroup;
            const int bpg = w->lGroup / epb; /* blocks per group: must be a power of two (128-element groups always are) */
            if (bpg < 1 || (bpg & (bpg - 1)) != 0) return KF_QUANT_ERR;
            a.gshift = __builtin_ctz(bpg);
        }
        jb.M = a.row_map ? L.n_hot : w->ne0; /* sparse forward: only the hot rows get slots */
        jb.qBias = w->qBias;
        if (L.mode == GEMV_PAIRED && j == 1) {
            if (w->ne0 != L.w[0]->ne0) return KF_INVALID_ARGS;
            continue; /* job 1 rides on job 0's slots */
        }
        rows_slots[j] = (jb.M + RPS - 1) / RPS;
        raw_slots += rows_slots[j];
    }
    if (L.mode == GEMV_PAIRED) a.njobs = 1;
    // one wave per spw slots; aim for ~4096 waves (16 per CU) on large matrices, never fewer than one slot each
    // small problems: one slot per wave (latency-bound, as many waves as slots); large ones: several rounds of resident waves so that
    // memory waits of one wave are covered by the dequant arithmetic of the others (measured: 25600x5120 q4 41 -> 33 us; the Qwen3-32B
    // q/k/v and o_proj launches, 22-28 MB each, 21.8 -> 19.1 and 16.1 -> 13.2 us; the 0.6B launches stay below the threshold)
    const long blocks_all = raw_slots * (long)nBlk * (64 / (1 << lpr_log2));
    long target_waves = L.target_waves > 0 ? L.target_waves : (blocks_all >= (1L << 19) ? 16384 : 4096);
    // the largest launches (>= 4 M blocks: the 25600-row FFN matrices of Qwen3-32B): two slots per wave through the buffer-load form, half the workgroups to
    // start and half the x staging (25600 x 5120: 27.6 -> 26.4 us, A/B in one run; smaller launches lose more from the coarser tail than they gain)
    if (L.target_waves <= 0 && blocks_all >= 4000000L && !a.row_map && fmt != FMT_Q4R) target_waves = 8192;
    if (g_knobs.gemv_waves > 0) target_waves = g_knobs.gemv_waves;
    long spw = (raw_slots + target_waves - 1) / target_waves;
    if (spw < 1) spw = 1;
    int G = spw >= 4 ? 4 : (spw >= 2 ? 2 : 1);
    if (L.mode == GEMV_PAIRED && G > 2) G = 2; /* two weight streams per slot: keep register pressure down */
    spw = (spw + G - 1) / G * G;
    a.spw = (int)spw;
    // every job starts on a wave boundary, so a wave never straddles two jobs
    long slots = 0;
    for (int j = 0; j < a.njobs; j++) {
        a.job[j].slot0 = (int)slots;
        slots += (rows_slots[j] + spw - 1) / spw * spw;
    }
    for (int j = a.njobs; j < 3; j++) a.job[j].slot0 = 0x7fffffff;
    a.total_slots = (int)slots;
    // buffer-load form of the long launches (gemv_kernel, STREAM): every byte offset a wave can form -- rows of the padded slot range included -- below
    // 2^31, and groups that never straddle two rows (then the G rows of a batch are a constant number of groups apart)
    a.stream_ok = 0;
    if (G > 1 && !a.row_map && fmt != FMT_Q4R) {
        long max_rows = 0;
        for (int j = 0; j < L.n; j++) max_rows = a.job[j].M > max_rows ? a.job[j].M : max_rows;
        const unsigned long long reach = ((unsigned long long)max_rows + (unsigned long long)(spw + G) * RPS) * nBlk * 16ull;
        const bool groups_ok = fmt < FMT_Q4 || (K % a.lGroup) == 0;
        if (reach < (1ull << 31) && groups_ok) a.stream_ok = 1;
        a.stream_ok = a.stream_ok && g_knobs.gemv_stream != 0;
    }
    const long waves = (slots + spw - 1) / spw;
    const int blocks = (int)((waves + 3) / 4);
    if (L.mode == GEMV_ARGMAX && blocks > KF_MAX_ARGMAX_PARTIALS) return KF_INTERNAL_ERR;
    const size_t smem = (size_t)K * 2 + 256;
    if (smem > 160 * 1024) return KF_INVALID_ARGS;
    dim3 grid(blocks);
    switch (fmt) {
        case FMT_BF16: launch_m<FMT_BF16>(a, L.mode, G, grid, smem, st); break;
        case FMT_F8: launch_m<FMT_F8>(a, L.mode, G, grid, smem, st); break;
        case FMT_Q4: {
            // table-lookup form whenever a 128-weight group is exactly one aligned quad of lanes: bit-identical to the arithmetic form (same
            // weights, same pairing, same summation order), 20 % fewer VALU instructions per weight
            const bool geom_ok = a.lGroup == 128 && (K % 128) == 0 && lpr_log2 >= 2;
            if (geom_ok && g_knobs.q4_perm != 0)
                launch_m<FMT_Q4P>(a, L.mode, G, grid, smem, st);
            else
                launch_m<FMT_Q4>(a, L.mode, G, grid, smem, st);
            break;
        }
        case FMT_Q2: {
            if (g_knobs.q2_tab != 0 && smem + 2048 <= 160 * 1024)
                launch_m<FMT_Q2T>(a, L.mode, G, grid, smem + 2048, st);
            else
                launch_m<FMT_Q2>(a, L.mode, G, grid, smem, st);
            break;
        }
        case FMT_Q4R: launch_m<FMT_Q4R>(a, L.mode, G, grid, smem, st); break;
        default: {
            if (g_knobs.q1_tab != 0 && smem + 4096 <= 160 * 1024)
                launch_m<FMT_Q1T>(a, L.mode, G, grid, smem + 4096, st);
            else
                launch_m<FMT_Q1>(a, L.mode, G, grid, smem, st);
            break;
        }
    }
    L.blocks = blocks;
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

#if !KF_GEMV_CANON
void argmax_finish_launch(hipStream_t st, const float* val, const int* idx, int n, int32_t* d_argmax, int32_t* d_state, int32_t* d_tokens_out) {
    hipLaunchKernelGGL(argmax_finish_kernel, dim3(1), dim3(256), 0, st, val, idx, n, d_argmax, d_state, d_tokens_out);
}
#endif

}  // namespace kf
