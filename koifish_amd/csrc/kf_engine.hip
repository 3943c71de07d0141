// kf_engine.hip -- the decode step's layer loop as ONE persistent launch (gfx950 / wave64).
//
// Replaces the per-neuron walk of Fish::ForwardOnRLS (gLLM.cpp:755-771) over SelfAttention::cuInfer (QKV.cu:617-702) and
// FFN::cuInfer (NeuronFuse.cu:615-656) for all layers of one token: what was 5 dependent launches per layer (kf_norm_linear,
// kf_attn_block, kf_linear, kf_norm_gateup_swiglu, kf_linear) becomes 6 phases inside one kernel whose workgroups stay resident,
//
//   P1 [RMSNorm + Q,K,V]  P2 [q/k-norm + RoPE + attention slice]  P3 [slice merge]  P4 [o_proj + residual]
//   P5 [RMSNorm + gate/up + SwiGLU]  P6 [down_proj + residual]
//
// because on this chip a dependent launch boundary plus the setup and first HBM miss of the next kernel costs ~3 us while a layer's
// 8.4 MB of weights stream in ~1.3 us (scratch/ub_overlap.hip).  One workgroup of 8 waves per CU:
//
//   * wave 7 ("poller") waits for the phase's input vector and stages it into LDS.  Producers publish every output as a tagged
//     granule -- one aligned 4-byte {bf16 value, 16-bit generation} (attention partials: 8-byte {fp32, 32-bit generation}) written by
//     sc1 (write-through) stores, 16 bytes per lane -- and the poller sweeps the granules with sc1 loads, all loads of a sweep in
//     flight, until every tag equals the generation it expects: the data is the flag (MI355X_MICROARCH.md "Valid forms", R2).  No
//     counters, no fences, no grid barrier; nothing depends on dispatch order or placement, only on all workgroups being resident
//     (grid <= number of CUs).  A sweep costs its whole round trip whether it succeeds or not and slows the stores it is waiting for,
//     so the FIRST sweep of a hand-off is issued a fixed delay behind the moment this workgroup's own rows of the feeding phase were
//     published (all workgroups do the same work in step): scratch/ub_handoff3.hip, 1.2 us per edge instead of 2.1-2.3.
//     The vectors that cross XCDs live in uncached device memory (an sc1 sweep costs 43 ns per KB there, 75 in cached memory).
//   * waves 0..6 hold the phase's 16-byte weight blocks in registers -- requested one phase ahead, unconditionally, so HBM latency is
//     off the chain and every wait is a counted s_waitcnt vmcnt(N) -- and multiply when the barrier behind the poller's staging opens.
//     The arithmetic is the mat-vec kernel's (kf_gemv_blocks.h: same lanes per row, same per-lane chain, same DPP tree) and the
//     attention kernel's (kf_attn_common.h; same slices, same 4-wave key interleave, same merge order), so every output bit equals the
//     multi-launch path's.
//
// A workgroup polls a buffer only when it has work that needs it: then every reader of generation g has finished before any
// producer of generation g+1 can have its own inputs complete, and buffers are reused across layers without a second handshake.
//
// Every geometry figure of a phase is a compile-time constant of the model shape (PlanT below: types with static members, never
// objects -- an object indexed by a run-time job number becomes a table in constant memory, and the load from it a vmcnt(0) drain
// of the weight prefetch that was just issued).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <type_traits>

#include "kf_engine_common.h"

namespace kf {


// ---- the exchange area (uncached memory): granule vectors at compile-time offsets (dwords), 256-byte aligned
struct EngXOff {
    int xA, qkv, ao, xB, act, part, hbest, tokg, end; /* tokg: one 8-byte granule {token id, epoch}: the id workgroup 0 picked, for the next step of a multi-step launch; */ /* part: 8-byte granules, KF_ATTN_MAX_SPLITS * (2 hd + 4) per head (EngCfg::PSH); hbest: 8-byte granules [ENG_NWG]: the head's per-workgroup maxima */
};
constexpr EngXOff eng_xoff(int dim, int qd, int kvd, int ffn, int hd) {
    EngXOff o{};
    o.xA = 0, o.qkv = o.xA + eng_gran_dw(dim), o.ao = o.qkv + eng_gran_dw(qd + 2 * kvd), o.xB = o.ao + eng_gran_dw(qd), o.act = o.xB + eng_gran_dw(dim);
    o.part = o.act + eng_gran_dw(ffn);
    o.hbest = o.part + eng_gran_dw(2 * (qd / hd) * KF_ATTN_MAX_SPLITS * (2 * hd + 4));
    o.tokg = o.hbest + eng_gran_dw(2 * ENG_NWG);
    o.end = o.tokg + eng_gran_dw(2);
    return o;
}
// ---- the XCD-local area (cached memory, plain stores: the lines stay in that XCD's L2): [tickets 1 KiB] [lqkv 8 x lq dwords] [lpart 8 x lp qwords]
constexpr int eng_lq_stride(int gq, int hd) { return ((gq * hd + 2 * hd) * 4 + 255) / 256 * 64; }                   /* dwords */
constexpr int eng_lp_stride(int gq, int hd) { return (gq * KF_ATTN_MAX_SPLITS * (2 * hd + 4) * 8 + 255) / 256 * 32; } /* qwords */
constexpr size_t eng_loc_bytes(int gq, int hd) { return 1024 + (size_t)8 * eng_lq_stride(gq, hd) * 4 + (size_t)8 * eng_lp_stride(gq, hd) * 8; }

struct EngArgs {
    const EngLayer* layers;
    int n_layer;
    int n_steps; /* decode steps of this launch (>= 2 only with the head and the pick inside: the picked id reaches the other workgroups as a tagged granule) */
    float eps, qk_eps;
    const float* rope_table;
    const int32_t* d_state; /* {token, pos} */
    const uint16_t* x_in;   /* plain bf16 [dim]: the embedding row; NULL: the engine takes the row itself from `emb` (TokenEmbed::cuInfer inside the launch) */
    const uint16_t* emb;    /* bf16 embedding table [emb_rows, dim] (engine_set_embedding) */
    const int32_t* d_forced;
    int emb_rows;
    uint16_t* x_out;        /* plain bf16 [dim]: the residual stream after the last layer */
    uint32_t* xch;          /* exchange area (EngXOff) */
    int* ws;                /* [0] epoch, [1] error word */
    char* loc;              /* XCD-local area */
    int nsp, chunk, merge_e; /* attention slices of this launch's position bound; merge elements per workgroup */
    int kv_stride;
    int max_seq;            /* rows of a layer's K / V cache: no position of a launch may reach past it */
    float qbias[7];         /* qBias of q k v o gate up down */
    int delay[6];           /* s_sleep units between "this workgroup's own rows of the feeding phase are published" and the first sweep of: x (P1), q|k|v (P2), slice
                               partials (P3), ao (P4), xB (P5), act (P6) */
    // the LM head + greedy pick as trailing phases (engine_set_head; head_w == NULL: the launch ends with x_out)
    g_u32x4 head_w;          /* bf16 [vocab, dim] */
    g_u16 head_norm;         /* final RMSNorm weight */
    uint16_t* logits;        /* bf16 [vocab] */
    int32_t* d_state_w;      /* {token, pos}: advanced by the pick (NULL: logits only) */
    int32_t* d_tokens_out;
    int vocab, head_on;
    unsigned long long* dbg; /* diagnostic instantiation only (KF_ENG_DEBUG): [layer][role][16] wall-clock stamps of workgroup dbg_wg */
    int dbg_wg;
};
#define ENG_STAMP(role, k)                                                                                                           \
    do {                                                                                                                         \
        if (DBG && wg == a.dbg_wg && lane == 0) a.dbg[((size_t)l * 2 + (role)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)


// ---- the same for a vector that needs no norm, swept by ALL waves of the workgroup: wave w takes the loads w, w + NWV, ... (a kilobyte each), repeats
// only its own until their tags match and stages them; the caller's barrier behind it makes the vector whole.  A sweep by one wave costs ~43 ns per kilobyte on
// top of the round trip (scratch/ub_handoff3.hip): eight waves divide that, and a stale piece is re-read alone instead of with the eleven others.
template <int XCH, int NLD, int NBLK, int NWV, bool F32X = true>
__device__ __forceinline__ void eng_poll_stage_part(const uint32_t* gsrc, uint32_t tag, u32x4* xs, int wave, int lane, int* ws, bool& dead, int* nsweeps = nullptr) {
    constexpr int n = NLD * 256, NR = (NLD + NWV - 1) / NWV;
    const __amdgpu_buffer_rsrc_t rs = eng_rsrc(gsrc, (uint32_t)n * 4u);
    const uint32_t tagw = tag << 16;
    u32x4 g[NR];
    for (int spins = 0;; spins++) {
        uint32_t bad = 0;
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const int r = wave + i * NWV; /* wave-uniform */
            g[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((r < NLD ? r : wave) * 64 + lane) * 16, 0, 16 /* sc1 */));
        }
#pragma unroll
        for (int i = 0; i < NR; i++) bad = (wave + i * NWV < NLD) ? tags_bad(g[i], tagw, bad) : bad;
        if (wave >= NLD || all_good(bad)) {
            if (nsweeps) *nsweeps = spins + 1;
            break;
        }
        if (dead || spins > ENG_SPIN_MAX) {
            if (!dead && lane == 0) atomicOr(ws + 1, 1);
            dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const int r = wave + i * NWV;
        if (r >= NLD) continue;
        const int e0 = 4 * (r * 64 + lane);
        const uint32_t o0 = (g[i].x & 0xffffu) | (g[i].y << 16), o1 = (g[i].z & 0xffffu) | (g[i].w << 16);
        if (F32X) {
            const int q = e0 >> 2, c = q / XCH, j = q - c * XCH;
            xs[j * NBLK + c] = u32x4{o0 << 16, o0 & 0xffff0000u, o1 << 16, o1 & 0xffff0000u};
        } else {
            const int q = e0 >> 3, c = q / XCH, j = q - c * XCH;
            reinterpret_cast<u32x2*>(xs + j * NBLK + c)[(e0 >> 2) & 1] = u32x2{o0, o1};
        }
    }
}

template <bool PAIRED, int MAXS>
struct MvRegs {
    u32x4 w[MAXS], w2[PAIRED ? MAXS : 1];
    uint16_t st[MAXS], ze[MAXS], st2[PAIRED ? MAXS : 1], ze2[PAIRED ? MAXS : 1];
};
template <class PL, int NCW>
constexpr int c_maxs() { return ((PL::spg + NCW - 1) / NCW) * PL::iters; }
// step k of compute wave cw: slot s0 + cw + (k / iters)*NCW of the job (matrix) the workgroup's rows belong to, iteration k % iters;
// s0 = the workgroup's first slot counted inside that matrix, Mj = its rows (a workgroup's slots never straddle two matrices)
template <class PL, int NCW>
__device__ __forceinline__ MvAt mv_at(int k, int s0, int cw, int lane, int Mj) {
    const int sub = lane >> PL::lpr_log2, ll = lane & (PL::LPR - 1);
    const int sl = k / PL::iters, it = k - sl * PL::iters;
    const int s_loc = cw + sl * NCW;
    MvAt q;
    q.row = (s0 + s_loc) * PL::RPS + sub;
    q.col = it * PL::LPR + ll;
    q.ok = s_loc < PL::spg && q.row < Mj && q.col < PL::nBlk;
    return q;
}
// unconditional loads of the wave's blocks (clamped indices; masks are applied at the multiply): m = the matrix of the workgroup's rows, m2 = the
// paired one (up_proj beside gate_proj)
// hotbits: the sparse forward's mask of this workgroup's rows (bit r = row s0 * RPS + r is computed): a cold row's blocks are never read (row 0's stand in; the product is masked)
template <class PL, int NCW, int FMT, int MAXS>
__device__ __forceinline__ void mv_prefetch(const EngMat m, const EngMat m2, int s0, int cw, int lane, int Mj, MvRegs<PL::PAIRED, MAXS>& R, uint32_t hotbits = 0xffffffffu) {
    constexpr bool GAMA = BlockDot<FMT>::HAS_GAMA;
    constexpr int gshift = FMT >= FMT_Q4 ? (FMT == FMT_Q4 || FMT == FMT_Q4P || FMT == FMT_Q1T || FMT == FMT_Q2T ? 2 : (FMT == FMT_Q2 ? 1 : 0)) : 0; /* 128-weight groups = four 32-weight blocks */
#pragma unroll
    for (int k = 0; k < MAXS; k++) {
        const MvAt q = mv_at<PL, NCW>(k, s0, cw, lane, Mj);
        int row = q.row < Mj ? q.row : Mj - 1;
        row = row > 0 ? row : 0;
        row = ((hotbits >> ((q.row - s0 * PL::RPS) & 31)) & 1u) ? row : 0;
        const int col = q.col < PL::nBlk ? q.col : PL::nBlk - 1;
        const uint32_t bidx = (uint32_t)row * (uint32_t)PL::nBlk + (uint32_t)col;
        if constexpr (FMT == FMT_Q1T) { /* this lane's dword of the 16-byte block: dword 3 holds elements 0 .. 31 */
            const uint32_t di = (bidx & ~3u) + (3u - (bidx & 3u));
            R.w[k] = u32x4{__builtin_nontemporal_load(reinterpret_cast<const uint32_t KF_GLOBAL*>(m.w) + di), 0u, 0u, 0u};
            if (PL::PAIRED) R.w2[k] = u32x4{__builtin_nontemporal_load(reinterpret_cast<const uint32_t KF_GLOBAL*>(m2.w) + di), 0u, 0u, 0u};
        } else if constexpr (FMT == FMT_Q2T) { /* this lane's half of the 16-byte block: bytes 8 .. 15 hold elements 0 .. 31 (dword 3 first), bytes 0 .. 7 elements 32 .. 63 */
            const uint32_t qi = (bidx & ~1u) + (1u - (bidx & 1u));
            const u32x2 h = __builtin_nontemporal_load(reinterpret_cast<const u32x2 KF_GLOBAL*>(m.w) + qi);
            R.w[k] = u32x4{h.x, h.y, 0u, 0u}; /* .y = the half's first 16 elements, .x = its last 16 */
            if (PL::PAIRED) {
                const u32x2 h2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 KF_GLOBAL*>(m2.w) + qi);
                R.w2[k] = u32x4{h2.x, h2.y, 0u, 0u};
            }
        } else {
            R.w[k] = __builtin_nontemporal_load(m.w + bidx);
            if (PL::PAIRED) R.w2[k] = __builtin_nontemporal_load(m2.w + bidx);
        }
        if (GAMA) {
            const uint32_t gi = bidx >> gshift;
            R.st[k] = m.step[gi], R.ze[k] = m.zero[gi];
            if (PL::PAIRED) R.st2[k] = m2.step[gi], R.ze2[k] = m2.zero[gi];
        }
    }
}
// epi(row, v, v2) runs in the lane that owns a finished row (row counted inside the matrix)
#ifndef ENG_P1_SHARE
#define ENG_P1_SHARE 1 /* the poller wave computes one of the workgroup's P1 row slots (0: seven waves, wave 0 takes two slots -- its second block of pair words costs 16 more registers and spills) */
#endif
#ifndef ENG_AH_REGS
#define ENG_AH_REGS 64 /* registers of dequantised pair words a lane may hold across a hand-off (MvPhase::AH) */
#endif
#ifndef ENG_AH_PAIR_WIDE
#define ENG_AH_PAIR_WIDE 0 /* the 2048-wide shape's gate | up phase: every block pair behind the barrier */
#endif
#ifndef ENG_AH_REGS_WIDE
#define ENG_AH_REGS_WIDE 16 /* the same for the 2048-wide shape, whose raw blocks in flight take 72 registers */
#endif
#ifndef ENG_COOP
#define ENG_COOP 1 /* the norm-free vectors (ao, act) are swept by all eight waves */
#endif
// the wave's blocks as bf16 pair words (BlockPrep): formed while the wave waits for the phase's activations.  (Round 4 also widened them to the fp32 operand pairs of the
// canonical v_pk_fma_f32 in front of the barrier: 32 registers per block -- the kernel spilled 30 - 180 registers and ran 20 % slower; the two conversions per pair stay behind
// the barrier, where they fill the bubbles of the dependent fma chains.)
template <bool PAIRED, int MAXS>
struct MvDeq {
    uint32_t p[MAXS][16], p2[PAIRED ? MAXS : 1][16];
};
// AH <= MAXS: the wave's first AH blocks are dequantised ahead of the barrier (D), the others behind it, block by block in front of their products (shapes whose blocks per
// lane do not fit the register file as pair words: Qwen3-1.7B's gate | up and down_proj phases hold 8 and 6 blocks per lane)
template <class PL, int NCW, int FMT, int MAXS, int AH>
__device__ __forceinline__ void mv_dequant(float qb, float qb2, int cw, int lane, const MvRegs<PL::PAIRED, MAXS>& R, MvDeq<PL::PAIRED, AH>& D, const u32x4* tab) {
#pragma unroll
    for (int k = 0; k < AH; k++) {
        if (cw + (k / PL::iters) * NCW >= PL::spg) continue; /* wave-uniform: this wave has no such slot */
        const float st = bf2f(R.st[k]);
        BlockPrep<FMT>::prep(R.w[k], st, bf2f(R.ze[k]), -(qb * st), lane, D.p[k], tab);
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("" : "+v"(D.p[k][i])); /* formed HERE, in front of the barrier, not sunk to the first use behind it */
        if (PL::PAIRED) {
            const float st2 = bf2f(R.st2[k]);
            BlockPrep<FMT>::prep(R.w2[k], st2, bf2f(R.ze2[k]), -(qb2 * st2), lane, D.p2[k], tab);
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("" : "+v"(D.p2[k][i]));
        }
    }
}
// mv_run on dequantised blocks: the same lanes, chains and tree
template <class PL, int NCW, int FMT, int MAXS, int AH, int AHN, bool CANON, typename Epi>
__device__ __forceinline__ void mv_run_deq(int s0, int cw, int lane, int Mj, const MvDeq<PL::PAIRED, AHN>& D, const MvRegs<PL::PAIRED, MAXS>& R, float qb, float qb2, const u32x4* tab,
                                           const u32x4* xs, uint32_t hotbits, Epi&& epi) {
    acc_t<CANON> acc{}, acc2{};
#pragma unroll
    for (int k = 0; k < MAXS; k++) {
        const int sl = k / PL::iters, it = k - sl * PL::iters;
        if (cw + sl * NCW >= PL::spg) continue;
        const MvAt q = mv_at<PL, NCW>(k, s0, cw, lane, Mj);
        const int col = q.col < PL::nBlk ? q.col : PL::nBlk - 1;
        if (it == 0) acc = acc_t<CANON>{}, acc2 = acc_t<CANON>{};
        const bool on = q.ok && ((hotbits >> ((q.row - s0 * PL::RPS) & 31)) & 1u);
        acc_t<CANON> r, r2{};
        if constexpr (AH == MAXS) {
            r = pairs_dot<CANON>(D.p[k], xs, col, PL::nBlk, acc);
            if (PL::PAIRED) r2 = pairs_dot<CANON>(D.p2[k], xs, col, PL::nBlk, acc2);
        } else {
            if (k < AH) {
                r = pairs_dot<CANON>(D.p[k < AH ? k : 0], xs, col, PL::nBlk, acc);
                if (PL::PAIRED) r2 = pairs_dot<CANON>(D.p2[PL::PAIRED && k < AH ? k : 0], xs, col, PL::nBlk, acc2);
            } else { /* this block's pair words now (the same BlockPrep: same bits) */
                uint32_t pw[16];
                const float st = bf2f(R.st[k]);
                BlockPrep<FMT>::prep(R.w[k], st, bf2f(R.ze[k]), -(qb * st), lane, pw, tab);
                r = pairs_dot<CANON>(pw, xs, col, PL::nBlk, acc);
                if (PL::PAIRED) {
                    const float st2 = bf2f(R.st2[k]);
                    BlockPrep<FMT>::prep(R.w2[k], st2, bf2f(R.ze2[k]), -(qb2 * st2), lane, pw, tab);
                    r2 = pairs_dot<CANON>(pw, xs, col, PL::nBlk, acc2);
                }
            }
        }
        acc = acc_pick(on, r, acc);
        if (PL::PAIRED) acc2 = acc_pick(on, r2, acc2);
        if (it == PL::iters - 1) {
            const float v = group_sum(acc_join(acc), PL::lpr_log2);
            float v2 = 0.f;
            if (PL::PAIRED) v2 = group_sum(acc_join(acc2), PL::lpr_log2);
            if ((lane & (PL::LPR - 1)) == 0 && q.ok) epi(q.row, v, v2);
        }
    }
}

// One mat-vec phase of a wave: its blocks are dequantised into bf16 pair words ahead of the hand-off (ahead) and multiplied behind it (run).  tab: the 1-bit selector table
template <class C, class PL, int NCW, int MAXS>
struct MvPhase {
    // blocks dequantised ahead: all of them while a lane holds at most ENG_AH_REGS registers of pair words (16 per block, 32 per gate | up pair); Qwen3-0.6B: every phase
    static constexpr int PER = PL::PAIRED ? 32 : 16, AH0 = (PL::PAIRED ? C::AHR_P : C::AHR) / PER, AH = MAXS <= AH0 ? MAXS : AH0; /* 0: every block behind the barrier */
    MvDeq<PL::PAIRED, (AH > 0 ? AH : 1)> d;
    float qb_, qb2_;
    const u32x4* tab_;
    __device__ __forceinline__ void ahead(float qb, float qb2, int cw, int lane, const MvRegs<PL::PAIRED, MAXS>& R, const u32x4* tab) {
        qb_ = qb, qb2_ = qb2, tab_ = tab;
        if constexpr (AH > 0) mv_dequant<PL, NCW, C::FMT, MAXS, AH>(qb, qb2, cw, lane, R, d, tab);
    }
    template <typename Epi>
    __device__ __forceinline__ void run(int s0, int cw, int lane, int Mj, const u32x4* xs, uint32_t hotbits, const MvRegs<PL::PAIRED, MAXS>& R, Epi&& epi) {
        mv_run_deq<PL, NCW, C::FMT, MAXS, AH, (AH > 0 ? AH : 1), C::CANON>(s0, cw, lane, Mj, d, R, qb_, qb2_, tab_, xs, hotbits, epi);
    }
};

// ------------------------------------------------------------------------------------------------ the kernel
// LDS: [layer table] [xs0] [xs1] [xrawA dim] [xrawB dim] [attention: qraw GQ*hd | kraw hd | vraw hd | qb GQ*hd | knew hd | wmax | comb] [outb] [cnt, pub]
struct EngLds {
    const EngLayer* lay;
    u32x4* xs[2];
    uint16_t *xrawA, *xrawB, *qraw, *kraw, *vraw, *qb, *knew;
    float* wmax;   /* [16] scratch of the head's arg-max */
    double* comb;  /* [NWV][GQ][hd + 2] the waves' fp64 sums {O[hd], L, m} of the attention slice */
    uint32_t* outb; /* [64] a phase's output granules of this workgroup, gathered so that ONE wave stores them 16 bytes per lane */
    int* cnt;       /* arrival counter of the compute waves that own rows of the phase */
    double* msc;    /* [ME][KF_ATTN_MAX_SPLITS] the slice partials of this workgroup's merge elements, transposed for the per-element chains; [ME] dwords behind it: its ao granules */
    int* pub;       /* [4] layers of P1 / P4 / P5 / P6 rows this workgroup has published so far: the poller starts sweeping for the phase's consumers' vector behind it */
    const u32x4* q1tab;      /* FMT_Q1T: 256 x 16 B of v_perm selectors (BlockDot<FMT_Q1T>) */
    const uint32_t* hotbits; /* [n_layer] sparse forward: bit r = this workgroup's r-th gate / up row is hot (all ones: dense) */
};
struct EngSlice { /* this workgroup's attention slice and merge share */
    int pos, len, nsp, kvh, split, h0, t0, t1, me0;
    bool has_unit, empty, own_new, has_merge;
    int xcc, rank;  /* XCD-mapped form: the XCD this workgroup runs on and its ticket there */
    int j1;         /* the matrix (0 q, 1 k, 2 v) this workgroup's P1 rows belong to */
    int s1;         /* first P1 slot of this workgroup, counted inside that matrix */
    int M1;         /* rows of that matrix */
    int q_out0;     /* index of its first P1 row in the vector its P1 rows are published to */
};

// The compute waves that own rows of a phase leave their granules in LDS; the wave that arrives last stores the workgroup's rows with ONE
// instruction, 16 bytes per lane: a 64-byte sector of the hand-off vector is then written by a few whole pieces instead of by sixteen
// separate 4-byte write-through stores (each a read-modify-write at the memory side).
// local = true: plain stores into a buffer of this XCD
__device__ __forceinline__ void wg_publish(const EngLds& L, int phase, uint32_t* buf, int idx0, int nrows, int nwaves, int lane, bool local = false) {
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(L.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != nwaves - 1) return;
    if (lane == 0) *L.cnt = 0;
    if (4 * lane < nrows) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(L.outb + 4 * lane);
        if (local)
            *reinterpret_cast<u32x4*>(buf + idx0 + 4 * lane) = v;
        else /* one descriptor for the workgroup's piece, the lane's 16 bytes as an offset (a per-lane base pointer would be served lane by lane) */
            __builtin_amdgcn_raw_buffer_store_b128(v, eng_rsrc(buf + idx0, (uint32_t)nrows * 4u), lane * 16, 0, 16 /* sc1 */);
    }
    if (lane == 0) __hip_atomic_fetch_add(L.pub + phase, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); /* the poller may start to sweep for what the consumers of these rows produce */
}

template <int FMT_, int GQ_, int HD_, int NWV_, int DIM_, int QD_, int KVD_, int FFN_, int NWG_, bool XMAP_, bool DBG_, bool CANON_ = true>
struct EngCfg {
    static constexpr int FMT = FMT_, GQ = GQ_, HD = HD_, NWV = NWV_, DIM = DIM_, QD = QD_, KVD = KVD_, FFN = FFN_, NWG = NWG_;
    static constexpr bool XMAP = XMAP_, DBG = DBG_;
    static constexpr int AHR = DIM_ > 1024 ? ENG_AH_REGS_WIDE : ENG_AH_REGS; /* MvPhase::AH: registers of pair words held across a hand-off (single-matrix phases) */
    static constexpr int AHR_P = DIM_ > 1024 ? ENG_AH_PAIR_WIDE : ENG_AH_REGS; /* the same for the gate | up phase (32 registers per block pair) */
    // CANON: the mat-vec phases and the head in the canonical order (one v_fma_f32 per product on fp32 operands: bit-exact against the oracle); false: v_dot2c_f32_bf16 on
    // bf16 pairs (kf_set_canonical(ctx, 0): <= 1 bf16 ulp per output from the oracle, fewer vector instructions).  The attention is canonical either way.
    static constexpr bool CANON = CANON_;
    static_assert(FMT_ == FMT_Q4 || FMT_ == FMT_Q4P || FMT_ == FMT_Q1T || FMT_ == FMT_Q2T, "storages with a BlockPrep: 32-weight blocks dequantised ahead of the hand-off (16 pair words each)");
    static constexpr bool F32X = CANON_; /* canonical order: activations staged as fp32 chunks (pairs_dot<true>: no conversion of the activation pair per product) */
    static constexpr int XCH = F32X ? 8 : 4; /* 16-byte chunks of x per 32-weight block */
    static constexpr int n_head = QD_ / HD_, n_kv = KVD_ / HD_;
    using SH = EngShape<FMT_, DIM_, QD_, KVD_, FFN_, NWG_>;
    static constexpr int xA = eng_xoff(DIM_, QD_, KVD_, FFN_, HD_).xA, qkv = eng_xoff(DIM_, QD_, KVD_, FFN_, HD_).qkv, ao = eng_xoff(DIM_, QD_, KVD_, FFN_, HD_).ao,
                         xB = eng_xoff(DIM_, QD_, KVD_, FFN_, HD_).xB, act = eng_xoff(DIM_, QD_, KVD_, FFN_, HD_).act, part = eng_xoff(DIM_, QD_, KVD_, FFN_, HD_).part,
                         hbest = eng_xoff(DIM_, QD_, KVD_, FFN_, HD_).hbest, tokg = eng_xoff(DIM_, QD_, KVD_, FFN_, HD_).tokg;
    // the LM head (bf16 [vocab, DIM]) as trailing phases of the same launch: the geometry gemv_launch picks for a many-row bf16 matrix of this width
    static constexpr int HnBlk = DIM_ / 8, Hlpr_log2 = c_lpr_log2(DIM_ / 8, 1L << 20), HLPR = 1 << Hlpr_log2, HRPS = 64 >> Hlpr_log2, Hiters = (HnBlk + HLPR - 1) / HLPR;
    static constexpr int lq_stride = eng_lq_stride(GQ_, HD_), lp_stride = eng_lp_stride(GQ_, HD_);
    // slice merge: every workgroup merges ME consecutive elements of one head (ME = the power of two >= q_dim / NWG).  A head's slice partials (fp64 sums of the
    // canonical softmax, kf_attn_common.h) lie as [hd / ME element groups][KF_ATTN_MAX_SPLITS slices][ME] values, a value = two 8-byte {32 bits, generation}
    // granules (low word, high word), then [KF_ATTN_MAX_SPLITS][4] granules {m, L low, L high, unused}: what a workgroup merges is contiguous
    static constexpr int ME = pow2_ceil((QD_ + NWG_ - 1) / NWG_), PSH = KF_ATTN_MAX_SPLITS * (2 * HD_ + 4), NLM = (KF_ATTN_MAX_SPLITS * ME * 2 + 127) / 128;
    static_assert(ME <= 64 && ME <= HD_, "merge elements per workgroup");
};
template <class C>
__device__ __forceinline__ uint32_t* eng_lqkv(const EngArgs& a, int xcc) { return reinterpret_cast<uint32_t*>(a.loc + 1024) + (size_t)xcc * C::lq_stride; }
template <class C>
__device__ __forceinline__ unsigned long long* eng_lpart(const EngArgs& a, int xcc) {
    return reinterpret_cast<unsigned long long*>(a.loc + 1024 + (size_t)8 * C::lq_stride * 4) + (size_t)xcc * C::lp_stride;
}

// ---- P2 for one wave: q/k-norm + RoPE + the canonical attention over this workgroup's slice, run by ALL 8 waves of the workgroup (the poller too: it has nothing
// to wait for between the staging of the raw heads and the slice partials).  The canonical softmax sums in fp64 against per-wave exponents (kf_attn_common.h), so
// how the keys are dealt to waves and lanes does not show in the result: wave w takes the keys t0 + w * KPW + group + 8 * KPW * u.
template <class C>
struct EngAttnState { /* the wave's K / V tiles of the slice, requested a layer ahead */
    static constexpr int U = 2; /* tiles per batch: 8 waves x (64 / LPK) keys x 2 = 64 keys (hd 128) */
    u32x4 kk[U], vv[U];
    uint16_t nw0, nw1; /* the q-norm (waves 0 .. GQ - 1: they prepare the q heads) or k-norm (wave GQ % NWV: the new key) weights of this lane's pair, requested at the top of the layer */
    float rc, rs;                /* the RoPE table's (cos, sin) of this lane's pair at the step's position: the same for every layer, read once */
};
template <class C>
__device__ __forceinline__ void eng_attn_rope(const EngArgs& a, const EngSlice& S, int lane, EngAttnState<C>& T) {
    T.rc = 1.f, T.rs = 0.f;
    if (a.rope_table && lane < (C::HD >> 1)) {
        const float* tab_pos = a.rope_table + (size_t)S.pos * C::HD;
        T.rc = tab_pos[2 * lane], T.rs = tab_pos[2 * lane + 1];
    }
}
template <class C>
__device__ __forceinline__ void eng_attn_normw(const EngLayer& ly, int wave, int lane, EngAttnState<C>& T) {
    const int half = C::HD >> 1, j = lane < half ? lane : half - 1;
    static_assert(C::GQ < C::NWV, "the q heads and the new key are prepared by different waves");
    g_u16 np = wave < C::GQ ? ly.norm_q : ly.norm_k; /* wave-uniform */
    T.nw0 = T.nw1 = 0;
    if (np) T.nw0 = np[j], T.nw1 = np[j + half];
}
template <class C>
__device__ __forceinline__ void eng_attn_issue(const EngArgs& a, const EngLayer& ly, const EngSlice& S, int wave, int lane, EngAttnState<C>& T, int b) {
    constexpr int hd = C::HD, LPK = hd >> 3, KPW = 64 / LPK, lpk_log2 = (C::HD == 128 ? 7 : 6) - 3, NWA = C::NWV, U = EngAttnState<C>::U;
    const int grp = lane >> lpk_log2, d0 = (lane & (LPK - 1)) * 8;
    const int tb = S.t0 + wave * KPW + grp + b * U * NWA * KPW;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int t = tb + u * NWA * KPW;
        T.kk[u] = T.vv[u] = u32x4{0, 0, 0, 0};
        if (t < S.t1) {
            const size_t off = (size_t)t * a.kv_stride + (size_t)S.kvh * hd + d0;
            T.kk[u] = *reinterpret_cast<const u32x4 KF_GLOBAL*>(ly.kcache + off);
            T.vv[u] = *reinterpret_cast<const u32x4 KF_GLOBAL*>(ly.vcache + off);
        }
    }
}
template <class C>
__device__ __forceinline__ void eng_attn_phase(const EngArgs& a, const EngLds& L, const EngSlice& S, const EngLayer& ly, uint32_t gen, uint32_t tag, int wave, int lane,
                                               EngAttnState<C>& T, int l, int wg) {
    constexpr bool DBG = C::DBG;
    constexpr int GQ = C::GQ, hd = C::HD, hd_log2 = C::HD == 128 ? 7 : 6, LPK = hd >> 3, KPW = 64 / LPK, lpk_log2 = hd_log2 - 3, NWA = C::NWV, U = EngAttnState<C>::U;
    constexpr bool XMAP = C::XMAP;
    constexpr int NQ = (GQ + NWA - 1) / NWA;
    const int tid = (wave << 6) | lane, pos = S.pos, nsp = S.nsp, kvh = S.kvh, h0 = S.h0, t1 = S.t1; /* | not +: with an opaque lane the sum would be re-associated and its invariant part kept in a register */
    const int grp = lane >> lpk_log2, d0 = (lane & (LPK - 1)) * 8;
    const int tstride = NWA * KPW, tstart = S.t0 + wave * KPW + grp;
    const int nbatch = (S.t1 - S.t0 + U * tstride - 1) / (U * tstride);
    __syncthreads(); /* raw heads staged */
    if (wave == 0) ENG_STAMP(1, 2);
    if (!S.empty) {
        { /* prologue: q heads of this group, and the new key when it lies in this slice (ROPE::cuInfer) */
            const bool rope = a.rope_table != nullptr;
            const bool qnorm = ly.norm_q != nullptr;
            const int half = hd >> 1, j = lane < half ? lane : half - 1;
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                const int hq = wave + i * NWA;
                if (hq < GQ) {
                    HeadRaw r;
                    r.x0 = L.qraw[hq * hd + j], r.x1 = L.qraw[hq * hd + j + half];
                    r.w0 = qnorm ? T.nw0 : r.x0, r.w1 = qnorm ? T.nw1 : r.x1;
                    prep_head_cs(r, qnorm, rope, T.rc, T.rs, hd, a.qk_eps, L.qb + hq * hd, nullptr, lane);
                }
            }
            if (S.own_new && wave == (GQ % NWA)) {
                HeadRaw r;
                r.x0 = L.kraw[j], r.x1 = L.kraw[j + half];
                r.w0 = ly.norm_k ? T.nw0 : r.x0, r.w1 = ly.norm_k ? T.nw1 : r.x1;
                prep_head_cs(r, ly.norm_k != nullptr, rope, T.rc, T.rs, hd, a.qk_eps, L.knew, nullptr, lane);
            }
        }
        __syncthreads(); /* heads prepared */
        if (wave == 0) ENG_STAMP(1, 11);
        if (S.own_new) { /* the cache rows of this position: the prepared key, the raw value (K.out / V.out alias them in the reference) */
            g_u16w krow = ly.kcache + (size_t)pos * a.kv_stride + (size_t)kvh * hd;
            g_u16w vrow = ly.vcache + (size_t)pos * a.kv_stride + (size_t)kvh * hd;
            if (tid < hd / 8) { /* 16 bytes per lane (the rows are 16-byte aligned: kv_stride and hd are multiples of 8) instead of a 2-byte store per thread */
                *reinterpret_cast<u32x4 KF_GLOBAL*>(const_cast<uint16_t KF_GLOBAL*>(krow) + 8 * tid) = *reinterpret_cast<const u32x4*>(L.knew + 8 * tid);
                *reinterpret_cast<u32x4 KF_GLOBAL*>(const_cast<uint16_t KF_GLOBAL*>(vrow) + 8 * tid) = *reinterpret_cast<const u32x4*>(L.vraw + 8 * tid);
            }
        }
        // the slice's sums: canonical (fp64 sums of exact weights: bit-exact against the oracle) when the mat-vec phases are, else the decode kernel's default arithmetic
        // (v_dot2c scores, v_exp_f32, fp32 sums); both keep integer reference exponents, so the combine below and the merge treat them alike
        using Acc = std::conditional_t<C::CANON, CanonAcc<GQ>, FastAcc<GQ>>;
        using Sum = std::conditional_t<C::CANON, double, float>;
        Acc A;
        A.init();
        float qf[GQ][8];
        u32x4 qp[GQ];
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            const u32x4 qv = *reinterpret_cast<const u32x4*>(L.qb + hq * hd + d0);
            const uint32_t q4[4] = {qv.x, qv.y, qv.z, qv.w};
            qp[hq] = qv;
#pragma unroll
            for (int i = 0; i < 4; i++) qf[hq][2 * i] = bf_lo(q4[i]), qf[hq][2 * i + 1] = bf_hi(q4[i]);
        }
        const float rden = 1.0f / sqrtf((float)hd);
        for (int b = 0; b < nbatch; b++) {
            const int tb = tstart + b * U * tstride;
            u32x4 ck[U], cv[U];
            bool valid[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int t = tb + u * tstride;
                valid[u] = t < t1;
                ck[u] = T.kk[u], cv[u] = T.vv[u];
                if (valid[u] && t == pos) ck[u] = *reinterpret_cast<const u32x4*>(L.knew + d0), cv[u] = *reinterpret_cast<const u32x4*>(L.vraw + d0);
            }
            if (b + 1 < nbatch) eng_attn_issue<C>(a, ly, S, wave, lane, T, b + 1);
            if constexpr (C::CANON) canon_batch<GQ, LPK, U>(A, qf, ck, cv, valid, lpk_log2, rden);
            else fast_batch<GQ, LPK, U>(A, qp, ck, cv, valid, lpk_log2, rden);
        }
        if (wave == 0) ENG_STAMP(1, 12);
        Sum* const comb = reinterpret_cast<Sum*>(L.comb);
        if constexpr (C::CANON) canon_wave_to_lds<GQ, LPK>(A, comb + (size_t)wave * GQ * (hd + 2), hd, lane, d0);
        else fast_wave_to_lds<GQ, LPK>(A, comb + (size_t)wave * GQ * (hd + 2), hd, lane, d0);
        if (wave == 0) ENG_STAMP(1, 13);
        __syncthreads(); /* the waves' sums in LDS */
        if (wave == 0) ENG_STAMP(1, 14);
        constexpr int PSD = hd + 2;
        for (int i = tid; i < GQ * hd; i += NWA * 64) { /* (two or four threads per element with DPP joins was measured: slower, 378 vs 370 us per step) */
            const int hq = i >> hd_log2, d = i & (hd - 1);
            float ms = -__builtin_inff();
#pragma unroll
            for (int sl = 0; sl < NWA; sl++) ms = fmaxf(ms, (float)comb[((size_t)sl * GQ + hq) * PSD + hd + 1]);
            double o, Ls;
            if constexpr (C::CANON) {
                o = 0.0, Ls = 0.0;
#pragma unroll
                for (int sl = 0; sl < NWA; sl++) {
                    const double* c = L.comb + ((size_t)sl * GQ + hq) * PSD;
                    const int e = canon_shift((float)c[hd + 1] - ms);
                    o += ldexp_d(c[d], e);
                    Ls += ldexp_d(c[hd], e);
                }
            } else {
                float of = 0.f, lf = 0.f;
#pragma unroll
                for (int sl = 0; sl < NWA; sl++) {
                    const float* c = comb + ((size_t)sl * GQ + hq) * PSD;
                    const int e = fast_shift(c[hd + 1] - ms);
                    of += __builtin_ldexpf(c[d], e);
                    lf += __builtin_ldexpf(c[hd], e);
                }
                o = (double)of, Ls = (double)lf; /* the merge takes the slice's sums as doubles either way */
            }
            if (nsp == 1) {
                st_gran(a.xch + C::ao + (h0 + hq) * hd + d, tag, f2bf((float)(o / Ls)));
            } else {
                constexpr int ME = C::ME, MAXSP = KF_ATTN_MAX_SPLITS;
                const size_t oi = ((size_t)(d / ME) * (MAXSP * ME) + (size_t)S.split * ME + (d & (ME - 1))) * 2, mi = (size_t)hd * MAXSP * 2 + (size_t)S.split * 4;
                const unsigned long long ob = __builtin_bit_cast(unsigned long long, o), lb = __builtin_bit_cast(unsigned long long, Ls), gg = (unsigned long long)gen << 32;
                if (XMAP) { /* plain stores into this XCD's partial buffer: a value = two {32 bits, generation} granules */
                    unsigned long long* dst = eng_lpart<C>(a, S.xcc) + (size_t)hq * C::PSH;
                    *reinterpret_cast<ulonglong2*>(dst + oi) = ulonglong2{gg | (ob & 0xffffffffull), gg | (ob >> 32)};
                    if (d == 0) {
                        *reinterpret_cast<ulonglong2*>(dst + mi) = ulonglong2{gg | __float_as_uint(ms), gg | (lb & 0xffffffffull)};
                        dst[mi + 2] = gg | (lb >> 32);
                    }
                } else {
                    unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.xch + C::part) + (size_t)(h0 + hq) * C::PSH;
                    st_gran64(dst + oi, gen, __uint_as_float((uint32_t)ob)), st_gran64(dst + oi + 1, gen, __uint_as_float((uint32_t)(ob >> 32)));
                    if (d == 0) st_gran64(dst + mi, gen, ms), st_gran64(dst + mi + 1, gen, __uint_as_float((uint32_t)lb)), st_gran64(dst + mi + 2, gen, __uint_as_float((uint32_t)(lb >> 32)));
                }
            }
        }
    } else if (nsp > 1) { /* empty slice: neutral partial (sums 0, exponent -inf) */
        for (int i = tid; i < GQ * hd; i += NWA * 64) {
            const int hq = i >> hd_log2, d = i & (hd - 1);
            constexpr int ME = C::ME, MAXSP = KF_ATTN_MAX_SPLITS;
            const size_t oi = ((size_t)(d / ME) * (MAXSP * ME) + (size_t)S.split * ME + (d & (ME - 1))) * 2, mi = (size_t)hd * MAXSP * 2 + (size_t)S.split * 4;
            const unsigned long long gg = (unsigned long long)gen << 32;
            if (XMAP) {
                unsigned long long* dst = eng_lpart<C>(a, S.xcc) + (size_t)hq * C::PSH;
                dst[oi] = gg, dst[oi + 1] = gg;
                if (d == 0) dst[mi] = gg | 0xff800000u, dst[mi + 1] = gg, dst[mi + 2] = gg;
            } else {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.xch + C::part) + (size_t)(h0 + hq) * C::PSH;
                st_gran64(dst + oi, gen, 0.f), st_gran64(dst + oi + 1, gen, 0.f);
                if (d == 0) st_gran64(dst + mi, gen, -__builtin_inff()), st_gran64(dst + mi + 1, gen, 0.f), st_gran64(dst + mi + 2, gen, 0.f);
            }
        }
    }
}

// the poller wave: per layer it stages P1's x, the slice's q/k/v heads, merges, stages P4's, P5's and P6's inputs
template <class C>
__device__ __forceinline__ void eng_poller_main(const EngArgs& a, const EngLds& L, const EngSlice& S, int epoch, int step, int wg, int lane) {
    using SH = typename C::SH;
    using P1 = typename SH::P1;
    using P4 = typename SH::P4;
    using P5 = typename SH::P5;
    using P6 = typename SH::P6;
    constexpr int FMT = C::FMT, GQ = C::GQ, HD = C::HD, NWV = C::NWV;
    constexpr bool XMAP = C::XMAP, DBG = C::DBG;
    constexpr int ND = C::DIM / 256, NQD = C::QD / 256, NF = C::FFN / 256;
    constexpr int XCH = C::XCH, hd = HD, hd_log2 = HD == 128 ? 7 : 6;
    bool dead = false;
    const bool has1 = XMAP ? true : wg * P1::spg < P1::total, has4 = wg * P4::spg < P4::total, has5 = wg * P5::spg < P5::total, has6 = wg * P6::spg < P6::total;
    // the poller's share of P1 (virtual compute wave NWV - 1)
    constexpr int NCW1 = ENG_P1_SHARE ? NWV : NWV - 1, S1 = c_maxs<P1, NCW1>(), NWP1 = P1::spg < NCW1 ? P1::spg : NCW1;
    constexpr bool P1_SHARE = ENG_P1_SHARE && P1::spg >= NWV; /* the poller owns a slot */
    const float qb1 = S.j1 == 0 ? a.qbias[0] : (S.j1 == 1 ? a.qbias[1] : a.qbias[2]);
    const int row0_1 = S.s1 * P1::RPS;
    MvRegs<false, S1> r1;
    auto mat1 = [&](int l) { /* the matrix of this workgroup's P1 rows in layer l */
        return L.lay[l].m[S.j1];
    };
    if (P1_SHARE) mv_prefetch<P1, NCW1, FMT, S1>(mat1(0), mat1(0), S.s1, NWV - 1, lane, S.M1, r1);
    EngAttnState<C> T;
#pragma unroll
    for (int u = 0; u < EngAttnState<C>::U; u++) T.kk[u] = T.vv[u] = u32x4{0, 0, 0, 0};
    eng_attn_rope<C>(a, S, lane, T);
    if (S.has_unit && !S.empty) eng_attn_issue<C>(a, L.lay[0], S, NWV - 1, lane, T, 0);
    int sw[4] = {0, 0, 0, 0};
    int swt[6] = {0, 0, 0, 0, 0, 0}; /* this step's sweeps of the poller per hand-off (x, q|k|v, slice partials, ao, xB, act): workgroup 0 adds them to the engine's statistics */
    for (int l = 0; l < a.n_layer; l++) {
        const EngLayer& ly = L.lay[l];
        const uint32_t gen = (uint32_t)epoch * (uint32_t)a.n_layer + (uint32_t)l, tag = gen & 0xffffu;
        if (S.has_unit && !S.empty) eng_attn_normw<C>(ly, NWV - 1, lane, T);
        MvPhase<C, P1, NCW1, S1> w1;
        if (P1_SHARE) w1.ahead(qb1, 0.f, NWV - 1, lane, r1, L.q1tab); /* its blocks were requested behind the previous layer's attention phase */
        // P1 (P4 adds this x as the residual)
        ENG_STAMP(0, 0);
        if (has1 || has4) {
            if (l == 0) {
                const uint16_t* x0 = a.x_in;
                if (!x0) { /* embed_kernel's row choice: the state's token, overridden by a teacher-forced id at this position */
                    int tok = a.d_state[0];
                    if (step > 0) { /* a later step of a multi-step launch: the id workgroup 0 picked at the end of the previous step, {id, epoch of this step} */
                        const __amdgpu_buffer_rsrc_t rt = eng_rsrc(a.xch + C::tokg, 8u);
                        for (int spins = 0;; spins++) {
                            const u32x2 g = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rt, 0, 0, 16 /* sc1 */));
                            if (g.y == (uint32_t)epoch) {
                                tok = (int)g.x;
                                break;
                            }
                            if (dead || spins > ENG_SPIN_MAX) {
                                if (!dead && lane == 0) atomicOr(a.ws + 1, 32);
                                dead = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    if (a.d_forced) {
                        const int f = a.d_forced[S.pos];
                        if (f >= 0) tok = f;
                    }
                    if (tok < 0 || tok >= a.emb_rows) tok = 0;
                    x0 = a.emb + (size_t)tok * C::DIM;
                }
                eng_poll_stage<XCH, ND, P1::nBlk, true, true, C::F32X>(nullptr, x0, tag, ly.norm_in, a.eps, L.xs[0], L.xrawA, lane, a.ws, dead, &sw[0], nullptr, 0, 0);
            } else {
                eng_poll_stage<XCH, ND, P1::nBlk, true, false, C::F32X>(a.xch + C::xA, nullptr, tag, ly.norm_in, a.eps, L.xs[0], L.xrawA, lane, a.ws, dead, &sw[0], has6 ? L.pub + 3 : nullptr, l,
                                                               a.delay[0]);
            }
        }
        ENG_STAMP(0, 1);
        __syncthreads();
        if (P1_SHARE && has1) { /* this wave's P1 rows, then the workgroup's publish like every other owner */
            w1.run(S.s1, NWV - 1, lane, S.M1, L.xs[0], 0xffffffffu, r1, [&](int row, float v, float) { L.outb[row - row0_1] = (tag << 16) | (uint32_t)f2bf(v); });
            if (XMAP)
                wg_publish(L, 0, eng_lqkv<C>(a, S.xcc), S.q_out0, P1::R, NWP1, lane, true);
            else
                wg_publish(L, 0, a.xch + C::qkv, S.q_out0, P1::R, NWP1, lane);
        }
        // P2: q heads of the group (GQ*hd granules), then k and v of the kv-head side by side in one piece
        if (S.has_unit) {
            // XCD-mapped form: the rows were published by workgroups of this XCD with plain stores into its own dense buffer [q GQ*hd | k hd | v hd]
            const __amdgpu_buffer_rsrc_t rs = XMAP ? eng_rsrc(eng_lqkv<C>(a, S.xcc), (uint32_t)(GQ * hd + 2 * hd) * 4u) : eng_rsrc(a.xch + C::qkv, (uint32_t)(C::QD + 2 * C::KVD) * 4u);
            const int q_src = XMAP ? 0 : S.h0 * hd;
            const uint32_t tagw = tag << 16;
            constexpr int NLQ = (GQ * hd + 255) / 256;
            u32x4 g[NLQ], gk;
            const int e_kv = 4 * lane; /* < hd: k, < 2hd: v */
            const bool kv_in = e_kv < 2 * hd;
            const int kv_src = XMAP ? GQ * hd + e_kv : (e_kv < hd ? C::QD + S.kvh * hd + e_kv : C::QD + C::KVD + S.kvh * hd + (e_kv - hd));
            eng_wait_pub(has1 ? L.pub + 0 : nullptr, l + 1, a.delay[1], dead);
            for (int spins = 0;; spins++) {
                uint32_t bad = 0;
#pragma unroll
                for (int r = 0; r < NLQ; r++) g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (q_src + 4 * (r * 64 + lane)) * 4, 0, 16));
                gk = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (kv_in ? kv_src : 0) * 4, 0, 16));
#pragma unroll
                for (int r = 0; r < NLQ; r++) bad = (4 * (r * 64 + lane) < GQ * hd) ? tags_bad(g[r], tagw, bad) : bad;
                bad = kv_in ? tags_bad(gk, tagw, bad) : bad;
                if (all_good(bad)) {
                    swt[1] += spins + 1;
                    break;
                }
                if (dead || spins > ENG_SPIN_MAX) {
                    if (!dead && lane == 0) atomicOr(a.ws + 1, 2);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int r = 0; r < NLQ; r++) {
                const int e0 = 4 * (r * 64 + lane);
                if (e0 < GQ * hd) *reinterpret_cast<u32x2*>(L.qraw + e0) = u32x2{(g[r].x & 0xffffu) | (g[r].y << 16), (g[r].z & 0xffffu) | (g[r].w << 16)};
            }
            if (kv_in) *reinterpret_cast<u32x2*>(L.kraw + e_kv) = u32x2{(gk.x & 0xffffu) | (gk.y << 16), (gk.z & 0xffffu) | (gk.w << 16)}; /* vraw = kraw + hd */
            ENG_STAMP(0, 2);
            if (P1_SHARE) { /* the next layer's P1 blocks of this wave: unconditional (the last layer requests its own again), and HERE -- loads return in order, so a
                               request in front of a poll holds that poll's sweep back by an HBM latency; the attention phase waits for nothing younger than its tiles */
                const int ln = l + 1 < a.n_layer ? l + 1 : l;
                mv_prefetch<P1, NCW1, FMT, S1>(mat1(ln), mat1(ln), S.s1, NWV - 1, lane, S.M1, r1);
            }
            eng_attn_phase<C>(a, L, S, ly, gen, tag, NWV - 1, lane, T, l, wg);
            /* the next layer's tiles (the last layer asks for its own again).  Round 4 moved this request of the poller behind the merge's sweep (loads return in order: the sweep's
               answer might wait for these sixteen HBM lines) and, with a second register set, in front of the attention phase: 387 / 392 us per launch against 383 / 390 here --
               the ao store then queues behind the tile requests, and the earlier sweep mostly comes too early */
            if (!S.empty) eng_attn_issue<C>(a, L.lay[l + 1 < a.n_layer ? l + 1 : l], S, NWV - 1, lane, T, 0);
        } else if (P1_SHARE) { /* a workgroup without an attention slice at this position: the same request, nothing to poll in front of it */
            const int ln = l + 1 < a.n_layer ? l + 1 : l;
            mv_prefetch<P1, NCW1, FMT, S1>(mat1(ln), mat1(ln), S.s1, NWV - 1, lane, S.M1, r1);
        }
        // P3: merge the slices of this workgroup's output elements: exact rescales to the largest exponent, fp64 sums, one division (kf_attn_common.h)
        ENG_STAMP(0, 3);
        if (S.has_merge) {
            constexpr int ME = C::ME, NLM = C::NLM, MAXSP = KF_ATTN_MAX_SPLITS;
            const int nsp = S.nsp, h = S.me0 >> hd_log2, dd = S.me0 & (hd - 1); /* XCD-mapped form: me0 counts inside the XCD's GQ heads */
            const unsigned long long* hbase = XMAP ? eng_lpart<C>(a, S.xcc) + (size_t)h * C::PSH : reinterpret_cast<const unsigned long long*>(a.xch + C::part) + (size_t)h * C::PSH;
            const __amdgpu_buffer_rsrc_t rs_o = eng_rsrc(hbase + (size_t)(dd / ME) * (MAXSP * ME * 2), (uint32_t)(MAXSP * ME * 2) * 8u);
            const __amdgpu_buffer_rsrc_t rs_ml = eng_rsrc(hbase + (size_t)hd * MAXSP * 2, (uint32_t)MAXSP * 32u);
            const int cnt = nsp * ME; /* values of this workgroup's elements: index sp * ME + e, two granules each */
            u32x4 go[NLM], gm0, gm1;
            const bool mine = lane < nsp;
            eng_wait_pub(nullptr, 0, a.delay[2], dead);
            ENG_STAMP(0, 9);
            int msw = 0;
            for (int spins = 0;; spins++) {
                uint32_t bad = 0;
                msw = spins + 1;
#pragma unroll
                for (int r = 0; r < NLM; r++) go[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_o, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
                gm0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ml, (mine ? lane : 0) * 32, 0, 16));
                gm1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ml, (mine ? lane : 0) * 32 + 16, 0, 16));
#pragma unroll
                for (int r = 0; r < NLM; r++) bad |= (r * 64 + lane) < cnt ? ((go[r].y ^ gen) | (go[r].w ^ gen)) : 0u;
                bad |= mine ? ((gm0.y ^ gen) | (gm0.w ^ gen) | (gm1.y ^ gen)) : 0u;
                if (all_good(bad)) break;
                if (dead || spins > ENG_SPIN_MAX) {
                    if (!dead && lane == 0) atomicOr(a.ws + 1, 4);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            ENG_STAMP(0, 11);
            swt[2] += msw;
            if (DBG && wg == a.dbg_wg && lane == 0) a.dbg[((size_t)l * 2) * 16 + 10] = (unsigned long long)msw;
            // transpose through LDS: element e's values over the slices contiguous for lane e (slices past nsp: 0)
#pragma unroll
            for (int r = 0; r < NLM; r++) {
                const int vi = r * 64 + lane;
                if (vi < MAXSP * ME) {
                    const int sp = vi / ME, e = vi - sp * ME;
                    L.msc[e * MAXSP + sp] = vi < cnt ? __builtin_bit_cast(double, ((unsigned long long)go[r].z << 32) | go[r].x) : 0.0;
                }
            }
            const float ms = mine ? __uint_as_float(gm0.x) : -__builtin_inff();
            const double ls = mine ? __builtin_bit_cast(double, ((unsigned long long)gm1.x << 32) | gm0.z) : 0.0;
            const float Mx = wave_max(ms);
            const int sh = canon_shift(ms - Mx);
            const double Lt = wave_sum_f64_fast(ldexp_d(ls, sh));
            // element e's sum over the slices by FOUR lanes (lane = 4 e + quarter: 8 slices each, then two quad adds): fp64 sums do not depend on their order
            int* shl = reinterpret_cast<int*>(L.msc + ME * MAXSP + 32); /* the slices' shifts, for lanes that walk other slices than their own */
            if (lane < MAXSP) shl[lane] = sh;
            double o = 0.0;
            {
                constexpr int QS = MAXSP / 4;
                const int e = (4 * ME <= 64) ? (lane >> 2) : lane, qq = (4 * ME <= 64) ? (lane & 3) : 0;
                const bool on = e < ME;
                const double* mv = L.msc + (on ? e : 0) * MAXSP + qq * QS;
                const int* sv = shl + qq * QS;
                if (4 * ME <= 64) {
#pragma unroll
                    for (int k = 0; k < QS; k++) o += ldexp_d(mv[k], sv[k]);
                    o += dpp_d<0xB1>(o); /* quad_perm xor 1 */
                    o += dpp_d<0x4E>(o); /* quad_perm xor 2 */
                } else {
#pragma unroll
                    for (int k = 0; k < MAXSP; k++) o += ldexp_d(mv[k], shl[k]);
                }
            }
            const uint32_t gr = (tag << 16) | (uint32_t)f2bf((float)(o / Lt));
            // lane 4 e (or e) holds element e's granule
            const bool holder = (4 * ME <= 64) ? ((lane & 3) == 0 && (lane >> 2) < ME) : (lane < ME);
            const int he = (4 * ME <= 64) ? (lane >> 2) : lane;
            uint32_t* const ao_dst = a.xch + C::ao + (XMAP ? S.h0 * hd : 0) + S.me0;
            if (ME >= 4) { /* 16 bytes per lane: a 4-byte write-through store is a read-modify-write at the memory side */
                uint32_t* mo = reinterpret_cast<uint32_t*>(L.msc + ME * MAXSP);
                if (holder) mo[he] = gr;
                if (4 * lane < ME) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(mo + 4 * lane), eng_rsrc(ao_dst, (uint32_t)ME * 4u), lane * 16, 0, 16 /* sc1 */);
            } else if (holder) {
                __hip_atomic_store(ao_dst + he, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // P4, P5 (P6 adds that x as the residual), P6
        ENG_STAMP(0, 4);
        if (ENG_COOP) { /* the ao vector: timed by this wave, swept by all eight */
            if (has4) eng_wait_pub(nullptr, 0, a.delay[3], dead);
            __syncthreads();
            if (has4) eng_poll_stage_part<XCH, NQD, P4::nBlk, NWV, C::F32X>(a.xch + C::ao, tag, L.xs[1], NWV - 1, lane, a.ws, dead, &sw[1]);
        } else if (has4) {
            eng_poll_stage<XCH, NQD, P4::nBlk, false, false, C::F32X>(a.xch + C::ao, nullptr, tag, nullptr, 0.f, L.xs[1], nullptr, lane, a.ws, dead, &sw[1], nullptr, 0, a.delay[3]);
        }
        ENG_STAMP(0, 5);
        __syncthreads();
        if (has5 || has6)
            eng_poll_stage<XCH, ND, P5::nBlk, true, false, C::F32X>(a.xch + C::xB, nullptr, tag, ly.norm_post, a.eps, L.xs[0], L.xrawB, lane, a.ws, dead, &sw[2], has4 ? L.pub + 1 : nullptr, l + 1,
                                                           a.delay[4]);
        ENG_STAMP(0, 6);
        __syncthreads();
        if (ENG_COOP) {
            if (has6) eng_wait_pub(has5 ? L.pub + 2 : nullptr, l + 1, a.delay[5], dead);
            __syncthreads();
            if (has6) eng_poll_stage_part<XCH, NF, P6::nBlk, NWV, C::F32X>(a.xch + C::act, tag, L.xs[1], NWV - 1, lane, a.ws, dead, &sw[3]);
        } else if (has6) {
            eng_poll_stage<XCH, NF, P6::nBlk, false, false, C::F32X>(a.xch + C::act, nullptr, tag, nullptr, 0.f, L.xs[1], nullptr, lane, a.ws, dead, &sw[3], has5 ? L.pub + 2 : nullptr, l + 1, a.delay[5]);
        }
        ENG_STAMP(0, 7);
        swt[0] += sw[0], swt[3] += sw[1], swt[4] += sw[2], swt[5] += sw[3];
        if (DBG && wg == a.dbg_wg && lane == 0)
            a.dbg[((size_t)l * 2) * 16 + 8] = (unsigned long long)sw[0] | ((unsigned long long)sw[1] << 16) | ((unsigned long long)sw[2] << 32) | ((unsigned long long)sw[3] << 48);
        __syncthreads();
    }
    // the engine's statistics (kf_engine_stats): sweeps the pollers of workgroup 0 (cross-XCD view) needed per hand-off, and the layers they cover
    if (wg == 0 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) atomicAdd(a.ws + 8 + i, swt[i]);
        atomicAdd(a.ws + 14, a.n_layer);
    }
}

// the compute waves
template <class C>
__device__ __forceinline__ void eng_compute_main(const EngArgs& a, const EngLds& L, const EngSlice& S, int epoch, int wg, int wave, int lane) {
    using SH = typename C::SH;
    using P1 = typename SH::P1;
    using P4 = typename SH::P4;
    using P5 = typename SH::P5;
    using P6 = typename SH::P6;
    constexpr int FMT = C::FMT, NWV = C::NWV;
    constexpr bool XMAP = C::XMAP, DBG = C::DBG;
    constexpr int NCW = NWV - 1;
    // P1 alone is shared with the poller wave (it is idle between staging x and the first q/k/v granules): NWV waves, so that the 0.6B shape's
    // 8 row-slots per workgroup are one step for every wave instead of two for wave 0
    constexpr int NCW1 = ENG_P1_SHARE ? NWV : NWV - 1;
    constexpr int S1 = c_maxs<P1, NCW1>(), S4 = c_maxs<P4, NCW>(), S5 = c_maxs<P5, NCW>(), S6 = c_maxs<P6, NCW>();
    // this workgroup's rows per phase (contiguous in the phase's output vector) and the waves that own some of them
    static_assert(P1::R % 4 == 0 && P4::R % 4 == 0 && P5::R % 4 == 0 && P6::R % 4 == 0 && P1::R <= 64 && P4::R <= 64 && P5::R <= 64 && P6::R <= 64, "rows per workgroup");
    static_assert(P1::total % P1::spg == 0 && P4::total % P4::spg == 0 && P5::total % P5::spg == 0 && P6::total % P6::spg == 0, "whole workgroups");
    static_assert(P1::S1 % P1::spg == 0 && P1::S2 % P1::spg == 0, "a workgroup's P1 rows belong to one matrix");
    constexpr int NWP1 = P1::spg < NCW1 ? P1::spg : NCW1, NWP4 = P4::spg < NCW ? P4::spg : NCW, NWP5 = P5::spg < NCW ? P5::spg : NCW, NWP6 = P6::spg < NCW ? P6::spg : NCW;
    const bool has1 = XMAP ? true : wg * P1::spg < P1::total, has4 = wg * P4::spg < P4::total, has5 = wg * P5::spg < P5::total, has6 = wg * P6::spg < P6::total;
    const float qb1 = S.j1 == 0 ? a.qbias[0] : (S.j1 == 1 ? a.qbias[1] : a.qbias[2]);
    const int row0_1 = S.s1 * P1::RPS;
    auto mat1 = [&](int l) {
        return L.lay[l].m[S.j1];
    };

    bool dead = false; /* a timed-out sweep of this wave: the error word is set, later sweeps return at once */
    MvRegs<false, S1> r1;
    MvRegs<false, S4> r4;
    MvRegs<true, S5> r5;
    MvRegs<false, S6> r6;
    EngAttnState<C> T;
#pragma unroll
    for (int u = 0; u < EngAttnState<C>::U; u++) T.kk[u] = T.vv[u] = u32x4{0, 0, 0, 0};
    mv_prefetch<P1, NCW1, FMT, S1>(mat1(0), mat1(0), S.s1, wave, lane, S.M1, r1);
    eng_attn_rope<C>(a, S, lane, T);
    if (S.has_unit && !S.empty) eng_attn_issue<C>(a, L.lay[0], S, wave, lane, T, 0);

    for (int l = 0; l < a.n_layer; l++) {
        const EngLayer& ly = L.lay[l];
        const uint32_t gen = (uint32_t)epoch * (uint32_t)a.n_layer + (uint32_t)l, tag = gen & 0xffffu, tag_next = (gen + 1u) & 0xffffu;
        const bool last = l == a.n_layer - 1;
        const int ln = last ? l : l + 1; /* the layer whose blocks are requested next: the last layer asks for its own again (unconditional requests keep every wait counted) */
        int lzm;
        asm volatile("s_mov_b32 %0, 0" : "=s"(lzm));
        const int lane_m = lane + lzm; /* as for the attention phase below: the mat-vec phases' per-lane indices are recomputed per layer rather than held in registers */
        if (S.has_unit && !S.empty) eng_attn_normw<C>(ly, wave, lane_m, T);
        // ================= P1: RMSNorm(x) -> Q, K, V rows  (every phase: the blocks are dequantised in front of the barrier the activations arrive behind)
        MvPhase<C, P1, NCW1, S1> w1; /* declared per layer: nothing of it is carried around the loop */
        w1.ahead(qb1, 0.f, wave, lane_m, r1, L.q1tab);
        __syncthreads();
        if (wave == 0) ENG_STAMP(1, 0);
        mv_prefetch<P4, NCW, FMT, S4>(ly.m[3], ly.m[3], wg * P4::spg, wave, lane_m, P4::M0, r4);
        w1.run(S.s1, wave, lane_m, S.M1, L.xs[0], 0xffffffffu, r1, [&](int row, float v, float) { L.outb[row - row0_1] = (tag << 16) | (uint32_t)f2bf(v); });
        if (has1 && wave < NWP1) {
            if (XMAP)
                wg_publish(L, 0, eng_lqkv<C>(a, S.xcc), S.q_out0, P1::R, NWP1, lane_m, true);
            else
                wg_publish(L, 0, a.xch + C::qkv, S.q_out0, P1::R, NWP1, lane_m);
        }
        // ================= P2: q/k-norm + RoPE + attention over this workgroup's slice (all 8 waves: eng_attn_phase), then the next layer's K/V tiles of the
        // slice (they do not depend on this token, except row `pos`, which is substituted; the last layer requests its own again)
        if (wave == 0) ENG_STAMP(1, 1);
        if (S.has_unit) {
            // the attention phase's per-lane addresses and offsets are layer-invariant, and hoisted out of the layer loop they would occupy ~50 registers through the
            // mat-vec phases (which then spill): an opaque zero added to the lane id makes them a few dozen instructions per layer instead
            int lz;
            asm volatile("s_mov_b32 %0, 0" : "=s"(lz));
            const int lane_a = lane + lz;
            eng_attn_phase<C>(a, L, S, ly, gen, tag, wave, lane_a, T, l, wg);
            if (wave == 0) ENG_STAMP(1, 3);
            if (DBG && wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ENG_STAMP(1, 10); /* partial stores acknowledged */
            }
            if (!S.empty) eng_attn_issue<C>(a, L.lay[ln], S, wave, lane_a, T, 0);
        }

        // ================= P4: o_proj + residual -> xB
        MvPhase<C, P4, NCW, S4> w4;
        w4.ahead(a.qbias[3], 0.f, wave, lane_m, r4, L.q1tab);
        if (ENG_COOP) {
            __syncthreads(); /* the poller has timed the first sweep */
            if (has4) eng_poll_stage_part<C::XCH, C::QD / 256, P4::nBlk, NWV, C::F32X>(a.xch + C::ao, tag, L.xs[1], wave, lane_m, a.ws, dead);
        }
        __syncthreads();
        if (wave == 0) ENG_STAMP(1, 4);
        const uint32_t hot5 = L.hotbits[l]; /* the sparse forward's mask of this workgroup's gate / up rows (all ones: dense) */
        mv_prefetch<P5, NCW, FMT, S5>(ly.m[4], ly.m[5], wg * P5::spg, wave, lane_m, P5::M0, r5, hot5);
        w4.run(wg * P4::spg, wave, lane_m, P4::M0, L.xs[1], 0xffffffffu, r4, [&](int row, float v, float) {
            const uint16_t o = f2bf(v);
            L.outb[row - wg * P4::R] = (tag << 16) | (uint32_t)f2bf(bf2f(L.xrawA[row]) + bf2f(o)); /* CU_add3: bf16(x + bf16(W.x)) */
        });
        if (has4 && wave < NWP4) wg_publish(L, 1, a.xch + C::xB, wg * P4::R, P4::R, NWP4, lane_m);
        // ================= P5: RMSNorm + gate/up + SwiGLU -> act
        if (wave == 0) ENG_STAMP(1, 5);
        MvPhase<C, P5, NCW, S5> w5;
        w5.ahead(a.qbias[4], a.qbias[5], wave, lane_m, r5, L.q1tab);
        __syncthreads();
        if (wave == 0) ENG_STAMP(1, 6);
        mv_prefetch<P6, NCW, FMT, S6>(ly.m[6], ly.m[6], wg * P6::spg, wave, lane_m, P6::M0, r6);
        w5.run(wg * P5::spg, wave, lane_m, P5::M0, L.xs[0], L.hotbits[l], r5, [&](int row, float v, float v2) {
            const float gt = round_bf16(v), up = round_bf16(v2); /* CU_swiglu_v0 on the two bf16-rounded projections */
            L.outb[row - wg * P5::R] = (tag << 16) | (uint32_t)f2bf((gt * up) / (1.0f + kf_expf(-gt)));
        });
        if (has5 && wave < NWP5) wg_publish(L, 2, a.xch + C::act, wg * P5::R, P5::R, NWP5, lane_m);
        // ================= P6: down_proj + residual -> x of the next layer
        if (wave == 0) ENG_STAMP(1, 7);
        MvPhase<C, P6, NCW, S6> w6;
        w6.ahead(a.qbias[6], 0.f, wave, lane_m, r6, L.q1tab);
        if (ENG_COOP) {
            __syncthreads();
            if (has6) eng_poll_stage_part<C::XCH, C::FFN / 256, P6::nBlk, NWV, C::F32X>(a.xch + C::act, tag, L.xs[1], wave, lane_m, a.ws, dead);
        }
        __syncthreads();
        if (wave == 0) ENG_STAMP(1, 8);
        mv_prefetch<P1, NCW1, FMT, S1>(mat1(ln), mat1(ln), S.s1, wave, lane_m, S.M1, r1);
        w6.run(wg * P6::spg, wave, lane_m, P6::M0, L.xs[1], 0xffffffffu, r6, [&](int row, float v, float) {
            const uint16_t o = f2bf(v);
            const uint16_t y = f2bf(bf2f(L.xrawB[row]) + bf2f(o));
            if (last) a.x_out[row] = y;
            L.outb[row - wg * P6::R] = (tag_next << 16) | (uint32_t)y;
        });
        if ((!last || a.head_on) && has6 && wave < NWP6) wg_publish(L, 3, a.xch + C::xA, wg * P6::R, P6::R, NWP6, lane_m);
        if (wave == 0) ENG_STAMP(1, 9);
    }
}


// ---- the LM head as trailing phases of the launch (Head4Token::cuInfer_1, NeuronFuse.cu:842-862: final RMSNorm, the [vocab, dim] mat-vec, first-maximum arg-max,
// GoPT.cpp:602-612): what kf_norm_lm_head + argmax_finish_kernel do as two more launches.  All 8 waves stream: a wave owns the row slots wave, wave + 8, ... of the
// workgroup's contiguous slot range, keeps HG slots (2 * HG 16-byte loads per lane) in flight, and holds its two x blocks in registers (a lane multiplies the same
// block columns of every row).  The arithmetic is gemv_kernel<FMT_BF16>'s: same lanes per row, same per-lane chain, same tree, bf16 store, arg-max over the stored values.
template <class C>
__device__ __forceinline__ void eng_head_main(const EngArgs& a, const EngLds& L, int epoch, bool more_steps, int wg, int wave, int lane) {
    constexpr int NWV = C::NWV, NWG = C::NWG, nBlk = C::HnBlk, LPR = C::HLPR, RPS = C::HRPS, ITERS = C::Hiters, HG = 4;
    constexpr int ND = C::DIM / 256;
    const int sub = lane >> C::Hlpr_log2, ll = lane & (LPR - 1);
    const int total = (a.vocab + RPS - 1) / RPS, spg = (total + NWG - 1) / NWG; /* slots in all, per workgroup */
    const int s_wg = wg * spg;
    int s_end = s_wg + spg;
    s_end = s_end < total ? s_end : total;
    const int nmine = s_end > s_wg + wave ? (s_end - s_wg - wave + NWV - 1) / NWV : 0; /* this wave's slots: s_wg + wave + NWV * i */
    const int nbatch = (nmine + HG - 1) / HG;
    u32x4 w[2][HG][ITERS];
    auto issue = [&](int b, int buf) {
#pragma unroll
        for (int g = 0; g < HG; g++) {
            int i = b * HG + g;
            i = i < nmine ? i : (nmine > 0 ? nmine - 1 : 0); /* unconditional requests (a conditional one would turn the waits behind it into drains): the tail re-reads the wave's last rows */
            int row = (s_wg + wave + NWV * i) * RPS + sub;
            row = row < a.vocab ? row : a.vocab - 1;
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                int col = it * LPR + ll;
                col = col < nBlk ? col : nBlk - 1;
                w[buf][g][it] = __builtin_nontemporal_load(a.head_w + (size_t)row * nBlk + col);
            }
        }
    };
    issue(0, 0); /* ahead of the hand-off of x: the first rows' HBM latency runs under it */
    const uint32_t gen = (uint32_t)(epoch + 1) * (uint32_t)a.n_layer, tag = gen & 0xffffu; /* the generation the last layer's down_proj published its rows with */
    if (wave == NWV - 1) {
        bool dead = false;
        const bool has6 = wg * C::SH::P6::spg < C::SH::P6::total;
        eng_poll_stage<1, ND, nBlk, true, false, false>(a.xch + C::xA, nullptr, tag, a.head_norm, a.eps, L.xs[0], nullptr, lane, a.ws, dead, nullptr, has6 ? L.pub + 3 : nullptr, a.n_layer,
                                                 a.delay[0]);
    }
    __syncthreads();
    float xf[ITERS][8]; /* this lane's block columns of the normed x, widened once: the lane multiplies the same columns of every row */
    uint32_t xp[ITERS][4]; /* the same as bf16 pairs (the v_dot2c form) */
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        int col = it * LPR + ll;
        col = col < nBlk ? col : nBlk - 1;
        const u32x4 xv = L.xs[0][col];
        const uint32_t x4[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int i = 0; i < 4; i++) xf[it][2 * i] = bf_lo(x4[i]), xf[it][2 * i + 1] = bf_hi(x4[i]), xp[it][i] = x4[i];
    }
    float best_v = -__builtin_inff();
    int best_i = 0x7fffffff;
    auto compute = [&](int b, int buf) {
#pragma unroll
        for (int g = 0; g < HG; g++) {
            const int i = b * HG + g;
            const int row = (s_wg + wave + NWV * i) * RPS + sub;
            acc_t<C::CANON> acc{};
#pragma unroll
            for (int it = 0; it < ITERS; it++) { /* BlockDot<FMT_BF16, CANON>: the block's 4 pairs in order (canonical: the even / odd chains, one v_pk_fma_f32 per pair) */
                const uint32_t w4[4] = {w[buf][g][it].x, w[buf][g][it].y, w[buf][g][it].z, w[buf][g][it].w};
                acc_t<C::CANON> r = acc;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if constexpr (C::CANON) r = pk_fma(f32x2_t{bf_lo(w4[i]), bf_hi(w4[i])}, f32x2_t{xf[it][2 * i], xf[it][2 * i + 1]}, r);
                    else r = dot2_bf16(w4[i], xp[it][i], r);
                }
                acc = acc_pick(it * LPR + ll < nBlk, r, acc);
            }
            const float v = group_sum(acc_join(acc), C::Hlpr_log2);
            if (ll == 0 && i < nmine && row < a.vocab) {
                const uint16_t o = f2bf(v);
                a.logits[row] = o;
                const float fv = bf2f(o);
                if (fv > best_v || (fv == best_v && row < best_i)) best_v = fv, best_i = row;
            }
        }
    };
    for (int b = 0; b < nbatch; b += 2) { /* two batches ping-pong: no register copies between steps */
        issue(b + 1, 1);
        compute(b, 0);
        if (b + 1 >= nbatch) break;
        issue(b + 2, 0);
        compute(b + 1, 1);
    }
    // first maximum over the workgroup's rows, published as one 8-byte granule {tag, bf16 value, row}; workgroup 0's poller picks over all of them
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const float ov = __shfl_xor(best_v, m, 64);
        const int oi = __shfl_xor(best_i, m, 64);
        if (ov > best_v || (ov == best_v && oi < best_i)) best_v = ov, best_i = oi;
    }
    float* rv = L.wmax; /* [NWV] values, [NWV] rows behind them (the attention scratch is free now) */
    int* ri = reinterpret_cast<int*>(L.wmax + NWV);
    if (lane == 0) rv[wave] = best_v, ri[wave] = best_i;
    __syncthreads();
    unsigned long long* hb = reinterpret_cast<unsigned long long*>(a.xch + C::hbest);
    if (wave == 0 && lane == 0) {
        for (int k = 1; k < NWV; k++)
            if (rv[k] > best_v || (rv[k] == best_v && ri[k] < best_i)) best_v = rv[k], best_i = ri[k];
        /* a workgroup without rows publishes (-inf, 0x7fffffff): it never wins */
        const unsigned long long gr = ((unsigned long long)((tag << 16) | (uint32_t)f2bf(best_v)) << 32) | (unsigned long long)(uint32_t)best_i;
        __hip_atomic_store(hb + wg, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wg == 0 && wave == NWV - 1 && a.d_state_w) { /* the pick (argmax_finish_kernel): NWG granules, 4 per lane */
        static_assert(NWG == 256, "four granules per lane");
        const __amdgpu_buffer_rsrc_t rs = eng_rsrc(hb, NWG * 8u);
        u32x4 g0, g1;
        bool ok = false;
        for (int z = 0; z < 24; z++) __builtin_amdgcn_s_sleep(1);
        for (int spins = 0; spins <= ENG_SPIN_MAX; spins++) {
            g0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 32, 0, 16 /* sc1 */));
            g1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 32 + 16, 0, 16));
            uint32_t bad = ((g0.y >> 16) ^ tag) | ((g0.w >> 16) ^ tag) | ((g1.y >> 16) ^ tag) | ((g1.w >> 16) ^ tag);
            if (all_good(bad)) {
                ok = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        const uint32_t hv[4] = {g0.y, g0.w, g1.y, g1.w}, hi[4] = {g0.x, g0.z, g1.x, g1.z};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float fv = bf2f((uint16_t)(hv[k] & 0xffffu));
            const int ix = (int)hi[k];
            if (fv > bv || (fv == bv && ix < bi)) bv = fv, bi = ix;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const float ov = __shfl_xor(bv, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
        }
        if (lane == 0) {
            const int err = __hip_atomic_load(a.ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok && err == 0 && bi >= 0 && bi < a.vocab) { /* never advance the decode state on a timed-out hand-off */
                const int p = a.d_state_w[1];
                if (a.d_tokens_out) a.d_tokens_out[p] = bi;
                a.d_state_w[0] = bi;
                a.d_state_w[1] = p + 1;
                if (more_steps) /* the next step of this launch starts from this id */
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)bi, (uint32_t)(epoch + 1)}, eng_rsrc(a.xch + C::tokg, 8u), 0, 0, 16 /* sc1 */);
            } else if (err == 0) {
                atomicOr(a.ws + 1, 16);
            }
        }
    }
}

// DIM, QD, KVD, FFN: the model's dim, q_dim, kv_dim, ffn; NWG: the grid (= CUs): sweeps and mat-vec geometry are straight-line code
template <class C>
__global__ void __launch_bounds__(C::NWV * 64) engine_kernel(const EngArgs a) {
    constexpr int hd = C::HD, GQ = C::GQ, NWV = C::NWV;
    constexpr bool XMAP = C::XMAP;
    using P1 = typename C::SH::P1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6) /* scalar: everything derived from it stays out of the vector registers */, wg = blockIdx.x;
    // ---- LDS carve
    // every buffer at a compile-time offset (the addresses fold into the LDS instructions' offset fields instead of living in registers through the whole launch);
    // the layer table, whose size is a run-time figure, lies behind them
    EngLds L;
    constexpr int maxK = C::DIM > C::QD ? (C::DIM > C::FFN ? C::DIM : C::FFN) : (C::QD > C::FFN ? C::QD : C::FFN);
    constexpr int xs_bytes = (maxK * 4 + 15) & ~15, xr_bytes = (C::DIM * 2 + 15) & ~15; /* xs: fp32 activations */
    constexpr size_t off = (size_t)2 * xs_bytes + 2 * xr_bytes;
    constexpr size_t fixed0 = (off + sizeof(uint16_t) * ((size_t)2 * GQ * hd + 3 * hd) + sizeof(double) * ((size_t)NWV * GQ * (hd + 2) + (size_t)C::ME * KF_ATTN_MAX_SPLITS + 64) + 4 * 16 + 4 * 64 + 32 + 15) & ~(size_t)15;
    constexpr size_t fixed_bytes = fixed0 + (C::FMT == FMT_Q1T ? 4096 : (C::FMT == FMT_Q2T ? 2048 : 0)); /* the 1-bit / 2-bit selector table */
    EngLayer* lay = reinterpret_cast<EngLayer*>(smem + fixed_bytes);
    L.lay = lay;
    L.q1tab = reinterpret_cast<const u32x4*>(smem + fixed0);
    uint32_t* hotbits = reinterpret_cast<uint32_t*>(smem + fixed_bytes + (((size_t)a.n_layer * sizeof(EngLayer) + 15) & ~(size_t)15)); /* [n_layer], behind the layer table */
    L.hotbits = hotbits;
    L.xs[0] = reinterpret_cast<u32x4*>(smem);
    L.xs[1] = reinterpret_cast<u32x4*>(smem + xs_bytes);
    L.xrawA = reinterpret_cast<uint16_t*>(smem + 2 * xs_bytes);
    L.xrawB = reinterpret_cast<uint16_t*>(smem + 2 * xs_bytes + xr_bytes);
    L.qraw = reinterpret_cast<uint16_t*>(smem + off); /* [GQ][hd] raw q heads of this workgroup's slice */
    L.kraw = L.qraw + GQ * hd, L.vraw = L.kraw + hd, L.qb = L.vraw + hd, L.knew = L.qb + GQ * hd;
    L.comb = reinterpret_cast<double*>(L.knew + hd); /* [NWV][GQ][hd + 2] doubles (the offset is a multiple of 16 bytes) */
    L.msc = L.comb + NWV * GQ * (hd + 2);             /* [ME][MAXSP] doubles + [64] dwords */
    L.wmax = reinterpret_cast<float*>(L.msc + C::ME * KF_ATTN_MAX_SPLITS + 64);
    L.outb = reinterpret_cast<uint32_t*>(L.wmax + 16);
    L.cnt = reinterpret_cast<int*>(L.outb + 64); /* [0] arrival counter, [1..2] XCD id and ticket */
    L.pub = L.cnt + 4;
    if (tid == 0) *L.cnt = 0;
    if (tid < 4) L.pub[tid] = 0;
    // ---- start: state, generation, tables
    EngSlice S;
    const int pos0 = a.d_state[1];
    S.pos = pos0;
    const int epoch0 = a.ws[0];
    if (a.ws[1] != 0) return; /* an earlier launch timed out (it could not become resident): nothing runs until the host has cleared the word (engine_reset) */
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.layers);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lay);
        const int nw = a.n_layer * (int)(sizeof(EngLayer) / 4);
        for (int i = tid; i < nw; i += NWV * 64) dst[i] = src[i];
    }
    if constexpr (C::FMT == FMT_Q1T) { /* selector table of BlockDot<FMT_Q1T> (kf_gemv.hip fills the same): entry B, dword p = bytes {2a, 2a+1, 2b, 2b+1}, a / b = bits 7-2p / 6-2p of B */
        if (tid < 256) {
            uint32_t e[4];
#pragma unroll
            for (int p = 0; p < 4; p++) e[p] = 0x01000100u + 0x0202u * ((tid >> (7 - 2 * p)) & 1u) + 0x02020000u * ((tid >> (6 - 2 * p)) & 1u);
            reinterpret_cast<u32x4*>(smem + fixed0)[tid] = u32x4{e[0], e[1], e[2], e[3]};
        }
    }
    if constexpr (C::FMT == FMT_Q2T) { /* selector table of BlockDot<FMT_Q2T> (kf_gemv.hip fills the same): entry B, dword p = bytes {2q, 2q+1, 2q', 2q'+1}, q / q' = the levels of elements 2p, 2p+1 of byte B */
        if (tid < 256) {
            uint32_t e[2];
#pragma unroll
            for (int p = 0; p < 2; p++) e[p] = 0x01000100u + 0x0202u * ((tid >> (6 - 4 * p)) & 3u) + 0x02020000u * ((tid >> (4 - 4 * p)) & 3u);
            reinterpret_cast<u32x2*>(smem + fixed0)[tid] = u32x2{e[0], e[1]};
        }
    }
    { /* sparse forward: the hot bits of this workgroup's gate / up rows, per layer (CS_Picker's hot[] read once per launch) */
        using P5 = typename C::SH::P5;
        static_assert(P5::R <= 32, "hot bits of a workgroup's gate / up rows in one word");
        for (int l = tid; l < a.n_layer; l += NWV * 64) {
            g_i32 hot = a.layers[l].hot;
            uint32_t bits = 0xffffffffu;
            if (hot) {
                bits = 0;
                for (int r = 0; r < P5::R; r++) {
                    const int row = wg * P5::R + r;
                    if (row < P5::M0 && hot[row] == 1) bits |= 1u << r;
                }
            }
            hotbits[l] = bits;
        }
    }
    __syncthreads();
    S.len = S.pos + 1, S.nsp = a.nsp;
    S.xcc = 0, S.rank = 0;
    int s1_abs; /* first P1 slot of this workgroup in the q | k | v slot order */
    if (XMAP) {
        // kv-head k lives on XCD k: its q/k/v rows, its attention slices and its merge are exchanged through that XCD's L2 only.
        // A workgroup learns its XCD from the hardware register and takes a ticket there (the tickets are zeroed again at the end).
        static_assert(!XMAP || (GQ * hd + 2 * hd) == (C::NWG / 8) * P1::R, "an XCD's workgroups share the q | k | v rows of one kv-head");
        int* tickets = reinterpret_cast<int*>(a.loc);
        int* xi = L.cnt + 1;
        if (tid == 0) {
            const int x = eng_xcc();
            xi[0] = x, xi[1] = __hip_atomic_fetch_add(tickets + x * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        S.xcc = xi[0], S.rank = xi[1];
        if (S.rank >= C::NWG / 8 && tid == 0) atomicOr(a.ws + 1, 8); /* not NWG / 8 workgroups per XCD: the polls below time out, the error word says why */
        const int r = S.rank & (C::NWG / 8 - 1), lr0 = r * P1::R; /* first of this workgroup's rows in the XCD's [q | k | v] list */
        const int j = lr0 < GQ * hd ? 0 : (lr0 < GQ * hd + hd ? 1 : 2);
        const int row = j == 0 ? S.xcc * GQ * hd + lr0 : (j == 1 ? S.xcc * hd + (lr0 - GQ * hd) : S.xcc * hd + (lr0 - GQ * hd - hd));
        s1_abs = (j == 0 ? 0 : (j == 1 ? P1::S1 : P1::S2)) + row / P1::RPS;
        S.q_out0 = lr0;
        S.has_unit = r < a.nsp;
        S.kvh = S.xcc, S.split = r;
        S.me0 = r * a.merge_e;
        S.has_merge = a.nsp > 1 && S.me0 < GQ * hd;
    } else {
        s1_abs = wg * P1::spg;
        const int j = s1_abs >= P1::S2 ? 2 : (s1_abs >= P1::S1 ? 1 : 0);
        S.q_out0 = (j == 0 ? 0 : (j == 1 ? C::QD : C::QD + C::KVD)) + (s1_abs - (j == 0 ? 0 : (j == 1 ? P1::S1 : P1::S2))) * P1::RPS;
        S.has_unit = wg < C::n_kv * a.nsp;
        S.kvh = S.has_unit ? wg / a.nsp : 0, S.split = S.has_unit ? wg - S.kvh * a.nsp : 0;
        S.me0 = wg * a.merge_e;
        S.has_merge = a.nsp > 1 && S.me0 < C::n_head * hd;
    }
    S.j1 = s1_abs >= P1::S2 ? 2 : (s1_abs >= P1::S1 ? 1 : 0);
    S.s1 = s1_abs - (S.j1 == 0 ? 0 : (S.j1 == 1 ? P1::S1 : P1::S2));
    S.M1 = S.j1 == 0 ? P1::M0 : (S.j1 == 1 ? P1::M1 : P1::M2);
    if (s1_abs >= P1::total) S.M1 = 0; /* a workgroup without P1 rows (shapes with fewer slots than workgroups): every step masked */
    S.h0 = S.kvh * GQ, S.t0 = S.split * a.chunk;
    // the steps of this launch: positions pos0, pos0 + 1, ... with the slice geometry of the launch bound; a step's id reaches the next step's layer 0 as a tagged
    // granule (eng_poller_main), everything else a step needs from its predecessor is in the KV cache
    const int nst = a.n_steps > 1 ? a.n_steps : 1;
    if (pos0 + nst > a.nsp * a.chunk || pos0 + nst > a.max_seq) { /* a position of this launch lies beyond the keys its slices cover (the caller's pos_bound does not hold) or beyond the
                                                                      cache rows (nsp * chunk rounds the bound up): refuse, loudly */
        if (tid == 0) atomicOr(a.ws + 1, 64);
        return;
    }
    for (int step = 0; step < nst; step++) {
        const int epoch = epoch0 + step;
        S.pos = pos0 + step, S.len = S.pos + 1;
        S.t1 = S.t0 + a.chunk < S.len ? S.t0 + a.chunk : S.len;
        S.empty = S.t0 >= S.len;
        S.own_new = S.has_unit && S.pos >= S.t0 && S.pos < S.t1;
        if (step > 0) { /* the workgroup's own publish counters count layers of ONE step; a step behind a timed-out one does not start */
            __syncthreads();
            if (tid < 4) L.pub[tid] = 0;
            if (tid == 0) L.cnt[3] = __hip_atomic_load(a.ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (L.cnt[3] != 0) break;
        }
        int sz; /* as inside the layer loop: nothing derived from the lane id is to be hoisted out of the step loop and held in registers across a whole step */
        asm volatile("s_mov_b32 %0, 0" : "=s"(sz));
        const int lane_s = lane + sz;
        if (wave == NWV - 1)
            eng_poller_main<C>(a, L, S, epoch, step, wg, lane_s);
        else
            eng_compute_main<C>(a, L, S, epoch, wg, wave, lane_s);
        if (a.head_on) eng_head_main<C>(a, L, epoch, step + 1 < nst, wg, wave, lane_s);
    }
    // the next launch's generation (workgroup 0 owns rows of the last phase, so every workgroup has read the epoch long before)
    if (wg == 0 && tid == 0) {
        a.ws[0] = epoch0 + nst;
        if (XMAP)
            for (int i = 0; i < 8; i++) reinterpret_cast<int*>(a.loc)[i * 32] = 0; /* every workgroup took its ticket before any could finish a layer */
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct EngineHost {
    EngArgs args;
    EngPlan plans[4];
    int fmt, GQ, hd, nwv, shape_class, xmap;
    int canon; /* 1: mat-vec phases in the canonical order (engine_set_canonical) */
    int dim, q_dim, kv_dim, ffn, n_kv, n_head;
    size_t smem;
    int n_cu;
    void* xmem;     /* the hand-off vectors every CU sweeps: device memory that is cached nowhere (hipDeviceMallocUncached), owned by the engine */
    size_t xbytes;
    void* ws;
    size_t ws_bytes;
    // first-sweep delays per attention slice count (the hand-offs' timing changes with the number of slices a launch cuts the context into): [nsp][6], and whether
    // engine_tune has measured that row on this device (else: the defaults)
    int delay_tab[KF_ATTN_MAX_SPLITS + 1][6];
    unsigned char tuned[KF_ATTN_MAX_SPLITS + 1];
};

// the instantiated model shapes: {GQA group, head_dim, dim, q_dim, ffn}
static int engine_shape_class(int GQ, int hd, int dim, int q_dim, int ffn) { /* kv_dim = q_dim / GQ */
    if (GQ == 2 && hd == 128 && dim == 1024 && q_dim == 2048 && ffn == 3072) return 1; /* Qwen3-0.6B (BASELINE config 2) */
    if (GQ == 2 && hd == 64 && dim == 256 && q_dim == 256 && ffn == 512) return 2;     /* the small parity-test shape */
    if (GQ == 2 && hd == 128 && dim == 2048 && q_dim == 2048 && ffn == 6144) return 3;  /* Qwen3-1.7B: the 0.6B head geometry on a 2048-wide stream (4-bit register-table storage only) */
    return 0;
}
static_assert(sizeof(EngPlan) == 80 && sizeof(EngLayer) == 224, "device table strides");
static int eng_epb(int fmt) { return fmt == FMT_BF16 ? 8 : (fmt == FMT_F8 ? 16 : (fmt == FMT_Q4 || fmt == FMT_Q4P ? 32 : (32 /* 2-bit / 1-bit: the engine's lanes take a 32-weight piece of a block each (eng_vepb) */))); }

// geometry of one phase; returns false when the shape is outside what the engine serves
static bool eng_plan(EngPlan& P, int fmt, int K, int njobs, const kf_weight* const* w, bool paired, int n_wg, bool& q4p_ok) {
    const int epb = eng_epb(fmt);
    if (K % epb != 0 || K % 8 != 0 || K > ENG_MAXLD * 256) return false;
    memset(&P, 0, sizeof(P));
    P.K = K, P.nBlk = K / epb;
    long rows = 0;
    for (int j = 0; j < njobs; j++)
        if (!(paired && j == 1)) rows += w[j]->ne0;
    P.lpr_log2 = gemv_lpr_log2(P.nBlk, rows);
    const int LPR = 1 << P.lpr_log2, RPS = 64 / LPR;
    P.iters = (P.nBlk + LPR - 1) / LPR;
    P.njobs = paired ? 1 : njobs;
    int slots = 0;
    for (int j = 0; j < njobs; j++) {
        const kf_weight* wj = w[j];
        if (gemv_fmt_of(wj) != (fmt == FMT_Q4P ? FMT_Q4 : fmt) || wj->ne1 != K || wj->qzeros || wj->qscales) return false;
        if (((uintptr_t)wj->data & 15) != 0) return false;
        if (fmt >= FMT_Q4) {
            if (!wj->gama || wj->lGroup <= 0 || (wj->lGroup % epb) != 0 || ((long)wj->ne0 * wj->ne1) % wj->lGroup != 0) return false;
            const int bpg = wj->lGroup / epb;
            if (bpg < 1 || (bpg & (bpg - 1)) != 0) return false;
            P.gshift = __builtin_ctz(bpg);
            if (!(wj->lGroup == 128 && (K % 128) == 0 && P.lpr_log2 >= 2)) q4p_ok = false;
        }
        P.M[j] = wj->ne0, P.qBias[j] = wj->qBias;
        if (paired && j == 1) {
            if (wj->ne0 != w[0]->ne0) return false;
            continue;
        }
        P.slot0[j] = slots;
        slots += (wj->ne0 + RPS - 1) / RPS;
    }
    for (int j = P.njobs; j < 3; j++) P.slot0[j] = 0x7fffffff;
    P.total_slots = slots;
    P.spg = (slots + n_wg - 1) / n_wg;
    return true;
}

// caller's workspace (cached device memory): [ws words 256 B] [EngLayer table] [XCD-local area: tickets | lqkv | lpart] [room for the exchange vectors, used
// only when the uncached allocation is refused]
size_t engine_ws_bytes(const kf_engine_desc* d) {
    const int q_dim = d->n_head * d->head_dim, kv_dim = d->n_kv * d->head_dim, GQ = d->n_kv > 0 ? d->n_head / d->n_kv : 1;
    size_t b = 256 + (((size_t)d->n_layer * sizeof(EngLayer) + 255) & ~(size_t)255);
    b += (eng_loc_bytes(GQ, d->head_dim) + 255) & ~(size_t)255;
    b += (size_t)eng_xoff(d->dim, q_dim, kv_dim, d->ffn, d->head_dim).end * 4 + 256;
    return b;
}

static void engine_release(EngineHost* E) {
    if (!E) return;
    if (E->args.dbg) (void)hipFree(E->args.dbg);
    if (E->xmem) (void)hipFree(E->xmem);
    delete E;
}

// the state every launch starts from: all granules "not written" (tags of generation 0xffff.., which no epoch below 2^16 / n_layer reaches soon), epoch 1, no
// error, tickets zero
static int engine_init_state(EngineHost* E, hipStream_t st) {
    EngArgs& a = E->args;
    if (E->xmem && hipMemsetAsync(E->xmem, 0xff, E->xbytes, st) != hipSuccess) return KF_HIP_CHECK;
    char* const loc0 = a.loc;
    char* const end = reinterpret_cast<char*>(E->ws) + E->ws_bytes;
    if (hipMemsetAsync(loc0, 0xff, (size_t)(end - loc0), st) != hipSuccess) return KF_HIP_CHECK;
    static const int init[16] = {1, 0}; /* epoch 1, no error, statistics (words 8 .. 14) zero */
    if (hipMemsetAsync(a.loc, 0, 1024, st) != hipSuccess || hipMemcpyAsync(a.ws, init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) return KF_HIP_CHECK;
    return KF_OK;
}

// why != NULL: *why names the reason of a refusal; dry: validation only (kf_engine_served) -- nothing is allocated, ws may be NULL
int engine_build(const kf_engine_desc* d, void* ws, size_t ws_bytes, hipStream_t st, EngineHost** out, const char** why, bool dry) {
    const char* dummy;
    if (!why) why = &dummy;
    *why = "bad arguments";
    if (!d || (!dry && (!ws || !out)) || d->n_layer < 1 || !d->layers) return KF_INVALID_ARGS;
    if (!dry && (ws_bytes < engine_ws_bytes(d) || ((uintptr_t)ws & 255) != 0)) return KF_INVALID_ARGS;
    const int hd = d->head_dim;
    *why = "head_dim must be 64 or 128 and n_head a multiple of n_kv";
    if ((hd != 64 && hd != 128) || d->n_kv <= 0 || d->n_head % d->n_kv != 0) return KF_UNSUPPORTED_DATATYPE;
    const int GQ = d->n_head / d->n_kv;
    const int shape_class = engine_shape_class(GQ, hd, d->dim, d->n_head * hd, d->ffn);
    *why = "model shape not instantiated: the engine's phase plans are compile-time types; built for Qwen3-0.6B (dim 1024, 16/8 heads of 128, ffn 3072), Qwen3-1.7B (dim 2048, same heads, ffn 6144; 4-bit layers) and the 256-wide test shape (dim 256, 4/2 heads of 64, ffn 512)";
    if (!shape_class) return KF_UNSUPPORTED_DATATYPE; /* not one of the instantiated model shapes: the per-layer launches remain */
    int dev = 0, n_cu = 0;
    *why = "HIP failure";
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) return KF_HIP_CHECK;
    *why = "the device has fewer than 256 compute units: one resident workgroup per CU is the engine's premise";
    if (n_cu < ENG_NWG) return KF_UNSUPPORTED_DATATYPE; /* one resident workgroup per CU is the engine's premise */
    n_cu = ENG_NWG;
    EngineHost* E = new EngineHost();
    memset(E, 0, sizeof(*E));
    EngArgs& a = E->args;
    a.n_layer = d->n_layer;
    E->dim = d->dim, E->n_head = d->n_head, E->n_kv = d->n_kv, E->q_dim = d->n_head * hd, E->kv_dim = d->n_kv * hd, E->ffn = d->ffn;
    a.kv_stride = d->kv_stride;
    a.max_seq = d->max_seq;
    a.eps = d->rms_eps, a.qk_eps = d->qk_eps, a.rope_table = d->rope_table;
    *why = "rope_table missing, kv_stride not a multiple of 8, or max_seq < 1";
    if (!a.rope_table || (a.kv_stride % 8) != 0 || a.max_seq < 1) {
        delete E;
        return KF_INVALID_ARGS;
    }
    // phases: every layer must have the same shapes and storage
    const kf_engine_layer& L0 = d->layers[0];
    int fmt = gemv_fmt_of(&L0.w[0]);
    *why = "layer storage not served: the engine is instantiated for 4-bit (RTN), 2-bit and 1-bit PackedQ layers in groups of 128; other storages keep the per-layer launches";
    if (fmt != FMT_Q4 && fmt != FMT_Q1 && fmt != FMT_Q2) { /* the engine is instantiated for the 4-bit PackedQ storage (BASELINE config 2); other storages keep the per-layer launches */
        delete E;
        return KF_UNSUPPORTED_DATATYPE;
    }
    *why = "a layer's matrices differ from layer 0's in shape, storage, grouping or alignment, or a norm weight / cache pointer is missing";
    bool q4p_ok = fmt == FMT_Q4;
    std::vector<EngLayer> tab(d->n_layer);
    for (int l = 0; l < d->n_layer; l++) {
        const kf_engine_layer& L = d->layers[l];
        const kf_weight* k1[3] = {&L.w[0], &L.w[1], &L.w[2]};
        const kf_weight* k4[1] = {&L.w[3]};
        const kf_weight* k5[2] = {&L.w[4], &L.w[5]};
        const kf_weight* k6[1] = {&L.w[6]};
        EngPlan p1, p4, p5, p6;
        EngPlan* const pl = E->plans;
        bool ok = eng_plan(p1, fmt, E->dim, 3, k1, false, n_cu, q4p_ok) && eng_plan(p4, fmt, E->q_dim, 1, k4, false, n_cu, q4p_ok) &&
                  eng_plan(p5, fmt, E->dim, 2, k5, true, n_cu, q4p_ok) && eng_plan(p6, fmt, E->ffn, 1, k6, false, n_cu, q4p_ok);
        ok = ok && L.w[0].ne0 == E->q_dim && L.w[1].ne0 == E->kv_dim && L.w[2].ne0 == E->kv_dim && L.w[3].ne0 == E->dim && L.w[4].ne0 == E->ffn && L.w[6].ne0 == E->dim;
        ok = ok && L.norm_in && L.norm_post && L.kcache && L.vcache && (((uintptr_t)L.kcache | (uintptr_t)L.vcache) & 15) == 0;
        if (l == 0) pl[0] = p1, pl[1] = p4, pl[2] = p5, pl[3] = p6;
        if (ok && l > 0) ok = !memcmp(&p1, &pl[0], sizeof(p1)) && !memcmp(&p4, &pl[1], sizeof(p4)) && !memcmp(&p5, &pl[2], sizeof(p5)) && !memcmp(&p6, &pl[3], sizeof(p6));
        if (!ok) {
            delete E;
            return KF_UNSUPPORTED_DATATYPE;
        }
        for (int j = 0; j < 7; j++) {
            const kf_weight& w = L.w[j];
            tab[l].m[j].w = (g_u32x4)(uintptr_t)w.data;
            tab[l].m[j].zero = tab[l].m[j].step = nullptr;
            if (fmt >= FMT_Q4) {
                tab[l].m[j].zero = (g_u16)(uintptr_t)(w.gama + w.ne0 + w.ne1);
                tab[l].m[j].step = (g_u16)(uintptr_t)(w.gama + w.ne0 + w.ne1 + (size_t)w.ne0 * w.ne1 / w.lGroup);
            }
        }
        if (l == 0)
            for (int j = 0; j < 7; j++) a.qbias[j] = (float)L.w[j].qBias;
        else
            for (int j = 0; j < 7; j++)
                if (a.qbias[j] != (float)L.w[j].qBias) {
                    delete E;
                    return KF_UNSUPPORTED_DATATYPE;
                }
        tab[l].norm_in = (g_u16)(uintptr_t)L.norm_in, tab[l].norm_post = (g_u16)(uintptr_t)L.norm_post;
        tab[l].norm_q = (g_u16)(uintptr_t)L.q_norm, tab[l].norm_k = (g_u16)(uintptr_t)L.k_norm;
        tab[l].kcache = (g_u16w)(uintptr_t)L.kcache, tab[l].vcache = (g_u16w)(uintptr_t)L.vcache;
        tab[l].hot = (g_i32)(uintptr_t)L.hot_ffn;
    }
    if (fmt == FMT_Q4 && q4p_ok) fmt = FMT_Q4P;
    if (fmt == FMT_Q1) fmt = FMT_Q1T; /* the LDS selector-table forms (same bits as the per-bit select forms) */
    if (fmt == FMT_Q2) fmt = FMT_Q2T;
    if (shape_class == 3 && fmt != FMT_Q4P) { /* the 2048-wide shape is instantiated for the register-table 4-bit storage only */
        delete E;
        *why = "the Qwen3-1.7B shape is instantiated for 4-bit (RTN, groups of 128) layers only";
        return KF_UNSUPPORTED_DATATYPE;
    }
    if (dry) {
        delete E;
        *why = "";
        return KF_OK;
    }
    E->fmt = fmt, E->GQ = GQ, E->hd = hd, E->n_cu = n_cu, E->nwv = ENG_NWV, E->shape_class = shape_class;
    E->canon = 1;
    E->xmap = shape_class == 1 || shape_class == 3 ? 1 : 0; /* 8 kv-heads on 8 XCDs (informational: the mapping is a template parameter of the instantiation) */
    // workspace carve
    char* p = reinterpret_cast<char*>(ws);
    E->ws = ws, E->ws_bytes = ws_bytes;
    a.ws = reinterpret_cast<int*>(p), p += 256;
    a.layers = reinterpret_cast<const EngLayer*>(p), p += ((size_t)d->n_layer * sizeof(EngLayer) + 255) & ~(size_t)255;
    a.loc = p, p += (eng_loc_bytes(GQ, hd) + 255) & ~(size_t)255;
    {
        /* s_sleep units (one trip of the wait loop ~ 40 ns) behind the own publish: x, q|k|v, slice partials, ao, xB, act.  Tuned on the 0.6B shape at 2 k keys
           (scratch/eng_ab.py): after the drains were removed 12,8,12,12,12,12 0.455 ms/step, 16,8,12,16,16,16 0.452, 20,12,16,20,20,20 0.464, 24,12,16,24,24,24 0.474 */
        const int dflt[6] = {16, 8, 20, 24, 16, 16};
        for (int i = 0; i < 6; i++) a.delay[i] = dflt[i];
        for (int n = 0; n <= KF_ATTN_MAX_SPLITS; n++)
            for (int i = 0; i < 6; i++) E->delay_tab[n][i] = dflt[i];
    }
    // The vectors that cross XCDs live in uncached device memory: an sc1 sweep of a cached (hipMalloc) line costs 75 ns per KB and CU, of an uncached one 43
    // (scratch/ub_handoff3.hip).  The XCD-local vectors (lqkv, lpart) stay in the caller's cached workspace: they are meant to live in that XCD's L2.
    E->xbytes = (size_t)eng_xoff(E->dim, E->q_dim, E->kv_dim, E->ffn, hd).end * 4;
    E->xmem = nullptr;
    if (hipExtMallocWithFlags(&E->xmem, E->xbytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        E->xmem = nullptr; /* the cached workspace serves (slower sweeps, same protocol) */
    }
    a.xch = reinterpret_cast<uint32_t*>(E->xmem ? E->xmem : (void*)p);
    if (engine_init_state(E, st) != KF_OK ||
        hipMemcpyAsync(const_cast<EngLayer*>(a.layers), tab.data(), tab.size() * sizeof(EngLayer), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        engine_release(E);
        *why = "HIP failure while initialising the workspace";
        return KF_HIP_CHECK;
    }
    // LDS
    int maxK = E->dim > E->q_dim ? E->dim : E->q_dim;
    if (E->ffn > maxK) maxK = E->ffn;
    const size_t xs_bytes = ((size_t)maxK * 4 + 15) & ~(size_t)15;
    size_t smem = (((size_t)d->n_layer * sizeof(EngLayer) + 15) & ~(size_t)15) + 2 * xs_bytes + 2 * (((size_t)E->dim * 2 + 15) & ~(size_t)15);
    smem += sizeof(uint16_t) * ((size_t)2 * GQ * hd + 3 * hd) + sizeof(double) * ((size_t)ENG_NWV * GQ * (hd + 2) + 64 * KF_ATTN_MAX_SPLITS + 64) + 4 * 16 + 4 * 64 + 32;
    smem += (fmt == FMT_Q1T ? 4096 : (fmt == FMT_Q2T ? 2048 : 0)) + (size_t)d->n_layer * 4 + 64; /* the 1-bit / 2-bit selector table; the layers' hot bits */
    smem = (smem + 15) & ~(size_t)15;
    if (smem > 160 * 1024) {
        engine_release(E);
        *why = "the layer table and the staging buffers exceed 160 KB of LDS (too many layers)";
        return KF_UNSUPPORTED_DATATYPE;
    }
    E->smem = smem;
    *out = E;
    *why = "";
    return KF_OK;
}
void engine_free(EngineHost* E) { engine_release(E); }
// diagnostic runs: per-phase wall-clock stamps of one workgroup (the DBG instantiation of the kernel)
int engine_debug_enable(EngineHost* E, int wg) {
    EngArgs& a = E->args;
    if (!a.dbg) {
        if (hipMalloc(&a.dbg, (size_t)a.n_layer * 2 * 16 * 8) != hipSuccess) {
            a.dbg = nullptr;
            return KF_HIP_CHECK;
        }
    }
    (void)hipMemset(a.dbg, 0, (size_t)a.n_layer * 2 * 16 * 8);
    a.dbg_wg = wg;
    return KF_OK;
}
void engine_set_delays(EngineHost* E, const int* d6) { /* every slice count */
    for (int n = 0; n <= KF_ATTN_MAX_SPLITS; n++)
        for (int i = 0; i < 6; i++) E->delay_tab[n][i] = d6[i];
}
int engine_debug_read(EngineHost* E, unsigned long long* h_out, int n_words) {
    if (!E->args.dbg) return 0;
    const int have = E->args.n_layer * 2 * 16;
    const int n = n_words < have ? n_words : have;
    if (hipMemcpy(h_out, E->args.dbg, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}

template <class C>
static bool engine_plans_match(const EngineHost* E) {
    using SH = typename C::SH;
    const CPlan c[4] = {SH::P1::plan(), SH::P4::plan(), SH::P5::plan(), SH::P6::plan()};
    for (int i = 0; i < 4; i++) {
        const EngPlan& r = E->plans[i];
        if (r.K != c[i].K || r.nBlk != c[i].nBlk || r.lpr_log2 != c[i].lpr_log2 || r.iters != c[i].iters || r.total_slots != c[i].total || r.spg != c[i].spg ||
            r.njobs != c[i].njobs)
            return false;
        for (int j = 0; j < r.njobs; j++)
            if (r.slot0[j] != c[i].slot0[j] || r.M[j] != c[i].M[j]) return false;
    }
    return true;
}
template <class C>
static int engine_go(EngineHost* E, hipStream_t st) {
    static int ready = 0; /* 1 ok, -1 the compile-time geometry is not the mat-vec launcher's */
    if (!ready) {
        if (!engine_plans_match<C>(E))
            ready = -1;
        else if (hipFuncSetAttribute((const void*)engine_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return KF_HIP_CHECK;
        else
            ready = 1;
    }
    if (ready < 0) return 1;
    hipLaunchKernelGGL((engine_kernel<C>), dim3(C::NWG), dim3(C::NWV * 64), E->smem, st, E->args);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}
template <int FMT>
static int engine_go_fmt(EngineHost* E, hipStream_t st) {
    switch (E->shape_class) {
        case 1: /* 8 kv-heads on 8 XCDs: the attention chain of a kv-head stays inside one XCD */
            if constexpr (FMT == FMT_Q4P) { /* the diagnostic (stamped) instantiations exist for the benchmark's storage only */
                if (E->args.dbg && E->canon) return engine_go<EngCfg<FMT_Q4P, 2, 128, ENG_NWV, 1024, 2048, 1024, 3072, ENG_NWG, true, true, true>>(E, st);
                if (E->args.dbg) return engine_go<EngCfg<FMT_Q4P, 2, 128, ENG_NWV, 1024, 2048, 1024, 3072, ENG_NWG, true, true, false>>(E, st);
            }
            if (!E->canon) return engine_go<EngCfg<FMT, 2, 128, ENG_NWV, 1024, 2048, 1024, 3072, ENG_NWG, true, false, false>>(E, st);
            return engine_go<EngCfg<FMT, 2, 128, ENG_NWV, 1024, 2048, 1024, 3072, ENG_NWG, true, false, true>>(E, st);
        case 2:
            if (!E->canon) return engine_go<EngCfg<FMT, 2, 64, ENG_NWV, 256, 256, 128, 512, ENG_NWG, false, false, false>>(E, st);
            return engine_go<EngCfg<FMT, 2, 64, ENG_NWV, 256, 256, 128, 512, ENG_NWG, false, false, true>>(E, st);
        case 3: /* gate | up and down_proj hold 8 and 6 blocks per lane: part of them is dequantised behind the hand-off (MvPhase::AH) */
            if constexpr (FMT == FMT_Q4P) {
                if (!E->canon) return engine_go<EngCfg<FMT_Q4P, 2, 128, ENG_NWV, 2048, 2048, 1024, 6144, ENG_NWG, true, false, false>>(E, st);
                return engine_go<EngCfg<FMT_Q4P, 2, 128, ENG_NWV, 2048, 2048, 1024, 6144, ENG_NWG, true, false, true>>(E, st);
            }
            return 1;
        default: return 1;
    }
}

// 1: this position bound is outside what the engine serves (the caller runs the multi-launch path), < 0 error
int engine_step(EngineHost* E, hipStream_t st, const uint16_t* x_in, uint16_t* x_out, const int32_t* d_state, int pos_bound, int with_head, int n_steps) {
    EngArgs& a = E->args;
    if ((!x_in && !a.emb) || !x_out || !d_state || pos_bound < 0 || pos_bound >= a.max_seq) return KF_INVALID_ARGS;
    if (with_head && !a.head_w) return KF_INVALID_ARGS;
    if (n_steps < 1 || (n_steps > 1 && (with_head != 2 || x_in))) return KF_INVALID_ARGS; /* several steps per launch: only with the pick inside and the embedding row read inside */
    a.n_steps = n_steps;
    a.head_on = with_head ? 1 : 0;
    a.d_state_w = with_head == 2 ? const_cast<int32_t*>(d_state) : nullptr; /* 2: head + greedy pick + state update; 1: logits only */
    const int nsp = attn_splits(pos_bound, E->n_kv);
    const int chunk = (pos_bound + 1 + nsp - 1) / nsp;
    const int NW = (E->GQ <= 2 && chunk > 128) ? 8 : 4;
    if (NW != 4 || E->n_kv * nsp > E->n_cu) return 1; /* the 8-wave slice form and more slices than workgroups are not restated here */
    a.nsp = nsp, a.chunk = chunk;
    for (int i = 0; i < 6; i++) a.delay[i] = E->delay_tab[nsp][i];
    int e = (E->n_head * E->hd + E->n_cu - 1) / E->n_cu, me = 1;
    while (me < e) me <<= 1;
    if (me > E->hd || me > 64) return 1;
    a.merge_e = me; /* = EngCfg::ME of the instantiated shapes (whole workgroups: q_dim / n_cu) */
    a.x_in = x_in, a.x_out = x_out, a.d_state = d_state;
    switch (E->fmt) {
        case FMT_Q4P: return engine_go_fmt<FMT_Q4P>(E, st);
        case FMT_Q4: return engine_go_fmt<FMT_Q4>(E, st);
        case FMT_Q1T: return engine_go_fmt<FMT_Q1T>(E, st);
        case FMT_Q2T: return engine_go_fmt<FMT_Q2T>(E, st);
        default: return 1;
    }
}
// bf16 embedding table for steps given x_in == NULL (the row is read inside the launch: one launch less per token)
int engine_set_embedding(EngineHost* E, const kf_weight* w, const int32_t* d_forced) {
    if (!w) {
        E->args.emb = nullptr, E->args.d_forced = nullptr, E->args.emb_rows = 0;
        return KF_OK;
    }
    if (w->type != KF_BF16 || w->quant != KF_QUANT_GROUP || w->qzeros || w->ne1 != E->dim || !w->data) return KF_UNSUPPORTED_DATATYPE;
    E->args.emb = reinterpret_cast<const uint16_t*>(w->data), E->args.d_forced = d_forced, E->args.emb_rows = w->ne0;
    return KF_OK;
}
// The LM head (bf16 [vocab, dim]) + final norm as trailing phases of the launch (engine_step with_head); NULL removes it.
int engine_set_head(EngineHost* E, const kf_weight* w, const uint16_t* norm_w, uint16_t* logits, int32_t* d_tokens_out) {
    EngArgs& a = E->args;
    if (!w) {
        a.head_w = nullptr, a.head_norm = nullptr, a.logits = nullptr, a.d_tokens_out = nullptr, a.vocab = 0;
        return KF_OK;
    }
    if (w->type != KF_BF16 || w->quant != KF_QUANT_GROUP || w->qzeros || w->ne1 != E->dim || !w->data || ((uintptr_t)w->data & 15) != 0 || !norm_w || !logits || w->ne0 < 64)
        return KF_UNSUPPORTED_DATATYPE;
    // the compile-time geometry of the head phases must be the mat-vec launcher's for this matrix (same lanes per row: same summation order)
    const int nBlk = E->dim / 8;
    if (gemv_lpr_log2(nBlk, w->ne0) != c_lpr_log2(nBlk, 1L << 20)) return KF_UNSUPPORTED_DATATYPE;
    a.head_w = (g_u32x4)(uintptr_t)w->data, a.head_norm = (g_u16)(uintptr_t)norm_w, a.logits = logits, a.d_tokens_out = d_tokens_out, a.vocab = w->ne0;
    return KF_OK;
}
int engine_error_word(EngineHost* E, hipStream_t st, int* h_err) {
    int v[2] = {0, 0};
    if (hipMemcpyAsync(v, E->args.ws, sizeof(v), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return KF_HIP_CHECK;
    *h_err = v[1];
    return KF_OK;
}
void engine_set_canonical(EngineHost* E, int on) { E->canon = on ? 1 : 0; }

// ---- self-calibration of the first-sweep delays.  The hand-off protocol is correct for ANY delay -- a sweep that comes too early is repeated -- but a
// repeated sweep costs its whole round trip on the critical path and a late one its lateness, and the best values depend on the device (clocks, fabric) and on how many
// slices the context is cut into.  engine_tune times the layers-only launch at the position the decode state holds (it reads the state's token and writes that position's
// K / V rows, as the real step is about to: idempotent, nothing advances) and walks the six delays by coordinate descent, smallest mean launch time wins.
static float eng_time_launches(EngineHost* E, hipStream_t st, uint16_t* x_out, const int32_t* d_state, int pos_bound, int reps, hipEvent_t e0, hipEvent_t e1) {
    if (hipEventRecord(e0, st) != hipSuccess) return -1.f;
    for (int i = 0; i < reps; i++)
        if (engine_step(E, st, nullptr, x_out, d_state, pos_bound, 0, 1) != KF_OK) return -1.f;
    float ms = 0.f;
    if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
    return ms * 1e3f / (float)reps;
}
int engine_tune(EngineHost* E, hipStream_t st, uint16_t* x_out, const int32_t* d_state, int pos_bound, int passes, float* us_before, float* us_after) {
    if (!E->args.emb || !x_out || !d_state || pos_bound < 0 || pos_bound >= E->args.max_seq) return KF_INVALID_ARGS;
    const int nsp = attn_splits(pos_bound, E->n_kv);
    if (nsp < 1 || nsp > KF_ATTN_MAX_SPLITS) return KF_INVALID_ARGS;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return KF_HIP_CHECK;
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return KF_HIP_CHECK;
    }
    int* d = E->delay_tab[nsp];
    int saved[6];
    for (int i = 0; i < 6; i++) saved[i] = d[i];
    constexpr int REPS = 6;
    auto measure = [&]() { /* the smaller of two batches: the first launches after a change of rhythm run slow */
        const float a0 = eng_time_launches(E, st, x_out, d_state, pos_bound, REPS, e0, e1), a1 = eng_time_launches(E, st, x_out, d_state, pos_bound, REPS, e0, e1);
        return (a0 < 0.f || a1 < 0.f) ? -1.f : (a0 < a1 ? a0 : a1);
    };
    int rc = KF_OK;
    float best = measure();
    if (us_before) *us_before = best;
    if (best < 0.f) rc = 1; /* not served at this position (or a HIP failure): nothing to tune */
    static const int kStep[3] = {8, 4, 2};
    for (int pass = 0; rc == KF_OK && pass < passes && pass < 3; pass++) {
        for (int i = 0; rc == KF_OK && i < 6; i++) {
            const int cur = d[i];
            int best_v = cur;
            for (int s = -2; s <= 2; s++) {
                if (s == 0) continue;
                const int v = cur + s * kStep[pass];
                if (v < 0 || v > 96) continue;
                d[i] = v;
                const float t = measure();
                if (t < 0.f) {
                    rc = KF_HIP_CHECK;
                    break;
                }
                if (t < best * 0.998f) best = t, best_v = v; /* keep the old value unless the gain is above the timing noise */
            }
            d[i] = best_v;
        }
    }
    int err = 0;
    if (engine_error_word(E, st, &err) != KF_OK || err != 0 || rc != KF_OK) { /* a timed-out poll while tuning: the defaults stay, the caller sees the error word through engine_check */
        for (int i = 0; i < 6; i++) d[i] = saved[i];
        if (rc == KF_OK) rc = KF_INTERNAL_ERR;
    } else {
        E->tuned[nsp] = 1;
    }
    if (us_after) *us_after = best;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
// statistics since creation / the last reset: out[0..5] = sweeps the poller of workgroup 0 issued per hand-off (x, q|k|v, slice partials, ao, xB, act), out[6] = polls per
// hand-off (layers stepped), out[7..12] = the delays in use at `pos_bound`, out[13] = 1 when engine_tune measured them
int engine_stats(EngineHost* E, hipStream_t st, int pos_bound, int* out14) {
    int w[16];
    if (hipMemcpyAsync(w, E->args.ws, sizeof(w), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return KF_HIP_CHECK;
    for (int i = 0; i < 7; i++) out14[i] = w[8 + i];
    int nsp = attn_splits(pos_bound < 0 ? 0 : pos_bound, E->n_kv);
    if (nsp < 1 || nsp > KF_ATTN_MAX_SPLITS) nsp = 1;
    for (int i = 0; i < 6; i++) out14[7 + i] = E->delay_tab[nsp][i];
    out14[13] = E->tuned[nsp];
    return KF_OK;
}
// after a timed-out poll (the error word latches and every later launch returns at once): all granules back to "not written", epoch 1, error word cleared
int engine_reset(EngineHost* E, hipStream_t st) {
    const int rc = engine_init_state(E, st);
    if (rc != KF_OK) return rc;
    return hipStreamSynchronize(st) == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
