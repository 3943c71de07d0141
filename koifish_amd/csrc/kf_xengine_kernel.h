// kf_xengine_kernel.h (the device code of kf_xengine.hip; round 6: a header, so that the storages' instantiations compile as separate translation units) -- EIGHT independent decoders per GPU, one per XCD: the decode step of kf_engine.hip re-shaped so that the CHIP streams bytes while each SEQUENCE
// waits on its hand-offs (round 5).
//
// kf_engine.hip gives one sequence all 256 CUs: a layer's 8.4 MB of weights sit in registers before the activations arrive, and the step is bound by its six all-to-all
// hand-offs per layer across 8 XCDs (0.22 of the HBM roofline, measured to be the floor of that design: DESIGN.md section 8.1).  The reference decodes one sequence per
// process (GoPT.cpp:1139-1180) and SURVEY section 8e scales the 0.6B model as independent replicas; this file puts the replicas INSIDE the package:
//
//   * 256 workgroups, one per CU.  A workgroup reads its XCD from HW_REG_XCC_ID and takes a ticket there: the 32 workgroups of XCD s are the decoder of sequence s
//     (placement-independent: nothing assumes which CU a block lands on, only that all 256 are resident).  The sequences share the weights and nothing else: own
//     K/V cache, own decode state, own forced ids, own logits.
//   * every hand-off vector of a decoder lives in cached memory that only its XCD touches: producers write tagged granules with PLAIN stores (they land in that XCD's
//     L2), the poller wave sweeps them with sc1 loads -- "the data is the flag" exactly as in kf_engine.hip, but an edge costs ~0.7 us instead of 1.2-2.3.
//   * 32 CUs cannot hold a layer in registers (261 KB of blocks per workgroup and layer), so the mat-vec phases STREAM: each compute wave walks its row slots with a ring
//     of DEPTH 16-byte blocks in flight, and in the last round of a phase the ring is refilled with the first blocks of the NEXT phase -- they do not depend on the
//     hand-off, so HBM latency stays off the chain.  The attention phase streams its K/V rows the same way (two batches of tiles in flight).
//
// The arithmetic is the canonical order of oracle/kf_oracle.c sections 4c and 6 (the library default): the same lanes per row, chain pairs and lane tree as
// gemv_kernel<.., CANON> (the geometry is a property of the matrix shape: PlanT, kf_engine_common.h), the order-free fp64 softmax sums of kf_attn_common.h.  Every id,
// logit and K/V row of every sequence equals what kf_engine.hip, the per-layer launches and the oracle produce for that sequence alone.
//
// Forms of the one kernel (XCfg): one decoder per XCD (12 waves, 168 registers) or two (two workgroups of 8 waves per CU, 128 registers: sequences x and x + 8); the model
// shapes of Qwen3-0.6B / 1.7B / 4B / 8B (GQA-4: four query heads per key tile, q | k | v as one fused matrix, wide vectors staged by all waves); and TP (XCfg::TP): ONE sequence
// whose eight tensor-parallel ranks are the eight XCDs -- o_proj / down_proj rows leave as fp32 partials into every rank's receive area, are summed in rank order by the
// workgroup that owns the rows and continue as a local hand-off (xe_publish push / xe_tp_reduce), the head in vocabulary shards with a pick across the XCDs.
//
// Replaces, for eight sequences at once: Fish::ForwardOnRLS (gLLM.cpp:722-787) over TokenEmbed::cuInfer (NeuronFuse.cu:176-218), SelfAttention::cuInfer (QKV.cu:617-702),
// FFN::cuInfer (NeuronFuse.cu:615-656), Head4Token::cuInfer_1 (NeuronFuse.cu:842-862) and sample_argmax (GoPT.cpp:602-612).
#include <stdlib.h>
#include <string.h>

#include <vector>

#pragma once
#include "kf_engine_common.h"

namespace kf {

constexpr int XE_NWG = 32;   /* workgroups of one decoder = the CUs of one XCD */
constexpr int XE_NXCD = 8;
constexpr int XE_GRID = XE_NWG * XE_NXCD; /* workgroups of a launch with ONE decoder per XCD; two decoders per XCD (WPC = 2): twice that, two workgroups per CU */
constexpr int XE_MAXSEQ = 32; /* 8 XCDs x up to 4 sequences per decoder (XCfg::NB) or 2 decoders (XCfg::WPC) */

struct XArgs {
    const EngLayer* layers; /* kcache / vcache = sequence 0's; sequence s at + s * kv_seq_stride elements */
    int n_layer, n_steps, n_seq;
    float eps, qk_eps;
    const float* rope_table;
    const uint16_t* emb; /* bf16 [emb_rows, DIM] */
    int emb_rows;
    int32_t* d_state;          /* [n_seq][4]: {token, pos, -, -} */
    const int32_t* d_forced;   /* [n_seq][forced_stride] or NULL */
    int32_t* d_tokens_out;     /* [n_seq][tokens_stride] or NULL */
    int forced_stride, tokens_stride;
    uint16_t* x_out;           /* [n_seq][DIM]: the residual stream after the last layer of the last step */
    uint16_t* logits;          /* [n_seq][vocab] */
    long long kv_seq_stride;
    int kv_stride, max_seq;
    float qbias[7];
    g_u32x4 head_w;
    g_u16 head_norm;
    int vocab, pick; /* pick: the greedy pick and the state update run inside (needed for n_steps > 1) */
    char* loc;       /* XCD-local exchange areas, loc_stride bytes each */
    size_t loc_stride;
    int* ws;         /* [0] (unused since round 6: the epoch is XArgs::epoch0), [1] error word, [16 + 32 x] ticket of XCD x, [17 + 32 x] its workgroups that have left */
    unsigned long long* dbg; /* diagnostic instantiation: [step][layer][16] stamps of (sequence dbg_seq, workgroup rank dbg_wg) */
    int dbg_seq, dbg_wg, dbg_steps;
    int deal_wl;    /* weight (x 8) of the compute waves that share the poller's SIMD: see xe_deal */
    // tensor parallel over the XCDs (XCfg::TP): ONE sequence, XCD r = rank r.  `layers` = [rank][n_layer] (each rank's shards and its kv-head's cache rows); the head in
    // vocabulary shards; head_w / logits / vocab above are unused
    g_u32x4 head_w_r[XE_NXCD];
    uint16_t* logits_r[XE_NXCD];
    int vocab_r[XE_NXCD], row0_r[XE_NXCD];
    unsigned long long* tp_recv; /* [rank][2][source rank][DIM] granules {fp32 partial | generation}: o_proj exchanges in buffer 0, down_proj exchanges in buffer 1 */
    unsigned long long* tp_best; /* [rank][source rank] {global row | tag16, bf16 value} */
    int stagger_us; /* two decoders per XCD: microseconds the second one starts behind the first */
    int epoch0;     /* the generation of the launch's first step (the host counts: + n_steps per launch; 1 after a reset) */
};

// error word bits: 1 a hand-off vector (x, ao, xB, act, head x), 2 q|k|v, 4 partials, 8 workgroups per XCD != 32, 16 pick, 32 token granule, 64 position beyond the cache, 2048 a TP exchange

constexpr int xe_p1_wgs(int dim, int epb, int qd, int kvd) { /* the most workgroups (32 ... 8) whose equal pieces of the q | k | v row slots do not straddle a matrix */
    for (int w = 32; w >= 8; w--) {
        const CPlan p = c_plan(dim, epb, qd, kvd, kvd, false, w);
        if (p.total == w * p.spg && p.slot0[1] % p.spg == 0 && p.slot0[2] % p.spg == 0) return w;
    }
    return 0;
}
template <int FMT_, int GQ_, int HD_, int NWV_, int DIM_, int QD_, int KVD_, int FFN_, int DEPTH_, bool DBG_, int WPC_ = 1, int AU_ = 2, bool TP_ = false, int NB_ = 1, int NP_ = 0>
struct XCfg {
    // NB: sequences per decoder (round 6).  The decoders of a launch unpack the SAME 4-bit blocks; with NB > 1 a decoder multiplies every unpacked block against the
    // activations of NB sequences (xcc + 8 b, b = 0 .. NB - 1) staged side by side in LDS -- the exact bf16-stepwise unpack (7 vector instructions per weight with its
    // bookkeeping) is paid once, each sequence adds its half-instruction per weight (v_pk_fma_f32) and keeps its own canonical chain, lane tree and rounding: every id, logit
    // and K / V row of every sequence stays bit for bit what the sequence alone produces.  Hand-offs carry NB vectors (one wait, NB sweeps), the attention, the K / V caches
    // and the decode state stay per sequence, the head's bf16 rows are read once for the NB sequences.
    static constexpr int NB = NB_;
    // TP: the eight XCDs are the eight ranks of ONE sequence (QD_, KVD_, FFN_: a rank's shard widths).  o_proj / down_proj are column shards: their rows leave as fp32
    // partials into every rank's receive area (the protocol of kf_tp.hip, inside the launch), each workgroup sums its 1 / 32 of the rows over the ranks in rank order,
    // adds the residual and publishes the slice inside its XCD -- from there on the hand-off is the local one.
    static constexpr bool TP = TP_;
    // The decoders of a launch stream the SAME head and the SAME layer weights (only a TP rank's shards are its own).  The head rows are read with PLAIN loads: with the
    // non-temporal hint (a stream read once) every decoder's 311 MB came from HBM; plain, the lines stay in the 256 MB memory-side cache for the other XCDs' decoders --
    // 16 sequences 3.46 -> 3.37 ms per step, 8 sequences 1.87 -> 1.83.  The layer weights: plain for one decoder per XCD (1.83 -> 1.82), non-temporal for two (plain: 3.43 --
    // two decoders' 8.4 MB layers and their K / V rows already fight for the XCD's 4 MB L2).
    static constexpr int WAUX = (TP_ || WPC_ > 1) ? 2 /* nt */ : 0;
    static constexpr bool KV_NT = GQ_ <= 4; /* a sequence's K / V rows, read once per step: non-temporal (head groups read them twice: plain) */
    // wide residual streams: EVERY wave of the workgroup sweeps, normalises and stages its own 1 KiB units of x / xB (the compute waves stand at the barrier behind that
    // staging anyway; ONE wave doing it took 14 us of a 192 us layer at 5120 values: scratch/xtp_time.py)
    static constexpr bool COOP = DIM_ >= 2048;
    static constexpr int FFNP = (FFN_ + 255) & ~255; /* the SwiGLU vector's exchange area in whole 1 KiB sweeps (a rank's 3200: 128 granules of padding, published as zeros) */
    static constexpr int AU = AU_; /* key tiles per attention batch and wave */
    static constexpr bool DEAL_CONTIG = WPC_ > 1 || NWV_ == 8; /* xe_deal: 7 compute waves = ONE beside the poller on its SIMD: a run per wave, that wave's shorter */
    static constexpr int WPC = WPC_; /* decoders per XCD = workgroups per CU: 2 lets one decoder's hand-off waits run under the other's arithmetic (the hardware interleaves the two workgroups' waves) */
    // NB pollers (waves NCW .. NWV - 1: wave NCW + b runs the hand-offs of sequence b, side by side) and NWV - NB compute waves.  (One poller staging four sequences' vectors one
    // after the other took 43 of a 169 us layer period; hand-offs run by compute waves beside their rows cost the attention loop its registers: 23 -> 32 us per layer.)
    // NP pollers for the NB sequences (NP_ = 0: one each): poller p runs the hand-offs of sequences p, p + NP, ... one after the other.  Two pollers for four sequences leave
    // ten of twelve waves to the arithmetic (the mat-vec phases and the attention are 86 % of the NB = 4 layer period; the hand-offs 12 %)
    static constexpr int NP = NP_ > 0 ? NP_ : NB_;
    static_assert(NB_ % NP == 0, "every poller the same number of sequences");
    static constexpr int FMT = FMT_, GQ = GQ_, HD = HD_, NWV = NWV_, NCW = NWV_ - NP, DIM = DIM_, QD = QD_, KVD = KVD_, FFN = FFN_, DEPTH = DEPTH_, NWG = XE_NWG;
    static constexpr bool DBG = DBG_;
    static_assert(FMT_ == FMT_Q4 || FMT_ == FMT_Q4P || FMT_ == FMT_Q1T || FMT_ == FMT_Q2T,
                  "4-bit PackedQ layers (arithmetic or register-table unpack), or -- round 6 -- 1-bit / 2-bit: one dword of a 128-element block, resp. one 8-byte half of a 64-element block per lane");
    static constexpr int VBYTES = FMT_ == FMT_Q1T ? 4 : (FMT_ == FMT_Q2T ? 8 : 16); /* bytes of a lane's 32-weight piece of the packed stream */
    static constexpr bool PREP = FMT_ == FMT_Q1T || FMT_ == FMT_Q2T;                  /* unpacked through BlockPrep<FMT> and the LDS selector table */
    static constexpr int n_head = QD_ / HD_, n_kv = KVD_ / HD_;
    // more than four query heads per kv-head (a TP rank of Qwen3-32B: 8 on 1): NG groups of GQW heads, each group its own workgroups over the same key slices (the fp64 sums
    // of eight heads would be 144 registers per lane) -- K / V rows are then read NG times, from this XCD's L2
    static constexpr int NG = GQ_ > 4 ? GQ_ / 4 : 1, GQW = GQ_ / NG;
    static_assert(GQ_ % NG == 0 && XE_NWG % (n_kv * NG) == 0, "whole workgroups per kv-head and head group");
    static constexpr int SPK = XE_NWG / (n_kv * NG); /* key slices (= workgroups) per kv-head and head group */
    // q | k | v rows: a workgroup's rows belong to ONE of the three matrices.  With 16 / 8 heads the 32 workgroups cut the row slots that way; for the GQA-4 shapes (32 / 8 heads:
    // 4 + 1 + 1 parts) only 24 equal pieces would (P1W0) -- the other eight workgroups would own no row of the phase: see FUSED below
    static constexpr int P1W0 = xe_p1_wgs(DIM_, eng_vepb<FMT_>(), QD_, KVD_);
    static_assert(P1W0 > 0, "no cut of the q | k | v row slots into whole-matrix pieces");
    // FUSED: where 32 equal pieces would straddle the matrices, the engine multiplies ONE matrix of QD + 2 KVD rows -- a copy of the three shards' blocks and zero / step words,
    // q rows then k rows then v rows, built once at create time inside the workspace (same rows, same lanes per row: the launch's rows were QD + 2 KVD all along) -- and all 32
    // workgroups own rows of the phase.  (The 8-on-1 test shape's 80 row slots do not divide by 32: it keeps the 20-workgroup cut.)
    static constexpr bool FUSED = P1W0 < XE_NWG && c_plan(DIM_, eng_vepb<FMT_>(), QD_ + 2 * KVD_, 0, 0, false, XE_NWG).total % XE_NWG == 0;
    static constexpr int P1W = FUSED ? XE_NWG : P1W0;
    struct SHF {
        using B = EngShape<FMT_, DIM_, QD_, KVD_, FFN_, XE_NWG>;
        using P1 = PlanT<DIM_, eng_vepb<FMT_>(), QD_ + 2 * KVD_, 0, 0, false, XE_NWG>;
        using P4 = typename B::P4;
        using P5 = typename B::P5;
        using P6 = typename B::P6;
    };
    using SH = std::conditional_t<FUSED, SHF, EngShape<FMT_, DIM_, QD_, KVD_, FFN_, XE_NWG, P1W0>>;
    static_assert(!FUSED || !PREP, "the fused q | k | v copy is laid out for 16-byte blocks");
    static_assert(!FUSED || SH::P1::lpr_log2 == EngShape<FMT_, DIM_, QD_, KVD_, FFN_, XE_NWG, P1W0>::P1::lpr_log2, "the fused matrix is walked with the launch's lanes per row");
    static constexpr int ME = QD_ / XE_NWG; /* ao elements a workgroup merges */
    static_assert(QD_ % XE_NWG == 0 && ME % 4 == 0 && ME <= HD_ && HD_ % ME == 0 && ME <= 128, "merge elements per workgroup");
    static constexpr int PSH = SPK * (2 * HD_ + 4); /* 8-byte granules of one head's slice partials: [HD / ME][SPK][ME] values x 2, then [SPK][4] {m, L lo, L hi, -} */
    // the XCD-local exchange area (dwords)
    static constexpr int xA = 0, qkv = xA + eng_gran_dw(DIM_), ao = qkv + eng_gran_dw(QD_ + 2 * KVD_), xB = ao + eng_gran_dw(QD_), act = xB + eng_gran_dw(DIM_),
                         part = act + eng_gran_dw(FFNP), hbest = part + eng_gran_dw(2 * n_head * PSH), tokg = hbest + eng_gran_dw(2 * XE_NWG), loc_dw = tokg + eng_gran_dw(2);
    // LM head (bf16 [vocab, DIM]): the geometry gemv_launch picks for a many-row bf16 matrix of this width
    static constexpr int HnBlk = DIM_ / 8, Hlpr_log2 = c_lpr_log2(DIM_ / 8, 1L << 20), HLPR = 1 << Hlpr_log2, HRPS = 64 >> Hlpr_log2, Hiters = (HnBlk + HLPR - 1) / HLPR;
    static constexpr int XCH = 8; /* fp32 activations: 16-byte chunks per 32-weight block */
    static constexpr int maxKc = DIM_ > QD_ ? (DIM_ > FFN_ ? DIM_ : FFN_) : (QD_ > FFN_ ? QD_ : FFN_);
    // the waves' attention sums (fp64, [NCW][GQ][hd + 2]) in the second activation buffer when the two would not fit side by side (Qwen3-8B: 2 x 48 KB of activations): the
    // buffer is idle between the barrier in front of q | k | v (down_proj of the layer before has read it) and the staging of the attention output, which waits for every
    // slice partial of the XCD -- so for this workgroup's, written after the last read of the sums
    static constexpr bool COMB_IN_XS1 = (size_t)maxKc * 8 + (size_t)DIM_ * 4 + sizeof(double) * (size_t)(NWV_ - 1) * GQW * (HD_ + 2) > (size_t)(WPC_ > 1 ? 70 : 140) * 1024; /* (two workgroups per CU: half the LDS each -- the 1.7B shape) */
    static_assert(!COMB_IN_XS1 || sizeof(double) * (size_t)(NWV_ - 1) * GQW * (HD_ + 2) <= (size_t)maxKc * 4, "the sums fit the buffer");
    static constexpr int XS = maxKc / 32; /* chunk stride of the staged activations (16-byte units): chunk j of block column c at [j * XS + c], whatever the phase's width */
    // every phase in whole rows and whole iterations: no masks at the multiply
    static constexpr bool EXACT = (QD_ + 2 * KVD_) % SH::P1::RPS == 0 && KVD_ % SH::P1::RPS == 0 && QD_ % SH::P1::RPS == 0 && DIM_ % SH::P4::RPS == 0 && FFN_ % SH::P5::RPS == 0 && DIM_ % SH::P6::RPS == 0 &&
                                  SH::P1::nBlk == SH::P1::iters * SH::P1::LPR && SH::P4::nBlk == SH::P4::iters * SH::P4::LPR && SH::P5::nBlk == SH::P5::iters * SH::P5::LPR &&
                                  SH::P6::nBlk == SH::P6::iters * SH::P6::LPR;
    static constexpr int maxR = (SH::P1::R > SH::P5::R ? SH::P1::R : SH::P5::R) > (SH::P4::R > SH::P6::R ? SH::P4::R : SH::P6::R) ? (SH::P1::R > SH::P5::R ? SH::P1::R : SH::P5::R)
                                                                                                                                      : (SH::P4::R > SH::P6::R ? SH::P4::R : SH::P6::R);
    static_assert(NB_ >= 1 && NB_ <= 4, "sequences per decoder");
    static_assert(!COOP || NP == NB_, "cooperative staging: a merge / norm scratch per sequence");
    static_assert(NB_ == 1 || (!TP_ && WPC_ == 1 && !FUSED && NG == 1 && (FMT_ == FMT_Q4P || PREP) && DIM_ / 256 <= 12 && FFNP / 256 <= 24 && !COMB_IN_XS1),
                  "the batched form: plain (non-TP, one workgroup per CU) decoders of the shapes whose vectors are staged by the poller in one sweep");
};
// LDS of a workgroup: NB per-sequence blocks (activations, raw residuals, the attention's head staging, the rows of the phase being published, head maxima), then what the
// sequences share in turn (the waves' attention sums, the merge scratch, counters), then the layer table
template <class C>
struct XLay {
    static constexpr int hd = C::HD, GQ = C::GQW, NCW = C::NCW;
    static constexpr int maxK = C::DIM > C::QD ? (C::DIM > C::FFN ? C::DIM : C::FFN) : (C::QD > C::FFN ? C::QD : C::FFN);
    static constexpr int xs_bytes = (maxK * 4 + 15) & ~15, xr_bytes = (C::DIM * 2 + 15) & ~15;
    static constexpr size_t o_attn = (size_t)2 * xs_bytes + 2 * xr_bytes;
    static constexpr size_t o_outb = (o_attn + sizeof(uint16_t) * ((size_t)2 * GQ * hd + 3 * hd) + 15) & ~(size_t)15;
    static constexpr size_t o_wmax = o_outb + 4 * (size_t)((C::maxR + 63) & ~63);
    static constexpr size_t seq_bytes = (o_wmax + 4 * 2 * 16 + 15) & ~(size_t)15;
    static constexpr size_t msc_bytes = (sizeof(double) * ((size_t)C::ME * C::SPK) + 4 * 64 + 4 * 128 + 15) & ~(size_t)15; /* the slice merge's scratch of ONE poller ([SPK][ME] fp64 + shifts + the merged granules; xe_coop_norm_stage: a slot per wave) */
    static constexpr size_t o_msc = (size_t)C::NB * seq_bytes;
    static constexpr size_t o_comb = o_msc + (size_t)C::NP * msc_bytes;
    static constexpr size_t o_cnt = o_comb + (C::COMB_IN_XS1 ? 0 : sizeof(double) * (size_t)NCW * GQ * (hd + 2));
    static constexpr size_t o_tab = (o_cnt + 64 + 15) & ~(size_t)15; /* FMT_Q1T: the 256 x 16 B selector table of BlockPrep<FMT_Q1T> */
    static constexpr size_t fixed_bytes = o_tab + (C::FMT == FMT_Q1T ? 4096 : (C::FMT == FMT_Q2T ? 2048 : 0));
};
constexpr size_t xe_loc_stride(int loc_dw) { return ((size_t)loc_dw * 4 + 4095) & ~(size_t)4095; }

#define XE_STAMP(k)                                                                                                                       \
    do {                                                                                                                                  \
        if (C::DBG && S.stamp && lane == 0) a.dbg[((size_t)S.step * a.n_layer + l) * 64 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

struct XLds {
    const EngLayer* lay;
    u32x4* xs[2];
    uint16_t *xrawA, *xrawB, *qraw, *kraw, *vraw, *qb, *knew;
    float* wmax;
    double* comb; /* [NCW][GQ][hd + 2] */
    double* msc;  /* [SPK][ME] + [SPK] shifts */
    uint32_t* outb;
    int* cnt;
    const u32x4* qtab; /* FMT_Q1T: the selector table */
    int* pub; /* [4] pieces of q | k | v, xB, act, x this workgroup has published since the launch began: its poller sleeps until then instead of sweeping (a spinning poller takes
                 issue slots from the two compute waves of its SIMD, which then finish last and hold the whole decoder's hand-off back) */
};
// the LDS pointers of sub-sequence b of a decoder (XCfg::NB > 1): the per-sequence block b, the shared parts as they are
template <class C>
__device__ __forceinline__ XLds xe_lds_view(const XLds& L, int b) {
    if constexpr (C::NB == 1) {
        return L;
    } else {
        const size_t sh = (size_t)b * XLay<C>::seq_bytes;
        auto mv = [&](auto* q) { return reinterpret_cast<decltype(q)>(reinterpret_cast<unsigned char*>(q) + sh); };
        XLds V = L;
        V.xs[0] = mv(L.xs[0]), V.xs[1] = mv(L.xs[1]), V.xrawA = mv(L.xrawA), V.xrawB = mv(L.xrawB);
        V.qraw = mv(L.qraw), V.kraw = mv(L.kraw), V.vraw = mv(L.vraw), V.qb = mv(L.qb), V.knew = mv(L.knew);
        V.wmax = mv(L.wmax), V.outb = mv(L.outb);
        V.msc = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(L.msc) + (size_t)(b % C::NP) * XLay<C>::msc_bytes); /* the scratch of the sequence's poller */
        return V;
    }
}
struct XSeq { /* this workgroup's place in its decoder, and the step's slice */
    int seq, r, step; /* seq: the decoder (its exchange area, its XCD); TP: = the rank */
    int sq;           /* the sequence whose state / forced ids / ids out this decoder follows (TP: 0) */
    int pos, len, kvh, split, h0, t0, t1, me0;
    bool empty, own_new, stamp, grp0; /* grp0: the first head group of its kv-head (it writes the new K / V row) */
    bool act;                         /* the sequence exists, is not parked and stands inside its cache: an inactive one touches nothing outside its own exchange area */
    int j1, s1, M1, q_out0;
    long long kv_off; /* elements: this sequence's K/V cache behind sequence 0's */
};

// ---- ring of 16-byte blocks in flight (per lane), with their group's step / zero.  ONE ring serves every phase: entry e of a phase is the wave's block e (single-matrix
// phases) or -- gate | up -- block e / 2 of gate_proj (e even) resp. up_proj (e odd), so a slot that falls idle in one phase's last round takes the next phase's entry of the
// same index whatever the two phases are.
template <int D>
struct XRing {
    u32x4 w[D];
    uint16_t st[D], ze[D];
};
// One mat-vec phase as RUN-TIME parameters of one wave (all wave-uniform: scalar registers).  The geometry figures are the compile-time constants of PlanT (the lanes per row,
// rows per wave step and steps per row gemv_launch picks for the matrices: the canonical summation order), handed to ONE copy of the streaming loop: four typed copies of the
// DEPTH-times unrolled block code were 150 KB of instructions.
struct XPhase {
    EngMat m, m2;    /* m2: up_proj beside gate_proj (paired) */
    float qb, qb2;
    int nBlk, lpr_log2, iters, rps_log2;
    int paired;
    int s0, Mj;      /* the workgroup's first slot counted inside the matrix, + the wave's first (XDeal::a); the matrix's rows */
    int sl_b;        /* the wave's k-th slot: s0 + k * sl_b */
    int n;           /* ring entries this wave walks: its slots x iters (x 2 paired) */
    int row0;        /* first row of the workgroup's piece, counted inside the matrix */
    uint32_t wbytes, gbytes;
};
// Which of a workgroup's row slots a compute wave walks: slot sl_a + k * sl_b for k = 0 .. n - 1 (XPhase).  Waves w and w + 4 share a SIMD (measured: HW_ID of waves 0 .. 7 =
// SIMD 1 3 0 2 1 3 0 2) and a SIMD's waves finish one after the other -- a phase lasts as long as the busiest SIMD.  The poller (wave NCW) sits on the SIMD of class NCW & 3,
// whose compute waves are fewer (11 compute waves: two there, three elsewhere; 7: ONE there, two elsewhere).
//  * one decoder per XCD: the slots are dealt to SIMDs first (slot s -> class s & 3), then round the waves of the class -- every SIMD the same share, and at any moment the
//    workgroup's waves read neighbouring kilobytes (measured against contiguous runs per wave: 1.88 ms per step of eight sequences against 1.98)
//  * two per XCD (7 compute waves; the other decoder's waves fill the idle issue slots, the order of the reads is mixed anyway): a contiguous run per wave, its length by the
//    wave's weight -- `wl` / 8 of a plain wave's share for the wave beside the poller (XArgs::deal_wl; 16 = equal SIMD shares; measured 14: 3.449 ms per step of sixteen
//    sequences, 16: 3.507, 8: 3.511)
struct XDeal {
    int a, b, n;
};
template <int NCW, int NPOLL>
__device__ __forceinline__ int xe_deal_cum(int cw, int wl) { /* weights of waves 0 .. cw - 1: `wl` for a wave that shares its SIMD with a poller (waves NCW .. NCW + NPOLL - 1), 8 otherwise */
    int tot = 0;
#pragma unroll
    for (int w = 0; w < NCW; w++) {
        bool beside = false;
#pragma unroll
        for (int i = 0; i < NPOLL; i++) beside = beside || (((NCW + i) & 3) == (w & 3));
        tot += w < cw ? (beside ? wl : 8) : 0;
    }
    return tot;
}
template <int NCW, bool CONTIG, int NPOLL = 1>
__device__ __forceinline__ XDeal xe_deal(int cw, int spg, int wl) {
    if constexpr (CONTIG) {
        const int tot = xe_deal_cum<NCW, NPOLL>(NCW, wl);
        const int f0 = (spg * xe_deal_cum<NCW, NPOLL>(cw, wl) + (tot >> 1)) / tot, f1 = (spg * xe_deal_cum<NCW, NPOLL>(cw + 1, wl) + (tot >> 1)) / tot;
        return XDeal{f0, 1, f1 - f0};
    } else {
        const int cls = cw & 3, idx = cw >> 2, nw = (NCW - cls + 3) / 4;
        const int per_cls = spg > cls ? (spg - cls + 3) / 4 : 0; /* slots of the class */
        return XDeal{cls + 4 * idx, 4 * nw, per_cls > idx ? (per_cls - idx + nw - 1) / nw : 0};
    }
}
struct XLaneGeo { /* the lane's place in a phase's row slots */
    uint32_t vblk; /* sub * nBlk + ll: the lane's block offset inside a slot's iteration */
    int sub, ll;
};
__device__ __forceinline__ XLaneGeo xe_lane_geo(const XPhase& P, int lane) {
    XLaneGeo g;
    g.sub = lane >> P.lpr_log2, g.ll = lane & ((1 << P.lpr_log2) - 1);
    g.vblk = (uint32_t)g.sub * (uint32_t)P.nBlk + (uint32_t)g.ll;
    return g;
}
// ONE unconditional set of loads into ring slot d: entry e of phase P (use_nx = false) or of the next phase NX (use_nx = true), or -- on = false -- a load that touches no memory
// (a zero-sized descriptor: every lane is out of range and reads 0).  No branch around a load anywhere in the streaming loop: the compiler counts the loads in flight only along
// straight-line code -- behind a conditional request it waits with vmcnt(0), i.e. for the block it has just asked for (measured: every entry then paid an HBM round trip).
// Buffer loads: block index = [scalar: the slot's first row and the iteration] + [lane: sub * nBlk + ll]; rows past the matrix read zeros (the descriptor's bound), columns
// past the row are masked at the multiply.
template <int NCW, int D, int WAUX, int FMT = FMT_Q4P>
__device__ __forceinline__ void xe_issue(const XPhase& P, const XLaneGeo& G, const XPhase& NX, const XLaneGeo& GN, bool use_nx, int e, int d, int cw, bool on, XRing<D>& R) {
    // every field read into a value FIRST, then chosen: `c ? NX.f : P.f` on two lvalues is a choice between two ADDRESSES followed by one load, which keeps both structs
    // in scratch memory (and every such read an indexed scratch load with a drain of the weight loads in front of it)
    auto pick = [](bool c, auto x, auto y) { return c ? x : y; };
    const int paired = pick(use_nx, +NX.paired, +P.paired), iters = pick(use_nx, +NX.iters, +P.iters), s0 = pick(use_nx, +NX.s0, +P.s0), sl_b = pick(use_nx, +NX.sl_b, +P.sl_b);
    const int rps_log2 = pick(use_nx, +NX.rps_log2, +P.rps_log2), lpr_log2 = pick(use_nx, +NX.lpr_log2, +P.lpr_log2), nBlk = pick(use_nx, +NX.nBlk, +P.nBlk);
    const int k = paired ? e >> 1 : e;
    const int sl = k / iters, it = k - sl * iters; /* (measured: a multiply-shift in place of this division by a run-time scalar is 3.5 % SLOWER: 6750 against 7000 tokens/s with 32 sequences) */
    const uint32_t ublk = (uint32_t)(((s0 + sl * sl_b) << rps_log2) * nBlk + (it << lpr_log2)); /* wave-uniform; a multiple of 4 */
    const bool second = paired && (e & 1);
    const g_u32x4 w_a = P.m.w, w_b = P.m2.w, w_c = NX.m.w, w_d = NX.m2.w;
    const g_u16 s_a = P.m.step, s_b = P.m2.step, s_c = NX.m.step, s_d = NX.m2.step;
    const g_u16 z_a = P.m.zero, z_b = P.m2.zero, z_c = NX.m.zero, z_d = NX.m2.zero;
    const g_u32x4 pw = pick(use_nx, pick(second, w_d, w_c), pick(second, w_b, w_a));
    const g_u16 ps = pick(use_nx, pick(second, s_d, s_c), pick(second, s_b, s_a));
    const g_u16 pz = pick(use_nx, pick(second, z_d, z_c), pick(second, z_b, z_a));
    const uint32_t wbytes = on ? pick(use_nx, +NX.wbytes, +P.wbytes) : 0u, gbytes = on ? pick(use_nx, +NX.gbytes, +P.gbytes) : 0u;
    const uint32_t vblk = pick(use_nx, +GN.vblk, +G.vblk);
    // (rows that are not whole iterations -- a TP rank's 100-block down_proj rows on 64 lanes: a lane past the row's end reads the NEXT row's first blocks, masked at the multiply.
    //  Sending those lanes past the buffer instead was measured: 2.9 % less fetch traffic, 1.7 - 2.9 % MORE time -- the five instructions per entry cost more than the bytes)
    if constexpr (FMT == FMT_Q1T) { /* this lane's DWORD of the 16-byte block of 128 elements (dword 3 holds elements 0 .. 31): virtual block v = 32 weights -> real block v / 4, dword 3 - v % 4 (ublk is a multiple of 4) */
        R.w[d] = u32x4{(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(eng_rsrc((const void*)pw, wbytes), (vblk & ~3u) * 4u + (3u - (vblk & 3u)) * 4u, ublk * 4u, WAUX), 0u, 0u, 0u};
    } else if constexpr (FMT == FMT_Q2T) { /* this lane's 8-byte HALF of the 16-byte block of 64 elements (bytes 8 .. 15 hold elements 0 .. 31): virtual block v -> half (v & ~1) + 1 - (v & 1) */
        const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(eng_rsrc((const void*)pw, wbytes), (vblk & ~1u) * 8u + (1u - (vblk & 1u)) * 8u, ublk * 8u, WAUX));
        R.w[d] = u32x4{h.x, h.y, 0u, 0u}; /* .y = the half's first 16 elements, .x = its last 16 (BlockPrep<FMT_Q2T>) */
    } else {
        R.w[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(eng_rsrc((const void*)pw, wbytes), vblk * 16u, ublk * 16u, WAUX /* XCfg::WAUX */));
    }
    R.st[d] = __builtin_amdgcn_raw_buffer_load_b16(eng_rsrc((const void*)ps, gbytes), (vblk >> 2) * 2u, (ublk >> 2) * 2u, 0);
    R.ze[d] = __builtin_amdgcn_raw_buffer_load_b16(eng_rsrc((const void*)pz, gbytes), (vblk >> 2) * 2u, (ublk >> 2) * 2u, 0);
}
// a phase's first entries into an idle ring
template <int NCW, int D, int WAUX, int FMT = FMT_Q4P>
__device__ __forceinline__ void xe_fill(const XPhase& P, int cw, int lane, XRing<D>& R) {
    const XLaneGeo G = xe_lane_geo(P, lane);
#pragma unroll
    for (int d = 0; d < D; d++) xe_issue<NCW, D, WAUX, FMT>(P, G, P, G, false, d, d, cw, d < P.n, R);
}
// one block's 32 products into the lane's chain pair: BlockDotF<FMT> (kf_gemv_blocks.h) on the fp32 activation chunks in LDS, chunk j of block column c at xf[j * XS + c]
// (XS: the chunk stride, a compile-time constant of the model shape -- the same for every phase, so the eight reads of a block are one address and immediate offsets)
template <int FMT, int XS, bool LOW>
__device__ __forceinline__ f32x2_t xe_block(u32x4 w, uint16_t st16, uint16_t ze16, float qb, const f32x4* xc, int lane, f32x2_t acc) {
    const float step = bf2f(st16), zero = bf2f(ze16), nb = -(qb * step);
    const uint32_t D[4] = {w.w, w.z, w.y, w.x};
    if constexpr (FMT == FMT_Q4P) {
        const float q0 = (float)((lane & 3) << 2);
        uint32_t r = pack_bf16x2(fmaf(q0, step, nb), fmaf(q0 + 1.0f, step, nb));
        const uint32_t P0 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        r = pack_bf16x2(fmaf(q0 + 2.0f, step, nb), fmaf(q0 + 3.0f, step, nb));
        const uint32_t P1 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
        const uint32_t tlm = __builtin_amdgcn_perm(P1, P0, 0x06040200u), thm = __builtin_amdgcn_perm(P1, P0, 0x07050301u);
        PermLut t;
        t.tl[0] = quad_bcast<0>(tlm), t.tl[1] = quad_bcast<1>(tlm), t.tl[2] = quad_bcast<2>(tlm), t.tl[3] = quad_bcast<3>(tlm);
        t.th[0] = quad_bcast<0>(thm), t.th[1] = quad_bcast<1>(thm), t.th[2] = quad_bcast<2>(thm), t.th[3] = quad_bcast<3>(thm);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            acc = perm_fma_dword(D[i], xc[(2 * i) * XS], xc[(2 * i + 1) * XS], t, acc);
            if constexpr (LOW) __builtin_amdgcn_sched_barrier(0); /* two workgroups per CU (128 registers): the next dword's activation chunks are read when this one's products are done, not
                                                                     all eight chunks (32 registers) at the top of the block */
        }
    } else {
        const float step16 = step * 0.0625f;
#pragma unroll
        for (int i = 0; i < 4; i++) acc = arith_fma_dword(D[i], xc[(2 * i) * XS], xc[(2 * i + 1) * XS], step, step16, nb, zero, acc);
    }
    return acc;
}
// Storages without a register-table form (1-bit): the block's 16 bf16 pair words by BlockPrep<FMT> (kf_gemv_blocks.h: the words kf_engine.hip multiplies), widened ONCE to the
// fp32 operand pairs, then every sequence's chain pair takes its sixteen v_pk_fma_f32 -- pairs_dot<true>'s products, element after element
template <int FMT, int XS, int NB, size_t SEQB>
__device__ __forceinline__ void xe_block_prep_nb(u32x4 w, uint16_t st16, uint16_t ze16, float qb, const f32x4* xc0, int lane, const u32x4* tab, f32x2_t (&acc)[NB]) {
    const float step = bf2f(st16), zero = bf2f(ze16), nb = -(qb * step);
    uint32_t pw[16];
    BlockPrep<FMT>::prep(w, step, zero, nb, lane, pw, tab);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const f32x2_t w0{bf_lo(pw[4 * d]), bf_hi(pw[4 * d])}, w1{bf_lo(pw[4 * d + 1]), bf_hi(pw[4 * d + 1])};
        const f32x2_t w2{bf_lo(pw[4 * d + 2]), bf_hi(pw[4 * d + 2])}, w3{bf_lo(pw[4 * d + 3]), bf_hi(pw[4 * d + 3])};
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const f32x4* xc = reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(xc0) + (size_t)b * SEQB);
            const f32x4 X0 = xc[(2 * d) * XS], X1 = xc[(2 * d + 1) * XS];
            f32x2_t a = acc[b];
            a = pk_fma(w0, f32x2_t{X0.x, X0.y}, a);
            a = pk_fma(w1, f32x2_t{X0.z, X0.w}, a);
            a = pk_fma(w2, f32x2_t{X1.x, X1.y}, a);
            a = pk_fma(w3, f32x2_t{X1.z, X1.w}, a);
            acc[b] = a;
        }
    }
}
// The same block against the activations of NB sequences (XCfg::NB > 1): the group's table and the block's 32 weights are formed ONCE (the byte-plane lookups and the fp32
// assembly of perm_fma_dword), then every sequence's chain pair takes its sixteen v_pk_fma_f32 -- element after element of the block, as xe_block does for one sequence
template <int XS, int NB, size_t SEQB>
__device__ __forceinline__ void xe_block_nb(u32x4 w, uint16_t st16, uint16_t ze16, float qb, const f32x4* xc0, int lane, f32x2_t (&acc)[NB]) {
    const float step = bf2f(st16), zero = bf2f(ze16), nb = -(qb * step);
    const uint32_t D[4] = {w.w, w.z, w.y, w.x};
    const float q0 = (float)((lane & 3) << 2);
    uint32_t r = pack_bf16x2(fmaf(q0, step, nb), fmaf(q0 + 1.0f, step, nb));
    const uint32_t P0 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    r = pack_bf16x2(fmaf(q0 + 2.0f, step, nb), fmaf(q0 + 3.0f, step, nb));
    const uint32_t P1 = pack_bf16x2(bf_lo(r) - zero, bf_hi(r) - zero);
    const uint32_t tlm = __builtin_amdgcn_perm(P1, P0, 0x06040200u), thm = __builtin_amdgcn_perm(P1, P0, 0x07050301u);
    PermLut t;
    t.tl[0] = quad_bcast<0>(tlm), t.tl[1] = quad_bcast<1>(tlm), t.tl[2] = quad_bcast<2>(tlm), t.tl[3] = quad_bcast<3>(tlm);
    t.th[0] = quad_bcast<0>(thm), t.th[1] = quad_bcast<1>(thm), t.th[2] = quad_bcast<2>(thm), t.th[3] = quad_bcast<3>(thm);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t le, he, lo, ho;
        perm_lookup4(D[i] >> 4, t, le, he);
        perm_lookup4(D[i], t, lo, ho);
        const f32x2_t w0{__uint_as_float(__builtin_amdgcn_perm(he, le, 0x07030c0cu)), __uint_as_float(__builtin_amdgcn_perm(ho, lo, 0x07030c0cu))}; /* elements 0, 1 of the dword */
        const f32x2_t w1{__uint_as_float(__builtin_amdgcn_perm(he, le, 0x06020c0cu)), __uint_as_float(__builtin_amdgcn_perm(ho, lo, 0x06020c0cu))}; /* 2, 3 */
        const f32x2_t w2{__uint_as_float(__builtin_amdgcn_perm(he, le, 0x05010c0cu)), __uint_as_float(__builtin_amdgcn_perm(ho, lo, 0x05010c0cu))}; /* 4, 5 */
        const f32x2_t w3{__uint_as_float(__builtin_amdgcn_perm(he, le, 0x04000c0cu)), __uint_as_float(__builtin_amdgcn_perm(ho, lo, 0x04000c0cu))}; /* 6, 7 */
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const f32x4* xc = reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(xc0) + (size_t)b * SEQB);
            const f32x4 X0 = xc[(2 * i) * XS], X1 = xc[(2 * i + 1) * XS];
            f32x2_t a = acc[b];
            a = pk_fma(w0, f32x2_t{X0.x, X0.y}, a);
            a = pk_fma(w1, f32x2_t{X0.z, X0.w}, a);
            a = pk_fma(w2, f32x2_t{X1.x, X1.y}, a);
            a = pk_fma(w3, f32x2_t{X1.z, X1.w}, a);
            acc[b] = a;
        }
    }
}
// One mat-vec phase of a compute wave: ONE copy of this loop serves every phase.  The ring holds the wave's first min(D, n) entries on entry (xe_fill / the previous phase's
// last round); entry e + D is requested when entry e has been multiplied, and in the last round slot d takes entry d of the NEXT phase NX (nx_on; they do not depend on the
// hand-off that separates the phases).  epi(row, v, v2) runs in the lane that owns a finished row (LDS only: no memory operation inside the loop but the refills).
template <class C, int D, typename Epi>
__device__ __forceinline__ void xe_mv_run(const XPhase& P, const XPhase& NX, bool nx_on, int cw, int lane, const u32x4* xs, XRing<D>& R, Epi&& epi, const u32x4* qtab = nullptr) {
    constexpr int NCW = C::NCW, XS = C::XS, NB = C::NB;
    static_assert((D % 2) == 0, "gate | up entries come in pairs");
    const int n = P.n, n_pad = n > 0 ? (n + D - 1) / D * D : D; /* at least one round: the last round is where the next phase's entries are requested */
    const f32x4* xf = reinterpret_cast<const f32x4*>(xs); /* sequence 0's activations; sequence b's lie XLay::seq_bytes x b behind them */
    const XLaneGeo G = xe_lane_geo(P, lane), GN = xe_lane_geo(NX, lane);
    f32x2_t acc[NB], acc2[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = f32x2_t{0.f, 0.f}, acc2[b] = f32x2_t{0.f, 0.f};
    for (int e0 = 0; e0 < n_pad; e0 += D) {
        const bool last = e0 + D >= n_pad;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int e = e0 + d;
            if (e < n) {
                const int k = P.paired ? e >> 1 : e;
                const int sl = k / P.iters, it = k - sl * P.iters;
                const int row = ((P.s0 + sl * P.sl_b) << P.rps_log2) + G.sub, colr = (it << P.lpr_log2) + G.ll;
                const bool second = P.paired && (d & 1);
                const float qb_a = P.qb, qb_b = P.qb2;
                if constexpr (NB == 1 && !C::PREP) {
                    if (it == 0) {
                        if (second) acc2[0] = f32x2_t{0.f, 0.f};
                        else acc[0] = f32x2_t{0.f, 0.f};
                    }
                    const f32x2_t in = second ? acc2[0] : acc[0];
                    f32x2_t o;
                    if constexpr (C::EXACT) { /* whole rows, whole iterations: nothing to mask */
                        o = xe_block<C::FMT, XS, (C::WPC > 1)>(R.w[d], R.st[d], R.ze[d], second ? qb_b : qb_a, xf + colr, lane, in);
                    } else {
                        const bool ok = row < P.Mj && colr < P.nBlk;
                        const int col = colr < P.nBlk ? colr : P.nBlk - 1;
                        o = acc_pick(ok, xe_block<C::FMT, XS, (C::WPC > 1)>(R.w[d], R.st[d], R.ze[d], second ? qb_b : qb_a, xf + col, lane, in), in);
                    }
                    if (second) acc2[0] = o;
                    else acc[0] = o;
                } else { /* NB sequences against one unpacked block */
                    f32x2_t in[NB], o[NB];
#pragma unroll
                    for (int b = 0; b < NB; b++) in[b] = it == 0 ? f32x2_t{0.f, 0.f} : (second ? acc2[b] : acc[b]), o[b] = in[b];
                    const bool ok = C::EXACT || (row < P.Mj && colr < P.nBlk);
                    const int col = (C::EXACT || colr < P.nBlk) ? colr : P.nBlk - 1;
                    if constexpr (C::PREP) xe_block_prep_nb<C::FMT, XS, NB, XLay<C>::seq_bytes>(R.w[d], R.st[d], R.ze[d], second ? qb_b : qb_a, xf + col, lane, qtab, o);
                    else xe_block_nb<XS, NB, XLay<C>::seq_bytes>(R.w[d], R.st[d], R.ze[d], second ? qb_b : qb_a, xf + col, lane, o);
#pragma unroll
                    for (int b = 0; b < NB; b++) {
                        const f32x2_t r = C::EXACT ? o[b] : acc_pick(ok, o[b], in[b]);
                        if (second) acc2[b] = r;
                        else acc[b] = r;
                    }
                }
                if (it == P.iters - 1 && (!P.paired || second)) {
#pragma unroll
                    for (int b = 0; b < NB; b++) {
                        const float v = group_sum(acc_join(acc[b]), P.lpr_log2);
                        float v2 = 0.f;
                        if (P.paired) v2 = group_sum(acc_join(acc2[b]), P.lpr_log2);
                        if (G.ll == 0 && (C::EXACT || row < P.Mj)) epi(b, row, v, v2);
                    }
                }
            }
            // refill slot d (always ONE set of loads: see xe_issue)
            const bool more = e + D < n, nxt = !more && last && nx_on && d < NX.n;
            xe_issue<NCW, D, C::WAUX, C::FMT>(P, G, NX, GN, nxt, more ? e + D : (nxt ? d : 0), d, cw, more || nxt, R);
        }
    }
}
// the waves that own rows of a phase leave their granules in LDS; the one that arrives last stores the workgroup's piece, 16 bytes per lane, with PLAIN stores (this XCD's L2)
// plain != NULL: the rows also as plain bf16 (the residual stream after the last layer: x_out)
// push (TP, o_proj / down_proj): the rows are fp32 partials; they go as {fp32 | generation} granules into this rank's slot of EVERY rank's receive area (push: rank 0's
// slot for these rows, push_stride granules from one rank's area to the next), agent-scope stores -- the areas are read from the other XCDs
// pad: zero granules behind the piece (the SwiGLU vector of a rank is swept in whole 1 KiB units)
// NB > 1: the pieces of the decoder's NB sequences, one after the other -- rows in L.outb + b * LBS bytes, destination dst + b * dst_bs dwords (plain + b * plain_bs elements);
// actm: bit b set = sequence b is active (an inactive one publishes nothing)
template <int NB = 1, size_t LBS = 0>
__device__ __forceinline__ void xe_publish(const XLds& L, int phase, uint32_t tag, uint32_t* dst, int nrows, int nwaves, int lane, uint16_t* plain = nullptr,
                                           unsigned long long* push = nullptr, size_t push_stride = 0, uint32_t tagx = 0, int pad = 0, size_t dst_bs = 0, size_t plain_bs = 0, uint32_t actm = 1u,
                                           g_i32 hot = nullptr /* phase 2, sparse forward: hot[] of this workgroup's first gate / up row on (D_matmul_sparse: a cold row's output is 0) */) {
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(L.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != nwaves - 1) return;
    if (lane == 0) *L.cnt = 0;
    if (push) {
        for (int i = lane; i < nrows; i += 64) {
            const float v = __uint_as_float(L.outb[i]);
#pragma unroll
            for (int d = 0; d < XE_NXCD; d++) st_gran64(push + (size_t)d * push_stride + i, tagx, v);
        }
        if (lane == 0) __hip_atomic_fetch_add(L.pub + phase, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (NB > 1 && !((actm >> b) & 1u)) continue;
        uint32_t* const ob = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(L.outb) + (size_t)b * LBS);
        uint32_t* const db = dst + (size_t)b * dst_bs;
        uint16_t* const pb = plain ? plain + (size_t)b * plain_bs : nullptr;
        for (int i = lane; i < pad; i += 64) db[nrows + i] = tag << 16;
        // one descriptor for the piece, the lane's 16 bytes as an offset: no 64-bit per-lane address (it was spilled, and its reload in front of the store drained the next phase's
        // weight loads in flight)
        const __amdgpu_buffer_rsrc_t rd = eng_rsrc(db, (uint32_t)nrows * 4u), rp = eng_rsrc(pb ? (const void*)pb : (const void*)db, pb ? (uint32_t)nrows * 2u : 0u);
        if (phase == 2) { /* gate | up: the rows were left as bf16 pairs {gate, up}: SwiGLU (CU_swiglu_v0) and the tag now */
            for (int i = lane; i < nrows; i += 64) {
                const uint32_t pr = ob[i];
                const float gt = bf_lo(pr), up = bf_hi(pr);
                const bool cold = hot != nullptr && hot[i] != 1; /* (the single-sequence engine and the masked launches publish the same zero) */
                ob[i] = (tag << 16) | (cold ? 0u : (uint32_t)f2bf((gt * up) / (1.0f + kf_expf(-gt))));
            }
        }
        for (int i = 4 * lane; i < nrows; i += 256) {
            const u32x4 g = *reinterpret_cast<const u32x4*>(ob + i);
            __builtin_amdgcn_raw_buffer_store_b128(g, rd, i * 4, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{(g.x & 0xffffu) | (g.y << 16), (g.z & 0xffffu) | (g.w << 16)}, rp, i * 2, 0, 0); /* out of range (dropped) without `plain` */
        }
    }
    if (lane == 0) __hip_atomic_fetch_add(L.pub + phase, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- TP: this workgroup's 1 / 32 of a column-shard exchange.  Rows wg * R .. + R of the eight ranks' fp32 partials (this rank's receive area, buffer `which`) are swept until
// every granule carries the exchange's generation, summed in rank order, rounded, added to the residual (tp_reduce_recv_kernel, kf_tp.hip: bf16(x + bf16(sum_r p_r))) and left
// as tagged granules in this XCD's area -- where the ordinary local hand-off picks the whole vector up.  plain: the rows also as plain bf16 (x_out)
template <class C>
__device__ __forceinline__ void xe_tp_reduce(const XArgs& a, const XSeq& S, int which, uint32_t tagx, const uint16_t* resid, uint32_t* dst_local, uint32_t tag16, uint16_t* plain, int lane, bool& dead) {
    constexpr int R = C::DIM / XE_NWG, NK = (R + 63) / 64;
    const int row0 = S.r * R;
    const unsigned long long* base = a.tp_recv + ((size_t)S.seq * 2 + which) * XE_NXCD * C::DIM + row0;
    u32x2 g[XE_NXCD][NK];
    for (int spins = 0;; spins++) {
        uint32_t bad = 0;
#pragma unroll
        for (int s = 0; s < XE_NXCD; s++) {
            const __amdgpu_buffer_rsrc_t rs = eng_rsrc(base + (size_t)s * C::DIM, (uint32_t)R * 8u);
#pragma unroll
            for (int k = 0; k < NK; k++) g[s][k] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (k * 64 + lane) * 8, 0, 16 /* sc1 */));
        }
#pragma unroll
        for (int s = 0; s < XE_NXCD; s++)
#pragma unroll
            for (int k = 0; k < NK; k++) bad |= (k * 64 + lane) < R ? (g[s][k].y ^ tagx) : 0u;
        if (all_good(bad)) break;
        if (dead || spins > ENG_SPIN_MAX) {
            if (!dead && lane == 0) atomicOr(a.ws + 1, 2048);
            dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
    }
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int i = k * 64 + lane;
        if (i < R) {
            float tot = __uint_as_float(g[0][k].x);
#pragma unroll
            for (int s = 1; s < XE_NXCD; s++) tot = tot + __uint_as_float(g[s][k].x);
            const uint16_t o = f2bf(bf2f(resid[row0 + i]) + bf2f(f2bf(tot)));
            dst_local[i] = (tag16 << 16) | (uint32_t)o;
            if (plain) plain[row0 + i] = o;
        }
    }
}

// ---- attention: the workgroup's key slice of its kv-head, streamed by the NCW compute waves (two batches of U tiles in flight per lane)
template <class C>
struct XAttn {
    static constexpr int U = C::AU;
    u32x4 kk[2][U], vv[2][U];
    uint16_t nw0, nw1;
    float rc[C::NB], rs[C::NB]; /* the lane's RoPE pair at the position of sub-sequence b */
};
template <class C>
__device__ __forceinline__ void xe_attn_issue(const XArgs& a, const EngLayer& ly, const XSeq& S, int cw, int lane, XAttn<C>& T, int b, int buf) {
    constexpr int hd = C::HD, LPK = hd >> 3, KPW = 64 / LPK, lpk_log2 = (C::HD == 128 ? 7 : 6) - 3, NWA = C::NCW, U = XAttn<C>::U;
    const int grp = lane >> lpk_log2, d0 = (lane & (LPK - 1)) * 8;
    const int tb = S.t0 + cw * KPW + grp + b * U * NWA * KPW;
    const int tmax = S.t1 > 0 ? S.t1 - 1 : 0; /* unconditional requests (a conditional one turns the waits behind it into drains): rows past the slice re-read its last row, masked at the use */
#pragma unroll
    for (int u = 0; u < U; u++) {
        int t = tb + u * NWA * KPW;
        t = t < tmax ? t : tmax;
        const size_t off = (size_t)S.kv_off + (size_t)t * a.kv_stride + (size_t)S.kvh * hd + d0;
        if constexpr (C::KV_NT) { /* a sequence's own rows, read once per step */
            T.kk[buf][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 KF_GLOBAL*>(ly.kcache + off));
            T.vv[buf][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 KF_GLOBAL*>(ly.vcache + off));
        } else {
            T.kk[buf][u] = *reinterpret_cast<const u32x4 KF_GLOBAL*>(ly.kcache + off);
            T.vv[buf][u] = *reinterpret_cast<const u32x4 KF_GLOBAL*>(ly.vcache + off);
        }
    }
}
// compute waves only (the poller meets the three barriers in xe_poller_main).  p4_fill: requests the first o_proj blocks, called when the last batch's tiles are in registers
// bi: the sub-sequence (XCfg::NB > 1: called once per sequence of the decoder, S and L being that sequence's views); p4_fill runs behind the key loop (the last sequence: the
// first o_proj blocks; before that: the next sequence's first tiles)
template <class C, typename Fill>
__device__ __forceinline__ void xe_attn_phase(const XArgs& a, const XLds& L, const XSeq& S, const EngLayer& ly, uint32_t gen, int cw, int lane, XAttn<C>& T, int l, Fill&& p4_fill, int bi = 0) {
    constexpr int GQ = C::GQW /* the heads of this workgroup's group */, hd = C::HD, hd_log2 = C::HD == 128 ? 7 : 6, LPK = hd >> 3, KPW = 64 / LPK, lpk_log2 = hd_log2 - 3, NWA = C::NCW, U = XAttn<C>::U;
    constexpr int NQ = (GQ + NWA - 1) / NWA;
    const int tid = (cw << 6) | lane, pos = S.pos, t1 = S.t1;
    const int grp = lane >> lpk_log2, d0 = (lane & (LPK - 1)) * 8;
    const int tstride = NWA * KPW, tstart = S.t0 + cw * KPW + grp;
    const int nbatch = S.empty ? 0 : (S.t1 - S.t0 + U * tstride - 1) / (U * tstride);
    __syncthreads(); /* raw heads staged */
    if (!S.empty) {
        const bool rope = a.rope_table != nullptr, qnorm = ly.norm_q != nullptr;
        const int half = hd >> 1, j = lane < half ? lane : half - 1;
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const int hq = cw + i * NWA;
            if (hq < GQ) {
                HeadRaw r;
                r.x0 = L.qraw[hq * hd + j], r.x1 = L.qraw[hq * hd + j + half];
                r.w0 = qnorm ? T.nw0 : r.x0, r.w1 = qnorm ? T.nw1 : r.x1;
                prep_head_cs(r, qnorm, rope, T.rc[bi], T.rs[bi], hd, a.qk_eps, L.qb + hq * hd, nullptr, lane);
            }
        }
        if (S.own_new && cw == (GQ % NWA)) {
            HeadRaw r;
            r.x0 = L.kraw[j], r.x1 = L.kraw[j + half];
            r.w0 = ly.norm_k ? T.nw0 : r.x0, r.w1 = ly.norm_k ? T.nw1 : r.x1;
            prep_head_cs(r, ly.norm_k != nullptr, rope, T.rc[bi], T.rs[bi], hd, a.qk_eps, L.knew, nullptr, lane);
        }
    }
    __syncthreads(); /* heads prepared */
    if (!S.empty) {
        if (S.own_new && S.grp0 && tid < hd / 8) { /* the cache rows of this position: the prepared key, the raw value (K.out / V.out alias them in the reference) */
            g_u16w krow = ly.kcache + (size_t)S.kv_off + (size_t)pos * a.kv_stride + (size_t)S.kvh * hd;
            g_u16w vrow = ly.vcache + (size_t)S.kv_off + (size_t)pos * a.kv_stride + (size_t)S.kvh * hd;
            *reinterpret_cast<u32x4 KF_GLOBAL*>(const_cast<uint16_t KF_GLOBAL*>(krow) + 8 * tid) = *reinterpret_cast<const u32x4*>(L.knew + 8 * tid);
            *reinterpret_cast<u32x4 KF_GLOBAL*>(const_cast<uint16_t KF_GLOBAL*>(vrow) + 8 * tid) = *reinterpret_cast<const u32x4*>(L.vraw + 8 * tid);
        }
        CanonAcc<GQ> A;
        A.init();
        float qf[GQ][8];
#pragma unroll
        for (int hq = 0; hq < GQ; hq++) {
            const u32x4 qv = *reinterpret_cast<const u32x4*>(L.qb + hq * hd + d0);
            const uint32_t q4[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int i = 0; i < 4; i++) qf[hq][2 * i] = bf_lo(q4[i]), qf[hq][2 * i + 1] = bf_hi(q4[i]);
        }
        const float rden = 1.0f / sqrtf((float)hd);
        auto batch = [&](int b, int buf) {
            const int tb = tstart + b * U * tstride;
            u32x4 ck[U], cv[U];
            bool valid[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int t = tb + u * tstride;
                valid[u] = t < t1;
                ck[u] = T.kk[buf][u], cv[u] = T.vv[buf][u];
                if (valid[u] && t == pos) ck[u] = *reinterpret_cast<const u32x4*>(L.knew + d0), cv[u] = *reinterpret_cast<const u32x4*>(L.vraw + d0);
            }
            canon_batch<GQ, LPK, U>(A, qf, ck, cv, valid, lpk_log2, rden);
        };
        for (int b = 0; b < nbatch; b += 2) { /* batch b sits in buffer 0 (requested by the phase before, or by the step below), b + 1 goes to buffer 1; requests past the slice are clamped */
            xe_attn_issue<C>(a, ly, S, cw, lane, T, b + 1, 1);
            batch(b, 0);
            if (b + 1 >= nbatch) break;
            xe_attn_issue<C>(a, ly, S, cw, lane, T, b + 2, 0);
            batch(b + 1, 1);
        }
        canon_wave_to_lds<GQ, LPK>(A, L.comb + (size_t)cw * GQ * (hd + 2), hd, lane, d0);
    }
    p4_fill(); /* the first o_proj blocks: two hand-offs (slice partials, ao) lie between here and their use -- inside the key loop the ring would cost the loop 48 registers */
    __syncthreads(); /* the waves' sums in LDS */
    // the slice's partial {O[hd], L, m} per query head, as {32 bits, generation} granule pairs in this XCD's partial area (an empty slice: sums 0, exponent -inf)
    constexpr int PSD = hd + 2, ME = C::ME, SPK = C::SPK;
    unsigned long long* const pbase = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint32_t*>(a.loc + (size_t)S.seq * a.loc_stride) + C::part);
    const unsigned long long gg = (unsigned long long)gen << 32;
    for (int i = tid; i < (S.act ? GQ * hd : 0); i += NWA * 64) {
        const int hq = i >> hd_log2, d = i & (hd - 1);
        double o = 0.0, Ls = 0.0;
        float ms = -__builtin_inff();
        if (!S.empty) {
#pragma unroll
            for (int sl = 0; sl < NWA; sl++) ms = fmaxf(ms, (float)L.comb[((size_t)sl * GQ + hq) * PSD + hd + 1]);
#pragma unroll
            for (int sl = 0; sl < NWA; sl++) {
                const double* c = L.comb + ((size_t)sl * GQ + hq) * PSD;
                const int e = canon_shift((float)c[hd + 1] - ms);
                o += ldexp_d(c[d], e);
                Ls += ldexp_d(c[hd], e);
            }
        }
        unsigned long long* dst = pbase + (size_t)(S.h0 + hq) * C::PSH;
        const size_t oi = ((size_t)(d / ME) * (SPK * ME) + (size_t)S.split * ME + (d & (ME - 1))) * 2, mi = (size_t)hd * SPK * 2 + (size_t)S.split * 4;
        const unsigned long long ob = __builtin_bit_cast(unsigned long long, o), lb = __builtin_bit_cast(unsigned long long, Ls);
        *reinterpret_cast<ulonglong2*>(dst + oi) = ulonglong2{gg | (ob & 0xffffffffull), gg | (ob >> 32)};
        if (d == 0) {
            *reinterpret_cast<ulonglong2*>(dst + mi) = ulonglong2{gg | __float_as_uint(ms), gg | (lb & 0xffffffffull)};
            dst[mi + 2] = gg | (lb >> 32);
        }
    }
}

// ---- cooperative RMSNorm staging (XCfg::COOP): wave w owns the 1 KiB units w, w + NWV, ... of the vector.  Sweep until tagged; raw bf16 -> xraw (the residual of the phase
// after next), fp64 sum of squares -> one LDS slot per wave; barrier; total = the slots in wave order (an fp64 sum: the order is far below an fp32 ulp -- the argument of
// oracle section 5); normalise and stage the own units as fp32 chunks.  The caller's phase barrier follows.
template <class C>
__device__ __forceinline__ void xe_coop_norm_stage(const XArgs& a, const XLds& L, const uint32_t* gsrc, uint32_t tag, g_u16 norm_w, u32x4* xs, uint16_t* xraw, int wave, int lane, bool* dead_io) {
    constexpr int ND = C::DIM / 256, NWV = C::NWV, NR = (ND + NWV - 1) / NWV, XCH = C::XCH, NBLK = C::XS;
    const __amdgpu_buffer_rsrc_t rs = eng_rsrc(gsrc, (uint32_t)ND * 1024u);
    u32x4 g[NR];
    u32x2 wn[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int r = wave + k * NWV;
        wn[k] = *reinterpret_cast<const u32x2 KF_GLOBAL*>(norm_w + 4 * ((r < ND ? r : 0) * 64 + lane));
    }
    const uint32_t tagw = tag << 16;
    bool dead = dead_io ? *dead_io : false;
    for (int spins = 0;; spins++) {
        uint32_t bad = 0;
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int r = wave + k * NWV;
            g[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((r < ND ? r : 0) * 64 + lane) * 16, 0, 16 /* sc1 */));
        }
#pragma unroll
        for (int k = 0; k < NR; k++) bad = (wave + k * NWV) < ND ? tags_bad(g[k], tagw, bad) : bad;
        if (all_good(bad)) break;
        if (dead || spins > ENG_SPIN_MAX || ((spins & 1023) == 1023 && __hip_atomic_load(a.ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            if (!dead && spins > ENG_SPIN_MAX && lane == 0) atomicOr(a.ws + 1, 1);
            dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
    }
    if (dead_io) *dead_io = dead;
    uint32_t p0[NR], p1[NR];
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int r = wave + k * NWV;
        p0[k] = (g[k].x & 0xffffu) | (g[k].y << 16), p1[k] = (g[k].z & 0xffffu) | (g[k].w << 16);
        if (r < ND) {
            const double x0 = (double)bf_lo(p0[k]), x1 = (double)bf_hi(p0[k]), x2 = (double)bf_lo(p1[k]), x3 = (double)bf_hi(p1[k]);
            ss = fma(x0, x0, ss), ss = fma(x1, x1, ss), ss = fma(x2, x2, ss), ss = fma(x3, x3, ss);
            *reinterpret_cast<u32x2*>(xraw + 4 * (r * 64 + lane)) = u32x2{p0[k], p1[k]};
        }
    }
    const double mine = wave_sum_f64_fast(ss);
    if (lane == 0) L.msc[wave] = mine;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; w++) tot += L.msc[w];
    const float mul = 1.0f / sqrtf(fmaf((float)tot, 1.0f / (float)C::DIM, a.eps));
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int r = wave + k * NWV;
        if (r < ND) {
            const int e0 = 4 * (r * 64 + lane), q = e0 >> 2, c = q / XCH, j = q - c * XCH;
            const uint32_t o0 = pack_bf16x2((bf_lo(p0[k]) * mul) * bf_lo(wn[k].x), (bf_hi(p0[k]) * mul) * bf_hi(wn[k].x));
            const uint32_t o1 = pack_bf16x2((bf_lo(p1[k]) * mul) * bf_lo(wn[k].y), (bf_hi(p1[k]) * mul) * bf_hi(wn[k].y));
            xs[j * NBLK + c] = u32x4{o0 << 16, o0 & 0xffff0000u, o1 << 16, o1 & 0xffff0000u};
        }
    }
}

// ---- the hand-offs of ONE sequence of a decoder (sweep until whole, stage into that sequence's LDS block; the slice merge): run by that sequence's poller wave
template <class C>
__device__ __forceinline__ void xe_ho_x(const XArgs& a, const XLds& Lb, const XSeq& Sb, const EngLayer& ly, int l, int epoch, uint32_t gen, int lane, bool& dead) { /* P1's x (P4 adds it as the residual) */
    constexpr int XCH = C::XCH, ND = C::DIM / 256, RT = C::DIM / XE_NWG;
    const uint32_t tag = gen & 0xffffu;
    uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)Sb.seq * a.loc_stride);
    if (l == 0) {
        int tok = a.d_state[Sb.sq * 4];
        if (Sb.step > 0) { /* the id workgroup 0 of this decoder picked at the end of the previous step: {id, epoch of this step} */
            const __amdgpu_buffer_rsrc_t rt = eng_rsrc(loc + C::tokg, 8u);
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
                __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
            }
        }
        if (a.d_forced) {
            const int f = a.d_forced[(size_t)Sb.sq * a.forced_stride + Sb.pos];
            if (f >= 0) tok = f;
        }
        if (tok < 0 || tok >= a.emb_rows) tok = 0;
        if constexpr (ND > 12) eng_poll_stage_norm_long<XCH, ND, C::XS, true, true, 8>(nullptr, a.emb + (size_t)tok * C::DIM, tag, ly.norm_in, a.eps, Lb.xs[0], Lb.xrawA, lane, a.ws, dead);
        else eng_poll_stage<XCH, ND, C::XS, true, true, true>(nullptr, a.emb + (size_t)tok * C::DIM, tag, ly.norm_in, a.eps, Lb.xs[0], Lb.xrawA, lane, a.ws, dead, nullptr, nullptr, 0, 0);
    } else {
        if constexpr (C::TP) /* the down_proj exchange of the layer before: this workgroup's rows summed over the ranks + the residual xB -> the local x area */
            xe_tp_reduce<C>(a, Sb, 1, 2u * (gen - 1u) + 2u, Lb.xrawB, loc + C::xA + Sb.r * RT, tag, nullptr, lane, dead);
        if constexpr (C::COOP) { /* every wave stages its own 1 KiB units of every sequence's vector: xe_coop_all, called by the poller and by the compute waves */
        } else if constexpr (ND > 12) eng_poll_stage_norm_long<XCH, ND, C::XS, false, true, 8>(loc + C::xA, nullptr, tag, ly.norm_in, a.eps, Lb.xs[0], Lb.xrawA, lane, a.ws, dead);
        else eng_poll_stage<XCH, ND, C::XS, true, false, true>(loc + C::xA, nullptr, tag, ly.norm_in, a.eps, Lb.xs[0], Lb.xrawA, lane, a.ws, dead, nullptr, nullptr, 0, 0);
    }
}
template <class C>
__device__ __forceinline__ void xe_ho_qkv(const XArgs& a, const XLds& Lb, const XSeq& Sb, uint32_t tag, int lane, bool& dead) { /* P2: the raw q heads of this workgroup's kv-head, its k and v rows */
    constexpr int GQ = C::GQW, hd = C::HD;
    uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)Sb.seq * a.loc_stride);
    const __amdgpu_buffer_rsrc_t rs = eng_rsrc(loc + C::qkv, (uint32_t)(C::QD + 2 * C::KVD) * 4u);
    const uint32_t tagw = tag << 16;
    constexpr int NLQ = (GQ * hd + 255) / 256;
    u32x4 g[NLQ], gk;
    const int e_kv = 4 * lane; /* < hd: k, < 2 hd: v */
    const bool kv_in = e_kv < 2 * hd;
    const int kv_src = e_kv < hd ? C::QD + Sb.kvh * hd + e_kv : C::QD + C::KVD + Sb.kvh * hd + (e_kv - hd);
    for (int spins = 0;; spins++) {
        uint32_t bad = 0;
#pragma unroll
        for (int r = 0; r < NLQ; r++) g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (Sb.h0 * hd + 4 * (r * 64 + lane)) * 4, 0, 16));
        gk = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (kv_in ? kv_src : 0) * 4, 0, 16));
#pragma unroll
        for (int r = 0; r < NLQ; r++) bad = (4 * (r * 64 + lane) < GQ * hd) ? tags_bad(g[r], tagw, bad) : bad;
        bad = kv_in ? tags_bad(gk, tagw, bad) : bad;
        if (all_good(bad)) break;
        if (dead || spins > ENG_SPIN_MAX) {
            if (!dead && lane == 0) atomicOr(a.ws + 1, 2);
            dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
    }
#pragma unroll
    for (int r = 0; r < NLQ; r++) {
        const int e0 = 4 * (r * 64 + lane);
        if (e0 < GQ * hd) *reinterpret_cast<u32x2*>(Lb.qraw + e0) = u32x2{(g[r].x & 0xffffu) | (g[r].y << 16), (g[r].z & 0xffffu) | (g[r].w << 16)};
    }
    if (kv_in) *reinterpret_cast<u32x2*>(Lb.kraw + e_kv) = u32x2{(gk.x & 0xffffu) | (gk.y << 16), (gk.z & 0xffffu) | (gk.w << 16)}; /* vraw = kraw + hd */
}
// P3: merge the SPK slices of this workgroup's ME output elements (exact rescales to the largest exponent, fp64 sums, one division: kf_attn_common.h) -> the ao area
template <class C>
__device__ __forceinline__ void xe_ho_merge(const XArgs& a, const XLds& Lb, const XSeq& Sb, uint32_t gen, int lane, bool& dead) {
    constexpr int hd = C::HD;
    const uint32_t tag = gen & 0xffffu;
    uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)Sb.seq * a.loc_stride);
    constexpr int ME = C::ME, SPK = C::SPK, NV = ME * SPK, NLM = (NV * 2 + 127) / 128; /* values; 16-byte loads (2 granules) per lane and sweep */
    const int h = Sb.me0 >> (hd == 128 ? 7 : 6), dd = Sb.me0 & (hd - 1);
    const unsigned long long* hbase = reinterpret_cast<const unsigned long long*>(loc + C::part) + (size_t)h * C::PSH;
    const __amdgpu_buffer_rsrc_t rs_o = eng_rsrc(hbase + (size_t)(dd / ME) * (SPK * ME * 2), (uint32_t)(SPK * ME * 2) * 8u);
    const __amdgpu_buffer_rsrc_t rs_ml = eng_rsrc(hbase + (size_t)hd * SPK * 2, (uint32_t)SPK * 32u);
    u32x4 go[NLM], gm0, gm1;
    const bool mine = lane < SPK;
    for (int spins = 0;; spins++) {
        uint32_t bad = 0;
#pragma unroll
        for (int r = 0; r < NLM; r++) go[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_o, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
        gm0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ml, (mine ? lane : 0) * 32, 0, 16));
        gm1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ml, (mine ? lane : 0) * 32 + 16, 0, 16));
#pragma unroll
        for (int r = 0; r < NLM; r++) bad |= (r * 64 + lane) < NV ? ((go[r].y ^ gen) | (go[r].w ^ gen)) : 0u;
        bad |= mine ? ((gm0.y ^ gen) | (gm0.w ^ gen) | (gm1.y ^ gen)) : 0u;
        if (all_good(bad)) break;
        if (dead || spins > ENG_SPIN_MAX) {
            if (!dead && lane == 0) atomicOr(a.ws + 1, 4);
            dead = true;
            break;
        }
        __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
    }
#pragma unroll
    for (int r = 0; r < NLM; r++) { /* value index vi = sp * ME + e */
        const int vi = r * 64 + lane;
        if (vi < NV) Lb.msc[vi] = __builtin_bit_cast(double, ((unsigned long long)go[r].z << 32) | go[r].x);
    }
    const float ms = mine ? __uint_as_float(gm0.x) : -__builtin_inff();
    const double ls = mine ? __builtin_bit_cast(double, ((unsigned long long)gm1.x << 32) | gm0.z) : 0.0;
    const float Mx = wave_max(ms);
    const int sh = canon_shift(ms - Mx);
    const double Lt = wave_sum_f64_fast(ldexp_d(ls, sh));
    int* shl = reinterpret_cast<int*>(Lb.msc + NV);
    if (mine) shl[lane] = sh;
    uint32_t* mo = reinterpret_cast<uint32_t*>(Lb.msc + NV) + 64;
#pragma unroll
    for (int e0 = 0; e0 < ME; e0 += 64) {
        const int e = e0 + lane;
        if (e < ME) {
            double o = 0.0;
#pragma unroll
            for (int sp = 0; sp < SPK; sp++) o += ldexp_d(Lb.msc[sp * ME + e], shl[sp]);
            mo[e] = (tag << 16) | (uint32_t)f2bf((float)(o / Lt));
        }
    }
    if (4 * lane < ME) *reinterpret_cast<u32x4*>(loc + C::ao + Sb.me0 + 4 * lane) = *reinterpret_cast<const u32x4*>(mo + 4 * lane);
}
template <class C>
__device__ __forceinline__ void xe_ho_ao(const XArgs& a, const XLds& Lb, const XSeq& Sb, uint32_t tag, int lane, bool& dead) { /* P4's ao */
    uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)Sb.seq * a.loc_stride);
    eng_poll_stage<C::XCH, C::QD / 256, C::XS, false, false, true>(loc + C::ao, nullptr, tag, nullptr, 0.f, Lb.xs[1], nullptr, lane, a.ws, dead, nullptr, nullptr, 0, 0);
}
template <class C>
__device__ __forceinline__ void xe_ho_xB(const XArgs& a, const XLds& Lb, const XSeq& Sb, const EngLayer& ly, uint32_t gen, int lane, bool& dead) { /* P5's xB (P6 adds it as the residual) */
    constexpr int XCH = C::XCH, ND = C::DIM / 256, RT = C::DIM / XE_NWG;
    const uint32_t tag = gen & 0xffffu;
    uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)Sb.seq * a.loc_stride);
    if constexpr (C::TP) xe_tp_reduce<C>(a, Sb, 0, 2u * gen + 1u, Lb.xrawA, loc + C::xB + Sb.r * RT, tag, nullptr, lane, dead); /* the o_proj exchange + the residual x */
    if constexpr (C::COOP) { /* xe_coop_all */
    } else if constexpr (ND > 12) eng_poll_stage_norm_long<XCH, ND, C::XS, false, true, 8>(loc + C::xB, nullptr, tag, ly.norm_post, a.eps, Lb.xs[0], Lb.xrawB, lane, a.ws, dead);
    else eng_poll_stage<XCH, ND, C::XS, true, false, true>(loc + C::xB, nullptr, tag, ly.norm_post, a.eps, Lb.xs[0], Lb.xrawB, lane, a.ws, dead, nullptr, nullptr, 0, 0);
}
template <class C>
__device__ __forceinline__ void xe_ho_act(const XArgs& a, const XLds& Lb, const XSeq& Sb, uint32_t tag, int lane, bool& dead, int* nsw) { /* P6's act */
    constexpr int XCH = C::XCH, NF = C::FFNP / 256;
    uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)Sb.seq * a.loc_stride);
    if constexpr (NF > 24) eng_poll_stage_long<XCH, NF, C::XS, 16>(loc + C::act, tag, Lb.xs[1], lane, a.ws, dead, nsw); /* a 9728-wide vector in one sweep: 152 registers */
    else eng_poll_stage<XCH, NF, C::XS, false, false, true>(loc + C::act, nullptr, tag, nullptr, 0.f, Lb.xs[1], nullptr, lane, a.ws, dead, nsw, nullptr, 0, 0);
}
// COOP shapes: this wave's units of x (which = 0: the x area, norm_in, xrawA) or xB (1: the xB area, norm_post, xrawB) of EVERY active sequence of the decoder -- called by
// every wave of the workgroup (a barrier per sequence inside xe_coop_norm_stage)
template <class C>
__device__ __forceinline__ void xe_coop_all(const XArgs& a, const XLds& L0, const XSeq (&SS)[C::NB], int which, uint32_t tag, const EngLayer& ly, int wave, int lane, bool* dead) {
#pragma unroll
    for (int b = 0; b < C::NB; b++) {
        if (C::NB > 1 && !SS[b].act) continue;
        const XLds Lb = xe_lds_view<C>(L0, b);
        uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)SS[b].seq * a.loc_stride);
        if (which == 0) xe_coop_norm_stage<C>(a, Lb, loc + C::xA, tag, ly.norm_in, Lb.xs[0], Lb.xrawA, wave, lane, dead);
        else xe_coop_norm_stage<C>(a, Lb, loc + C::xB, tag, ly.norm_post, Lb.xs[0], Lb.xrawB, wave, lane, dead);
    }
}
// ---- poller wave PB of a workgroup: the hand-offs of the decoder's sequences PB, PB + NP, ... (one after the other), and every barrier; stamps: poller 0's first sequence
template <class C, int PB>
__device__ __forceinline__ void xe_poller_main(const XArgs& a, const XLds& L0, const XSeq (&SS)[C::NB], int epoch, int lane) {
    constexpr int NB = C::NB, NP = C::NP, NMINE = NB / NP;
    static_assert(C::DIM % 256 == 0 && C::QD % 256 == 0 && (C::TP || C::FFN % 256 == 0), "hand-off vectors in 1 KiB pieces");
    XSeq S = SS[PB];
    if (PB > 0) S.stamp = false;
    const XLds L = xe_lds_view<C>(L0, PB);
    bool dead = false;
    for (int l = 0; l < a.n_layer; l++) {
        const EngLayer& ly = L.lay[l];
        const uint32_t gen = (uint32_t)epoch * (uint32_t)a.n_layer + (uint32_t)l, tag = gen & 0xffffu;
        XE_STAMP(0);
        if (C::DBG && !C::TP && S.stamp && lane == 0) a.dbg[((size_t)S.step * a.n_layer + l) * 64 + 31] = __builtin_amdgcn_s_memtime(); /* the shader clock beside the 100 MHz stamp: the frequency the CU really runs at */
        if (l > 0) {
            eng_wait_pub(L.pub + 3, S.step * a.n_layer + l, 0, dead);
            XE_STAMP(12);
        }
#pragma unroll
        for (int k = 0; k < NMINE; k++)
            if (NB == 1 || SS[PB + k * NP].act) xe_ho_x<C>(a, xe_lds_view<C>(L0, PB + k * NP), SS[PB + k * NP], ly, l, epoch, gen, lane, dead);
        if constexpr (C::COOP) {
            if (l > 0) xe_coop_all<C>(a, L0, SS, 0, tag, ly, C::NCW + PB, lane, &dead);
        }
        if (C::TP && l > 0) XE_STAMP(31); /* (TP: the slot of the shader-clock stamp) this workgroup's rows of the down_proj exchange are summed */
        XE_STAMP(1);
        __syncthreads(); /* B1 */
        if (!(C::P1W < XE_NWG && S.r >= C::P1W)) eng_wait_pub(L.pub + 0, S.step * a.n_layer + l + 1, 0, dead); /* (a workgroup without q | k | v rows publishes none) */
        XE_STAMP(9);
#pragma unroll
        for (int k = 0; k < NMINE; k++)
            if (!SS[PB + k * NP].empty) xe_ho_qkv<C>(a, xe_lds_view<C>(L0, PB + k * NP), SS[PB + k * NP], tag, lane, dead); /* only a slice with keys needs the heads (an inactive sequence is an empty one) */
        XE_STAMP(2);
#pragma unroll
        for (int b = 0; b < NB; b++) { /* the compute waves' attention, sequence after sequence (xe_attn_phase) */
            __syncthreads(); /* raw heads staged */
            __syncthreads(); /* heads prepared */
            __syncthreads(); /* the waves' sums in LDS */
        }
        XE_STAMP(3);
#pragma unroll
        for (int k = 0; k < NMINE; k++)
            if (NB == 1 || SS[PB + k * NP].act) xe_ho_merge<C>(a, xe_lds_view<C>(L0, PB + k * NP), SS[PB + k * NP], gen, lane, dead);
        XE_STAMP(5);
#pragma unroll
        for (int k = 0; k < NMINE; k++)
            if (NB == 1 || SS[PB + k * NP].act) xe_ho_ao<C>(a, xe_lds_view<C>(L0, PB + k * NP), SS[PB + k * NP], tag, lane, dead);
        XE_STAMP(6);
        __syncthreads(); /* B4 */
        eng_wait_pub(L.pub + 1, S.step * a.n_layer + l + 1, 0, dead);
        XE_STAMP(10);
#pragma unroll
        for (int k = 0; k < NMINE; k++)
            if (NB == 1 || SS[PB + k * NP].act) xe_ho_xB<C>(a, xe_lds_view<C>(L0, PB + k * NP), SS[PB + k * NP], ly, gen, lane, dead);
        if constexpr (C::COOP) xe_coop_all<C>(a, L0, SS, 1, tag, ly, C::NCW + PB, lane, &dead);
        if constexpr (C::TP) XE_STAMP(13); /* (TP: the slot of the act sweep count) this workgroup's rows of the o_proj exchange are summed */
        XE_STAMP(7);
        __syncthreads(); /* B5 */
        eng_wait_pub(L.pub + 2, S.step * a.n_layer + l + 1, 0, dead);
        XE_STAMP(11);
        int nsw_act = 0;
#pragma unroll
        for (int k = 0; k < NMINE; k++)
            if (NB == 1 || SS[PB + k * NP].act) xe_ho_act<C>(a, xe_lds_view<C>(L0, PB + k * NP), SS[PB + k * NP], tag, lane, dead, &nsw_act);
        if (C::DBG && !C::TP && S.stamp && lane == 0) a.dbg[((size_t)S.step * a.n_layer + l) * 64 + 13] = (unsigned long long)nsw_act;
        XE_STAMP(8);
        __syncthreads(); /* B6 */
    }
}

// ---- the compute waves
template <class C>
__device__ __forceinline__ void xe_compute_main(const XArgs& a, const XLds& L, const XSeq (&SS)[C::NB], int epoch, int cw, int lane, XRing<C::DEPTH>& R) {
    constexpr int NB = C::NB;
    const XSeq& S = SS[0]; /* the workgroup's place (and, NB == 1, the sequence) */
    using SH = typename C::SH;
    using P1 = typename SH::P1;
    using P4 = typename SH::P4;
    using P5 = typename SH::P5;
    using P6 = typename SH::P6;
    constexpr int NCW = C::NCW, D = C::DEPTH;
    static_assert(P1::R % 4 == 0 && P4::R % 4 == 0 && P5::R % 4 == 0 && P6::R % 4 == 0, "16-byte pieces");
    static_assert(P1::total == C::P1W * P1::spg && P4::total == XE_NWG * P4::spg && P5::total == XE_NWG * P5::spg && P6::total == XE_NWG * P6::spg, "every workgroup owns rows of every phase");
    static_assert(C::FUSED || (P1::S1 % P1::spg == 0 && P1::S2 % P1::spg == 0), "a workgroup's P1 rows belong to one matrix");
    static_assert(!C::TP || (P4::R == C::DIM / XE_NWG && P6::R == C::DIM / XE_NWG && C::WPC == 1), "TP: a workgroup sums the rows it produced (xe_tp_reduce)");
    uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)S.seq * a.loc_stride);
    const float qb1 = S.j1 == 0 ? a.qbias[0] : (S.j1 == 1 ? a.qbias[1] : a.qbias[2]);
    const int wg = S.r;
    XAttn<C> T;
    uint32_t actm = 0; /* bit b: sequence b of the decoder is active */
#pragma unroll
    for (int b = 0; b < NB; b++) {
        actm |= SS[b].act ? (1u << b) : 0u;
        T.rc[b] = 1.f, T.rs[b] = 0.f;
        if (a.rope_table && lane < (C::HD >> 1)) {
            const float* tab_pos = a.rope_table + (size_t)SS[b].pos * C::HD;
            T.rc[b] = tab_pos[2 * lane], T.rs[b] = tab_pos[2 * lane + 1];
        }
    }
    const size_t loc_bs = (size_t)XE_NXCD * a.loc_stride / 4; /* dwords from sequence b's exchange area to sequence b + 1's (sequences xcc + 8 b) */
    // phase q of a layer (0: q | k | v, 1: o_proj, 2: gate | up, 3: down_proj) as scalars: every field by a chain of selects on q -- never a struct chosen among four (the
    // compiler keeps such a value in scratch memory and every later field read becomes an indexed scratch load, in front of which it drains the weight loads in flight)
    auto sel4 = [](int q, auto v0, auto v1, auto v2, auto v3) { return q == 0 ? v0 : (q == 1 ? v1 : (q == 2 ? v2 : v3)); };
    auto phase_of = [&](int q, const EngLayer& ly) {
        auto uni = [](auto ptr) {
            const unsigned long long v = (unsigned long long)(uintptr_t)ptr;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return (decltype(ptr))(uintptr_t)(((unsigned long long)hi << 32) | lo);
        };
        const int j = sel4(q, S.j1, 3, 4, 6), j2 = q == 2 ? 5 : j;
        XPhase P;
        P.m.w = uni(ly.m[j].w), P.m.step = uni(ly.m[j].step), P.m.zero = uni(ly.m[j].zero);
        P.m2.w = uni(ly.m[j2].w), P.m2.step = uni(ly.m[j2].step), P.m2.zero = uni(ly.m[j2].zero);
        P.qb = sel4(q, qb1, a.qbias[3], a.qbias[4], a.qbias[6]), P.qb2 = a.qbias[5];
        P.nBlk = sel4(q, P1::nBlk, P4::nBlk, P5::nBlk, P6::nBlk), P.lpr_log2 = sel4(q, P1::lpr_log2, P4::lpr_log2, P5::lpr_log2, P6::lpr_log2);
        P.iters = sel4(q, P1::iters, P4::iters, P5::iters, P6::iters), P.rps_log2 = 6 - P.lpr_log2, P.paired = q == 2 ? 1 : 0;
        P.s0 = sel4(q, S.s1, wg * P4::spg, wg * P5::spg, wg * P6::spg), P.Mj = sel4(q, S.M1, P4::M0, P5::M0, P6::M0);
        P.row0 = P.s0 << P.rps_log2;
        const int spg = sel4(q, P1::spg, P4::spg, P5::spg, P6::spg);
        const XDeal dl = xe_deal<NCW, C::DEAL_CONTIG, C::NP>(cw, spg, a.deal_wl);
        P.s0 += __builtin_amdgcn_readfirstlane(dl.a), P.sl_b = __builtin_amdgcn_readfirstlane(dl.b);
        P.n = __builtin_amdgcn_readfirstlane(dl.n) * P.iters * (P.paired ? 2 : 1);
        if (C::P1W < XE_NWG && q == 0 && wg >= C::P1W) P.n = 0;
        P.wbytes = (uint32_t)P.Mj * (uint32_t)P.nBlk * (uint32_t)C::VBYTES, P.gbytes = (uint32_t)P.Mj * (uint32_t)(P.nBlk / 4) * 2u;
        return P;
    };
    static_assert(P1::nBlk % 4 == 0 && P4::nBlk % 4 == 0 && P5::nBlk % 4 == 0 && P6::nBlk % 4 == 0 && P1::LPR % 4 == 0 && P4::LPR % 4 == 0 && P5::LPR % 4 == 0 && P6::LPR % 4 == 0,
                  "a group's four blocks start at a multiple of four");
    xe_fill<NCW, D, C::WAUX, C::FMT>(phase_of(0, L.lay[0]), cw, lane, R);
    for (int l = 0; l < a.n_layer; l++) {
        const EngLayer& ly = L.lay[l];
        const uint32_t gen = (uint32_t)epoch * (uint32_t)a.n_layer + (uint32_t)l, tag = gen & 0xffffu, tag_next = (gen + 1u) & 0xffffu;
        const bool last = l == a.n_layer - 1;
        const EngLayer& lyn = L.lay[last ? l : l + 1];
        if (NB > 1 || !S.empty) { /* q-norm (waves that prepare a q head) / k-norm (the wave that prepares the new key) weights of this lane's pair */
            const int half = C::HD >> 1, j = lane < half ? lane : half - 1;
            g_u16 np = cw < C::GQW ? ly.norm_q : ly.norm_k;
            T.nw0 = T.nw1 = 0;
            if (np) T.nw0 = np[j], T.nw1 = np[j + half];
        }
        for (int q = 0; q < 4; q++) {
            // q = 0: RMSNorm(x) -> q | k | v rows, then (below) the attention; 1: o_proj + residual -> xB; 2: RMSNorm + gate | up + SwiGLU -> act; 3: down_proj + residual -> next x
            const XPhase P = phase_of(q, ly);
            const XPhase NX = phase_of(q == 3 ? 0 : q + 1, q == 3 ? lyn : ly);
            const bool nx_on = q == 0 ? false : (q == 3 ? !last : true); /* behind q | k | v comes the attention (its first tiles are requested instead); the ring is not carried through the head phase */
            const u32x4* xs = (q == 0 || q == 2) ? L.xs[0] : L.xs[1];
            const int nrows = q == 0 ? P1::R : (q == 1 ? P4::R : (q == 2 ? P5::R : P6::R));
            const int spg = q == 0 ? P1::spg : (q == 1 ? P4::spg : (q == 2 ? P5::spg : P6::spg));
            int nwp = 0; /* waves that own rows of the phase */
#pragma unroll
            for (int w = 0; w < NCW; w++) nwp += xe_deal<NCW, C::DEAL_CONTIG, C::NP>(w, spg, a.deal_wl).n > 0 ? 1 : 0;
            uint32_t* const dst = loc + (q == 0 ? C::qkv + S.q_out0 : (q == 1 ? C::xB + wg * P4::R : (q == 2 ? C::act + wg * P5::R : C::xA + wg * P6::R)));
            if constexpr (C::COOP) { /* this wave's share of the sweep + RMSNorm + staging of x (layers behind the first: layer 0's row comes from the embedding table) / xB */
                if (q == 0 && l > 0) xe_coop_all<C>(a, L, SS, 0, tag, ly, cw, lane, nullptr);
                if (q == 2) xe_coop_all<C>(a, L, SS, 1, tag, ly, cw, lane, nullptr);
            }
            __syncthreads(); /* the phase's activations are staged (B1 / B4 / B5 / B6) */
            if (cw == 0) XE_STAMP(16 + 2 * q);
            xe_mv_run<C, D>(
                P, NX, nx_on, cw, lane, xs, R,
                [&](int b, int row, float v, float v2) {
                    const XLds Lb = xe_lds_view<C>(L, b);
                    uint32_t g;
                    if (q == 0) {
                        g = (tag << 16) | (uint32_t)f2bf(v);
                    } else if (C::TP && (q == 1 || q == 3)) {
                        g = __float_as_uint(v); /* a column shard's row: the un-rounded fp32 partial (xe_publish pushes it to every rank) */
                    } else if (q == 1) {
                        g = (tag << 16) | (uint32_t)f2bf(bf2f(Lb.xrawA[row]) + bf2f(f2bf(v))); /* CU_add3: bf16(x + bf16(W.x)) */
                    } else if (q == 2) {
                        g = pack_bf16x2(v, v2); /* the two bf16-rounded projections; CU_swiglu_v0 on them runs in xe_publish, once per row with every lane busy (here: the whole wave would walk
                                                   the exponential and the division for two rows) */
                    } else {
                        g = (tag_next << 16) | (uint32_t)f2bf(bf2f(Lb.xrawB[row]) + bf2f(f2bf(v)));
                    }
                    Lb.outb[row - P.row0] = g;
                },
                L.qtab);
            XE_STAMP(32 + 8 * q + (cw & 7)); /* this wave's rows of the phase are done */
            if (C::DBG && S.stamp && lane == 0 && q == 0) {
                unsigned hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                a.dbg[((size_t)S.step * a.n_layer + l) * 64 + ((cw & 7) < 2 ? 14 + (cw & 7) : 23 + (cw & 7))] = hw; /* slots 14, 15, 25 .. 30: where (SIMD, CU) the wave runs */
            }
            if (q == 0 && (NB == 1 || SS[0].act)) xe_attn_issue<C>(a, ly, SS[0], cw, lane, T, 0, 0); /* the slice's first tiles (rows of earlier positions; row `pos` is substituted): the q | k | v hand-off hides them */
            if (xe_deal<NCW, C::DEAL_CONTIG, C::NP>(cw, spg, a.deal_wl).n > 0 && !(C::P1W < XE_NWG && q == 0 && wg >= C::P1W)) {
                if constexpr (C::TP) {
                    // o_proj (q == 1) / down_proj (q == 3): this rank's slot [buffer][S.seq][rows wg * R ..] of rank 0's receive area, the other ranks' areas 2 * 8 * DIM granules apart
                    unsigned long long* push = (q == 1 || q == 3) ? a.tp_recv + ((size_t)(q == 3 ? 1 : 0) * XE_NXCD + S.seq) * C::DIM + wg * nrows : nullptr;
                    xe_publish(L, q, tag, dst, nrows, nwp, lane, nullptr, push, (size_t)2 * XE_NXCD * C::DIM, 2u * gen + (q == 3 ? 2u : 1u), (q == 2 && wg == XE_NWG - 1) ? C::FFNP - C::FFN : 0);
                } else {
                    xe_publish<NB, XLay<C>::seq_bytes>(L, q, tag, dst, nrows, nwp, lane, (q == 3 && last) ? a.x_out + (size_t)S.seq * C::DIM + wg * P6::R : nullptr, nullptr, 0, 0, 0, loc_bs,
                                                       (size_t)XE_NXCD * C::DIM, actm, (q == 2 && ly.hot) ? ly.hot + wg * P5::R : nullptr);
                }
            }
            if (cw == 0) XE_STAMP(17 + 2 * q);
            if (q == 0) { /* q/k-norm + RoPE + attention over the workgroup's key slice; the slice partial into the XCD's partial area; then the first o_proj blocks */
#pragma unroll
                for (int b = 0; b < NB; b++) { /* sequence after sequence (own positions, own slices, own K / V rows); behind a sequence's key loop the next one's first tiles are requested */
                    xe_attn_phase<C>(
                        a, xe_lds_view<C>(L, b), SS[b], ly, gen, cw, lane, T, l,
                        [&]() {
                            if (b + 1 < NB) {
                                if (SS[b + 1 < NB ? b + 1 : b].act) xe_attn_issue<C>(a, ly, SS[b + 1 < NB ? b + 1 : b], cw, lane, T, 0, 0);
                            } else {
                                xe_fill<NCW, D, C::WAUX, C::FMT>(phase_of(1, ly), cw, lane, R);
                            }
                        },
                        b);
                }
                if (cw == 0) XE_STAMP(24);
            }
        }
    }
}

// ---- the LM head + greedy pick as trailing phases (eng_head_main of kf_engine.hip on 32 workgroups): final RMSNorm, the [vocab, dim] bf16 mat-vec with the arithmetic
// of gemv_kernel<FMT_BF16, .., CANON> (same lanes per row, chain pair, tree, bf16 store), first-maximum arg-max over the stored values
template <class C>
__device__ __forceinline__ void xe_head_main(const XArgs& a, const XLds& L, const XSeq (&SS)[C::NB], int epoch, bool more_steps, int wave, int lane) {
    constexpr int NWV = C::NWV, NWG = XE_NWG, nBlk = C::HnBlk, LPR = C::HLPR, RPS = C::HRPS, ITERS = C::Hiters, NB = C::NB;
#ifndef XE_HEAD_HG_NB4
#define XE_HEAD_HG_NB4 2
#endif
    constexpr int HG = NB > 2 ? XE_HEAD_HG_NB4 : (ITERS <= 4 ? 4 : (ITERS <= 5 ? 2 : 1)); /* row slots per batch and wave, two batches in flight: 2 x HG x ITERS x 4 registers (2560-wide rows: 80, 4096-wide: 64) */
    constexpr int ND = C::DIM / 256;
    const XSeq& S = SS[0];
    const int wg = S.r;
    // TP: this rank's vocabulary shard (rows row0 .. of the full head), its logits; chosen by compares (an index into the kernel arguments at run time is a copy in scratch)
    uint16_t* logits = a.logits + (size_t)S.seq * a.vocab; /* sequence b's: + 8 b vocab */
    g_u32x4 head_w = a.head_w;
    int vocab = a.vocab, row0g = 0;
    if constexpr (C::TP) {
#pragma unroll
        for (int r = 0; r < XE_NXCD; r++)
            if (r == S.seq) logits = a.logits_r[r], head_w = a.head_w_r[r], vocab = a.vocab_r[r], row0g = a.row0_r[r];
    }
    const int sub = lane >> C::Hlpr_log2, ll = lane & (LPR - 1);
    const int total = (vocab + RPS - 1) / RPS, spg = (total + NWG - 1) / NWG;
    const int s_wg = wg * spg;
    int s_end = s_wg + spg;
    s_end = s_end < total ? s_end : total;
    const int nmine = s_end > s_wg + wave ? (s_end - s_wg - wave + NWV - 1) / NWV : 0;
    const int nbatch = (nmine + HG - 1) / HG;
    u32x4 w[2][HG][ITERS];
    auto issue = [&](int b, int buf) {
#pragma unroll
        for (int g = 0; g < HG; g++) {
            int i = b * HG + g;
            i = i < nmine ? i : (nmine > 0 ? nmine - 1 : 0);
            int row = (s_wg + wave + NWV * i) * RPS + sub;
            row = row < vocab ? row : vocab - 1;
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                int col = it * LPR + ll;
                col = col < nBlk ? col : nBlk - 1;
                if constexpr (C::TP) w[buf][g][it] = __builtin_nontemporal_load(head_w + (size_t)row * nBlk + col); /* a rank's vocabulary shard: read once */
                else w[buf][g][it] = head_w[(size_t)row * nBlk + col]; /* the same rows for every decoder (XCfg::WAUX) */
            }
        }
    };
    const uint32_t gen = (uint32_t)(epoch + 1) * (uint32_t)a.n_layer, tag = gen & 0xffffu; /* the generation the last layer's down_proj published its rows with */
    if (wave >= C::NCW) { /* the pollers: each its sequence's final x */
        bool dead = false;
        if constexpr (C::TP) { /* the last layer's down_proj exchange: this workgroup's rows -> the local x area (rank 0: also x_out) */
            uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)S.seq * a.loc_stride);
            eng_wait_pub(L.pub + 3, (S.step + 1) * a.n_layer, 0, dead);
            xe_tp_reduce<C>(a, S, 1, 2u * (gen - 1u) + 2u, L.xrawB, loc + C::xA + wg * (C::DIM / XE_NWG), tag, S.seq == 0 ? a.x_out : nullptr, lane, dead);
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (wave == C::NCW + b % C::NP && (NB == 1 || SS[b].act)) {
                const XLds Lb = xe_lds_view<C>(L, b);
                uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)SS[b].seq * a.loc_stride);
                if constexpr (ND > 12) eng_poll_stage_norm_long<1, ND, nBlk, false, false, 8>(loc + C::xA, nullptr, tag, a.head_norm, a.eps, Lb.xs[0], Lb.xrawA, lane, a.ws, dead);
                else eng_poll_stage<1, ND, nBlk, true, false, false>(loc + C::xA, nullptr, tag, a.head_norm, a.eps, Lb.xs[0], nullptr, lane, a.ws, dead, nullptr, L.pub + 3, (S.step + 1) * a.n_layer, 0);
            }
        }
    } else {
        issue(0, 0); /* ahead of the hand-off of x (the pollers' own first rows are requested behind their sweeps: loads return in order) */
    }
    __syncthreads();
    if (wave >= C::NCW) issue(0, 0);
    float xf[NB][ITERS][8];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const XLds Lb = xe_lds_view<C>(L, b);
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            int col = it * LPR + ll;
            col = col < nBlk ? col : nBlk - 1;
            const u32x4 xv = Lb.xs[0][col];
            const uint32_t x4[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int i = 0; i < 4; i++) xf[b][it][2 * i] = bf_lo(x4[i]), xf[b][it][2 * i + 1] = bf_hi(x4[i]);
        }
    }
    float best_v[NB];
    int best_i[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) best_v[b] = -__builtin_inff(), best_i[b] = 0x7fffffff;
    auto compute = [&](int bt, int buf) {
#pragma unroll
        for (int g = 0; g < HG; g++) {
            const int i = bt * HG + g;
            const int row = (s_wg + wave + NWV * i) * RPS + sub;
#pragma unroll
            for (int b = 0; b < NB; b++) { /* the rows are read once; every sequence of the decoder takes its own chain pair over them */
                f32x2_t acc{0.f, 0.f};
#pragma unroll
                for (int it = 0; it < ITERS; it++) {
                    const uint32_t w4[4] = {w[buf][g][it].x, w[buf][g][it].y, w[buf][g][it].z, w[buf][g][it].w};
                    f32x2_t r = acc;
#pragma unroll
                    for (int i2 = 0; i2 < 4; i2++) r = pk_fma(f32x2_t{bf_lo(w4[i2]), bf_hi(w4[i2])}, f32x2_t{xf[b][it][2 * i2], xf[b][it][2 * i2 + 1]}, r);
                    acc = acc_pick(it * LPR + ll < nBlk, r, acc);
                }
                const float v = group_sum(acc_join(acc), C::Hlpr_log2);
                if (ll == 0 && i < nmine && row < vocab && (NB == 1 || SS[b].act)) {
                    const uint16_t o = f2bf(v);
                    logits[(size_t)b * XE_NXCD * vocab + row] = o;
                    const float fv = bf2f(o);
                    if (fv > best_v[b] || (fv == best_v[b] && row < best_i[b])) best_v[b] = fv, best_i[b] = row;
                }
            }
        }
    };
    for (int b = 0; b < nbatch; b += 2) {
        issue(b + 1, 1);
        compute(b, 0);
        if (b + 1 >= nbatch) break;
        issue(b + 2, 0);
        compute(b + 1, 1);
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const float ov = __shfl_xor(best_v[b], m, 64);
            const int oi = __shfl_xor(best_i[b], m, 64);
            if (ov > best_v[b] || (ov == best_v[b] && oi < best_i[b])) best_v[b] = ov, best_i[b] = oi;
        }
        const XLds Lb = xe_lds_view<C>(L, b);
        float* rv = Lb.wmax;
        int* ri = reinterpret_cast<int*>(Lb.wmax + NWV);
        if (lane == 0) rv[wave] = best_v[b], ri[wave] = best_i[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const XSeq& Sb = SS[b];
        if (NB > 1 && !Sb.act) continue;
        const XLds Lb = xe_lds_view<C>(L, b);
        uint32_t* const loc = reinterpret_cast<uint32_t*>(a.loc + (size_t)Sb.seq * a.loc_stride);
        float* rv = Lb.wmax;
        int* ri = reinterpret_cast<int*>(Lb.wmax + NWV);
        unsigned long long* hb = reinterpret_cast<unsigned long long*>(loc + C::hbest);
        if (wave == 0 && lane == 0) {
            float bv0 = best_v[b];
            int bi0 = best_i[b];
            for (int k = 1; k < NWV; k++)
                if (rv[k] > bv0 || (rv[k] == bv0 && ri[k] < bi0)) bv0 = rv[k], bi0 = ri[k];
            hb[wg] = ((unsigned long long)((tag << 16) | (uint32_t)f2bf(bv0)) << 32) | (unsigned long long)(uint32_t)bi0;
        }
        if (wg == 0 && wave == C::NCW + b % C::NP && a.pick) { /* the pick over the decoder's 32 workgroup maxima (sequence b's poller): two granules per lane */
            const __amdgpu_buffer_rsrc_t rs = eng_rsrc(hb, NWG * 8u);
            u32x4 g0{0, 0, 0, 0};
            bool ok = false;
            const bool mine = lane < NWG / 2;
            for (int spins = 0; spins <= ENG_SPIN_MAX; spins++) {
                g0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (mine ? lane : 0) * 16, 0, 16 /* sc1 */));
                const uint32_t bad = ((g0.y >> 16) ^ tag) | ((g0.w >> 16) ^ tag);
                if (all_good(bad)) {
                    ok = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
            }
            float bv = -__builtin_inff();
            int bi = 0x7fffffff;
            if (mine) {
                const uint32_t hv[2] = {g0.y, g0.w}, hi[2] = {g0.x, g0.z};
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const float fv = bf2f((uint16_t)(hv[k] & 0xffffu));
                    const int ix = (int)hi[k];
                    if (fv > bv || (fv == bv && ix < bi)) bv = fv, bi = ix;
                }
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) {
                const float ov = __shfl_xor(bv, m, 64);
                const int oi = __shfl_xor(bi, m, 64);
                if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
            }
            int vocab_all = vocab;
            if constexpr (C::TP) { /* the rank's maximum (GLOBAL row) to every rank; then the first maximum over the ranks (tp_pick_kernel, kf_tp.hip: lowest row among equals) */
                vocab_all = 0;
#pragma unroll
                for (int r = 0; r < XE_NXCD; r++) vocab_all += a.vocab_r[r];
                if (lane < XE_NXCD) {
                    const unsigned long long gr = ((unsigned long long)((tag << 16) | (uint32_t)f2bf(bv)) << 32) | (unsigned long long)(uint32_t)(bi + row0g);
                    __hip_atomic_store(a.tp_best + (size_t)lane * XE_NXCD + S.seq, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const __amdgpu_buffer_rsrc_t rb = eng_rsrc(a.tp_best + (size_t)S.seq * XE_NXCD, XE_NXCD * 8u);
                u32x2 gb{0, 0};
                const bool src = lane < XE_NXCD;
                bool ok2 = false;
                for (int spins = 0; spins <= ENG_SPIN_MAX; spins++) {
                    gb = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rb, (src ? lane : 0) * 8, 0, 16 /* sc1 */));
                    if (all_good((gb.y >> 16) ^ tag)) {
                        ok2 = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(KF_SWEEP_SLEEP);
                }
                ok = ok && ok2;
                bv = src ? bf2f((uint16_t)(gb.y & 0xffffu)) : -__builtin_inff(), bi = src ? (int)gb.x : 0x7fffffff;
#pragma unroll
                for (int m = 4; m > 0; m >>= 1) {
                    const float ov = __shfl_xor(bv, m, 64);
                    const int oi = __shfl_xor(bi, m, 64);
                    if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
                }
            }
            if (lane == 0) {
                const int err = __hip_atomic_load(a.ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ok && err == 0 && bi >= 0 && bi < vocab_all) { /* never advance the decode state on a timed-out hand-off */
                    if (!C::TP || S.seq == 0) { /* TP: every rank knows the id (its next embedding row); rank 0 keeps the books */
                        int32_t* st = a.d_state + Sb.sq * 4;
                        const int p = st[1];
                        if (a.d_tokens_out) a.d_tokens_out[(size_t)Sb.sq * a.tokens_stride + p] = bi;
                        st[0] = bi, st[1] = p + 1;
                    }
                    if (more_steps) *reinterpret_cast<u32x2*>(loc + C::tokg) = u32x2{(uint32_t)bi, (uint32_t)(epoch + 1)};
                } else if (err == 0) {
                    atomicOr(a.ws + 1, 16);
                }
            }
        }
    }
}

template <class C>
__global__ void __launch_bounds__(C::NWV * 64, (C::NWV * C::WPC + 3) / 4 /* waves per SIMD: the register budget that lets WPC workgroups share a CU */) xengine_kernel(const XArgs a) {
    constexpr int hd = C::HD, GQ = C::GQW, NWV = C::NWV, NB = C::NB;
    using P1 = typename C::SH::P1;
    using LY = XLay<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- LDS carve (compile-time offsets: XLay; the layer table behind them)
    XLds L;
    static_assert(NWV <= 16, "wmax scratch");
    EngLayer* lay = reinterpret_cast<EngLayer*>(smem + LY::fixed_bytes);
    L.lay = lay;
    L.xs[0] = reinterpret_cast<u32x4*>(smem);
    L.xs[1] = reinterpret_cast<u32x4*>(smem + LY::xs_bytes);
    L.xrawA = reinterpret_cast<uint16_t*>(smem + 2 * LY::xs_bytes);
    L.xrawB = reinterpret_cast<uint16_t*>(smem + 2 * LY::xs_bytes + LY::xr_bytes);
    L.qraw = reinterpret_cast<uint16_t*>(smem + LY::o_attn);
    L.kraw = L.qraw + GQ * hd, L.vraw = L.kraw + hd, L.qb = L.vraw + hd, L.knew = L.qb + GQ * hd;
    L.outb = reinterpret_cast<uint32_t*>(smem + LY::o_outb);
    L.wmax = reinterpret_cast<float*>(smem + LY::o_wmax);
    L.comb = reinterpret_cast<double*>(smem + (C::COMB_IN_XS1 ? (size_t)LY::xs_bytes : LY::o_comb));
    L.msc = reinterpret_cast<double*>(smem + LY::o_msc);
    L.cnt = reinterpret_cast<int*>(smem + LY::o_cnt);
    L.pub = L.cnt + 8;
    L.qtab = reinterpret_cast<const u32x4*>(smem + LY::o_tab);
    if constexpr (C::FMT == FMT_Q1T) { /* selector table of BlockDot<FMT_Q1T> (kf_gemv.hip / kf_engine.hip fill the same): entry B, dword p = bytes {2a, 2a+1, 2b, 2b+1}, a / b = bits 7-2p / 6-2p of B */
        if (tid < 256) {
            uint32_t e[4];
#pragma unroll
            for (int p = 0; p < 4; p++) e[p] = 0x01000100u + 0x0202u * ((tid >> (7 - 2 * p)) & 1u) + 0x02020000u * ((tid >> (6 - 2 * p)) & 1u);
            reinterpret_cast<u32x4*>(smem + LY::o_tab)[tid] = u32x4{e[0], e[1], e[2], e[3]};
        }
    }
    if constexpr (C::FMT == FMT_Q2T) { /* selector table of BlockDot<FMT_Q2T>: entry B, dword p = bytes {2q, 2q+1, 2q', 2q'+1}, q / q' = the levels of elements 2p, 2p+1 of byte B */
        if (tid < 256) {
            uint32_t e[2];
#pragma unroll
            for (int p = 0; p < 2; p++) e[p] = 0x01000100u + 0x0202u * ((tid >> (6 - 4 * p)) & 3u) + 0x02020000u * ((tid >> (4 - 4 * p)) & 3u);
            reinterpret_cast<u32x2*>(smem + LY::o_tab)[tid] = u32x2{e[0], e[1]};
        }
    }
    if (tid == 0) *L.cnt = 0;
    if (tid < 4) L.pub[tid] = 0;
    if (a.ws[1] != 0) return; /* an earlier launch timed out: nothing runs until the host has cleared the word (xengine_reset) */
    // ---- which decoder, which place in it: the XCD from the hardware register, a ticket there; with two decoders per XCD the first 32 tickets (the workgroups the
    // dispatcher placed first: one per CU) are decoder 0, the next 32 decoder 1 -- sequences x and x + 8.  NB > 1: ONE decoder per XCD whose sequences are x + 8 b.
    XSeq SS[NB];
    XSeq& S = SS[0];
    int xcc;
    {
        int* xi = L.cnt + 1;
        if (tid == 0) {
            const int x = eng_xcc();
            xi[0] = x, xi[1] = __hip_atomic_fetch_add(a.ws + 16 + x * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        xcc = xi[0];
        S.seq = xcc + XE_NXCD * ((xi[1] / XE_NWG) % C::WPC), S.r = xi[1] & (XE_NWG - 1);
        if (xi[1] >= XE_NWG * C::WPC && tid == 0) atomicOr(a.ws + 1, 8); /* not 32 (64) workgroups on this XCD: the polls time out, the word says why */
    }
    auto leave = [&]() { /* the last workgroup of the XCD to leave zeroes its ticket counter for the next launch */
        if (tid == 0) {
            const int d = __hip_atomic_fetch_add(a.ws + 17 + xcc * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == XE_NWG * C::WPC - 1) {
                __hip_atomic_store(a.ws + 16 + xcc * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.ws + 17 + xcc * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    const int epoch0 = a.epoch0; /* the generation of this launch's first step: counted by the host (xengine_steps), never written by the kernel -- a decoder whose sequences are
                                    all parked leaves at once, and no workgroup may find the word already moved on */
    if (C::WPC == 2 && S.seq >= XE_NXCD && a.stagger_us > 0) { /* the second decoder of an XCD starts late: the two then stand in different phases, and one's hand-off waits and K / V streaming run
                                                                  under the other's mat-vec arithmetic (started together they stay in lockstep: same phase, same wait, no overlap) */
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime() + (unsigned long long)a.stagger_us * 100ull;
        while (__builtin_amdgcn_s_memrealtime() < t1) __builtin_amdgcn_s_sleep(8);
    }
    // ---- the decoder's sequences: which exist, are not parked ({token, pos, parked, status}: d_state[s][2] != 0) and stand inside their cache rows for every step of this launch.
    // A sequence that does not is INACTIVE for the launch -- it touches nothing outside its own exchange area and says why in its own status word (d_state[s][3]: 64 = a
    // position of the launch lies beyond the cache; ADVICE r05: one finished sequence must not stop the others) -- and a decoder without an active sequence leaves.
    const int nst = a.n_steps > 1 ? a.n_steps : 1;
    int pos0[NB];
    bool any_act = false;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        XSeq& Sb = SS[b];
        Sb = S;
        Sb.seq = S.seq + XE_NXCD * b;
        Sb.sq = C::TP ? 0 : Sb.seq;
        Sb.act = C::TP ? (Sb.seq < a.n_seq) : false;
        pos0[b] = 0;
        if (C::TP || Sb.seq < a.n_seq) {
            const int p = a.d_state[Sb.sq * 4 + 1]; /* (TP: rank 0 moves the state on at the end of a step -- which no workgroup reaches before every workgroup has passed here) */
            pos0[b] = p;
            if constexpr (C::TP) {
                if (p < 0 || p + nst > a.max_seq) { /* a position of this launch lies beyond the cache rows: refuse, loudly */
                    if (tid == 0) atomicOr(a.ws + 1, 64);
                    leave();
                    return;
                }
            } else {
                const bool parked = a.d_state[Sb.sq * 4 + 2] != 0;
                const bool inside = p >= 0 && p + nst <= a.max_seq;
                Sb.act = !parked && inside;
                if (!parked && !inside && S.r == 0 && tid == 0) atomicOr(a.d_state + Sb.sq * 4 + 3, 64);
            }
        }
        Sb.kv_off = C::TP ? 0 : (long long)Sb.seq * a.kv_seq_stride; /* TP: the rank's cache rows are in its layer table */
        any_act = any_act || Sb.act;
    }
    if (!any_act) { /* a decoder without a sequence: its workgroups leave */
        leave();
        return;
    }
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.layers + (C::TP ? (size_t)S.seq * a.n_layer : 0)); /* TP: this rank's table */
        uint32_t* dst = reinterpret_cast<uint32_t*>(lay);
        const int nw = a.n_layer * (int)(sizeof(EngLayer) / 4);
        for (int i = tid; i < nw; i += NWV * 64) dst[i] = src[i];
    }
    __syncthreads();
    {
        const int s1_abs = S.r * P1::spg;
        const int j1 = s1_abs >= P1::S2 ? 2 : (s1_abs >= P1::S1 ? 1 : 0);
        const int s1 = s1_abs - (j1 == 0 ? 0 : (j1 == 1 ? P1::S1 : P1::S2));
        const int per_kv = C::SPK * C::NG, idx = S.r % per_kv, grp = idx / C::SPK;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            XSeq& Sb = SS[b];
            Sb.j1 = j1, Sb.s1 = s1;
            Sb.M1 = j1 == 0 ? P1::M0 : (j1 == 1 ? P1::M1 : P1::M2);
            Sb.q_out0 = (j1 == 0 ? 0 : (j1 == 1 ? C::QD : C::QD + C::KVD)) + s1 * P1::RPS;
            Sb.kvh = S.r / per_kv, Sb.split = idx - grp * C::SPK, Sb.h0 = Sb.kvh * C::GQ + grp * C::GQW, Sb.me0 = S.r * C::ME;
            Sb.grp0 = grp == 0;
        }
    }
    XRing<C::DEPTH> R;
    for (int step = 0; step < nst; step++) {
        const int epoch = epoch0 + step;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            XSeq& Sb = SS[b];
            Sb.step = step, Sb.pos = pos0[b] + step, Sb.len = Sb.pos + 1;
            Sb.stamp = C::DBG && a.dbg && Sb.seq == a.dbg_seq && Sb.r == a.dbg_wg && step < a.dbg_steps;
            const int chunk = (((Sb.len + C::SPK - 1) / C::SPK) + 63) & ~63; /* keys per slice: the context cut into SPK pieces, whole 64-key runs */
            Sb.t0 = Sb.split * chunk;
            Sb.t1 = Sb.t0 + chunk < Sb.len ? Sb.t0 + chunk : Sb.len;
            Sb.empty = Sb.t0 >= Sb.len || !Sb.act;
            Sb.own_new = Sb.act && Sb.pos >= Sb.t0 && Sb.pos < Sb.t1;
            if (!Sb.act) Sb.pos = 0, Sb.len = 1, Sb.t0 = 0, Sb.t1 = 0;
        }
        if (step > 0) { /* a step behind a timed-out one does not start */
            __syncthreads();
            if (tid == 0) L.cnt[3] = __hip_atomic_load(a.ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (L.cnt[3] != 0) break;
        }
        if (wave >= C::NCW) { /* poller wave - NCW */
            if constexpr (C::NP == 1) {
                xe_poller_main<C, 0>(a, L, SS, epoch, lane);
            } else if constexpr (C::NP == 2) {
                if (wave == C::NCW) xe_poller_main<C, 0>(a, L, SS, epoch, lane);
                else xe_poller_main<C, 1>(a, L, SS, epoch, lane);
            } else {
                static_assert(C::NP == 4, "one, two or four pollers");
                if (wave == C::NCW) xe_poller_main<C, 0>(a, L, SS, epoch, lane);
                else if (wave == C::NCW + 1) xe_poller_main<C, 1>(a, L, SS, epoch, lane);
                else if (wave == C::NCW + 2) xe_poller_main<C, 2>(a, L, SS, epoch, lane);
                else xe_poller_main<C, 3>(a, L, SS, epoch, lane);
            }
        } else {
            xe_compute_main<C>(a, L, SS, epoch, wave, lane, R);
        }
        if (a.head_w) xe_head_main<C>(a, L, SS, epoch, step + 1 < nst, wave, lane);
    }
    leave();
}

// ------------------------------------------------------------------------------------------------ host side: what every translation unit that instantiates the kernel needs
struct XEngineHost {
    XArgs args;
    int shape_class, fmt;
    int dim, q_dim, kv_dim, ffn, n_head, n_kv, hd;
    size_t smem, loc_stride;
    void* ws;
    size_t ws_bytes;
    int nwv, depth; /* the instantiation in use */
    int deal_wl;    /* 0: the form's default (xengine_go) */
    size_t tp_bytes; /* TP: bytes of the receive + pick areas behind the exchange areas (reset with them) */
    int epoch;       /* the generation the next launch starts at (XArgs::epoch0) */
    int batch;       /* sequences per decoder of the form in use (XCfg::NB): 1, 2 or 4 */
    int variant_set; /* xengine_set_variant was called with a waves x depth pair (tuning runs) */
    int two_wpc;     /* n_seq 9 .. 16 through the round-5 form (two decoders per XCD, two workgroups per CU) instead of the batched one: A/B hook */
};

template <class C>
static size_t xe_smem(int n_layer) {
    return XLay<C>::fixed_bytes + (((size_t)n_layer * sizeof(EngLayer) + 15) & ~(size_t)15);
}
template <class C>
static int xengine_go(XEngineHost* E, hipStream_t st) {
    static unsigned long long ready = 0; /* bit d: the attribute is set on device d (it is per device: ADVICE r05) */
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return KF_HIP_CHECK;
    if (dev < 0 || dev >= 64 || !((ready >> dev) & 1ull)) {
        if (hipFuncSetAttribute((const void*)xengine_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return KF_HIP_CHECK;
        if (dev >= 0 && dev < 64) ready |= 1ull << dev;
    }
    size_t smem = xe_smem<C>(E->args.n_layer);
    if (C::WPC == 2 && smem < 54 * 1024) smem = 54 * 1024; /* two workgroups per CU, never three: a third would be a workgroup of some decoder queued behind its own peers */
    if (smem * C::WPC > 160 * 1024) return KF_UNSUPPORTED_DATATYPE;
    E->args.deal_wl = E->deal_wl > 0 ? E->deal_wl : (C::WPC > 1 ? 14 : (C::NCW == 7 ? 11 : 16)); /* xe_deal: the share of the compute wave beside the poller (two decoders per XCD) */
    hipLaunchKernelGGL((xengine_kernel<C>), dim3(XE_GRID * C::WPC), dim3(C::NWV * 64), smem, st, E->args);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
