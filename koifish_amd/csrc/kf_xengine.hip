// kf_xengine.hip -- host side of the XCD-confined decode engines (the kernel: kf_xengine_kernel.h): shape classes, workspaces, the layer tables, the instantiations of the
// 4-bit forms and their dispatch.  (kf_xengine_q1.hip: the 1-bit forms.)
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kf_xengine_kernel.h"

namespace kf {

// ------------------------------------------------------------------------------------------------ host side

// kf_xengine_q1.hip: the 1-bit forms (their own translation unit: the instantiations compile side by side)
int xengine_go_lowbit(XEngineHost* E, hipStream_t st);
size_t xe_smem_lowbit(int fmt, int shape_class, int n_seq, int n_layer);
static int xe_shape_class(int GQ, int hd, int dim, int q_dim, int ffn) {
    if (GQ == 2 && hd == 128 && dim == 1024 && q_dim == 2048 && ffn == 3072) return 1; /* Qwen3-0.6B (BASELINE config 2) */
    if (GQ == 2 && hd == 64 && dim == 256 && q_dim == 256 && ffn == 512) return 2;     /* the small parity-test shape */
    if (GQ == 2 && hd == 128 && dim == 2048 && q_dim == 2048 && ffn == 6144) return 3;  /* Qwen3-1.7B: the streaming phases do not care how many blocks a lane walks */
    if (GQ == 4 && hd == 128 && dim == 2560 && q_dim == 4096 && ffn == 9728) return 4;  /* Qwen3-4B (32 / 8 heads; cases/tutorial/history.md:4-6) */
    if (GQ == 4 && hd == 128 && dim == 4096 && q_dim == 4096 && ffn == 12288) return 5; /* Qwen3-8B */
    if (GQ == 8 && hd == 64 && dim == 256 && q_dim == 512 && ffn == 512) return 6;      /* parity-test shape: 8 query heads on ONE kv-head (two head groups), 20 workgroups with q | k | v rows */
    return 0;
}
template <int NWV, int DEPTH, bool DBG, int WPC, int AU = 2, int NB = 1>
using XC3 = XCfg<FMT_Q4P, 2, 128, NWV, 2048, 2048, 1024, 6144, DEPTH, DBG, WPC, AU, false, NB>;
template <int NWV, int DEPTH, bool DBG, int WPC, int AU = 2, int NB = 1, int NP = 0>
using XC1 = XCfg<FMT_Q4P, 2, 128, NWV, 1024, 2048, 1024, 3072, DEPTH, DBG, WPC, AU, false, NB, NP>;
template <int NWV, int DEPTH, bool DBG, int WPC, int AU = 2, int NB = 1, int NP = 0>
using XC2 = XCfg<FMT_Q4P, 2, 64, NWV, 256, 256, 128, 512, DEPTH, DBG, WPC, AU, false, NB, NP>;
// the GQA-4 shapes: one decoder per XCD, 8 waves (two per SIMD: 256 registers -- the attention sums of four query heads are 36 fp64 values per lane)
using XC4 = XCfg<FMT_Q4P, 4, 128, 8, 2560, 4096, 1024, 9728, 8, false, 1, 2>;
using XC5 = XCfg<FMT_Q4P, 4, 128, 8, 4096, 4096, 1024, 12288, 8, false, 1, 2>;
using XC6 = XCfg<FMT_Q4P, 8, 64, 12, 256, 512, 64, 512, 6, false, 1, 2>;
using XC4W = XCfg<FMT_Q4P, 4, 128, 12, 2560, 4096, 1024, 9728, 6, false, 1, 1>; /* 11 compute waves at 168 registers: the attention loop spills, the mat-vec phases (nine tenths of the bytes) have the waves */
using XC5W = XCfg<FMT_Q4P, 4, 128, 12, 4096, 4096, 1024, 12288, 6, false, 1, 1>;
static int xe_loc_dw(int shape_class) {
    switch (shape_class) {
        case 1: return XC1<9, 8, false, 1>::loc_dw;
        case 3: return XC3<9, 8, false, 1>::loc_dw;
        case 4: return XC4::loc_dw;
        case 5: return XC5::loc_dw;
        case 6: return XC6::loc_dw;
    }
    return XC2<9, 8, false, 1>::loc_dw;
}

static bool xe_class_fused(int sc); /* below the shape aliases */
static size_t xe_smem_class3_two(int n_layer);
template <template <int, int, bool, int, int, int, int> class XC>
static size_t xe_shape_smem(int n_seq, int n_layer, bool two_wpc);
// FUSED shapes (XCfg::FUSED): q | k | v of a layer as ONE matrix -- blocks, then the zero words, then the step words of the QD + 2 KVD rows -- copied once into the workspace
static size_t xe_fused_layer_bytes(const kf_engine_desc* d) {
    const size_t rows = (size_t)(d->n_head + 2 * d->n_kv) * d->head_dim, nblk = d->dim / 32, grp = d->dim / 128;
    return ((rows * nblk * 16 + 255) & ~(size_t)255) + 2 * ((rows * grp * 2 + 255) & ~(size_t)255);
}
static int xe_fuse_qkv(const kf_engine_desc* d, EngLayer* tab, const float* qbias, char*& p, hipStream_t st) {
    if (qbias[0] != qbias[1] || qbias[0] != qbias[2]) return KF_UNSUPPORTED_DATATYPE; /* one zero point for the fused rows */
    const size_t nblk = d->dim / 32, grp = d->dim / 128;
    const size_t rows_all = (size_t)(d->n_head + 2 * d->n_kv) * d->head_dim;
    for (int l = 0; l < d->n_layer; l++) {
        char* wdst = p;
        char* zdst = wdst + ((rows_all * nblk * 16 + 255) & ~(size_t)255);
        char* sdst = zdst + ((rows_all * grp * 2 + 255) & ~(size_t)255);
        size_t r0 = 0;
        for (int j = 0; j < 3; j++) {
            const kf_weight& w = d->layers[l].w[j];
            const size_t rows = (size_t)w.ne0;
            const uint16_t* zero = w.gama + w.ne0 + w.ne1;
            const uint16_t* step = zero + rows * grp;
            if (hipMemcpyAsync(wdst + r0 * nblk * 16, w.data, rows * nblk * 16, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                hipMemcpyAsync(zdst + r0 * grp * 2, zero, rows * grp * 2, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                hipMemcpyAsync(sdst + r0 * grp * 2, step, rows * grp * 2, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return KF_HIP_CHECK;
            r0 += rows;
        }
        tab[l].m[0].w = (g_u32x4)(uintptr_t)wdst, tab[l].m[0].zero = (g_u16)(uintptr_t)zdst, tab[l].m[0].step = (g_u16)(uintptr_t)sdst;
        p += xe_fused_layer_bytes(d);
    }
    return KF_OK;
}

size_t xengine_ws_bytes(const kf_engine_desc* d) {
    const int hd = d->head_dim, GQ = d->n_kv > 0 ? d->n_head / d->n_kv : 1;
    const int sc = xe_shape_class(GQ, hd, d->dim, d->n_head * hd, d->ffn);
    size_t b = 4096 + (((size_t)d->n_layer * sizeof(EngLayer) + 255) & ~(size_t)255);
    b += (size_t)XE_MAXSEQ * xe_loc_stride(sc ? xe_loc_dw(sc) : 0) + 4096;
    if (sc && xe_class_fused(sc)) b += (size_t)d->n_layer * xe_fused_layer_bytes(d) + 256;
    return b;
}
static int xengine_init_state(XEngineHost* E, hipStream_t st) {
    XArgs& a = E->args;
    if (hipMemsetAsync(a.loc, 0xff, (size_t)(E->tp_bytes ? XE_NXCD : XE_MAXSEQ) * E->loc_stride, st) != hipSuccess) return KF_HIP_CHECK;
    if (E->tp_bytes && hipMemsetAsync(a.tp_recv, 0xff, E->tp_bytes, st) != hipSuccess) return KF_HIP_CHECK;
    static int init[16 + 32 * XE_NXCD];
    memset(init, 0, sizeof(init));
    init[0] = 1; /* no error, tickets zero */
    if (hipMemcpyAsync(a.ws, init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) return KF_HIP_CHECK;
    E->epoch = 1;
    return KF_OK;
}
void xengine_free(XEngineHost* E) {
    if (!E) return;
    if (E->args.dbg) (void)hipFree(E->args.dbg);
    delete E;
}
// one model's (one TP rank's) layer table out of its descriptor; qbias: the seven matrices' zero points of layer 0 (every layer, every rank: the same)
// *fmt_io: the storage every matrix of the model carries (0 on entry = not known yet): FMT_Q4P (4-bit PackedQ) or FMT_Q1T (1-bit, round 6)
static int xe_fill_layers(const kf_engine_desc* d, EngLayer* tab, float* qbias, bool qbias_set, bool& q4p_ok, int* fmt_io = nullptr, bool allow_hot = false) {
    const int hd = d->head_dim, q_dim = d->n_head * hd, kv_dim = d->n_kv * hd;
    const int Ks[7] = {d->dim, d->dim, d->dim, q_dim, d->dim, d->dim, d->ffn}, Ms[7] = {q_dim, kv_dim, kv_dim, d->dim, d->ffn, d->ffn, d->dim};
    for (int l = 0; l < d->n_layer; l++) {
        const kf_engine_layer& Ly = d->layers[l];
        if (Ly.hot_ffn && !allow_hot) return KF_UNSUPPORTED_DATATYPE;
        if (!Ly.norm_in || !Ly.norm_post || !Ly.kcache || !Ly.vcache || (((uintptr_t)Ly.kcache | (uintptr_t)Ly.vcache) & 15) != 0) return KF_UNSUPPORTED_DATATYPE;
        for (int j = 0; j < 7; j++) {
            const kf_weight& w = Ly.w[j];
            const int wf = gemv_fmt_of(&w), ef = wf == FMT_Q4 ? FMT_Q4P : (wf == FMT_Q1 ? FMT_Q1T : (wf == FMT_Q2 ? FMT_Q2T : -1));
            if (ef < 0 || (!fmt_io && ef != FMT_Q4P) || (fmt_io && *fmt_io != 0 && *fmt_io != ef)) return KF_UNSUPPORTED_DATATYPE;
            if (fmt_io) *fmt_io = ef;
            if (w.ne0 != Ms[j] || w.ne1 != Ks[j] || w.qzeros || w.qscales || !w.gama || w.lGroup != 128 || (Ks[j] % 128) != 0 || ((uintptr_t)w.data & 15) != 0)
                return KF_UNSUPPORTED_DATATYPE;
            const long rows = j < 3 ? (long)q_dim + 2 * kv_dim : (j == 4 || j == 5 ? (long)d->ffn : (long)Ms[j]);
            if (gemv_lpr_log2(Ks[j] / 32, rows) < 2) q4p_ok = false; /* the register-table form needs a group's four blocks in one aligned lane quad (1-bit: the four dwords of a block in one) */
            tab[l].m[j].w = (g_u32x4)(uintptr_t)w.data;
            tab[l].m[j].zero = (g_u16)(uintptr_t)(w.gama + w.ne0 + w.ne1);
            tab[l].m[j].step = (g_u16)(uintptr_t)(w.gama + w.ne0 + w.ne1 + (size_t)w.ne0 * w.ne1 / w.lGroup);
            if (l == 0 && !qbias_set) qbias[j] = (float)w.qBias;
            else if (qbias[j] != (float)w.qBias) return KF_UNSUPPORTED_DATATYPE;
        }
        tab[l].norm_in = (g_u16)(uintptr_t)Ly.norm_in, tab[l].norm_post = (g_u16)(uintptr_t)Ly.norm_post;
        tab[l].norm_q = (g_u16)(uintptr_t)Ly.q_norm, tab[l].norm_k = (g_u16)(uintptr_t)Ly.k_norm;
        tab[l].kcache = (g_u16w)(uintptr_t)Ly.kcache, tab[l].vcache = (g_u16w)(uintptr_t)Ly.vcache;
        tab[l].hot = (g_i32)(uintptr_t)Ly.hot_ffn; /* sparse forward: CS_Picker's hot[ffn] on the device (kf_abi.h kf_engine_layer::hot_ffn); NULL: dense */
    }
    return KF_OK;
}
int xengine_build(const kf_engine_desc* d, int n_seq, long long kv_seq_stride, void* ws, size_t ws_bytes, hipStream_t st, XEngineHost** out, const char** why, bool dry) {
    const char* dummy;
    if (!why) why = &dummy;
    *why = "bad arguments";
    if (!d || (!dry && (!ws || !out)) || d->n_layer < 1 || !d->layers || n_seq < 1 || n_seq > XE_MAXSEQ || kv_seq_stride < 0) return KF_INVALID_ARGS;
    const int hd = d->head_dim;
    *why = "head_dim must be 64 or 128 and n_head a multiple of n_kv";
    if ((hd != 64 && hd != 128) || d->n_kv <= 0 || d->n_head % d->n_kv != 0) return KF_UNSUPPORTED_DATATYPE;
    const int GQ = d->n_head / d->n_kv, q_dim = d->n_head * hd, kv_dim = d->n_kv * hd;
    const int sc = xe_shape_class(GQ, hd, d->dim, q_dim, d->ffn);
    *why = "model shape not instantiated for the XCD-confined engine: built for Qwen3-0.6B (dim 1024, 16/8 heads of 128, ffn 3072), Qwen3-1.7B (dim 2048, same heads, ffn 6144), Qwen3-4B "
           "(dim 2560, 32/8 heads, ffn 9728), Qwen3-8B (dim 4096, 32/8 heads, ffn 12288) and the 256-wide test shape";
    if (!sc) return KF_UNSUPPORTED_DATATYPE;
    *why = "the GQA-4 / GQA-8 shapes (Qwen3-4B / 8B, the 8-on-1 test shape) run one decoder per XCD: at most 8 sequences (two workgroups per CU would need 2 x 78 KB (2 x 98 KB) of activations in LDS)";
    if (sc >= 4 && n_seq > XE_NXCD) return KF_UNSUPPORTED_DATATYPE;
    *why = "more than 16 sequences (four per decoder) are served for the Qwen3-0.6B shape and the 256-wide test shape only";
    if (sc == 3 && n_seq > 2 * XE_NXCD) return KF_UNSUPPORTED_DATATYPE;
    *why = "the model is too deep for this many sequences: the workgroup's activations of every sequence of a decoder + the layer table must fit 160 KB of LDS";
    if (sc == 1 && xe_shape_smem<XC1>(n_seq, d->n_layer, false) > 160 * 1024) return KF_UNSUPPORTED_DATATYPE;
    if (sc == 2 && xe_shape_smem<XC2>(n_seq, d->n_layer, false) > 160 * 1024) return KF_UNSUPPORTED_DATATYPE;
    if (sc == 4 && xe_smem<XC4W>(d->n_layer) > 160 * 1024) return KF_UNSUPPORTED_DATATYPE;
    if (sc == 5 && xe_smem<XC5W>(d->n_layer) > 160 * 1024) return KF_UNSUPPORTED_DATATYPE;
    if (sc == 3 && n_seq <= XE_NXCD && xe_smem<XC3<12, 6, false, 1>>(d->n_layer) > 160 * 1024) return KF_UNSUPPORTED_DATATYPE;
    *why = "two decoders per XCD (more than 8 sequences) do not fit this shape and depth: two workgroups per CU need 2 x the activations + the layer table in 160 KB of LDS";
    if (sc == 3 && n_seq > XE_NXCD && 2 * xe_smem_class3_two(d->n_layer) > 160 * 1024 && xe_smem<XC3<8, 8, false, 1, 2, 2>>(d->n_layer) > 160 * 1024) return KF_UNSUPPORTED_DATATYPE;
    if (!dry && (ws_bytes < xengine_ws_bytes(d) || ((uintptr_t)ws & 255) != 0)) {
        *why = "workspace too small or not 256-byte aligned";
        return KF_INVALID_ARGS;
    }
    int dev = 0, n_cu = 0;
    *why = "HIP failure";
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return KF_HIP_CHECK;
    *why = "the device does not show 256 compute units (8 XCDs of 32): one resident workgroup per CU, 32 per XCD, is the premise";
    if (n_cu != XE_GRID) return KF_UNSUPPORTED_DATATYPE;
    *why = "rope_table missing, kv_stride not a multiple of 8, or max_seq < 1";
    if (!d->rope_table || (d->kv_stride % 8) != 0 || d->max_seq < 1) return KF_INVALID_ARGS;
    *why = "layer storage not served: 4-bit (RTN), 2-bit or 1-bit (YinYang) PackedQ layers in groups of 128 with 16-byte aligned blocks, every matrix the same storage, every layer the same shapes (FFN dense or with a hot-row mask)";
    std::vector<EngLayer> tab(d->n_layer);
    float qbias[7] = {0};
    bool q4p_ok = true;
    int fmt = 0;
    if (xe_fill_layers(d, tab.data(), qbias, false, q4p_ok, &fmt, true) != KF_OK) return KF_UNSUPPORTED_DATATYPE;
    if (!q4p_ok) return KF_UNSUPPORTED_DATATYPE;
    *why = "1-bit / 2-bit PackedQ layers are served for the Qwen3-0.6B shape and the 256-wide test shape";
    if (fmt != FMT_Q4P && sc != 1 && sc != 2) return KF_UNSUPPORTED_DATATYPE;
    *why = "the model is too deep for this many sequences: the workgroup's activations of every sequence of a decoder + the layer table must fit 160 KB of LDS";
    if (fmt != FMT_Q4P && xe_smem_lowbit(fmt, sc, n_seq, d->n_layer) > 160 * 1024) return KF_UNSUPPORTED_DATATYPE;
    if (dry) {
        *why = "";
        return KF_OK;
    }
    XEngineHost* E = new XEngineHost();
    memset(E, 0, sizeof(*E));
    XArgs& a = E->args;
    E->shape_class = sc, E->fmt = fmt, E->dim = d->dim, E->q_dim = q_dim, E->kv_dim = kv_dim, E->ffn = d->ffn, E->n_head = d->n_head, E->n_kv = d->n_kv, E->hd = hd;
    E->nwv = 12, E->depth = 6; /* 11 compute waves + the poller: three waves per SIMD (measured best: 1.89 ms per step of eight sequences against 1.92 with 9 x 8; 1.82 since the head rows are read with plain loads) */
    a.n_layer = d->n_layer, a.n_seq = n_seq, a.kv_seq_stride = kv_seq_stride, a.kv_stride = d->kv_stride, a.max_seq = d->max_seq;
    a.eps = d->rms_eps, a.qk_eps = d->qk_eps, a.rope_table = d->rope_table;
    for (int j = 0; j < 7; j++) a.qbias[j] = qbias[j];
    char* p = reinterpret_cast<char*>(ws);
    E->ws = ws, E->ws_bytes = ws_bytes;
    a.ws = reinterpret_cast<int*>(p), p += 4096;
    a.layers = reinterpret_cast<const EngLayer*>(p), p += ((size_t)d->n_layer * sizeof(EngLayer) + 255) & ~(size_t)255;
    p = reinterpret_cast<char*>(((uintptr_t)p + 4095) & ~(uintptr_t)4095);
    E->loc_stride = xe_loc_stride(xe_loc_dw(sc));
    a.loc = p, a.loc_stride = E->loc_stride;
    if (xe_class_fused(sc)) {
        char* fp = reinterpret_cast<char*>(((uintptr_t)(p + (size_t)XE_MAXSEQ * E->loc_stride) + 255) & ~(uintptr_t)255);
        const int frc = xe_fuse_qkv(d, tab.data(), qbias, fp, st);
        if (frc != KF_OK) {
            xengine_free(E);
            *why = "q / k / v carry different zero points (the fused copy of the three takes one), or a HIP failure while copying";
            return frc;
        }
    }
    if (xengine_init_state(E, st) != KF_OK || hipMemcpyAsync(const_cast<EngLayer*>(a.layers), tab.data(), tab.size() * sizeof(EngLayer), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        xengine_free(E);
        *why = "HIP failure while initialising the workspace";
        return KF_HIP_CHECK;
    }
    *out = E;
    *why = "";
    return KF_OK;
}
// ---- tensor parallel over the XCDs: ONE sequence of a model too wide for one XCD's share to be a decoder of its own -- eight ranks = eight XCDs (XCfg::TP)
// a rank of Qwen3-32B under TP = 8: 8 query heads on 1 kv-head, q_dim 1024, ffn 3200 (koifish_amd/tp.py TPPlan)
using XC7 = XCfg<FMT_Q4P, 8, 128, 12, 5120, 1024, 128, 3200, 6, false, 1, 1, true>;
using XC7D = XCfg<FMT_Q4P, 8, 128, 12, 5120, 1024, 128, 3200, 6, true, 1, 1, true>; /* + the per-phase stamps of one workgroup of one rank */
static size_t xe_smem_class3_two(int n_layer) { return xe_smem<XC3<8, 4, false, 2, 1>>(n_layer); }
// a rank of Qwen3-8B under TP = 8 (round 6: ONE sequence of a GQA-4 model on the eight XCDs): 4 query heads on 1 kv-head, q_dim 512, ffn 1536 (12 groups of 128)
using XC8 = XCfg<FMT_Q4P, 4, 128, 12, 4096, 512, 128, 1536, 6, false, 1, 1, true>;
// a rank of Qwen3-4B under TP = 8: ffn 9728 = 76 groups of 128 does not split into eight whole-group column shards -- the model is run with its FFN padded to 80 groups (512
// zero rows of gate / up, 512 zero columns of down_proj: exact zeros in every sum), 1280 per rank
using XC9 = XCfg<FMT_Q4P, 4, 128, 12, 2560, 512, 128, 1280, 2, false, 1, 1, true>; /* (ring depth, ms per token at 2 k keys: 6 2.33, 4 2.25, 2 2.22; 8 waves 2.42 - 2.48, 16 waves 2.87) */
static_assert(XC7::FUSED && XC8::FUSED && XC9::FUSED, "the TP forms multiply q | k | v as one fused matrix");
static bool xe_class_fused(int sc) { return sc == 4 ? XC4::FUSED : (sc == 5 ? XC5::FUSED : (sc == 6 ? XC6::FUSED : (sc == 7 ? XC7::FUSED : ((sc == 8 || sc == 9) ? true : false)))); }
static bool xe_class_tp(int sc) { return sc >= 7 && sc <= 9; }
static int xe_tp_shape(const kf_engine_desc* d) { /* the TP shape class of a rank's card, 0: not instantiated */
    if (d->head_dim == 128 && d->n_head == 8 && d->n_kv == 1 && d->dim == 5120 && d->ffn == 3200) return 7;
    if (d->head_dim == 128 && d->n_head == 4 && d->n_kv == 1 && d->dim == 4096 && d->ffn == 1536) return 8;
    if (d->head_dim == 128 && d->n_head == 4 && d->n_kv == 1 && d->dim == 2560 && d->ffn == 1280) return 9;
    return 0;
}
static int xe_tp_loc_dw(int sc) { return sc == 8 ? XC8::loc_dw : (sc == 9 ? XC9::loc_dw : XC7::loc_dw); }
static size_t xe_tp_recv_granules(int dim) { return (size_t)XE_NXCD * 2 * XE_NXCD * dim; }
size_t xengine_ws_bytes_tp(const kf_engine_desc* d0) {
    const int sc = xe_tp_shape(d0);
    size_t b = 4096 + (((size_t)XE_NXCD * d0->n_layer * sizeof(EngLayer) + 255) & ~(size_t)255);
    b += (size_t)XE_NXCD * xe_loc_stride(xe_tp_loc_dw(sc)) + 4096;
    b += xe_tp_recv_granules(d0->dim) * 8 + (size_t)XE_NXCD * XE_NXCD * 8 + 4096;
    b += (size_t)XE_NXCD * d0->n_layer * xe_fused_layer_bytes(d0) + 256;
    return b;
}
int xengine_build_tp(const kf_engine_desc* const* ds, int world, void* ws, size_t ws_bytes, hipStream_t st, XEngineHost** out, const char** why) {
    const char* dummy;
    if (!why) why = &dummy;
    *why = "bad arguments";
    if (!ds || !ws || !out || world < 1) return KF_INVALID_ARGS;
    *why = "tensor parallel over the XCDs: exactly 8 ranks (one per XCD)";
    if (world != XE_NXCD) return KF_UNSUPPORTED_DATATYPE;
    for (int r = 0; r < world; r++)
        if (!ds[r] || !ds[r]->layers || ds[r]->n_layer < 1 || ds[r]->n_layer != ds[0]->n_layer) return KF_INVALID_ARGS;
    *why = "rank shape not instantiated: built for the TP = 8 ranks of Qwen3-32B (dim 5120, 8 / 1 heads of 128, ffn 3200 per rank), Qwen3-8B (dim 4096, 4 / 1 heads, ffn 1536) and Qwen3-4B with "
           "its FFN padded to 10240 (dim 2560, 4 / 1 heads, ffn 1280)";
    const int sc = xe_tp_shape(ds[0]);
    for (int r = 0; r < world; r++)
        if (!sc || xe_tp_shape(ds[r]) != sc) return KF_UNSUPPORTED_DATATYPE;
    const kf_engine_desc* d = ds[0];
    if (ws_bytes < xengine_ws_bytes_tp(d) || ((uintptr_t)ws & 255) != 0) {
        *why = "workspace too small or not 256-byte aligned";
        return KF_INVALID_ARGS;
    }
    int dev = 0, n_cu = 0;
    *why = "HIP failure";
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return KF_HIP_CHECK;
    *why = "the device does not show 256 compute units (8 XCDs of 32): one resident workgroup per CU, 32 per XCD, is the premise";
    if (n_cu != XE_GRID) return KF_UNSUPPORTED_DATATYPE;
    *why = "rope_table missing, kv_stride not a multiple of 8, max_seq < 1, or the ranks disagree";
    for (int r = 0; r < world; r++)
        if (!ds[r]->rope_table || (ds[r]->kv_stride % 8) != 0 || ds[r]->max_seq < 1 || ds[r]->max_seq != d->max_seq || ds[r]->kv_stride != d->kv_stride || ds[r]->rms_eps != d->rms_eps ||
            ds[r]->qk_eps != d->qk_eps)
            return KF_INVALID_ARGS;
    *why = "layer storage not served: 4-bit PackedQ (RTN) layers in groups of 128 with 16-byte aligned blocks, dense FFN, every layer and rank the same shapes";
    std::vector<EngLayer> tab((size_t)world * d->n_layer);
    float qbias[7] = {0};
    bool q4p_ok = true;
    for (int r = 0; r < world; r++)
        if (xe_fill_layers(ds[r], tab.data() + (size_t)r * d->n_layer, qbias, r > 0, q4p_ok) != KF_OK) return KF_UNSUPPORTED_DATATYPE;
    if (!q4p_ok) return KF_UNSUPPORTED_DATATYPE;
    XEngineHost* E = new XEngineHost();
    memset(E, 0, sizeof(*E));
    XArgs& a = E->args;
    E->shape_class = sc, E->fmt = FMT_Q4P, E->dim = d->dim, E->q_dim = d->n_head * d->head_dim, E->kv_dim = d->n_kv * d->head_dim, E->ffn = d->ffn, E->n_head = d->n_head, E->n_kv = d->n_kv, E->hd = d->head_dim;
    E->nwv = 12, E->depth = 6;
    a.n_layer = d->n_layer, a.n_seq = XE_NXCD /* decoders = ranks */, a.kv_seq_stride = 0, a.kv_stride = d->kv_stride, a.max_seq = d->max_seq;
    a.eps = d->rms_eps, a.qk_eps = d->qk_eps, a.rope_table = d->rope_table;
    for (int j = 0; j < 7; j++) a.qbias[j] = qbias[j];
    char* p = reinterpret_cast<char*>(ws);
    E->ws = ws, E->ws_bytes = ws_bytes;
    a.ws = reinterpret_cast<int*>(p), p += 4096;
    a.layers = reinterpret_cast<const EngLayer*>(p), p += (tab.size() * sizeof(EngLayer) + 255) & ~(size_t)255;
    p = reinterpret_cast<char*>(((uintptr_t)p + 4095) & ~(uintptr_t)4095);
    E->loc_stride = xe_loc_stride(xe_tp_loc_dw(sc));
    a.loc = p, a.loc_stride = E->loc_stride, p += (size_t)XE_NXCD * E->loc_stride;
    a.tp_recv = reinterpret_cast<unsigned long long*>(p), p += xe_tp_recv_granules(d->dim) * 8;
    a.tp_best = reinterpret_cast<unsigned long long*>(p);
    E->tp_bytes = xe_tp_recv_granules(d->dim) * 8 + (size_t)XE_NXCD * XE_NXCD * 8;
    {
        char* fp = reinterpret_cast<char*>(((uintptr_t)(p + (size_t)XE_NXCD * XE_NXCD * 8) + 255) & ~(uintptr_t)255);
        for (int r = 0; r < world; r++) {
            const int frc = xe_fuse_qkv(ds[r], tab.data() + (size_t)r * d->n_layer, qbias, fp, st);
            if (frc != KF_OK) {
                xengine_free(E);
                *why = "q / k / v carry different zero points (the fused copy of the three takes one), or a HIP failure while copying";
                return frc;
            }
        }
    }
    if (xengine_init_state(E, st) != KF_OK || hipMemcpyAsync(const_cast<EngLayer*>(a.layers), tab.data(), tab.size() * sizeof(EngLayer), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        xengine_free(E);
        *why = "HIP failure while initialising the workspace";
        return KF_HIP_CHECK;
    }
    *out = E;
    *why = "";
    return KF_OK;
}
// the head in vocabulary shards (rank r: rows row0[r] .. of the full matrix), logits: the full vector, the shards in rank order
int xengine_set_head_tp(XEngineHost* E, const kf_weight* const* ws, const int* row0, const uint16_t* norm_w, uint16_t* logits, int32_t* d_tokens_out, int tokens_stride) {
    XArgs& a = E->args;
    if (!E->tp_bytes || !ws || !row0 || !norm_w || !logits) return KF_INVALID_ARGS;
    const int nBlk = E->dim / 8;
    long at = 0;
    for (int r = 0; r < XE_NXCD; r++) {
        const kf_weight* w = ws[r];
        if (!w || w->type != KF_BF16 || w->quant != KF_QUANT_GROUP || w->qzeros || w->ne1 != E->dim || !w->data || ((uintptr_t)w->data & 15) != 0 || w->ne0 < 64 || row0[r] != at) return KF_UNSUPPORTED_DATATYPE;
        if (gemv_lpr_log2(nBlk, w->ne0) != c_lpr_log2(nBlk, 1L << 20)) return KF_UNSUPPORTED_DATATYPE; /* same lanes per row as the mat-vec launcher: same summation order */
        a.head_w_r[r] = (g_u32x4)(uintptr_t)w->data, a.logits_r[r] = logits + at, a.vocab_r[r] = w->ne0, a.row0_r[r] = row0[r];
        at += w->ne0;
    }
    a.head_w = a.head_w_r[0], a.vocab = a.vocab_r[0], a.logits = logits; /* (head_w: "a head is set") */
    a.head_norm = (g_u16)(uintptr_t)norm_w, a.d_tokens_out = d_tokens_out, a.tokens_stride = tokens_stride;
    return KF_OK;
}
#ifndef XE_VARIANTS
#define XE_VARIANTS 1 /* the tuning instantiations (waves per workgroup x ring depth) beside the defaults.  Measured and dropped (ms per step of eight sequences at 2 k keys, default
                         12 x 6: 1.89): 13 x 6 2.22, 16 x 4 2.17 (128 registers: spills), four key tiles per attention batch 2.29 - 2.37 (spills) */
#endif
// n_seq <= 8: one decoder per XCD (12 waves, 168 registers), one sequence each.  More: the BATCHED form (round 6) -- still one decoder per XCD, every unpacked block multiplied
// against the activations of 2 (n_seq <= 16) or 4 (n_seq <= 32) sequences.  The round-5 form of 9 .. 16 sequences (two decoders per XCD, two workgroups of 8 waves per CU,
// 128 registers) stays behind XEngineHost::two_wpc as the A/B reference.
template <template <int, int, bool, int, int, int, int> class XC, int NWV, int DEPTH, bool DBG, int WPC, int AU, int NB, int NP = 0>
using XF = XC<NWV, DEPTH, DBG, WPC, AU, NB, NP>; /* (a template template parameter carries no default arguments) */
static int xe_batch_of(const XEngineHost* E) { return E->args.n_seq <= XE_NXCD ? 1 : ((E->args.n_seq <= 2 * XE_NXCD) ? (E->two_wpc ? 1 : 2) : 4); }
template <template <int, int, bool, int, int, int, int> class XC>
static int xengine_go_shape(XEngineHost* E, hipStream_t st) {
    const int nb = xe_batch_of(E);
    const bool two = E->args.n_seq > XE_NXCD && nb == 1;
#ifndef XE_ONLY_DEFAULT
    const bool dbg = E->args.dbg != nullptr;
    if (dbg && nb == 4) return xengine_go<XF<XC, 12, 2, true, 1, 2, 4>>(E, st);
    if (dbg && nb == 2) return xengine_go<XF<XC, 12, 4, true, 1, 2, 2>>(E, st);
    if (dbg && nb == 1) return two ? xengine_go<XF<XC, 8, 4, true, 2, 1, 1>>(E, st) : xengine_go<XF<XC, 12, 2, true, 1, 2, 1>>(E, st);
#endif
    if (nb == 4) { /* 8 compute waves + the four sequences' pollers (168 registers), where the four sequences' activations + the layer table fit the LDS; else 4 + 4 waves */
#ifdef XE_NB4_VARIANTS /* tuning builds only */
        if (E->variant_set && E->nwv == 12 && E->depth == 8) return xengine_go<XF<XC, 12, 8, false, 1, 2, 4>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 4) return xengine_go<XF<XC, 12, 4, false, 1, 2, 4>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 61) return xengine_go<XF<XC, 12, 6, false, 1, 1, 4>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 41) return xengine_go<XF<XC, 12, 4, false, 1, 1, 4>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 21) return xengine_go<XF<XC, 12, 2, false, 1, 1, 4>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 6) return xengine_go<XF<XC, 12, 6, false, 1, 2, 4>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 22) return xengine_go<XF<XC, 12, 2, false, 1, 2, 4, 2>>(E, st); /* two pollers for the four sequences */
        if (E->variant_set && E->nwv == 12 && E->depth == 12) return xengine_go<XF<XC, 12, 2, false, 1, 2, 4, 1>>(E, st); /* one */
        if (E->variant_set && E->nwv == 12 && E->depth == 44) return xengine_go<XF<XC, 12, 4, false, 1, 2, 4, 4>>(E, st);
#endif
        // ring depth 2 (measured, 32 sequences at 2 k keys, tokens/s: depth 8 5320 -- spills --, 6 6970, 4 7360, 2 7815; one key tile per attention batch 7480 / 7720 at depth 6 / 2): with
        // four sequences' chain pairs and activation chunks live, every register the ring does not hold is worth more than a deeper queue -- twelve waves hide the latency
        if (!(E->variant_set && E->nwv == 8) && xe_smem<XF<XC, 12, 2, false, 1, 2, 4>>(E->args.n_layer) <= 160 * 1024) return xengine_go<XF<XC, 12, 2, false, 1, 2, 4>>(E, st);
        return xengine_go<XF<XC, 8, 8, false, 1, 2, 4>>(E, st);
    }
    if (nb == 2) {
#ifdef XE_NB4_VARIANTS /* tuning builds only */
        if (E->variant_set && E->nwv == 12 && E->depth == 61) return xengine_go<XF<XC, 12, 6, false, 1, 1, 2>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 6) return xengine_go<XF<XC, 12, 6, false, 1, 2, 2>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 41) return xengine_go<XF<XC, 12, 4, false, 1, 1, 2>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 2) return xengine_go<XF<XC, 12, 2, false, 1, 2, 2>>(E, st);
        if (E->variant_set && E->nwv == 12 && E->depth == 14) return xengine_go<XF<XC, 12, 4, false, 1, 2, 2, 1>>(E, st); /* one poller for the two sequences */
        if (E->variant_set && E->nwv == 12 && E->depth == 12) return xengine_go<XF<XC, 12, 2, false, 1, 2, 2, 1>>(E, st);
#endif
        if (E->variant_set && E->nwv == 8) return xengine_go<XF<XC, 8, 8, false, 1, 2, 2>>(E, st);
        return xengine_go<XF<XC, 12, 4, false, 1, 2, 2>>(E, st); /* (16 sequences: depth 6 5990, 4 6250, 2 6200 tokens/s) */
    }
#ifdef XE_NB4_VARIANTS
    if (!two && E->variant_set && E->nwv == 12 && E->depth == 4) return xengine_go<XF<XC, 12, 4, false, 1, 2, 1>>(E, st);
#endif
    if (!two && E->variant_set && E->nwv == 12 && E->depth == 6) return xengine_go<XF<XC, 12, 6, false, 1, 2, 1>>(E, st);
    if (!two && E->variant_set && E->nwv == 9) return xengine_go<XF<XC, 9, 8, false, 1, 2, 1>>(E, st);
    return two ? xengine_go<XF<XC, 8, 4, false, 2, 1, 1>>(E, st) : xengine_go<XF<XC, 12, 2, false, 1, 2, 1>>(E, st); /* (8 sequences: depth 6 4370, 4 4570, 2 4630 tokens/s) */
}
// the LDS the chosen form needs (classes 1 and 2), so that create / served can refuse a model too deep for it instead of the first step
template <template <int, int, bool, int, int, int, int> class XC>
static size_t xe_shape_smem(int n_seq, int n_layer, bool two_wpc) {
    if (n_seq <= XE_NXCD) return xe_smem<XF<XC, 12, 2, false, 1, 2, 1>>(n_layer);
    if (n_seq <= 2 * XE_NXCD) return two_wpc ? 2 * (xe_smem<XF<XC, 8, 4, false, 2, 1, 1>>(n_layer) < 54 * 1024 ? (size_t)54 * 1024 : xe_smem<XF<XC, 8, 4, false, 2, 1, 1>>(n_layer)) : xe_smem<XF<XC, 12, 4, false, 1, 2, 2>>(n_layer);
    const size_t s12 = xe_smem<XF<XC, 12, 2, false, 1, 2, 4>>(n_layer), s8 = xe_smem<XF<XC, 8, 8, false, 1, 2, 4>>(n_layer);
    return s12 < s8 ? s12 : s8;
}
// n_steps decode steps of every sequence in ONE launch; with_head: 0 layers only (x_out), 1 + logits, 2 + greedy pick and state update (needed for n_steps > 1)
int xengine_steps(XEngineHost* E, hipStream_t st, int32_t* d_state, uint16_t* x_out, int with_head, int n_steps) {
    XArgs& a = E->args;
    if (!a.emb || !d_state || !x_out || n_steps < 1 || (n_steps > 1 && with_head != 2)) return KF_INVALID_ARGS;
    if (with_head && !a.head_w) return KF_INVALID_ARGS;
    XArgs save = a;
    a.d_state = d_state, a.x_out = x_out, a.n_steps = n_steps, a.pick = with_head == 2 ? 1 : 0;
    a.epoch0 = E->epoch, E->epoch += n_steps; /* generations never repeat between resets (a 31-bit count of steps) */
    if (!with_head) a.head_w = nullptr;
    int rc;
    if (E->fmt != FMT_Q4P) rc = xengine_go_lowbit(E, st);
    else
#ifdef XE_NB4_VARIANTS /* tuning builds: the ring depth of the other shapes */
    if (E->shape_class == 3 && a.n_seq <= XE_NXCD && E->variant_set && E->depth == 4) rc = xengine_go<XC3<12, 4, false, 1>>(E, st);
    else if (E->shape_class == 3 && a.n_seq <= XE_NXCD && E->variant_set && E->depth == 2) rc = xengine_go<XC3<12, 2, false, 1>>(E, st);
    else if (E->shape_class == 4 && E->variant_set && E->depth == 4) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 12, 2560, 4096, 1024, 9728, 4, false, 1, 1>>(E, st);
    else if (E->shape_class == 4 && E->variant_set && E->depth == 2) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 12, 2560, 4096, 1024, 9728, 2, false, 1, 1>>(E, st);
    else if (E->shape_class == 5 && E->variant_set && E->depth == 2) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 12, 4096, 4096, 1024, 12288, 2, false, 1, 1>>(E, st);
    else
#endif
    if (E->shape_class == 3) { /* the default instantiations only (no tuning variants, no stamps) */
#ifdef XE_C3_VARIANTS /* tuning builds: the two-sequences-per-decoder form of the 1.7B shape */
        if (a.n_seq > XE_NXCD && E->variant_set && E->nwv == 12 && E->depth == 2) rc = xengine_go<XC3<12, 2, false, 1, 2, 2>>(E, st);
        else if (a.n_seq > XE_NXCD && E->variant_set && E->nwv == 12 && E->depth == 21) rc = xengine_go<XC3<12, 2, false, 1, 1, 2>>(E, st);
        else if (a.n_seq > XE_NXCD && E->variant_set && E->nwv == 12 && E->depth == 4) rc = xengine_go<XC3<12, 4, false, 1, 2, 2>>(E, st);
        else if (a.n_seq > XE_NXCD && E->variant_set && E->nwv == 8 && E->depth == 4) rc = xengine_go<XC3<8, 4, false, 1, 2, 2>>(E, st);
        else if (a.n_seq > XE_NXCD && E->variant_set && E->nwv == 8 && E->depth == 6) rc = xengine_go<XC3<8, 6, false, 1, 2, 2>>(E, st);
        else if (a.n_seq > XE_NXCD && E->variant_set && E->nwv == 8 && E->depth == 2) rc = xengine_go<XC3<8, 2, false, 1, 2, 2>>(E, st);
        else
#endif
        // two sequences per decoder: 6 compute waves + 2 pollers at 256 registers (measured at 2 k keys, 16 sequences: 2180 tokens/s; 12 waves at 168 registers spill in the
        // streaming loops: 1650 - 1690; round 5's two decoders per XCD 1600; eight sequences, one per decoder: 1940)
        if (a.n_seq > XE_NXCD && !E->two_wpc && xe_smem<XC3<8, 8, false, 1, 2, 2>>(a.n_layer) <= 160 * 1024) rc = xengine_go<XC3<8, 8, false, 1, 2, 2>>(E, st); /* (ring depth 4 / 6 / 8: 2180 / 2230 / 2240) */
        else rc = a.n_seq > XE_NXCD ? xengine_go<XC3<8, 4, false, 2, 1>>(E, st) : xengine_go<XC3<12, 6, false, 1>>(E, st);
    }
    else if (E->shape_class == 4)
        rc = E->nwv == 8 ? xengine_go<XC4>(E, st) : xengine_go<XC4W>(E, st);
    else if (E->shape_class == 5)
        rc = E->nwv == 8 ? xengine_go<XC5>(E, st) : xengine_go<XC5W>(E, st);
    else if (E->shape_class == 6)
        rc = xengine_go<XC6>(E, st);
    else if (E->shape_class == 8) {
#ifdef XE_TP9_VARIANTS
        if (E->variant_set && E->nwv == 12 && E->depth == 2) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 12, 4096, 512, 128, 1536, 2, false, 1, 1, true>>(E, st);
        else
#endif
        rc = xengine_go<XC8>(E, st);
    }
    else if (E->shape_class == 9) {
#ifdef XE_TP9_VARIANTS /* tuning builds (scratch/xtp_time.py CONFIG=qwen3-4b VARIANT=): stamps, ring depth, the 8-wave 256-register form, two key tiles per batch */
        if (E->args.dbg) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 12, 2560, 512, 128, 1280, 6, true, 1, 1, true>>(E, st);
        else if (E->variant_set && E->nwv == 12 && E->depth == 6) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 12, 2560, 512, 128, 1280, 6, false, 1, 1, true>>(E, st);
        else if (E->variant_set && E->nwv == 12 && E->depth == 4) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 12, 2560, 512, 128, 1280, 4, false, 1, 1, true>>(E, st);
        else if (E->variant_set && E->nwv == 8 && E->depth == 8) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 8, 2560, 512, 128, 1280, 8, false, 1, 2, true>>(E, st);
        else if (E->variant_set && E->nwv == 8 && E->depth == 4) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 8, 2560, 512, 128, 1280, 4, false, 1, 2, true>>(E, st);
        else if (E->variant_set && E->nwv == 16 && E->depth == 2) rc = xengine_go<XCfg<FMT_Q4P, 4, 128, 16, 2560, 512, 128, 1280, 2, false, 1, 1, true>>(E, st);
        else
#endif
        rc = xengine_go<XC9>(E, st);
    }
    else if (E->shape_class == 7) {
        if (E->args.dbg) rc = xengine_go<XC7D>(E, st);
#ifdef XE_TP_VARIANTS /* tuning builds only (scratch/xtp_time.py VARIANT=): ring depth 8 / 4, two key tiles per attention batch, the 8-wave 256-register form.  Measured,
                         ms per step of 16 layers at 4 k keys: default (12 waves, depth 6, one tile) 2.802, depth 8 2.895, depth 4 2.786, two tiles 3.157, 8 waves 2.896 */
        else if (E->depth == 8) rc = xengine_go<XCfg<FMT_Q4P, 8, 128, 12, 5120, 1024, 128, 3200, 8, false, 1, 1, true>>(E, st);
        else if (E->depth == 4) rc = xengine_go<XCfg<FMT_Q4P, 8, 128, 12, 5120, 1024, 128, 3200, 4, false, 1, 1, true>>(E, st);
        else if (E->depth == 62) rc = xengine_go<XCfg<FMT_Q4P, 8, 128, 12, 5120, 1024, 128, 3200, 6, false, 1, 2, true>>(E, st);
        else if (E->nwv == 8) rc = xengine_go<XCfg<FMT_Q4P, 8, 128, 8, 5120, 1024, 128, 3200, 8, false, 1, 2, true>>(E, st);
#endif
        else rc = xengine_go<XC7>(E, st);
    }
    else
        rc = E->shape_class == 1 ? xengine_go_shape<XC1>(E, st) : xengine_go_shape<XC2>(E, st);
    a.head_w = save.head_w;
    return rc;
}
int xengine_set_embedding(XEngineHost* E, const kf_weight* w, const int32_t* d_forced, int forced_stride) {
    if (!w || w->type != KF_BF16 || w->quant != KF_QUANT_GROUP || w->qzeros || w->ne1 != E->dim || !w->data) return KF_UNSUPPORTED_DATATYPE;
    E->args.emb = reinterpret_cast<const uint16_t*>(w->data), E->args.emb_rows = w->ne0, E->args.d_forced = d_forced, E->args.forced_stride = forced_stride;
    return KF_OK;
}
int xengine_set_head(XEngineHost* E, const kf_weight* w, const uint16_t* norm_w, uint16_t* logits, int32_t* d_tokens_out, int tokens_stride) {
    XArgs& a = E->args;
    if (E->tp_bytes) return KF_INVALID_ARGS; /* a TP engine takes its head in vocabulary shards: kf_xengine_set_head_tp */
    if (!w) { /* no head: kf_xengine_steps then refuses (every step of the ABI ends in the head); kept so that a head can be taken away before its weight is freed */
        a.head_w = nullptr, a.head_norm = nullptr, a.logits = nullptr, a.d_tokens_out = nullptr, a.vocab = 0;
        return KF_OK;
    }
    if (w->type != KF_BF16 || w->quant != KF_QUANT_GROUP || w->qzeros || w->ne1 != E->dim || !w->data || ((uintptr_t)w->data & 15) != 0 || !norm_w || !logits || w->ne0 < 64)
        return KF_UNSUPPORTED_DATATYPE;
    const int nBlk = E->dim / 8;
    if (gemv_lpr_log2(nBlk, w->ne0) != c_lpr_log2(nBlk, 1L << 20)) return KF_UNSUPPORTED_DATATYPE; /* same lanes per row as the mat-vec launcher: same summation order */
    a.head_w = (g_u32x4)(uintptr_t)w->data, a.head_norm = (g_u16)(uintptr_t)norm_w, a.logits = logits, a.d_tokens_out = d_tokens_out, a.tokens_stride = tokens_stride, a.vocab = w->ne0;
    return KF_OK;
}
int xengine_error_word(XEngineHost* E, hipStream_t st, int* h_err) {
    int v[2] = {0, 0};
    if (hipMemcpyAsync(v, E->args.ws, sizeof(v), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return KF_HIP_CHECK;
    *h_err = v[1];
    return KF_OK;
}
int xengine_reset(XEngineHost* E, hipStream_t st) {
    const int rc = xengine_init_state(E, st);
    if (rc != KF_OK) return rc;
    return hipStreamSynchronize(st) == hipSuccess ? KF_OK : KF_HIP_CHECK;
}
void xengine_set_variant(XEngineHost* E, int nwv, int depth) {
    if (nwv == 0) { /* tuning hook: depth = the stagger of the second decoder in microseconds */
        E->args.stagger_us = depth;
        return;
    }
    if (nwv == -1) { /* tuning hook: depth = XArgs::deal_wl (0: the default of the form) */
        E->deal_wl = depth;
        return;
    }
    if (nwv == -2) { /* A/B hook: depth != 0 = 9 .. 16 sequences through the round-5 form (two decoders per XCD) instead of the batched one */
        E->two_wpc = depth != 0;
        return;
    }
    E->nwv = nwv, E->depth = depth, E->variant_set = 1;
}
int xengine_debug_enable(XEngineHost* E, int seq, int wg, int max_steps) {
    XArgs& a = E->args;
    const size_t bytes = (size_t)max_steps * a.n_layer * 64 * 8;
    if (a.dbg) (void)hipFree(a.dbg), a.dbg = nullptr;
    if (seq < 0) return KF_OK;
    if (hipMalloc(&a.dbg, bytes) != hipSuccess) {
        a.dbg = nullptr;
        return KF_HIP_CHECK;
    }
    (void)hipMemset(a.dbg, 0, bytes);
    a.dbg_seq = seq, a.dbg_wg = wg, a.dbg_steps = max_steps;
    return KF_OK;
}
int xengine_debug_read(XEngineHost* E, unsigned long long* h_out, int n_words) {
    if (!E->args.dbg) return 0;
    if (hipMemcpy(h_out, E->args.dbg, (size_t)n_words * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n_words;
}

}  // namespace kf
