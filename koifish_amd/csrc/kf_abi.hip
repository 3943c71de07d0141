// kf_abi.hip -- extern "C" boundary of libkf_hip.so (include/kf_abi.h).  Argument checks, stream plumbing and
// hipGraph capture live here; the kernels are in kf_gemv.hip / kf_attn.hip / kf_ops.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>


#include <mutex>
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "kf_kernels.h"

struct kf_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    bool capturing;
    float* amax_val; /* per-workgroup partial maxima for kf_lm_head when the caller passes no scratch */
    int* amax_idx;
    int canonical;    /* 1: the decode kernels sum in the canonical order of oracle/kf_oracle.c sections 4c / 6 (bit-exact against the oracle; the default); 0: v_dot2c / fp32 forms */
    void* scratch;    /* caller-owned workspace of kf_linear (kf_set_scratch): AWQ slice partials, or a weight dequantised to bf16 */
    size_t scratch_bytes;
    // resident GetDataX copies for token batches (kf_set_dequant_arena): caller-owned memory, filled on first use, keyed by the data pointers of the matrices a route
    // multiplies in one launch and the form they are laid out in (DEQ_STACK: back to back; DEQ_ILV: gate | up interleaved in blocks of 16 rows)
    struct DeqCopy {
        const void* key[3];
        int form;
        size_t off;
    };
    void* arena;
    size_t arena_bytes, arena_used;
    std::vector<DeqCopy> copies;
};
enum { DEQ_STACK = 0, DEQ_ILV = 1 };
struct kf_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(KF_HIP_CHECK, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHKCTX(c) \
    if (!(c)) return fail(KF_INVALID_ARGS, "%s: null kf_ctx", __func__)
#define RET(code) \
    do {          \
        int c_ = (code); \
        if (c_ != KF_OK) return fail(c_, "%s failed with %d", __func__, c_); \
        return KF_OK;    \
    } while (0)

static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" {

const char* kf_last_error(void) { return g_err; }
const char* kf_version(void) { return "koifish_amd 0.1 (gfx950)"; }

int kf_init(int device, void* stream, kf_ctx** out) {
    if (!out) return fail(KF_INVALID_ARGS, "kf_init: out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(KF_HIP_CHECK, "kf_init: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(KF_INVALID_ARGS, "kf_init: device %d of %d", device, n);
    HIPCHK(hipSetDevice(device));
    kf_ctx* c = new kf_ctx();
    c->device = device;
    c->capturing = false;
    if (stream) {
        c->stream = (hipStream_t)stream, c->own_stream = false;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    HIPCHK(hipMalloc(&c->amax_val, sizeof(float) * kf::KF_MAX_ARGMAX_PARTIALS));
    HIPCHK(hipMalloc(&c->amax_idx, sizeof(int) * kf::KF_MAX_ARGMAX_PARTIALS));
    c->scratch = nullptr, c->scratch_bytes = 0;
    c->arena = nullptr, c->arena_bytes = c->arena_used = 0;
    c->canonical = 1;
    *out = c;
    return KF_OK;
}
// bf16 view of a weight: the data itself (bf16 storage) or its dequantisation into the caller's scratch (kf_set_scratch)
static int lib_weight_bf16(kf_ctx* c, const kf_weight* w, const uint16_t** out) {
    if (w->type == KF_BF16) {
        *out = (const uint16_t*)w->data;
        return KF_OK;
    }
    const size_t need = (size_t)w->ne0 * w->ne1 * 2;
    if (need > c->scratch_bytes || !c->scratch)
        return fail(KF_INVALID_ARGS, "kf_linear: this weight needs %zu bytes of scratch (kf_linear_scratch_bytes), kf_set_scratch gave %zu", need, c->scratch_bytes);
    const int r = kf::dequant_launch(c->stream, w, (uint16_t*)c->scratch);
    *out = (const uint16_t*)c->scratch;
    return r;
}
static size_t up256z(size_t v) { return (v + 255) & ~(size_t)255; }
// The bf16 copies of n_w group-quantised matrices of one input width, laid out as `form` says: from the arena when the caller lent one (dequantised on first use and
// kept), otherwise dequantised into the scratch for this call.  KF_OK: *out holds them; 1: no room in either; < 0 error.
static int deq_copies(kf_ctx* c, int n_w, const kf_weight* const* w, int form, const uint16_t** out, bool* resident = nullptr, bool arena_only = false) {
    size_t need = 0;
    for (int i = 0; i < n_w; i++) need += up256z((size_t)w[i]->ne0 * w[i]->ne1 * 2);
    if (resident) *resident = false;
    if (c->arena) {
        for (const kf_ctx::DeqCopy& e : c->copies) {
            bool same = e.form == form;
            for (int i = 0; i < 3; i++) same = same && e.key[i] == (i < n_w ? w[i]->data : nullptr);
            if (same) {
                *out = (const uint16_t*)((const char*)c->arena + e.off);
                if (resident) *resident = true;
                return KF_OK;
            }
        }
    }
    const bool keep = c->arena && !c->capturing && c->arena_used + need <= c->arena_bytes; /* a captured launch would replay the fill with every replay: those use the scratch */
    char* dst = keep ? (char*)c->arena + c->arena_used : (char*)c->scratch;
    if (!keep && (arena_only || !c->scratch || c->scratch_bytes < need)) return 1;
    size_t off = 0;
    for (int i = 0; i < n_w; i++) {
        const int r = form == DEQ_ILV ? kf::dequant_launch(c->stream, w[i], (uint16_t*)dst, n_w, i) : kf::dequant_launch(c->stream, w[i], (uint16_t*)(dst + off));
        if (r != KF_OK) return r < 0 ? r : KF_HIP_CHECK;
        off += up256z((size_t)w[i]->ne0 * w[i]->ne1 * 2);
    }
    if (keep) {
        kf_ctx::DeqCopy e;
        for (int i = 0; i < 3; i++) e.key[i] = i < n_w ? w[i]->data : nullptr;
        e.form = form, e.off = c->arena_used;
        c->copies.push_back(e);
        c->arena_used += need;
        if (resident) *resident = true;
    }
    *out = (const uint16_t*)dst;
    return KF_OK;
}

int kf_destroy(kf_ctx* c) {
    if (!c) return KF_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->amax_val), (void)hipFree(c->amax_idx);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return KF_OK;
}
int kf_sync(kf_ctx* c) {
    CHKCTX(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    return KF_OK;
}
int kf_malloc(kf_ctx* c, size_t bytes, void** out) {
    CHKCTX(c);
    if (!out) return fail(KF_INVALID_ARGS, "kf_malloc: out is null");
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(KF_OUTOF_GPUMEMORY, "kf_malloc(%zu): %s", bytes, hipGetErrorString(e));
    return KF_OK;
}
int kf_free(kf_ctx* c, void* p) {
    CHKCTX(c);
    if (p) HIPCHK(hipFree(p));
    return KF_OK;
}
int kf_memset(kf_ctx* c, void* p, int v, size_t bytes) {
    CHKCTX(c);
    HIPCHK(hipMemsetAsync(p, v, bytes, c->stream));
    return KF_OK;
}
int kf_memset32(kf_ctx* c, void* p, int32_t v, size_t count) {
    CHKCTX(c);
    if (!p || ((uintptr_t)p & 3)) return fail(KF_INVALID_ARGS, "kf_memset32: null or unaligned pointer");
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)p, v, count, c->stream));
    return KF_OK;
}
int kf_h2d(kf_ctx* c, void* dst, const void* src, size_t bytes) {
    CHKCTX(c);
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KF_OK;
}
int kf_d2h(kf_ctx* c, void* dst, const void* src, size_t bytes) {
    CHKCTX(c);
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KF_OK;
}
int kf_d2d(kf_ctx* c, void* dst, const void* src, size_t bytes) {
    CHKCTX(c);
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return KF_OK;
}

int kf_graph_begin(kf_ctx* c) {
    CHKCTX(c);
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_graph_begin: already capturing");
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    return KF_OK;
}
int kf_graph_end(kf_ctx* c, kf_graph** out) {
    CHKCTX(c);
    if (!c->capturing || !out) return fail(KF_INVALID_ARGS, "kf_graph_end: not capturing");
    c->capturing = false;
    kf_graph* g = new kf_graph();
    hipError_t e = hipStreamEndCapture(c->stream, &g->graph);
    if (e != hipSuccess) {
        delete g;
        return fail(KF_HIP_CHECK, "hipStreamEndCapture: %s", hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g->graph);
        delete g;
        return fail(KF_HIP_CHECK, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    *out = g;
    return KF_OK;
}
int kf_graph_launch(kf_ctx* c, kf_graph* g) {
    CHKCTX(c);
    if (!g) return fail(KF_INVALID_ARGS, "kf_graph_launch: null graph");
    HIPCHK(hipGraphLaunch(g->exec, c->stream));
    return KF_OK;
}
int kf_graph_destroy(kf_graph* g) {
    if (!g) return KF_OK;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return KF_OK;
}

int kf_event_create(void** ev) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void*)e;
    return KF_OK;
}
int kf_event_record(kf_ctx* c, void* ev) {
    CHKCTX(c);
    HIPCHK(hipEventRecord((hipEvent_t)ev, c->stream));
    return KF_OK;
}
int kf_event_elapsed_ms(void* a, void* b, float* ms) {
    HIPCHK(hipEventSynchronize((hipEvent_t)b));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return KF_OK;
}
int kf_event_destroy(void* ev) {
    if (ev) HIPCHK(hipEventDestroy((hipEvent_t)ev));
    return KF_OK;
}

// ------------------------------------------------------------------------------------------------ operators
static int check_weight(const kf_weight* w, const char* who) {
    if (!w || !w->data) return fail(KF_INVALID_ARGS, "%s: null weight", who);
    if (w->ne0 <= 0 || w->ne1 <= 0) return fail(KF_INVALID_ARGS, "%s: bad shape %d x %d", who, w->ne0, w->ne1);
    if (!al16(w->data)) return fail(KF_BLAS_UNALIGN, "%s: weight data not 16-byte aligned", who);
    if (w->qzeros || w->qscales) { /* AutoAWQ layout */
        if (!w->qzeros || !w->qscales || w->type != KF_Q4 || w->lGroup != 128 || (w->ne0 % 8) || (w->ne1 % 128) || !al16(w->qscales))
            return fail(KF_QUANT_ERR, "%s: malformed AWQ weight (%d x %d, group %d)", who, w->ne0, w->ne1, w->lGroup);
        return KF_OK;
    }
    if ((w->quant == KF_QUANT_ROW_LUT && (w->type == KF_Q3 || w->type == KF_Q2)) || (w->quant == KF_QUANT_ROW_RTN && w->type == KF_Q2)) {
        /* 3- / 2-bit row forms: dequant-only storage (kf_dequant, kf_linear through it, kf_quantize for NF3) */
        if (!w->gama) return fail(KF_QUANT_ERR, "%s: row-quantised weight without gama", who);
        if (w->ne1 % 8) return fail(KF_BLAS_UNALIGN, "%s: rows of %d weights are not whole 8-weight units", who, w->ne1);
        return KF_OK;
    }
    if (w->quant == KF_QUANT_ROW_RTN) return fail(KF_QUANT_ERR, "%s: row-RTN storage exists for KF_Q2 only (type %d)", who, w->type);
    if (w->quant == KF_QUANT_ROW_LUT) { /* row-codebook 4-bit storage (GeQuant::RT_NormalF): nibble stream + 16 table entries per row */
        if (w->type != KF_Q4 || !w->gama) return fail(KF_QUANT_ERR, "%s: malformed row-LUT weight (type %d)", who, w->type);
        if (w->ne1 % 32) return fail(KF_BLAS_UNALIGN, "%s: row-LUT rows of %d weights are not 16-byte aligned", who, w->ne1);
        if (!al16(w->gama + w->ne0 + w->ne1)) return fail(KF_BLAS_UNALIGN, "%s: row-LUT tables not 16-byte aligned (ne0 = %d must be a multiple of 8)", who, w->ne0);
        return KF_OK;
    }
    if (w->quant != KF_QUANT_GROUP) return fail(KF_UNSUPPORTED_DATATYPE, "%s: unknown quant mode %d", who, w->quant);
    switch (w->type) {
        case KF_BF16: case KF_F8E5M2: break;
        case KF_Q4: case KF_T_SIGN: case KF_BOOL1: case KF_T_BINARY:
            if (!w->gama) return fail(KF_QUANT_ERR, "%s: quantised weight without gama", who);
            if (w->lGroup <= 0 || ((long long)w->ne0 * w->ne1) % w->lGroup) return fail(KF_QUANT_ERR, "%s: bad group size %d", who, w->lGroup);
            break;
        default: return fail(KF_UNSUPPORTED_DATATYPE, "%s: unsupported weight type %d", who, w->type);
    }
    return KF_OK;
}

int kf_dequant(kf_ctx* c, const kf_weight* w, kf_bf16* out) {
    CHKCTX(c);
    int r = check_weight(w, "kf_dequant");
    if (r) return r;
    if (!out || !al16(out)) return fail(KF_BLAS_UNALIGN, "kf_dequant: out null/unaligned");
    if (w->qzeros) RET(kf::awq_dequant_launch(c->stream, w, out));
    RET(kf::dequant_launch(c->stream, w, out));
}
int kf_quantize(kf_ctx* c, const kf_weight* w, const kf_bf16* src, int symmetric) {
    CHKCTX(c);
    int r = check_weight(w, "kf_quantize");
    if (r) return r;
    if (!src) return fail(KF_INVALID_ARGS, "kf_quantize: null src");
    RET(kf::quantize_launch(c->stream, w, src, symmetric));
}

static void init_args(kf_ctx* c, kf::GemvLaunch& L) { memset(&L, 0, sizeof(L)); L.args.alpha = 1.0f; L.canon = c->canonical; }

static const int KF_DEQ_GEMM_MIN = 2048; /* token rows from which a quantised weight is dequantised once and multiplied by the bf16 tile kernel (when scratch was handed over) */
size_t kf_linear_scratch_bytes(const kf_weight* w, int nTok) {
    if (!w || nTok < 1) return 0;
    if (w->qzeros) return kf::awq_scratch_bytes(w);
    if (w->quant != KF_QUANT_GROUP) return (size_t)w->ne0 * w->ne1 * 2; /* row forms: GetDataX into the scratch (the 4-bit row codebook only for token batches the tile kernels do not cover) */
    // training-size batches of a quantised weight: one dequantise pass (a few % of the product at >= 2048 rows) buys the 256 x 256 bf16 tile kernel (kf_gemm3.hip)
    if (w->type != KF_BF16 && nTok >= KF_DEQ_GEMM_MIN && w->ne0 >= 256 && (w->ne1 % 64) == 0) return (size_t)w->ne0 * w->ne1 * 2;
    return 0;
}
int kf_set_canonical(kf_ctx* c, int on) {
    CHKCTX(c);
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_set_canonical: not while capturing");
    c->canonical = on ? 1 : 0;
    return KF_OK;
}
int kf_get_canonical(kf_ctx* c) { return c ? c->canonical : 0; }
int kf_set_dequant_arena(kf_ctx* c, void* arena, size_t bytes) {
    CHKCTX(c);
    if (arena && !al16(arena)) return fail(KF_BLAS_UNALIGN, "kf_set_dequant_arena: unaligned");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_set_dequant_arena: not while capturing");
    c->arena = arena, c->arena_bytes = arena ? bytes : 0, c->arena_used = 0;
    c->copies.clear();
    return KF_OK;
}
size_t kf_dequant_arena_used(kf_ctx* c) { return c ? c->arena_used : 0; }
size_t kf_resident_scratch_bytes(void) { return kf::gemm3_sk_ws_bytes(); }
int kf_set_scratch(kf_ctx* c, void* scratch, size_t bytes) {
    CHKCTX(c);
    if (scratch && !al16(scratch)) return fail(KF_BLAS_UNALIGN, "kf_set_scratch: unaligned");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_set_scratch: not while capturing (captured launches hold the old pointer)");
    c->scratch = scratch, c->scratch_bytes = scratch ? bytes : 0;
    return KF_OK;
}

int kf_linear(kf_ctx* c, const kf_weight* w, const kf_bf16* x, kf_bf16* y, const kf_bf16* bias, int nTok, float alpha, float beta, uint32_t epilogue,
              const kf_bf16* residual) {
    CHKCTX(c);
    int r = check_weight(w, "kf_linear");
    if (r) return r;
    if (nTok < 1) return fail(KF_INVALID_ARGS, "kf_linear: nTok=%d", nTok);
    if (!x || !y || !al16(x)) return fail(KF_BLAS_UNALIGN, "kf_linear: x/y null or x unaligned");
    if ((epilogue & KF_EPI_RESIDUAL) && !residual) return fail(KF_INVALID_ARGS, "kf_linear: residual epilogue without residual");
    if (nTok > 1 && ((w->ne1 * 2) % 16 != 0)) return fail(KF_BLAS_UNALIGN, "kf_linear: token rows of x are not 16-byte aligned");
    // nTok > 1 (SLP::Forw with a batch of tokens: x [nTok, ne1] row-major, y [nTok, ne0]): the MFMA tile kernels of kf_gemm.hip from 8 rows
    // up when the shape is covered (K a multiple of 128, 16-byte aligned rows), otherwise one mat-vec launch per token row.
    if (w->qzeros) { /* AutoAWQ layout: its own transposed mat-vec */
        const size_t need = kf::awq_scratch_bytes(w);
        if (need > c->scratch_bytes || !c->scratch)
            return fail(KF_INVALID_ARGS, "kf_linear: the AWQ mat-vec needs %zu bytes of scratch (kf_linear_scratch_bytes), kf_set_scratch gave %zu", need, c->scratch_bytes);
        for (int t = 0; t < nTok; t++) {
            int rc = kf::awq_linear_launch(c->stream, w, x + (size_t)t * w->ne1, y + (size_t)t * w->ne0, bias, alpha, beta,
                                           (epilogue & KF_EPI_RESIDUAL) ? residual + (size_t)t * w->ne0 : nullptr, (float*)c->scratch);
            if (rc != KF_OK) return fail(rc, "kf_linear (AWQ) failed with %d", rc);
        }
        return KF_OK;
    }
    const int gemm_min = kf::g_knobs.gemm_min; /* token rows from which the MFMA tile kernel replaces the per-token mat-vec loop (8) */
    if (w->quant != KF_QUANT_GROUP && w->type != KF_Q4) {
        // 3- / 2-bit row forms: GetDataX into the scratch, then the bf16 product, whatever the batch (the reference's own order; no in-place mat-vec)
        const uint16_t* Wd = nullptr;
        r = lib_weight_bf16(c, w, &Wd);
        if (r != KF_OK) return fail(r, "kf_linear (row-form dequant) failed with %d", r);
        kf_weight wb;
        memset(&wb, 0, sizeof(wb));
        wb.data = Wd, wb.type = KF_BF16, wb.ne0 = w->ne0, wb.ne1 = w->ne1;
        return kf_linear(c, &wb, x, y, bias, nTok, alpha, beta, epilogue, residual);
    }
    if (c->arena && nTok >= kf::g_knobs.resident_min && w->type != KF_BF16 && w->quant == KF_QUANT_GROUP && w->ne0 >= 128 && (w->ne1 % 64) == 0) {
        // a resident copy (kf_set_dequant_arena): nothing to dequantise -- the bf16 tile kernels of kf_gemm3.hip from g_knobs.resident_min token rows (the 1024-row o_proj / down_proj of a
        // long prompt as 256 tiles of 64 x 128 or 64 x 64)
        const kf_weight* one[1] = {w};
        const uint16_t* W = nullptr;
        r = deq_copies(c, 1, one, DEQ_STACK, &W, nullptr, true);
        if (r < 0) return fail(r, "kf_linear (resident dequantised copy) failed with %d", r);
        if (r == KF_OK) {
            kf_weight wb;
            memset(&wb, 0, sizeof(wb));
            wb.data = W, wb.type = KF_BF16, wb.ne0 = w->ne0, wb.ne1 = w->ne1;
            const bool lend = c->scratch && c->scratch_bytes >= kf::gemm3_sk_ws_bytes() && al16(c->scratch); /* the scratch holds no copy on this route: split-K slots */
            const int rc = kf::gemm_launch(c->stream, &wb, x, w->ne1, nTok, y, w->ne0, bias, alpha, beta, (epilogue & KF_EPI_RESIDUAL) ? residual : nullptr, w->ne0, lend ? c->scratch : nullptr,
                                           lend ? c->scratch_bytes : 0);
            if (rc < 0) return fail(rc, "kf_linear (bf16 tile GEMM on the resident copy) failed with %d", rc);
            if (rc == KF_OK) return KF_OK;
        }
    }
    if (nTok >= KF_DEQ_GEMM_MIN && w->type != KF_BF16 && w->quant == KF_QUANT_GROUP && w->ne0 >= 256 && (w->ne1 % 64) == 0 && c->scratch &&
        c->scratch_bytes >= (size_t)w->ne0 * w->ne1 * 2 &&
        ((long)((w->ne0 + 255) / 256) * ((nTok + 255) / 256) >= 128 || (long)((w->ne0 + 127) / 128) * ((nTok + 127) / 128) >= 256 /* the 128 x 128 form: M 1024 from 4096 rows */)) {
        // GetDataX into the caller's scratch, then the 256 x 256 bf16 tile kernel: the reference's own order, on our own kernels
        r = kf::dequant_launch(c->stream, w, (uint16_t*)c->scratch);
        if (r != KF_OK) return fail(r, "kf_linear (dequantise for the large-batch tile kernel) failed with %d", r);
        kf_weight wb;
        memset(&wb, 0, sizeof(wb));
        wb.data = c->scratch, wb.type = KF_BF16, wb.ne0 = w->ne0, wb.ne1 = w->ne1;
        const int rc = kf::gemm_launch(c->stream, &wb, x, w->ne1, nTok, y, w->ne0, bias, alpha, beta, (epilogue & KF_EPI_RESIDUAL) ? residual : nullptr, w->ne0);
        if (rc < 0) return fail(rc, "kf_linear (large-batch bf16 tile GEMM) failed with %d", rc);
        if (rc == KF_OK) return KF_OK;
    }
    if (nTok >= gemm_min) {
        const int rc = kf::gemm_launch(c->stream, w, x, w->ne1, nTok, y, w->ne0, bias, alpha, beta, (epilogue & KF_EPI_RESIDUAL) ? residual : nullptr, w->ne0);
        if (rc < 0) return fail(rc, "kf_linear (token-batch GEMM) failed with %d", rc);
        if (rc == KF_OK) return KF_OK;
    }
    if (nTok >= gemm_min && w->quant == KF_QUANT_ROW_LUT && c->scratch && c->scratch_bytes >= (size_t)w->ne0 * w->ne1 * 2) {
        // row-codebook storage whose shape the in-register-unpack tile kernels do not cover: the reference's own order -- GetDataX into the scratch,
        // then the bf16 product on the dequantised copy.  Mat-vecs (below) read the nibble stream directly.
        const uint16_t* Wd = nullptr;
        r = lib_weight_bf16(c, w, &Wd);
        if (r != KF_OK) return fail(r, "kf_linear (row-LUT dequant) failed with %d", r);
        kf_weight wb;
        memset(&wb, 0, sizeof(wb));
        wb.data = Wd, wb.type = KF_BF16, wb.ne0 = w->ne0, wb.ne1 = w->ne1;
        const int rc = kf::gemm_launch(c->stream, &wb, x, w->ne1, nTok, y, w->ne0, bias, alpha, beta, (epilogue & KF_EPI_RESIDUAL) ? residual : nullptr, w->ne0);
        if (rc < 0) return fail(rc, "kf_linear (row-LUT token-batch GEMM) failed with %d", rc);
        if (rc == KF_OK) return KF_OK;
    }
    for (int t = 0; t < nTok; t++) {
        kf::GemvLaunch L;
        init_args(c, L);
        L.n = 1, L.w[0] = w, L.mode = kf::GEMV_PLAIN;
        L.args.x = x + (size_t)t * w->ne1, L.args.job[0].y = y + (size_t)t * w->ne0;
        L.args.bias = bias, L.args.alpha = alpha, L.args.beta = beta;
        L.args.residual = (epilogue & KF_EPI_RESIDUAL) ? residual + (size_t)t * w->ne0 : nullptr;
        int rc = kf::gemv_launch(c->stream, L);
        if (rc != KF_OK) return fail(rc, "kf_linear failed with %d", rc);
    }
    return KF_OK;
}

int kf_linear_f32(kf_ctx* c, const kf_weight* w, const kf_bf16* x, float* y) {
    CHKCTX(c);
    int r = check_weight(w, "kf_linear_f32");
    if (r) return r;
    if (!x || !y || !al16(x)) return fail(KF_BLAS_UNALIGN, "kf_linear_f32: x/y null or x unaligned");
    kf::GemvLaunch L;
    init_args(c, L);
    L.n = 1, L.w[0] = w, L.mode = kf::GEMV_PLAIN;
    L.args.x = x, L.args.yf = y, L.args.job[0].y = nullptr;
    RET(kf::gemv_launch(c->stream, L));
}
int kf_tp_reduce(kf_ctx* c, const float* partials, int n_ranks, int n, const kf_bf16* residual, kf_bf16* out) {
    CHKCTX(c);
    if (!partials || !out || n_ranks < 1 || n < 1) return fail(KF_INVALID_ARGS, "kf_tp_reduce: bad args");
    RET(kf::tp_reduce_launch(c->stream, partials, n_ranks, n, residual, out));
}

// ---- tensor-parallel exchange over peer-mapped receive areas (kf_tp.hip)
static int tp_check(const kf_tp_comm* t, const char* who) {
    if (!t || t->world < 1 || t->world > 8 || t->rank < 0 || t->rank >= t->world || t->n_max < 1 || t->per_step < 1 || !t->recv || !t->d_step || !t->d_err)
        return fail(KF_INVALID_ARGS, "%s: bad kf_tp_comm", who);
    for (int r = 0; r < t->world; r++)
        if (!t->peer[r]) return fail(KF_INVALID_ARGS, "%s: peer %d not set", who, r);
    return KF_OK;
}
static unsigned long long* tp_vec_slot(void* area, const kf_tp_comm* t, uint32_t index, int rank) {
    return (unsigned long long*)area + ((size_t)(index & 1u) * t->world + rank) * t->n_max;
}
size_t kf_tp_push_bytes(uint32_t per_step) { return (size_t)per_step * sizeof(kf::TpPushDev); }
int kf_tp_commit(kf_ctx* c, const kf_tp_comm* t) {
    CHKCTX(c);
    int r = tp_check(t, "kf_tp_commit");
    if (r) return r;
    if (!t->d_push) return fail(KF_INVALID_ARGS, "kf_tp_commit: d_push is null (kf_tp_push_bytes(per_step) bytes of device memory)");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_tp_commit: not while capturing");
    std::vector<kf::TpPushDev> tab(t->per_step);
    for (uint32_t i = 0; i < t->per_step; i++) {
        kf::TpPushDev& d = tab[i];
        memset(&d, 0, sizeof(d));
        for (int p = 0; p < t->world; p++) d.peer[p] = tp_vec_slot(t->peer[p], t, i, t->rank);
        d.step = t->d_step, d.per_step = t->per_step, d.index = i, d.world = t->world;
    }
    HIPCHK(hipMemcpyAsync(t->d_push, tab.data(), tab.size() * sizeof(kf::TpPushDev), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KF_OK;
}
size_t kf_tp_recv_bytes(int world, int n_max) { return world < 1 || n_max < 1 ? 0 : ((size_t)2 * world * n_max + (size_t)2 * world) * 8; }
int kf_tp_alloc(kf_ctx* c, size_t bytes, void** out) {
    CHKCTX(c);
    if (!out || !bytes) return fail(KF_INVALID_ARGS, "kf_tp_alloc: bad args");
    // uncached (fine-grained) device memory: remote writes become visible to this device's polling loads inside a running kernel
    // No fallback to cached (coarse-grained) memory: a peer's stores into it are not guaranteed to reach a running kernel's polls, and every exchange would end in
    // a 2^24-spin timeout with no hint of the cause.
    hipError_t e = hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(KF_OUTOF_GPUMEMORY, "kf_tp_alloc(%zu): uncached device memory refused (%s); a cached allocation cannot serve as a receive area", bytes, hipGetErrorString(e));
    }
    HIPCHK(hipMemset(*out, 0, bytes));
    return KF_OK;
}
int kf_tp_ipc_export(kf_ctx* c, void* p, unsigned char handle[64]) {
    CHKCTX(c);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, p));
    memcpy(handle, &h, 64);
    return KF_OK;
}
int kf_tp_ipc_open(kf_ctx* c, const unsigned char handle[64], void** out) {
    CHKCTX(c);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    HIPCHK(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
    return KF_OK;
}
int kf_tp_ipc_close(kf_ctx* c, void* p) {
    CHKCTX(c);
    HIPCHK(hipIpcCloseMemHandle(p));
    return KF_OK;
}
int kf_linear_f32_push(kf_ctx* c, const kf_weight* w, const kf_bf16* x, const kf_tp_comm* t, uint32_t index) {
    CHKCTX(c);
    int r = check_weight(w, "kf_linear_f32_push");
    if (r) return r;
    if ((r = tp_check(t, "kf_linear_f32_push"))) return r;
    if (!x || !al16(x)) return fail(KF_BLAS_UNALIGN, "kf_linear_f32_push: x null or unaligned");
    if (w->ne0 > t->n_max || index + 1 >= t->per_step) return fail(KF_INVALID_ARGS, "kf_linear_f32_push: %d rows / index %u do not fit the comm", w->ne0, index);
    kf::GemvLaunch L;
    init_args(c, L);
    L.n = 1, L.w[0] = w, L.mode = kf::GEMV_PLAIN;
    L.args.x = x, L.args.job[0].y = nullptr;
    if (!t->d_push) return fail(KF_INVALID_ARGS, "kf_linear_f32_push: kf_tp_commit has not been called");
    L.args.tp = reinterpret_cast<const kf::TpPushDev*>(t->d_push) + index;
    RET(kf::gemv_launch(c->stream, L));
}
int kf_tp_reduce_recv(kf_ctx* c, const kf_tp_comm* t, uint32_t index, int n, const kf_bf16* residual, kf_bf16* out) {
    CHKCTX(c);
    int r = tp_check(t, "kf_tp_reduce_recv");
    if (r) return r;
    if (!out || n < 1 || n > t->n_max || index + 1 >= t->per_step) return fail(KF_INVALID_ARGS, "kf_tp_reduce_recv: bad args");
    RET(kf::tp_reduce_recv_launch(c->stream, tp_vec_slot(t->recv, t, index, 0), t->world, t->n_max, n, t->d_step, t->per_step, index, residual, out, t->d_err));
}
int kf_tp_lm_head(kf_ctx* c, const kf_bf16* x, const kf_bf16* norm_w, float eps, const kf_weight* w, kf_bf16* logits, int row0, const kf_tp_comm* t, void* scratch) {
    CHKCTX(c);
    int r = tp_check(t, "kf_tp_lm_head");
    if (r) return r;
    if ((r = check_weight(w, "kf_tp_lm_head"))) return r;
    if (!x || !al16(x) || !logits) return fail(KF_BLAS_UNALIGN, "kf_tp_lm_head: x null/unaligned or no logits buffer");
    float* av = scratch ? (float*)scratch : c->amax_val;
    int* ai = scratch ? (int*)((float*)scratch + kf::KF_MAX_ARGMAX_PARTIALS) : c->amax_idx;
    kf::GemvLaunch L;
    init_args(c, L);
    L.n = 1, L.w[0] = w, L.mode = kf::GEMV_ARGMAX;
    L.args.x = x, L.args.norm_w = norm_w, L.args.eps = eps, L.args.job[0].y = logits;
    L.args.amax_val = av, L.args.amax_idx = ai;
    r = kf::gemv_launch(c->stream, L);
    if (r) return fail(r, "kf_tp_lm_head: gemv failed with %d", r);
    unsigned long long* peers[8];
    for (int p = 0; p < t->world; p++) peers[p] = (unsigned long long*)t->peer[p] + (size_t)2 * t->world * t->n_max + 2 * t->rank;
    RET(kf::tp_argmax_push_launch(c->stream, av, ai, L.blocks, row0, peers, t->world, t->d_step, t->per_step, t->per_step - 1));
}
int kf_tp_pick(kf_ctx* c, const kf_tp_comm* t, int32_t* d_state, int32_t* d_tokens_out) {
    CHKCTX(c);
    int r = tp_check(t, "kf_tp_pick");
    if (r) return r;
    if (!d_state) return fail(KF_INVALID_ARGS, "kf_tp_pick: d_state is null");
    RET(kf::tp_pick_launch(c->stream, (const unsigned long long*)t->recv + (size_t)2 * t->world * t->n_max, t->world, t->d_step, t->per_step, t->per_step - 1, d_state,
                           d_tokens_out, t->d_err, 0x7fffffff));
}

int kf_norm_linear(kf_ctx* c, const kf_bf16* x, const kf_bf16* norm_w, float eps, int n_w, const kf_weight* const* w, kf_bf16* const* y,
                   const int64_t* y_pos_stride, int pos, const int32_t* d_pos) {
    CHKCTX(c);
    if (n_w < 1 || n_w > 3 || !w || !y) return fail(KF_INVALID_ARGS, "kf_norm_linear: n_w=%d", n_w);
    if (!x || !al16(x) || (norm_w && !al16(norm_w))) return fail(KF_BLAS_UNALIGN, "kf_norm_linear: x/norm_w null or unaligned");
    kf::GemvLaunch L;
    init_args(c, L);
    L.n = n_w, L.mode = kf::GEMV_PLAIN;
    for (int i = 0; i < n_w; i++) {
        int r = check_weight(w[i], "kf_norm_linear");
        if (r) return r;
        if (!y[i]) return fail(KF_INVALID_ARGS, "kf_norm_linear: y[%d] null", i);
        L.w[i] = w[i];
        L.args.job[i].y = y[i];
        L.args.job[i].y_pos_stride = y_pos_stride ? y_pos_stride[i] : 0;
    }
    L.args.x = x, L.args.norm_w = norm_w, L.args.eps = eps, L.args.pos = pos, L.args.d_pos = d_pos;
    RET(kf::gemv_launch(c->stream, L));
}

int kf_norm_gateup_swiglu(kf_ctx* c, const kf_bf16* x, const kf_bf16* norm_w, float eps, const kf_weight* gate, const kf_weight* up, kf_bf16* act) {
    CHKCTX(c);
    int r = check_weight(gate, "kf_norm_gateup_swiglu");
    if (r) return r;
    r = check_weight(up, "kf_norm_gateup_swiglu");
    if (r) return r;
    if (!x || !act || !al16(x)) return fail(KF_BLAS_UNALIGN, "kf_norm_gateup_swiglu: x/act");
    kf::GemvLaunch L;
    init_args(c, L);
    L.n = 2, L.w[0] = gate, L.w[1] = up, L.mode = kf::GEMV_PAIRED;
    L.args.x = x, L.args.norm_w = norm_w, L.args.eps = eps, L.args.job[0].y = act;
    RET(kf::gemv_launch(c->stream, L));
}

// ---- sparse forward (hot rows)
int kf_hot_rows(kf_ctx* c, const int32_t* d_hot, int n, int32_t* d_rows, int32_t* d_count) {
    CHKCTX(c);
    if (!d_hot || !d_rows || !d_count || n < 1) return fail(KF_INVALID_ARGS, "kf_hot_rows: null pointer or n < 1");
    RET(kf::hot_rows_launch(c->stream, d_hot, n, d_rows, d_count));
}
int kf_linear_masked(kf_ctx* c, const kf_weight* w, const kf_bf16* x, kf_bf16* y, const kf_bf16* bias, const int32_t* d_rows, int n_hot) {
    CHKCTX(c);
    int r = check_weight(w, "kf_linear_masked");
    if (r) return r;
    if (!x || !y || !al16(x) || !d_rows || n_hot < 0 || n_hot > w->ne0) return fail(KF_INVALID_ARGS, "kf_linear_masked: null / unaligned pointer or n_hot outside [0, ne0]");
    if (w->qzeros) return fail(KF_UNSUPPORTED_DATATYPE, "kf_linear_masked: AutoAWQ-layout weights are served by kf_linear only");
    r = kf::cold_fill_launch(c->stream, y, bias, w->ne0); /* cold rows: 0 (+ bias); the hot rows are overwritten below */
    if (r != KF_OK || n_hot == 0) RET(r);
    kf::GemvLaunch L;
    init_args(c, L);
    L.n = 1, L.w[0] = w, L.mode = kf::GEMV_PLAIN, L.n_hot = n_hot;
    L.args.x = x, L.args.job[0].y = y, L.args.bias = bias, L.args.row_map = d_rows;
    RET(kf::gemv_launch(c->stream, L));
}
int kf_norm_gateup_swiglu_masked(kf_ctx* c, const kf_bf16* x, const kf_bf16* norm_w, float eps, const kf_weight* gate, const kf_weight* up, kf_bf16* act,
                                 const int32_t* d_rows, int n_hot) {
    CHKCTX(c);
    int r = check_weight(gate, "kf_norm_gateup_swiglu_masked");
    if (r) return r;
    r = check_weight(up, "kf_norm_gateup_swiglu_masked");
    if (r) return r;
    if (!x || !act || !al16(x) || !d_rows || n_hot < 0 || n_hot > gate->ne0) return fail(KF_INVALID_ARGS, "kf_norm_gateup_swiglu_masked: bad pointer or n_hot");
    r = kf::cold_fill_launch(c->stream, act, nullptr, gate->ne0); /* SwiGLU of two zero projections is zero */
    if (r != KF_OK || n_hot == 0) RET(r);
    kf::GemvLaunch L;
    init_args(c, L);
    L.n = 2, L.w[0] = gate, L.w[1] = up, L.mode = kf::GEMV_PAIRED, L.n_hot = n_hot;
    L.args.x = x, L.args.norm_w = norm_w, L.args.eps = eps, L.args.job[0].y = act, L.args.row_map = d_rows;
    RET(kf::gemv_launch(c->stream, L));
}

int kf_rmsnorm(kf_ctx* c, const kf_bf16* x, const kf_bf16* w, kf_bf16* y, int rows, int dim, float eps, float* rstd) {
    CHKCTX(c);
    if (!x || !w || !y) return fail(KF_INVALID_ARGS, "kf_rmsnorm: null pointer");
    if (dim % 2 != 0) return fail(KF_RMS_PARAMS, "rmsnorm dim %d is not divisible by 2", dim);
    RET(kf::rmsnorm_launch(c->stream, x, w, y, rows, dim, eps, rstd));
}

int kf_rope_table_host(float* t, int n_pos, int hd, float theta) {
    if (!t || n_pos <= 0 || hd <= 0 || (hd & 1)) return fail(KF_INVALID_ARGS, "kf_rope_table_host: bad args");
    for (int p = 0; p < n_pos; p++)
        for (int j = 0; j < hd / 2; j++) {
            const float inv_freq = 1.0f / powf(theta, (float)(j * 2) / (float)hd);
            const float angle = (float)p * inv_freq;
            t[((size_t)p * (hd / 2) + j) * 2] = cosf(angle);
            t[((size_t)p * (hd / 2) + j) * 2 + 1] = sinf(angle);
        }
    return KF_OK;
}

int kf_qknorm_rope(kf_ctx* c, kf_bf16* q, kf_bf16* k, const kf_bf16* wq, const kf_bf16* wk, const float* table, int pos, const int32_t* d_pos, int n_head,
                   int n_kv, int hd, float eps) {
    CHKCTX(c);
    if (!q) return fail(KF_INVALID_ARGS, "kf_qknorm_rope: null q");
    RET(kf::qknorm_rope_launch(c->stream, q, k, wq, wk, table, pos, d_pos, n_head, n_kv, hd, eps));
}

static size_t attn_part_bytes(int n_head, int hd) { return sizeof(double) * (size_t)n_head * kf::KF_ATTN_MAX_SPLITS * (hd + 2); } /* {O[hd], L, m} fp64 per (head, slice) */
size_t kf_attn_scratch_bytes(int n_head, int hd) { return attn_part_bytes(n_head, hd) + kf::KF_ATTN_CNT_BYTES; /* + arrival counters */ }

int kf_attn_decode(kf_ctx* c, const kf_bf16* q, const kf_bf16* kc, const kf_bf16* vc, kf_bf16* out, int pos, const int32_t* d_pos, int n_head, int n_kv, int hd,
                   int kv_stride, void* scratch) {
    CHKCTX(c);
    if (!q || !kc || !vc || !out || !scratch) return fail(KF_INVALID_ARGS, "kf_attn_decode: null pointer");
    if (!al16(kc) || !al16(vc) || (kv_stride % 8)) return fail(KF_BLAS_UNALIGN, "kf_attn_decode: cache not 16-byte aligned");
    kf::AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.canon = c->canonical;
    a.q = q, a.kcache = const_cast<kf_bf16*>(kc), a.vcache = vc, a.out = out, a.part = (float*)((char*)scratch + kf::KF_ATTN_CNT_BYTES);
    a.pos = pos, a.d_pos = d_pos, a.n_head = n_head, a.n_kv = n_kv, a.hd = hd, a.kv_stride = kv_stride;
    a.counters = (int*)scratch; /* the first KF_ATTN_CNT_BYTES: arrival counters (fixed place whatever the shape) */
    RET(kf::attn_launch(c->stream, a));
}

int kf_attn_block(kf_ctx* c, const kf_bf16* q_raw, const kf_bf16* k_raw, kf_bf16* kc, const kf_bf16* vc, kf_bf16* out, const kf_bf16* wq, const kf_bf16* wk,
                  const float* table, int pos, const int32_t* d_pos, int n_head, int n_kv, int hd, int kv_stride, float eps, void* scratch) {
    CHKCTX(c);
    if (!q_raw || !k_raw || !kc || !vc || !out || !scratch || !table) return fail(KF_INVALID_ARGS, "kf_attn_block: null pointer");
    if (!al16(kc) || !al16(vc) || (kv_stride % 8)) return fail(KF_BLAS_UNALIGN, "kf_attn_block: cache not 16-byte aligned");
    kf::AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.canon = c->canonical;
    a.q = q_raw, a.k_raw = k_raw, a.kcache = kc, a.vcache = vc, a.out = out, a.part = (float*)((char*)scratch + kf::KF_ATTN_CNT_BYTES);
    a.wq_norm = wq, a.wk_norm = wk, a.rope_table = table, a.eps = eps;
    a.pos = pos, a.d_pos = d_pos, a.n_head = n_head, a.n_kv = n_kv, a.hd = hd, a.kv_stride = kv_stride;
    a.counters = (int*)scratch;
    RET(kf::attn_launch(c->stream, a));
}

int kf_swiglu(kf_ctx* c, const kf_bf16* gate, const kf_bf16* up, kf_bf16* out, int n) {
    CHKCTX(c);
    if (!gate || !up || !out || n <= 0) return fail(KF_INVALID_ARGS, "kf_swiglu: bad args");
    RET(kf::swiglu_launch(c->stream, gate, up, out, n));
}
int kf_add(kf_ctx* c, const kf_bf16* a, const kf_bf16* b, kf_bf16* out, int n) {
    CHKCTX(c);
    if (!a || !b || !out || n <= 0) return fail(KF_INVALID_ARGS, "kf_add: bad args");
    RET(kf::add_launch(c->stream, a, b, out, n));
}
int kf_embed(kf_ctx* c, const kf_weight* w, int token, const int32_t* d_token, kf_bf16* out) {
    CHKCTX(c);
    int r = check_weight(w, "kf_embed");
    if (r) return r;
    if (!out || !al16(out)) return fail(KF_BLAS_UNALIGN, "kf_embed: out null/unaligned");
    if (!d_token && (token < 0 || token >= w->ne0)) return fail(KF_INVALID_ARGS, "kf_embed: token %d outside [0,%d)", token, w->ne0);
    RET(kf::embed_launch(c->stream, w, token, d_token, nullptr, nullptr, out));
}
int kf_embed_state(kf_ctx* c, const kf_weight* w, const int32_t* d_state, const int32_t* d_forced, kf_bf16* out) {
    CHKCTX(c);
    int r = check_weight(w, "kf_embed_state");
    if (r) return r;
    if (!out || !al16(out) || !d_state) return fail(KF_INVALID_ARGS, "kf_embed_state: bad args");
    RET(kf::embed_launch(c->stream, w, 0, nullptr, d_state, d_forced, out));
}

// ---- token batch (prefill)
int kf_embed_batch(kf_ctx* c, const kf_weight* w, const int32_t* d_tokens, int n_tok, kf_bf16* out) {
    CHKCTX(c);
    int r = check_weight(w, "kf_embed_batch");
    if (r) return r;
    if (!out || !al16(out) || !d_tokens || n_tok < 1) return fail(KF_INVALID_ARGS, "kf_embed_batch: bad args");
    RET(kf::embed_launch(c->stream, w, 0, d_tokens, nullptr, nullptr, out, n_tok));
}
// Large token batches of matrices that share their input (Q | K | V, gate | up): GetDataX of each into the caller's scratch, back to back, then ONE launch of the
// 256 x 256 bf16 tile kernel over the stacked rows (each matrix a multiple of 256 rows; its rows go to its own output).  A 1024-row K or V projection alone is 4 x 8
// tiles at 2048 tokens -- an eighth of the chip -- and took 32 us on the in-register-unpack kernels; stacked with Q it is 128 tiles.
static const int KF_MULTI_DEQ_MIN = 1024; /* token rows from which the stacked route is taken (with a dequantise per call; resident copies: g_knobs.resident_min) */
static int multi_min(const kf_ctx* c) { return c && c->arena && kf::g_knobs.resident_min < KF_MULTI_DEQ_MIN ? kf::g_knobs.resident_min : KF_MULTI_DEQ_MIN; }
/* up256z: a no-op on these sizes (multiples of 256 rows x 64 columns): the stacked rows are contiguous */
static bool multi_deq_ok(int n_w, const kf_weight* const* w, int nTok, size_t* need, const kf_ctx* c = nullptr) {
    size_t tot = 0;
    long rows = 0;
    if (n_w < 2 || n_w > 3 || nTok < multi_min(c)) return false;
    for (int i = 0; i < n_w; i++) {
        if (w[i]->qzeros || w[i]->quant != KF_QUANT_GROUP || w[i]->ne0 < 256 || (w[i]->ne0 % 256) != 0 || (w[i]->ne1 % 64) != 0 || w[i]->ne1 != w[0]->ne1) return false;
        tot += up256z((size_t)w[i]->ne0 * w[i]->ne1 * 2), rows += w[i]->ne0;
    }
    if ((rows / 256) * ((nTok + 255) / 256) < 64 && (rows / 128) * ((nTok + 127) / 128) < 64) return false; /* fewer tiles than that: the in-register-unpack kernels */
    *need = tot;
    return true;
}
size_t kf_linear_multi_scratch_bytes(int n_w, const kf_weight* const* w, int nTok) {
    size_t need = 0;
    if (!w) return 0;
    for (int i = 0; i < n_w; i++)
        if (!w[i]) return 0;
    return multi_deq_ok(n_w, w, nTok, &need) ? need : 0;
}
// KF_OK done, 1 not this route, < 0 error
static int multi_deq_route(kf_ctx* c, int n_w, const kf_weight* const* w, const kf_bf16* x, kf_bf16* const* y, int nTok, const kf::G3Rope* rope = nullptr) {
    size_t need = 0;
    if (!multi_deq_ok(n_w, w, nTok, &need, c) || !al16(x)) return 1;
    int M[3] = {0, 0, 0};
    for (int i = 0; i < n_w; i++) M[i] = w[i]->ne0;
    const uint16_t* W = nullptr;
    const int r = deq_copies(c, n_w, w, DEQ_STACK, &W);
    if (r != KF_OK) return r;
    return kf::gemm3_multi_launch(c->stream, n_w, W, M, w[0]->ne1, x, w[0]->ne1, nTok, y, rope);
}
int kf_qkv_rope_batch(kf_ctx* c, const kf_weight* wq, const kf_weight* wk, const kf_weight* wv, const kf_bf16* x, kf_bf16* q, kf_bf16* k, kf_bf16* v, int nTok, const kf_bf16* wq_norm,
                      const kf_bf16* wk_norm, const float* rope_table, int pos0, int n_head, int n_kv, int hd, float eps) {
    return kf_qkv_rope_seqs(c, wq, wk, wv, x, q, k, v, nTok, 0, wq_norm, wk_norm, rope_table, pos0, n_head, n_kv, hd, eps);
}
int kf_qkv_rope_seqs(kf_ctx* c, const kf_weight* wq, const kf_weight* wk, const kf_weight* wv, const kf_bf16* x, kf_bf16* q, kf_bf16* k, kf_bf16* v, int nTok, int seq_len,
                     const kf_bf16* wq_norm, const kf_bf16* wk_norm, const float* rope_table, int pos0, int n_head, int n_kv, int hd, float eps) {
    CHKCTX(c);
    if (!wq || !wk || !wv || !x || !q || !k || !v || nTok < 1 || pos0 < 0 || seq_len < 0 || (seq_len > 0 && (pos0 != 0 || nTok % seq_len)))
        return fail(KF_INVALID_ARGS, "kf_qkv_rope_batch / _seqs: bad args");
    const kf_weight* ws[3] = {wq, wk, wv};
    kf_bf16* ys[3] = {q, k, v};
    if (hd == 128 && nTok >= multi_min(c) && wq->ne0 == n_head * hd && wk->ne0 == n_kv * hd) { /* one launch: the stacked tile GEMM with q/k-norm + RoPE in its epilogue */
        for (int i = 0; i < 3; i++) {
            const int r = check_weight(ws[i], "kf_qkv_rope_batch");
            if (r) return r;
        }
        const kf::G3Rope rp = {wq_norm, wk_norm, rope_table, pos0, eps, seq_len};
        const int rc = multi_deq_route(c, 3, ws, x, ys, nTok, &rp);
        if (rc < 0) return fail(rc, "kf_qkv_rope_batch (stacked tile GEMM + RoPE epilogue) failed with %d", rc);
        if (rc == KF_OK) return KF_OK;
    }
    const int r = kf_linear_multi(c, 3, ws, x, ys, nTok);
    if (r) return r;
    if (seq_len > 0) return kf_qknorm_rope_train(c, q, k, wq_norm, wk_norm, rope_table, nTok, seq_len, wq->ne0, wk->ne0, n_head, n_kv, hd, eps, nullptr, nullptr);
    return kf_qknorm_rope_batch(c, q, k, wq_norm, wk_norm, rope_table, pos0, nTok, wq->ne0, wk->ne0, n_head, n_kv, hd, eps);
}
int kf_linear_multi(kf_ctx* c, int n_w, const kf_weight* const* w, const kf_bf16* x, kf_bf16* const* y, int nTok) {
    CHKCTX(c);
    if (n_w < 1 || n_w > 3 || !w || !y || !x || nTok < 1) return fail(KF_INVALID_ARGS, "kf_linear_multi: bad args (n_w=%d nTok=%d)", n_w, nTok);
    for (int i = 0; i < n_w; i++) {
        int r = check_weight(w[i], "kf_linear_multi");
        if (r) return r;
        if (!y[i]) return fail(KF_INVALID_ARGS, "kf_linear_multi: y[%d] null", i);
        if (w[i]->ne1 != w[0]->ne1) return fail(KF_INVALID_ARGS, "kf_linear_multi: the matrices do not share the input width");
    }
    if (n_w > 1 && nTok >= multi_min(c)) {
        const int rc = multi_deq_route(c, n_w, w, x, y, nTok);
        if (rc < 0) return fail(rc, "kf_linear_multi (dequantise + stacked tile GEMM) failed with %d", rc);
        if (rc == KF_OK) return KF_OK;
    }
    // (bf16 storage from g3_first rows: kf_linear multiplies each matrix on the kf_gemm3.hip tile kernels -- the stacked in-register launch would be another summation order)
    const bool bf16_tiles = w[0]->type == KF_BF16 && nTok >= kf::g_knobs.g3_first;
    if (n_w > 1 && nTok >= 8 && !bf16_tiles) {
        const int rc = kf::gemm_multi_launch(c->stream, n_w, w, x, w[0]->ne1, nTok, y);
        if (rc < 0) return fail(rc, "kf_linear_multi failed with %d", rc);
        if (rc == KF_OK) return KF_OK;
    }
    for (int i = 0; i < n_w; i++) {
        int r = kf_linear(c, w[i], x, y[i], nullptr, nTok, 1.0f, 0.0f, KF_EPI_NONE, nullptr);
        if (r) return r;
    }
    return KF_OK;
}
int kf_gateup_swiglu_batch(kf_ctx* c, const kf_weight* gate, const kf_weight* up, const kf_bf16* x, kf_bf16* act, kf_bf16* up_scratch, int nTok) {
    CHKCTX(c);
    int r = check_weight(gate, "kf_gateup_swiglu_batch");
    if (r) return r;
    r = check_weight(up, "kf_gateup_swiglu_batch");
    if (r) return r;
    if (!x || !act || !up_scratch || nTok < 1) return fail(KF_INVALID_ARGS, "kf_gateup_swiglu_batch: bad args");
    if (gate->ne0 != up->ne0 || gate->ne1 != up->ne1) return fail(KF_INVALID_ARGS, "kf_gateup_swiglu_batch: gate and up shapes differ");
    if (nTok >= multi_min(c)) { /* gate | up dequantised interleaved, ONE tile-GEMM launch with the SwiGLU expression in its epilogue (on the two bf16-rounded projections,
                                       as the paired kernel and swiglu_kernel form it) */
        const kf_weight* ws[2] = {gate, up};
        size_t need = 0;
        const uint16_t* W = nullptr;
        if (multi_deq_ok(2, ws, nTok, &need, c) && al16(x) && gate->ne0 % 128 == 0 && deq_copies(c, 2, ws, DEQ_ILV, &W) == KF_OK) {
            const int rc = kf::gemm3_swiglu_launch(c->stream, W, gate->ne0, gate->ne1, x, gate->ne1, nTok, act);
            if (rc < 0) return fail(rc, "kf_gateup_swiglu_batch (interleaved dequantise + tile GEMM with SwiGLU epilogue) failed with %d", rc);
            if (rc == KF_OK) return KF_OK;
        }
        kf_bf16* ys[2] = {act, up_scratch};
        const int rc = multi_deq_route(c, 2, ws, x, ys, nTok);
        if (rc < 0) return fail(rc, "kf_gateup_swiglu_batch (dequantise + stacked tile GEMM) failed with %d", rc);
        if (rc == KF_OK) return kf_swiglu(c, act, up_scratch, act, (size_t)nTok * gate->ne0);
    }
    if (nTok >= 8 && !(gate->type == KF_BF16 && nTok >= kf::g_knobs.g3_first)) { /* bf16 storage from g3_first rows: two kf_linear (tile kernels) + kf_swiglu, as kf_linear_multi */
        const int rc = kf::gemm_paired_launch(c->stream, gate, up, x, gate->ne1, nTok, act);
        if (rc < 0) return fail(rc, "kf_gateup_swiglu_batch failed with %d", rc);
        if (rc == KF_OK) return KF_OK;
    }
    r = kf_linear(c, gate, x, act, nullptr, nTok, 1.0f, 0.0f, KF_EPI_NONE, nullptr);
    if (r) return r;
    r = kf_linear(c, up, x, up_scratch, nullptr, nTok, 1.0f, 0.0f, KF_EPI_NONE, nullptr);
    if (r) return r;
    return kf_swiglu(c, act, up_scratch, act, nTok * gate->ne0);
}
int kf_qknorm_rope_batch(kf_ctx* c, kf_bf16* q, kf_bf16* k, const kf_bf16* wq, const kf_bf16* wk, const float* table, int pos0, int n_tok, int64_t q_stride,
                         int64_t k_stride, int n_head, int n_kv, int hd, float eps) {
    CHKCTX(c);
    if (!q || n_tok < 1 || pos0 < 0) return fail(KF_INVALID_ARGS, "kf_qknorm_rope_batch: bad args");
    if (hd % 2) return fail(KF_RMS_PARAMS, "head_dim %d is not divisible by 2", hd);
    RET(kf::qknorm_rope_launch(c->stream, q, k, wq, wk, table, pos0, nullptr, n_head, n_kv, hd, eps, n_tok, q_stride, k_stride));
}
int kf_qknorm_rope_train(kf_ctx* c, kf_bf16* q, kf_bf16* k, const kf_bf16* wq, const kf_bf16* wk, const float* table, int n_tok, int seq_len, int64_t q_stride,
                         int64_t k_stride, int n_head, int n_kv, int hd, float eps, float* rstd_q, float* rstd_k) {
    CHKCTX(c);
    if (!q || n_tok < 1 || seq_len < 1) return fail(KF_INVALID_ARGS, "kf_qknorm_rope_train: bad args");
    if (hd % 2) return fail(KF_RMS_PARAMS, "head_dim %d is not divisible by 2", hd);
    RET(kf::qknorm_rope_launch(c->stream, q, k, wq, wk, table, 0, nullptr, n_head, n_kv, hd, eps, n_tok, q_stride, k_stride, seq_len, rstd_q, rstd_k));
}
int kf_attn_prefill(kf_ctx* c, const kf_bf16* q, const kf_bf16* kc, const kf_bf16* vc, kf_bf16* out, int pos0, int n_tok, int64_t q_stride, int n_head, int n_kv,
                    int hd, int kv_stride) {
    CHKCTX(c);
    if (!q || !kc || !vc || !out || n_tok < 1 || pos0 < 0) return fail(KF_INVALID_ARGS, "kf_attn_prefill: bad args");
    if (!al16(kc) || !al16(vc) || (kv_stride % 8)) return fail(KF_BLAS_UNALIGN, "kf_attn_prefill: cache not 16-byte aligned");
    if (n_tok >= 8) { /* fewer tokens: the per-token form of the decode kernel */
        const int rc = kf::attn_prefill_mfma_launch(c->stream, q, kc, vc, out, pos0, n_tok, q_stride, n_head, n_kv, hd, kv_stride);
        if (rc < 0) return fail(rc, "kf_attn_prefill: launch failed (%d)", rc);
        if (rc == KF_OK) return KF_OK;
    }
    kf::AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.canon = c->canonical;
    a.q = q, a.kcache = const_cast<kf_bf16*>(kc), a.vcache = vc, a.out = out;
    a.pos = pos0, a.n_head = n_head, a.n_kv = n_kv, a.hd = hd, a.kv_stride = kv_stride;
    a.n_tok = n_tok, a.q_stride = q_stride, a.one_slice = 1;
    RET(kf::attn_launch(c->stream, a));
}

int kf_attn_prefill_batch(kf_ctx* c, const kf_bf16* q, const kf_bf16* k, const kf_bf16* v, kf_bf16* out, int n_tok, int64_t q_stride, int n_head, int n_kv, int hd,
                          int kv_stride, int n_seq) {
    CHKCTX(c);
    if (!q || !k || !v || !out || n_tok < 1 || n_seq < 1) return fail(KF_INVALID_ARGS, "kf_attn_prefill_batch: bad args");
    if (!al16(k) || !al16(v) || (kv_stride % 8)) return fail(KF_BLAS_UNALIGN, "kf_attn_prefill_batch: k / v rows not 16-byte aligned");
    const int rc = kf::attn_prefill_mfma_launch(c->stream, q, k, v, out, 0, n_tok, q_stride, n_head, n_kv, hd, kv_stride, n_seq);
    if (rc == 1) return fail(KF_INVALID_ARGS, "kf_attn_prefill_batch: shape not covered by the tile kernel (head_dim 64 / 128, n_head / n_kv in 1, 2, 4, 8, 16-byte aligned rows)");
    RET(rc);
}
int kf_attn_prefill_batch_strided(kf_ctx* c, const kf_bf16* q, const kf_bf16* k, const kf_bf16* v, kf_bf16* out, int n_tok, int64_t q_stride, int64_t out_stride, int n_head,
                                  int n_kv, int hd, int kv_stride, int n_seq) {
    CHKCTX(c);
    if (!q || !k || !v || !out || n_tok < 1 || n_seq < 1 || out_stride < (int64_t)n_head * hd) return fail(KF_INVALID_ARGS, "kf_attn_prefill_batch_strided: bad args");
    if (!al16(k) || !al16(v) || (kv_stride % 8)) return fail(KF_BLAS_UNALIGN, "kf_attn_prefill_batch_strided: k / v rows not 16-byte aligned");
    const int rc = kf::attn_prefill_mfma_launch(c->stream, q, k, v, out, 0, n_tok, q_stride, n_head, n_kv, hd, kv_stride, n_seq, out_stride);
    if (rc == 1) return fail(KF_INVALID_ARGS, "kf_attn_prefill_batch_strided: shape not covered by the tile kernel (head_dim 64 / 128, n_head / n_kv in 1, 2, 4, 8, 16-byte aligned rows)");
    RET(rc);
}

int kf_set_state(kf_ctx* c, int32_t* d_state, int token, int pos) {
    CHKCTX(c);
    if (!d_state) return fail(KF_INVALID_ARGS, "kf_set_state: null state");
    RET(kf::set_state_launch(c->stream, d_state, token, pos));
}

size_t kf_head_scratch_bytes(void) { return (sizeof(float) + sizeof(int)) * (size_t)kf::KF_MAX_ARGMAX_PARTIALS; }

static int head_impl(kf_ctx* c, const kf_bf16* x, const kf_bf16* norm_w, float eps, const kf_weight* w, kf_bf16* logits, int32_t* d_argmax, int32_t* d_state,
                     int32_t* d_tokens_out, void* scratch, const char* who) {
    int r = check_weight(w, who);
    if (r) return r;
    if (!x || !al16(x)) return fail(KF_BLAS_UNALIGN, "%s: x null/unaligned", who);
    float* av = scratch ? (float*)scratch : c->amax_val;
    int* ai = scratch ? (int*)((float*)scratch + kf::KF_MAX_ARGMAX_PARTIALS) : c->amax_idx;
    kf_bf16* lg = logits;
    if (!lg) return fail(KF_INVALID_ARGS, "%s: logits buffer required (vocab*2 bytes)", who);
    kf::GemvLaunch L;
    init_args(c, L);
    L.n = 1, L.w[0] = w, L.mode = kf::GEMV_ARGMAX;
    L.args.x = x, L.args.norm_w = norm_w, L.args.eps = eps, L.args.job[0].y = lg;
    L.args.amax_val = av, L.args.amax_idx = ai;
    r = kf::gemv_launch(c->stream, L);
    if (r) return fail(r, "%s: gemv failed with %d", who, r);
    // (the pick stays its own 4-us launch: folded into this kernel behind an arrival ticket -- write-through partial stores, drained, one atomic per workgroup -- it
    // measured 2-5 us SLOWER per step than the separate launch, and with a release fence per workgroup 170 us slower)
    if (d_argmax || d_state) kf::argmax_finish_launch(c->stream, av, ai, L.blocks, d_argmax, d_state, d_tokens_out); /* both NULL: logits only */
    return hipGetLastError() == hipSuccess ? KF_OK : fail(KF_HIP_CHECK, "%s: launch failed", who);
}
int kf_lm_head(kf_ctx* c, const kf_weight* w, const kf_bf16* x, kf_bf16* logits, int32_t* d_argmax_out, void* scratch) {
    CHKCTX(c);
    return head_impl(c, x, nullptr, 0.f, w, logits, d_argmax_out, nullptr, nullptr, scratch, "kf_lm_head");
}
int kf_norm_lm_head(kf_ctx* c, const kf_bf16* x, const kf_bf16* norm_w, float eps, const kf_weight* w, kf_bf16* logits, int32_t* d_state, int32_t* d_tokens_out,
                    void* scratch) {
    CHKCTX(c);
    return head_impl(c, x, norm_w, eps, w, logits, nullptr, d_state, d_tokens_out, scratch, "kf_norm_lm_head");
}

int kf_sample(kf_ctx* c, const kf_bf16* logits, int n, int top_k, float temperature, float top_p, uint64_t* d_rng_state, int32_t* d_token, int32_t* d_state,
              int32_t* d_tokens_out, const int32_t* d_forced, int n_forced) {
    CHKCTX(c);
    if (!logits || !d_rng_state || (!d_token && !d_state)) return fail(KF_INVALID_ARGS, "kf_sample: null pointer");
    int r = kf::sample_launch(c->stream, logits, n, top_k, temperature, top_p, (unsigned long long*)d_rng_state, d_token, d_state, d_tokens_out, d_forced, n_forced);
    if (r == KF_INVALID_ARGS)
        return fail(r, "kf_sample: needs 2 <= top_k < n/2 (TOPK_heap::Select asserts it), top_k <= 1024, temperature > 0, top_p > 0 (got k=%d n=%d T=%g p=%g)",
                    top_k, n, temperature, top_p);
    RET(r);
}

int kf_sample_topk(kf_ctx* c, const kf_bf16* logits, int n, int top_k, float temperature, float top_p, uint64_t* d_rng_state, int32_t* d_token, int32_t* d_state,
                   int32_t* d_tokens_out, const int32_t* d_forced, int n_forced) {
    CHKCTX(c);
    if (!logits || !d_rng_state || (!d_token && !d_state)) return fail(KF_INVALID_ARGS, "kf_sample_topk: null pointer");
    int r = kf::sample_launch(c->stream, logits, n, top_k, temperature, top_p, (unsigned long long*)d_rng_state, d_token, d_state, d_tokens_out, d_forced, n_forced, 1);
    if (r == KF_INVALID_ARGS) return fail(r, "kf_sample_topk: needs 2 <= top_k < n/2, top_k <= 1024, temperature > 0, top_p > 0 (got k=%d n=%d T=%g p=%g)", top_k, n, temperature, top_p);
    RET(r);
}
int kf_layernorm(kf_ctx* c, const kf_bf16* x, const kf_bf16* w, const kf_bf16* bias, kf_bf16* y, int rows, int dim, float eps, float* mean, float* rstd) {
    CHKCTX(c);
    if (!x || !w || !y || rows < 1 || dim < 1) return fail(KF_INVALID_ARGS, "kf_layernorm: bad args");
    RET(kf::layernorm_launch(c->stream, x, w, bias, y, rows, dim, eps, mean, rstd));
}
int kf_gelu(kf_ctx* c, const kf_bf16* x, kf_bf16* y, size_t n) {
    CHKCTX(c);
    if (!x || !y || n == 0) return fail(KF_INVALID_ARGS, "kf_gelu: bad args");
    RET(kf::gelu_launch(c->stream, x, y, n));
}
// scratch layout of kf_linear_backward: [W bf16 OC*IC][middle][bias slabs fp64 ceil(n/256)*OC], each 256-B aligned; the middle region is either the transposed copies
// [W^T bf16 IC*OC][deltaIn^T bf16 OC*n][inp^T bf16 IC*n] of the small-shape path or the partial-tile slots of the large tile kernel's stream-K form (kf_gemm3.hip)
static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t lbw_mid_bytes(int OC, int IC, int n) {
    const size_t t = up256((size_t)OC * IC * 2) + up256((size_t)OC * n * 2) + up256((size_t)IC * n * 2), k = up256(kf::gemm3_sk_ws_bytes());
    return t > k ? t : k;
}
size_t kf_linear_backward_scratch_bytes(int OC, int IC, int n) {
    if (OC < 1 || IC < 1 || n < 1) return 0;
    return up256((size_t)OC * IC * 2) + lbw_mid_bytes(OC, IC, n) + up256((size_t)((n + 255) / 256) * OC * 8);
}
int kf_linear_backward(kf_ctx* c, const kf_weight* w, const kf_bf16* deltaIn, const kf_bf16* inp, kf_bf16* delta, kf_bf16* gW, kf_bf16* gBias, int n, int accumulate_delta,
                       void* scratch) {
    CHKCTX(c);
    int r = check_weight(w, "kf_linear_backward");
    if (r) return r;
    if (w->qzeros) return fail(KF_UNSUPPORTED_DATATYPE, "kf_linear_backward: AutoAWQ-layout weights are inference-only");
    if (!deltaIn || !scratch || n < 1) return fail(KF_INVALID_ARGS, "kf_linear_backward: null deltaIn / scratch or n < 1");
    if ((gW && !inp) || (!delta && !gW && !gBias)) return fail(KF_INVALID_ARGS, "kf_linear_backward: gW needs inp; nothing to compute");
    const int OC = w->ne0, IC = w->ne1;
    // the MFMA kernels contract over multiples of 64 and want 16-byte aligned rows
    if ((OC % 64) || OC < 128 || (IC % 8) || (gW && ((n % 64) || n < 128)))
        return fail(KF_INVALID_ARGS, "kf_linear_backward: OC (and n, for the weight gradient) must be multiples of 64 and >= 128, IC a multiple of 8 (got %d %d %d)", OC, IC, n);
    if (!al16(deltaIn) || (inp && !al16(inp)) || (delta && !al16(delta)) || (gW && !al16(gW)) || ((uintptr_t)scratch & 255))
        return fail(KF_BLAS_UNALIGN, "kf_linear_backward: tensors must be 16-byte aligned, scratch 256-byte aligned");
    char* p = (char*)scratch;
    uint16_t* Wd = (uint16_t*)p;    p += up256((size_t)OC * IC * 2);
    uint16_t* WdT = (uint16_t*)p;   p += up256((size_t)OC * IC * 2);
    uint16_t* dInT = (uint16_t*)p;  p += up256((size_t)OC * n * 2);
    uint16_t* inpT = (uint16_t*)p;
    double* slabs = (double*)((char*)scratch + up256((size_t)OC * IC * 2) + lbw_mid_bytes(OC, IC, n));
    void* const sk_ws = WdT; /* the middle region: stream-K slots when the large tile kernel takes the shape, the transposed copies otherwise */
    const size_t sk_bytes = lbw_mid_bytes(OC, IC, n);
    if (gBias) {
        r = kf::colsum_add_launch(c->stream, deltaIn, gBias, n, OC, slabs);
        if (r != KF_OK) return fail(r, "kf_linear_backward: bias column sums failed with %d", r);
    }
    if (delta) { /* delta [n, IC] (+)= deltaIn [n, OC] . W [OC, IC]: rows of W^T are contiguous in the contraction index OC */
        const uint16_t* Wsrc = (const uint16_t*)w->data;
        r = KF_OK;
        if (w->type != KF_BF16) r = kf::dequant_launch(c->stream, w, Wd), Wsrc = Wd;
        if (r != KF_OK) return fail(r, "kf_linear_backward: dequantise of the weight failed with %d", r);
        // large shapes: the 256x256 tile kernel reads W as the K-MAJOR operand it already is (contraction over its OC rows): no transpose
        r = kf::gemm3_km_launch(c->stream, Wsrc, IC, true, deltaIn, OC, false, n, IC, OC, delta, IC, nullptr, 1.0f, accumulate_delta ? 1.0f : 0.0f, sk_ws, sk_bytes);
        if (r < 0) return fail(r, "kf_linear_backward: input-gradient GEMM failed with %d", r);
    }
    if (delta && r == 1) {
        r = w->type == KF_BF16 ? kf::dequant_launch(c->stream, w, Wd) : KF_OK;
        if (r == KF_OK) r = kf::transpose_bf16_launch(c->stream, Wd, WdT, OC, IC);
        if (r != KF_OK) return fail(r, "kf_linear_backward: dequantise / transpose of the weight failed with %d", r);
        kf_weight wt;
        memset(&wt, 0, sizeof(wt));
        wt.data = WdT, wt.type = KF_BF16, wt.ne0 = IC, wt.ne1 = OC;
        r = kf::gemm_launch(c->stream, &wt, deltaIn, OC, n, delta, IC, nullptr, 1.0f, accumulate_delta ? 1.0f : 0.0f, nullptr, IC);
        if (r != KF_OK) return fail(r < 0 ? r : KF_INVALID_ARGS, "kf_linear_backward: input-gradient GEMM not covered (%d)", r);
    }
    int rg = 1;
    if (gW) { /* both operands are k-major for the contraction over the n token rows: inp [n][IC] is the "weight" side, deltaIn [n][OC] the "token" side */
        rg = kf::gemm3_km_launch(c->stream, inp, IC, true, deltaIn, OC, true, OC, IC, n, gW, IC, nullptr, 1.0f, 1.0f, sk_ws, sk_bytes);
        if (rg < 0) return fail(rg, "kf_linear_backward: weight-gradient GEMM failed with %d", rg);
    }
    if (gW && rg == 1) { /* gW [OC, IC] += deltaIn^T [OC, n] . inp [n, IC]: "weight" = inp^T [IC, n], "tokens" = the OC rows of deltaIn^T, contraction over n */
        r = kf::transpose_bf16_launch(c->stream, deltaIn, dInT, n, OC);
        if (r == KF_OK) r = kf::transpose_bf16_launch(c->stream, inp, inpT, n, IC);
        if (r != KF_OK) return fail(r, "kf_linear_backward: operand transposes failed with %d", r);
        kf_weight xt;
        memset(&xt, 0, sizeof(xt));
        xt.data = inpT, xt.type = KF_BF16, xt.ne0 = IC, xt.ne1 = n;
        r = kf::gemm_launch(c->stream, &xt, dInT, n, OC, gW, IC, nullptr, 1.0f, 1.0f, nullptr, IC);
        if (r != KF_OK) return fail(r < 0 ? r : KF_INVALID_ARGS, "kf_linear_backward: weight-gradient GEMM not covered (%d)", r);
    }
    return KF_OK;
}
size_t kf_attn_backward_scratch_bytes(int T, int n_head, int n_seq) { return (T < 1 || n_head < 1 || n_seq < 1) ? 0 : sizeof(float) * 2 * (size_t)T * n_head * n_seq; }
int kf_attn_backward(kf_ctx* c, const kf_bf16* q, const kf_bf16* k, const kf_bf16* v, long long ld_qkv, const kf_bf16* o, const kf_bf16* dO, long long ld_o, kf_bf16* dq,
                     kf_bf16* dk, kf_bf16* dv, long long ld_d, int T, int n_head, int n_kv, int hd, int n_seq, void* scratch) {
    CHKCTX(c);
    if (!q || !k || !v || !o || !dO || !dq || !dk || !dv || !scratch || n_seq < 1 || n_kv < 1 || n_head % n_kv) return fail(KF_INVALID_ARGS, "kf_attn_backward: null pointer, n_seq < 1 or n_head not a multiple of n_kv");
    if (!al16(q) || !al16(k) || !al16(v) || !al16(o) || !al16(dO) || !al16(dq) || !al16(dk) || !al16(dv) || (ld_qkv % 8) || (ld_o % 8) || (ld_d % 8) || ((uintptr_t)scratch & 3))
        return fail(KF_BLAS_UNALIGN, "kf_attn_backward: rows must be 16-byte aligned");
    if (ld_qkv < (long long)n_head * hd || ld_o < (long long)n_head * hd || ld_d < (long long)n_head * hd) return fail(KF_INVALID_ARGS, "kf_attn_backward: row stride below n_head * head_dim");
    if (T < 1 || n_head < 1) return fail(KF_INVALID_ARGS, "kf_attn_backward: T / n_head < 1");
    int r = kf::attn_backward_mfma_launch(c->stream, q, k, v, ld_qkv, o, dO, ld_o, dq, dk, dv, ld_d, T, n_head, hd, n_seq, (float*)scratch, n_kv, ld_qkv, ld_d);
    if (r == 1) r = KF_UNSUPPORTED_DATATYPE; /* shape outside the MFMA tile kernels */
    if (r == KF_UNSUPPORTED_DATATYPE) return fail(r, "kf_attn_backward: head_dim %d not covered (64, 128)", hd);
    RET(r);
}
int kf_embed_pos(kf_ctx* c, const kf_bf16* wte, long long ldw, const kf_bf16* wpe, const int32_t* tokens, int B, int T, int C, int V, kf_bf16* out) {
    CHKCTX(c);
    if (!wte || !wpe || !tokens || !out) return fail(KF_INVALID_ARGS, "kf_embed_pos: null pointer");
    if (!al16(wte) || !al16(wpe) || !al16(out)) return fail(KF_BLAS_UNALIGN, "kf_embed_pos: tensors must be 16-byte aligned");
    const int r = kf::embed_pos_launch(c->stream, wte, ldw, wpe, tokens, B, T, C, V, out);
    if (r == KF_INVALID_ARGS) return fail(r, "kf_embed_pos: needs B, T, V >= 1, C a multiple of 8, ldw >= C a multiple of 8 (got %d %d %d %d %lld)", B, T, C, V, ldw);
    RET(r);
}
int kf_argmax_rows_state(kf_ctx* c, const kf_bf16* logits, long long ld, int n, int n_rows, const int32_t* d_seq, int32_t* d_states, int32_t* d_tokens_out, int tokens_stride) {
    CHKCTX(c);
    if (!logits || !d_seq || !d_states) return fail(KF_INVALID_ARGS, "kf_argmax_rows_state: null pointer");
    const int r = kf::argmax_rows_state_launch(c->stream, logits, ld, n, n_rows, d_seq, d_states, d_tokens_out, tokens_stride);
    if (r == KF_INVALID_ARGS) return fail(r, "kf_argmax_rows_state: needs n, n_rows >= 1 and ld >= n");
    RET(r);
}
int kf_copy_blocks(kf_ctx* c, void* const* d_dst_table, size_t dst_offset, const void* src, size_t src_stride, size_t block_bytes, int n_blocks) {
    CHKCTX(c);
    if (!d_dst_table || !src) return fail(KF_INVALID_ARGS, "kf_copy_blocks: null pointer");
    if (!al16(src)) return fail(KF_BLAS_UNALIGN, "kf_copy_blocks: src must be 16-byte aligned");
    const int r = kf::copy_blocks_launch(c->stream, d_dst_table, dst_offset, src, src_stride, block_bytes, n_blocks);
    if (r == KF_INVALID_ARGS) return fail(r, "kf_copy_blocks: 1 .. 65535 blocks, sizes / strides / offset multiples of 16 bytes");
    RET(r);
}
int kf_memset2d(kf_ctx* c, void* p, size_t pitch, int value, size_t width, size_t rows) {
    CHKCTX(c);
    if (!p || width > pitch) return fail(KF_INVALID_ARGS, "kf_memset2d: null pointer or width > pitch");
    if (!width || !rows) return KF_OK;
    RET(hipMemset2DAsync(p, pitch, value, width, rows, c->stream) == hipSuccess ? KF_OK : KF_HIP_CHECK);
}
int kf_embed_backward(kf_ctx* c, kf_bf16* dwte, long long ldw, kf_bf16* dwpe, const kf_bf16* dout, const int32_t* tokens, int B, int T, int C, int V) {
    CHKCTX(c);
    if (!dout || !tokens || (!dwte && !dwpe)) return fail(KF_INVALID_ARGS, "kf_embed_backward: null pointer");
    if (!al16(dout) || (dwte && !al16(dwte)) || (dwpe && !al16(dwpe))) return fail(KF_BLAS_UNALIGN, "kf_embed_backward: tensors must be 16-byte aligned");
    const int r = kf::embed_backward_launch(c->stream, dwte, ldw, dwpe, dout, tokens, B, T, C, V);
    if (r == KF_INVALID_ARGS) return fail(r, "kf_embed_backward: needs B, T, V >= 1, C a multiple of 8 up to 8192, ldw >= C a multiple of 8 (got %d %d %d %d %lld)", B, T, C, V, ldw);
    RET(r);
}
size_t kf_norm_backward_scratch_bytes(int rows, int dim, int is_layernorm) {
    return sizeof(double) * (size_t)kf::norm_backward_groups(rows < 1 ? 1 : rows) * (is_layernorm ? 2 : 1) * (size_t)(dim < 0 ? 0 : dim);
}
int kf_norm_backward(kf_ctx* c, kf_bf16* dinp, kf_bf16* dweight, kf_bf16* dbias, const kf_bf16* dout, const kf_bf16* inp, const kf_bf16* weight, const float* mean,
                     const float* rstd, int rows, int dim, void* scratch) {
    CHKCTX(c);
    if (!dinp || !dweight || !dout || !inp || !weight || !rstd || !scratch) return fail(KF_INVALID_ARGS, "kf_norm_backward: null pointer");
    if (rows < 1 || dim < 8 || (dim % 8) != 0 || dim > 8192) return fail(KF_INVALID_ARGS, "kf_norm_backward: needs rows >= 1 and dim a multiple of 8 up to 8192 (got %d x %d)", rows, dim);
    if (!al16(dinp) || !al16(dout) || !al16(inp) || !al16(weight) || ((uintptr_t)scratch & 7)) return fail(KF_BLAS_UNALIGN, "kf_norm_backward: tensors must be 16-byte aligned");
    RET(kf::norm_backward_launch(c->stream, dinp, dweight, dbias, dout, inp, weight, mean, rstd, rows, dim, (double*)scratch));
}
int kf_rope_backward(kf_ctx* c, kf_bf16* d, const float* rope_table, int pos0, int n_tok, int seq_len, long long stride, int n_head, int hd) {
    CHKCTX(c);
    if (!d || !rope_table || pos0 < 0) return fail(KF_INVALID_ARGS, "kf_rope_backward: bad args");
    RET(kf::rope_backward_launch(c->stream, d, rope_table, pos0, n_tok, seq_len, stride, n_head, hd));
}
int kf_gelu_backward(kf_ctx* c, kf_bf16* d_in_out, const kf_bf16* x, size_t n) {
    CHKCTX(c);
    if (!d_in_out || !x || n == 0) return fail(KF_INVALID_ARGS, "kf_gelu_backward: bad args");
    RET(kf::gelu_backward_launch(c->stream, d_in_out, x, n));
}
int kf_swiglu_backward(kf_ctx* c, kf_bf16* delta_in_out, kf_bf16* delta_gate, const kf_bf16* gate, const kf_bf16* up, size_t n) {
    CHKCTX(c);
    if (!delta_in_out || !delta_gate || !gate || !up || n == 0) return fail(KF_INVALID_ARGS, "kf_swiglu_backward: bad args");
    RET(kf::swiglu_backward_launch(c->stream, delta_in_out, delta_gate, gate, up, n));
}
int kf_fused_classifier(kf_ctx* c, kf_bf16* logits, float* losses, kf_bf16* probs, float dloss, const int32_t* targets, int B, int T, int V, int P,
                        const int32_t* mask, int write_dlogits) {
    CHKCTX(c);
    if (!logits || !losses || !targets) return fail(KF_INVALID_ARGS, "kf_fused_classifier: null pointer");
    if (B < 1 || T < 1 || V < 1 || P < V) return fail(KF_INVALID_ARGS, "kf_fused_classifier: needs B, T, V >= 1 and P >= V (got %d %d %d %d)", B, T, V, P);
    if ((P % 8) != 0 || !al16(logits) || (probs && !al16(probs)))
        return fail(KF_BLAS_UNALIGN, "kf_fused_classifier: rows must be 16-byte aligned (P = %d a multiple of 8, like the reference's padded vocabulary)", P);
    RET(kf::fused_classifier_launch(c->stream, logits, losses, probs, dloss, targets, (long)B * T, V, P, mask, write_dlogits));
}
int kf_adamw(kf_ctx* c, kf_bf16* params, kf_bf16* grads, void* gm, void* gv, size_t n, int mv_type, float learning_rate, float beta1, float beta2,
             float beta1_correction, float beta2_correction, float eps, float weight_decay, float grad_scale, uint32_t seed, int32_t* d_status) {
    CHKCTX(c);
    if (!params || !grads || !gm || !gv) return fail(KF_INVALID_ARGS, "kf_adamw: null pointer");
    if (mv_type != KF_BF16 && mv_type != KF_F32) return fail(KF_UNSUPPORTED_DATATYPE, "kf_adamw: moments must be bf16 or f32 (got %d)", mv_type);
    if (n == 0 || n % 8) return fail(KF_INVALID_ARGS, "kf_adamw: n = %zu is not a positive multiple of 8 (TASKA_1p1 asserts N %% typ128::size == 0)", n);
    if (!al16(params) || !al16(grads) || !al16(gm) || !al16(gv)) return fail(KF_BLAS_UNALIGN, "kf_adamw: tensors must be 16-byte aligned");
    RET(kf::adamw_launch(c->stream, params, grads, gm, gv, n, mv_type == KF_BF16, learning_rate, beta1, beta2, beta1_correction, beta2_correction, eps,
                         weight_decay, grad_scale, seed, d_status));
}

// ---- persistent decode engine
struct kf_engine {
    kf::EngineHost* h;
};
size_t kf_engine_workspace_bytes(const kf_engine_desc* d) { return d ? kf::engine_ws_bytes(d) : 0; }
int kf_engine_create(kf_ctx* c, const kf_engine_desc* d, void* ws, size_t ws_bytes, kf_engine** out) {
    CHKCTX(c);
    if (!d || !ws || !out) return fail(KF_INVALID_ARGS, "kf_engine_create: null argument");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_engine_create: not while capturing");
    kf::EngineHost* h = nullptr;
    const char* why = "";
    const int rc = kf::engine_build(d, ws, ws_bytes, c->stream, &h, &why);
    if (rc != KF_OK) return fail(rc, "kf_engine_create: %s", why);
    kf_engine* e = new kf_engine();
    e->h = h;
    *out = e;
    return KF_OK;
}
int kf_engine_step(kf_ctx* c, kf_engine* e, const kf_bf16* x_in, kf_bf16* x_out, const int32_t* d_state, int pos_bound) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_engine_step: null engine");
    kf::engine_set_canonical(e->h, c->canonical);
    const int rc = kf::engine_step(e->h, c->stream, x_in, x_out, d_state, pos_bound);
    if (rc < 0) return fail(rc, "kf_engine_step failed with %d", rc);
    return rc;
}
int kf_engine_step_head(kf_ctx* c, kf_engine* e, const kf_bf16* x_in, kf_bf16* x_out, int32_t* d_state, int pos_bound, int pick) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_engine_step_head: null engine");
    kf::engine_set_canonical(e->h, c->canonical);
    const int rc = kf::engine_step(e->h, c->stream, x_in, x_out, d_state, pos_bound, pick ? 2 : 1);
    if (rc < 0) return fail(rc, "kf_engine_step_head failed with %d (no head set?)", rc);
    return rc;
}
int kf_engine_steps_head(kf_ctx* c, kf_engine* e, kf_bf16* x_out, int32_t* d_state, int pos_bound, int n_steps) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_engine_steps_head: null engine");
    if (n_steps < 1) return fail(KF_INVALID_ARGS, "kf_engine_steps_head: n_steps %d", n_steps);
    kf::engine_set_canonical(e->h, c->canonical);
    const int rc = kf::engine_step(e->h, c->stream, nullptr, x_out, d_state, pos_bound, 2, n_steps);
    if (rc < 0) return fail(rc, "kf_engine_steps_head failed with %d (no head / embedding set?)", rc);
    return rc;
}
int kf_engine_set_head(kf_ctx* c, kf_engine* e, const kf_weight* head_or_null, const kf_bf16* final_norm_w, kf_bf16* logits, int32_t* d_tokens_out) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_engine_set_head: null engine");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_engine_set_head: not while capturing");
    const int rc = kf::engine_set_head(e->h, head_or_null, final_norm_w, logits, d_tokens_out);
    if (rc != KF_OK) return fail(rc, "kf_engine_set_head: the in-launch head takes a bf16 [vocab, dim] matrix of the engine's width, a norm weight and a logits buffer");
    return KF_OK;
}
int kf_engine_reset(kf_ctx* c, kf_engine* e) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_engine_reset: null engine");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_engine_reset: not while capturing");
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(KF_HIP_CHECK, "kf_engine_reset: HIP failure");
    const int rc = kf::engine_reset(e->h, c->stream);
    if (rc != KF_OK) return fail(rc, "kf_engine_reset: HIP failure");
    return KF_OK;
}
int kf_engine_set_embedding(kf_ctx* c, kf_engine* e, const kf_weight* embed_or_null, const int32_t* d_forced) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_engine_set_embedding: null engine");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_engine_set_embedding: not while capturing");
    const int rc = kf::engine_set_embedding(e->h, embed_or_null, d_forced);
    if (rc != KF_OK) return fail(rc, "kf_engine_set_embedding: the fused row read takes a bf16 table of the engine's width (other storages keep kf_embed_state)");
    return KF_OK;
}
int kf_engine_served(kf_ctx* c, const kf_engine_desc* d, char* why, size_t why_bytes) {
    CHKCTX(c);
    const char* w = "";
    const int rc = kf::engine_build(d, nullptr, 0, c->stream, nullptr, &w, true);
    if (why && why_bytes > 0) snprintf(why, why_bytes, "%s", w);
    if (rc == KF_OK) return KF_OK;
    return rc == KF_UNSUPPORTED_DATATYPE ? KF_ENGINE_NOT_SERVED : fail(rc, "kf_engine_served: %s", w);
}
int kf_engine_tune(kf_ctx* c, kf_engine* e, kf_bf16* x_out, const int32_t* d_state, int pos_bound, int passes, float* us_before, float* us_after) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_engine_tune: null engine");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_engine_tune: not while capturing");
    kf::engine_set_canonical(e->h, c->canonical);
    const int rc = kf::engine_tune(e->h, c->stream, x_out, d_state, pos_bound, passes, us_before, us_after);
    if (rc < 0) return fail(rc, "kf_engine_tune failed with %d (no embedding table set, or a poll timed out)", rc);
    return rc;
}
int kf_engine_stats(kf_ctx* c, kf_engine* e, int pos_bound, kf_engine_statistics* out) {
    CHKCTX(c);
    if (!e || !e->h || !out) return fail(KF_INVALID_ARGS, "kf_engine_stats: null argument");
    int w[14];
    const int rc = kf::engine_stats(e->h, c->stream, pos_bound, w);
    if (rc != KF_OK) return fail(rc, "kf_engine_stats: HIP failure");
    for (int i = 0; i < 6; i++) out->sweeps[i] = w[i], out->delay[i] = w[7 + i];
    out->polls = w[6], out->tuned = w[13];
    return KF_OK;
}
int kf_engine_check(kf_ctx* c, kf_engine* e) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_engine_check: null engine");
    int err = 0;
    const int rc = kf::engine_error_word(e->h, c->stream, &err);
    if (rc != KF_OK) return fail(rc, "kf_engine_check: HIP failure");
    if (err) return fail(KF_INTERNAL_ERR, "kf_engine: a hand-off poll timed out (error word 0x%x): the launch was not fully resident", err);
    return KF_OK;
}
// ---- XCD-confined decode engines (kf_xengine.hip)
struct kf_xengine {
    kf::XEngineHost* h;
};
size_t kf_xengine_workspace_bytes(const kf_engine_desc* d) { return d ? kf::xengine_ws_bytes(d) : 0; }
int kf_xengine_create(kf_ctx* c, const kf_engine_desc* d, int n_seq, int64_t kv_seq_stride, void* ws, size_t ws_bytes, kf_xengine** out) {
    CHKCTX(c);
    if (!d || !ws || !out) return fail(KF_INVALID_ARGS, "kf_xengine_create: null argument");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_xengine_create: not while capturing");
    if (n_seq < 1 || n_seq > KF_XENGINE_MAX_SEQ) return fail(KF_INVALID_ARGS, "kf_xengine_create: n_seq %d outside 1 .. %d (up to four sequences per XCD)", n_seq, KF_XENGINE_MAX_SEQ);
    kf::XEngineHost* h = nullptr;
    const char* why = "";
    const int rc = kf::xengine_build(d, n_seq, (long long)kv_seq_stride, ws, ws_bytes, c->stream, &h, &why);
    if (rc != KF_OK) return fail(rc, "kf_xengine_create: %s", why);
    kf_xengine* e = new kf_xengine();
    e->h = h;
    *out = e;
    return KF_OK;
}
size_t kf_xengine_workspace_bytes_tp(const kf_engine_desc* d) { return d ? kf::xengine_ws_bytes_tp(d) : 0; }
int kf_xengine_create_tp(kf_ctx* c, const kf_engine_desc* const* ds, int world, void* ws, size_t ws_bytes, kf_xengine** out) {
    CHKCTX(c);
    if (!ds || !ws || !out) return fail(KF_INVALID_ARGS, "kf_xengine_create_tp: null argument");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_xengine_create_tp: not while capturing");
    kf::XEngineHost* h = nullptr;
    const char* why = "";
    const int rc = kf::xengine_build_tp(ds, world, ws, ws_bytes, c->stream, &h, &why);
    if (rc != KF_OK) return fail(rc, "kf_xengine_create_tp: %s", why);
    kf_xengine* e = new kf_xengine();
    e->h = h;
    *out = e;
    return KF_OK;
}
int kf_xengine_set_head_tp(kf_ctx* c, kf_xengine* e, const kf_weight* const* shards, const int32_t* row0, const kf_bf16* final_norm_w, kf_bf16* logits, int32_t* d_tokens_out, int tokens_stride) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_xengine_set_head_tp: null engine");
    const int rc = kf::xengine_set_head_tp(e->h, shards, row0, final_norm_w, logits, d_tokens_out, tokens_stride);
    if (rc != KF_OK) return fail(rc, "kf_xengine_set_head_tp: eight bf16 [rows, dim] vocabulary shards in rank order (row0 = the rows before), a norm weight and a logits buffer");
    return KF_OK;
}
int kf_xengine_served(kf_ctx* c, const kf_engine_desc* d, char* why, size_t why_bytes) {
    CHKCTX(c);
    if (!d) return fail(KF_INVALID_ARGS, "kf_xengine_served: null descriptor");
    const char* w = "";
    const int rc = kf::xengine_build(d, 1, 0, nullptr, 0, c->stream, nullptr, &w, true);
    if (why && why_bytes) snprintf(why, why_bytes, "%s", w);
    if (rc == KF_OK) return KF_OK;
    return rc == KF_UNSUPPORTED_DATATYPE ? KF_ENGINE_NOT_SERVED : fail(rc, "kf_xengine_served: %s", w);
}
int kf_xengine_set_embedding(kf_ctx* c, kf_xengine* e, const kf_weight* embed, const int32_t* d_forced, int forced_stride) {
    CHKCTX(c);
    if (!e || !e->h || !embed) return fail(KF_INVALID_ARGS, "kf_xengine_set_embedding: null argument");
    if (d_forced && forced_stride < 1) return fail(KF_INVALID_ARGS, "kf_xengine_set_embedding: forced_stride %d", forced_stride);
    const int rc = kf::xengine_set_embedding(e->h, embed, d_forced, forced_stride);
    if (rc != KF_OK) return fail(rc, "kf_xengine_set_embedding: the row read inside the launch takes a bf16 table of the engine's width");
    return KF_OK;
}
int kf_xengine_set_head(kf_ctx* c, kf_xengine* e, const kf_weight* head_or_null, const kf_bf16* final_norm_w, kf_bf16* logits, int32_t* d_tokens_out, int tokens_stride) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_xengine_set_head: null engine");
    const int rc = kf::xengine_set_head(e->h, head_or_null, final_norm_w, logits, d_tokens_out, tokens_stride);
    if (rc != KF_OK) return fail(rc, "kf_xengine_set_head: the in-launch head takes a bf16 [vocab, dim] matrix of the engine's width, a norm weight and a logits buffer");
    return KF_OK;
}
int kf_xengine_steps(kf_ctx* c, kf_xengine* e, kf_bf16* x_out, int32_t* d_state, int n_steps, int pick) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_xengine_steps: null engine");
    if (!c->canonical) return fail(KF_UNSUPPORTED_DATATYPE, "kf_xengine_steps: the XCD-confined engines run the canonical summation order only (kf_set_canonical(ctx, 1))");
    if (n_steps < 1 || (n_steps > 1 && !pick)) return fail(KF_INVALID_ARGS, "kf_xengine_steps: n_steps %d (several steps per launch need the pick inside)", n_steps);
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_xengine_steps: not while capturing (the launch's generation is a kernel argument the host counts: a replayed launch would repeat it)");
    const int rc = kf::xengine_steps(e->h, c->stream, d_state, x_out, pick ? 2 : 1, n_steps);
    if (rc == KF_UNSUPPORTED_DATATYPE) return fail(rc, "kf_xengine_steps: the form for this many sequences does not fit the LDS at this depth");
    if (rc != KF_OK) return fail(rc, "kf_xengine_steps failed with %d (no embedding / head set?)", rc);
    return KF_OK;
}
int kf_xengine_check(kf_ctx* c, kf_xengine* e) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_xengine_check: null engine");
    int err = 0;
    const int rc = kf::xengine_error_word(e->h, c->stream, &err);
    if (rc != KF_OK) return fail(rc, "kf_xengine_check: HIP failure");
    if (err) return fail(KF_INTERNAL_ERR, "kf_xengine: error word 0x%x (8: an XCD did not get 32 workgroups; 64: a position beyond the cache rows (TP form); others: a hand-off poll timed out -- the launch was not fully resident)", err);
    return KF_OK;
}
int kf_xengine_reset(kf_ctx* c, kf_xengine* e) {
    CHKCTX(c);
    if (!e || !e->h) return fail(KF_INVALID_ARGS, "kf_xengine_reset: null engine");
    if (c->capturing) return fail(KF_INVALID_ARGS, "kf_xengine_reset: not while capturing");
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(KF_HIP_CHECK, "kf_xengine_reset: HIP failure");
    const int rc = kf::xengine_reset(e->h, c->stream);
    if (rc != KF_OK) return fail(rc, "kf_xengine_reset: HIP failure");
    return KF_OK;
}
int kf_xengine_destroy(kf_xengine* e) {
    if (e) {
        kf::xengine_free(e->h);
        delete e;
    }
    return KF_OK;
}
int kfdbg_xengine_variant(kf_xengine* e, int nwv, int depth) {
    if (!e || !e->h) return -1;
    kf::xengine_set_variant(e->h, nwv, depth);
    return 0;
}
int kfdbg_xengine_stamps_enable(kf_xengine* e, int seq, int wg, int max_steps) { return (e && e->h) ? kf::xengine_debug_enable(e->h, seq, wg, max_steps) : -1; }
int kfdbg_xengine_stamps(kf_xengine* e, unsigned long long* h_out, int n_words) { return (e && e->h) ? kf::xengine_debug_read(e->h, h_out, n_words) : -1; }
// ---- diagnostics (NOT part of the ABI header; tests and scratch/ only)
// the per-phase stamps of one workgroup of the engine: enable (the diagnostic instantiation of the kernel serves the next launches), read
int kfdbg_engine_stamps_enable(kf_engine* e, int wg) { return (e && e->h) ? kf::engine_debug_enable(e->h, wg) : -1; }
int kfdbg_engine_stamps(kf_engine* e, unsigned long long* h_out, int n_words) { return (e && e->h) ? kf::engine_debug_read(e->h, h_out, n_words) : -1; }
int kfdbg_engine_set_delays(kf_engine* e, const int* d6) {
    if (!e || !e->h || !d6) return -1;
    kf::engine_set_delays(e->h, d6);
    return 0;
}
// development knobs (kf::Knobs): a kernel form against the form it replaces, inside one process
int kfdbg_set_knob(const char* name, long value) {
    if (!name) return -1;
    kf::Knobs& k = kf::g_knobs;
    if (!strcmp(name, "q4_perm")) k.q4_perm = (int)value;
    else if (!strcmp(name, "q2_tab")) k.q2_tab = (int)value;
    else if (!strcmp(name, "q1_tab")) k.q1_tab = (int)value;
    else if (!strcmp(name, "gemv_waves")) k.gemv_waves = value;
    else if (!strcmp(name, "gemv_stream")) k.gemv_stream = (int)value;
    else if (!strcmp(name, "gemv_xf2")) k.gemv_xf2 = (int)value;
    else if (!strcmp(name, "gemm_min")) k.gemm_min = (int)value;
    else if (!strcmp(name, "g3_tiles")) k.g3_tiles = (int)value;
    else if (!strcmp(name, "g3_first")) k.g3_first = (int)value;
    else if (!strcmp(name, "g3_wide")) k.g3_wide = (int)value;
    else if (!strcmp(name, "attn_gq_split")) k.attn_gq_split = (int)value;
    else if (!strcmp(name, "g3_mid_min")) k.g3_mid_min = (int)value;
    else if (!strcmp(name, "resident_min")) k.resident_min = (int)value;
    else if (!strcmp(name, "attn_pair_min")) k.attn_pair_min = (int)value;
    else return -1;
    return 0;
}
int kf_engine_destroy(kf_engine* e) {
    if (e) {
        kf::engine_free(e->h);
        delete e;
    }
    return KF_OK;
}

}  // extern "C"
