// kf_kernels.h -- internal launch interfaces between the ABI layer (kf_abi.hip) and the kernel files.
#pragma once
#include "../../include/kf_abi.h"
#include "kf_device.h"

namespace kf {

enum {
    FMT_BF16 = 0, FMT_F8 = 1, FMT_Q4 = 2, FMT_Q2 = 3, FMT_Q1 = 4,
    FMT_Q4P = 5, /* 4-bit through the register-table lookup (mat-vec only) */
    FMT_Q4R = 6, /* 4-bit row codebook (KF_QUANT_ROW_LUT): byte-packed nibbles + 16 bf16 table entries per row */
    FMT_Q1T = 7, /* 1-bit through an LDS table of v_perm selectors (mat-vec only; bit-identical to FMT_Q1) */
    FMT_Q2T = 8  /* 2-bit, the same way (bit-identical to FMT_Q2) */
};
inline bool is_row_lut(const kf_weight* w) { return w->quant != KF_QUANT_GROUP; } /* any row-wise card (kf_lut.hip); the mat-vec kernel takes Q4 + ROW_LUT only */
enum { GEMV_PLAIN = 0, GEMV_PAIRED = 1, GEMV_ARGMAX = 2 };
constexpr int KF_MAX_ARGMAX_PARTIALS = 4096;
constexpr int KF_ATTN_MAX_SPLITS = 32;
constexpr int KF_ATTN_CNT_BYTES = 16384; /* arrival counters at the head of the attention scratch */

struct GemvJob {
    const void* w;
    const uint16_t* zero;
    const uint16_t* step;
    uint16_t* y;
    long long y_pos_stride; /* elements added per position (KV-cache row aliasing), 0 otherwise */
    int M;
    int qBias;
    int slot0;
};

struct TpPushDev { /* one per (rank, exchange index), written once by kf_tp_commit */
    unsigned long long* peer[8]; /* this rank's slot of the exchange's buffer in every rank's receive area */
    const unsigned* step;        /* device word: generation */
    unsigned per_step, index;
    int world, pad_;
};
struct GemvArgs {
    GemvJob job[3];
    int njobs;
    int K, nBlk, lpr_log2, iters, lGroup, gshift;
    int spw, total_slots;
    const uint16_t* x;
    const uint16_t* norm_w; /* non-NULL: RMSNorm prologue */
    float eps, inv_dim;
    const uint16_t* residual;
    const uint16_t* bias;
    float* yf; /* non-NULL: write fp32 row dots here instead of bf16 outputs (single job) */
    float alpha, beta;
    const int* d_pos;
    int pos;
    float* amax_val;
    int* amax_idx;
    // tensor-parallel push (kf_linear_f32_push): the un-rounded fp32 row dot goes, tagged, into this rank's slot of every rank's receive area.  ONE pointer to a
    // descriptor in device memory: kernel arguments are fetched before a launch's first load, and the 96 bytes of the descriptor inside this struct cost every
    // mat-vec launch of the decode step 0.37 us (measured: 0.742 -> 0.782 ms per step on the per-layer path).
    const struct TpPushDev* tp; /* NULL: no push */
    int stream_ok;          /* long dense launches: buffer-load form allowed (offsets < 2^31, groups inside rows) */
    const int32_t* row_map; /* non-NULL: the sparse forward -- slot rows index this list of hot rows (job.M = their number); weights and outputs use row_map[row] */
};

struct GemvLaunch {
    GemvArgs args;
    const kf_weight* w[3];
    int n;
    int mode;
    long target_waves; /* 0: default */
    int n_hot;         /* args.row_map != NULL: number of entries */
    int canon;         /* the canonical summation order (one v_pk_fma_f32 per weight pair: an even and an odd chain per lane; oracle/kf_oracle.c section 4c) instead of v_dot2c_f32_bf16 */
    int blocks;        /* out */
};

int gemv_launch(hipStream_t st, GemvLaunch& L);
int gemv_launch_dot2(hipStream_t st, GemvLaunch& L);  /* kf_gemv.hip */
int gemv_launch_canon(hipStream_t st, GemvLaunch& L); /* kf_gemv_canon.hip: the same file, canonical instantiation */
// tensor-parallel exchange (kf_tp.hip)
int tp_reduce_recv_launch(hipStream_t st, const unsigned long long* slots, int R, int n_max, int n, const unsigned* d_step, unsigned per_step, unsigned index,
                          const uint16_t* residual, uint16_t* out, int* d_err);
int tp_argmax_push_launch(hipStream_t st, const float* val, const int* idx, int n, int row0, unsigned long long* const* peers, int R, const unsigned* d_step,
                          unsigned per_step, unsigned index);
int tp_pick_launch(hipStream_t st, const unsigned long long* pairs, int R, unsigned* d_step, unsigned per_step, unsigned index, int32_t* d_state, int32_t* d_tokens_out,
                   int* d_err, int vocab);
int gemv_lpr_log2(int nBlk, long rows); /* lanes per row of a mat-vec launch (kf_gemv.hip) */
int gemv_lpr_log2_fmt(int fmt, int K, long rows); /* the same per storage (1-bit: K / 32 virtual blocks, divided by four) */
int gemv_fmt_of(const kf_weight* w);    /* FMT_* of a weight, < 0: not served by the mat-vec kernel */
void argmax_finish_launch(hipStream_t st, const float* val, const int* idx, int n, int32_t* d_argmax, int32_t* d_state, int32_t* d_tokens_out);

// ---- token-batch GEMM on MFMA (kf_gemm.hip): KF_OK launched, 1 = shape not eligible (caller loops the mat-vec), < 0 error
// the 256 x 256 tile kernel on K-MAJOR operands (kf_gemm3.hip): the two GEMMs of SLP::Back without transposes.  1 = shape not served
int gemm3_km_launch(hipStream_t st, const uint16_t* A, long long lda, bool akm, const uint16_t* B, long long ldb, bool bkm, int n, int M, int K, uint16_t* y, long long ldy,
                    const uint16_t* bias, float alpha, float beta, void* ws, size_t ws_bytes);
size_t gemm3_sk_ws_bytes();
struct G3Rope { /* ROPE::cuInfer folded into the stacked Q | K | V launch's epilogue (kf_gemm3.hip g3_epilogue_qkrope) */
    const uint16_t *wq, *wk; /* q / k norm weights [128] or NULL */
    const float* table;      /* RoPE (cos, sin) table or NULL */
    int pos0;
    float eps;
    int seq_len; /* > 0: rows are sequences of seq_len tokens back to back, positions pos0 .. pos0 + seq_len - 1 in each */
};
int gemm3_multi_launch(hipStream_t st, int n_w, const uint16_t* Wcat, const int* M, int K, const uint16_t* x, long long ldx, int n, uint16_t* const* y, const G3Rope* rope = nullptr);
int gemm3_swiglu_launch(hipStream_t st, const uint16_t* Wilv, int ffn, int K, const uint16_t* x, long long ldx, int n, uint16_t* act); /* kf_gemm3.hip */
int gemm_launch(hipStream_t st, const kf_weight* w, const uint16_t* x, long long ldx, int n, uint16_t* y, long long ldy, const uint16_t* bias, float alpha,
                float beta, const uint16_t* residual, long long ldr, void* ws = nullptr, size_t ws_bytes = 0); /* ws: split-K slots of the small bf16 tiles (gemm3_sk_ws_bytes) */

int gemm_multi_launch(hipStream_t st, int n_w, const kf_weight* const* w, const uint16_t* x, long long ldx, int n, uint16_t* const* y);
int gemm_paired_launch(hipStream_t st, const kf_weight* gate, const kf_weight* up, const uint16_t* x, long long ldx, int n, uint16_t* act);

// ---- attention (kf_attn.hip)
struct AttnArgs {
    const uint16_t* q;     /* raw or prepared q [n_head*hd] */
    const uint16_t* k_raw; /* NULL: keys come from the cache only */
    uint16_t* kcache;
    const uint16_t* vcache;
    const uint16_t* wq_norm;
    const uint16_t* wk_norm;
    const float* rope_table; /* NULL: q is already normed+roped */
    float* part;             /* [n_head][n_splits][hd + 2] fp64 partials {O, L, m} (kf_attn.hip reads it as doubles) */
    int* counters;           /* [n_kv] arrival counters, zero between launches */
    uint16_t* out;
    const int* d_pos;
    int pos;
    int n_head, n_kv, hd, kv_stride, n_splits, chunk, cnt_stride;
    float eps, inv_sqrt_hd_den;
    int n_tok;          /* token batch (prefill): position pos + token */
    int one_slice;      /* every (kv-head, token) is handled by one workgroup: no scratch, no hand-off */
    long long q_stride; /* elements between the q (and out) rows of consecutive tokens */
    int canon;          /* the canonical softmax (fp64 sums, exact rescales; bit-exact against the oracle's CANON mode) instead of the fp32 form */
    int gq_split;       /* set by attn_launch: the query heads of a kv-head are dealt to this many workgroups (blockIdx.y = kv-head * gq_split + part), each with GQ / gq_split heads:
                           the canonical form at 8 query heads per kv-head needs 411 registers in one workgroup (one wave per SIMD, accumulators in AGPRs), 224 in two */
};
int attn_launch(hipStream_t st, AttnArgs& a);
int attn_splits(int pos_bound, int n_kv);
// token-batch causal attention on MFMA (kf_attn_prefill.hip): KF_OK launched, 1 = shape not covered, < 0 error
int attn_prefill_mfma_launch(hipStream_t st, const uint16_t* q, const uint16_t* kc, const uint16_t* vc, uint16_t* out, int pos0, int n_tok, long long q_stride,
                             int n_head, int n_kv, int hd, int kv_stride, int n_seq = 1, long long out_stride = 0 /* 0: q_stride */);
int qknorm_rope_launch(hipStream_t st, uint16_t* q, uint16_t* k, const uint16_t* wq, const uint16_t* wk, const float* table, int pos,
                       const int* d_pos, int n_head, int n_kv, int hd, float eps, int n_tok = 1, long long q_stride = 0, long long k_stride = 0, int seq_len = 0,
                       float* rstd_q = nullptr, float* rstd_k = nullptr);

// ---- persistent decode engine (kf_engine.hip)
struct EngineHost;
// Development knobs (process-wide, default = the product's choice).  Not part of the ABI and never read from the environment: tests and the scripts under
// scratch/ set them through kfdbg_set_knob (kf_abi.hip) to compare a kernel form with the form it replaces inside one process.
struct Knobs {
    int q4_perm = 1;      /* 4-bit mat-vec through the register-table lookup (0: the arithmetic form; same bits) */
    int q2_tab = 1;       /* 2-bit mat-vec through the LDS selector table (0: the arithmetic form; same bits) */
    int g3_tiles = 3;     /* smallest bf16 tile gemm3_launch may pick: 0 = 128 x 128 only (round 3), 1 = + 64 x 128, 3 = + 64 x 64 (kf_gemm3.hip) */
    int resident_min = 320;  /* token rows from which the token-batch routes use RESIDENT dequantised copies (kf_set_dequant_arena) + the bf16 tile kernels; without an arena: 1024 */
    int attn_pair_min = 256;  /* prompt tokens from which kf_attn_prefill takes its paired two-key-half form (when there is about one workgroup per CU or fewer) */
    int g3_wide = 1;      /* gate | up + SwiGLU on 192 x 256 tiles when the 256 x 256 ones would leave CUs idle */
    int g3_mid_min = 192; /* 64 x 128 tiles from this many of them, 64 x 64 below */
    int attn_gq_split = 4; /* canonical decode attention of a GQA-8 model: workgroups per (kv-head, slice), 2 or 4 */
    int g3_first = 256;   /* token rows from which bf16 operands try the kf_gemm3.hip tile kernels before the 32 x 32 direct kernel */
    int q1_tab = 1;       /* 1-bit mat-vec through the LDS selector table (0: the per-bit select form; same bits) */
    long gemv_waves = 0;  /* > 0: waves a mat-vec launch aims for (0: the launcher's rule) */
    int gemv_stream = 1;  /* buffer-load form of the long mat-vec launches (0: off) */
    int gemv_xf2 = 1;     /* canonical 4-bit rows too long for fp32 activations in 48 KiB of LDS: two windows of half the block columns (0: bf16 activations, widened per product) */
    int gemm_min = 8;     /* token rows from which the MFMA tile kernels replace the per-token mat-vec loop */
};
extern Knobs g_knobs;
size_t engine_ws_bytes(const kf_engine_desc* d);
int engine_build(const kf_engine_desc* d, void* ws, size_t ws_bytes, hipStream_t st, EngineHost** out, const char** why = nullptr, bool dry = false);
int engine_tune(EngineHost* E, hipStream_t st, uint16_t* x_out, const int32_t* d_state, int pos_bound, int passes, float* us_before, float* us_after);
int engine_stats(EngineHost* E, hipStream_t st, int pos_bound, int* out14);
int engine_step(EngineHost* E, hipStream_t st, const uint16_t* x_in, uint16_t* x_out, const int32_t* d_state, int pos_bound, int with_head = 0, int n_steps = 1); /* 1: not served */
int engine_set_head(EngineHost* E, const kf_weight* w, const uint16_t* norm_w, uint16_t* logits, int32_t* d_tokens_out);
int engine_set_embedding(EngineHost* E, const kf_weight* w, const int32_t* d_forced);
int engine_error_word(EngineHost* E, hipStream_t st, int* h_err);
void engine_set_canonical(EngineHost* E, int on); /* every phase: canonical order (1, the default) or the v_dot2c / fp32 forms */
int engine_reset(EngineHost* E, hipStream_t st); /* after a timed-out poll: exchange state re-initialised, error word cleared */
void engine_free(EngineHost* E);
int engine_debug_read(EngineHost* E, unsigned long long* h_out, int n_words);
int engine_debug_enable(EngineHost* E, int wg);       /* per-phase stamps of workgroup wg from the next launch on (the diagnostic instantiation of the kernel) */
void engine_set_delays(EngineHost* E, const int* d6); /* tuning runs */

// ---- XCD-confined decode engines: up to eight independent sequences per launch, one per XCD (kf_xengine.hip)
struct XEngineHost;
size_t xengine_ws_bytes(const kf_engine_desc* d);
int xengine_build(const kf_engine_desc* d, int n_seq, long long kv_seq_stride, void* ws, size_t ws_bytes, hipStream_t st, XEngineHost** out, const char** why = nullptr, bool dry = false);
int xengine_steps(XEngineHost* E, hipStream_t st, int32_t* d_state, uint16_t* x_out, int with_head, int n_steps);
size_t xengine_ws_bytes_tp(const kf_engine_desc* rank0);                                /* tensor parallel over the XCDs: one sequence, rank r on XCD r */
int xengine_build_tp(const kf_engine_desc* const* ranks, int world, void* ws, size_t ws_bytes, hipStream_t st, XEngineHost** out, const char** why = nullptr);
int xengine_set_head_tp(XEngineHost* E, const kf_weight* const* shards, const int* row0, const uint16_t* norm_w, uint16_t* logits, int32_t* d_tokens_out, int tokens_stride);
int xengine_set_embedding(XEngineHost* E, const kf_weight* w, const int32_t* d_forced, int forced_stride);
int xengine_set_head(XEngineHost* E, const kf_weight* w, const uint16_t* norm_w, uint16_t* logits, int32_t* d_tokens_out, int tokens_stride);
int xengine_error_word(XEngineHost* E, hipStream_t st, int* h_err);
int xengine_reset(XEngineHost* E, hipStream_t st);
void xengine_free(XEngineHost* E);
void xengine_set_variant(XEngineHost* E, int nwv, int depth);             /* tuning runs: waves per workgroup x ring depth (instantiated pairs only) */
int xengine_debug_enable(XEngineHost* E, int seq, int wg, int max_steps); /* per-phase stamps of one workgroup of one decoder (seq < 0: off) */
int xengine_debug_read(XEngineHost* E, unsigned long long* h_out, int n_words);

// ---- small ops (kf_ops.hip)
int rmsnorm_launch(hipStream_t st, const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int dim, float eps, float* rstd);
int layernorm_launch(hipStream_t st, const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* y, int rows, int dim, float eps, float* mean, float* rstd);
int gelu_launch(hipStream_t st, const uint16_t* x, uint16_t* y, size_t n);
int rope_backward_launch(hipStream_t st, uint16_t* d, const float* table, int pos0, int n_tok, int seq_len, long long stride, int n_head, int hd);
int gelu_backward_launch(hipStream_t st, uint16_t* d_in_out, const uint16_t* x, size_t n);
int swiglu_backward_launch(hipStream_t st, uint16_t* delta_in_out, uint16_t* delta_gate, const uint16_t* gate, const uint16_t* up, size_t n);
// LayerNorm / RMSNorm backward (kf_norm_bwd.hip); scratch: norm_backward_groups(rows) * (mean ? 2 : 1) * C doubles
int norm_backward_groups(int rows);
int norm_backward_launch(hipStream_t st, uint16_t* dinp, uint16_t* dweight, uint16_t* dbias, const uint16_t* dout, const uint16_t* inp, const uint16_t* weight,
                         const float* mean, const float* rstd, int rows, int C, double* scratch);
int bias_residual_launch(hipStream_t st, uint16_t* y, const uint16_t* bias, const uint16_t* residual, size_t n, int M); /* kf_ops.hip */
// linear backward helpers (kf_linear_bwd.hip)
int transpose_bf16_launch(hipStream_t st, const uint16_t* in, uint16_t* out, int R, int C);
int colsum_add_launch(hipStream_t st, const uint16_t* x, uint16_t* dst, int n, int C, double* scratch); /* scratch: ceil(n / 256) * C doubles */
// causal MHA backward on MFMA tiles (kf_attn_bwd_mfma.hip); scratch: 2 * n_seq * n_head * T floats; 1 = shape not covered
int attn_backward_mfma_launch(hipStream_t st, const uint16_t* q, const uint16_t* k, const uint16_t* v, long long ld_qkv, const uint16_t* o, const uint16_t* dO, long long ld_o,
                              uint16_t* dq, uint16_t* dk, uint16_t* dv, long long ld_d, int T, int n_head, int hd, int n_seq, float* scratch, int n_kv, long long ld_kv,
                              long long ld_dkv); /* kf_attn_bwd_mfma.hip: 1 = not covered */
// embedding backward (kf_embed_bwd.hip)
int argmax_rows_state_launch(hipStream_t st, const uint16_t* logits, long long ld, int n, int n_rows, const int* d_seq, int32_t* states, int32_t* tokens_out, int tokens_stride);
int copy_blocks_launch(hipStream_t st, void* const* dst_table, size_t dst_offset, const void* src, size_t src_stride, size_t block_bytes, int n_blocks);
int embed_pos_launch(hipStream_t st, const uint16_t* wte, long long ldw, const uint16_t* wpe, const int* tokens, int B, int T, int C, int V, uint16_t* out);
int embed_backward_launch(hipStream_t st, uint16_t* dwte, long long ldw, uint16_t* dwpe, const uint16_t* dout, const int* tokens, int B, int T, int C, int V);
// fused classifier (kf_loss.hip): cross-entropy loss per row + logit gradient in place
int fused_classifier_launch(hipStream_t st, uint16_t* logits, float* losses, uint16_t* probs, float dloss, const int* targets, long rows, int V, int P,
                            const int* mask, int write_dlogits);
int swiglu_launch(hipStream_t st, const uint16_t* gate, const uint16_t* up, uint16_t* out, int n);
int add_launch(hipStream_t st, const uint16_t* a, const uint16_t* b, uint16_t* out, int n);
int embed_launch(hipStream_t st, const kf_weight* w, int token, const int32_t* d_token, const int32_t* d_state, const int32_t* d_forced,
                 uint16_t* out, int n_tok = 1);
int dequant_launch(hipStream_t st, const kf_weight* w, uint16_t* out, int ilv_n = 1, int ilv_i = 0); /* ilv_n > 1: rows interleaved with ilv_n - 1 other matrices in blocks of 16 (kf_ops.hip) */
int adamw_launch(hipStream_t st, uint16_t* params, uint16_t* grads, void* gm, void* gv, size_t n, int mv_bf16, float lr, float beta1, float beta2, float b1c,
                 float b2c, float eps, float wd, float grad_scale, unsigned int seed, int* status);
int sample_launch(hipStream_t st, const uint16_t* logits, int n, int top_k, float temperature, float top_p, unsigned long long* rng, int32_t* d_token,
                  int32_t* d_state, int32_t* d_tokens_out, const int32_t* d_forced, int n_forced, int true_topk = 0);
int quantize_launch(hipStream_t st, const kf_weight* w, const uint16_t* src, int symmetric);
// row-codebook 4-bit storage (kf_lut.hip): NF4 quantiser, dequant, embedding rows
int lut_quantize_launch(hipStream_t st, const kf_weight* w, const uint16_t* src);
int lut_dequant_launch(hipStream_t st, const kf_weight* w, uint16_t* out);
int lut_embed_launch(hipStream_t st, const kf_weight* w, int token, const int32_t* d_token, const int32_t* d_state, const int32_t* d_forced, uint16_t* out, int n_tok);
// ---- AutoAWQ layout (kf_awq.hip)
size_t awq_scratch_bytes(const kf_weight* w);
int awq_linear_launch(hipStream_t st, const kf_weight* w, const uint16_t* x, uint16_t* y, const uint16_t* bias, float alpha, float beta,
                      const uint16_t* residual, float* scratch);
int awq_dequant_launch(hipStream_t st, const kf_weight* w, uint16_t* out);

int set_state_launch(hipStream_t st, int32_t* d_state, int token, int pos);
int hot_rows_launch(hipStream_t st, const int32_t* hot, int n, int32_t* rows, int32_t* count);
int cold_fill_launch(hipStream_t st, uint16_t* y, const uint16_t* bias, int n);
int tp_reduce_launch(hipStream_t st, const float* partials, int R, int n, const uint16_t* residual, uint16_t* out);

}  // namespace kf
