/*
 * kf_abi.h -- C ABI of libkf_hip.so: the MI355X (gfx950) implementation of Koifish's quantized
 * transformer forward path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless
 * its name starts with h_.  All functions return 0 (KF_OK) or a negative KOIFISH_* code
 * (src/g_def_x.hpp:21-83) and never exit(); kf_last_error() returns the text.
 *
 * Each entry point replaces one seam of the reference; the seam is cited as
 * file:line into gruai/koifish.  A kf_ctx is single-threaded by contract (the reference drives one
 * stream from one host thread, NeuronFuse.cu:29); kernels never allocate (GTensor owns memory).
 * bf16 is carried as uint16_t bit patterns (floatX = floatGama = __nv_bfloat16, g_float.hpp:246-262).
 */
#ifndef KF_ABI_H
#define KF_ABI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t kf_bf16;
typedef struct kf_ctx kf_ctx;
typedef struct kf_graph kf_graph;

/* error codes: values of src/g_def_x.hpp:21-83 that this path can raise */
#define KF_OK 0
#define KF_INTERNAL_ERR (-11)          /* KOIFISH_INTERNAL_ERR */
#define KF_INVALID_ARGS (-20)          /* KOIFISH_INVALID_ARGS */
#define KF_OUTOF_GPUMEMORY (-100)      /* KOIFISH_OUTOF_GPUMEMORY */
#define KF_QUANT_ERR (-701)            /* KOIFISH_QUANT_ERR */
#define KF_UNSUPPORTED_DATATYPE (-1000) /* KOIFISH_UNSUPPORTED_DATATYPE */
#define KF_HIP_CHECK (-1400)           /* KOIFISH_CUDA_CHECK */
#define KF_ADAMW_MV (-5100)            /* KOIFISH_ADAMW_MV: a non-finite parameter or step met by kf_adamw (written to *d_status) */
#define KF_BLAS_UNALIGN (-2000)        /* KOIFISH_BLAS_UNALIGN: a pointer is not 16-byte aligned (gemm.cu:119-122) */
#define KF_RMS_PARAMS (-2100)          /* KOIFISH_RMS_PARAMS */

/* typNUMBER (src/g_float.hpp:84-117), same numeric values */
enum kf_dtype {
    KF_F32 = 0, KF_F64, KF_F16, KF_BF16, KF_F8E5M2, KF_F8E4M3, KF_U8, KF_I8, KF_U16, KF_I16, KF_U32, KF_I32, KF_U64, KF_I64,
    KF_Q4, KF_Q3, KF_Q2, KF_T_SIGN, KF_T_SEQ, KF_BOOL1, KF_T_BINARY, KF_T_BINARY_3, KF_T_BINARY_TILE
};

/*
 * A weight as GTensor + its quant card hand it to a kernel (GTensor.hpp:168-490, TASKA_quant
 * GeQuant.cpp:1297-1350).  `data` is the start of the `data||gama` allocation: packed Packed128 stream
 * (PackedQ.hpp:28-60; W[ne0,ne1] row-major flattened, groups of lGroup consecutive elements) or plain
 * bf16 / f8e5m2 elements.  `gama` = gama_T(GAMA) = data + szData: bf16 [R_SCALE ne0][C_SCALE ne1]
 * [ZERO nGroup][STEP nGroup] (GTensor.cpp:456-510); NULL for unquantised types.
 */
typedef struct kf_weight {
    const void* data;
    const kf_bf16* gama;
    int32_t type;   /* kf_dtype: KF_BF16, KF_F8E5M2, KF_Q4, KF_T_SIGN, KF_BOOL1, KF_T_BINARY */
    int32_t ne0;    /* rows  = out features (OC) */
    int32_t ne1;    /* cols  = in features  (IC) */
    int32_t nGroup; /* ne0*ne1/lGroup */
    int32_t lGroup; /* T_group, 128 */
    int32_t qMin, qMax, qBias;
    /* Vendor AutoAWQ tensors (GTensor::qZero / qScale, GeQuant.cpp:410; CU_Q42X_awq quantizer.cu:131-156).  Both non-NULL select the
     * AWQ GEMM layout: type = KF_Q4, data = qweight int32 [ne1, ne0/8] (TRANSPOSED: input rows), qzeros int32 [ne1/128, ne0/8],
     * qscales fp16 [ne1/128, ne0]; gama is unused.  Served by kf_linear and kf_dequant (which then writes [ne1, ne0], TransA = 0). */
    const void* qzeros;
    const void* qscales;
    /* Quant card mode (QUANT_MODE, GeQuant.hpp).  0 = KF_QUANT_GROUP: everything above.  KF_QUANT_ROW_LUT: the row-codebook 4-bit storage of
     * GeQuant::RT_NormalF / _row_lut (GeQuant.cpp:696-755; QUANT_MODE::RTNf "NF4" and the generic LUT mode share it): type = KF_Q4,
     * data = the nibble stream BIT_SET_k writes (CLI_params.cpp:2177-2190: element i of the row-major matrix in byte i/2, even i in the HIGH nibble),
     * gama = bf16 [R_SCALE ne0][C_SCALE ne1][LUT ne0 x 16] (GTensor.cpp:456-510 with the LUT at +ne0+ne1); a weight is lut[row][nibble]
     * (CU_Q42X_NF4 / CU_Q42X_lut, quantizer.cu:583-652, rc_normal = 0 as LowBit_worker's only sweep entry leaves it, GeQuant.cpp:844).
     * lGroup / nGroup / qMin / qMax / qBias are not used.  ne1 must be a multiple of 32.
     * The same mode with type = KF_Q3 / KF_Q2 is the 8- / 4-entry form (CU_Q32X_NF3 / CU_Q32X_ / CU_Q22X_, quantizer.cu:690-792): a 3- / 2-bit stream,
     * most significant bit first, 8 weights per 3 / 2 bytes, [LUT ne0 x 8] / [LUT ne0 x 4]; ne1 a multiple of 8.  KF_QUANT_ROW_RTN (type = KF_Q2) is
     * CU_Q22X_RTN (quantizer.cu:655-688): gama = [R][C][(zero, step) x ne0], weight = bf16(zero + bf16(step * id)).  These three are served the way the
     * reference serves them -- kf_dequant (GetDataX), and kf_linear as dequant + the bf16 product; kf_quantize builds the 3-bit normal-float form
     * (RT_NormalF asserts bits == 4 || 3).  CU_Q32X_RTN is not restated: it reads every row from row 0's stream (quantizer.cu:802, no row offset). */
    int32_t quant;
    int32_t reserved_;
} kf_weight;
#define KF_QUANT_GROUP 0
#define KF_QUANT_ROW_LUT 1
#define KF_QUANT_ROW_RTN 2

/* kf_linear epilogue flags */
#define KF_EPI_NONE 0u
#define KF_EPI_RESIDUAL 1u /* y = bf16(residual + bf16(W.x))   (CU_add3 after proj_cat/down: QKV.cu:687, NeuronFuse.cu:642) */

/* ---- lifetime: InitCUDA / CUDA_cleanup / SYNC_STREAM (QKV.cu:501-571,600-615; E_GPU.cpp:152-175) ---- */
int kf_init(int device, void* hip_stream_or_null, kf_ctx** out);
int kf_destroy(kf_ctx* ctx);
int kf_sync(kf_ctx* ctx);
const char* kf_last_error(void);
const char* kf_version(void);
/* device memory (huTensor::Alloc / SerialGamaData: huTensor.cu:385-458) -- thin wrappers so that host code
 * above this ABI needs no HIP headers */
int kf_malloc(kf_ctx* ctx, size_t bytes, void** out);
int kf_free(kf_ctx* ctx, void* p);
int kf_memset(kf_ctx* ctx, void* p, int value, size_t bytes);
/* `count` 32-bit words of `value`, on the context's stream, no host sync (kf_h2d synchronises: a flag or a seed set between launches goes through this) */
int kf_memset32(kf_ctx* ctx, void* p, int32_t value, size_t count);
int kf_h2d(kf_ctx* ctx, void* dst, const void* h_src, size_t bytes);
int kf_d2h(kf_ctx* ctx, void* h_dst, const void* src, size_t bytes);
int kf_d2d(kf_ctx* ctx, void* dst, const void* src, size_t bytes);
/* hipGraph capture of whatever is launched between begin/end on the ctx stream; replay with kf_graph_launch */
int kf_graph_begin(kf_ctx* ctx);
int kf_graph_end(kf_ctx* ctx, kf_graph** out);
int kf_graph_launch(kf_ctx* ctx, kf_graph* g);
int kf_graph_destroy(kf_graph* g);
/* HIP events on the ctx stream (bench.py times kernels with these) */
int kf_event_create(void** ev);
int kf_event_record(kf_ctx* ctx, void* ev);
int kf_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* synchronises on ev_stop */
int kf_event_destroy(void* ev);

/* ---- position / token source.  Every decode kernel takes `pos` by value AND an optional device
 * pointer d_pos: when d_pos != NULL the kernel reads *d_pos instead (so one captured graph serves many
 * positions) and `pos` only bounds the launch geometry (it must be >= *d_pos). ---- */

/* GTensor::GetDataX (GTensor.hpp:399, quantizer.cu:249-392): bf16 view of any weight into `out` [ne0*ne1] */
int kf_dequant(kf_ctx* ctx, const kf_weight* w, kf_bf16* out);

/* device quantiser, GeQuant::RTN_x / YinYang (GeQuant.cpp:428-628; device twin CU_XtoQ128_ T.cu:105-175):
 * src bf16 [ne0*ne1] -> w->data (packed) and w->gama zero/step.  `w` must be fully described. symmetric: RTN only.
 * KF_F8E5M2: the storage conversion of Float2T<f8e5> (g_float.hpp:433-443): float -> half (RNE) -> high byte. */
int kf_quantize(kf_ctx* ctx, const kf_weight* w, const kf_bf16* src, int symmetric);

/* SLP::Forw -> TASKA_AxB::blasLt -> CU_mm_blasLt (Neuron.hpp:418, NeuronFuse.cu:305-381, GTensor.hpp:703-741,
 * gemm.cu:93-214): y[nTok, ne0] = alpha * x[nTok, ne1] . W^T (+ beta*y) (+ bias), fp32 accumulate, bf16 out,
 * computed straight from the packed stream (no GetDataX round trip).  x [nTok, ne1], y [nTok, ne0] row-major; from 8 token rows up
 * the product runs on MFMA tiles fed by in-register unpack (kf_gemm.hip), below that (or for K not a multiple of 128) one mat-vec per row:
 * the same values up to fp32 summation order.
 * epilogue KF_EPI_RESIDUAL adds `residual` [nTok, ne0]; `residual` MAY alias `y` (the in-place form of SelfAttention / FFN::cuFlow): every path reads a
 * residual element before it stores that element.  x must not alias y.
 * Kernels never allocate: the three storages served by "dequantise, then multiply" (the reference's own order) -- AutoAWQ tensors, the 3- / 2-bit row
 * forms, and the 4-bit row codebook for token batches the tile kernels do not cover -- use a caller-owned workspace: kf_linear_scratch_bytes says how
 * much a weight needs (0 for every PackedQ / bf16 / f8 weight), kf_set_scratch hands the context a buffer (device memory, 16-byte aligned, the caller's
 * to free after the last call that used it; not while capturing, since captured launches hold the pointer).  Too small or missing: KF_INVALID_ARGS. */
/* Summation order of the decode kernels (mat-vecs, LM head, decode attention, the persistent engine's mat-vec phases).
 *   1 (the default since round 4): the CANONICAL order kernels and the CPU oracle share (oracle/kf_oracle.c sections 4c and 6 "CANON"): every lane keeps two fused
 *     multiply-add chains (even / odd elements: one v_pk_fma_f32 per weight pair) joined by a balanced tree, the softmax works on exact power-of-two scalings with
 *     fp64 sums -- logits, greedy ids and KV rows equal the oracle's BIT FOR BIT (tests/test_gpu_canonical.py, bench.py cpu_baseline parity pass).  It is the order
 *     bench.py times.  (The persistent engine follows the switch in every phase: mat-vecs, attention slice, head.)
 *   0: products by v_dot2c_f32_bf16, the softmax in fp32 -- <= 1 bf16 ulp per output from the oracle (its order is not reproducible on a host: the instruction's
 *     internal rounding has no bit-exact model); the fewest vector instructions per weight: a few per cent faster on the latency-bound 0.6B step, ~25 % on the
 *     VALU-bound mat-vecs of a 32B model.  Greedy ids may differ from the oracle's at near-ties of the two best logits.
 * The reference's own order is cuBLASLt's and unspecified (gemm.cu:126).  Not while capturing. */
int kf_set_canonical(kf_ctx* ctx, int on);
int kf_get_canonical(kf_ctx* ctx);
size_t kf_linear_scratch_bytes(const kf_weight* w, int nTok);
int kf_set_scratch(kf_ctx* ctx, void* scratch, size_t bytes);
/* RESIDENT dequantised copies for token batches (prompt prefill).  The reference dequantises a quantised weight in front of EVERY token-batch product
 * (GTensor::GetDataX -> cuBLASLt, SLP::Forw); with 288 GB of HBM the bf16 form of an inference model's matrices can simply stay: the routes of kf_linear,
 * kf_linear_multi, kf_qkv_rope_batch and kf_gateup_swiglu_batch that dequantise into the scratch (>= 1024 token rows; with an arena from 320) put the copy into `arena` instead the first
 * time they meet a weight (or a stacked Q | K | V / interleaved gate | up set) and find it there afterwards -- same values, same tile kernels, no dequantise
 * launches, and kf_linear takes the bf16 tile kernels from 320 rows (o_proj / down_proj of a prompt: 64 x 128 / 64 x 64 tiles of kf_gemm3.hip, in k-pieces when a
 * kf_set_scratch workspace of kf_resident_scratch_bytes() is lent).
 * Copies are keyed by the weights' data pointers: the caller promises the weights do not change while the arena holds them (inference; after a weight update
 * call kf_set_dequant_arena again -- it forgets every copy).  arena: device memory, 16-byte aligned, the caller's; NULL switches the feature off; when it is
 * full, further weights go through the scratch as before.  kf_dequant_arena_used: bytes filled so far.  Not while capturing. */
int kf_set_dequant_arena(kf_ctx* ctx, void* arena, size_t bytes);
size_t kf_dequant_arena_used(kf_ctx* ctx);
size_t kf_resident_scratch_bytes(void); /* a kf_set_scratch workspace of at least this size lets kf_linear's resident-copy route cut short token batches into k-pieces */
int kf_linear(kf_ctx* ctx, const kf_weight* w, const kf_bf16* x, kf_bf16* y, const kf_bf16* bias, int nTok, float alpha, float beta,
              uint32_t epilogue, const kf_bf16* residual);

/* ---- tensor parallel (no counterpart in the reference, which is single-GPU: QKV.cu:503) ----
 * column-split projections (o_proj, down_proj) give un-rounded fp32 partial row dots per rank ... */
int kf_linear_f32(kf_ctx* ctx, const kf_weight* w_shard, const kf_bf16* x_shard, float* y_partial);
/* ... that are combined in rank order after an all-gather: out = bf16(residual + bf16(sum_r partials[r][:])) */
int kf_tp_reduce(kf_ctx* ctx, const float* partials, int n_ranks, int n, const kf_bf16* residual_or_null, kf_bf16* out);

/* The exchange itself without a collective library: every rank owns a receive area that every other rank can address (xGMI peer-to-peer mapping
 * of kf_tp_alloc memory, exported / opened with the kf_tp_ipc_* calls; ranks of one process just pass pointers).  kf_linear_f32_push is kf_linear_f32
 * whose epilogue stores each fp32 row dot, tagged, into this rank's slot of EVERY rank's area (one 8-byte system-scope store per rank and row);
 * kf_tp_reduce_recv waits for the tags of all ranks' slots and forms out = bf16(residual + bf16(sum_r partial_r)) in rank order -- the bits of
 * kf_tp_reduce.  kf_tp_lm_head runs the LM head on this rank's vocabulary rows and pushes its first maximum (value, GLOBAL row) to every rank;
 * kf_tp_pick takes the first maximum over the ranks, updates the decode state {token, pos + 1} like kf_norm_lm_head does, and advances the
 * generation word.  All of it is plain kernel launches on the context's stream: the TP step captures into a hipGraph.
 * Area layout (8-byte granules): [2 exchange buffers][world][n_max] vector slots, then [world][2] pick slots; kf_tp_recv_bytes gives the size;
 * the area must be zero before the first step (kf_tp_alloc zeroes it).  `index` numbers the exchanges of one step (0 .. per_step - 2; the
 * pick uses per_step - 1); exchanges with even index use buffer 0, odd ones buffer 1.  A poll that outlasts ~2^24 re-reads sets *d_err
 * (1 + rank waited for) instead of hanging. */
typedef struct kf_tp_comm {
    int32_t rank, world; /* world <= 8 */
    int32_t n_max;       /* rows of a vector slot */
    uint32_t per_step;   /* exchanges per token including the pick */
    void* recv;          /* this rank's receive area */
    void* peer[8];       /* peer[r]: rank r's receive area as addressable from this process (peer[rank] == recv) */
    uint32_t* d_step;    /* device word, zero at start: generation */
    int32_t* d_err;      /* device word, zero at start */
    void* d_push;        /* kf_tp_push_bytes(per_step) bytes of device memory: the push descriptors, written by kf_tp_commit */
} kf_tp_comm;
size_t kf_tp_recv_bytes(int world, int n_max);
size_t kf_tp_push_bytes(uint32_t per_step);
int kf_tp_commit(kf_ctx* ctx, const kf_tp_comm* comm); /* once every peer[] is set, before the first kf_linear_f32_push (not while capturing) */
int kf_tp_alloc(kf_ctx* ctx, size_t bytes, void** out);              /* device memory other processes / devices may map; zeroed */
int kf_tp_ipc_export(kf_ctx* ctx, void* p, unsigned char handle[64]); /* hipIpcGetMemHandle */
int kf_tp_ipc_open(kf_ctx* ctx, const unsigned char handle[64], void** out);
int kf_tp_ipc_close(kf_ctx* ctx, void* p);
int kf_linear_f32_push(kf_ctx* ctx, const kf_weight* w_shard, const kf_bf16* x_shard, const kf_tp_comm* comm, uint32_t index);
int kf_tp_reduce_recv(kf_ctx* ctx, const kf_tp_comm* comm, uint32_t index, int n, const kf_bf16* residual_or_null, kf_bf16* out);
int kf_tp_lm_head(kf_ctx* ctx, const kf_bf16* x, const kf_bf16* norm_w, float eps, const kf_weight* w_shard, kf_bf16* logits_shard, int row0,
                  const kf_tp_comm* comm, void* head_scratch);
int kf_tp_pick(kf_ctx* ctx, const kf_tp_comm* comm, int32_t* d_state, int32_t* d_tokens_out);

/* ---- sparse ("EOE" / hot-neuron) forward: D_matmul_sparse (src/Utils/GST_float.cpp:306-318) with the hot[] array of CS_Picker
 * (src/Manifold/SparseNeuron.cpp:20-29; Neuron.hpp:265-285: one int per FFN row, 1 = hot):  y[i] = (hot[i] == 1 ? W[i,:].x : 0) (+ bias[i]).
 * The mask is turned into a list of hot rows once (kf_hot_rows: ascending indices, their number to *d_count) and the products walk that list, so
 * cold rows cost no HBM traffic: algorithmic bytes = n_hot / ne0 of the dense product's.  Hot rows carry every bit of kf_linear's rows.
 * kf_norm_gateup_swiglu_masked is the FFN form (FFN::cuInfer with the mask on the gate / up rows): cold rows of `act` are SwiGLU(0, 0) = 0. */
int kf_hot_rows(kf_ctx* ctx, const int32_t* d_hot, int n, int32_t* d_rows, int32_t* d_count);
int kf_linear_masked(kf_ctx* ctx, const kf_weight* w, const kf_bf16* x, kf_bf16* y, const kf_bf16* bias_or_null, const int32_t* d_rows, int n_hot);
int kf_norm_gateup_swiglu_masked(kf_ctx* ctx, const kf_bf16* x, const kf_bf16* norm_w_or_null, float eps, const kf_weight* gate, const kf_weight* up,
                                 kf_bf16* act, const int32_t* d_rows, int n_hot);

/* LayerNormal::cuFlow -> CU_rms_infer (Neuron.hpp:453, T.cu:561-573, layernorm.cuh:800-859) */
int kf_rmsnorm(kf_ctx* ctx, const kf_bf16* x, const kf_bf16* w, kf_bf16* y, int rows, int dim, float eps, float* rstd_or_null);

/* ROPE::cuInfer (Neuron.hpp:364, rope.cu:645-672): per-head RMSNorm (CU_rms_forward_v2) then rotate-half RoPE
 * (CU_rope2_v0) on q [n_head*hd] and on the new key row k [n_kv*hd], in place.  wq_norm/wk_norm may be NULL.
 * rope_table: fp32 [max_pos][hd/2][2] = (cos, sin) built by kf_rope_table_host. */
int kf_qknorm_rope(kf_ctx* ctx, kf_bf16* q, kf_bf16* k, const kf_bf16* wq_norm, const kf_bf16* wk_norm, const float* rope_table, int pos,
                   const int32_t* d_pos, int n_head, int n_kv, int hd, float eps);
/* host: fills h_table [n_pos][hd/2][2] with cosf/sinf(pos / powf(theta, 2j/hd)) (operator.cuh:734-772) */
int kf_rope_table_host(float* h_table, int n_pos, int hd, float theta);

/* attention triple of SelfAttention::cuInfer (QKV.cu:669-673; operator.cuh:572-632,251-277,649-668), fused,
 * full causal length.  q bf16 [n_head*hd] (already normed+roped), kcache/vcache: layer base, row t at
 * t*kv_stride elements; out bf16 [n_head*hd]; scratch: kf_attn_scratch_bytes(), zero-filled ONCE by the caller (the
 * kernel keeps its arrival counters at zero between launches). */
int kf_attn_decode(kf_ctx* ctx, const kf_bf16* q, const kf_bf16* kcache, const kf_bf16* vcache, kf_bf16* out, int pos, const int32_t* d_pos,
                   int n_head, int n_kv, int hd, int kv_stride, void* scratch);
size_t kf_attn_scratch_bytes(int n_head, int hd);

/* Relu::Forw -> CU_swiglu_v0 (Neuron.hpp:384, Activation.cu:85-93) */
int kf_swiglu(kf_ctx* ctx, const kf_bf16* gate, const kf_bf16* up, kf_bf16* out, int n);
/* T1p1(CU_add3) (QKV.cu:687, packedN.cuh:866-875) */
int kf_add(kf_ctx* ctx, const kf_bf16* a, const kf_bf16* b, kf_bf16* out, int n);
/* TokenEmbed::cuInfer/OnEmbed (NeuronFuse.cu:176-218, embed.cuh:54-132).  token by value, or *d_token */
int kf_embed(kf_ctx* ctx, const kf_weight* w, int token, const int32_t* d_token, kf_bf16* out);
/* Head4Token::cuInfer_1 + sample_argmax (NeuronFuse.cu:842-862, GoPT.cpp:602-612): logits bf16 [ne0] (required) and
 * the greedy id (first maximum) to d_argmax_out (device int32; NULL: logits only, no pick).  scratch: kf_head_scratch_bytes(). */
int kf_lm_head(kf_ctx* ctx, const kf_weight* w, const kf_bf16* x, kf_bf16* logits, int32_t* d_argmax_out, void* scratch_or_null);
size_t kf_head_scratch_bytes(void);

/* ---- fused forms used by the decode step (same arithmetic as the calls above, fewer launches) ---- */
/* [RMSNorm(x, norm_w)] -> up to 3 projections sharing the normed input.  y[i] rows go to y[i], or, when
 * y_pos_stride[i] != 0, to y[i] + pos*y_pos_stride[i] (K.out/V.out alias the KV-cache row: TGraph.cpp:198-207) */
int kf_norm_linear(kf_ctx* ctx, const kf_bf16* x, const kf_bf16* norm_w_or_null, float eps, int n_w, const kf_weight* const* w,
                   kf_bf16* const* y, const int64_t* y_pos_stride, int pos, const int32_t* d_pos);
/* [RMSNorm] -> gate & up -> SwiGLU -> act [ffn]   (FFN::cuInfer NeuronFuse.cu:615-637) */
int kf_norm_gateup_swiglu(kf_ctx* ctx, const kf_bf16* x, const kf_bf16* norm_w_or_null, float eps, const kf_weight* gate, const kf_weight* up,
                          kf_bf16* act);
/* q/k-norm + RoPE + attention in one pass: q raw [n_head*hd], k_raw [n_kv*hd] (the new key before norm/rope,
 * written normed+roped into kcache row pos), v row must already sit in vcache row pos.
 * Requirement on the caches (also kf_attn_decode, kf_engine_step): with d_pos the launch is laid out for positions up to `pos` (the bound) and loads K / V rows
 * up to that bound before the device position is known; rows past the real position are masked by a zero probability, not by a select, so they must hold
 * FINITE values (allocate the caches zero-filled, as KVCache::Init does: 0 * NaN would poison the output). */
int kf_attn_block(kf_ctx* ctx, const kf_bf16* q_raw, const kf_bf16* k_raw, kf_bf16* kcache, const kf_bf16* vcache, kf_bf16* out,
                  const kf_bf16* wq_norm, const kf_bf16* wk_norm, const float* rope_table, int pos, const int32_t* d_pos, int n_head, int n_kv,
                  int hd, int kv_stride, float eps, void* scratch);
/* [final RMSNorm] + LM head + greedy pick; then state update for graph replay: d_state[0] = next token,
 * d_state[1] += 1 (position), d_tokens_out[old pos] = next token (when non-NULL).  d_state == NULL: logits only (kf_sample follows). */
int kf_norm_lm_head(kf_ctx* ctx, const kf_bf16* x, const kf_bf16* norm_w_or_null, float eps, const kf_weight* w, kf_bf16* logits,
                    int32_t* d_state, int32_t* d_tokens_out, void* scratch);
/* writes the decode state {token, pos} (a one-thread kernel: capturable, no host staging buffer) */
int kf_set_state(kf_ctx* ctx, int32_t* d_state, int token, int pos);
/* embed lookup driven by device state: token = (d_forced && d_forced[pos] >= 0) ? d_forced[pos] : d_state[0] */
int kf_embed_state(kf_ctx* ctx, const kf_weight* w, const int32_t* d_state, const int32_t* d_forced, kf_bf16* out);

/* GeneratOnPrompt::Sample, non-greedy branch (GoPT.cpp:614-630; LogitsInfo::TopK / UpdateLogits / TopP / Qu_FlipCoin, GoPT.cpp:632-790):
 * picks the next token from bf16 logits[n] on the device -- candidate set exactly as TOPK_heap::Select builds it (indices
 * 0..k-2 plus the first maximum over i >= k-1), softmax with temperature, top-p cut, xorshift64* coin.  d_rng_state: one uint64 in device
 * memory (LogitsInfo::rng_state, seeded with config.common.seed), advanced once per call.  Writes *d_token (when non-NULL) and, when
 * d_state is non-NULL, the decode-state update of kf_norm_lm_head.  temperature == 0 or top_k == 1 is the greedy path: use kf_lm_head.
 * d_forced (optional, n_forced entries, -1 = free): when the token of the NEXT position is teacher-forced (prompt prefill through the
 * decode path) nothing is drawn and the rng state is left alone, as the reference's prefill loop never samples (GoPT.cpp:1139-1146).
 * KF_INVALID_ARGS unless 2 <= top_k < n/2 (the reference asserts it), top_k <= 1024, temperature > 0, top_p > 0. */
int kf_sample(kf_ctx* ctx, const kf_bf16* logits, int n, int top_k, float temperature, float top_p, uint64_t* d_rng_state, int32_t* d_token,
              int32_t* d_state, int32_t* d_tokens_out, const int32_t* d_forced, int n_forced);

/* kf_sample with the candidate set the reference's TopK is evidently meant to keep: the top_k LARGEST logits (ties at the k-th value towards the
 * lower token index; candidates ordered by logit, equal logits by index).  Not what the reference computes (its heap keeps indices 0..k-2, see
 * kf_sample); offered beside it.  Everything else -- softmax with temperature, top-p cut, xorshift coin, state update -- is identical. */
int kf_sample_topk(kf_ctx* ctx, const kf_bf16* logits, int n, int top_k, float temperature, float top_p, uint64_t* d_rng_state, int32_t* d_token,
                   int32_t* d_state, int32_t* d_tokens_out, const int32_t* d_forced, int n_forced);

/* ---- GPT-2 family forward pieces (BASELINE config 3) */
/* LayerNorm forward with affine weight and bias (LayerNormal::cuFlow for the GPT-2 family -> CU_lm_forward, layernorm.cuh:226-300):
 * y = bf16((x - mean) * rstd * w + b), rstd = 1/sqrtf(var + eps); bias may be NULL; mean / rstd [rows] fp32 are optional outputs. */
int kf_layernorm(kf_ctx* ctx, const kf_bf16* x, const kf_bf16* w, const kf_bf16* bias_or_null, kf_bf16* y, int rows, int dim, float eps, float* mean_or_null,
                 float* rstd_or_null);
/* GELU, tanh form (Relu::Forw GELU -> gelu_forward_kernel2, Activation.cu:23-40) */
int kf_gelu(kf_ctx* ctx, const kf_bf16* x, kf_bf16* y, size_t n);

/* kf_qknorm_rope_batch for a training batch: n_tok rows = sequences of seq_len tokens back to back, positions 0 .. seq_len - 1 in each; the per-head
 * 1/rms of the q / k norms (fp32, before the bf16 rounding the forward applies to it) go to rstd_q [n_tok * n_head] / rstd_k [n_tok * n_kv] when given:
 * what kf_norm_backward needs for the q/k-norm backward. */
int kf_qknorm_rope_train(kf_ctx* ctx, kf_bf16* q, kf_bf16* k, const kf_bf16* wq_norm, const kf_bf16* wk_norm, const float* rope_table, int n_tok, int seq_len, int64_t q_stride,
                         int64_t k_stride, int n_head, int n_kv, int head_dim, float eps, float* rstd_q_or_null, float* rstd_k_or_null);

/* kf_attn_prefill for n_seq independent sequences of n_tok tokens each (a training batch), positions 0 .. n_tok - 1, in ONE launch: sequence s owns rows
 * s * n_tok .. (s + 1) * n_tok - 1 of q / out (row stride q_stride) and of k / v (row stride kv_stride; e.g. the column blocks of a fused [B*T, 3C] buffer).
 * Same arithmetic as kf_attn_prefill (the MFMA tile kernel).  head_dim 64 or 128. */
int kf_attn_prefill_batch(kf_ctx* ctx, const kf_bf16* q, const kf_bf16* k, const kf_bf16* v, kf_bf16* out, int n_tok, int64_t q_stride, int n_head, int n_kv, int head_dim,
                          int kv_stride, int n_seq);

/* the same with a row stride of its own for out: a training step reads q out of the fused [rows, 3C] buffer (q_stride 3C) and writes a dense [rows, C] out (out_stride C),
 * which saves the copy of the q columns (GPT-2's c_attn output: QKV.cu's fused qkv, /root/reference/src/Device/CUDA/QKV.cu) */
int kf_attn_prefill_batch_strided(kf_ctx* ctx, const kf_bf16* q, const kf_bf16* k, const kf_bf16* v, kf_bf16* out, int n_tok, int64_t q_stride, int64_t out_stride, int n_head,
                                  int n_kv, int head_dim, int kv_stride, int n_seq);

/* Causal attention backward for n_seq sequences of T tokens each, stored back to back in every tensor (the training path's SDPA backward:
 * cudnn-frontend in the reference, QKV.cu:130-315, 427-447).  q: rows of n_head * head_dim, k / v: rows of n_kv * head_dim (GQA: n_head a multiple of
 * n_kv), all with row stride ld_qkv (e.g. the column blocks of a fused q|k|v buffer); o (the forward output) and dO: n_head * head_dim with stride ld_o;
 * dq (n_head heads) and dk, dv (n_kv heads, summed over the query heads of a group) with stride ld_d.  scale 1/sqrt(head_dim), fp32 softmax recomputed
 * from q and k, bf16 stores.  head_dim 64 or 128 (KF_UNSUPPORTED_DATATYPE otherwise).
 * scratch: kf_attn_backward_scratch_bytes(T, n_head, n_seq) bytes (the rows' log-sum-exp and dO.O). */
size_t kf_attn_backward_scratch_bytes(int T, int n_head, int n_seq);
int kf_attn_backward(kf_ctx* ctx, const kf_bf16* q, const kf_bf16* k, const kf_bf16* v, long long ld_qkv, const kf_bf16* o, const kf_bf16* dO, long long ld_o, kf_bf16* dq,
                     kf_bf16* dk, kf_bf16* dv, long long ld_d, int T, int n_head, int n_kv, int head_dim, int n_seq, void* scratch);

/* Embedding forward of a training batch (encoder_forward -> encoder_forward_kernel3, kernel/embed.cuh:20-45,372-376): out[bt][c] = bf16(wte[tokens[bt]][c] + wpe[bt % T][c]), fp32 sum, round to nearest.
 * wte rows ldw apart; an id outside [0, V) reads row 0. */
int kf_embed_pos(kf_ctx* ctx, const kf_bf16* wte, long long ldw, const kf_bf16* wpe, const int32_t* tokens, int B, int T, int C, int V, kf_bf16* out);
/* The greedy pick (sample_argmax, GoPT.cpp:602-612: first maximum over float(logits)) of n_rows rows of bf16 logits [n] (rows ld apart) at once, each followed by the
 * decode-state update of kf_norm_lm_head for sequence d_seq[row] of a [n_seq][4] state array {token, pos, ..}: d_tokens_out[seq * tokens_stride + pos] = id (when non-NULL),
 * state = {id, pos + 1}.  The head of a batch of prompts: ONE product over the rows' hidden states (kf_linear), one pick launch. */
int kf_argmax_rows_state(kf_ctx* ctx, const kf_bf16* logits, long long ld, int n, int n_rows, const int32_t* d_seq, int32_t* d_states, int32_t* d_tokens_out, int tokens_stride);
/* block b (b < n_blocks) of src, blocks src_stride bytes apart, copied to d_dst_table[b] + dst_offset -- d_dst_table is a DEVICE array of device pointers (16-byte aligned
 * destinations); sizes, strides and the offset multiples of 16 bytes.  One launch scatters the K / V rows of a batch of prompts into the prompts' own caches
 * (the reference re-aims K.out / V.out at the cache rows of ONE sequence, TGraph.cpp:198-207). */
int kf_copy_blocks(kf_ctx* ctx, void* const* d_dst_table, size_t dst_offset, const void* src, size_t src_stride, size_t block_bytes, int n_blocks);
/* value into `width` bytes of each of `rows` rows `pitch` bytes apart, on the context's stream (the padded logit columns of a step) */
int kf_memset2d(kf_ctx* ctx, void* p, size_t pitch, int value, size_t width, size_t rows);

/* Embedding backward (encoder_backward, kernel/embed.cuh:380-470): dout [B*T, C] is the gradient of  wte[tokens[bt]] + wpe[t].
 *   dwpe [T, C]        += sum over the batch                         (NULL: no position table, e.g. a RoPE model)
 *   dwte [V, ldw >= C] += the rows of every position holding that token, in ascending position order   (NULL: skipped)
 * fp32 sums, bf16(sum + old) stores; tokens outside [0, V) are skipped.  Deterministic (no atomics), all on the device. */
int kf_embed_backward(kf_ctx* ctx, kf_bf16* dwte_or_null, long long ldw, kf_bf16* dwpe_or_null, const kf_bf16* dout, const int32_t* tokens, int B, int T, int C, int V);

/* Linear layer backward (SLP::Back, NeuronFuse.cu:495-547; matmul_backward, kernel/gemm.cu:326-370).  w [OC, IC] in any PackedQ / f8 / bf16 storage,
 * dequantised to bf16 first exactly as the reference's GetDataX does; deltaIn [n, OC] is the gradient of the layer's output, inp [n, IC] its input:
 *   delta [n, IC]  = (accumulate_delta ? delta : 0) + deltaIn . W        (NULL: skipped)
 *   gW    [OC, IC] += deltaIn^T . inp                                     (NULL: skipped -- isFixWeight; the bf16 gradient of a quantised weight's master copy)
 *   gBias [OC]     += column sums of deltaIn                              (NULL: no bias)
 * fp32 accumulation, bf16 stores (beta = 1 adds the stored bf16 value).  OC a multiple of 64 and >= 128 (so is n when gW is wanted), IC a multiple of 8.
 * scratch: kf_linear_backward_scratch_bytes(OC, IC, n) bytes of device memory, 256-byte aligned (the dequantised weight; the fp32 partial tiles of the split-K weight-gradient
 * launch, >= 64 MiB, or the transposed copies of shapes too small for the 256 x 256 tile kernel; the bias column-sum slabs).  Contents need not be initialised and
 * are not kept between calls; results do not depend on them (the split-K partial sums are added in a fixed order). */
size_t kf_linear_backward_scratch_bytes(int OC, int IC, int n);
int kf_linear_backward(kf_ctx* ctx, const kf_weight* w, const kf_bf16* deltaIn, const kf_bf16* inp_or_null, kf_bf16* delta_or_null, kf_bf16* gW_or_null,
                       kf_bf16* gBias_or_null, int n, int accumulate_delta, void* scratch);

/* LayerNorm / RMSNorm backward (LayerNormal::cuFlow, backward branch, T.cu:605-646: layernorm_backward -> layernorm_backward_kernel10, layernorm.cuh:311-503;
 * RMS: CU_rms_back_llmc, layernorm.cuh:863-1051).  mean == NULL selects RMSNorm.  dinp (the residual-path gradient on entry) becomes
 * bf16(dinp + dL/dinp); dweight (and dbias, LayerNorm with a bias) accumulate the sums over the rows: bf16(sum + old).  rstd (and mean) are the forward's
 * per-row outputs.  scratch: kf_norm_backward_scratch_bytes(rows, dim, mean != NULL) bytes of device memory, 8-byte aligned. */
size_t kf_norm_backward_scratch_bytes(int rows, int dim, int is_layernorm);
int kf_norm_backward(kf_ctx* ctx, kf_bf16* dinp, kf_bf16* dweight, kf_bf16* dbias_or_null, const kf_bf16* dout, const kf_bf16* inp, const kf_bf16* weight,
                     const float* mean_or_null, const float* rstd, int rows, int dim, void* scratch);

/* RoPE backward (the transpose of the rotate-half rotation of kf_qknorm_rope), in place on a gradient d [n_tok rows of n_head * head_dim, row stride
 * `stride`]: row t belongs to position pos0 + t % seq_len (seq_len = n_tok for one sequence; a batch of equal-length sequences stored back to back
 * passes its sequence length).  The q/k-norm that precedes RoPE in the forward is an RMSNorm over head_dim: its backward is kf_norm_backward with
 * rows = n_tok * n_head and dim = head_dim. */
int kf_rope_backward(kf_ctx* ctx, kf_bf16* d, const float* rope_table, int pos0, int n_tok, int seq_len, long long stride, int n_head, int head_dim);

/* Activation backward (Relu::Back, Activation.cu:283-320).  GELU, in place on the incoming gradient (Activation_backward_inplace ->
 * gelu_backward_inplace_kernel, Activation.cu:42-78): d = bf16(gelu'(x) * d) with x the pre-activation.  SwiGLU (CU_swiglu_back_v0,
 * Activation.cu:245-260): delta_gate = bf16(delta * up * sig * (1 + gate * (1 - sig))), delta_in_out = bf16(delta * gate * sig), sig = sigmoid(gate). */
int kf_gelu_backward(kf_ctx* ctx, kf_bf16* d_in_out, const kf_bf16* x, size_t n);
int kf_swiglu_backward(kf_ctx* ctx, kf_bf16* delta_in_out, kf_bf16* delta_gate, const kf_bf16* gate, const kf_bf16* up, size_t n);

/* Fused classifier of the GPT-2 / training forward (fused_classifier, src/Device/CUDA/kernel/fused_classifier.cuh:68-140, launched by Head4Token at
 * NeuronFuse.cu:923 as <<<dB*T, 1024>>>(logits, losses, nullptr, rLoss, targets, dB, T, V, Vp, devMask, write_dlogits)): for every row of
 * logits [B*T, P] (V valid entries, P the padded row length, a multiple of 8):  losses[row] -= log(softmax(row)[target])  (accumulates, as the
 * reference does), then -- write_dlogits != 0 -- the row is overwritten by bf16((prob - onehot(target)) * dloss), and probs (may be NULL) gets
 * bf16(prob).  Rows whose mask word (may be NULL) has bit 0x10000 (MASK_FLAG::F_IGNORE_LOSS, DataLoader.hpp:78) are skipped entirely.
 * Same thread decomposition and reduction order as the reference; exp / log are this library's fixed fp32 recipes. */
int kf_fused_classifier(kf_ctx* ctx, kf_bf16* logits, float* losses, kf_bf16* probs_or_null, float dloss, const int32_t* targets, int B, int T, int V, int P,
                        const int32_t* mask_or_null, int write_dlogits);

/* ---- training kernel path (BASELINE config 3), first piece: the AdamW parameter update CU_adamw_p (src/Device/CUDA/Optimizer.cu:393-442)
 * as PIPE_Adamw::Update launches it (Optimizer.cu:630-646; TASKA_1p1, packedN.cuh:612-643: 512 threads x 8 bf16 per thread).
 * params / grads bf16 [n] (grads are zeroed), gm / gv: first and second moments, mv_type KF_BF16 (floatMV = bf16) or KF_F32.
 * g = grad_scale*grad; m = sAtB(g, m, beta1); v = sAtB(g*g, v, beta2); step = (m/beta1_correction) / (sqrtf(v/beta2_correction) + eps);
 * p -= lr*weight_decay*p + lr*step.  bf16 stores use the reference's seeded stochastic rounding (CU_Float2T<bf16>, packedN.cuh:62-72;
 * SquirrelNoise5 keyed on the launch geometry, which this entry reproduces), so results are bit-identical to the restatement in oracle/.
 * n must be a multiple of 8.  A thread that meets a non-finite parameter or step stores nothing and writes KF_ADAMW_MV to *d_status
 * (device int32, optional). */
int kf_adamw(kf_ctx* ctx, kf_bf16* params, kf_bf16* grads, void* gm, void* gv, size_t n, int mv_type, float learning_rate, float beta1, float beta2,
             float beta1_correction, float beta2_correction, float eps, float weight_decay, float grad_scale, uint32_t seed, int32_t* d_status);

/* ---- token batch (prompt prefill).  The reference feeds the prompt one token at a time through the decode path (Fish::Chat,
 * GoPT.cpp:1139-1146); its batched forward exists only on the training side (SelfAttention::cuFlow / ROPE::cuFlow,
 * NeuronFuse.cu:692-731, rope.cu).  These entries run the same per-token arithmetic for n_tok consecutive positions at once;
 * kf_linear / kf_rmsnorm / kf_swiglu already take row batches. */
/* n_w <= 3 matrices that share the input rows x [nTok, ne1] (Q, K, V): y[i] [nTok, w[i]->ne0].  One launch when a tile kernel
 * covers the shape, otherwise kf_linear per matrix.  Batches of >= 1024 rows (>= 320 with resident copies, kf_set_dequant_arena) of group-quantised matrices whose row counts are multiples of
 * 256 are dequantised back to back into the kf_set_scratch workspace (kf_linear_multi_scratch_bytes says how large; 0 = route not taken)
 * and multiplied by ONE launch of the 256 x 256 bf16 tile kernel -- the reference's own order (GetDataX, then the GEMM); its fp32 summation
 * order differs from the in-register-unpack kernels of smaller batches, as kf_linear's own large-batch path does. */
size_t kf_linear_multi_scratch_bytes(int n_w, const kf_weight* const* w, int nTok);
int kf_linear_multi(kf_ctx* ctx, int n_w, const kf_weight* const* w, const kf_bf16* x, kf_bf16* const* y, int nTok);
/* act [nTok, ne0] = silu(x . gate^T) * (x . up^T) for nTok rows (FFN::cuFlow: gate.Forw, up.Forw, Relu::Forw SWIG,
 * NeuronFuse.cu:615-656 with a batch): one launch, bit-identical to kf_linear x 2 + kf_swiglu; up_scratch [nTok, ne0] is used by
 * the unfused fallback and by the large-batch route (>= 1024 rows, >= 320 with resident copies: gate | up interleaved, SwiGLU in the tile GEMM's epilogue; or stacked as in kf_linear_multi, then kf_swiglu). */
int kf_gateup_swiglu_batch(kf_ctx* ctx, const kf_weight* gate, const kf_weight* up, const kf_bf16* x, kf_bf16* act, kf_bf16* up_scratch, int nTok);
/* out[t] = row d_tokens[t] of the table, t < n_tok (TokenEmbed::OnEmbed for a batch, NeuronFuse.cu:176-207) */
int kf_embed_batch(kf_ctx* ctx, const kf_weight* w, const int32_t* d_tokens, int n_tok, kf_bf16* out);
/* kf_qknorm_rope for tokens t < n_tok at positions pos0 + t: q + t*q_stride, k + t*k_stride (k may be cache rows: k_stride = kv_stride) */
int kf_qknorm_rope_batch(kf_ctx* ctx, kf_bf16* q, kf_bf16* k, const kf_bf16* wq_norm, const kf_bf16* wk_norm, const float* rope_table, int pos0, int n_tok,
                         int64_t q_stride, int64_t k_stride, int n_head, int n_kv, int hd, float eps);
/* SelfAttention::cuFlow's projection step for a token batch in one call: Q | K | V of the same x (K / V rows may be the cache rows themselves, TGraph.cpp:198-207) followed by
 * ROPE::cuInfer on q and k (kf_qknorm_rope_batch).  For >= 1024 tokens (>= 320 with resident copies) and head_dim 128 both happen in ONE launch -- the stacked tile GEMM with the q/k-norm and the rotation in
 * its epilogue (a 128-row tile is one head) -- otherwise kf_linear_multi + kf_qknorm_rope_batch.  Same arithmetic either way (token batches: MFMA summation order, tolerances). */
int kf_qkv_rope_batch(kf_ctx* ctx, const kf_weight* wq, const kf_weight* wk, const kf_weight* wv, const kf_bf16* x, kf_bf16* q, kf_bf16* k, kf_bf16* v, int nTok,
                      const kf_bf16* wq_norm_or_null, const kf_bf16* wk_norm_or_null, const float* rope_table_or_null, int pos0, int n_head, int n_kv, int head_dim, float eps);
/* the same for nTok rows that are SEQUENCES of seq_len tokens back to back (a batch of prompts; a training batch): positions 0 .. seq_len - 1 in each (pos0 must be 0, nTok a
 * multiple of seq_len; seq_len 0 = kf_qkv_rope_batch).  Unfused route: kf_linear_multi + kf_qknorm_rope_train. */
int kf_qkv_rope_seqs(kf_ctx* ctx, const kf_weight* wq, const kf_weight* wk, const kf_weight* wv, const kf_bf16* x, kf_bf16* q, kf_bf16* k, kf_bf16* v, int nTok, int seq_len,
                     const kf_bf16* wq_norm_or_null, const kf_bf16* wk_norm_or_null, const float* rope_table_or_null, int pos0, int n_head, int n_kv, int head_dim, float eps);
/* causal attention of tokens t < n_tok (position pos0 + t attends to cache rows 0 .. pos0 + t, which must already hold the prepared
 * keys / values of the batch); q and out rows are q_stride elements apart; same arithmetic as kf_attn_decode, one workgroup per
 * (kv-head, token), no scratch. */
int kf_attn_prefill(kf_ctx* ctx, const kf_bf16* q, const kf_bf16* kcache, const kf_bf16* vcache, kf_bf16* out, int pos0, int n_tok, int64_t q_stride,
                    int n_head, int n_kv, int hd, int kv_stride);

/* ---- the decode step's layer loop as ONE persistent launch (kf_engine.hip).  Replaces, for all layers of one token, the neuron walk of
 * Fish::ForwardOnRLS (gLLM.cpp:755-771) over SelfAttention::cuInfer (QKV.cu:617-702) and FFN::cuInfer (NeuronFuse.cu:615-656): the same
 * arithmetic as kf_norm_linear -> kf_attn_block -> kf_linear -> kf_norm_gateup_swiglu -> kf_linear per layer (every output bit equal), with the
 * five launches per layer turned into phases of one resident kernel that hand their vectors over through tagged granules in `workspace`.
 * One workgroup per CU; the launch must have the GPU's CUs to itself (every poll is bounded: a launch that cannot become fully resident sets the
 * engine's error word instead of hanging; kf_engine_check reads it).  w[7] = q k v o gate up down; all layers the same shapes and storage.
 * kf_engine_step returns KF_ENGINE_NOT_SERVED (1) when the position bound needs an attention slicing the engine does not restate: the caller
 * then issues the per-layer calls.  x_in / x_out: plain bf16 [dim] (the embedding row in, the residual stream after the last layer out; they may
 * alias).  pos_bound bounds the position as in the other decode entries (graph replay); the position itself is d_state[1]. */
typedef struct kf_engine kf_engine;
typedef struct kf_engine_layer {
    kf_weight w[7];
    const kf_bf16 *norm_in, *norm_post, *q_norm, *k_norm; /* q_norm / k_norm may be NULL */
    kf_bf16 *kcache, *vcache;                            /* layer base; row t at t*kv_stride elements */
    const int32_t* hot_ffn; /* sparse forward (D_matmul_sparse, GST_float.cpp:306-318): CS_Picker's hot[ffn] on the DEVICE, 1 = the gate / up row is computed, anything else = 0
                               (SwiGLU(0, 0) = 0: that element of the FFN's hidden vector is zero); cold rows are never read.  NULL: dense.  Read once per launch. */
} kf_engine_layer;
typedef struct kf_engine_desc {
    int32_t n_layer, dim, n_head, n_kv, head_dim, ffn, kv_stride;
    int32_t max_seq; /* rows of every layer's K / V cache (> 0): a launch whose positions reach past it is refused (KF_INVALID_ARGS from the bound, error word bit 64 from the state) */
    float rms_eps, qk_eps;
    const float* rope_table;
    const kf_engine_layer* layers; /* HOST array [n_layer] of device pointers */
} kf_engine_desc;
#define KF_ENGINE_NOT_SERVED 1
size_t kf_engine_workspace_bytes(const kf_engine_desc* desc);
/* workspace: device memory, 256-byte aligned, kf_engine_workspace_bytes(desc) bytes, owned by the caller, initialised here.
 * KF_UNSUPPORTED_DATATYPE: shapes / storage outside what the engine serves (the per-layer calls remain). */
int kf_engine_create(kf_ctx* ctx, const kf_engine_desc* desc, void* workspace, size_t workspace_bytes, kf_engine** out);
int kf_engine_step(kf_ctx* ctx, kf_engine* e, const kf_bf16* x_in, kf_bf16* x_out, const int32_t* d_state, int pos_bound);
/* TokenEmbed::cuInfer inside the launch: with a bf16 embedding table set, kf_engine_step may be given x_in == NULL and reads the row of the state's token
 * (or of d_forced[pos] when that is >= 0, as kf_embed_state does) itself.  NULL table: back to x_in.  KF_UNSUPPORTED_DATATYPE for other storages. */
int kf_engine_set_embedding(kf_ctx* ctx, kf_engine* e, const kf_weight* embed_bf16_or_null, const int32_t* d_forced_or_null);
/* Head4Token::cuInfer_1 (NeuronFuse.cu:842-862) + sample_argmax (GoPT.cpp:602-612) as trailing phases of the same launch: with a bf16 head [vocab, dim] and
 * the final RMSNorm weight set, kf_engine_step_head runs the layers, the final norm, the LM-head mat-vec (logits [vocab] bf16, every bit equal to
 * kf_norm_lm_head's) and -- pick != 0 -- the greedy first-maximum pick with the decode-state update of kf_norm_lm_head (d_tokens_out[pos] = id,
 * d_state = {id, pos + 1}): one launch per token instead of three.  A launch whose error word is set never advances the state.  NULL head: removed.
 * KF_UNSUPPORTED_DATATYPE for other head storages (the caller keeps kf_norm_lm_head). */
int kf_engine_set_head(kf_ctx* ctx, kf_engine* e, const kf_weight* head_bf16_or_null, const kf_bf16* final_norm_w, kf_bf16* logits, int32_t* d_tokens_out_or_null);
int kf_engine_step_head(kf_ctx* ctx, kf_engine* e, const kf_bf16* x_in, kf_bf16* x_out, int32_t* d_state, int pos_bound, int pick);
/* n_steps consecutive greedy decode steps in ONE launch (Fish::ForwardOnRLS + Head4Token::cuInfer_1 + sample_argmax, n_steps times: GoPT.cpp:1155-1180): the step of
 * kf_engine_step_head(x_in = NULL, pick = 1) repeated inside the kernel -- the id a step picks reaches the next step's embedding read as a tagged granule instead of
 * through a launch boundary (~8 us per step).  Every position d_state[1] .. d_state[1] + n_steps - 1 must lie under pos_bound (the bound fixes the attention slicing of
 * the launch); d_tokens_out, d_forced and d_state are read and written per step as by the single step; logits / x_out hold the last step's.  Same bits as n_steps launches. */
int kf_engine_steps_head(kf_ctx* ctx, kf_engine* e, kf_bf16* x_out, int32_t* d_state, int pos_bound, int n_steps);
/* Why a model is (not) served: KF_OK, or KF_ENGINE_NOT_SERVED with the reason in `why` (shape not instantiated, storage, device too small, ...): validation only, nothing is
 * allocated (desc->layers as for kf_engine_create).  Fish::EnsureEngine logs it once. */
int kf_engine_served(kf_ctx* ctx, const kf_engine_desc* desc, char* why, size_t why_bytes);
/* First-sweep delays of the hand-offs, measured on THIS device.  A consumer's first sweep of a hand-off vector is issued a fixed delay behind the moment its own workgroup
 * published its rows of the feeding phase; too early costs a second sweep, too late its lateness.  kf_engine_tune times the layers-only launch at the position d_state holds
 * (reads the state's token, rewrites that position's K / V rows with the values the real step is about to write; the state does not advance) and walks the six delays by
 * coordinate descent (`passes` <= 3 rounds of 8, 4, 2 sleep units; ~100 launches per round), for the attention slice count of `pos_bound`.  Results never depend on the
 * delays.  Needs the embedding table (kf_engine_set_embedding).  us_before / us_after (optional): mean launch time with the old and the chosen delays. */
int kf_engine_tune(kf_ctx* ctx, kf_engine* e, kf_bf16* x_out, const int32_t* d_state, int pos_bound, int passes, float* us_before, float* us_after);
typedef struct kf_engine_statistics {
    int32_t sweeps[6]; /* sweeps the poller of workgroup 0 issued since creation / reset, per hand-off: x (P1), q|k|v (P2), slice partials (P3), ao (P4), xB (P5), act (P6) */
    int32_t polls;     /* polls per hand-off in the same span (= layers stepped): sweeps[i] / polls = sweeps per poll, 1.0 when every first sweep came back complete */
    int32_t delay[6];  /* the delays in use at pos_bound's slice count (s_sleep units) */
    int32_t tuned;     /* 1: measured by kf_engine_tune on this device, 0: the built-in defaults */
} kf_engine_statistics;
int kf_engine_stats(kf_ctx* ctx, kf_engine* e, int pos_bound, kf_engine_statistics* out);
int kf_engine_check(kf_ctx* ctx, kf_engine* e); /* synchronises; KF_INTERNAL_ERR when a hand-off poll has timed out since creation or the last kf_engine_reset */
/* After KF_INTERNAL_ERR from kf_engine_check: the error word latches and every later launch of the engine returns at once without output.  kf_engine_reset
 * synchronises, puts the hand-off state back to its initial one and clears the word; the steps since the failure have to be redone. */
int kf_engine_reset(kf_ctx* ctx, kf_engine* e);
int kf_engine_destroy(kf_engine* e);

/* ---- EIGHT decoders per GPU, one per XCD, for up to 32 independent sequences (kf_xengine.hip; rounds 5 - 6).  The reference decodes ONE sequence per process (Fish::Chat, GoPT.cpp:1139-1180) and
 * scales a small model by running more processes; on this part a single sequence cannot keep the HBM busy (its step is a chain of hand-offs), so the replicas move
 * INSIDE the package: the 32 workgroups of XCD s are the decoder of sequence s, every hand-off stays in that XCD's L2, and the chip streams eight sequences' bytes at
 * once.  The sequences share the weights (the kf_engine_desc's layer table, the embedding, the head) and nothing else; per sequence: a K/V cache (sequence s at
 * layers[].kcache + s * kv_seq_stride elements), a decode state {token, pos, -, -} (d_state + 4 s), forced ids (d_forced + s * forced_stride), ids out, logits
 * (logits + s * vocab), the residual stream out (x_out + s * dim).  Each sequence's ids, logits and K/V rows are bit for bit those kf_engine_steps_head, the per-layer
 * calls and the oracle give for that sequence alone (canonical order only: kf_set_canonical(ctx, 0) is refused).  One workgroup per CU, 32 per XCD: the launch must have
 * the GPU to itself (bounded polls, error word, kf_xengine_check / _reset as for kf_engine).  Sequences may stand at different positions.
 * n_seq <= 8: sequence s on XCD s, one workgroup per CU.  More (round 6): still ONE decoder per XCD, each decoding 2 (n_seq <= 16) or 4 (n_seq <= 32) sequences -- XCD x takes
 * sequences x, x + 8, x + 16, x + 24.  Every 4-bit block a decoder streams is unpacked ONCE (the exact bf16-stepwise dequantisation, T.cu:274, is what bounds these engines) and
 * multiplied against the activations of all its sequences; each sequence keeps its own canonical chains, attention, K / V cache and state, so its bits do not change.
 * Served: 4-bit PackedQ (RTN, groups of 128) layers of the Qwen3-0.6B, 1.7B, 4B and 8B shapes (and two small test shapes), bf16 embedding / head; more than 8
 * sequences for the 0.6B / 1.7B shapes (1.7B: at most 16), more than 16 for the 0.6B shape.  Round 6: 1-bit and 2-bit PackedQ (YinYang) layers for the 0.6B shape and the 256-wide
 * test shape (1-bit: BASELINE config 5's storage), and FFNs with a hot-row mask (kf_engine_layer::hot_ffn: a cold gate / up row publishes a zero, D_matmul_sparse).  Weights are read in place and must not
 * change while an engine built on them lives; for the GQA-4 shapes (and the TP form below) the engine keeps a fused COPY of every layer's q | k | v rows in its workspace, made at
 * create time: after a weight update destroy the engine and create it again.
 * d_state [n_seq][4] = {token, pos, parked, status}.  parked != 0: the launch skips the sequence (the other sequences of its decoder go on).  A launch one of whose positions
 * would lie beyond a sequence's cache rows (pos + n_steps > max_seq) skips THAT sequence and sets bit 64 of its status word -- the others decode on; nothing is shared but
 * the weights, so one finished sequence never stops the rest (the caller clears the word when it re-aims the sequence). */
typedef struct kf_xengine kf_xengine;
#define KF_XENGINE_MAX_SEQ 32
size_t kf_xengine_workspace_bytes(const kf_engine_desc* desc);
int kf_xengine_create(kf_ctx* ctx, const kf_engine_desc* desc, int n_seq, int64_t kv_seq_stride, void* workspace, size_t workspace_bytes, kf_xengine** out);
int kf_xengine_served(kf_ctx* ctx, const kf_engine_desc* desc, char* why, size_t why_bytes); /* KF_OK or KF_ENGINE_NOT_SERVED + the reason */
/* TokenEmbed::cuInfer inside the launch (bf16 table, required); d_forced [n_seq][forced_stride] or NULL */
int kf_xengine_set_embedding(kf_ctx* ctx, kf_xengine* e, const kf_weight* embed_bf16, const int32_t* d_forced_or_null, int forced_stride);
/* Head4Token::cuInfer_1 + sample_argmax inside the launch: logits [n_seq][vocab] bf16, d_tokens_out [n_seq][tokens_stride] or NULL */
int kf_xengine_set_head(kf_ctx* ctx, kf_xengine* e, const kf_weight* head_bf16_or_null, const kf_bf16* final_norm_w, kf_bf16* logits, int32_t* d_tokens_out_or_null, int tokens_stride);
/* n_steps greedy decode steps of EVERY sequence in one launch (Fish::ForwardOnRLS + Head4Token::cuInfer_1 + sample_argmax per sequence and step): d_state [n_seq][4] is
 * read and advanced per step, x_out [n_seq][dim] holds the last step's residual stream.  pick = 0 (n_steps = 1 only): logits without the pick, the state stays. */
int kf_xengine_steps(kf_ctx* ctx, kf_xengine* e, kf_bf16* x_out, int32_t* d_state, int n_steps, int pick); /* (launched directly: refused inside kf_graph_begin / _end) */
int kf_xengine_check(kf_ctx* ctx, kf_xengine* e); /* synchronises; KF_INTERNAL_ERR when a hand-off poll has timed out since creation or the last kf_xengine_reset */
int kf_xengine_reset(kf_ctx* ctx, kf_xengine* e);
int kf_xengine_destroy(kf_xengine* e);
/* Tensor parallel over the XCDs (round 5): ONE sequence of a model whose TP = 8 ranks (koifish_amd/tp.py TPPlan; the partitioning of SURVEY 8e) run as the eight XCDs of one
 * launch -- rank r's 32 workgroups stream rank r's shards; q | k | v, the attention of the rank's kv-head and gate | up stay inside the XCD; the column shards (o_proj, down_proj)
 * leave as fp32 partials into every rank's receive area (the {value | generation} protocol of kf_tp_*, inside the launch), summed in rank order: the bits of kf_tp_reduce_recv.
 * Served: the ranks of Qwen3-32B (per rank: dim 5120, 8 / 1 heads of 128, ffn 3200), 4-bit PackedQ layers, bf16 embedding (replicated) and head (vocabulary shards).
 * rank_descs[r]: rank r's layers (its shards, its kv-head's cache rows: kv_stride = 128).  kf_xengine_set_embedding / _steps / _check / _reset / _destroy as above with
 * n_seq = 1 (d_state [4], x_out [dim]); logits: the FULL vector, the shards' rows in rank order.  No reference counterpart (QKV.cu:503 is single-GPU). */
size_t kf_xengine_workspace_bytes_tp(const kf_engine_desc* rank0_desc);
int kf_xengine_create_tp(kf_ctx* ctx, const kf_engine_desc* const* rank_descs, int world, void* workspace, size_t workspace_bytes, kf_xengine** out);
int kf_xengine_set_head_tp(kf_ctx* ctx, kf_xengine* e, const kf_weight* const* head_shards, const int32_t* row0, const kf_bf16* final_norm_w, kf_bf16* logits, int32_t* d_tokens_out_or_null, int tokens_stride);

#ifdef __cplusplus
}
#endif
#endif
