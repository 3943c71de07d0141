// kf_host.hpp -- host side of the decode path, ABOVE the C ABI (include/kf_abi.h): a C++ mirror of the
// reference's neuron / tensor interface for this path.  Class and method names, argument meaning and error
// behaviour follow gruai/koifish so that a maintainer can read one against the other:
//   GTensor            src/Tensor/GTensor.hpp:168-490        (data||gama blob, type, ne[], quant card)
//   KVCache            src/Utils/Cache.{hpp,cpp}:14-57       ([n_layer, max_seq, kv_dim] bf16 pair)
//   LayerNormal::cuFlow src/Manifold/Neuron.hpp:438-458, src/Device/CUDA/T.cu:561-573
//   SLP::Forw          src/Manifold/Neuron.hpp:397-430, src/Device/CUDA/NeuronFuse.cu:305-381
//   ROPE::cuInfer      src/Device/CUDA/kernel/rope.cu:645-672
//   SelfAttention::cuInfer / _devQKV   src/Device/CUDA/QKV.cu:617-702, src/Manifold/TGraph.cpp:198-207
//   FFN::cuInfer       src/Device/CUDA/NeuronFuse.cu:615-656
//   TokenEmbed::cuInfer src/Device/CUDA/NeuronFuse.cu:176-218
//   Head4Token::cuInfer_1 src/Device/CUDA/NeuronFuse.cu:842-862
//   Fish::ForwardOnRLS / Chat  src/Manifold/gLLM.cpp:722-787, src/Manifold/GoPT.cpp:1111-1235
// No HIP headers here: device memory and launches go through the kf_* C ABI only (plain g++ builds this file).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/kf_abi.h"

namespace koifish {

using floatX = kf_bf16;
enum class typNUMBER : uint8_t { F32 = 0, F64, F16, BF16, F8E5M2, F8E4M3, U8, I8, U16, I16, U32, I32, U64, I64, Q4, Q3, Q2, T_SIGN, T_SEQ, BOOL1, T_BINARY };

struct Fish;

#define KF_TRY(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != KF_OK) return rc_; \
    } while (0)

// Quant card fields that cross the kernel seam (QUANT_CARD / GeQuant, GeQuant.cpp:107-124)
struct QuantCard {
    int bits = 16, T_group = 128, qMin = 0, qMax = 0, qBias = 0;
    bool isNormalFloat = false;  // QUANT_MODE::RTNf (GeQuant.cpp:853): row-codebook storage -> kf_weight.quant = KF_QUANT_ROW_LUT
};

// A device tensor: `data||gama` in one allocation, as huTensor::Alloc lays it out (GTensor.cpp:456-510).
struct GTensor {
    std::string name;
    void* data = nullptr;
    size_t szData = 0, szGama = 0;
    typNUMBER type = typNUMBER::BF16;
    int ne[4] = {1, 1, 1, 1};
    QuantCard quant;
    bool owned = false;
    kf_ctx* ctx = nullptr;
    std::shared_ptr<GTensor> qZero, qScale;  // vendor AutoAWQ tensors (GeQuant.cpp:410): data = qweight [in, out/8]; ne = {out, in}

    ~GTensor();
    size_t size() const { return (size_t)ne[0] * ne[1] * ne[2] * ne[3]; }
    int nGroup() const { return (szGama && quant.T_group > 0) ? (int)(size() / quant.T_group) : 0; }
    floatX* gama_T() const { return szGama ? reinterpret_cast<floatX*>(reinterpret_cast<uint8_t*>(data) + szData) : nullptr; }
    kf_weight desc() const;  // what TASKA_quant / TASKA_AxB hand to the kernels
    // SerialGamaData H2D (huTensor.cu:413-458): allocate and copy a host `data||gama` blob
    int LoadBlob(kf_ctx* c, const void* h_blob, size_t nbytes);
    int Alloc(kf_ctx* c, size_t nbytes);
};
using hGTensor = std::shared_ptr<GTensor>;
hGTensor GT(kf_ctx* c, const std::string& name, typNUMBER tp, int n0, int n1 = 1);
inline floatX* ToX(const hGTensor& t) { return t ? reinterpret_cast<floatX*>(t->data) : nullptr; }

struct KVCache {
    enum CTYPE { KV_KEY = 0, KV_VAL };
    hGTensor key, val;
    int n_layer = 0, max_seq_len = 0, kv_dim = 0;
    int Init(kf_ctx* c, int n_layer, int max_seq, int kv_dim);
    void* Get(CTYPE type, int layer, int pos) const;  // Cache.cpp:42-57
};

struct GeNeuron {
    std::string name;
    Fish* hFish = nullptr;
    int layid = 0;  // 1-based like the reference (hCache->Get(.., layid - 1, 0))
    virtual ~GeNeuron() {}
};

struct LayerNormal : GeNeuron {
    hGTensor w, out;
    float rms_eps = 1e-6f;
    int nHead = 0, ldTH = 0;
    hGTensor cuFlow(hGTensor inpDelta, int flag = 0);
};

struct Relu : GeNeuron {
    int Forw(hGTensor out, hGTensor gate, hGTensor inp, int flag = 0);  // CU_swiglu_v0
};

struct SLP : GeNeuron {
    hGTensor w, b, out;
    int nIn = 0, nOut = 0;
    bool Empty() const { return !w; }
    // rhs = W.lhs (+b);  returns 0 or -1 like the reference (NeuronFuse.cu:305-381)
    int Forw(hGTensor rhs, hGTensor lhs, hGTensor toGelu = nullptr, Relu* hRelu = nullptr, int flag = 0);
    int Forw(floatX* rhs, const floatX* lhs, uint32_t epilogue = 0, const floatX* residual = nullptr);
};

struct SelfAttention;
struct ROPE : GeNeuron {
    LayerNormal *hnQ = nullptr, *hnK = nullptr;
    int n_head = 0, n_head_kv = 0, head_dim = 0;
    float theta = 1e6f;
    hGTensor cuInfer(SelfAttention* hQKV, uint32_t seed, int pos, int flag = 0);
};

struct SelfAttention : GeNeuron {
    LayerNormal norm, normQ, normK;
    SLP Q, K, V, proj_cat;
    ROPE rope;
    KVCache* hCache = nullptr;
    hGTensor out;
    int n_head = 0, n_head_kv = 0, head_dim = 0, q_dim = 0, kv_dim = 0;
    bool isSeparateQKV = true;
    void _devQKV(int pos);  // K.out / V.out alias the KV-cache row (TGraph.cpp:198-207)
    hGTensor cuInfer(hGTensor inpL, int flag = 0);
    // n tokens at positions pos0.. in one pass (the forward of SelfAttention::cuFlow, NeuronFuse.cu:692-731, with the decode arithmetic)
    int cuFlow(floatX* bx, int pos0, int n);
};

struct FFN : GeNeuron {
    LayerNormal norm;
    SLP gate, up, down;
    Relu relu;
    hGTensor out;
    int latent = 0;
    // sparse forward (CS_Picker::hot, SparseNeuron.cpp:20-29): the hot rows of gate / up as a device list; n_hot < 0 = dense
    hGTensor hot_rows, hot_mask;  // hot_mask: CS_Picker's hot[ffn] itself on the device (the persistent engine reads the mask, the per-layer launches the list)
    int n_hot = -1;
    hGTensor cuInfer(hGTensor hIn, int flag = 0);
    int cuFlow(floatX* bx, int n);
};

struct TokenEmbed : GeNeuron {
    hGTensor w, out;
    hGTensor cuInfer(int token, int flag = 0);
};

struct Head4Token : GeNeuron {
    SLP proj;
    hGTensor preLogits;
    hGTensor cuInfer_1(hGTensor inp_, int flag = 0);
};

struct MODEL_CARD {  // the fields of CLI_params / MODEL_CARD this path reads (CLI_params.hpp, cases/qwen3/*.json)
    int nEmbed = 0, nLayer = 0, n_head = 0, n_head_kv = 0, head_dim = 0, n_ff = 0, vocab = 0, n_ctx = 0;
    float rms_eps = 1e-6f, qk_eps = 1e-6f, rope_theta = 1e6f;
    bool tie_word_embeddings = true;
};

// Shared scratch, named after GST_MemBuffer's gBUFF members (GST_MemBuffer.hpp:46-48)
struct MemBuffer {
    hGTensor tmpFF1;    // Q.out
    hGTensor kraw;      // new key before norm/rope (fused path)
    hGTensor scratch;   // proj_cat out / gate-act
    hGTensor delta;     // down out
    hGTensor upOut;     // up out (unfused path)
    hGTensor normed;    // norm.out
    hGTensor attn_ws;   // split-KV partials
    hGTensor head_ws;   // arg-max partials
    hGTensor residual;  // alias, not owned
    // token-batch (prefill) activations, [prefill_chunk, .] rows, allocated by the first Prefill
    hGTensor bX, bNorm, bQ, bAttn, bGate, bUp;
    int32_t* d_ptok = nullptr;
    int rows = 0;       // rows of the token-batch buffers
};

// CHAT_SAMPLER (CLI_params.hpp:663-683): the fields GeneratOnPrompt::Sample reads.  temperature == 0 or top_k == 1 -> sample_argmax.
// (The reference defaults to temperature 0.6; this host defaults to greedy, the mode every parity test and the bench use.)
struct CHAT_SAMPLER {
    float temperature = 0.0f;
    float top_p = 0.95f;
    int top_k = 50;
    uint64_t seed = 42;
    bool true_topk = false;  // false: the candidate set the reference's TOPK_heap keeps (kf_sample); true: the k largest logits (kf_sample_topk)
    bool greedy() const { return temperature == 0.0f || top_k == 1; }
};

// Tensor-parallel decode (no counterpart in the single-GPU reference, QKV.cu:503): this Fish holds ONE rank's shard -- the
// card carries the LOCAL head / ffn / vocab counts and the full nEmbed -- and exchanges fp32 partials through peer-mapped receive areas
// (kf_tp_* of the ABI).  q/k/v/gate/up are row shards, o_proj / down_proj column shards, the head a vocabulary shard, the embedding replicated.
struct TPState {
    int rank = 0, world = 1, vocab_row0 = 0;
    kf_tp_comm comm;
    void* area = nullptr;
    bool committed = false;             // kf_tp_commit done for the current peer set
    std::vector<void*> opened;          // IPC mappings of other processes' areas (closed with the Fish)
    std::vector<kf_graph*> group_graphs;  // ranks of ONE process stepped in lock-step on one stream: rank 0 keeps the group's graphs, one per bucket
};

struct Fish {
    MODEL_CARD config;
    TPState tp;
    int TPInit(int rank, int world, int vocab_row0);  // allocates this rank's receive area; peers are set afterwards
    int TPSetPeer(int r, void* area);
    // One phase of a TP step: 0 embed | 1 attention half of `layer` up to the o_proj push | 2 its reduce | 3 FFN half up to the down_proj push |
    // 4 its reduce | 5 LM-head shard + arg-max push | 6 pick.  A rank that owns its GPU enqueues all of them in order (EnqueueStep); ranks that
    // share one stream are enqueued phase by phase across ranks (a reduce must not sit in the queue ahead of the pushes it waits for).
    int TPCommit();
    int TPPhase(int phase, int layer);
    int EnqueueStepTP();
    CHAT_SAMPLER samp_params;
    uint64_t* d_rng = nullptr;  // LogitsInfo::rng_state on the device
    int SetSampler(const CHAT_SAMPLER& s);  // also (re)seeds the device rng state
    int HeadAndPick(const floatX* x_last);  // [final norm + LM head] then arg-max or Sample, and the decode-state update
    kf_ctx* ctx = nullptr;
    int fuse_level = 1;  // 0: one launch per reference kernel; 1: fused launches (5 per layer)
    // the layer loop of a decode step as one persistent launch (kf_engine_*): used by EnqueueStep when fuse_level >= 1, the model's shapes and
    // storage are served and the position bound is; otherwise the per-layer launches run.  Same arithmetic, bit for bit.
    void* lin_scratch = nullptr;  // kf_set_scratch: workspace of the dequantise-then-multiply storages (AutoAWQ, row forms), sized when weights are set
    size_t lin_scratch_bytes = 0;
    void* deq_arena = nullptr;  // kf_set_dequant_arena: resident bf16 copies of the layers' quantised matrices for long prompts (EnsureResident)
    size_t deq_arena_bytes = 0, resident_max_bytes = (size_t)16 << 30;
    // OFF by default: the copies are keyed by the blobs' addresses, so a caller that hands weights over as device pointers (kfh_set_weight is_device = 1) and later
    // re-quantises or updates them IN PLACE would prefill on stale bf16 copies while the decode reads the live data.  kfh_set_prefill_resident opts in and takes that promise
    // ("the quantised data does not change under me"); kfh_weights_changed drops the copies (and the engine's table) after an in-place update.
    int prefill_resident = 0;
    bool resident_tried = false;
    int EnsureResident(int PC);
    void DropResident();
    int EnsureLinearScratch(const kf_weight& w, int nTok);
    bool use_engine = true;
    kf_engine* engine = nullptr;
    void* engine_ws = nullptr;
    int engine_state = 0;  // 0 not tried, 1 built, -1 not served
    unsigned weights_gen = 0;  // counts DropEngine calls (a weight set again, kfh_weights_changed): engines other objects built on this Fish compare it with their own copy
    bool engine_embed = false;  // the engine reads the (bf16) embedding row itself
    bool engine_head = false;   // ... and runs the final norm, the (bf16) LM head and the greedy pick as trailing phases of its launch
    int masked_layers = 0;      // layers with a hot-row mask (kfh_set_hot): the engine walks dense FFNs only
    int EnsureEngine();
    void DropEngineTable();     // the engine's device table and the captured graphs only (a hot-row mask changed)
    void DropEngine();          // weights / norms / caches changed: the engine's device table and the captured graphs hold stale pointers
    int EngineCheck();          // synchronises; a timed-out hand-off is reported ONCE (KF_INTERNAL_ERR), the engine reset so that later steps run again
    int engine_steps = 0;  // steps enqueued (or captured) through the engine
    std::string engine_why;  // why the engine does not serve this model (kf_engine_served), "" when it does
    int engine_autotune = 0;  // > 0: passes of kf_engine_tune run once per position bucket, at the first step inside it (kfh_set_engine_autotune; off in the library: ~0.2 s of
                              // host-blocking launches inside RunSteps -- bench.py opts in).  A failed tune is never fatal: the defaults stay, the engine is reset, the decode goes on.
    std::vector<unsigned char> bucket_tuned;
    KVCache cache;
    MemBuffer gBUFF;
    TokenEmbed embed;
    std::vector<std::unique_ptr<SelfAttention>> attn;
    std::vector<std::unique_ptr<FFN>> ffn;
    LayerNormal final_norm;
    Head4Token head;
    float* rope_table = nullptr;  // device [n_ctx][hd/2][2]
    int32_t* d_state = nullptr;   // {token, pos, 0, 0}
    int32_t* d_forced = nullptr;  // [n_ctx] teacher-forced ids, -1 = free running
    int32_t* d_tokens_out = nullptr;  // [n_ctx] greedy id produced at each position
    int tok_pos = 0;              // hBatch->tok_pos
    bool graph_mode = false;      // launches take position/token from d_state
    bool state_tokens = false;    // eager per-kernel steps (fuse_level 0, e.g. AutoAWQ weights): position from the host, token from d_state
    std::vector<kf_graph*> graphs;  // one per position bucket
    std::vector<int> graph_bound;
    hGTensor x;                   // residual stream

    ~Fish();
    // the attention kernels are instantiated for query-group sizes 1, 2, 4, 8 and head_dim 64 / 128 (every Qwen3 size except the 14B's group of 5)
    static bool ShapeServed(const MODEL_CARD& card, std::string& why);
    int Build(const MODEL_CARD& card, int device, void* stream);
    int pos_bound() const;  // launch-geometry bound for the current bucket
    // one token through every neuron (Fish::ForwardOnRLS, gLLM.cpp:722-787)
    int ForwardOnRLS(int token, int pos);
    int EnqueueStep(int bound);  // the launch sequence of one step, positions/tokens from d_state
    kf_graph* GraphFor(int pos);
    int SetState(int token, int pos);
    // token-serial prefill then greedy decode (Fish::Chat, GoPT.cpp:1139-1159); ids of the new tokens to out
    int Generate(const int* prompt, int n_prompt, int n_new, int* out, bool use_graph);
    // replay n decode steps starting at `pos` with whatever d_forced holds (bench / long runs); no host sync
    bool OneLaunchStep(); /* the decode step is one kernel launch: launched directly, no graph */
    int RunSteps(int pos, int n, bool use_graph);
    // n prompt tokens at positions pos0.. through every layer as token batches (MFMA tile kernels), then the head on the last one:
    // afterwards the KV cache holds rows pos0..pos0+n-1, d_state = {greedy next token, pos0+n}, d_tokens_out[pos0+n-1] = that token.
    // The reference prefills token by token (Fish::Chat, GoPT.cpp:1139-1146); same arithmetic per token, fp32 sums in MFMA order.
    int Prefill(const int* tokens, int n, int pos0);
    int PrefillReady(int min_rows = 0);  // the token-batch buffers ([rows >= min_rows]), scratch and resident copies: allocated by the first prefill, grown by a larger batch
    // rows per token batch (clamped to n_ctx).  Measured, Qwen3-0.6B 4-bit, prompt filling the context: 2047 tokens 20.9 / 15.5 / 9.8 ms at 512 / 1024 / 2048 rows,
    // 8191 tokens 92.3 / 53.9 / 42.7 / 36.6 ms at 1024 / 2048 / 4096 / 8192 (scratch/prefill_chunk.py): the tile kernels want many rows per launch.
    int prefill_chunk = 8192;
    int prefill_mode = 0;  // Generate: 0 token-serial prefill like the reference, 1 batched
};

// Eight independent decoders on ONE GPU, one per XCD, sharing a Fish's weights (kf_xengine_* of the ABI; round 5).  The reference decodes one sequence per process
// (Fish::Chat, GoPT.cpp:1139-1180) and scales a small model by running more processes; here the "processes" are the 32-workgroup halves of one launch: sequence s has its own
// KVCache (Cache.cpp:14-57, one per sequence), decode state, forced ids, ids out and logits -- everything the reference's per-process Fish owns except the weights.
// Every sequence's ids / logits / K / V rows are those Fish::RunSteps produces for it alone (canonical order), bit for bit.
struct XcdReplicas {
    Fish* hFish = nullptr;  // the weights (not owned)
    int n_seq = 0;
    kf_xengine* engine = nullptr;
    void* engine_ws = nullptr;
    hGTensor key, val;      // [n_seq][n_layer][n_ctx][kv_dim] bf16
    hGTensor logits, x;     // [n_seq][vocab], [n_seq][nEmbed]
    int32_t* d_state = nullptr;       // [n_seq][4]: {token, pos, parked, status}
    int32_t* d_forced = nullptr;      // [n_seq][n_ctx], -1 = free running
    int32_t* d_tokens_out = nullptr;  // [n_seq][n_ctx]
    std::string why;        // why the model is not served, "" when it is
    int steps_per_launch = 32;
    long long steps_run = 0;
    size_t engine_ws_bytes = 0;
    unsigned built_gen = 0;  // Fish::weights_gen the engine was built on
    ~XcdReplicas();
    int Build(Fish* f, int n_seq_);
    int MakeEngine(bool allocate);
    int Fresh();          // re-creates the engine when the Fish's weights changed since it was built (kfh_weights_changed, a weight set again)
    int Park(int seq, bool on);
    int Status(int seq, int32_t* out4);
    int SetForced(int seq, const int32_t* ids, int n);
    int SetState(int seq, int token, int pos);
    int Prefill(int seq, const int* tokens, int n);  // the sequence's prompt through Fish::Prefill (token batches on the tile kernels), its K / V rows into the sequence's cache; the sequence then stands behind the prompt
    int RunSteps(int n);  // n greedy steps of EVERY sequence from wherever each stands; no host sync
    // a queue of prompts answered through the slots (Fish::Chat's rounds over DEBUG.prompts, GoPT.cpp:1111-1180, n_seq at once); see kf_host.cpp
    int Chat(const int32_t* prompts, const int32_t* prompt_len, int n_req, int stride, int max_new, int eos, int32_t* out, int32_t* out_len, long long* stats,
             const int32_t* max_new_each = nullptr);  // max_new_each: a limit of its own per request (<= max_new)
    // S prompts at once (rows = S x T, T = the longest, shorter ones padded): ONE token batch through the tile kernels -- the rows of a prompt attend to that prompt only
    // (kf_attn_prefill_batch), positions restart per prompt (kf_qknorm_rope_train), every prompt's K / V rows scattered into its slot's cache (kf_copy_blocks), the head on
    // each prompt's last row.  What Fish::Chat's token-serial prefill loop (GoPT.cpp:1139-1146) does for one prompt, for S of them in the launches of one.
    int PrefillBatch(const int* slots, const int32_t* tokens, const int* lens, int S, int stride);
    int prefill_batch = 1;        // Chat: prompts prefilled together when several slots are free (1: one by one, the bits of Fish::Prefill)
    hGTensor bK, bV, bL;          // [rows, kv_dim] K / V rows of a prompt batch before they are scattered; [n_seq, vocab] the batch's logits
    void** d_dst = nullptr;       // [3][n_seq] device tables of the scatters' destinations (K, V, logits), then the batch's slots as int32
    CHAT_SAMPLER samp_params;     // Chat's sampler (greedy by default); non-greedy: one launch per token, kf_sample per occupied slot
    uint64_t* d_rng = nullptr;    // [n_seq] xorshift states, seeded per request (seed + request index)
    int SetSampler(const CHAT_SAMPLER& s);
    int Check();          // synchronises; a timed-out hand-off is reported once (KF_INTERNAL_ERR) and the engine reset
    size_t kv_seq_elems() const;
};

// ONE sequence of a model split over eight tensor-parallel ranks, the ranks as the eight XCDs of ONE launch (kf_xengine_create_tp): the Fish of every rank holds its shards
// (koifish_amd/tp.py build_native_rank); this object owns the ranks' K / V rows, the sequence's state / forced ids / ids out, the full logits vector.
struct XcdTP {
    std::vector<Fish*> ranks;  // not owned
    kf_xengine* engine = nullptr;
    void* engine_ws = nullptr;
    hGTensor key, val;         // [rank][n_layer][n_ctx][kv_dim of a rank] bf16
    hGTensor logits, x;        // [vocab] (the shards in rank order), [nEmbed]
    int32_t* d_state = nullptr;
    int32_t* d_forced = nullptr;
    int32_t* d_tokens_out = nullptr;
    int vocab = 0;
    std::string why;
    int steps_per_launch = 16;
    std::vector<unsigned> built_gen;  // the ranks' Fish::weights_gen at Build: a rank whose weights changed since makes RunSteps refuse (the engine holds a fused copy of q | k | v)
    ~XcdTP();
    int Build(Fish** fs, int world);
    int SetForced(const int32_t* ids, int n);
    int SetState(int token, int pos);
    int RunSteps(int n);
    int Check();
};

}  // namespace koifish
