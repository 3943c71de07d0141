// kf_train.cpp -- ONE whole training step of the hybrid-precision GPT-2 of BASELINE config 3, sequenced in C++ behind the C ABI (part of libkf_host.so).
//
// The reference's step ("training"):  Fish::Train -> Optimizer loop -> Fish::ForwardOnRLS (gLLM.cpp:722-787: the neurons in graph order) ->
// BackwardOnRLS (gLLM.cpp:656-686: the reverse walk, SLP::Back NeuronFuse.cu:495-563) -> Optimizer::UpdateTensorParam -> CU_adamw_ (Optimizer.cu:135-160) ->
// the re-quantisation of every updated matrix (CU_XtoQ128_ / Float2T<f8e5>, T.cu:105-175).  Here the same order as three straight loops over a TABLE of device
// pointers: the caller (Python owns the buffers: koifish_amd/train_step.py; a C++ host would kf_malloc them) registers every trained tensor {master, gradient, two
// moments, optional quantised blob} and every kept activation once; a step is then ABI calls only -- no allocation, no host <-> device copy, no host sync.
//
// Order of the registered tensors (kfh_gpt2_set_param index):  per block l, 12 in a row: qkv.w qkv.b proj.w proj.b fc.w fc.b proj2.w proj2.b ln1.w ln1.b ln2.w
// ln2.b;  then wte, wpe, lnf.w, lnf.b.   The AdamW seed of tensor i at optimizer step t is seed + 7919 t + i (one seed per launch, as the reference draws one per
// tensor update).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "kf_host.hpp"

namespace koifish {

struct TrainTensor {
    kf_bf16 *p = nullptr, *g = nullptr;
    void *m = nullptr, *v = nullptr;
    long long n = 0;
    bool decay = false, has_blob = false, requant = false;
    kf_weight blob;  // what the forward multiplies (f8e5m2 / 4-bit PackedQ / the bf16 master itself for the tied head)
};

// the activations one block keeps for its backward (nothing is recomputed)
struct BlockActs {
    kf_bf16 *x, *h1, *qkv, *att, *x2, *h2, *f, *g;
    float *m1, *r1, *m2, *r2;
};

struct GPT2Trainer {
    kf_ctx* ctx = nullptr;
    int C = 0, H = 0, NL = 0, V = 0, Vp = 0, B = 0, T = 0, N = 0, hd = 0;
    std::vector<TrainTensor> params;
    std::vector<BlockActs> acts;
    kf_bf16 *xf = nullptr, *hf = nullptr, *logits = nullptr, *dx = nullptr, *dh = nullptr, *dqkv = nullptr, *datt = nullptr, *d4 = nullptr;
    float *mf = nullptr, *rf = nullptr, *losses = nullptr;
    void *sc_lin = nullptr, *sc_ln = nullptr, *sc_at = nullptr;
    const int32_t* ids = nullptr;  // of the last Forward (the embedding backward scatters by them)
    long long t = 0;               // optimizer steps taken

    enum { QKV_W = 0, QKV_B, PROJ_W, PROJ_B, FC_W, FC_B, PROJ2_W, PROJ2_B, LN1_W, LN1_B, LN2_W, LN2_B, PER_BLOCK };
    TrainTensor& P(int l, int k) { return params[(size_t)l * PER_BLOCK + k]; }
    TrainTensor& Wte() { return params[(size_t)NL * PER_BLOCK]; }
    TrainTensor& Wpe() { return params[(size_t)NL * PER_BLOCK + 1]; }
    TrainTensor& LnfW() { return params[(size_t)NL * PER_BLOCK + 2]; }
    TrainTensor& LnfB() { return params[(size_t)NL * PER_BLOCK + 3]; }

    int Ready() const {
        for (const TrainTensor& e : params)
            if (!e.p || !e.g || !e.m || !e.v || e.n < 8 || (e.n & 7)) return KF_INVALID_ARGS;
        for (const BlockActs& a : acts)
            if (!a.x || !a.h1 || !a.qkv || !a.att || !a.x2 || !a.h2 || !a.f || !a.g || !a.m1 || !a.r1 || !a.m2 || !a.r2) return KF_INVALID_ARGS;
        if (!xf || !hf || !logits || !dx || !dh || !dqkv || !datt || !d4 || !mf || !rf || !losses || !sc_lin || !sc_ln || !sc_at) return KF_INVALID_ARGS;
        for (int l = 0; l < NL; l++)
            for (int k : {QKV_W, PROJ_W, FC_W, PROJ2_W})
                if (!params[(size_t)l * PER_BLOCK + k].has_blob) return KF_INVALID_ARGS;
        return params[(size_t)NL * PER_BLOCK].has_blob ? KF_OK : KF_INVALID_ARGS;
    }
    // SLP::Forw (NeuronFuse.cu:305-381): y = x . W^T + b (+ residual)
    int Lin(TrainTensor& w, const kf_bf16* x, kf_bf16* y, TrainTensor* bias, const kf_bf16* res) {
        return kf_linear(ctx, &w.blob, x, y, bias ? bias->p : nullptr, N, 1.0f, 0.0f, res ? 1u : 0u, res);
    }
    int LN(const kf_bf16* x, TrainTensor& w, TrainTensor& b, kf_bf16* y, float* mean, float* rstd) { return kf_layernorm(ctx, x, w.p, b.p, y, N, C, 1e-5f, mean, rstd); }
    // SLP::Back (NeuronFuse.cu:495-563): weight / bias gradients into the tensors' own buffers, delta to the layer below
    int LinBack(TrainTensor& w, const kf_bf16* dIn, const kf_bf16* inp, kf_bf16* delta, TrainTensor* bias) {
        return kf_linear_backward(ctx, &w.blob, dIn, inp, delta, w.g, bias ? bias->g : nullptr, N, 0, sc_lin);
    }
    // kf_norm_backward ADDS into dweight / dbias: the per-tensor gradients are zero here (kf_adamw zeroes what it consumed)
    int LNBack(kf_bf16* dxx, const kf_bf16* dout, const kf_bf16* inp, TrainTensor& w, TrainTensor& b, const float* mean, const float* rstd) {
        return kf_norm_backward(ctx, dxx, w.g, b.g, dout, inp, w.p, mean, rstd, N, C, sc_ln);
    }

    // TokenEmbed, NL x [LayerNorm, qkv, causal attention, proj + residual, LayerNorm, fc, GELU, proj2 + residual], LayerNorm, tied head, fused classifier:
    // per-row losses in `losses`, the logit gradients of the MEAN loss in `logits`
    int Forward(const int32_t* d_ids, const int32_t* d_tgt) {
        KF_TRY(Ready());
        if (!d_ids || !d_tgt) return KF_INVALID_ARGS;
        KF_TRY(kf_embed_pos(ctx, Wte().p, C, Wpe().p, d_ids, B, T, C, Vp, acts[0].x));
        for (int l = 0; l < NL; l++) {
            BlockActs& a = acts[l];
            KF_TRY(LN(a.x, P(l, LN1_W), P(l, LN1_B), a.h1, a.m1, a.r1));
            KF_TRY(Lin(P(l, QKV_W), a.h1, a.qkv, &P(l, QKV_B), nullptr));
            KF_TRY(kf_attn_prefill_batch_strided(ctx, a.qkv, a.qkv + C, a.qkv + 2 * C, a.att, T, 3 * (int64_t)C, C, H, H, hd, 3 * C, B)); /* q / k / v: column blocks of the fused rows */
            KF_TRY(Lin(P(l, PROJ_W), a.att, a.x2, &P(l, PROJ_B), a.x));
            KF_TRY(LN(a.x2, P(l, LN2_W), P(l, LN2_B), a.h2, a.m2, a.r2));
            KF_TRY(Lin(P(l, FC_W), a.h2, a.f, &P(l, FC_B), nullptr));
            KF_TRY(kf_gelu(ctx, a.f, a.g, (size_t)N * 4 * C));
            KF_TRY(Lin(P(l, PROJ2_W), a.g, l + 1 < NL ? acts[l + 1].x : xf, &P(l, PROJ2_B), a.x2));
        }
        KF_TRY(LN(xf, LnfW(), LnfB(), hf, mf, rf));
        KF_TRY(Lin(Wte(), hf, logits, nullptr, nullptr));
        KF_TRY(kf_memset(ctx, losses, 0, (size_t)N * 4));
        KF_TRY(kf_fused_classifier(ctx, logits, losses, nullptr, 1.0f / (float)N, d_tgt, B, T, V, Vp, nullptr, 1));
        ids = d_ids;
        return KF_OK;
    }
    int Backward() {
        KF_TRY(Ready());
        if (!ids) return KF_INVALID_ARGS;
        if (Vp > V) KF_TRY(kf_memset2d(ctx, logits + V, (size_t)Vp * 2, 0, (size_t)(Vp - V) * 2, (size_t)N)); /* the padded vocabulary columns carry no gradient */
        KF_TRY(LinBack(Wte(), logits, hf, dh, nullptr));
        KF_TRY(kf_memset(ctx, dx, 0, (size_t)N * C * 2));
        KF_TRY(LNBack(dx, dh, xf, LnfW(), LnfB(), mf, rf));
        for (int l = NL - 1; l >= 0; l--) {
            BlockActs& a = acts[l];
            KF_TRY(LinBack(P(l, PROJ2_W), dx, a.g, d4, &P(l, PROJ2_B)));
            KF_TRY(kf_gelu_backward(ctx, d4, a.f, (size_t)N * 4 * C));
            KF_TRY(LinBack(P(l, FC_W), d4, a.h2, dh, &P(l, FC_B)));
            KF_TRY(LNBack(dx, dh, a.x2, P(l, LN2_W), P(l, LN2_B), a.m2, a.r2));
            KF_TRY(LinBack(P(l, PROJ_W), dx, a.att, datt, &P(l, PROJ_B)));
            KF_TRY(kf_attn_backward(ctx, a.qkv, a.qkv + C, a.qkv + 2 * C, 3 * (long long)C, a.att, datt, C, dqkv, dqkv + C, dqkv + 2 * C, 3 * (long long)C, T, H, H, hd, B, sc_at));
            KF_TRY(LinBack(P(l, QKV_W), dqkv, a.h1, dh, &P(l, QKV_B)));
            KF_TRY(LNBack(dx, dh, a.x, P(l, LN1_W), P(l, LN1_B), a.m1, a.r1));
        }
        return kf_embed_backward(ctx, Wte().g, C, Wpe().g, dx, ids, B, T, C, Vp);
    }
    // CU_adamw_ on every tensor (its own master, moments and gradient; seeded stochastic rounding), then the re-quantisation of every quantised matrix from its
    // updated master.  kf_adamw zeroes the gradients it has consumed.
    int Update(float lr, double beta1, double beta2, float eps, float wd, uint32_t seed) {
        KF_TRY(Ready());
        t++;
        const float b1c = (float)(1.0 - std::pow(beta1, (double)t)), b2c = (float)(1.0 - std::pow(beta2, (double)t)); /* the bias corrections, in double like the host side of the reference */
        for (size_t i = 0; i < params.size(); i++) {
            TrainTensor& e = params[i];
            KF_TRY(kf_adamw(ctx, e.p, e.g, e.m, e.v, (size_t)e.n, KF_BF16, lr, (float)beta1, (float)beta2, b1c, b2c, eps, e.decay ? wd : 0.0f, 1.0f,
                            (uint32_t)((seed + 7919ull * (unsigned long long)t + i) & 0xFFFFFFFFull), nullptr));
            if (e.requant) KF_TRY(kf_quantize(ctx, &e.blob, e.p, 0));
        }
        return KF_OK;
    }
};

}  // namespace koifish

using koifish::GPT2Trainer;

extern "C" {
void* kfh_gpt2_create(kf_ctx* ctx, int C, int H, int NL, int V, int Vp, int B, int T) {
    if (!ctx || C < 8 || H < 1 || C % H || NL < 1 || V < 1 || Vp < V || B < 1 || T < 1) return nullptr;
    GPT2Trainer* g = new GPT2Trainer;
    g->ctx = ctx, g->C = C, g->H = H, g->NL = NL, g->V = V, g->Vp = Vp, g->B = B, g->T = T, g->N = B * T, g->hd = C / H;
    g->params.resize((size_t)NL * GPT2Trainer::PER_BLOCK + 4);
    g->acts.resize(NL);
    memset(g->acts.data(), 0, sizeof(koifish::BlockActs) * NL);
    return g;
}
void kfh_gpt2_destroy(void* h) { delete reinterpret_cast<GPT2Trainer*>(h); }
int kfh_gpt2_n_params(void* h) { return (int)reinterpret_cast<GPT2Trainer*>(h)->params.size(); }
// blob: the descriptor of what the forward reads (null: the tensor is not multiplied as a weight); requant != 0: kf_quantize(blob, master) after every update
int kfh_gpt2_set_param(void* h, int index, void* p, void* g, void* m, void* v, long long n, int decay, const kf_weight* blob, int requant) {
    GPT2Trainer* t = reinterpret_cast<GPT2Trainer*>(h);
    if (index < 0 || index >= (int)t->params.size() || !p || !g || !m || !v || n < 8 || (n & 7)) return KF_INVALID_ARGS;
    koifish::TrainTensor& e = t->params[index];
    e.p = (kf_bf16*)p, e.g = (kf_bf16*)g, e.m = m, e.v = v, e.n = n, e.decay = decay != 0, e.has_blob = blob != nullptr, e.requant = blob && requant;
    if (blob) e.blob = *blob;
    return KF_OK;
}
// ptrs: x h1 m1 r1 qkv att x2 h2 m2 r2 f g
int kfh_gpt2_set_block_acts(void* h, int layer, void* const* ptrs) {
    GPT2Trainer* t = reinterpret_cast<GPT2Trainer*>(h);
    if (layer < 0 || layer >= t->NL || !ptrs) return KF_INVALID_ARGS;
    koifish::BlockActs& a = t->acts[layer];
    a.x = (kf_bf16*)ptrs[0], a.h1 = (kf_bf16*)ptrs[1], a.m1 = (float*)ptrs[2], a.r1 = (float*)ptrs[3], a.qkv = (kf_bf16*)ptrs[4], a.att = (kf_bf16*)ptrs[5];
    a.x2 = (kf_bf16*)ptrs[6], a.h2 = (kf_bf16*)ptrs[7], a.m2 = (float*)ptrs[8], a.r2 = (float*)ptrs[9], a.f = (kf_bf16*)ptrs[10], a.g = (kf_bf16*)ptrs[11];
    return KF_OK;
}
// ptrs: xf hf mf rf logits losses dx dh dqkv datt d4 scratch_linear_backward scratch_norm_backward scratch_attn_backward
int kfh_gpt2_set_buffers(void* h, void* const* ptrs) {
    GPT2Trainer* t = reinterpret_cast<GPT2Trainer*>(h);
    if (!ptrs) return KF_INVALID_ARGS;
    t->xf = (kf_bf16*)ptrs[0], t->hf = (kf_bf16*)ptrs[1], t->mf = (float*)ptrs[2], t->rf = (float*)ptrs[3], t->logits = (kf_bf16*)ptrs[4], t->losses = (float*)ptrs[5];
    t->dx = (kf_bf16*)ptrs[6], t->dh = (kf_bf16*)ptrs[7], t->dqkv = (kf_bf16*)ptrs[8], t->datt = (kf_bf16*)ptrs[9], t->d4 = (kf_bf16*)ptrs[10];
    t->sc_lin = ptrs[11], t->sc_ln = ptrs[12], t->sc_at = ptrs[13];
    return KF_OK;
}
int kfh_gpt2_forward(void* h, const int32_t* d_ids, const int32_t* d_tgt) { return reinterpret_cast<GPT2Trainer*>(h)->Forward(d_ids, d_tgt); }
int kfh_gpt2_backward(void* h) { return reinterpret_cast<GPT2Trainer*>(h)->Backward(); }
int kfh_gpt2_update(void* h, float lr, double beta1, double beta2, float eps, float wd, uint32_t seed) {
    return reinterpret_cast<GPT2Trainer*>(h)->Update(lr, beta1, beta2, eps, wd, seed);
}
int kfh_gpt2_step(void* h, const int32_t* d_ids, const int32_t* d_tgt, float lr, double beta1, double beta2, float eps, float wd, uint32_t seed) {
    GPT2Trainer* t = reinterpret_cast<GPT2Trainer*>(h);
    KF_TRY(t->Forward(d_ids, d_tgt));
    KF_TRY(t->Backward());
    return t->Update(lr, beta1, beta2, eps, wd, seed);
}
long long kfh_gpt2_steps_taken(void* h) { return reinterpret_cast<GPT2Trainer*>(h)->t; }
}
