// kf_host.cpp -- see kf_host.hpp.  Plain C++17, no HIP headers; every device action is a kf_* ABI call.
#include "kf_host.hpp"

#include <cassert>
#include <cstdio>
#include <cstring>

namespace koifish {

// ------------------------------------------------------------------------------------------------ GTensor
GTensor::~GTensor() {
    if (owned && data && ctx) kf_free(ctx, data);
}
kf_weight GTensor::desc() const {
    kf_weight w;
    std::memset(&w, 0, sizeof(w));
    w.data = data;
    w.gama = gama_T();
    w.type = (int32_t)type;
    w.ne0 = ne[0], w.ne1 = ne[1];
    w.lGroup = quant.T_group;
    w.nGroup = nGroup();
    w.qMin = quant.qMin, w.qMax = quant.qMax, w.qBias = quant.qBias;
    if (quant.isNormalFloat) w.quant = KF_QUANT_ROW_LUT, w.nGroup = 0, w.lGroup = 0;
    if (qZero && qScale) w.qzeros = qZero->data, w.qscales = qScale->data, w.nGroup = (int)(size() / 128), w.gama = nullptr;
    return w;
}
int GTensor::Alloc(kf_ctx* c, size_t nbytes) {
    ctx = c;
    KF_TRY(kf_malloc(c, nbytes, &data));
    owned = true;
    return KF_OK;
}
int GTensor::LoadBlob(kf_ctx* c, const void* h_blob, size_t nbytes) {
    KF_TRY(Alloc(c, nbytes));
    return kf_h2d(c, data, h_blob, nbytes);
}
hGTensor GT(kf_ctx* c, const std::string& name, typNUMBER tp, int n0, int n1) {
    auto t = std::make_shared<GTensor>();
    t->name = name, t->type = tp, t->ne[0] = n0, t->ne[1] = n1;
    size_t bytes = (size_t)n0 * n1 * (tp == typNUMBER::F32 || tp == typNUMBER::I32 ? 4 : 2);
    t->szData = bytes;
    if (t->Alloc(c, bytes) != KF_OK) return nullptr;
    kf_memset(c, t->data, 0, bytes);
    return t;
}

// ------------------------------------------------------------------------------------------------ KVCache
int KVCache::Init(kf_ctx* c, int nl, int ms, int kvd) {
    n_layer = nl, max_seq_len = ms, kv_dim = kvd;
    key = GT(c, "kv.key", typNUMBER::BF16, nl * ms, kvd);
    val = GT(c, "kv.val", typNUMBER::BF16, nl * ms, kvd);
    return (key && val) ? KF_OK : KF_OUTOF_GPUMEMORY;
}
void* KVCache::Get(CTYPE type, int layer, int pos) const {
    floatX* base = ToX(type == KV_KEY ? key : val);
    return base + ((size_t)layer * max_seq_len + pos) * kv_dim;
}

// ------------------------------------------------------------------------------------------------ neurons
hGTensor LayerNormal::cuFlow(hGTensor inp, int) {
    // P_CHAT_1 && nHead == 0  ->  CU_rms_infer (T.cu:569-573)
    if (kf_rmsnorm(hFish->ctx, ToX(inp), ToX(w), ToX(out), 1, ldTH, rms_eps, nullptr) != KF_OK) return nullptr;
    return out;
}
int Relu::Forw(hGTensor out, hGTensor gate, hGTensor inp, int) {
    return kf_swiglu(hFish->ctx, ToX(gate), ToX(inp), ToX(out), inp->ne[0]) == KF_OK ? 0 : -1;
}
int SLP::Forw(floatX* rhs, const floatX* lhs, uint32_t epilogue, const floatX* residual) {
    kf_weight wd = w->desc();
    return kf_linear(hFish->ctx, &wd, lhs, rhs, b ? ToX(b) : nullptr, 1, 1.0f, 0.0f, epilogue, residual) == KF_OK ? 0 : -1;
}
int SLP::Forw(hGTensor rhs, hGTensor lhs, hGTensor toGelu, Relu*, int) {
    floatX* dst = toGelu ? ToX(toGelu) : ToX(rhs);  // rhs = to_gelu ? to_gelu : rhs (NeuronFuse.cu:332)
    return Forw(dst, ToX(lhs));
}

hGTensor ROPE::cuInfer(SelfAttention* hQKV, uint32_t, int pos, int) {
    Fish* f = hFish;
    floatX* q = ToX(hQKV->Q.out);
    floatX* k = reinterpret_cast<floatX*>(hQKV->hCache->Get(KVCache::KV_KEY, hQKV->layid - 1, pos));
    int rc = kf_qknorm_rope(f->ctx, q, k, hnQ ? ToX(hnQ->w) : nullptr, hnK ? ToX(hnK->w) : nullptr, f->rope_table, pos, nullptr, n_head, n_head_kv, head_dim,
                            hnQ ? hnQ->rms_eps : 1e-6f);
    return rc == KF_OK ? hQKV->Q.out : nullptr;
}

void SelfAttention::_devQKV(int) { /* K.out/V.out are resolved per launch from the cache base + pos (see cuInfer) */ }

hGTensor SelfAttention::cuInfer(hGTensor inpL, int) {
    Fish* f = hFish;
    kf_ctx* c = f->ctx;
    const int pos = f->tok_pos, L = layid - 1;
    floatX* key_cache = reinterpret_cast<floatX*>(hCache->Get(KVCache::KV_KEY, L, 0));
    floatX* val_cache = reinterpret_cast<floatX*>(hCache->Get(KVCache::KV_VAL, L, 0));
    f->gBUFF.residual = inpL;
    const int32_t* d_pos = f->graph_mode ? f->d_state + 1 : nullptr;
    const int bound = f->graph_mode ? f->pos_bound() : pos;
    if (f->fuse_level == 0) {
        if (f->graph_mode) return nullptr;  // the per-kernel path resolves cache rows on the host
        hGTensor inpQ = norm.cuFlow(inpL);
        if (!inpQ) return nullptr;
        floatX* krow = key_cache + (size_t)pos * kv_dim;
        floatX* vrow = val_cache + (size_t)pos * kv_dim;
        if (Q.Forw(ToX(Q.out), ToX(inpQ)) || K.Forw(krow, ToX(inpQ)) || V.Forw(vrow, ToX(inpQ))) return nullptr;
        if (!rope.cuInfer(this, 42, pos)) return nullptr;
        if (kf_attn_decode(c, ToX(Q.out), key_cache, val_cache, ToX(Q.out), pos, nullptr, n_head, n_head_kv, head_dim, kv_dim, f->gBUFF.attn_ws->data) != KF_OK)
            return nullptr;
        if (proj_cat.Forw(ToX(f->gBUFF.scratch), ToX(Q.out))) return nullptr;
        if (kf_add(c, ToX(f->gBUFF.residual), ToX(f->gBUFF.scratch), ToX(out), f->config.nEmbed) != KF_OK) return nullptr;  // CU_add3
        return out;
    }
    // fused: [norm + Q,K,V] -> [q/k-norm + RoPE + attention] -> [proj_cat + residual]
    kf_weight wq = Q.w->desc(), wk = K.w->desc(), wv = V.w->desc(), wo = proj_cat.w->desc();
    const kf_weight* ws[3] = {&wq, &wk, &wv};
    kf_bf16* ys[3] = {ToX(Q.out), ToX(f->gBUFF.kraw), val_cache};
    const int64_t strides[3] = {0, 0, (int64_t)kv_dim};
    if (kf_norm_linear(c, ToX(inpL), ToX(norm.w), norm.rms_eps, 3, ws, ys, strides, bound, d_pos) != KF_OK) return nullptr;
    if (kf_attn_block(c, ToX(Q.out), ToX(f->gBUFF.kraw), key_cache, val_cache, ToX(f->gBUFF.scratch), normQ.w ? ToX(normQ.w) : nullptr,
                      normK.w ? ToX(normK.w) : nullptr, f->rope_table, bound, d_pos, n_head, n_head_kv, head_dim, kv_dim, normQ.rms_eps,
                      f->gBUFF.attn_ws->data) != KF_OK)
        return nullptr;
    if (kf_linear(c, &wo, ToX(f->gBUFF.scratch), ToX(out), nullptr, 1, 1.0f, 0.0f, KF_EPI_RESIDUAL, ToX(inpL)) != KF_OK) return nullptr;
    return out;
}

hGTensor FFN::cuInfer(hGTensor hIn, int) {
    Fish* f = hFish;
    kf_ctx* c = f->ctx;
    f->gBUFF.residual = hIn;
    if (f->fuse_level == 0) {
        hGTensor xn = norm.cuFlow(hIn);
        if (!xn) return nullptr;
        hGTensor tGelu = f->gBUFF.scratch, up_out = f->gBUFF.upOut;
        if (n_hot >= 0) {  // D_matmul_sparse on both projections
            kf_weight wg = gate.w->desc(), wu = up.w->desc();
            const int32_t* rows = reinterpret_cast<const int32_t*>(hot_rows->data);
            if (kf_linear_masked(c, &wg, ToX(xn), ToX(tGelu), nullptr, rows, n_hot) != KF_OK || kf_linear_masked(c, &wu, ToX(xn), ToX(up_out), nullptr, rows, n_hot) != KF_OK)
                return nullptr;
        } else if (gate.Forw(tGelu, xn) || up.Forw(up_out, xn))
            return nullptr;
        if (relu.Forw(tGelu, tGelu, up_out)) return nullptr;
        if (down.Forw(ToX(f->gBUFF.delta), ToX(tGelu))) return nullptr;
        if (kf_add(c, ToX(f->gBUFF.residual), ToX(f->gBUFF.delta), ToX(out), f->config.nEmbed) != KF_OK) return nullptr;
        return out;
    }
    kf_weight wg = gate.w->desc(), wu = up.w->desc(), wd = down.w->desc();
    if (n_hot >= 0) {
        if (kf_norm_gateup_swiglu_masked(c, ToX(hIn), ToX(norm.w), norm.rms_eps, &wg, &wu, ToX(f->gBUFF.scratch), reinterpret_cast<const int32_t*>(hot_rows->data),
                                         n_hot) != KF_OK)
            return nullptr;
    } else if (kf_norm_gateup_swiglu(c, ToX(hIn), ToX(norm.w), norm.rms_eps, &wg, &wu, ToX(f->gBUFF.scratch)) != KF_OK)
        return nullptr;
    if (kf_linear(c, &wd, ToX(f->gBUFF.scratch), ToX(out), nullptr, 1, 1.0f, 0.0f, KF_EPI_RESIDUAL, ToX(hIn)) != KF_OK) return nullptr;
    return out;
}

// ---- token batches (prefill)
int SelfAttention::cuFlow(floatX* bx, int pos0, int n) {
    Fish* f = hFish;
    kf_ctx* c = f->ctx;
    const int L = layid - 1, C = f->config.nEmbed;
    floatX* key_cache = reinterpret_cast<floatX*>(hCache->Get(KVCache::KV_KEY, L, 0));
    floatX* val_cache = reinterpret_cast<floatX*>(hCache->Get(KVCache::KV_VAL, L, 0));
    floatX* krows = key_cache + (size_t)pos0 * kv_dim;
    floatX* vrows = val_cache + (size_t)pos0 * kv_dim;
    floatX *bn = ToX(f->gBUFF.bNorm), *bq = ToX(f->gBUFF.bQ), *ba = ToX(f->gBUFF.bAttn);
    kf_weight wq = Q.w->desc(), wk = K.w->desc(), wv = V.w->desc(), wo = proj_cat.w->desc();
    KF_TRY(kf_rmsnorm(c, bx, ToX(norm.w), bn, n, C, norm.rms_eps, nullptr));
    const kf_weight* ws[3] = {&wq, &wk, &wv};
    kf_bf16* ys[3] = {bq, krows, vrows};  // K.out / V.out alias the cache rows (_devQKV)
    (void)ws, (void)ys;
    KF_TRY(kf_qkv_rope_batch(c, &wq, &wk, &wv, bn, bq, krows, vrows, n, normQ.w ? ToX(normQ.w) : nullptr, normK.w ? ToX(normK.w) : nullptr, f->rope_table, pos0, n_head, n_head_kv,
                             head_dim, normQ.rms_eps));
    KF_TRY(kf_attn_prefill(c, bq, key_cache, val_cache, ba, pos0, n, q_dim, n_head, n_head_kv, head_dim, kv_dim));
    return kf_linear(c, &wo, ba, bx, nullptr, n, 1.0f, 0.0f, KF_EPI_RESIDUAL, bx);
}

int FFN::cuFlow(floatX* bx, int n) {
    Fish* f = hFish;
    kf_ctx* c = f->ctx;
    const int C = f->config.nEmbed;
    floatX *bn = ToX(f->gBUFF.bNorm), *bg = ToX(f->gBUFF.bGate), *bu = ToX(f->gBUFF.bUp);
    kf_weight wg = gate.w->desc(), wu = up.w->desc(), wd = down.w->desc();
    KF_TRY(kf_rmsnorm(c, bx, ToX(norm.w), bn, n, C, norm.rms_eps, nullptr));
    KF_TRY(kf_gateup_swiglu_batch(c, &wg, &wu, bn, bg, bu, n));
    return kf_linear(c, &wd, bg, bx, nullptr, n, 1.0f, 0.0f, KF_EPI_RESIDUAL, bx);
}

hGTensor TokenEmbed::cuInfer(int token, int) {
    Fish* f = hFish;
    kf_weight wd = w->desc();
    int rc = (f->graph_mode || f->state_tokens) ? kf_embed_state(f->ctx, &wd, f->d_state, f->d_forced, ToX(out)) : kf_embed(f->ctx, &wd, token, nullptr, ToX(out));
    return rc == KF_OK ? out : nullptr;
}

hGTensor Head4Token::cuInfer_1(hGTensor inp_, int) {
    Fish* f = hFish;
    kf_weight wd = proj.w->desc();
    int rc;
    if (f->fuse_level == 0 && !f->state_tokens) {
        hGTensor xn = f->final_norm.cuFlow(inp_);
        if (!xn) return nullptr;
        rc = kf_lm_head(f->ctx, &wd, ToX(xn), ToX(preLogits), f->d_state + 2, f->gBUFF.head_ws->data);
    } else {
        rc = f->HeadAndPick(ToX(inp_));
    }
    return rc == KF_OK ? preLogits : nullptr;
}

// ------------------------------------------------------------------------------------------------ Fish
Fish::~Fish() {
    for (auto g : graphs) kf_graph_destroy(g);
    for (auto g : tp.group_graphs)
        if (g) kf_graph_destroy(g);
    if (ctx) {
        for (void* p : tp.opened) kf_tp_ipc_close(ctx, p);
        if (tp.area) kf_free(ctx, tp.area);
    }
    if (ctx && lin_scratch) kf_free(ctx, lin_scratch);
    if (ctx && deq_arena) kf_free(ctx, deq_arena);
    if (engine) kf_engine_destroy(engine);
    if (ctx && engine_ws) kf_free(ctx, engine_ws);
    if (ctx) {
        if (rope_table) kf_free(ctx, rope_table);
        if (d_state) kf_free(ctx, d_state);
        if (d_forced) kf_free(ctx, d_forced);
        if (d_tokens_out) kf_free(ctx, d_tokens_out);
        if (gBUFF.d_ptok) kf_free(ctx, gBUFF.d_ptok);
        if (d_rng) kf_free(ctx, d_rng);
    }
    attn.clear(), ffn.clear();
    embed = TokenEmbed(), head = Head4Token(), final_norm = LayerNormal();
    gBUFF = MemBuffer(), cache = KVCache(), x.reset();
    if (ctx) kf_destroy(ctx);
}

bool Fish::ShapeServed(const MODEL_CARD& c, std::string& why) {
    const int gq = c.n_head_kv > 0 ? c.n_head / c.n_head_kv : 0;
    if (c.n_head_kv <= 0 || c.n_head % c.n_head_kv != 0 || !(gq == 1 || gq == 2 || gq == 4 || gq == 8)) {
        why = "query heads per kv head = " + std::to_string(c.n_head) + " / " + std::to_string(c.n_head_kv) + " (served: 1, 2, 4, 8)";
        return false;
    }
    if (c.head_dim != 64 && c.head_dim != 128) {
        why = "head_dim " + std::to_string(c.head_dim) + " (served: 64, 128)";
        return false;
    }
    if (c.nEmbed % 8 != 0 || c.n_ff % 8 != 0) {
        why = "hidden / intermediate size must be a multiple of 8";
        return false;
    }
    return true;
}
static thread_local std::string g_host_err;
int Fish::Build(const MODEL_CARD& card, int device, void* stream) {
    if (!ShapeServed(card, g_host_err)) {
        g_host_err = "Fish::Build: " + g_host_err;
        return KF_UNSUPPORTED_DATATYPE;
    }
    config = card;
    KF_TRY(kf_init(device, stream, &ctx));
    const int C = card.nEmbed, hd = card.head_dim, qd = card.n_head * hd, kvd = card.n_head_kv * hd;
    KF_TRY(cache.Init(ctx, card.nLayer, card.n_ctx, kvd));
    x = GT(ctx, "x", typNUMBER::BF16, C);
    gBUFF.tmpFF1 = GT(ctx, "tmpFF1", typNUMBER::BF16, qd);
    gBUFF.kraw = GT(ctx, "kraw", typNUMBER::BF16, kvd);
    gBUFF.scratch = GT(ctx, "scratch", typNUMBER::BF16, std::max(std::max(qd, C), card.n_ff));
    gBUFF.delta = GT(ctx, "delta", typNUMBER::BF16, C);
    gBUFF.upOut = GT(ctx, "upOut", typNUMBER::BF16, card.n_ff);
    gBUFF.normed = GT(ctx, "normed", typNUMBER::BF16, C);
    gBUFF.attn_ws = GT(ctx, "attn_ws", typNUMBER::F32, (int)(kf_attn_scratch_bytes(card.n_head, hd) / 4));
    gBUFF.head_ws = GT(ctx, "head_ws", typNUMBER::F32, (int)(kf_head_scratch_bytes() / 4));
    if (!x || !gBUFF.tmpFF1 || !gBUFF.kraw || !gBUFF.scratch || !gBUFF.delta || !gBUFF.upOut || !gBUFF.normed || !gBUFF.attn_ws || !gBUFF.head_ws)
        return KF_OUTOF_GPUMEMORY;
    // RoPE (cos,sin) table on the host libm, uploaded once
    {
        std::vector<float> tab((size_t)card.n_ctx * hd);
        KF_TRY(kf_rope_table_host(tab.data(), card.n_ctx, hd, card.rope_theta));
        KF_TRY(kf_malloc(ctx, tab.size() * 4, (void**)&rope_table));
        KF_TRY(kf_h2d(ctx, rope_table, tab.data(), tab.size() * 4));
    }
    KF_TRY(kf_malloc(ctx, 16, (void**)&d_state));
    KF_TRY(kf_memset(ctx, d_state, 0, 16));
    KF_TRY(kf_malloc(ctx, (size_t)card.n_ctx * 4, (void**)&d_forced));
    KF_TRY(kf_memset(ctx, d_forced, 0xff, (size_t)card.n_ctx * 4));
    KF_TRY(kf_malloc(ctx, (size_t)card.n_ctx * 4, (void**)&d_tokens_out));
    KF_TRY(kf_memset(ctx, d_tokens_out, 0xff, (size_t)card.n_ctx * 4));

    embed.hFish = this, embed.name = "embed_tokens", embed.out = x;
    for (int l = 0; l < card.nLayer; l++) {
        auto a = std::make_unique<SelfAttention>();
        a->hFish = this, a->layid = l + 1, a->name = "layers." + std::to_string(l) + ".self_attn";
        a->n_head = card.n_head, a->n_head_kv = card.n_head_kv, a->head_dim = hd, a->q_dim = qd, a->kv_dim = kvd;
        a->hCache = &cache, a->out = x;
        a->norm.hFish = this, a->norm.ldTH = C, a->norm.rms_eps = card.rms_eps, a->norm.out = gBUFF.normed;
        a->normQ.hFish = a->normK.hFish = this, a->normQ.nHead = card.n_head, a->normK.nHead = card.n_head_kv;
        a->normQ.ldTH = a->normK.ldTH = hd, a->normQ.rms_eps = a->normK.rms_eps = card.qk_eps;
        for (SLP* s : {&a->Q, &a->K, &a->V, &a->proj_cat}) s->hFish = this;
        a->Q.out = gBUFF.tmpFF1, a->Q.nIn = C, a->Q.nOut = qd;
        a->rope.hFish = this, a->rope.hnQ = &a->normQ, a->rope.hnK = &a->normK;
        a->rope.n_head = card.n_head, a->rope.n_head_kv = card.n_head_kv, a->rope.head_dim = hd, a->rope.theta = card.rope_theta;
        attn.push_back(std::move(a));
        auto m = std::make_unique<FFN>();
        m->hFish = this, m->layid = l + 1, m->name = "layers." + std::to_string(l) + ".mlp", m->latent = card.n_ff, m->out = x;
        m->norm.hFish = this, m->norm.ldTH = C, m->norm.rms_eps = card.rms_eps, m->norm.out = gBUFF.normed;
        for (SLP* s : {&m->gate, &m->up, &m->down}) s->hFish = this;
        m->relu.hFish = this;
        ffn.push_back(std::move(m));
    }
    final_norm.hFish = this, final_norm.ldTH = C, final_norm.rms_eps = card.rms_eps, final_norm.out = gBUFF.normed;
    head.hFish = this, head.proj.hFish = this;
    head.preLogits = GT(ctx, "preLogits", typNUMBER::BF16, card.vocab);
    return head.preLogits ? KF_OK : KF_OUTOF_GPUMEMORY;
}

// position buckets: one captured graph each; the bound fixes the attention slice count of that graph
static const int kBuckets[] = {64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072};
static int bucket_of(int pos) {
    int i = 0;
    while (kBuckets[i] <= pos) i++;
    return i;
}
int Fish::pos_bound() const {
    int b = kBuckets[bucket_of(tok_pos)] - 1;
    return b < config.n_ctx - 1 ? b : config.n_ctx - 1;
}

int Fish::EnsureLinearScratch(const kf_weight& w, int nTok) {
    const size_t need = kf_linear_scratch_bytes(&w, nTok);
    if (need <= lin_scratch_bytes) return KF_OK;
    KF_TRY(kf_sync(ctx));
    void* p = nullptr;
    KF_TRY(kf_malloc(ctx, need, &p));
    KF_TRY(kf_set_scratch(ctx, p, need));
    if (lin_scratch) kf_free(ctx, lin_scratch);
    lin_scratch = p, lin_scratch_bytes = need;
    return KF_OK;
}

// The persistent decode engine over this model's layers: built on first use (every weight must be set), not while capturing.
int Fish::EnsureEngine() {
    if (engine_state != 0) return engine_state > 0 ? KF_OK : KF_ENGINE_NOT_SERVED;
    engine_state = -1;
    std::vector<kf_engine_layer> L(config.nLayer);
    for (int l = 0; l < config.nLayer; l++) {
        SelfAttention* a = attn[l].get();
        FFN* m = ffn[l].get();
        SLP* s[7] = {&a->Q, &a->K, &a->V, &a->proj_cat, &m->gate, &m->up, &m->down};
        for (int j = 0; j < 7; j++) {
            if (!s[j]->w || s[j]->b) {
                engine_why = "a layer matrix is missing or carries a bias";
                return KF_ENGINE_NOT_SERVED;
            }
            L[l].w[j] = s[j]->w->desc();
        }
        if (!a->norm.w || !m->norm.w) {
            engine_why = "a norm weight is missing";
            return KF_ENGINE_NOT_SERVED;
        }
        L[l].hot_ffn = m->n_hot >= 0 ? reinterpret_cast<const int32_t*>(m->hot_mask->data) : nullptr; /* the sparse forward inside the launch: cold gate / up rows never read, zeros published */
        L[l].norm_in = ToX(a->norm.w), L[l].norm_post = ToX(m->norm.w);
        L[l].q_norm = a->normQ.w ? ToX(a->normQ.w) : nullptr, L[l].k_norm = a->normK.w ? ToX(a->normK.w) : nullptr;
        L[l].kcache = reinterpret_cast<floatX*>(cache.Get(KVCache::KV_KEY, l, 0));
        L[l].vcache = reinterpret_cast<floatX*>(cache.Get(KVCache::KV_VAL, l, 0));
    }
    kf_engine_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_layer = config.nLayer, d.dim = config.nEmbed, d.n_head = config.n_head, d.n_kv = config.n_head_kv, d.head_dim = config.head_dim, d.ffn = config.n_ff;
    d.kv_stride = config.n_head_kv * config.head_dim;
    d.max_seq = config.n_ctx;
    d.rms_eps = config.rms_eps, d.qk_eps = config.qk_eps, d.rope_table = rope_table, d.layers = L.data();
    {
        char why[320];
        why[0] = 0;
        const int served = kf_engine_served(ctx, &d, why, sizeof(why));
        engine_why = why;
        if (served != KF_OK) return KF_ENGINE_NOT_SERVED;
    }
    const size_t bytes = kf_engine_workspace_bytes(&d);
    if (kf_malloc(ctx, bytes, &engine_ws) != KF_OK) return KF_OUTOF_GPUMEMORY;
    if (kf_engine_create(ctx, &d, engine_ws, bytes, &engine) != KF_OK) {
        kf_free(ctx, engine_ws);
        engine_ws = nullptr, engine = nullptr;
        return KF_ENGINE_NOT_SERVED;
    }
    engine_state = 1;
    // a bf16 embedding table is read inside the launch (graph-mode steps): one launch less per token; other storages keep kf_embed_state
    kf_weight we = embed.w->desc();
    engine_embed = kf_engine_set_embedding(ctx, engine, &we, d_forced) == KF_OK;
    // final norm + bf16 LM head + greedy pick as trailing phases of the same launch (other head storages keep kf_norm_lm_head)
    engine_head = false;
    if (head.proj.w && final_norm.w && head.preLogits && final_norm.rms_eps == config.rms_eps) {
        kf_weight wh = head.proj.w->desc();
        engine_head = kf_engine_set_head(ctx, engine, &wh, ToX(final_norm.w), ToX(head.preLogits), d_tokens_out) == KF_OK;
    }
    return KF_OK;
}
void Fish::DropEngineTable() {
    for (auto& g : graphs)
        if (g) kf_graph_destroy(g), g = nullptr;
    if (engine) {
        kf_sync(ctx);
        kf_engine_destroy(engine);
        engine = nullptr;
    }
    if (engine_ws) kf_free(ctx, engine_ws), engine_ws = nullptr;
    engine_state = 0, engine_embed = engine_head = false;
}
void Fish::DropEngine() {
    weights_gen++; /* what XcdReplicas / XcdTP objects built on this Fish compare their own copy with */
    DropEngineTable();
    bucket_tuned.clear(); /* the measured delays belonged to that engine */
    DropResident(); /* the resident bf16 copies are keyed by the old tensors' addresses too */
}
void Fish::DropResident() {
    if (deq_arena) {
        kf_sync(ctx);
        kf_set_dequant_arena(ctx, nullptr, 0);
        kf_free(ctx, deq_arena), deq_arena = nullptr, deq_arena_bytes = 0;
    }
    resident_tried = false;
}
// Resident bf16 copies of the layers' quantised matrices for long prompts (kf_set_dequant_arena): 2 bytes per weight -- 0.9 GB for Qwen3-0.6B, 62 GB for Qwen3-32B -- of a
// 288 GB part, instead of five dequantise launches per layer and prompt.  Sized and allocated by the first Prefill after the weights were set (never inside a launch
// sequence); skipped when the copies would not fit `resident_max_bytes` or the allocation fails (the per-call scratch route stays).
int Fish::EnsureResident(int PC) {
    if (deq_arena || resident_tried || !prefill_resident || PC < 1024) return KF_OK;
    resident_tried = true;
    size_t total = 0;
    auto add = [&](const SLP& s) {
        if (!s.w) return;
        const kf_weight d = s.w->desc();
        if (d.type != KF_BF16 && d.quant == KF_QUANT_GROUP && !d.qzeros) total += ((size_t)d.ne0 * d.ne1 * 2 + 255) & ~(size_t)255;
    };
    for (int l = 0; l < config.nLayer; l++) add(attn[l]->Q), add(attn[l]->K), add(attn[l]->V), add(attn[l]->proj_cat), add(ffn[l]->gate), add(ffn[l]->up), add(ffn[l]->down);
    if (total == 0 || total > resident_max_bytes) return KF_OK;
    KF_TRY(kf_sync(ctx));
    void* p = nullptr;
    if (kf_malloc(ctx, total, &p) != KF_OK) return KF_OK; /* no room: not an error, the scratch route serves */
    deq_arena = p, deq_arena_bytes = total;
    KF_TRY(kf_set_dequant_arena(ctx, p, total));
    const size_t need = kf_resident_scratch_bytes(); /* the scratch holds no dequantised copy any more: it lends the small tile kernels their split-K slots */
    if (need > lin_scratch_bytes) {
        void* q = nullptr;
        KF_TRY(kf_malloc(ctx, need, &q));
        KF_TRY(kf_set_scratch(ctx, q, need));
        if (lin_scratch) kf_free(ctx, lin_scratch);
        lin_scratch = q, lin_scratch_bytes = need;
    }
    return KF_OK;
}
int Fish::EngineCheck() {
    if (!engine || engine_steps == 0) return KF_OK;
    const int rc = kf_engine_check(ctx, engine);
    if (rc == KF_INTERNAL_ERR) { /* the word latches: every launch since the failure was a no-op.  Report it once, start over from a clean hand-off state */
        for (auto& g : graphs)
            if (g) kf_graph_destroy(g), g = nullptr;
        kf_engine_reset(ctx, engine);
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------ tensor parallel
int Fish::TPInit(int rank, int world, int vocab_row0) {
    if (world < 2 || world > 8 || rank < 0 || rank >= world) return KF_INVALID_ARGS;
    std::memset(&tp.comm, 0, sizeof(tp.comm));
    tp.rank = rank, tp.world = world, tp.vocab_row0 = vocab_row0;
    tp.comm.rank = rank, tp.comm.world = world, tp.comm.n_max = config.nEmbed, tp.comm.per_step = 2u * (uint32_t)config.nLayer + 1u;
    const size_t bytes = kf_tp_recv_bytes(world, config.nEmbed);
    const size_t pbytes = kf_tp_push_bytes(tp.comm.per_step);
    KF_TRY(kf_tp_alloc(ctx, bytes + 64 + pbytes, &tp.area)); /* + the generation and error words and the push descriptors behind the area */
    tp.comm.d_push = reinterpret_cast<uint8_t*>(tp.area) + bytes + 64;
    tp.comm.recv = tp.area, tp.comm.peer[rank] = tp.area;
    tp.comm.d_step = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(tp.area) + bytes);
    tp.comm.d_err = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(tp.area) + bytes + 32);
    use_engine = false; /* the persistent engine serves whole layers of one GPU */
    return KF_OK;
}
int Fish::TPSetPeer(int r, void* area) {
    if (tp.world < 2 || r < 0 || r >= tp.world || !area) return KF_INVALID_ARGS;
    tp.comm.peer[r] = area;
    tp.committed = false;
    return KF_OK;
}
int Fish::TPCommit() { /* the push descriptors are written once, when every peer's area is known (never inside a capture) */
    if (tp.committed || tp.world < 2) return KF_OK;
    for (int r = 0; r < tp.world; r++)
        if (!tp.comm.peer[r]) return KF_INVALID_ARGS;
    KF_TRY(kf_tp_commit(ctx, &tp.comm));
    tp.committed = true;
    return KF_OK;
}
int Fish::TPPhase(int phase, int l) {
    KF_TRY(TPCommit());
    const int32_t* d_pos = graph_mode ? d_state + 1 : nullptr;
    const int bound = graph_mode ? pos_bound() : tok_pos;
    const int C = config.nEmbed;
    switch (phase) {
        case 0: return embed.cuInfer(-1) ? KF_OK : KF_INTERNAL_ERR;
        case 1: {  // [norm + q,k,v rows of this rank] -> [q/k-norm + RoPE + attention over the local kv heads] -> o_proj column shard, pushed
            SelfAttention* a = attn[l].get();
            floatX* key_cache = reinterpret_cast<floatX*>(cache.Get(KVCache::KV_KEY, l, 0));
            floatX* val_cache = reinterpret_cast<floatX*>(cache.Get(KVCache::KV_VAL, l, 0));
            kf_weight wq = a->Q.w->desc(), wk = a->K.w->desc(), wv = a->V.w->desc(), wo = a->proj_cat.w->desc();
            const kf_weight* ws[3] = {&wq, &wk, &wv};
            kf_bf16* ys[3] = {ToX(a->Q.out), ToX(gBUFF.kraw), val_cache};
            const int64_t strides[3] = {0, 0, (int64_t)a->kv_dim};
            KF_TRY(kf_norm_linear(ctx, ToX(x), ToX(a->norm.w), a->norm.rms_eps, 3, ws, ys, strides, bound, d_pos));
            KF_TRY(kf_attn_block(ctx, ToX(a->Q.out), ToX(gBUFF.kraw), key_cache, val_cache, ToX(gBUFF.scratch), a->normQ.w ? ToX(a->normQ.w) : nullptr,
                                 a->normK.w ? ToX(a->normK.w) : nullptr, rope_table, bound, d_pos, a->n_head, a->n_head_kv, a->head_dim, a->kv_dim, a->normQ.rms_eps,
                                 gBUFF.attn_ws->data));
            return kf_linear_f32_push(ctx, &wo, ToX(gBUFF.scratch), &tp.comm, 2u * (uint32_t)l);
        }
        case 2: return kf_tp_reduce_recv(ctx, &tp.comm, 2u * (uint32_t)l, C, ToX(x), ToX(x));
        case 3: {
            FFN* m = ffn[l].get();
            kf_weight wg = m->gate.w->desc(), wu = m->up.w->desc(), wd = m->down.w->desc();
            KF_TRY(kf_norm_gateup_swiglu(ctx, ToX(x), ToX(m->norm.w), m->norm.rms_eps, &wg, &wu, ToX(gBUFF.scratch)));
            return kf_linear_f32_push(ctx, &wd, ToX(gBUFF.scratch), &tp.comm, 2u * (uint32_t)l + 1u);
        }
        case 4: return kf_tp_reduce_recv(ctx, &tp.comm, 2u * (uint32_t)l + 1u, C, ToX(x), ToX(x));
        case 5: {
            kf_weight wh = head.proj.w->desc();
            return kf_tp_lm_head(ctx, ToX(x), ToX(final_norm.w), final_norm.rms_eps, &wh, ToX(head.preLogits), tp.vocab_row0, &tp.comm, gBUFF.head_ws->data);
        }
        case 6: return kf_tp_pick(ctx, &tp.comm, d_state, d_tokens_out);
    }
    return KF_INVALID_ARGS;
}
int Fish::EnqueueStepTP() {
    KF_TRY(TPPhase(0, 0));
    for (int l = 0; l < config.nLayer; l++)
        for (int ph = 1; ph <= 4; ph++) KF_TRY(TPPhase(ph, l));
    KF_TRY(TPPhase(5, 0));
    return TPPhase(6, 0);
}
// ranks of one process on one stream: phase by phase across the ranks
static int tp_group_enqueue(Fish** fs, int R) {
    for (int r = 0; r < R; r++) KF_TRY(fs[r]->TPPhase(0, 0));
    for (int l = 0; l < fs[0]->config.nLayer; l++)
        for (int ph = 1; ph <= 4; ph++)
            for (int r = 0; r < R; r++) KF_TRY(fs[r]->TPPhase(ph, l));
    for (int r = 0; r < R; r++) KF_TRY(fs[r]->TPPhase(5, 0));
    for (int r = 0; r < R; r++) KF_TRY(fs[r]->TPPhase(6, 0));
    return KF_OK;
}

int Fish::EnqueueStep(int bound) {
    if (tp.world > 1) return EnqueueStepTP();
    if (use_engine && fuse_level >= 1 && engine_state > 0 && engine_embed && (graph_mode || state_tokens)) { /* embedding row read inside the launch */
        if (engine_head) { /* ... and the final norm, the LM head and the greedy pick: ONE launch per token */
            const int rc = kf_engine_step_head(ctx, engine, nullptr, ToX(x), d_state, bound, samp_params.greedy() ? 1 : 0);
            if (rc < 0) return rc;
            if (rc == KF_OK) {
                engine_steps++;
                if (samp_params.greedy()) return KF_OK;
                return (samp_params.true_topk ? kf_sample_topk : kf_sample)(ctx, ToX(head.preLogits), config.vocab, samp_params.top_k, samp_params.temperature, samp_params.top_p,
                                                                             d_rng, nullptr, d_state, d_tokens_out, d_forced, config.n_ctx);
            }
        } else {
            const int rc = kf_engine_step(ctx, engine, nullptr, ToX(x), d_state, bound);
            if (rc < 0) return rc;
            if (rc == KF_OK) {
                engine_steps++;
                return head.cuInfer_1(x) ? KF_OK : KF_INTERNAL_ERR;
            }
        }
    }
    hGTensor cur = embed.cuInfer(-1);
    if (!cur) return KF_INTERNAL_ERR;
    if (use_engine && fuse_level >= 1 && engine_state > 0 && !(engine_embed && (graph_mode || state_tokens))) {
        const int rc = kf_engine_step(ctx, engine, ToX(cur), ToX(x), d_state, bound);
        if (rc < 0) return rc;
        if (rc == KF_OK) {
            engine_steps++;
            return head.cuInfer_1(x) ? KF_OK : KF_INTERNAL_ERR;
        }
    }
    for (int l = 0; l < config.nLayer; l++) {
        cur = attn[l]->cuInfer(cur);
        if (!cur) return KF_INTERNAL_ERR;
        cur = ffn[l]->cuInfer(cur);
        if (!cur) return KF_INTERNAL_ERR;
    }
    return head.cuInfer_1(cur) ? KF_OK : KF_INTERNAL_ERR;
}

int Fish::ForwardOnRLS(int token, int pos) {
    if (pos < 0 || pos >= config.n_ctx || token < 0 || token >= config.vocab) return KF_INVALID_ARGS;
    tok_pos = pos;
    graph_mode = false;
    KF_TRY(kf_set_state(ctx, d_state, token, pos));
    hGTensor cur = embed.cuInfer(token);
    if (!cur) return KF_INTERNAL_ERR;
    for (int l = 0; l < config.nLayer; l++) {
        cur = attn[l]->cuInfer(cur);
        if (!cur) return KF_INTERNAL_ERR;
        cur = ffn[l]->cuInfer(cur);
        if (!cur) return KF_INTERNAL_ERR;
    }
    return head.cuInfer_1(cur) ? KF_OK : KF_INTERNAL_ERR;
}

int Fish::SetState(int token, int pos) { return kf_set_state(ctx, d_state, token, pos); }

kf_graph* Fish::GraphFor(int pos) {
    const int b = bucket_of(pos);
    if ((int)graphs.size() <= b) graphs.resize(b + 1, nullptr), graph_bound.resize(b + 1, 0);
    if (!graphs[b]) {
        if (use_engine && engine_state == 0) EnsureEngine(); /* not while capturing */
        tok_pos = pos;
        graph_mode = true;
        const int save = fuse_level;
        fuse_level = 1;
        if (kf_graph_begin(ctx) != KF_OK) return nullptr;
        int rc = EnqueueStep(pos_bound());
        kf_graph* g = nullptr;
        int rc2 = kf_graph_end(ctx, &g);
        fuse_level = save;
        graph_mode = false;
        if (rc != KF_OK || rc2 != KF_OK) return nullptr;
        graphs[b] = g, graph_bound[b] = pos_bound();
    }
    return graphs[b];
}

// A step that is ONE kernel launch (the engine with the embedding row, the LM head and the greedy pick inside) gains nothing from a captured graph: replaying a
// one-node graph costs ~6 us more per step than the launch itself (measured, scratch/graph_vs_eager.py: 0.4222 vs 0.4158 ms at positions 1900-2040).
bool Fish::OneLaunchStep() {
    if (tp.world > 1 || !use_engine || fuse_level == 0) return false;
    if (engine_state == 0) EnsureEngine();
    return engine_state > 0 && engine_embed && engine_head && samp_params.greedy();
}

int Fish::RunSteps(int pos, int n, bool use_graph) {
    if (pos < 0 || pos + n > config.n_ctx) return KF_INVALID_ARGS;
    KF_TRY(TPCommit());
    if (use_graph && n > 1 && OneLaunchStep()) { /* runs of steps inside one position bucket: ONE launch each (kf_engine_steps_head), at most kStepsPerLaunch steps */
        constexpr int kStepsPerLaunch = 16;
        int i = 0;
        while (i < n) {
            const int p = pos + i, b = bucket_of(p);
            int m = 1;
            while (i + m < n && m < kStepsPerLaunch && bucket_of(pos + i + m) == b) m++;
            tok_pos = p;
            if (engine_autotune > 0 && engine_embed) { /* once per position bucket: the hand-off delays measured at the first step inside it */
                if (bucket_tuned.empty()) bucket_tuned.assign((size_t)bucket_of(config.n_ctx - 1) + 1, 0);
                if (!bucket_tuned[b]) {
                    bucket_tuned[b] = 1;
                    const int trc = kf_engine_tune(ctx, engine, ToX(x), d_state, pos_bound(), engine_autotune, nullptr, nullptr);
                    if (trc < 0) { /* a tuning problem (a poll timed out while a delay was being tried) is not a decode error: the built-in delays stay (kf_engine_tune restored
                                      them), the latched error word is cleared, and the step below runs */
                        if (kf_engine_reset(ctx, engine) != KF_OK) return trc;
                        engine_autotune = 0;
                    }
                }
            }
            const int rc = kf_engine_steps_head(ctx, engine, ToX(x), d_state, pos_bound(), m);
            if (rc < 0) return rc;
            if (rc != KF_OK) break; /* not served at this position: the per-step path below takes the rest */
            engine_steps += m;
            i += m;
        }
        if (i == n) return KF_OK;
        pos += i, n -= i;
    }
    for (int i = 0; i < n; i++) {
        const int p = pos + i;
        if (fuse_level == 0) {  // per-kernel launches only (AutoAWQ weights): eager, position from the host, token from the device state
            tok_pos = p;
            graph_mode = false, state_tokens = true;
            int rc = EnqueueStep(pos_bound());
            state_tokens = false;
            KF_TRY(rc);
        } else if (use_graph && !OneLaunchStep()) {
            kf_graph* g = GraphFor(p);
            if (!g) return KF_INTERNAL_ERR;
            KF_TRY(kf_graph_launch(ctx, g));
        } else {
            if (use_engine && engine_state == 0) EnsureEngine();
            tok_pos = p;
            graph_mode = true;  // positions/tokens still come from d_state, launches are eager
            const int save = fuse_level;
            fuse_level = 1;
            int rc = EnqueueStep(pos_bound());
            fuse_level = save;
            graph_mode = false;
            KF_TRY(rc);
        }
    }
    return KF_OK;
}

int Fish::Prefill(const int* tokens, int n, int pos0) {
    if (n < 1 || pos0 < 0 || pos0 + n > config.n_ctx) return KF_INVALID_ARGS;
    for (int i = 0; i < n; i++)
        if (tokens[i] < 0 || tokens[i] >= config.vocab) return KF_INVALID_ARGS;
    const int PC = prefill_chunk < config.n_ctx ? prefill_chunk : config.n_ctx, C = config.nEmbed;
    KF_TRY(PrefillReady());
    floatX* bx = ToX(gBUFF.bX);
    kf_weight we = embed.w->desc();
    int m = 0;
    for (int c0 = 0; c0 < n; c0 += PC) {
        m = n - c0 < PC ? n - c0 : PC;
        KF_TRY(kf_h2d(ctx, gBUFF.d_ptok, tokens + c0, (size_t)m * 4));
        KF_TRY(kf_embed_batch(ctx, &we, gBUFF.d_ptok, m, bx));
        for (int l = 0; l < config.nLayer; l++) {
            KF_TRY(attn[l]->cuFlow(bx, pos0 + c0, m));
            KF_TRY(ffn[l]->cuFlow(bx, m));
        }
    }
    // head on the last token; the state update leaves {next token, pos0 + n}
    KF_TRY(kf_d2d(ctx, x->data, bx + (size_t)(m - 1) * C, (size_t)C * 2));
    KF_TRY(SetState(tokens[n - 1], pos0 + n - 1));
    tok_pos = pos0 + n - 1;
    return HeadAndPick(ToX(x));
}

// the token-batch buffers ([prefill_chunk rows]) and the tile kernels' scratch / resident copies: allocated by the first prefill, never inside a later one
int Fish::PrefillReady(int min_rows) {
    int PC = prefill_chunk < config.n_ctx ? prefill_chunk : config.n_ctx; /* one prompt never has more rows than the context; a batch of prompts (XcdReplicas::PrefillBatch) may */
    const int C = config.nEmbed, qd = config.n_head * config.head_dim;
    if (min_rows > prefill_chunk) return KF_INVALID_ARGS;
    PC = min_rows > PC ? min_rows : PC;
    if (gBUFF.bX && gBUFF.rows < PC) { /* a later, larger batch: the buffers grow once (never inside a captured region: prefill launches are eager) */
        KF_TRY(kf_sync(ctx));
        gBUFF.bX.reset(), gBUFF.bNorm.reset(), gBUFF.bQ.reset(), gBUFF.bAttn.reset(), gBUFF.bGate.reset(), gBUFF.bUp.reset();
        kf_free(ctx, gBUFF.d_ptok), gBUFF.d_ptok = nullptr;
        if (!deq_arena) resident_tried = false; /* the larger batch may be one the resident bf16 copies serve (>= 1024 rows) */
    }
    if (!gBUFF.bX) {
        gBUFF.rows = PC;
        gBUFF.bX = GT(ctx, "bX", typNUMBER::BF16, C, PC);
        gBUFF.bNorm = GT(ctx, "bNorm", typNUMBER::BF16, C, PC);
        gBUFF.bQ = GT(ctx, "bQ", typNUMBER::BF16, qd, PC);
        gBUFF.bAttn = GT(ctx, "bAttn", typNUMBER::BF16, qd, PC);
        gBUFF.bGate = GT(ctx, "bGate", typNUMBER::BF16, config.n_ff, PC);
        gBUFF.bUp = GT(ctx, "bUp", typNUMBER::BF16, config.n_ff, PC);
        if (!gBUFF.bX || !gBUFF.bNorm || !gBUFF.bQ || !gBUFF.bAttn || !gBUFF.bGate || !gBUFF.bUp) return KF_OUTOF_GPUMEMORY;
        KF_TRY(kf_malloc(ctx, (size_t)PC * 4, (void**)&gBUFF.d_ptok));
        // the stacked large-batch routes (Q | K | V and gate | up dequantised back to back, one tile-GEMM launch each): their workspace, sized here -- the first
        // Prefill allocates the batch buffers anyway; the launches themselves never allocate
        for (int l = 0; l < config.nLayer; l++) {
            kf_weight wq = attn[l]->Q.w->desc(), wk = attn[l]->K.w->desc(), wv = attn[l]->V.w->desc(), wg = ffn[l]->gate.w->desc(), wu = ffn[l]->up.w->desc();
            const kf_weight* qkv[3] = {&wq, &wk, &wv};
            const kf_weight* gu[2] = {&wg, &wu};
            size_t need = kf_linear_multi_scratch_bytes(3, qkv, PC), need2 = kf_linear_multi_scratch_bytes(2, gu, PC);
            need = need > need2 ? need : need2;
            if (need > lin_scratch_bytes) {
                KF_TRY(kf_sync(ctx));
                void* p = nullptr;
                KF_TRY(kf_malloc(ctx, need, &p));
                KF_TRY(kf_set_scratch(ctx, p, need));
                if (lin_scratch) kf_free(ctx, lin_scratch);
                lin_scratch = p, lin_scratch_bytes = need;
            }
        }
    }
    return EnsureResident(gBUFF.rows);
}

int Fish::HeadAndPick(const floatX* x_last) {
    kf_weight wh = head.proj.w->desc();
    if (samp_params.greedy())
        return kf_norm_lm_head(ctx, x_last, ToX(final_norm.w), final_norm.rms_eps, &wh, ToX(head.preLogits), d_state, d_tokens_out, gBUFF.head_ws->data);
    KF_TRY(kf_norm_lm_head(ctx, x_last, ToX(final_norm.w), final_norm.rms_eps, &wh, ToX(head.preLogits), nullptr, nullptr, gBUFF.head_ws->data));
    return (samp_params.true_topk ? kf_sample_topk : kf_sample)(ctx, ToX(head.preLogits), config.vocab, samp_params.top_k, samp_params.temperature, samp_params.top_p,
                                                                 d_rng, nullptr, d_state, d_tokens_out, d_forced, config.n_ctx);
}

int Fish::SetSampler(const CHAT_SAMPLER& s) {
    if (!s.greedy()) {
        const int k = s.top_k < config.vocab ? s.top_k : config.vocab;
        if (k < 2 || k >= config.vocab / 2 || k > 1024 || !(s.temperature > 0.0f) || !(s.top_p > 0.0f)) return KF_INVALID_ARGS;
    }
    const bool was_greedy = samp_params.greedy();
    samp_params = s;
    if (!was_greedy || !s.greedy()) { /* the captured step graphs end in a different pick, or hold the old sampler arguments */
        for (auto& g : graphs)
            if (g) kf_graph_destroy(g), g = nullptr;
    }
    if (!d_rng) KF_TRY(kf_malloc(ctx, 8, (void**)&d_rng));
    return kf_h2d(ctx, d_rng, &samp_params.seed, 8);
}

int Fish::Generate(const int* prompt, int n_prompt, int n_new, int* out, bool use_graph) {
    if (n_prompt < 1 || n_new < 1 || n_prompt + n_new - 1 > config.n_ctx) return KF_INVALID_ARGS;
    std::vector<int32_t> forced(config.n_ctx, -1);
    for (int i = 0; i < n_prompt; i++) forced[i] = prompt[i];
    KF_TRY(kf_h2d(ctx, d_forced, forced.data(), forced.size() * 4));
    const int total = n_prompt + n_new - 1;
    if (prefill_mode == 1 && n_prompt > 1) {
        KF_TRY(Prefill(prompt, n_prompt, 0));
        if (n_new > 1) KF_TRY(RunSteps(n_prompt, n_new - 1, use_graph));
    } else {
        KF_TRY(SetState(prompt[0], 0));
        KF_TRY(RunSteps(0, total, use_graph));
    }
    std::vector<int32_t> toks(config.n_ctx);
    KF_TRY(kf_d2h(ctx, toks.data(), d_tokens_out, toks.size() * 4));
    KF_TRY(EngineCheck()); /* a launch that could not become resident leaves an error word, never a hang */
    for (int i = 0; i < n_new; i++) out[i] = toks[n_prompt - 1 + i];
    return KF_OK;
}

// ------------------------------------------------------------------------------------------------ XCD-confined replicas
XcdReplicas::~XcdReplicas() {
    if (!hFish) return;
    kf_ctx* ctx = hFish->ctx;
    if (engine) {
        kf_sync(ctx);
        kf_xengine_destroy(engine);
    }
    if (engine_ws) kf_free(ctx, engine_ws);
    if (d_state) kf_free(ctx, d_state);
    if (d_forced) kf_free(ctx, d_forced);
    if (d_tokens_out) kf_free(ctx, d_tokens_out);
    if (d_rng) kf_free(ctx, d_rng);
    if (d_dst) kf_free(ctx, d_dst);
}
size_t XcdReplicas::kv_seq_elems() const {
    const MODEL_CARD& c = hFish->config;
    return (size_t)c.nLayer * c.n_ctx * c.n_head_kv * c.head_dim;
}
// the engine over the Fish's weights AS THEY ARE NOW (their device addresses go into the engine's layer table; the GQA-4 forms copy q | k | v): called by Build, and again by
// the first use after the Fish's weights changed (kfh_weights_changed / a weight set again: Fish::weights_gen) -- a decode through this object never reads stale weights
int XcdReplicas::MakeEngine(bool allocate) {
    Fish* f = hFish;
    kf_ctx* ctx = f->ctx;
    const MODEL_CARD& c = f->config;
    const int kvd = c.n_head_kv * c.head_dim;
    std::vector<kf_engine_layer> L(c.nLayer);
    for (int l = 0; l < c.nLayer; l++) {
        SelfAttention* a = f->attn[l].get();
        FFN* m = f->ffn[l].get();
        SLP* s[7] = {&a->Q, &a->K, &a->V, &a->proj_cat, &m->gate, &m->up, &m->down};
        for (int j = 0; j < 7; j++) {
            if (!s[j]->w || s[j]->b) {
                why = "a layer matrix is missing or carries a bias";
                return KF_ENGINE_NOT_SERVED;
            }
            L[l].w[j] = s[j]->w->desc();
        }
        if (!a->norm.w || !m->norm.w) {
            why = "a norm weight is missing";
            return KF_ENGINE_NOT_SERVED;
        }
        L[l].hot_ffn = m->n_hot >= 0 ? reinterpret_cast<const int32_t*>(m->hot_mask->data) : nullptr; /* the sparse forward (round 6): cold gate / up rows publish zeros */
        L[l].norm_in = ToX(a->norm.w), L[l].norm_post = ToX(m->norm.w);
        L[l].q_norm = a->normQ.w ? ToX(a->normQ.w) : nullptr, L[l].k_norm = a->normK.w ? ToX(a->normK.w) : nullptr;
        L[l].kcache = L[l].vcache = reinterpret_cast<floatX*>(f->cache.Get(KVCache::KV_KEY, l, 0)); /* stand-ins for the validation call below; set after the allocation */
    }
    kf_engine_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_layer = c.nLayer, d.dim = c.nEmbed, d.n_head = c.n_head, d.n_kv = c.n_head_kv, d.head_dim = c.head_dim, d.ffn = c.n_ff;
    d.kv_stride = kvd, d.max_seq = c.n_ctx;
    d.rms_eps = c.rms_eps, d.qk_eps = c.qk_eps, d.rope_table = f->rope_table, d.layers = L.data();
    {
        char w[320];
        w[0] = 0;
        const int served = kf_xengine_served(ctx, &d, w, sizeof(w));
        why = w;
        if (served != KF_OK) return served < 0 ? served : KF_ENGINE_NOT_SERVED;
    }
    if (!f->embed.w || !f->head.proj.w || !f->final_norm.w || f->final_norm.rms_eps != c.rms_eps) {
        why = "embedding / head / final norm missing";
        return KF_ENGINE_NOT_SERVED;
    }
    const size_t seq_elems = kv_seq_elems();
    if (allocate) { // per sequence: K / V cache, state {token, pos, parked, status}, forced ids, ids out, logits, residual stream
        key = GT(ctx, "xr.key", typNUMBER::BF16, kvd, c.n_ctx * c.nLayer * n_seq);
        val = GT(ctx, "xr.val", typNUMBER::BF16, kvd, c.n_ctx * c.nLayer * n_seq);
        logits = GT(ctx, "xr.logits", typNUMBER::BF16, c.vocab, n_seq);
        x = GT(ctx, "xr.x", typNUMBER::BF16, c.nEmbed, n_seq);
        if (!key || !val || !logits || !x) return KF_OUTOF_GPUMEMORY;
        KF_TRY(kf_memset(ctx, key->data, 0, seq_elems * n_seq * 2));
        KF_TRY(kf_memset(ctx, val->data, 0, seq_elems * n_seq * 2));
        KF_TRY(kf_malloc(ctx, (size_t)n_seq * 16, (void**)&d_state));
        KF_TRY(kf_malloc(ctx, (size_t)n_seq * c.n_ctx * 4, (void**)&d_forced));
        KF_TRY(kf_malloc(ctx, (size_t)n_seq * c.n_ctx * 4, (void**)&d_tokens_out));
        KF_TRY(kf_memset(ctx, d_state, 0, (size_t)n_seq * 16));
        KF_TRY(kf_memset(ctx, d_forced, 0xff, (size_t)n_seq * c.n_ctx * 4));
        KF_TRY(kf_memset(ctx, d_tokens_out, 0, (size_t)n_seq * c.n_ctx * 4));
    }
    for (int l = 0; l < c.nLayer; l++) {
        L[l].kcache = ToX(key) + (size_t)l * c.n_ctx * kvd;
        L[l].vcache = ToX(val) + (size_t)l * c.n_ctx * kvd;
    }
    const size_t bytes = kf_xengine_workspace_bytes(&d);
    if (!engine_ws || bytes > engine_ws_bytes) {
        if (engine_ws) kf_free(ctx, engine_ws), engine_ws = nullptr;
        KF_TRY(kf_malloc(ctx, bytes, &engine_ws));
        engine_ws_bytes = bytes;
    }
    int rc = kf_xengine_create(ctx, &d, n_seq, (int64_t)seq_elems, engine_ws, engine_ws_bytes, &engine);
    if (rc != KF_OK) {
        why = kf_last_error();
        return rc;
    }
    kf_weight we = f->embed.w->desc(), wh = f->head.proj.w->desc();
    rc = kf_xengine_set_embedding(ctx, engine, &we, d_forced, c.n_ctx);
    if (rc == KF_OK) rc = kf_xengine_set_head(ctx, engine, &wh, ToX(f->final_norm.w), ToX(logits), d_tokens_out, c.n_ctx);
    if (rc != KF_OK) {
        why = "the embedding table and the LM head must be bf16 (read inside the launch)";
        return rc == KF_UNSUPPORTED_DATATYPE ? KF_ENGINE_NOT_SERVED : rc;
    }
    built_gen = f->weights_gen;
    return KF_OK;
}
int XcdReplicas::Build(Fish* f, int n_seq_) {
    if (!f || n_seq_ < 1 || n_seq_ > KF_XENGINE_MAX_SEQ) return KF_INVALID_ARGS;
    hFish = f, n_seq = n_seq_;
    return MakeEngine(true);
}
int XcdReplicas::Fresh() {
    if (engine && built_gen == hFish->weights_gen) return KF_OK;
    kf_ctx* ctx = hFish->ctx;
    if (engine) {
        KF_TRY(kf_sync(ctx));
        kf_xengine_destroy(engine), engine = nullptr;
    }
    return MakeEngine(false); /* caches, states, forced ids and ids out stay: the sequences go on where they stand, on the new weights */
}
int XcdReplicas::SetForced(int seq, const int32_t* ids, int n) {
    if (seq < 0 || seq >= n_seq || n < 0 || n > hFish->config.n_ctx) return KF_INVALID_ARGS;
    std::vector<int32_t> row(hFish->config.n_ctx, -1);
    for (int i = 0; i < n; i++) row[i] = ids[i];
    return kf_h2d(hFish->ctx, d_forced + (size_t)seq * hFish->config.n_ctx, row.data(), row.size() * 4);
}
int XcdReplicas::SetState(int seq, int token, int pos) {
    if (seq < 0 || seq >= n_seq || pos < 0 || pos >= hFish->config.n_ctx || token < 0 || token >= hFish->config.vocab) return KF_INVALID_ARGS;
    KF_TRY(kf_memset32(hFish->ctx, d_state + 4 * seq + 3, 0, 1)); /* the status word of the re-aimed sequence */
    return kf_set_state(hFish->ctx, d_state + 4 * seq, token, pos);
}
// CHAT_SAMPLER for Chat (greedy by default): one rng word per sequence, seeded per request
int XcdReplicas::SetSampler(const CHAT_SAMPLER& sp) {
    if (!sp.greedy()) {
        const int k = sp.top_k < hFish->config.vocab ? sp.top_k : hFish->config.vocab;
        if (k < 2 || k >= hFish->config.vocab / 2 || k > 1024 || !(sp.temperature > 0.0f) || !(sp.top_p > 0.0f)) return KF_INVALID_ARGS;
        if (!d_rng) KF_TRY(kf_malloc(hFish->ctx, (size_t)n_seq * 8, (void**)&d_rng));
    }
    samp_params = sp;
    return KF_OK;
}
// A parked sequence is skipped by the launches (the others decode on); its cache, state and ids stay.  Status: 0, or 64 = the last launch would have left the
// sequence's cache rows and skipped it (kf_abi.h: d_state [n_seq][4] = {token, pos, parked, status}).
int XcdReplicas::Park(int seq, bool on) {
    if (seq < 0 || seq >= n_seq) return KF_INVALID_ARGS;
    return kf_memset32(hFish->ctx, d_state + 4 * seq + 2, on ? 1 : 0, 1); /* on the stream, no host sync: the queue flips these between launches */
}
int XcdReplicas::Status(int seq, int32_t* out4) {
    if (seq < 0 || seq >= n_seq || !out4) return KF_INVALID_ARGS;
    return kf_d2h(hFish->ctx, out4, d_state + 4 * seq, 16);
}
// The prompt half of "prefill + decode" for one of the sequences: the model's own batched prefill (Fish::Prefill: the reference prefills token by token through the decode
// path, GoPT.cpp:1139-1146), then the prompt's K / V rows of every layer move into the sequence's cache and the sequence stands at {first generated id, n}.
int XcdReplicas::Prefill(int seq, const int* tokens, int n) {
    const MODEL_CARD& c = hFish->config;
    if (seq < 0 || seq >= n_seq || !tokens || n < 1 || n >= c.n_ctx) return KF_INVALID_ARGS;
    kf_ctx* ctx = hFish->ctx;
    KF_TRY(Fresh());
    {   /* the sequence's cache has the model's own layout ([layer][row][kv_dim]): for the length of the call the model's cache IS the sequence's, the rows land where they stay */
        struct Aim {
            KVCache& kc;
            void *k0, *v0;
            Aim(KVCache& c_, void* k, void* v) : kc(c_), k0(c_.key->data), v0(c_.val->data) { kc.key->data = k, kc.val->data = v; }
            ~Aim() { kc.key->data = k0, kc.val->data = v0; }
        } aim(hFish->cache, ToX(key) + (size_t)seq * kv_seq_elems(), ToX(val) + (size_t)seq * kv_seq_elems());
        KF_TRY(hFish->Prefill(tokens, n, 0));
    }
    KF_TRY(kf_d2d(ctx, ToX(logits) + (size_t)seq * c.vocab, ToX(hFish->head.preLogits), (size_t)c.vocab * 2));   /* the last prompt token's logits */
    KF_TRY(kf_d2d(ctx, d_state + 4 * seq, hFish->d_state, 8));                                                   /* {the id picked behind the prompt, n} */
    KF_TRY(kf_memset(ctx, d_state + 4 * seq + 3, 0, 4));                                                         /* status: clear */
    KF_TRY(kf_d2d(ctx, d_tokens_out + (size_t)seq * c.n_ctx + (n - 1), hFish->d_tokens_out + (n - 1), 4));      /* ids out: position n - 1 holds that id, as after decode steps */
    return KF_OK;
}
int XcdReplicas::PrefillBatch(const int* slots, const int32_t* tokens, const int* lens, int S, int stride) {
    Fish* f = hFish;
    const MODEL_CARD& c = f->config;
    kf_ctx* ctx = f->ctx;
    if (!slots || !tokens || !lens || S < 1 || S > n_seq || stride < 1) return KF_INVALID_ARGS;
    int T = 0;
    for (int i = 0; i < S; i++) {
        if (slots[i] < 0 || slots[i] >= n_seq || lens[i] < 1 || lens[i] > stride || lens[i] >= c.n_ctx) return KF_INVALID_ARGS;
        for (int j = 0; j < i; j++)
            if (slots[j] == slots[i]) return KF_INVALID_ARGS;
        for (int t = 0; t < lens[i]; t++)
            if (tokens[(size_t)i * stride + t] < 0 || tokens[(size_t)i * stride + t] >= c.vocab) return KF_INVALID_ARGS;
        T = lens[i] > T ? lens[i] : T;
    }
    /* rows of a prompt behind its end (padding up to the batch's longest): finite values that nobody reads before a decode step rewrites them */
    const int R = S * T, C = c.nEmbed;
    const int qd = c.n_head * c.head_dim, kvd = c.n_head_kv * c.head_dim;
    if (T >= c.n_ctx || R > f->prefill_chunk) return KF_INVALID_ARGS;
    KF_TRY(Fresh());
    KF_TRY(f->PrefillReady(R));
    if (!bK || bK->ne[1] < R) {
        KF_TRY(kf_sync(ctx));
        const int rows = f->gBUFF.rows;
        bK = GT(ctx, "xr.bK", typNUMBER::BF16, kvd, rows), bV = GT(ctx, "xr.bV", typNUMBER::BF16, kvd, rows);
        if (!bK || !bV) return KF_OUTOF_GPUMEMORY;
        if (!d_dst) KF_TRY(kf_malloc(ctx, (size_t)3 * n_seq * sizeof(void*) + (size_t)n_seq * 4, (void**)&d_dst)); /* three tables of destinations (K, V, logits) + the slots */
        if (!bL) bL = GT(ctx, "xr.bL", typNUMBER::BF16, c.vocab, n_seq);
        if (!bL) return KF_OUTOF_GPUMEMORY;
    }
    std::vector<int32_t> rows((size_t)R, 0);
    std::vector<void*> dst((size_t)3 * n_seq + (n_seq + 1) / 2, nullptr); /* the slots ride behind the pointer tables as int32 */
    int32_t* h_slots = reinterpret_cast<int32_t*>(dst.data() + (size_t)3 * n_seq);
    for (int i = 0; i < S; i++) {
        memcpy(rows.data() + (size_t)i * T, tokens + (size_t)i * stride, (size_t)lens[i] * 4);
        dst[i] = ToX(key) + (size_t)slots[i] * kv_seq_elems(), dst[n_seq + i] = ToX(val) + (size_t)slots[i] * kv_seq_elems();
        dst[2 * n_seq + i] = ToX(logits) + (size_t)slots[i] * c.vocab;
        h_slots[i] = slots[i];
    }
    const int32_t* d_slots = reinterpret_cast<const int32_t*>(d_dst + (size_t)3 * n_seq);
    KF_TRY(kf_h2d(ctx, f->gBUFF.d_ptok, rows.data(), (size_t)R * 4));
    KF_TRY(kf_h2d(ctx, d_dst, dst.data(), (size_t)3 * n_seq * sizeof(void*) + (size_t)n_seq * 4));
    floatX *bx = ToX(f->gBUFF.bX), *bn = ToX(f->gBUFF.bNorm), *bq = ToX(f->gBUFF.bQ), *ba = ToX(f->gBUFF.bAttn), *bk = ToX(bK), *bv = ToX(bV);
    kf_weight we = f->embed.w->desc();
    KF_TRY(kf_embed_batch(ctx, &we, f->gBUFF.d_ptok, R, bx));
    for (int l = 0; l < c.nLayer; l++) {
        SelfAttention& a = *f->attn[l];
        kf_weight wq = a.Q.w->desc(), wk = a.K.w->desc(), wv = a.V.w->desc(), wo = a.proj_cat.w->desc();
        KF_TRY(kf_rmsnorm(ctx, bx, ToX(a.norm.w), bn, R, C, a.norm.rms_eps, nullptr));
        KF_TRY(kf_qkv_rope_seqs(ctx, &wq, &wk, &wv, bn, bq, bk, bv, R, T, a.normQ.w ? ToX(a.normQ.w) : nullptr, a.normK.w ? ToX(a.normK.w) : nullptr, f->rope_table, 0, c.n_head,
                                c.n_head_kv, c.head_dim, a.normQ.rms_eps)); /* positions restart with every prompt */
        KF_TRY(kf_attn_prefill_batch(ctx, bq, bk, bv, ba, T, qd, c.n_head, c.n_head_kv, c.head_dim, kvd, S)); /* the rows of a prompt see that prompt's keys only */
        const size_t off = (size_t)l * c.n_ctx * kvd * 2, blk = (size_t)T * kvd * 2;
        KF_TRY(kf_copy_blocks(ctx, d_dst, off, bk, blk, blk, S));
        KF_TRY(kf_copy_blocks(ctx, d_dst + n_seq, off, bv, blk, blk, S));
        KF_TRY(kf_linear(ctx, &wo, ba, bx, nullptr, R, 1.0f, 0.0f, KF_EPI_RESIDUAL, bx));
        KF_TRY(f->ffn[l]->cuFlow(bx, R));
    }
    // the head on every prompt's last row: state {last prompt token, len - 1} -> {picked id, len}, ids out [len - 1] = that id, the row's logits into the slot's
    // (the head matrix -- 311 MB on Qwen3-0.6B -- is read ONCE for all prompts: the last rows normed side by side, one product, one pick launch, one scatter of the logits)
    kf_weight wh = f->head.proj.w->desc();
    const bool together = S > 1 && c.vocab % 8 == 0;
    for (int i = 0; i < S; i++) {
        const int s = slots[i], n = lens[i];
        KF_TRY(kf_set_state(ctx, d_state + 4 * s, tokens[(size_t)i * stride + n - 1], n - 1));
        KF_TRY(kf_memset(ctx, d_state + 4 * s + 3, 0, 4));
        if (together)
            KF_TRY(kf_rmsnorm(ctx, bx + ((size_t)i * T + n - 1) * C, ToX(f->final_norm.w), bn + (size_t)i * C, 1, C, f->final_norm.rms_eps, nullptr));
        else
            KF_TRY(kf_norm_lm_head(ctx, bx + ((size_t)i * T + n - 1) * C, ToX(f->final_norm.w), f->final_norm.rms_eps, &wh, ToX(logits) + (size_t)s * c.vocab, d_state + 4 * s,
                                   d_tokens_out + (size_t)s * c.n_ctx, f->gBUFF.head_ws->data));
    }
    if (together) {
        KF_TRY(kf_linear(ctx, &wh, bn, ToX(bL), nullptr, S, 1.0f, 0.0f, 0u, nullptr));
        KF_TRY(kf_argmax_rows_state(ctx, ToX(bL), c.vocab, c.vocab, S, d_slots, d_state, d_tokens_out, c.n_ctx));
        KF_TRY(kf_copy_blocks(ctx, d_dst + 2 * n_seq, 0, ToX(bL), (size_t)c.vocab * 2, (size_t)c.vocab * 2, S));
    }
    return KF_OK;
}
int XcdReplicas::RunSteps(int n) {
    if (n < 1) return KF_INVALID_ARGS;
    KF_TRY(Fresh());
    for (int i = 0; i < n;) {
        const int m = n - i < steps_per_launch ? n - i : steps_per_launch;
        KF_TRY(kf_xengine_steps(hFish->ctx, engine, ToX(x), d_state, m, 1));
        i += m, steps_run += m;
    }
    return KF_OK;
}
// A queue of prompts answered through the sequences' slots: what Fish::Chat does round by round over DEBUG.prompts (GoPT.cpp:1111-1180: prefill the prompt, then sample
// until the tokenizer's EOS or the context is full, then the next prompt), with n_seq rounds in flight at once.  A free slot takes the next prompt (Prefill), the launches
// decode every occupied slot, a finished round's slot is parked until the queue refills it; an answer ends at `eos` (eos < 0: never), at max_new ids, or at the last cache
// row.  out [n_req][max_new] (-1 behind an answer's end), out_len [n_req]; every answer equals Fish::Generate's on the same prompt (same prefill, same decode arithmetic).
// With eos >= 0 the ids of each launch are read back (one sync per launch) and a sequence may run up to steps_per_launch - 1 ids past its EOS before its slot is freed --
// those ids are dropped; stats [4] = {launches, steps, prefills, sequence-steps decoded and dropped}.
//
// With a non-greedy sampler (SetSampler: GeneratOnPrompt::Sample, GoPT.cpp:614-630 -- temperature, top-k, top-p, xorshift coin) the launches run ONE step each and leave the
// logits (pick = 0); kf_sample then draws every occupied slot's id from its own logits with the slot's own rng state, seeded at the request's start with seed + request
// index: answer r equals Fish::Generate's on prompt r under SetSampler(seed + r).
int XcdReplicas::Chat(const int32_t* prompts, const int32_t* prompt_len, int n_req, int stride, int max_new, int eos, int32_t* out, int32_t* out_len, long long* stats,
                      const int32_t* max_new_each) {
    const MODEL_CARD& c = hFish->config;
    if (!prompts || !prompt_len || !out || !out_len || n_req < 1 || stride < 1 || max_new < 1) return KF_INVALID_ARGS;
    if (max_new_each)
        for (int r = 0; r < n_req; r++)
            if (max_new_each[r] < 1 || max_new_each[r] > max_new) return KF_INVALID_ARGS;
    const bool sampled = !samp_params.greedy();
    auto draw = [&](int s) -> int {  // the slot's next id from its logits: state {token, pos} -> {id, pos + 1}, ids out [pos] = id
        return (samp_params.true_topk ? kf_sample_topk : kf_sample)(hFish->ctx, ToX(logits) + (size_t)s * c.vocab, c.vocab, samp_params.top_k, samp_params.temperature,
                                                                     samp_params.top_p, d_rng + s, nullptr, d_state + 4 * s, d_tokens_out + (size_t)s * c.n_ctx,
                                                                     d_forced + (size_t)s * c.n_ctx, c.n_ctx);
    };
    for (int r = 0; r < n_req; r++)
        if (prompt_len[r] < 1 || prompt_len[r] > stride || prompt_len[r] >= c.n_ctx) return KF_INVALID_ARGS;
    kf_ctx* ctx = hFish->ctx;
    KF_TRY(Fresh());
    struct Slot { int req = -1, len = 0, want = 0, have = 0; };
    std::vector<Slot> slot(n_seq);
    std::vector<int32_t> row(c.n_ctx);
    long long st[4] = {0, 0, 0, 0};
    int next = 0, done = 0;
    for (int s = 0; s < n_seq; s++) KF_TRY(Park(s, true));
    auto finish = [&](int s, int n_ids) -> int {  // the slot's answer out, the slot parked
        Slot& q = slot[s];
        KF_TRY(kf_d2h(ctx, row.data(), d_tokens_out + (size_t)s * c.n_ctx, (size_t)(q.len - 1 + q.have) * 4));
        for (int i = 0; i < max_new; i++) out[(size_t)q.req * max_new + i] = i < n_ids ? row[q.len - 1 + i] : -1;
        out_len[q.req] = n_ids;
        st[3] += q.have - n_ids;
        q.req = -1, done++;
        return Park(s, true);
    };
    auto serve = [&]() -> int {
        while (done < n_req) {
            for (bool again = true; again && next < n_req;) { /* refill: the free slots take the next prompts -- together (PrefillBatch) when several are free and asked for */
                again = false;
                std::vector<int> fs;
                const int PC = hFish->prefill_chunk; /* rows of one token batch */
                int tmax = 0;
                for (int s = 0; s < n_seq && next + (int)fs.size() < n_req && (int)fs.size() < (prefill_batch > 1 ? prefill_batch : 1); s++) {
                    if (slot[s].req >= 0) continue;
                    const int len = prompt_len[next + (int)fs.size()], t2 = len > tmax ? len : tmax;
                    if (!fs.empty() && (long long)(fs.size() + 1) * t2 > PC) break; /* the batch's rows fit the token-batch buffers */
                    tmax = t2, fs.push_back(s);
                }
                if (fs.empty()) break;
                const int m = (int)fs.size(), r0 = next;
                for (int i = 0; i < m; i++) {
                    Slot& q = slot[fs[i]];
                    q.req = next++, q.len = prompt_len[q.req], q.have = 1;  // the prefill picks the answer's first id
                    const int room = c.n_ctx - q.len;                       // the ids the cache has rows for: the id behind row p needs row p
                    const int asked = max_new_each ? max_new_each[q.req] : max_new; /* a limit of its own per request, or the common one */
                    q.want = asked < room + 1 ? asked : room + 1;
                    KF_TRY(kf_memset32(ctx, d_forced + (size_t)fs[i] * c.n_ctx, -1, (size_t)c.n_ctx)); /* free running */
                }
                if (m > 1)
                    KF_TRY(PrefillBatch(fs.data(), prompts + (size_t)r0 * stride, prompt_len + r0, m, stride));
                else
                    KF_TRY(Prefill(fs[0], prompts + (size_t)r0 * stride, prompt_len[r0]));
                st[2] += m;
                for (int i = 0; i < m; i++) {
                    const int s = fs[i];
                    Slot& q = slot[s];
                    if (sampled) { /* the prefill left the last prompt token's logits in the slot's logits and picked greedily: draw the answer's first id instead */
                        const uint64_t seed = samp_params.seed + (uint64_t)q.req;
                        KF_TRY(kf_set_state(ctx, reinterpret_cast<int32_t*>(d_rng + s), (int)(uint32_t)seed, (int)(uint32_t)(seed >> 32))); /* two words on the stream */
                        KF_TRY(kf_set_state(ctx, d_state + 4 * s, prompts[(size_t)q.req * stride + q.len - 1], q.len - 1));
                        KF_TRY(draw(s));
                    }
                    if (eos >= 0) {
                        int32_t first;
                        KF_TRY(kf_d2h(ctx, &first, d_tokens_out + (size_t)s * c.n_ctx + (q.len - 1), 4));
                        if (first == eos) q.want = 1;
                    }
                    if (q.have >= q.want) { KF_TRY(finish(s, q.want)); continue; }  // a one-id answer: the slot takes another prompt at once
                    KF_TRY(Park(s, false));
                }
                again = true; /* more free slots may be waiting (the batch limit, the buffer's rows, a one-id answer) */
            }
            int k = sampled ? 1 : steps_per_launch, active = 0;
            for (int s = 0; s < n_seq; s++)
                if (slot[s].req >= 0) active++, k = slot[s].want - slot[s].have < k ? slot[s].want - slot[s].have : k;
            if (!active) continue;
            KF_TRY(kf_xengine_steps(ctx, engine, ToX(x), d_state, k, sampled ? 0 : 1));
            if (sampled)
                for (int s = 0; s < n_seq; s++)
                    if (slot[s].req >= 0) KF_TRY(draw(s));
            st[0]++, st[1] += k, steps_run += k;
            for (int s = 0; s < n_seq; s++) {
                Slot& q = slot[s];
                if (q.req < 0) continue;
                const int had = q.have;
                q.have += k;
                int n_ids = q.have >= q.want ? q.want : -1;
                if (eos >= 0) {
                    KF_TRY(kf_d2h(ctx, row.data(), d_tokens_out + (size_t)s * c.n_ctx + (q.len - 1 + had), (size_t)k * 4));
                    for (int i = 0; i < k; i++)
                        if (row[i] == eos) { n_ids = had + i + 1; break; }
                }
                if (n_ids >= 0) KF_TRY(finish(s, n_ids));
            }
        }
        return KF_OK;
    };
    const int rc = serve();
    for (int s = 0; s < n_seq; s++) { /* whatever happened, the object leaves as it came: every slot free running */
        const int r2 = Park(s, false);
        if (rc == KF_OK && r2 != KF_OK) return r2;
    }
    if (stats) for (int i = 0; i < 4; i++) stats[i] = st[i];
    return rc != KF_OK ? rc : Check();
}
int XcdReplicas::Check() {
    if (!engine) return KF_OK;
    const int rc = kf_xengine_check(hFish->ctx, engine);
    if (rc == KF_INTERNAL_ERR) kf_xengine_reset(hFish->ctx, engine);
    return rc;
}

// ---- tensor parallel over the XCDs
XcdTP::~XcdTP() {
    if (ranks.empty()) return;
    kf_ctx* ctx = ranks[0]->ctx;
    kf_sync(ctx);
    if (engine) kf_xengine_destroy(engine);
    if (engine_ws) kf_free(ctx, engine_ws);
    if (d_state) kf_free(ctx, d_state);
    if (d_forced) kf_free(ctx, d_forced);
    if (d_tokens_out) kf_free(ctx, d_tokens_out);
}
int XcdTP::Build(Fish** fs, int world) {
    if (!fs || world < 1) return KF_INVALID_ARGS;
    for (int r = 0; r < world; r++)
        if (!fs[r]) return KF_INVALID_ARGS;
    ranks.assign(fs, fs + world);
    Fish* f0 = fs[0];
    kf_ctx* ctx = f0->ctx;
    const MODEL_CARD& c = f0->config;
    const int kvd = c.n_head_kv * c.head_dim;
    const size_t rank_elems = (size_t)c.nLayer * c.n_ctx * kvd;
    key = GT(ctx, "xtp.key", typNUMBER::BF16, kvd, c.n_ctx * c.nLayer * world);
    val = GT(ctx, "xtp.val", typNUMBER::BF16, kvd, c.n_ctx * c.nLayer * world);
    if (!key || !val) return KF_OUTOF_GPUMEMORY;
    KF_TRY(kf_memset(ctx, key->data, 0, rank_elems * world * 2));
    KF_TRY(kf_memset(ctx, val->data, 0, rank_elems * world * 2));
    std::vector<std::vector<kf_engine_layer>> Ls(world);
    std::vector<kf_engine_desc> ds(world);
    std::vector<const kf_engine_desc*> dp(world);
    std::vector<kf_weight> heads(world);
    std::vector<const kf_weight*> hp(world);
    std::vector<int32_t> row0(world);
    vocab = 0;
    for (int r = 0; r < world; r++) {
        Fish* f = fs[r];
        const MODEL_CARD& cr = f->config;
        if (cr.nLayer != c.nLayer || cr.n_ctx != c.n_ctx || cr.nEmbed != c.nEmbed || cr.n_head_kv * cr.head_dim != kvd) {
            why = "the ranks' cards disagree";
            return KF_INVALID_ARGS;
        }
        Ls[r].resize(c.nLayer);
        for (int l = 0; l < c.nLayer; l++) {
            SelfAttention* a = f->attn[l].get();
            FFN* m = f->ffn[l].get();
            SLP* s[7] = {&a->Q, &a->K, &a->V, &a->proj_cat, &m->gate, &m->up, &m->down};
            for (int j = 0; j < 7; j++) {
                if (!s[j]->w || s[j]->b) {
                    why = "a layer matrix is missing or carries a bias";
                    return KF_ENGINE_NOT_SERVED;
                }
                Ls[r][l].w[j] = s[j]->w->desc();
            }
            if (!a->norm.w || !m->norm.w || m->n_hot >= 0) {
                why = "a norm weight is missing or a hot-row mask is set";
                return KF_ENGINE_NOT_SERVED;
            }
            Ls[r][l].hot_ffn = nullptr;
            Ls[r][l].norm_in = ToX(a->norm.w), Ls[r][l].norm_post = ToX(m->norm.w);
            Ls[r][l].q_norm = a->normQ.w ? ToX(a->normQ.w) : nullptr, Ls[r][l].k_norm = a->normK.w ? ToX(a->normK.w) : nullptr;
            Ls[r][l].kcache = ToX(key) + (size_t)r * rank_elems + (size_t)l * c.n_ctx * kvd;
            Ls[r][l].vcache = ToX(val) + (size_t)r * rank_elems + (size_t)l * c.n_ctx * kvd;
        }
        kf_engine_desc& d = ds[r];
        std::memset(&d, 0, sizeof(d));
        d.n_layer = c.nLayer, d.dim = cr.nEmbed, d.n_head = cr.n_head, d.n_kv = cr.n_head_kv, d.head_dim = cr.head_dim, d.ffn = cr.n_ff;
        d.kv_stride = kvd, d.max_seq = c.n_ctx;
        d.rms_eps = cr.rms_eps, d.qk_eps = cr.qk_eps, d.rope_table = f0->rope_table, d.layers = Ls[r].data();
        dp[r] = &d;
        if (!f->embed.w || !f->head.proj.w || !f->final_norm.w) {
            why = "embedding / head / final norm missing";
            return KF_ENGINE_NOT_SERVED;
        }
        heads[r] = f->head.proj.w->desc(), hp[r] = &heads[r], row0[r] = vocab;
        vocab += cr.vocab;
    }
    logits = GT(ctx, "xtp.logits", typNUMBER::BF16, vocab, 1);
    x = GT(ctx, "xtp.x", typNUMBER::BF16, c.nEmbed, 1);
    if (!logits || !x) return KF_OUTOF_GPUMEMORY;
    KF_TRY(kf_malloc(ctx, 16, (void**)&d_state));
    KF_TRY(kf_malloc(ctx, (size_t)c.n_ctx * 4, (void**)&d_forced));
    KF_TRY(kf_malloc(ctx, (size_t)c.n_ctx * 4, (void**)&d_tokens_out));
    KF_TRY(kf_memset(ctx, d_state, 0, 16));
    KF_TRY(kf_memset(ctx, d_forced, 0xff, (size_t)c.n_ctx * 4));
    KF_TRY(kf_memset(ctx, d_tokens_out, 0, (size_t)c.n_ctx * 4));
    const size_t bytes = kf_xengine_workspace_bytes_tp(dp[0]);
    KF_TRY(kf_malloc(ctx, bytes, &engine_ws));
    int rc = kf_xengine_create_tp(ctx, dp.data(), world, engine_ws, bytes, &engine);
    if (rc != KF_OK) {
        why = kf_last_error();
        return rc == KF_UNSUPPORTED_DATATYPE ? KF_ENGINE_NOT_SERVED : rc;
    }
    kf_weight we = f0->embed.w->desc();
    rc = kf_xengine_set_embedding(ctx, engine, &we, d_forced, c.n_ctx);
    if (rc == KF_OK) rc = kf_xengine_set_head_tp(ctx, engine, hp.data(), row0.data(), ToX(f0->final_norm.w), ToX(logits), d_tokens_out, c.n_ctx);
    if (rc != KF_OK) {
        why = kf_last_error();
        return rc == KF_UNSUPPORTED_DATATYPE ? KF_ENGINE_NOT_SERVED : rc;
    }
    built_gen.clear();
    for (int r = 0; r < world; r++) built_gen.push_back(fs[r]->weights_gen);
    return KF_OK;
}
int XcdTP::SetForced(const int32_t* ids, int n) {
    const int n_ctx = ranks[0]->config.n_ctx;
    if (n < 0 || n > n_ctx) return KF_INVALID_ARGS;
    std::vector<int32_t> row(n_ctx, -1);
    for (int i = 0; i < n; i++) row[i] = ids[i];
    return kf_h2d(ranks[0]->ctx, d_forced, row.data(), row.size() * 4);
}
int XcdTP::SetState(int token, int pos) {
    if (pos < 0 || pos >= ranks[0]->config.n_ctx || token < 0 || token >= vocab) return KF_INVALID_ARGS;
    return kf_set_state(ranks[0]->ctx, d_state, token, pos);
}
int XcdTP::RunSteps(int n) {
    if (!engine || n < 1) return KF_INVALID_ARGS;
    for (size_t r = 0; r < ranks.size(); r++)
        if (r >= built_gen.size() || ranks[r]->weights_gen != built_gen[r]) { /* never a step on stale shards (the engine's tables and its fused q | k | v copy are of the old weights) */
            why = "a rank's weights changed since this engine was built (kfh_weights_changed / a weight set again): destroy it and create it again";
            return KF_INVALID_ARGS;
        }
    for (int i = 0; i < n;) {
        const int m = n - i < steps_per_launch ? n - i : steps_per_launch;
        KF_TRY(kf_xengine_steps(ranks[0]->ctx, engine, ToX(x), d_state, m, 1));
        i += m;
    }
    return KF_OK;
}
int XcdTP::Check() {
    if (!engine) return KF_OK;
    const int rc = kf_xengine_check(ranks[0]->ctx, engine);
    if (rc == KF_INTERNAL_ERR) kf_xengine_reset(ranks[0]->ctx, engine);
    return rc;
}

}  // namespace koifish

// ================================================================================================ C entry points
// Flat C surface over the classes above for ctypes (tests, bench.py) -- host-side convenience, not part of the
// kernel ABI.  Handles are koifish::Fish*.
using namespace koifish;
extern "C" {

void* kfh_create(int device, void* stream, int dim, int n_layer, int n_head, int n_kv, int head_dim, int ffn, int vocab, int n_ctx, float rms_eps, float qk_eps,
                 float theta, int* rc_out) {
    Fish* f = new Fish();
    MODEL_CARD c;
    c.nEmbed = dim, c.nLayer = n_layer, c.n_head = n_head, c.n_head_kv = n_kv, c.head_dim = head_dim, c.n_ff = ffn, c.vocab = vocab, c.n_ctx = n_ctx;
    c.rms_eps = rms_eps, c.qk_eps = qk_eps, c.rope_theta = theta;
    int rc = f->Build(c, device, stream);
    if (rc_out) *rc_out = rc;
    if (rc != KF_OK) {
        delete f;
        return nullptr;
    }
    return f;
}
void kfh_destroy(void* h) { delete reinterpret_cast<Fish*>(h); }
const char* kfh_host_error(void) { return g_host_err.c_str(); } /* why the last kfh_create failed */
void* kfh_ctx(void* h) { return reinterpret_cast<Fish*>(h)->ctx; }
int kfh_set_fuse_level(void* h, int lvl) {
    reinterpret_cast<Fish*>(h)->fuse_level = lvl;
    return KF_OK;
}
// sparse forward: h_hot[ffn] (1 = hot, CS_Picker::hot) for one layer's FFN, NULL = dense again.  The mask becomes a device row list here (load time).
int kfh_set_hot(void* h, int layer, const int32_t* h_hot, int n) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (layer < 0 || layer >= f->config.nLayer) return KF_INVALID_ARGS;
    FFN* m = f->ffn[layer].get();
    if (h_hot && n != f->config.n_ff) return KF_INVALID_ARGS; /* arguments first: a rejected call changes nothing */
    f->DropEngineTable(); /* the engine's layer table (and the captured graphs) hold the masks: rebuilt on the next step; the resident bf16 copies and the measured delays do not
                             depend on them and stay */
    f->weights_gen++;     /* an XcdReplicas built on this Fish holds the masks' addresses in its layer table too: its next use re-creates the engine */
    if (!h_hot) {
        if (m->n_hot >= 0) f->masked_layers--;
        m->n_hot = -1, m->hot_rows.reset(), m->hot_mask.reset();
        return KF_OK;
    }
    hGTensor mask = GT(f->ctx, "hot", typNUMBER::I32, n), rows = GT(f->ctx, "hot_rows", typNUMBER::I32, n + 4);
    if (!mask || !rows) return KF_OUTOF_GPUMEMORY;
    KF_TRY(kf_h2d(f->ctx, mask->data, h_hot, (size_t)n * 4));
    int32_t* d_count = reinterpret_cast<int32_t*>(rows->data) + n;
    KF_TRY(kf_hot_rows(f->ctx, reinterpret_cast<const int32_t*>(mask->data), n, reinterpret_cast<int32_t*>(rows->data), d_count));
    int32_t cnt = 0;
    KF_TRY(kf_d2h(f->ctx, &cnt, d_count, 4));
    if (m->n_hot < 0) f->masked_layers++;
    m->hot_rows = rows, m->hot_mask = mask, m->n_hot = cnt;
    return KF_OK;
}
int kfh_n_hot(void* h, int layer) {
    Fish* f = reinterpret_cast<Fish*>(h);
    return (layer < 0 || layer >= f->config.nLayer) ? KF_INVALID_ARGS : f->ffn[layer]->n_hot;
}
// the persistent decode engine: on (default) / off; captured step graphs are dropped so that the next step is captured the new way
int kfh_set_engine(void* h, int on) {
    Fish* f = reinterpret_cast<Fish*>(h);
    f->use_engine = on != 0;
    for (auto& g : f->graphs)
        if (g) kf_graph_destroy(g), g = nullptr;
    return KF_OK;
}
// summation order of the decode kernels (kf_set_canonical): 1 (default) the canonical order shared with the CPU oracle, 0 the v_dot2c forms; captured graphs are dropped
int kfh_set_canonical(void* h, int on) {
    Fish* f = reinterpret_cast<Fish*>(h);
    for (auto& g : f->graphs)
        if (g) kf_graph_destroy(g), g = nullptr;
    return kf_set_canonical(f->ctx, on);
}
// steps enqueued (or captured) through the engine so far; -1: the engine does not serve this model
int kfh_engine_steps(void* h) {
    Fish* f = reinterpret_cast<Fish*>(h);
    return f->engine_state < 0 ? -1 : f->engine_steps;
}
// diagnostics (scratch/eng_stamps.py, scratch/eng_ab.py): per-phase stamps of one workgroup of the engine; the first-sweep delays of its hand-offs
extern "C" int kfdbg_engine_stamps(kf_engine* e, unsigned long long* h_out, int n_words);
extern "C" int kfdbg_engine_stamps_enable(kf_engine* e, int wg);
extern "C" int kfdbg_engine_set_delays(kf_engine* e, const int* d6);
int kfh_engine_stamps(void* h, unsigned long long* out, int n) {
    Fish* f = reinterpret_cast<Fish*>(h);
    return f->engine ? kfdbg_engine_stamps(f->engine, out, n) : -1;
}
int kfh_engine_stamps_enable(void* h, int wg) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (f->engine_state == 0) f->EnsureEngine();
    for (auto& g : f->graphs)
        if (g) kf_graph_destroy(g), g = nullptr; /* captured launches hold the product instantiation */
    return f->engine ? kfdbg_engine_stamps_enable(f->engine, wg) : -1;
}
int kfh_engine_set_delays(void* h, const int* d6) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (f->engine_state == 0) f->EnsureEngine();
    for (auto& g : f->graphs)
        if (g) kf_graph_destroy(g), g = nullptr;
    return f->engine ? kfdbg_engine_set_delays(f->engine, d6) : -1;
}
// why the engine does not serve this model ("" when it does; valid once a step has been tried or kfh_engine_stamps_enable / kfh_engine_only forced the build)
const char* kfh_engine_why(void* h) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (f->engine_state == 0) f->EnsureEngine();
    return f->engine_why.c_str();
}
// kf_engine_tune at the position the decode state holds; us[0] / us[1] = mean launch time before / after
int kfh_engine_tune(void* h, int passes, float* us2) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (f->engine_state == 0) f->EnsureEngine();
    if (f->engine_state <= 0 || !f->engine_embed) return KF_ENGINE_NOT_SERVED;
    int32_t st[4];
    KF_TRY(kf_d2h(f->ctx, st, f->d_state, 16));
    f->tok_pos = st[1];
    float b = 0.f, a = 0.f;
    const int rc = kf_engine_tune(f->ctx, f->engine, ToX(f->x), f->d_state, f->pos_bound(), passes, &b, &a);
    if (us2) us2[0] = b, us2[1] = a;
    return rc;
}
// resident bf16 copies for long prompts: on (default) / off, and the byte budget above which a model goes without (0 keeps the current one); takes effect at the next Prefill
/* after an IN-PLACE change of weight data this Fish was given as device pointers: every derived copy (resident bf16 copies, the engine's tables, captured graphs) is dropped */
int kfh_weights_changed(void* h) {
    reinterpret_cast<Fish*>(h)->DropEngine();
    return KF_OK;
}
int kfh_set_prefill_resident(void* h, int on, size_t max_bytes) {
    Fish* f = reinterpret_cast<Fish*>(h);
    f->DropResident();
    f->prefill_resident = on ? 1 : 0;
    if (max_bytes) f->resident_max_bytes = max_bytes;
    return KF_OK;
}
size_t kfh_resident_bytes(void* h) {
    Fish* f = reinterpret_cast<Fish*>(h);
    return f->deq_arena ? kf_dequant_arena_used(f->ctx) : 0;
}
int kfh_set_engine_autotune(void* h, int passes) {
    Fish* f = reinterpret_cast<Fish*>(h);
    f->engine_autotune = passes < 0 ? 0 : passes;
    return KF_OK;
}
// statistics of the engine's hand-offs at `pos` (kf_engine_statistics as 14 ints: sweeps[6], polls, delay[6], tuned)
int kfh_engine_stats(void* h, int pos, int32_t* out14) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (!f->engine) return KF_ENGINE_NOT_SERVED;
    kf_engine_statistics s;
    f->tok_pos = pos;
    KF_TRY(kf_engine_stats(f->ctx, f->engine, f->pos_bound(), &s));
    for (int i = 0; i < 6; i++) out14[i] = s.sweeps[i], out14[7 + i] = s.delay[i];
    out14[6] = s.polls, out14[13] = s.tuned;
    return KF_OK;
}
// n launches of the engine alone at the position d_state holds (bench.py times the kernel with events around this); the residual stream is
// re-read from the embedding each time so that the values stay those of a real step
int kfh_engine_only(void* h, int n) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (f->engine_state == 0) f->EnsureEngine();
    if (f->engine_state <= 0) return KF_ENGINE_NOT_SERVED;
    int32_t st[4];
    KF_TRY(kf_d2h(f->ctx, st, f->d_state, 16));
    f->tok_pos = st[1];
    f->graph_mode = true;
    int rc = KF_OK;
    for (int i = 0; i < n && rc == KF_OK; i++) rc = kf_engine_step(f->ctx, f->engine, ToX(f->x), ToX(f->x), f->d_state, f->pos_bound());
    f->graph_mode = false;
    return rc;
}
// synchronises; KF_INTERNAL_ERR when one of the engine's hand-off polls has timed out
int kfh_engine_check(void* h) {
    Fish* f = reinterpret_cast<Fish*>(h);
    return f->EngineCheck();
}

static SLP* slot_of(Fish* f, int layer, int slot) {
    if (layer < 0) return slot == 1 ? &f->head.proj : nullptr;
    if (layer >= f->config.nLayer) return nullptr;
    SelfAttention* a = f->attn[layer].get();
    FFN* m = f->ffn[layer].get();
    switch (slot) {
        case 0: return &a->Q;
        case 1: return &a->K;
        case 2: return &a->V;
        case 3: return &a->proj_cat;
        case 4: return &m->gate;
        case 5: return &m->up;
        case 6: return &m->down;
    }
    return nullptr;
}

// slot: 0 q,1 k,2 v,3 o,4 gate,5 up,6 down (layer >= 0); layer = -1: slot 0 embed_tokens, 1 lm_head.
// Either a host blob (`data||gama`, copied to a fresh device allocation) or an existing device pointer.
static int set_weight_impl(void* h, int layer, int slot, int type, int ne0, int ne1, const void* blob, size_t blob_bytes, size_t szData, int is_device, int lGroup,
                           int qMin, int qMax, int qBias, bool normal_float);
int kfh_set_weight(void* h, int layer, int slot, int type, int ne0, int ne1, const void* blob, size_t blob_bytes, size_t szData, int is_device, int lGroup,
                   int qMin, int qMax, int qBias) {
    return set_weight_impl(h, layer, slot, type, ne0, ne1, blob, blob_bytes, szData, is_device, lGroup, qMin, qMax, qBias, false);
}
// the same for a tensor whose quant card says isNormalFloat (QUANT_MODE::RTNf): Q4 nibble stream || gama with a 16-entry table per row
int kfh_set_weight_lut(void* h, int layer, int slot, int ne0, int ne1, const void* blob, size_t blob_bytes, size_t szData, int is_device) {
    return set_weight_impl(h, layer, slot, (int)typNUMBER::Q4, ne0, ne1, blob, blob_bytes, szData, is_device, 0, 0, 15, 0, true);
}
static int set_weight_impl(void* h, int layer, int slot, int type, int ne0, int ne1, const void* blob, size_t blob_bytes, size_t szData, int is_device, int lGroup,
                           int qMin, int qMax, int qBias, bool normal_float) {
    Fish* f = reinterpret_cast<Fish*>(h);
    auto t = std::make_shared<GTensor>();
    t->type = (typNUMBER)type, t->ne[0] = ne0, t->ne[1] = ne1;
    t->szData = szData, t->szGama = blob_bytes - szData;
    t->quant.T_group = lGroup, t->quant.qMin = qMin, t->quant.qMax = qMax, t->quant.qBias = qBias;
    t->quant.isNormalFloat = normal_float;
    if (is_device) {
        t->data = const_cast<void*>(blob), t->ctx = f->ctx, t->owned = false;
    } else {
        int rc = t->LoadBlob(f->ctx, blob, blob_bytes);
        if (rc != KF_OK) return rc;
    }
    f->DropEngine(); /* the engine's device table (and every captured graph) holds the old tensors' addresses */
    if (layer < 0 && slot == 0) {
        f->embed.w = t;
        return KF_OK;
    }
    SLP* s = slot_of(f, layer, slot);
    if (!s) return KF_INVALID_ARGS;
    s->w = t, s->nOut = ne0, s->nIn = ne1;
    return f->EnsureLinearScratch(t->desc(), f->prefill_chunk); /* load time: the launches themselves never allocate */
}
// tie_word_embeddings: lm_head shares embed_tokens' tensor (Neuron.cpp:349-356)
int kfh_tie_head(void* h) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (!f->embed.w) return KF_INVALID_ARGS;
    f->DropEngine();
    f->head.proj.w = f->embed.w, f->head.proj.nOut = f->embed.w->ne[0], f->head.proj.nIn = f->embed.w->ne[1];
    return KF_OK;
}
// slot: 0 input_layernorm, 1 post_attention_layernorm, 2 q_norm, 3 k_norm; layer = -1: final norm.  bf16 [n]
int kfh_set_norm(void* h, int layer, int slot, const void* w, int n, int is_device) {
    Fish* f = reinterpret_cast<Fish*>(h);
    auto t = std::make_shared<GTensor>();
    t->type = typNUMBER::BF16, t->ne[0] = n, t->szData = (size_t)n * 2;
    if (is_device) {
        t->data = const_cast<void*>(w), t->ctx = f->ctx;
    } else {
        int rc = t->LoadBlob(f->ctx, w, (size_t)n * 2);
        if (rc != KF_OK) return rc;
    }
    f->DropEngine();
    if (layer < 0) {
        f->final_norm.w = t;
        return KF_OK;
    }
    if (layer >= f->config.nLayer) return KF_INVALID_ARGS;
    SelfAttention* a = f->attn[layer].get();
    switch (slot) {
        case 0: a->norm.w = t; break;
        case 1: f->ffn[layer]->norm.w = t; break;
        case 2: a->normQ.w = t; break;
        case 3: a->normK.w = t; break;
        default: return KF_INVALID_ARGS;
    }
    return KF_OK;
}

// one eager step; logits (bf16[vocab]) and hidden copied to host when non-null; returns the greedy id or < 0
int kfh_forward(void* h, int token, int pos, uint16_t* h_logits) {
    Fish* f = reinterpret_cast<Fish*>(h);
    int rc = f->ForwardOnRLS(token, pos);
    if (rc != KF_OK) return rc;
    int32_t st[4];
    rc = kf_d2h(f->ctx, st, f->d_state, 16);
    if (rc != KF_OK) return rc;
    if (h_logits) {
        rc = kf_d2h(f->ctx, h_logits, f->head.preLogits->data, (size_t)f->config.vocab * 2);
        if (rc != KF_OK) return rc;
    }
    return f->fuse_level == 0 ? st[2] : st[0];
}
int kfh_generate(void* h, const int* prompt, int n_prompt, int n_new, int* out, int use_graph) {
    return reinterpret_cast<Fish*>(h)->Generate(prompt, n_prompt, n_new, out, use_graph != 0);
}
int kfh_set_forced(void* h, const int32_t* forced, int n) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (n > f->config.n_ctx) return KF_INVALID_ARGS;
    return kf_h2d(f->ctx, f->d_forced, forced, (size_t)n * 4);
}
int kfh_prefill(void* h, const int* tokens, int n, int pos0) { return reinterpret_cast<Fish*>(h)->Prefill(tokens, n, pos0); }
int kfh_set_prefill_mode(void* h, int mode, int chunk) {
    Fish* f = reinterpret_cast<Fish*>(h);
    f->prefill_mode = mode;
    if (chunk > 0 && !f->gBUFF.bX) f->prefill_chunk = chunk;
    return KF_OK;
}
int kfh_set_sampler(void* h, float temperature, float top_p, int top_k, uint64_t seed) {
    CHAT_SAMPLER s;
    s.temperature = temperature, s.top_p = top_p, s.top_k = top_k & 0xFFFF, s.seed = seed;
    s.true_topk = (top_k & 0x10000) != 0;  // bit 16 of top_k: candidates = the k largest logits (kf_sample_topk)
    return reinterpret_cast<Fish*>(h)->SetSampler(s);
}
int kfh_set_state(void* h, int token, int pos) { return reinterpret_cast<Fish*>(h)->SetState(token, pos); }
int kfh_run_steps(void* h, int pos, int n, int use_graph) { return reinterpret_cast<Fish*>(h)->RunSteps(pos, n, use_graph != 0); }
int kfh_sync(void* h) { return kf_sync(reinterpret_cast<Fish*>(h)->ctx); }
// the ids of the steps run so far; KF_INTERNAL_ERR (once) when a hand-off of the persistent engine timed out since the last check: the ids behind the failure are not valid
int kfh_get_tokens(void* h, int32_t* out, int n) {
    Fish* f = reinterpret_cast<Fish*>(h);
    KF_TRY(f->EngineCheck());
    return kf_d2h(f->ctx, out, f->d_tokens_out, (size_t)n * 4);
}
void* kfh_kcache(void* h) { return reinterpret_cast<Fish*>(h)->cache.key->data; }
void* kfh_vcache(void* h) { return reinterpret_cast<Fish*>(h)->cache.val->data; }
void* kfh_logits(void* h) { return reinterpret_cast<Fish*>(h)->head.preLogits->data; }
void* kfh_hidden(void* h) { return reinterpret_cast<Fish*>(h)->x->data; }
// ---- tensor parallel
int kfh_tp_init(void* h, int rank, int world, int vocab_row0) { return reinterpret_cast<Fish*>(h)->TPInit(rank, world, vocab_row0); }
void* kfh_tp_area(void* h) { return reinterpret_cast<Fish*>(h)->tp.area; }
int kfh_tp_set_peer(void* h, int r, void* area) { return reinterpret_cast<Fish*>(h)->TPSetPeer(r, area); }
int kfh_tp_export(void* h, unsigned char* handle64) {
    Fish* f = reinterpret_cast<Fish*>(h);
    return f->tp.area ? kf_tp_ipc_export(f->ctx, f->tp.area, handle64) : KF_INVALID_ARGS;
}
int kfh_tp_open_peer(void* h, int r, const unsigned char* handle64) {
    Fish* f = reinterpret_cast<Fish*>(h);
    void* p = nullptr;
    KF_TRY(kf_tp_ipc_open(f->ctx, handle64, &p));
    f->tp.opened.push_back(p);
    return f->TPSetPeer(r, p);
}
// synchronises; KF_INTERNAL_ERR when a poll of this rank ran out of spins (a peer never pushed)
int kfh_tp_check(void* h) {
    Fish* f = reinterpret_cast<Fish*>(h);
    if (f->tp.world < 2) return KF_OK;
    KF_TRY(kf_sync(f->ctx));
    int32_t e = 0;
    KF_TRY(kf_d2h(f->ctx, &e, f->tp.comm.d_err, 4));
    return e == 0 ? KF_OK : KF_INTERNAL_ERR;
}
// R ranks of this process (all on the stream of rank 0) run n decode steps from `pos` in lock-step; tokens / positions come from each rank's d_state
int kfh_tp_group_run(void** hs, int R, int pos, int n, int use_graph) {
    if (R < 2 || R > 8) return KF_INVALID_ARGS;
    Fish* fs[8];
    for (int r = 0; r < R; r++) {
        fs[r] = reinterpret_cast<Fish*>(hs[r]);
        if (fs[r]->tp.world != R || fs[r]->tp.rank != r) return KF_INVALID_ARGS;
    }
    if (pos < 0 || pos + n > fs[0]->config.n_ctx) return KF_INVALID_ARGS;
    for (int r = 0; r < R; r++) KF_TRY(fs[r]->TPCommit());
    auto& graphs = fs[0]->tp.group_graphs;
    for (int i = 0; i < n; i++) {
        for (int r = 0; r < R; r++) fs[r]->tok_pos = pos + i, fs[r]->graph_mode = true;
        int rc = KF_OK;
        if (use_graph) {
            const int b = bucket_of(pos + i);
            if ((int)graphs.size() <= b) graphs.resize(b + 1, nullptr);
            if (!graphs[b]) {
                rc = kf_graph_begin(fs[0]->ctx);
                if (rc == KF_OK) {
                    rc = tp_group_enqueue(fs, R);
                    kf_graph* g = nullptr;
                    const int rc2 = kf_graph_end(fs[0]->ctx, &g);
                    if (rc == KF_OK) rc = rc2;
                    if (rc == KF_OK) graphs[b] = g;
                }
            }
            if (rc == KF_OK) rc = kf_graph_launch(fs[0]->ctx, graphs[b]);
        } else {
            rc = tp_group_enqueue(fs, R);
        }
        for (int r = 0; r < R; r++) fs[r]->graph_mode = false;
        KF_TRY(rc);
    }
    return KF_OK;
}

// ---- XCD-confined replicas: handles are koifish::XcdReplicas*
void* kfh_xr_create(void* fish, int n_seq, int* rc_out) {
    XcdReplicas* r = new XcdReplicas();
    const int rc = r->Build(reinterpret_cast<Fish*>(fish), n_seq);
    if (rc_out) *rc_out = rc;
    if (rc != KF_OK) {
        g_host_err = r->why;
        delete r;
        return nullptr;
    }
    return r;
}
void kfh_xr_destroy(void* h) { delete reinterpret_cast<XcdReplicas*>(h); }
int kfh_xr_set_forced(void* h, int seq, const int32_t* ids, int n) { return reinterpret_cast<XcdReplicas*>(h)->SetForced(seq, ids, n); }
int kfh_xr_set_state(void* h, int seq, int token, int pos) { return reinterpret_cast<XcdReplicas*>(h)->SetState(seq, token, pos); }
int kfh_xr_prefill(void* h, int seq, const int* tokens, int n) { return reinterpret_cast<XcdReplicas*>(h)->Prefill(seq, tokens, n); }
int kfh_xr_run_steps(void* h, int n) { return reinterpret_cast<XcdReplicas*>(h)->RunSteps(n); }
int kfh_xr_check(void* h) { return reinterpret_cast<XcdReplicas*>(h)->Check(); }
int kfh_xr_park(void* h, int seq, int on) { return reinterpret_cast<XcdReplicas*>(h)->Park(seq, on != 0); }
int kfh_xr_prefill_batch(void* h, const int* slots, const int32_t* tokens, const int* lens, int S, int stride) {
    return reinterpret_cast<XcdReplicas*>(h)->PrefillBatch(slots, tokens, lens, S, stride);
}
int kfh_xr_set_prefill_batch(void* h, int n) {
    XcdReplicas* r = reinterpret_cast<XcdReplicas*>(h);
    if (n < 1 || n > r->n_seq) return KF_INVALID_ARGS;
    r->prefill_batch = n;
    return KF_OK;
}
int kfh_xr_set_sampler(void* h, float temperature, float top_p, int top_k, uint64_t seed) {
    CHAT_SAMPLER s;
    s.temperature = temperature, s.top_p = top_p, s.top_k = top_k & 0xFFFF, s.seed = seed;
    s.true_topk = (top_k & 0x10000) != 0;
    return reinterpret_cast<XcdReplicas*>(h)->SetSampler(s);
}
int kfh_xr_chat(void* h, const int32_t* prompts, const int32_t* prompt_len, int n_req, int stride, int max_new, int eos, int32_t* out, int32_t* out_len, long long* stats) {
    return reinterpret_cast<XcdReplicas*>(h)->Chat(prompts, prompt_len, n_req, stride, max_new, eos, out, out_len, stats);
}
// max_new_each [n_req] (or NULL): request r's own limit, <= max_new (the width of out's rows)
int kfh_xr_chat_each(void* h, const int32_t* prompts, const int32_t* prompt_len, int n_req, int stride, int max_new, int eos, int32_t* out, int32_t* out_len, long long* stats,
                     const int32_t* max_new_each) {
    return reinterpret_cast<XcdReplicas*>(h)->Chat(prompts, prompt_len, n_req, stride, max_new, eos, out, out_len, stats, max_new_each);
}
int kfh_xr_status(void* h, int seq, int32_t* out4) { return reinterpret_cast<XcdReplicas*>(h)->Status(seq, out4); }
int kfh_xr_set_steps_per_launch(void* h, int n) {
    if (n < 1 || n > 4096) return KF_INVALID_ARGS;
    reinterpret_cast<XcdReplicas*>(h)->steps_per_launch = n;
    return KF_OK;
}
int kfh_xr_get_tokens(void* h, int seq, int32_t* out, int n) {
    XcdReplicas* r = reinterpret_cast<XcdReplicas*>(h);
    if (seq < 0 || seq >= r->n_seq || n < 0 || n > r->hFish->config.n_ctx) return KF_INVALID_ARGS;
    return kf_d2h(r->hFish->ctx, out, r->d_tokens_out + (size_t)seq * r->hFish->config.n_ctx, (size_t)n * 4);
}
int kfh_xr_get_state(void* h, int seq, int32_t* out2) {
    XcdReplicas* r = reinterpret_cast<XcdReplicas*>(h);
    if (seq < 0 || seq >= r->n_seq) return KF_INVALID_ARGS;
    return kf_d2h(r->hFish->ctx, out2, r->d_state + 4 * seq, 8);
}
void* kfh_xr_logits(void* h, int seq) {
    XcdReplicas* r = reinterpret_cast<XcdReplicas*>(h);
    return ToX(r->logits) + (size_t)seq * r->hFish->config.vocab;
}
void* kfh_xr_hidden(void* h, int seq) {
    XcdReplicas* r = reinterpret_cast<XcdReplicas*>(h);
    return ToX(r->x) + (size_t)seq * r->hFish->config.nEmbed;
}
void* kfh_xr_kcache(void* h, int seq) {
    XcdReplicas* r = reinterpret_cast<XcdReplicas*>(h);
    return ToX(r->key) + (size_t)seq * r->kv_seq_elems();
}
void* kfh_xr_vcache(void* h, int seq) {
    XcdReplicas* r = reinterpret_cast<XcdReplicas*>(h);
    return ToX(r->val) + (size_t)seq * r->kv_seq_elems();
}
void* kfh_xtp_create(void** fishes, int world, int* rc_out) {
    XcdTP* t = new XcdTP();
    const int rc = t->Build(reinterpret_cast<Fish**>(fishes), world);
    if (rc_out) *rc_out = rc;
    if (rc != KF_OK) {
        g_host_err = t->why;
        delete t;
        return nullptr;
    }
    return t;
}
void kfh_xtp_destroy(void* h) { delete reinterpret_cast<XcdTP*>(h); }
int kfh_xtp_set_forced(void* h, const int32_t* ids, int n) { return reinterpret_cast<XcdTP*>(h)->SetForced(ids, n); }
int kfh_xtp_set_state(void* h, int token, int pos) { return reinterpret_cast<XcdTP*>(h)->SetState(token, pos); }
int kfh_xtp_run_steps(void* h, int n) { return reinterpret_cast<XcdTP*>(h)->RunSteps(n); }
int kfh_xtp_check(void* h) { return reinterpret_cast<XcdTP*>(h)->Check(); }
int kfh_xtp_set_steps_per_launch(void* h, int n) {
    if (n < 1 || n > 4096) return KF_INVALID_ARGS;
    reinterpret_cast<XcdTP*>(h)->steps_per_launch = n;
    return KF_OK;
}
int kfh_xtp_get_tokens(void* h, int32_t* out, int n) {
    XcdTP* t = reinterpret_cast<XcdTP*>(h);
    if (n < 0 || n > t->ranks[0]->config.n_ctx) return KF_INVALID_ARGS;
    return kf_d2h(t->ranks[0]->ctx, out, t->d_tokens_out, (size_t)n * 4);
}
int kfh_xtp_vocab(void* h) { return reinterpret_cast<XcdTP*>(h)->vocab; }
void* kfh_xtp_logits(void* h) { return ToX(reinterpret_cast<XcdTP*>(h)->logits); }
void* kfh_xtp_hidden(void* h) { return ToX(reinterpret_cast<XcdTP*>(h)->x); }
void* kfh_xtp_kcache(void* h) { return ToX(reinterpret_cast<XcdTP*>(h)->key); }
void* kfh_xtp_vcache(void* h) { return ToX(reinterpret_cast<XcdTP*>(h)->val); }
extern "C" int kfdbg_xengine_variant(kf_xengine* e, int nwv, int depth);
extern "C" int kfdbg_xengine_stamps_enable(kf_xengine* e, int seq, int wg, int max_steps);
extern "C" int kfdbg_xengine_stamps(kf_xengine* e, unsigned long long* h_out, int n_words);
int kfh_xtp_variant(void* h, int nwv, int depth) { return kfdbg_xengine_variant(reinterpret_cast<XcdTP*>(h)->engine, nwv, depth); }
int kfh_xtp_stamps_enable(void* h, int rank, int wg, int max_steps) { return kfdbg_xengine_stamps_enable(reinterpret_cast<XcdTP*>(h)->engine, rank, wg, max_steps); }
int kfh_xtp_stamps(void* h, unsigned long long* out, int n) { return kfdbg_xengine_stamps(reinterpret_cast<XcdTP*>(h)->engine, out, n); }
int kfh_xr_variant(void* h, int nwv, int depth) { return kfdbg_xengine_variant(reinterpret_cast<XcdReplicas*>(h)->engine, nwv, depth); }
int kfh_xr_stamps_enable(void* h, int seq, int wg, int max_steps) { return kfdbg_xengine_stamps_enable(reinterpret_cast<XcdReplicas*>(h)->engine, seq, wg, max_steps); }
int kfh_xr_stamps(void* h, unsigned long long* out, int n) { return kfdbg_xengine_stamps(reinterpret_cast<XcdReplicas*>(h)->engine, out, n); }

int kfh_num_graphs(void* h) {
    int n = 0;
    for (auto g : reinterpret_cast<Fish*>(h)->graphs) n += g != nullptr;
    return n;
}
}
