"""BASELINE config 3 as a whole training step at toy size, every operator through the C-ABI: forward (token + position embedding, two hybrid GPT-2 blocks
-- attention matrices f8e5m2, MLP matrices 4-bit --, final LayerNorm, tied bf16 head, fused classifier) and backward (head, LayerNorm, MLP, GELU,
attention, QKV, embedding), against torch autograd in fp64 on the dequantised weights.  bf16 activations and gradients against fp64: the
gradients must agree to a few percent of their scale (max 2^-5, rms 2^-7; observed: max 0.9 %, rms 0.2 %)."""
import ctypes as C

import numpy as np
import pytest
import torch

from koifish_amd import lib as L
from oracle import oracle as O
from tests.conftest import bf16_t, u16

pytestmark = pytest.mark.gpu


def _lin(ctx, dw, x, n, m, bias=None, residual=None):
    y = torch.zeros(n, m, dtype=torch.bfloat16, device=ctx.device)
    d = dw.desc()
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), bias.data_ptr() if bias is not None else None, n, 1.0, 0.0, 1 if residual is not None else 0,
                             residual.data_ptr() if residual is not None else None) == 0, ctx.hip.kf_last_error()
    return y


@pytest.mark.parametrize("Bn,T", [(2, 64), (8, 256)])   # 128 rows: the hand-written GEMM kernels; 2048 rows: the dequantise + vendor-GEMM path
def test_gpt2_toy_training_step_vs_autograd(ctx, Bn, T):
    C_, H, NL, V, Vp = 128, 2, 2, 250, 256
    hd, N = C_ // H, Bn * T
    dev = ctx.device
    rng = np.random.default_rng(71)
    mk = lambda *s, std=0.08: O.f32_to_bf16(rng.normal(0, std, size=s).astype(np.float32))
    lnw = lambda: O.f32_to_bf16((1 + rng.normal(0, 0.1, C_)).astype(np.float32))
    f64 = lambda a: torch.tensor(O.bf16_to_f32(a).astype(np.float64))
    wte = np.zeros((Vp, C_), np.uint16)
    wte[:V] = mk(V, C_, std=0.2)
    wpe = mk(T, C_, std=0.05)
    ids = rng.integers(0, V, N).astype(np.int32)
    tgt = rng.integers(0, V, N).astype(np.int32)
    blocks = []
    for _ in range(NL):
        W = {"qkv": (mk(3 * C_, C_), mk(3 * C_), L.F8E5M2), "proj": (mk(C_, C_), mk(C_), L.F8E5M2), "fc": (mk(4 * C_, C_), mk(4 * C_), L.Q4), "proj2": (mk(C_, 4 * C_), mk(C_), L.Q4)}
        ow = {k: O.quantize(v_[0], v_[0].shape[0], v_[0].shape[1], v_[2]) for k, v_ in W.items()}
        blocks.append(dict(W=W, ow=ow, ln=(lnw(), mk(C_), lnw(), mk(C_)), dw={k: ctx.upload_blob(W[k][2], W[k][0].shape[0], W[k][0].shape[1], ow[k].blob()) for k in W},
                           db={k: bf16_t(W[k][1], dev) for k in W}, lnd=None))
    lnf = (lnw(), mk(C_))
    dhead = ctx.upload_blob(L.BF16, Vp, C_, O.quantize(wte, Vp, C_, L.BF16).blob())
    zeros = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
    stat = lambda: torch.zeros(N, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ forward on the device
    pos = np.tile(np.arange(T), Bn)
    e_tok, e_pos = bf16_t(wte[ids], dev), bf16_t(wpe[pos], dev)
    x = zeros(N, C_)
    assert ctx.hip.kf_add(ctx.h, e_tok.data_ptr(), e_pos.data_ptr(), x.data_ptr(), N * C_) == 0
    saved = []
    for b in blocks:
        l1w, l1b, l2w, l2b = (bf16_t(a, dev) for a in b["ln"])
        b["lnd"] = (l1w, l1b, l2w, l2b)
        h1, m1, r1 = zeros(N, C_), stat(), stat()
        assert ctx.hip.kf_layernorm(ctx.h, x.data_ptr(), l1w.data_ptr(), l1b.data_ptr(), h1.data_ptr(), N, C_, 1e-5, m1.data_ptr(), r1.data_ptr()) == 0
        qkv = _lin(ctx, b["dw"]["qkv"], h1, N, 3 * C_, bias=b["db"]["qkv"])
        att = zeros(N, C_)
        for s_ in range(Bn):
            sl = slice(s_ * T, (s_ + 1) * T)
            qc = qkv[sl, :C_].contiguous()
            assert ctx.hip.kf_attn_prefill(ctx.h, qc.data_ptr(), qkv[sl, C_:2 * C_].data_ptr(), qkv[sl, 2 * C_:].data_ptr(), att[sl].data_ptr(), 0, T, C_, H, H, hd, 3 * C_) == 0
        x2 = _lin(ctx, b["dw"]["proj"], att, N, C_, bias=b["db"]["proj"], residual=x)
        h2, m2, r2 = zeros(N, C_), stat(), stat()
        assert ctx.hip.kf_layernorm(ctx.h, x2.data_ptr(), l2w.data_ptr(), l2b.data_ptr(), h2.data_ptr(), N, C_, 1e-5, m2.data_ptr(), r2.data_ptr()) == 0
        fpre = _lin(ctx, b["dw"]["fc"], h2, N, 4 * C_, bias=b["db"]["fc"])
        g = torch.zeros_like(fpre)
        assert ctx.hip.kf_gelu(ctx.h, fpre.data_ptr(), g.data_ptr(), fpre.numel()) == 0
        xo = _lin(ctx, b["dw"]["proj2"], g, N, C_, bias=b["db"]["proj2"], residual=x2)
        saved.append(dict(x=x, h1=h1, m1=m1, r1=r1, qkv=qkv, att=att, x2=x2, h2=h2, m2=m2, r2=r2, fpre=fpre, g=g))
        x = xo
    lfw, lfb = bf16_t(lnf[0], dev), bf16_t(lnf[1], dev)
    hf, mf, rf = zeros(N, C_), stat(), stat()
    assert ctx.hip.kf_layernorm(ctx.h, x.data_ptr(), lfw.data_ptr(), lfb.data_ptr(), hf.data_ptr(), N, C_, 1e-5, mf.data_ptr(), rf.data_ptr()) == 0
    logits = _lin(ctx, dhead, hf, N, Vp)
    losses = torch.zeros(N, dtype=torch.float32, device=dev)
    td = torch.from_numpy(tgt).to(dev)
    assert ctx.hip.kf_fused_classifier(ctx.h, logits.data_ptr(), losses.data_ptr(), None, 1.0 / N, td.data_ptr(), Bn, T, V, Vp, None, 1) == 0
    # the padded vocabulary columns carry no gradient (the reference's head GEMM runs over V, not Vp, columns of the logit gradient)
    logits[:, V:] = 0

    # ------------------------------------------------------------------ backward on the device
    def lin_bwd(dw, dIn, inp, n_out_cols, want_bias=True):
        d = dw.desc()
        OC, IC = dIn.shape[1], inp.shape[1]
        delta, gW, gB = zeros(N, IC), zeros(OC, IC), zeros(OC)
        sc = torch.empty(ctx.hip.kf_linear_backward_scratch_bytes(OC, IC, N) + 256, dtype=torch.uint8, device=dev)
        sp = (sc.data_ptr() + 255) & ~255
        assert ctx.hip.kf_linear_backward(ctx.h, C.byref(d), dIn.data_ptr(), inp.data_ptr(), delta.data_ptr(), gW.data_ptr(), gB.data_ptr() if want_bias else None, N, 0, sp) == 0, \
            ctx.hip.kf_last_error()
        return delta, gW, gB

    def ln_bwd(dx, dout, inp, w, mean, rstd):
        gw, gb = zeros(C_), zeros(C_)
        sc = torch.empty(ctx.hip.kf_norm_backward_scratch_bytes(N, C_, 1) // 8 + 1, dtype=torch.float64, device=dev)
        assert ctx.hip.kf_norm_backward(ctx.h, dx.data_ptr(), gw.data_ptr(), gb.data_ptr(), dout.data_ptr(), inp.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), N, C_,
                                        sc.data_ptr()) == 0, ctx.hip.kf_last_error()
        return gw, gb

    grads = {}
    dhf, g_wte, _ = lin_bwd(dhead, logits, hf, C_, want_bias=False)
    dx = zeros(N, C_)
    grads["lnf.w"], grads["lnf.b"] = ln_bwd(dx, dhf, x, lfw, mf, rf)
    att_sc = torch.zeros(ctx.hip.kf_attn_backward_scratch_bytes(T, H, Bn) // 4 + 1, dtype=torch.float32, device=dev)
    for li in reversed(range(NL)):
        b, s_ = blocks[li], saved[li]
        dg, grads["%d.proj2.w" % li], grads["%d.proj2.b" % li] = lin_bwd(b["dw"]["proj2"], dx, s_["g"], 4 * C_)
        assert ctx.hip.kf_gelu_backward(ctx.h, dg.data_ptr(), s_["fpre"].data_ptr(), dg.numel()) == 0
        dh2, grads["%d.fc.w" % li], grads["%d.fc.b" % li] = lin_bwd(b["dw"]["fc"], dg, s_["h2"], C_)
        grads["%d.ln2.w" % li], grads["%d.ln2.b" % li] = ln_bwd(dx, dh2, s_["x2"], b["lnd"][2], s_["m2"], s_["r2"])
        datt, grads["%d.proj.w" % li], grads["%d.proj.b" % li] = lin_bwd(b["dw"]["proj"], dx, s_["att"], C_)
        dqkv = zeros(N, 3 * C_)
        qkv = s_["qkv"]
        assert ctx.hip.kf_attn_backward(ctx.h, qkv[:, :C_].data_ptr(), qkv[:, C_:2 * C_].data_ptr(), qkv[:, 2 * C_:].data_ptr(), 3 * C_, s_["att"].data_ptr(), datt.data_ptr(), C_,
                                        dqkv[:, :C_].data_ptr(), dqkv[:, C_:2 * C_].data_ptr(), dqkv[:, 2 * C_:].data_ptr(), 3 * C_, T, H, H, hd, Bn, att_sc.data_ptr()) == 0, ctx.hip.kf_last_error()
        dh1, grads["%d.qkv.w" % li], grads["%d.qkv.b" % li] = lin_bwd(b["dw"]["qkv"], dqkv, s_["h1"], C_)
        grads["%d.ln1.w" % li], grads["%d.ln1.b" % li] = ln_bwd(dx, dh1, s_["x"], b["lnd"][0], s_["m1"], s_["r1"])
    g_wpe = zeros(T, C_)
    idd = torch.from_numpy(ids).to(dev)
    assert ctx.hip.kf_embed_backward(ctx.h, g_wte.data_ptr(), C_, g_wpe.data_ptr(), dx.data_ptr(), idd.data_ptr(), Bn, T, C_, Vp) == 0   # tied: on top of the head's gradient
    ctx.sync()
    grads["wte"], grads["wpe"] = g_wte, g_wpe
    dev_loss = float(losses.mean())

    # ------------------------------------------------------------------ the same model in torch, fp64, on the dequantised weights
    P = {}
    leaf = lambda a: a.clone().requires_grad_(True)
    P["wte"], P["wpe"] = leaf(f64(wte)), leaf(f64(wpe))
    P["lnf.w"], P["lnf.b"] = leaf(f64(lnf[0])), leaf(f64(lnf[1]))
    for li, b in enumerate(blocks):
        for k in ("qkv", "proj", "fc", "proj2"):
            P["%d.%s.w" % (li, k)] = leaf(f64(O.dequant(b["ow"][k])).reshape(b["W"][k][0].shape))
            P["%d.%s.b" % (li, k)] = leaf(f64(b["W"][k][1]))
        for j, nm in enumerate(("ln1.w", "ln1.b", "ln2.w", "ln2.b")):
            P["%d.%s" % (li, nm)] = leaf(f64(b["ln"][j]))
    F = torch.nn.functional
    xt = P["wte"][torch.from_numpy(ids).long()] + P["wpe"][torch.from_numpy(pos).long()]
    for li in range(NL):
        g_ = lambda nm: P["%d.%s" % (li, nm)]
        h1 = F.layer_norm(xt, (C_,), g_("ln1.w"), g_("ln1.b"), 1e-5)
        qkv = h1 @ g_("qkv.w").T + g_("qkv.b")
        sp4 = lambda t_: t_.reshape(Bn, T, H, hd).transpose(1, 2)
        at = F.scaled_dot_product_attention(sp4(qkv[:, :C_]), sp4(qkv[:, C_:2 * C_]), sp4(qkv[:, 2 * C_:]), is_causal=True).transpose(1, 2).reshape(N, C_)
        x2 = xt + at @ g_("proj.w").T + g_("proj.b")
        h2 = F.layer_norm(x2, (C_,), g_("ln2.w"), g_("ln2.b"), 1e-5)
        xt = x2 + F.gelu(h2 @ g_("fc.w").T + g_("fc.b"), approximate="tanh") @ g_("proj2.w").T + g_("proj2.b")
    hft = F.layer_norm(xt, (C_,), P["lnf.w"], P["lnf.b"], 1e-5)
    loss = F.cross_entropy((hft @ P["wte"].T)[:, :V], torch.from_numpy(tgt).long())
    loss.backward()
    ref_loss = float(loss.detach())
    assert abs(dev_loss - ref_loss) <= 2.0 ** -7 * ref_loss
    worst = []
    for name, gd in grads.items():
        ref = P[name].grad.numpy()
        got = O.bf16_to_f32(u16(gd)).astype(np.float64).reshape(ref.shape)
        sc_ = np.abs(ref).max()
        mx, rms = np.abs(got - ref).max() / sc_, np.sqrt(((got - ref) ** 2).mean()) / sc_
        worst.append((mx, rms, name))
        assert mx <= 2.0 ** -5 and rms <= 2.0 ** -7, "%s: max %.4f rms %.4f of scale" % (name, mx, rms)
    print("largest gradient deviations (max, rms, tensor):", sorted(worst, reverse=True)[:3])


def test_qwen3_toy_training_step_vs_autograd(ctx):
    """The Qwen3 family's training step at toy size through the ABI: RMSNorm, Q / K / V (4-bit), per-head q/k RMSNorm + rotate-half RoPE, grouped-query
    causal attention, o_proj + residual, RMSNorm, gate / up / SwiGLU / down + residual, final RMSNorm, tied head, loss -- and the backward of each --
    against torch autograd in fp64 on the dequantised weights."""
    Bn, T, dim, H, KV, hd, ffn, NL, V, Vp, theta, eps = 2, 64, 128, 4, 2, 64, 256, 2, 250, 256, 10000.0, 1e-6
    N, Cq, Ck = Bn * T, H * hd, KV * hd
    dev = ctx.device
    rng = np.random.default_rng(91)
    mk = lambda *s, std=0.08: O.f32_to_bf16(rng.normal(0, std, size=s).astype(np.float32))
    nw = lambda n: O.f32_to_bf16((1 + rng.normal(0, 0.1, n)).astype(np.float32))
    f64 = lambda a: torch.tensor(O.bf16_to_f32(a).astype(np.float64))
    zeros = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
    wte = np.zeros((Vp, dim), np.uint16)
    wte[:V] = mk(V, dim, std=0.2)
    ids = rng.integers(0, V, N).astype(np.int32)
    tgt = rng.integers(0, V, N).astype(np.int32)
    shapes = {"q": (Cq, dim), "k": (Ck, dim), "v": (Ck, dim), "o": (dim, Cq), "gate": (ffn, dim), "up": (ffn, dim), "down": (dim, ffn)}
    layers = []
    for _ in range(NL):
        W = {k_: mk(*sh) for k_, sh in shapes.items()}
        ow = {k_: O.quantize(W[k_], shapes[k_][0], shapes[k_][1], L.Q4) for k_ in W}
        norms = {"n1": nw(dim), "n2": nw(dim), "qn": nw(hd), "kn": nw(hd)}
        layers.append(dict(ow=ow, dw={k_: ctx.upload_blob(L.Q4, shapes[k_][0], shapes[k_][1], ow[k_].blob()) for k_ in W}, norms=norms,
                           nd={k_: bf16_t(v_, dev) for k_, v_ in norms.items()}))
    nf = nw(dim)
    nf_d = bf16_t(nf, dev)
    dhead = ctx.upload_blob(L.BF16, Vp, dim, O.quantize(wte, Vp, dim, L.BF16).blob())
    table = ctx.rope_table(T, hd, theta)
    stat = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)

    def rms(x, w, rows, d_):
        y, r = zeros(rows, d_), stat(rows)
        assert ctx.hip.kf_rmsnorm(ctx.h, x.data_ptr(), w.data_ptr(), y.data_ptr(), rows, d_, eps, r.data_ptr()) == 0
        return y, r

    # ------------------------------------------------------------------ forward
    x = bf16_t(wte[ids], dev)
    saved = []
    for ly in layers:
        h1, r1 = rms(x, ly["nd"]["n1"], N, dim)
        qkv = zeros(N, Cq + 2 * Ck)   # q | k | v column blocks
        for nm, c0, w_ in (("q", 0, Cq), ("k", Cq, Ck), ("v", Cq + Ck, Ck)):
            qkv[:, c0:c0 + w_] = _lin(ctx, ly["dw"][nm], h1, N, w_)
        raw = qkv.clone()   # pre-norm q / k for the backward
        rq, rk = stat(N * H), stat(N * KV)
        # the fused per-head RMSNorm + RoPE of the inference path, positions 0 .. T - 1 of every sequence, with the per-head 1/rms kept for the backward
        assert ctx.hip.kf_qknorm_rope_train(ctx.h, qkv[:, :Cq].data_ptr(), qkv[:, Cq:].data_ptr(), ly["nd"]["qn"].data_ptr(), ly["nd"]["kn"].data_ptr(), table.data_ptr(), N, T,
                                            Cq + 2 * Ck, Cq + 2 * Ck, H, KV, hd, eps, rq.data_ptr(), rk.data_ptr()) == 0
        qc = qkv[:, :Cq].contiguous()
        att = zeros(N, Cq)
        assert ctx.hip.kf_attn_prefill_batch(ctx.h, qc.data_ptr(), qkv[:, Cq:].data_ptr(), qkv[:, Cq + Ck:].data_ptr(), att.data_ptr(), T, Cq, H, KV, hd, Cq + 2 * Ck, Bn) == 0
        x2 = _lin(ctx, ly["dw"]["o"], att, N, dim, residual=x)
        h2, r2 = rms(x2, ly["nd"]["n2"], N, dim)
        gate, up = _lin(ctx, ly["dw"]["gate"], h2, N, ffn), _lin(ctx, ly["dw"]["up"], h2, N, ffn)
        act = zeros(N, ffn)
        assert ctx.hip.kf_swiglu(ctx.h, gate.data_ptr(), up.data_ptr(), act.data_ptr(), N * ffn) == 0
        xo = _lin(ctx, ly["dw"]["down"], act, N, dim, residual=x2)
        saved.append(dict(x=x, h1=h1, r1=r1, raw=raw, rq=rq, rk=rk, qkv=qkv, att=att, x2=x2, h2=h2, r2=r2, gate=gate, up=up, act=act))
        x = xo
    hf, rf = rms(x, nf_d, N, dim)
    logits = _lin(ctx, dhead, hf, N, Vp)
    losses = torch.zeros(N, dtype=torch.float32, device=dev)
    td = torch.from_numpy(tgt).to(dev)
    assert ctx.hip.kf_fused_classifier(ctx.h, logits.data_ptr(), losses.data_ptr(), None, 1.0 / N, td.data_ptr(), Bn, T, V, Vp, None, 1) == 0
    logits[:, V:] = 0

    # ------------------------------------------------------------------ backward
    def lin_bwd(dw, dIn, inp, delta=None, acc=0):
        d = dw.desc()
        OC, IC = dIn.shape[1], inp.shape[1]
        delta = delta if delta is not None else zeros(N, IC)
        gW = zeros(OC, IC)
        sc = torch.empty(ctx.hip.kf_linear_backward_scratch_bytes(OC, IC, N) + 256, dtype=torch.uint8, device=dev)
        assert ctx.hip.kf_linear_backward(ctx.h, C.byref(d), dIn.data_ptr(), inp.data_ptr(), delta.data_ptr(), gW.data_ptr(), None, N, acc, (sc.data_ptr() + 255) & ~255) == 0, \
            ctx.hip.kf_last_error()
        return delta, gW

    def rms_bwd(dx, dout, inp, w, rstd, rows, d_):
        gw = zeros(d_)
        sc = torch.empty(ctx.hip.kf_norm_backward_scratch_bytes(rows, d_, 0) // 8 + 1, dtype=torch.float64, device=dev)
        assert ctx.hip.kf_norm_backward(ctx.h, dx.data_ptr(), gw.data_ptr(), None, dout.data_ptr(), inp.data_ptr(), w.data_ptr(), None, rstd.data_ptr(), rows, d_, sc.data_ptr()) == 0, \
            ctx.hip.kf_last_error()
        return gw

    grads = {}
    dhf, g_wte = lin_bwd(dhead, logits, hf)
    dx = zeros(N, dim)
    grads["nf"] = rms_bwd(dx, dhf, x, nf_d, rf, N, dim)
    att_sc = torch.zeros(ctx.hip.kf_attn_backward_scratch_bytes(T, H, Bn) // 4 + 1, dtype=torch.float32, device=dev)
    W_ = Cq + 2 * Ck
    for li in reversed(range(NL)):
        ly, s_ = layers[li], saved[li]
        dact, grads["%d.down" % li] = lin_bwd(ly["dw"]["down"], dx, s_["act"])
        dgate = zeros(N, ffn)
        assert ctx.hip.kf_swiglu_backward(ctx.h, dact.data_ptr(), dgate.data_ptr(), s_["gate"].data_ptr(), s_["up"].data_ptr(), N * ffn) == 0   # dact becomes d(up)
        dh2, grads["%d.up" % li] = lin_bwd(ly["dw"]["up"], dact, s_["h2"])
        _, grads["%d.gate" % li] = lin_bwd(ly["dw"]["gate"], dgate, s_["h2"], delta=dh2, acc=1)
        grads["%d.n2" % li] = rms_bwd(dx, dh2, s_["x2"], ly["nd"]["n2"], s_["r2"], N, dim)
        datt, grads["%d.o" % li] = lin_bwd(ly["dw"]["o"], dx, s_["att"])
        dqkv = zeros(N, W_)
        qkv = s_["qkv"]
        assert ctx.hip.kf_attn_backward(ctx.h, qkv[:, :Cq].data_ptr(), qkv[:, Cq:].data_ptr(), qkv[:, Cq + Ck:].data_ptr(), W_, s_["att"].data_ptr(), datt.data_ptr(), Cq,
                                        dqkv[:, :Cq].data_ptr(), dqkv[:, Cq:].data_ptr(), dqkv[:, Cq + Ck:].data_ptr(), W_, T, H, KV, hd, Bn, att_sc.data_ptr()) == 0, ctx.hip.kf_last_error()
        # RoPE backward on dq, dk (in place), then the q/k-norm backward per head row
        assert ctx.hip.kf_rope_backward(ctx.h, dqkv[:, :Cq].data_ptr(), table.data_ptr(), 0, N, T, W_, H, hd) == 0
        assert ctx.hip.kf_rope_backward(ctx.h, dqkv[:, Cq:].data_ptr(), table.data_ptr(), 0, N, T, W_, KV, hd) == 0
        dq_post, dk_post = dqkv[:, :Cq].contiguous().view(N * H, hd), dqkv[:, Cq:Cq + Ck].contiguous().view(N * KV, hd)
        dq_raw, dk_raw = zeros(N * H, hd), zeros(N * KV, hd)
        grads["%d.qn" % li] = rms_bwd(dq_raw, dq_post, s_["raw"][:, :Cq].contiguous().view(N * H, hd), ly["nd"]["qn"], s_["rq"], N * H, hd)
        grads["%d.kn" % li] = rms_bwd(dk_raw, dk_post, s_["raw"][:, Cq:Cq + Ck].contiguous().view(N * KV, hd), ly["nd"]["kn"], s_["rk"], N * KV, hd)
        dh1, grads["%d.q" % li] = lin_bwd(ly["dw"]["q"], dq_raw.view(N, Cq), s_["h1"])
        _, grads["%d.k" % li] = lin_bwd(ly["dw"]["k"], dk_raw.view(N, Ck), s_["h1"], delta=dh1, acc=1)
        _, grads["%d.v" % li] = lin_bwd(ly["dw"]["v"], dqkv[:, Cq + Ck:].contiguous(), s_["h1"], delta=dh1, acc=1)
        grads["%d.n1" % li] = rms_bwd(dx, dh1, s_["x"], ly["nd"]["n1"], s_["r1"], N, dim)
    idd = torch.from_numpy(ids).to(dev)
    assert ctx.hip.kf_embed_backward(ctx.h, g_wte.data_ptr(), dim, None, dx.data_ptr(), idd.data_ptr(), Bn, T, dim, Vp) == 0
    ctx.sync()
    grads["wte"] = g_wte
    dev_loss = float(losses.mean())

    # ------------------------------------------------------------------ torch, fp64
    P = {"wte": f64(wte).clone().requires_grad_(True), "nf": f64(nf).clone().requires_grad_(True)}
    for li, ly in enumerate(layers):
        for k_ in shapes:
            P["%d.%s" % (li, k_)] = f64(O.dequant(ly["ow"][k_])).reshape(shapes[k_]).clone().requires_grad_(True)
        for k_ in ("n1", "n2", "qn", "kn"):
            P["%d.%s" % (li, k_)] = f64(ly["norms"][k_]).clone().requires_grad_(True)
    F = torch.nn.functional
    rmsn = lambda t_, w_: t_ * torch.rsqrt((t_ * t_).mean(-1, keepdim=True) + eps) * w_
    posv = torch.arange(T, dtype=torch.float64)
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float64) / hd))
    ang = posv[:, None] * inv[None, :]
    cs, sn = torch.cos(ang), torch.sin(ang)   # [T, hd/2]

    def rope(t_):   # [Bn, T, heads, hd], rotate-half
        a, b = t_[..., :hd // 2], t_[..., hd // 2:]
        c_, s2 = cs[None, :, None, :], sn[None, :, None, :]
        return torch.cat([a * c_ - b * s2, a * s2 + b * c_], dim=-1)

    xt = P["wte"][torch.from_numpy(ids).long()]
    for li in range(NL):
        g_ = lambda nm: P["%d.%s" % (li, nm)]
        h1 = rmsn(xt, g_("n1"))
        q = rope(rmsn((h1 @ g_("q").T).reshape(Bn, T, H, hd), g_("qn"))).transpose(1, 2)
        k = rope(rmsn((h1 @ g_("k").T).reshape(Bn, T, KV, hd), g_("kn"))).transpose(1, 2)
        v = (h1 @ g_("v").T).reshape(Bn, T, KV, hd).transpose(1, 2)
        k, v = k.repeat_interleave(H // KV, dim=1), v.repeat_interleave(H // KV, dim=1)
        at = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(N, Cq)
        x2 = xt + at @ g_("o").T
        h2 = rmsn(x2, g_("n2"))
        xt = x2 + (F.silu(h2 @ g_("gate").T) * (h2 @ g_("up").T)) @ g_("down").T
    loss = F.cross_entropy((rmsn(xt, P["nf"]) @ P["wte"].T)[:, :V], torch.from_numpy(tgt).long())
    loss.backward()
    ref_loss = float(loss.detach())
    assert abs(dev_loss - ref_loss) <= 2.0 ** -7 * ref_loss
    worst = []
    for name, gd in grads.items():
        ref = P[name].grad.numpy()
        got = O.bf16_to_f32(u16(gd)).astype(np.float64).reshape(ref.shape)
        sc_ = np.abs(ref).max()
        mx, rms_ = np.abs(got - ref).max() / sc_, np.sqrt(((got - ref) ** 2).mean()) / sc_
        worst.append((round(mx, 4), round(rms_, 5), name))
        # the 64-element q/k-norm weight gradients sum bf16-rounded rows over all tokens and heads (and the forward rounds their rstd to bf16, as
        # the reference does): twice the rms allowance of the matrices
        rms_tol = 2.0 ** -6 if name.endswith(("qn", "kn")) else 2.0 ** -7
        assert mx <= 2.0 ** -5 and rms_ <= rms_tol, "%s: max %.4f rms %.4f of scale" % (name, mx, rms_)
    print("largest gradient deviations (max, rms, tensor):", sorted(worst, reverse=True)[:3])


def _ref_loss_fp64(Cn, H, NL, V, Bn, T, P, ids, tgt):
    """the toy GPT-2 in torch fp64 on given (already dequantised) parameters -> mean cross entropy"""
    F = torch.nn.functional
    hd, N = Cn // H, Bn * T
    pos = np.tile(np.arange(T), Bn)
    xt = P["wte"][torch.from_numpy(ids).long()] + P["wpe"][torch.from_numpy(pos).long()]
    for li in range(NL):
        g_ = lambda nm: P["%d.%s" % (li, nm)]
        h1 = F.layer_norm(xt, (Cn,), g_("ln1.w"), g_("ln1.b"), 1e-5)
        qkv = h1 @ g_("qkv.w").T + g_("qkv.b")
        sp4 = lambda t_: t_.reshape(Bn, T, H, hd).transpose(1, 2)
        at = F.scaled_dot_product_attention(sp4(qkv[:, :Cn]), sp4(qkv[:, Cn:2 * Cn]), sp4(qkv[:, 2 * Cn:]), is_causal=True).transpose(1, 2).reshape(N, Cn)
        x2 = xt + at @ g_("proj.w").T + g_("proj.b")
        h2 = F.layer_norm(x2, (Cn,), g_("ln2.w"), g_("ln2.b"), 1e-5)
        xt = x2 + F.gelu(h2 @ g_("fc.w").T + g_("fc.b"), approximate="tanh") @ g_("proj2.w").T + g_("proj2.b")
    hft = F.layer_norm(xt, (Cn,), P["lnf.w"], P["lnf.b"], 1e-5)
    return F.cross_entropy((hft @ P["wte"].T)[:, :V], torch.from_numpy(tgt).long())


def test_gpt2_two_consecutive_steps_with_update(ctx):
    """BASELINE config 3 as ONE loop (koifish_amd/train_step.py, the code bench.py's config3 leg times at full size): forward + loss -> backward into per-tensor gradient
    buffers -> kf_adamw on the model's own bf16 masters and moments (seeded stochastic rounding, Optimizer.cu:135-160) -> kf_quantize of every matrix back into its f8 / 4-bit
    blob (T.cu:105-175) -- twice, on the same batch, so that the second step's loss depends on the first step's update.  Checked per step:
      * the loss against torch fp64 on the oracle-dequantised blobs the step actually read (2^-7 relative),
      * the update: every master, both moments and the zeroed gradient equal the ORACLE's CU_adamw restatement applied to the device's own gradients, bit for bit,
      * the re-quantisation: every blob equals the oracle's quantiser applied to the updated master, byte for byte,
    and the loss falls from step 1 to step 2 to step 3 (the parameters really moved)."""
    from koifish_amd.train_step import GPT2Step, MATS
    Cn, H, NL, V, Vp, Bn, T = 128, 2, 2, 250, 256, 2, 64
    N = Bn * T
    rng = np.random.default_rng(303)
    bf = lambda a: torch.from_numpy(O.f32_to_bf16(a.astype(np.float32)).view(np.int16)).view(torch.bfloat16)
    mk = lambda *s, std=0.08: bf(rng.normal(0, std, size=s))
    lnw = lambda: bf(1 + rng.normal(0, 0.1, Cn))
    shapes = dict(qkv=(3 * Cn, Cn), proj=(Cn, Cn), fc=(4 * Cn, Cn), proj2=(Cn, 4 * Cn))
    wte = torch.zeros(Vp, Cn, dtype=torch.bfloat16)
    wte[:V] = mk(V, Cn, std=0.2)
    masters = dict(wte=wte, wpe=mk(T, Cn, std=0.05), lnf=(lnw(), mk(Cn)),
                   blocks=[dict({k: (mk(*shapes[k]), mk(shapes[k][0])) for k in MATS}, ln=(lnw(), mk(Cn), lnw(), mk(Cn))) for _ in range(NL)])
    st = GPT2Step(ctx, Cn, H, NL, V, Vp, Bn, T, masters=masters)
    ids = rng.integers(0, V, N).astype(np.int32)
    tgt = rng.integers(0, V, N).astype(np.int32)
    d_ids, d_tgt = torch.from_numpy(ids).to(ctx.device), torch.from_numpy(tgt).to(ctx.device)
    hp = dict(lr=2e-3, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.1, seed=99)
    f64 = lambda a_u16: torch.tensor(O.bf16_to_f32(a_u16).astype(np.float64))
    TYPE_OF = {L.F8E5M2: L.F8E5M2, L.Q4: L.Q4}

    def params_as_read():
        """what the step's forward multiplies: the dequantised blobs (oracle dequantiser on the device's bytes) and the bf16 tensors"""
        P = {}
        for e in st.params:
            nm = e["name"]
            key = nm.replace("h", "", 1) if nm.startswith("h") and nm[1].isdigit() else nm
            if e["type"] in TYPE_OF:
                ne0, ne1 = e["p"].shape
                P[key] = f64(u16(ctx.dequant(e["blob"]))).reshape(ne0, ne1)   # kf_dequant is itself bit-exact against the oracle (tests/test_gpu_ops.py)
            else:
                P[key] = f64(u16(e["p"])).reshape(tuple(e["p"].shape))
        return P
    losses = []
    for step in range(3):
        P = params_as_read()
        ref = float(_ref_loss_fp64(Cn, H, NL, V, Bn, T, P, ids, tgt))
        st.forward(d_ids, d_tgt)
        ctx.sync()
        dev_loss = float(st.losses.mean())
        assert abs(dev_loss - ref) <= 2.0 ** -7 * ref, "step %d: loss %.5f vs fp64 %.5f" % (step, dev_loss, ref)
        losses.append(dev_loss)
        if step == 2:
            break
        st.backward()
        ctx.sync()
        before = [(u16(e["p"]).copy(), u16(e["g"]).copy(), u16(e["m"]).copy(), u16(e["v"]).copy()) for e in st.params]
        assert all(g.any() for _, g, _, _ in before), "every tensor received a gradient"
        st.update(**hp)
        ctx.sync()
        t = st.t
        b1c, b2c = 1.0 - hp["beta1"] ** t, 1.0 - hp["beta2"] ** t
        for i, (e, (p0, g0, m0, v0)) in enumerate(zip(st.params, before)):
            p, g, m, v = (a.reshape(-1).copy() for a in (p0, g0, m0, v0))
            assert O.adamw(p, g, m, v, hp["lr"], hp["beta1"], hp["beta2"], b1c, b2c, hp["eps"], hp["wd"] if e["wd"] else 0.0, 1.0, (hp["seed"] + 7919 * t + i) & 0xFFFFFFFF) == 0
            assert np.array_equal(u16(e["p"]).reshape(-1), p), "step %d: master %s differs from the oracle's AdamW" % (step, e["name"])
            assert np.array_equal(u16(e["m"]).reshape(-1), m) and np.array_equal(u16(e["v"]).reshape(-1), v), e["name"]
            assert not u16(e["g"]).any()
            assert not np.array_equal(p, p0.reshape(-1)), "%s did not move" % e["name"]
            if e["type"] in TYPE_OF:   # the blob the next forward reads = the oracle's quantiser on the updated master
                ne0, ne1 = e["p"].shape
                ow = O.quantize(u16(e["p"]).reshape(ne0, ne1), ne0, ne1, e["type"])
                assert np.array_equal(e["blob"].blob.cpu().numpy(), np.frombuffer(ow.blob(), dtype=np.uint8)), "step %d: blob of %s" % (step, e["name"])
    assert losses[1] < losses[0] and losses[2] < losses[1], losses
    print("toy losses over three forwards on one batch:", ["%.4f" % v for v in losses])


def test_embed_pos_and_strided_attention(ctx):
    """the two operators the step's C++ sequencer uses instead of torch ops:
      * kf_embed_pos == bf16(wte[id] + wpe[t]) in fp32, round to nearest (encoder_forward_kernel3, kernel/embed.cuh:20-45) bit for bit, an out-of-range id reading row 0;
      * kf_attn_prefill_batch_strided reading q out of the fused [rows, 3C] buffer and writing a dense [rows, C] out == kf_attn_prefill_batch on a copy of the q columns,
        bit for bit;
      * kf_memset2d clears exactly the padded columns."""
    hip, dev = ctx.hip, ctx.device
    rng = np.random.default_rng(8)
    Bn, T, Cn, V, H = 3, 70, 128, 300, 2
    N = Bn * T
    wte = O.f32_to_bf16(rng.normal(0, 0.3, (V, Cn)).astype(np.float32))
    wpe = O.f32_to_bf16(rng.normal(0, 0.1, (T, Cn)).astype(np.float32))
    ids = rng.integers(0, V, N).astype(np.int32)
    ids[5], ids[9] = -3, V + 10
    d_wte, d_wpe, d_ids = bf16_t(wte, dev), bf16_t(wpe, dev), torch.from_numpy(ids).to(dev)
    out = torch.zeros(N, Cn, dtype=torch.bfloat16, device=dev)
    assert hip.kf_embed_pos(ctx.h, d_wte.data_ptr(), Cn, d_wpe.data_ptr(), d_ids.data_ptr(), Bn, T, Cn, V, out.data_ptr()) == 0, hip.kf_last_error()
    ctx.sync()
    safe = np.where((ids < 0) | (ids >= V), 0, ids)
    ref = O.f32_to_bf16(O.bf16_to_f32(wte[safe]) + O.bf16_to_f32(np.tile(wpe, (Bn, 1))))
    assert np.array_equal(u16(out), ref)
    assert hip.kf_embed_pos(ctx.h, d_wte.data_ptr(), Cn, d_wpe.data_ptr(), d_ids.data_ptr(), Bn, T, Cn + 4, V, out.data_ptr()) != 0   # C not a multiple of 8
    # attention on the fused rows
    qkv = bf16_t(O.f32_to_bf16(rng.normal(0, 1.0, (N, 3 * Cn)).astype(np.float32)), dev)
    qc = qkv[:, :Cn].contiguous()
    a0 = torch.zeros(N, Cn, dtype=torch.bfloat16, device=dev)
    a1 = torch.zeros(N, Cn, dtype=torch.bfloat16, device=dev)
    assert hip.kf_attn_prefill_batch(ctx.h, qc.data_ptr(), qkv[:, Cn:].data_ptr(), qkv[:, 2 * Cn:].data_ptr(), a0.data_ptr(), T, Cn, H, H, Cn // H, 3 * Cn, Bn) == 0, hip.kf_last_error()
    assert hip.kf_attn_prefill_batch_strided(ctx.h, qkv.data_ptr(), qkv[:, Cn:].data_ptr(), qkv[:, 2 * Cn:].data_ptr(), a1.data_ptr(), T, 3 * Cn, Cn, H, H, Cn // H, 3 * Cn, Bn) == 0, hip.kf_last_error()
    ctx.sync()
    assert u16(a0).any() and np.array_equal(u16(a0), u16(a1))
    assert hip.kf_attn_prefill_batch_strided(ctx.h, qkv.data_ptr(), qkv[:, Cn:].data_ptr(), qkv[:, 2 * Cn:].data_ptr(), a1.data_ptr(), T, 3 * Cn, Cn - 8, H, H, Cn // H, 3 * Cn, Bn) != 0   # out rows would overlap
    # the padded columns of a [rows, Vp] buffer
    Vp = 320
    lg = torch.full((N, Vp), 1.5, dtype=torch.bfloat16, device=dev)
    assert hip.kf_memset2d(ctx.h, lg.data_ptr() + V * 2, Vp * 2, 0, (Vp - V) * 2, N) == 0, hip.kf_last_error()
    ctx.sync()
    h = u16(lg)
    assert not h[:, V:].any() and (h[:, :V] == h[0, 0]).all()
