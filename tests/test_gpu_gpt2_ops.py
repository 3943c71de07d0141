"""GPT-2 family forward pieces (BASELINE config 3): LayerNorm, GELU, the K = 1600 GEMM, and one transformer block assembled from the ABI
(LayerNorm -> fused QKV GEMM + bias -> causal MHA -> projection + bias + residual -> LayerNorm -> FC + bias -> GELU -> projection + bias +
residual) against the same block assembled from the oracle's operators."""
import ctypes as C

import numpy as np
import pytest
import torch

from koifish_amd import lib as L
from oracle import oracle as O
from tests.conftest import bf16_t, close_bf16, u16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,dim", [(1, 768), (5, 1600), (64, 4096), (3, 50)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_layernorm_bit_exact(ctx, rows, dim, with_bias):
    rng = np.random.default_rng(rows * dim)
    x = O.f32_to_bf16(rng.normal(0.3, 2.0, size=(rows, dim)).astype(np.float32))
    w = O.f32_to_bf16((1 + rng.normal(0, 0.1, size=dim)).astype(np.float32))
    b = O.f32_to_bf16(rng.normal(0, 0.1, size=dim).astype(np.float32)) if with_bias else None
    y = torch.zeros(rows, dim, dtype=torch.bfloat16, device=ctx.device)
    mean = torch.zeros(rows, dtype=torch.float32, device=ctx.device)
    rstd = torch.zeros(rows, dtype=torch.float32, device=ctx.device)
    xd, wd = bf16_t(x, ctx.device), bf16_t(w, ctx.device)
    bd = bf16_t(b, ctx.device) if with_bias else None
    assert ctx.hip.kf_layernorm(ctx.h, xd.data_ptr(), wd.data_ptr(), bd.data_ptr() if with_bias else None, y.data_ptr(), rows, dim, 1e-5, mean.data_ptr(),
                                rstd.data_ptr()) == 0
    ctx.sync()
    ry, rm, rs = O.layernorm(x, w, b, 1e-5, want_stats=True)
    assert np.array_equal(u16(y), ry)
    assert np.array_equal(mean.cpu().numpy(), rm) and np.array_equal(rstd.cpu().numpy(), rs)


def test_gelu_bit_exact(ctx):
    rng = np.random.default_rng(5)
    x = O.f32_to_bf16(np.concatenate([rng.normal(0, 3, 70000), [0.0, -0.0, 12.0, -12.0, 1e-30, 40.0, -40.0]]).astype(np.float32))
    y = torch.zeros(x.size, dtype=torch.bfloat16, device=ctx.device)
    xd_ = bf16_t(x, ctx.device)   # named: a temporary is freed as soon as data_ptr() returns
    assert ctx.hip.kf_gelu(ctx.h, xd_.data_ptr(), y.data_ptr(), x.size) == 0
    ctx.sync()
    assert np.array_equal(u16(y), O.gelu(x))
    xf = O.bf16_to_f32(x).astype(np.float64)
    exact = 0.5 * xf * (1 + np.tanh(np.sqrt(2 / np.pi) * (xf + 0.044715 * xf ** 3)))
    assert np.abs(O.bf16_to_f32(u16(y)) - exact).max() <= 2.0 ** -8 * np.abs(exact).max()


def _linear(ctx, dw, x, n, m, bias=None, residual=None):
    y = torch.zeros(n, m, dtype=torch.bfloat16, device=ctx.device)
    d = dw.desc()
    rc = ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), bias.data_ptr() if bias is not None else None, n, 1.0, 0.0,
                           1 if residual is not None else 0, residual.data_ptr() if residual is not None else None)
    assert rc == 0, ctx.hip.kf_last_error()
    return y


@pytest.mark.parametrize("t", [L.Q4, L.F8E5M2, L.BF16])
@pytest.mark.parametrize("n", [96, 700])
def test_gemm_k1600(ctx, t, n):
    """n_embd = 1600: K is a multiple of 64 but not of 128 (4-bit groups straddle rows: 12.5 groups per row); 96 rows take the direct kernel,
    700 the staged tiles with a half-empty last k tile"""
    m, k = 4800, 1600
    rng = np.random.default_rng(t)
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(m, k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(n, k)).astype(np.float32))
    ow = O.quantize(w, m, k, t)
    dw = ctx.upload_blob(t, m, k, ow.blob())
    y = u16(_linear(ctx, dw, bf16_t(x, ctx.device), n, m))
    exact = O.bf16_to_f32(x).astype(np.float64) @ O.bf16_to_f32(O.dequant(ow)).astype(np.float64).T
    assert np.abs(O.bf16_to_f32(y) - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 1e-6
    for tt in (0, n - 1):
        assert close_bf16(y[tt], O.linear(ow, x[tt])).all()


def test_gpt2_block_forward(ctx):
    """one GPT-2 block, hybrid storage as in cases/gpt2/1558M_F8_B80: attention matrices f8e5m2, MLP matrices RTN 4-bit; T = 96 tokens"""
    C_, H, T = 256, 4, 96
    hd = C_ // H
    rng = np.random.default_rng(2)
    mk = lambda *s, std=0.05: O.f32_to_bf16(rng.normal(0, std, size=s).astype(np.float32))
    x = mk(T, C_, std=1.0)
    ln1w, ln1b, ln2w, ln2b = (O.f32_to_bf16((1 + rng.normal(0, 0.1, C_)).astype(np.float32)), mk(C_), O.f32_to_bf16((1 + rng.normal(0, 0.1, C_)).astype(np.float32)), mk(C_))
    W = {"qkv": (mk(3 * C_, C_), mk(3 * C_), L.F8E5M2), "proj": (mk(C_, C_), mk(C_), L.F8E5M2), "fc": (mk(4 * C_, C_), mk(4 * C_), L.Q4), "proj2": (mk(C_, 4 * C_), mk(C_), L.Q4)}
    ow = {k: O.quantize(v[0], v[0].shape[0], v[0].shape[1], v[2]) for k, v in W.items()}
    dw = {k: ctx.upload_blob(W[k][2], W[k][0].shape[0], W[k][0].shape[1], ow[k].blob()) for k in W}
    db = {k: bf16_t(W[k][1], ctx.device) for k in W}
    dev = ctx.device
    # ---- device
    xd = bf16_t(x, dev)
    h1 = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
    l1w_d, l1b_d, l2w_d, l2b_d = (bf16_t(a_, dev) for a_ in (ln1w, ln1b, ln2w, ln2b))
    assert ctx.hip.kf_layernorm(ctx.h, xd.data_ptr(), l1w_d.data_ptr(), l1b_d.data_ptr(), h1.data_ptr(), T, C_, 1e-5, None, None) == 0
    qkv = _linear(ctx, dw["qkv"], h1, T, 3 * C_, bias=db["qkv"])
    att = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
    q, k, v = qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:]
    # q rows are 3C apart inside the fused buffer; the attention entry takes one stride for q and out, so q goes through a compact copy
    qc = q.contiguous()
    assert ctx.hip.kf_attn_prefill(ctx.h, qc.data_ptr(), k.data_ptr(), v.data_ptr(), att.data_ptr(), 0, T, C_, H, H, hd, 3 * C_) == 0, ctx.hip.kf_last_error()
    x2 = _linear(ctx, dw["proj"], att, T, C_, bias=db["proj"], residual=xd)
    h2 = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
    assert ctx.hip.kf_layernorm(ctx.h, x2.data_ptr(), l2w_d.data_ptr(), l2b_d.data_ptr(), h2.data_ptr(), T, C_, 1e-5, None, None) == 0
    f = _linear(ctx, dw["fc"], h2, T, 4 * C_, bias=db["fc"])
    g = torch.zeros_like(f)
    assert ctx.hip.kf_gelu(ctx.h, f.data_ptr(), g.data_ptr(), f.numel()) == 0
    out = u16(_linear(ctx, dw["proj2"], g, T, C_, bias=db["proj2"], residual=x2))
    ctx.sync()
    # ---- oracle, token by token
    r1 = O.layernorm(x, ln1w, ln1b, 1e-5)
    rqkv = np.stack([O.linear(ow["qkv"], r1[t], bias=W["qkv"][1]) for t in range(T)])
    rq, rk, rv = rqkv[:, :C_], rqkv[:, C_:2 * C_], rqkv[:, 2 * C_:]
    ratt = np.stack([O.attn_decode(rq[t], rk[:t + 1], rv[:t + 1], t, H, H, hd, mode=O.ATTN_FUSED) for t in range(T)])
    rx2 = np.stack([O.add(x[t], O.linear(ow["proj"], ratt[t], bias=W["proj"][1])) for t in range(T)])
    r2 = O.layernorm(rx2, ln2w, ln2b, 1e-5)
    rf = np.stack([O.linear(ow["fc"], r2[t], bias=W["fc"][1]) for t in range(T)])
    rg = O.gelu(rf)
    rout = np.stack([O.add(rx2[t], O.linear(ow["proj2"], rg[t], bias=W["proj2"][1])) for t in range(T)])
    a, b = O.bf16_to_f32(out), O.bf16_to_f32(rout)
    assert np.abs(a - b).max() <= 2.0 ** -6 * np.abs(b).max()
    assert np.sqrt(((a - b) ** 2).mean()) <= 2.0 ** -9 * np.abs(b).max()


def test_gpt2_tiny_step(ctx):
    """BASELINE config 3 end to end at toy size: token + position embedding, two hybrid blocks (attention f8e5m2, MLP 4-bit), final LayerNorm, tied bf16
    LM head, fused classifier (mean loss, logit gradients), then one AdamW step -- the ABI calls against the same chain of oracle operators."""
    C_, H, T, NL, V, Vp = 128, 2, 48, 2, 203, 208
    hd = C_ // H
    rng = np.random.default_rng(31)
    mk = lambda *s, std=0.08: O.f32_to_bf16(rng.normal(0, std, size=s).astype(np.float32))
    lnw = lambda: O.f32_to_bf16((1 + rng.normal(0, 0.1, C_)).astype(np.float32))
    dev = ctx.device
    wte = np.zeros((Vp, C_), np.uint16)
    wte[:V] = mk(V, C_, std=0.2)
    wpe = mk(T, C_, std=0.05)
    ids = rng.integers(0, V, T).astype(np.int32)
    tgt = rng.integers(0, V, T).astype(np.int32)
    blocks = []
    for _ in range(NL):
        W = {"qkv": (mk(3 * C_, C_), mk(3 * C_), L.F8E5M2), "proj": (mk(C_, C_), mk(C_), L.F8E5M2), "fc": (mk(4 * C_, C_), mk(4 * C_), L.Q4), "proj2": (mk(C_, 4 * C_), mk(C_), L.Q4)}
        ow = {k: O.quantize(v[0], v[0].shape[0], v[0].shape[1], v[2]) for k, v in W.items()}
        blocks.append(dict(W=W, ow=ow, ln=(lnw(), mk(C_), lnw(), mk(C_)), dw={k: ctx.upload_blob(W[k][2], W[k][0].shape[0], W[k][0].shape[1], ow[k].blob()) for k in W},
                           db={k: bf16_t(W[k][1], dev) for k in W}))
    lnf = (lnw(), mk(C_))
    ohead = O.quantize(wte, Vp, C_, L.BF16)
    dhead = ctx.upload_blob(L.BF16, Vp, C_, ohead.blob())
    # ---- device
    e_tok, e_pos = bf16_t(wte[ids], dev), bf16_t(wpe, dev)
    x = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
    assert ctx.hip.kf_add(ctx.h, e_tok.data_ptr(), e_pos.data_ptr(), x.data_ptr(), T * C_) == 0
    for b in blocks:
        l1w, l1b, l2w, l2b = (bf16_t(a, dev) for a in b["ln"])
        h1 = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
        assert ctx.hip.kf_layernorm(ctx.h, x.data_ptr(), l1w.data_ptr(), l1b.data_ptr(), h1.data_ptr(), T, C_, 1e-5, None, None) == 0
        qkv = _linear(ctx, b["dw"]["qkv"], h1, T, 3 * C_, bias=b["db"]["qkv"])
        att = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
        qc = qkv[:, :C_].contiguous()
        assert ctx.hip.kf_attn_prefill(ctx.h, qc.data_ptr(), qkv[:, C_:2 * C_].data_ptr(), qkv[:, 2 * C_:].data_ptr(), att.data_ptr(), 0, T, C_, H, H, hd, 3 * C_) == 0
        x2 = _linear(ctx, b["dw"]["proj"], att, T, C_, bias=b["db"]["proj"], residual=x)
        h2 = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
        assert ctx.hip.kf_layernorm(ctx.h, x2.data_ptr(), l2w.data_ptr(), l2b.data_ptr(), h2.data_ptr(), T, C_, 1e-5, None, None) == 0
        f = _linear(ctx, b["dw"]["fc"], h2, T, 4 * C_, bias=b["db"]["fc"])
        g = torch.zeros_like(f)
        assert ctx.hip.kf_gelu(ctx.h, f.data_ptr(), g.data_ptr(), f.numel()) == 0
        x = _linear(ctx, b["dw"]["proj2"], g, T, C_, bias=b["db"]["proj2"], residual=x2)
    hf = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
    lnf_w, lnf_b = bf16_t(lnf[0], dev), bf16_t(lnf[1], dev)
    assert ctx.hip.kf_layernorm(ctx.h, x.data_ptr(), lnf_w.data_ptr(), lnf_b.data_ptr(), hf.data_ptr(), T, C_, 1e-5, None, None) == 0
    logits = _linear(ctx, dhead, hf, T, Vp)
    logits_fwd = u16(logits).copy()
    losses = torch.zeros(T, dtype=torch.float32, device=dev)
    td = torch.from_numpy(tgt).to(dev)
    assert ctx.hip.kf_fused_classifier(ctx.h, logits.data_ptr(), losses.data_ptr(), None, 1.0 / T, td.data_ptr(), 1, T, V, Vp, None, 1) == 0
    ctx.sync()
    # ---- oracle
    rx = np.stack([O.add(wte[ids[t]], wpe[t]) for t in range(T)])
    for b in blocks:
        W, ow = b["W"], b["ow"]
        r1 = O.layernorm(rx, b["ln"][0], b["ln"][1], 1e-5)
        rqkv = np.stack([O.linear(ow["qkv"], r1[t], bias=W["qkv"][1]) for t in range(T)])
        ratt = np.stack([O.attn_decode(rqkv[t, :C_], rqkv[:t + 1, C_:2 * C_], rqkv[:t + 1, 2 * C_:], t, H, H, hd, mode=O.ATTN_FUSED) for t in range(T)])
        rx2 = np.stack([O.add(rx[t], O.linear(ow["proj"], ratt[t], bias=W["proj"][1])) for t in range(T)])
        r2 = O.layernorm(rx2, b["ln"][2], b["ln"][3], 1e-5)
        rg = O.gelu(np.stack([O.linear(ow["fc"], r2[t], bias=W["fc"][1]) for t in range(T)]))
        rx = np.stack([O.add(rx2[t], O.linear(ow["proj2"], rg[t], bias=W["proj2"][1])) for t in range(T)])
    rh = O.layernorm(rx, lnf[0], lnf[1], 1e-5)
    rlog = np.ascontiguousarray(np.stack([O.linear(ohead, rh[t]) for t in range(T)]))
    a, bb = O.bf16_to_f32(logits_fwd[:, :V]), O.bf16_to_f32(rlog[:, :V])
    assert np.abs(a - bb).max() <= 2.0 ** -6 * np.abs(bb).max()
    rloss = np.zeros(T, np.float32)
    rgrad = rlog.copy()
    O.fused_classifier(rgrad, rloss, tgt, V, dloss=1.0 / T)
    got = losses.cpu().numpy()
    assert abs(got.mean() - rloss.mean()) <= 2.0 ** -7 * rloss.mean()
    # the classifier itself, on the device's own logits: bit for bit
    rloss2, rgrad2 = np.zeros(T, np.float32), logits_fwd.copy()
    O.fused_classifier(rgrad2, rloss2, tgt, V, dloss=1.0 / T)
    assert np.array_equal(got, rloss2) and np.array_equal(u16(logits), rgrad2)
    # ---- one AdamW step on the embedding table with a synthetic gradient (the backward GEMMs are not part of this round): bit for bit
    n = Vp * C_
    grad = O.f32_to_bf16(rng.normal(0, 0.01, n).astype(np.float32))
    p_d, g_d = bf16_t(wte.reshape(-1), dev), bf16_t(grad, dev)
    m_d, v_d = torch.zeros(n, dtype=torch.bfloat16, device=dev), torch.zeros(n, dtype=torch.bfloat16, device=dev)
    assert ctx.hip.kf_adamw(ctx.h, p_d.data_ptr(), g_d.data_ptr(), m_d.data_ptr(), v_d.data_ptr(), n, L.BF16, 3e-4, 0.9, 0.95, 0.1, 0.05, 1e-8, 0.1, 1.0, 1234, None) == 0
    ctx.sync()
    rp, rg2, rm, rv = wte.reshape(-1).copy(), grad.copy(), np.zeros(n, np.uint16), np.zeros(n, np.uint16)
    O.adamw(rp, rg2, rm, rv, 3e-4, 0.9, 0.95, 0.1, 0.05, 1e-8, 0.1, 1.0, 1234)
    assert np.array_equal(u16(p_d), rp) and np.array_equal(u16(m_d), rm) and np.array_equal(u16(v_d), rv)


def test_activation_backward_bit_exact(ctx):
    rng = np.random.default_rng(8)
    n = 8 * 1024 + 40
    x = O.f32_to_bf16(np.concatenate([rng.normal(0, 2.0, n - 5), [0.0, 11.0, -11.0, 40.0, -40.0]]).astype(np.float32))
    d = O.f32_to_bf16(rng.normal(0, 1.0, n).astype(np.float32))
    dd = bf16_t(d, ctx.device)
    xd_ = bf16_t(x, ctx.device)
    assert ctx.hip.kf_gelu_backward(ctx.h, dd.data_ptr(), xd_.data_ptr(), n) == 0
    ctx.sync()
    assert np.array_equal(u16(dd), O.gelu_backward(d, x))
    g = O.f32_to_bf16(rng.normal(0, 2.0, n).astype(np.float32))
    u = O.f32_to_bf16(rng.normal(0, 2.0, n).astype(np.float32))
    dd = bf16_t(d, ctx.device)
    dg = torch.zeros(n, dtype=torch.bfloat16, device=ctx.device)
    gd_, ud_ = bf16_t(g, ctx.device), bf16_t(u, ctx.device)
    assert ctx.hip.kf_swiglu_backward(ctx.h, dd.data_ptr(), dg.data_ptr(), gd_.data_ptr(), ud_.data_ptr(), n) == 0
    ctx.sync()
    r_up, r_gate = O.swiglu_backward(d, g, u)
    assert np.array_equal(u16(dd), r_up) and np.array_equal(u16(dg), r_gate)


@pytest.mark.parametrize("rows,dim", [(1, 768), (37, 1600), (300, 1024), (700, 5120), (5, 8192), (9, 8), (4100, 128), (8200, 64), (5000, 512), (4097, 256)])
@pytest.mark.parametrize("ln", [True, False])
def test_norm_backward_bit_exact(ctx, rows, dim, ln):
    rng = np.random.default_rng(rows * dim + ln)
    mk = lambda *s_, std=1.0, mu=0.0: O.f32_to_bf16(rng.normal(mu, std, size=s_).astype(np.float32))
    x, dout, dinp0 = mk(rows, dim, std=1.5, mu=0.2), mk(rows, dim), mk(rows, dim, std=0.5)
    w, dw0, db0 = O.f32_to_bf16((1 + rng.normal(0, 0.2, dim)).astype(np.float32)), mk(dim, std=0.5), mk(dim, std=0.5)
    xf = O.bf16_to_f32(x).astype(np.float64)
    mean = xf.mean(axis=1).astype(np.float32) if ln else None
    var = ((xf - (xf.mean(axis=1, keepdims=True) if ln else 0.0)) ** 2).mean(axis=1)
    rstd = (1.0 / np.sqrt(var + 1e-5)).astype(np.float32)
    dev = ctx.device
    d_dinp, d_dw, d_db = bf16_t(dinp0, dev), bf16_t(dw0, dev), bf16_t(db0, dev)
    nbytes = ctx.hip.kf_norm_backward_scratch_bytes(rows, dim, int(ln))
    scratch = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev)
    md = torch.from_numpy(mean).to(dev) if ln else None
    rd = torch.from_numpy(rstd).to(dev)
    do_d, x_d, w_d = bf16_t(dout, dev), bf16_t(x, dev), bf16_t(w, dev)
    assert ctx.hip.kf_norm_backward(ctx.h, d_dinp.data_ptr(), d_dw.data_ptr(), d_db.data_ptr() if ln else None, do_d.data_ptr(), x_d.data_ptr(),
                                    w_d.data_ptr(), md.data_ptr() if ln else None, rd.data_ptr(), rows, dim, scratch.data_ptr()) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    r_dinp, r_dw, r_db = dinp0.copy(), dw0.copy(), db0.copy()
    O.norm_backward(r_dinp, r_dw, r_db if ln else None, dout, x, w, mean, rstd)
    assert np.array_equal(u16(d_dinp), r_dinp)
    assert np.array_equal(u16(d_dw), r_dw)
    assert np.array_equal(u16(d_db), r_db)


def test_activation_kernels_unaligned_and_tails(ctx):
    """the vectorised element-wise kernels on pointers that are not 16-byte aligned (one element per thread) and on lengths with a tail"""
    rng = np.random.default_rng(12)
    for n, off in ((1, 0), (7, 0), (8, 0), (1031, 0), (1031, 3), (4096, 5)):
        x = O.f32_to_bf16(rng.normal(0, 2.0, n).astype(np.float32))
        d = O.f32_to_bf16(rng.normal(0, 1.0, n).astype(np.float32))
        xb = torch.zeros(n + 16, dtype=torch.bfloat16, device=ctx.device)
        db = torch.zeros(n + 16, dtype=torch.bfloat16, device=ctx.device)
        yb = torch.zeros(n + 16, dtype=torch.bfloat16, device=ctx.device)
        xb[off:off + n] = bf16_t(x, ctx.device)
        db[off:off + n] = bf16_t(d, ctx.device)
        assert ctx.hip.kf_gelu(ctx.h, xb[off:].data_ptr(), yb[off:].data_ptr(), n) == 0
        assert ctx.hip.kf_gelu_backward(ctx.h, db[off:].data_ptr(), xb[off:].data_ptr(), n) == 0
        ctx.sync()
        assert np.array_equal(u16(yb)[off:off + n], O.gelu(x)) and not u16(yb)[off + n:].any() and not u16(yb)[:off].any()
        assert np.array_equal(u16(db)[off:off + n], O.gelu_backward(d, x)) and not u16(db)[off + n:].any()


@pytest.mark.parametrize("B,T,C_,V,ldw", [(2, 48, 128, 61, 128), (4, 96, 1600, 300, 1600), (3, 40, 2056, 17, 2064), (1, 300, 64, 5, 64)])
def test_embed_backward_bit_exact(ctx, B, T, C_, V, ldw):
    rng = np.random.default_rng(B * T + C_)
    N = B * T
    dout = O.f32_to_bf16(rng.normal(0, 1.0, (N, C_)).astype(np.float32))
    tokens = rng.integers(0, V, N).astype(np.int32)
    tokens[rng.integers(0, N, 3)] = -1   # masked positions are skipped
    dwte0 = O.f32_to_bf16(rng.normal(0, 1.0, (V, ldw)).astype(np.float32))
    dwpe0 = O.f32_to_bf16(rng.normal(0, 1.0, (T, C_)).astype(np.float32))
    dev = ctx.device
    d_wte, d_wpe, d_out = bf16_t(dwte0, dev), bf16_t(dwpe0, dev), bf16_t(dout, dev)
    d_tok = torch.from_numpy(tokens).to(dev)
    assert ctx.hip.kf_embed_backward(ctx.h, d_wte.data_ptr(), ldw, d_wpe.data_ptr(), d_out.data_ptr(), d_tok.data_ptr(), B, T, C_, V) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    r_wte, r_wpe = dwte0.copy(), dwpe0.copy()
    O.embed_backward(r_wte, r_wpe, dout, tokens, B, T, V)
    assert np.array_equal(u16(d_wte), r_wte)
    assert np.array_equal(u16(d_wpe), r_wpe)
    # independent check of the oracle itself: fp64 scatter-add
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    ref = f(dwte0)[:, :C_].copy()
    for i, tk in enumerate(tokens):
        if 0 <= tk < V:
            ref[tk] += f(dout[i])
    assert np.abs(f(r_wte)[:, :C_] - ref).max() <= 2.0 ** -8 * np.abs(ref).max() + 1e-6
    assert np.array_equal(r_wte[:, C_:], dwte0[:, C_:])


@pytest.mark.parametrize("T,H,hd", [(48, 2, 64), (64, 1, 64), (130, 3, 64), (257, 2, 64), (100, 2, 128), (300, 1, 128)])
@pytest.mark.parametrize("fused_layout", [False, True])
def test_attn_backward_vs_oracle(ctx, T, H, hd, fused_layout):
    """causal MHA backward: separate [T, C] tensors, and q / k / v (and their gradients) as the column blocks of fused [T, 3C] buffers"""
    C_ = H * hd
    rng = np.random.default_rng(T * 13 + H)
    mk = lambda: O.f32_to_bf16(rng.normal(0, 1.0, (T, C_)).astype(np.float32))
    q, k, v, dO = mk(), mk(), mk(), mk()
    dev = ctx.device
    if fused_layout:
        qkv = bf16_t(np.concatenate([q, k, v], axis=1), dev)
        qd, kd, vd, ld = qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:], 3 * C_
        dqkv = torch.zeros(T, 3 * C_, dtype=torch.bfloat16, device=dev)
        dqd, dkd, dvd, ldd = dqkv[:, :C_], dqkv[:, C_:2 * C_], dqkv[:, 2 * C_:], 3 * C_
    else:
        qd, kd, vd, ld = bf16_t(q, dev), bf16_t(k, dev), bf16_t(v, dev), C_
        dqd, dkd, dvd = (torch.zeros(T, C_, dtype=torch.bfloat16, device=dev) for _ in range(3))
        ldd = C_
    od = torch.zeros(T, C_, dtype=torch.bfloat16, device=dev)
    qc = qd.contiguous()
    assert ctx.hip.kf_attn_prefill(ctx.h, qc.data_ptr(), kd.data_ptr(), vd.data_ptr(), od.data_ptr(), 0, T, C_, H, H, hd, ld) == 0, ctx.hip.kf_last_error()
    dOd = bf16_t(dO, dev)
    scratch = torch.zeros(ctx.hip.kf_attn_backward_scratch_bytes(T, H, 1) // 4 + 1, dtype=torch.float32, device=dev)
    assert ctx.hip.kf_attn_backward(ctx.h, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ld, od.data_ptr(), dOd.data_ptr(), C_, dqd.data_ptr(), dkd.data_ptr(), dvd.data_ptr(), ldd,
                                    T, H, H, hd, 1, scratch.data_ptr()) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    r_dq, r_dk, r_dv = O.attn_backward(q, k, v, u16(od), dO, H, hd)
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    for name, got, ref in (("dq", dqd, r_dq), ("dk", dkd, r_dk), ("dv", dvd, r_dv)):
        g, r = f(u16(got)), f(ref)
        assert np.abs(g - r).max() <= 2.0 ** -7 * np.abs(r).max(), name
        assert np.sqrt(((g - r) ** 2).mean()) <= 2.0 ** -9 * np.abs(r).max(), name


def test_attn_backward_rejects(ctx):
    z = torch.zeros(64 * 128, dtype=torch.bfloat16, device=ctx.device)
    s = torch.zeros(1024, dtype=torch.float32, device=ctx.device)
    p = z.data_ptr()
    assert ctx.hip.kf_attn_backward(ctx.h, p, p, p, 96, p, p, 96, p, p, p, 96, 16, 1, 1, 96, 1, s.data_ptr()) < 0   # head_dim 96: not covered
    assert ctx.hip.kf_attn_backward(ctx.h, p, p, p, 32, p, p, 128, p, p, p, 128, 16, 1, 1, 64, 1, s.data_ptr()) == -20     # stride below n_head * head_dim


def test_attn_backward_batched_sequences(ctx):
    """n_seq sequences in one call == the sequences one by one"""
    T, H, hd, Bn = 96, 2, 64, 3
    C_ = H * hd
    rng = np.random.default_rng(3)
    dev = ctx.device
    mk = lambda: bf16_t(O.f32_to_bf16(rng.normal(0, 1.0, (Bn * T, C_)).astype(np.float32)), dev)
    q, k, v, dO = mk(), mk(), mk(), mk()
    o = torch.zeros(Bn * T, C_, dtype=torch.bfloat16, device=dev)
    for b in range(Bn):
        sl = slice(b * T, (b + 1) * T)
        assert ctx.hip.kf_attn_prefill(ctx.h, q[sl].data_ptr(), k[sl].data_ptr(), v[sl].data_ptr(), o[sl].data_ptr(), 0, T, C_, H, H, hd, C_) == 0
    outs = []
    for mode in ("batched", "single"):
        dq, dk, dv = (torch.zeros(Bn * T, C_, dtype=torch.bfloat16, device=dev) for _ in range(3))
        sc = torch.zeros(ctx.hip.kf_attn_backward_scratch_bytes(T, H, Bn) // 4 + 1, dtype=torch.float32, device=dev)
        if mode == "batched":
            assert ctx.hip.kf_attn_backward(ctx.h, q.data_ptr(), k.data_ptr(), v.data_ptr(), C_, o.data_ptr(), dO.data_ptr(), C_, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), C_, T, H, H, hd, Bn,
                                            sc.data_ptr()) == 0
        else:
            for b in range(Bn):
                sl = slice(b * T, (b + 1) * T)
                assert ctx.hip.kf_attn_backward(ctx.h, q[sl].data_ptr(), k[sl].data_ptr(), v[sl].data_ptr(), C_, o[sl].data_ptr(), dO[sl].data_ptr(), C_, dq[sl].data_ptr(), dk[sl].data_ptr(),
                                                dv[sl].data_ptr(), C_, T, H, H, hd, 1, sc.data_ptr()) == 0
        ctx.sync()
        outs.append((u16(dq), u16(dk), u16(dv)))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_attn_prefill_batch_equals_per_sequence(ctx):
    T, H, hd, Bn = 100, 2, 64, 3
    C_ = H * hd
    rng = np.random.default_rng(9)
    dev = ctx.device
    qkv = bf16_t(O.f32_to_bf16(rng.normal(0, 1.0, (Bn * T, 3 * C_)).astype(np.float32)), dev)
    qc = qkv[:, :C_].contiguous()
    o1, o2 = (torch.zeros(Bn * T, C_, dtype=torch.bfloat16, device=dev) for _ in range(2))
    assert ctx.hip.kf_attn_prefill_batch(ctx.h, qc.data_ptr(), qkv[:, C_:].data_ptr(), qkv[:, 2 * C_:].data_ptr(), o1.data_ptr(), T, C_, H, H, hd, 3 * C_, Bn) == 0, ctx.hip.kf_last_error()
    for b in range(Bn):
        sl = slice(b * T, (b + 1) * T)
        assert ctx.hip.kf_attn_prefill(ctx.h, qc[sl].data_ptr(), qkv[sl, C_:2 * C_].data_ptr(), qkv[sl, 2 * C_:].data_ptr(), o2[sl].data_ptr(), 0, T, C_, H, H, hd, 3 * C_) == 0
    ctx.sync()
    assert np.array_equal(u16(o1), u16(o2))


@pytest.mark.parametrize("T,H,KV,hd", [(96, 4, 2, 64), (130, 8, 2, 128), (70, 2, 1, 64)])
def test_attn_backward_gqa_vs_oracle(ctx, T, H, KV, hd):
    """grouped-query attention backward: q | k | v as column blocks of one fused buffer with H + 2 KV heads per row; dk, dv sum over the group"""
    Cq, Ck = H * hd, KV * hd
    W = Cq + 2 * Ck
    rng = np.random.default_rng(T + H)
    dev = ctx.device
    qkv_h = O.f32_to_bf16(rng.normal(0, 1.0, (T, W)).astype(np.float32))
    dO_h = O.f32_to_bf16(rng.normal(0, 1.0, (T, Cq)).astype(np.float32))
    qkv = bf16_t(qkv_h, dev)
    qc = qkv[:, :Cq].contiguous()
    o = torch.zeros(T, Cq, dtype=torch.bfloat16, device=dev)
    assert ctx.hip.kf_attn_prefill_batch(ctx.h, qc.data_ptr(), qkv[:, Cq:].data_ptr(), qkv[:, Cq + Ck:].data_ptr(), o.data_ptr(), T, Cq, H, KV, hd, W, 1) == 0, ctx.hip.kf_last_error()
    dO = bf16_t(dO_h, dev)
    dqkv = torch.zeros(T, W, dtype=torch.bfloat16, device=dev)
    sc = torch.zeros(ctx.hip.kf_attn_backward_scratch_bytes(T, H, 1) // 4 + 1, dtype=torch.float32, device=dev)
    assert ctx.hip.kf_attn_backward(ctx.h, qkv[:, :Cq].data_ptr(), qkv[:, Cq:].data_ptr(), qkv[:, Cq + Ck:].data_ptr(), W, o.data_ptr(), dO.data_ptr(), Cq,
                                    dqkv[:, :Cq].data_ptr(), dqkv[:, Cq:].data_ptr(), dqkv[:, Cq + Ck:].data_ptr(), W, T, H, KV, hd, 1, sc.data_ptr()) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    r_dq, r_dk, r_dv = O.attn_backward(qkv_h[:, :Cq], qkv_h[:, Cq:Cq + Ck], qkv_h[:, Cq + Ck:], u16(o), dO_h, H, hd, n_kv=KV)
    got = u16(dqkv)
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    for name, g_, r_ in (("dq", got[:, :Cq], r_dq), ("dk", got[:, Cq:Cq + Ck], r_dk), ("dv", got[:, Cq + Ck:], r_dv)):
        g, r = f(g_), f(r_)
        assert np.abs(g - r).max() <= 2.0 ** -7 * np.abs(r).max(), name
        assert np.sqrt(((g - r) ** 2).mean()) <= 2.0 ** -9 * np.abs(r).max(), name


def test_rope_backward_bit_exact(ctx):
    T, H, hd, theta = 37, 3, 64, 1e6
    rng = np.random.default_rng(6)
    d = O.f32_to_bf16(rng.normal(0, 1.0, (2 * T, H * hd + 16)).astype(np.float32))   # two sequences of T rows, padded row stride
    dd = bf16_t(d, ctx.device)
    table = ctx.rope_table(T, hd, theta)
    assert ctx.hip.kf_rope_backward(ctx.h, dd.data_ptr(), table.data_ptr(), 0, 2 * T, T, H * hd + 16, H, hd) == 0
    ctx.sync()
    got = u16(dd)
    for t in range(2 * T):
        assert np.array_equal(got[t, :H * hd], O.rope_backward(d[t, :H * hd], H, hd, t % T, theta)), t
    assert np.array_equal(got[:, H * hd:], d[:, H * hd:])
    # the transpose property: <rope(x), g> == <x, rope_backward(g)> up to bf16 rounding
    x = O.f32_to_bf16(rng.normal(0, 1.0, H * hd).astype(np.float32))
    g = d[5, :H * hd]
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    lhs, rhs = float(f(O.rope(x, H, hd, 5, theta)) @ f(g)), float(f(x) @ f(O.rope_backward(g, H, hd, 5, theta)))
    assert abs(lhs - rhs) <= 2.0 ** -6 * (abs(lhs) + 1.0)
