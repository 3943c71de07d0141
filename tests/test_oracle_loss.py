"""CPU: the oracle's fused classifier (oracle/kf_oracle.c kfo_fused_classifier, restating fused_classifier.cuh:68-140) against an fp64 softmax
cross-entropy, and its fixed-recipe logf against libm.  No reference vectors exist for this kernel (parity unpinned)."""
import numpy as np
import pytest

from oracle import oracle as O
bf16_bits, bits_to_f32 = O.f32_to_bf16, O.bf16_to_f32


def test_logf_within_2ulp_of_libm():
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(-80, 80, 4000)), rng.uniform(0.5, 2.0, 4000), [1.0, 2.0, 0.5, 1e-40, 3e-39, 1.17549435e-38, 3.4e38]]).astype(np.float32)
    got = O.logf(x)
    ref64 = np.log(x.astype(np.float64))
    ref = ref64.astype(np.float32)
    ulp = np.abs(np.spacing(ref))
    # near log(1) = 0 the result is tiny: allow 2 ulp of the result or 2^-24 relative to |x - 1|
    err = np.abs(got.astype(np.float64) - ref64)
    assert np.all(err <= 2.0 * ulp + 1e-45), float(np.max(err / ulp))
    assert O.logf([0.0])[0] == -np.inf and np.isnan(O.logf([-1.0])[0]) and O.logf([np.inf])[0] == np.inf
    assert O.logf([1.0])[0] == 0.0


@pytest.mark.parametrize("rows,V,P", [(5, 50257, 50264), (3, 1000, 1000), (4, 66, 72), (2, 9000, 9008), (3, 7, 8)])
def test_fused_classifier_vs_fp64(rows, V, P):
    rng = np.random.default_rng(rows * 1000 + V)
    lg = np.zeros((rows, P), np.uint16)
    lg[:, :V] = bf16_bits((rng.standard_normal((rows, V)) * 3.0).astype(np.float32))
    lg[:, V:] = 0x7fc0  # padding must never be read
    tg = rng.integers(0, V, rows).astype(np.int32)
    x = bits_to_f32(lg[:, :V]).astype(np.float64)
    mx = x.max(axis=1, keepdims=True)
    e = np.exp(x - mx)
    pr = e / e.sum(axis=1, keepdims=True)
    loss_ref = -np.log(pr[np.arange(rows), tg])
    dloss = 1.0 / rows
    losses = np.full(rows, 0.25, np.float32)  # accumulates
    work = lg.copy()
    probs = O.fused_classifier(work, losses, tg, V, dloss=dloss, want_probs=True)
    assert np.allclose(losses - 0.25, loss_ref, rtol=2e-6, atol=2e-6)
    onehot = np.zeros((rows, V))
    onehot[np.arange(rows), tg] = 1.0
    d_ref = (pr - onehot) * dloss
    d_got = bits_to_f32(work[:, :V]).astype(np.float64)
    assert np.all(np.abs(d_got - d_ref) <= np.abs(d_ref) * 2.0 ** -8 + 1e-9)
    p_got = bits_to_f32(probs[:, :V]).astype(np.float64)
    assert np.all(np.abs(p_got - pr) <= pr * 2.0 ** -8 + 1e-12)
    assert np.array_equal(work[:, V:], lg[:, V:])  # padding untouched


def test_fused_classifier_mask_and_flags():
    rng = np.random.default_rng(11)
    rows, V, P = 6, 300, 304
    lg = np.zeros((rows, P), np.uint16)
    lg[:, :V] = bf16_bits(rng.standard_normal((rows, V)).astype(np.float32))
    tg = rng.integers(0, V, rows).astype(np.int32)
    mask = np.array([0, 0x10000, 1, 0x10001, 0, 0x10000], np.int32)
    losses = np.zeros(rows, np.float32)
    work = lg.copy()
    O.fused_classifier(work, losses, tg, V, mask=mask)
    skip = (mask & 0x10000) != 0
    assert np.all(losses[skip] == 0) and np.all(losses[~skip] > 0)
    assert np.array_equal(work[skip], lg[skip]) and not np.array_equal(work[~skip], lg[~skip])
    # write_dlogits = 0: loss only
    losses2 = np.zeros(rows, np.float32)
    work2 = lg.copy()
    O.fused_classifier(work2, losses2, tg, V, write_dlogits=False)
    assert np.array_equal(work2, lg) and np.array_equal(losses2[~skip], losses[~skip])


def test_activation_backward_vs_autograd():
    """oracle GELU / SwiGLU backward against torch autograd in fp64 on the same bf16 inputs (tolerance: one bf16 rounding of the result)"""
    import torch
    rng = np.random.default_rng(5)
    n = 4096
    x = O.f32_to_bf16(np.concatenate([rng.normal(0, 2.0, n - 6), [0.0, 12.0, -12.0, 30.0, -30.0, 1e-3]]).astype(np.float32))
    d = O.f32_to_bf16(rng.normal(0, 1.0, n).astype(np.float32))
    xt = torch.tensor(O.bf16_to_f32(x).astype(np.float64), requires_grad=True)
    y = torch.nn.functional.gelu(xt, approximate="tanh")
    y.backward(torch.tensor(O.bf16_to_f32(d).astype(np.float64)))
    got = O.bf16_to_f32(O.gelu_backward(d, x)).astype(np.float64)
    ref = xt.grad.numpy()
    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0 ** -8 + 1e-6)
    g = O.f32_to_bf16(rng.normal(0, 2.0, n).astype(np.float32))
    u = O.f32_to_bf16(rng.normal(0, 2.0, n).astype(np.float32))
    gt = torch.tensor(O.bf16_to_f32(g).astype(np.float64), requires_grad=True)
    ut = torch.tensor(O.bf16_to_f32(u).astype(np.float64), requires_grad=True)
    (torch.nn.functional.silu(gt) * ut).backward(torch.tensor(O.bf16_to_f32(d).astype(np.float64)))
    d_up, d_gate = O.swiglu_backward(d, g, u)
    for got_b, ref in ((d_up, ut.grad.numpy()), (d_gate, gt.grad.numpy())):
        got = O.bf16_to_f32(got_b).astype(np.float64)
        assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0 ** -8 + 1e-6)


@pytest.mark.parametrize("ln", [True, False])
def test_norm_backward_vs_autograd(ln):
    """oracle LayerNorm / RMSNorm backward against torch autograd in fp64 (same bf16 inputs, forward statistics from the fp64 forward)"""
    import torch
    rng = np.random.default_rng(17 + ln)
    rows, C_ = 37, 264
    x = O.f32_to_bf16(rng.normal(0.2, 1.5, (rows, C_)).astype(np.float32))
    w = O.f32_to_bf16((1 + rng.normal(0, 0.2, C_)).astype(np.float32))
    dout = O.f32_to_bf16(rng.normal(0, 1.0, (rows, C_)).astype(np.float32))
    xt = torch.tensor(O.bf16_to_f32(x).astype(np.float64), requires_grad=True)
    wt = torch.tensor(O.bf16_to_f32(w).astype(np.float64), requires_grad=True)
    bt = torch.zeros(C_, dtype=torch.float64, requires_grad=True)
    eps = 1e-5
    if ln:
        mean = xt.mean(dim=1, keepdim=True)
        rstd = 1.0 / torch.sqrt(((xt - mean) ** 2).mean(dim=1, keepdim=True) + eps)
        y = (xt - mean) * rstd * wt + bt
    else:
        mean = None
        rstd = 1.0 / torch.sqrt((xt ** 2).mean(dim=1, keepdim=True) + eps)
        y = xt * rstd * wt
    y.backward(torch.tensor(O.bf16_to_f32(dout).astype(np.float64)))
    dinp0 = O.f32_to_bf16(rng.normal(0, 0.5, (rows, C_)).astype(np.float32))
    dw0 = O.f32_to_bf16(rng.normal(0, 0.5, C_).astype(np.float32))
    db0 = O.f32_to_bf16(rng.normal(0, 0.5, C_).astype(np.float32))
    dinp, dw, db = dinp0.copy(), dw0.copy(), db0.copy()
    O.norm_backward(dinp, dw, db if ln else None, dout, x, w, mean.detach().numpy().ravel().astype(np.float32) if ln else None,
                    rstd.detach().numpy().ravel().astype(np.float32))
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    ref_dinp = f(dinp0) + xt.grad.numpy()
    assert np.all(np.abs(f(dinp) - ref_dinp) <= np.abs(ref_dinp) * 2.0 ** -8 + 2e-3 * np.abs(xt.grad.numpy()).max())
    ref_dw = f(dw0) + wt.grad.numpy()
    assert np.all(np.abs(f(dw) - ref_dw) <= np.abs(ref_dw) * 2.0 ** -8 + 1e-2)
    if ln:
        ref_db = f(db0) + bt.grad.numpy()
        assert np.all(np.abs(f(db) - ref_db) <= np.abs(ref_db) * 2.0 ** -8 + 1e-2)
    else:
        assert np.array_equal(db, db0)


def test_attn_backward_vs_autograd():
    import torch
    rng = np.random.default_rng(23)
    T, H, hd = 37, 3, 64
    mk = lambda: O.f32_to_bf16(rng.normal(0, 1.0, (T, H * hd)).astype(np.float32))
    q, k, v, dO = mk(), mk(), mk(), mk()
    f = lambda a: torch.tensor(O.bf16_to_f32(a).astype(np.float64)).reshape(T, H, hd).transpose(0, 1)   # [H, T, hd]
    qt, kt, vt = (f(a).clone().requires_grad_(True) for a in (q, k, v))
    ot = torch.nn.functional.scaled_dot_product_attention(qt, kt, vt, is_causal=True)
    ot.backward(f(dO))
    o = O.f32_to_bf16(ot.detach().transpose(0, 1).reshape(T, H * hd).numpy().astype(np.float32))
    dq, dk, dv = O.attn_backward(q, k, v, o, dO, H, hd)
    for got, ref_t in ((dq, qt.grad), (dk, kt.grad), (dv, vt.grad)):
        ref = ref_t.transpose(0, 1).reshape(T, H * hd).numpy()
        g = O.bf16_to_f32(got).astype(np.float64)
        assert np.abs(g - ref).max() <= 2.0 ** -7 * np.abs(ref).max()   # D uses the bf16-rounded o: a little beyond one bf16 rounding
