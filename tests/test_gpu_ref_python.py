"""The HIP kernels against outputs of the reference's own Python programs (tests/golden/ref_python_vectors.npz, see tests/test_oracle_ref_python.py):
AutoAWQ dequant, causal grouped-query attention (token-batch and decode forms), RMS normalisation -- through the C ABI."""
import os

import numpy as np
import pytest
import torch

from conftest import bf16_t, u16

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python_vectors.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_awq_dequant(ctx, O, gold):
    from koifish_amd.runtime import AWQDevWeight
    qw, qz, sc = gold["awq_qweight"], gold["awq_qzeros"], gold["awq_scales"]
    n_in, n_out = qw.shape[0], qw.shape[1] * 8
    dw = AWQDevWeight(n_out, n_in, torch.from_numpy(qw.copy()).to(ctx.device), torch.from_numpy(qz.copy()).to(ctx.device), torch.from_numpy(sc.view(np.int16).copy()).to(ctx.device))
    got, ref = O.bf16_to_f32(u16(ctx.dequant(dw))), O.bf16_to_f32(gold["awq_dequant"])
    assert np.abs(got - ref).max() <= 2.0 ** -7 * np.abs(ref).max() and (got != ref).mean() < 0.08


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_attention_prefill_and_decode(ctx, O, gold, tag):
    q, k, v, ref = gold["att_%s_q" % tag], gold["att_%s_k" % tag], gold["att_%s_v" % tag], gold["att_%s_out" % tag]
    T, HQ, D = q.shape
    HK = k.shape[1]
    qd, kvd = HQ * D, HK * D
    q_t, kc_t, vc_t = bf16_t(q.reshape(T, qd), ctx.device), bf16_t(k.reshape(T, kvd), ctx.device), bf16_t(v.reshape(T, kvd), ctx.device)
    out_t = torch.zeros(T, qd, dtype=torch.bfloat16, device=ctx.device)
    assert ctx.hip.kf_attn_prefill(ctx.h, q_t.data_ptr(), kc_t.data_ptr(), vc_t.data_ptr(), out_t.data_ptr(), 0, T, qd, HQ, HK, D, kvd) == 0
    scale = np.abs(ref).max()
    got = O.bf16_to_f32(u16(out_t)).reshape(T, HQ, D)
    assert np.abs(got - ref).max() <= 2.0 ** -6 * scale
    for t in (0, 1, T // 2, T - 1):
        o = O.bf16_to_f32(u16(ctx.attn_decode(q_t[t].contiguous(), kc_t, vc_t, t, HQ, HK, D))).reshape(HQ, D)
        assert np.abs(o - ref[t]).max() <= 2.0 ** -6 * scale, "t=%d" % t


def test_rmsnorm(ctx, O, gold):
    x, ref = gold["rms_x"], gold["rms_out"]
    ones = torch.ones(x.shape[1], dtype=torch.bfloat16, device=ctx.device)
    got = O.bf16_to_f32(u16(ctx.rmsnorm(bf16_t(x, ctx.device), ones, eps=1e-12)))
    assert np.abs(got - ref).max() <= 2.0 ** -8 * np.abs(ref).max()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_linear(ctx, O, gold, tag):
    """kf_linear on bf16 weights -- the mat-vec for one row, the token-batch kernels for 40 -- against the reference's own statement of the product (tl_gemm.py:150-151)"""
    import ctypes as C
    from koifish_amd import lib as L
    x, w, ref = gold["gemm_%s_x" % tag], gold["gemm_%s_w" % tag], gold["gemm_%s_out" % tag]
    n, (M, K) = x.shape[0], w.shape
    dw = ctx.upload_blob(L.BF16, M, K, w)
    d = dw.desc()
    xd = bf16_t(x, ctx.device)
    y = torch.zeros(n, M, dtype=torch.bfloat16, device=ctx.device)
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), xd.data_ptr(), y.data_ptr(), None, n, 1.0, 0.0, 0, None) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    got = O.bf16_to_f32(u16(y))
    assert np.abs(got - ref).max() <= 2.0 ** -8 * np.abs(ref).max() + 1e-6
    assert np.allclose(got, ref, rtol=1e-2, atol=1e-2)
