"""Edge cases of the ABI on the GPU: ragged shapes, the last cache position, first token, odd group counts, aliasing outputs,
null/unaligned arguments, determinism under repetition."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import bf16_t, u16, ulp_diff_bf16, close_bf16
from koifish_amd import lib as L
from koifish_amd import synth
from helpers import oracle_model, prompt_ids
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rand_w(rng, m, k, std=0.02):
    return O.f32_to_bf16(rng.normal(0, std, size=(m, k)).astype(np.float32))


@pytest.mark.parametrize("t", [L.Q4, L.BF16, L.F8E5M2, L.T_SIGN, L.BOOL1])
@pytest.mark.parametrize("shape", [(1, 128), (3, 256), (7, 384), (129, 640), (2, 12800), (1000, 1152), (33, 3200)])
def test_linear_ragged_shapes(ctx, t, shape):
    """row counts that are not multiples of the rows-per-wave, K that is not a power of two, a single row, a single group"""
    m, k = shape
    rng = np.random.default_rng(m * 7 + k)
    w = rand_w(rng, m, k)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    ow = O.quantize(w, m, k, t)
    dw = ctx.upload_blob(t, m, k, ow.blob())
    y = u16(ctx.linear(dw, bf16_t(x, ctx.device)))
    assert close_bf16(y, O.linear(ow, x)).all()
    assert np.array_equal(u16(ctx.dequant(dw)), O.dequant(ow))


def test_outputs_do_not_spill_past_their_rows(ctx):
    """guard bytes around y stay untouched for a row count that leaves masked lanes"""
    rng = np.random.default_rng(5)
    m, k = 37, 1024
    ow = O.quantize(rand_w(rng, m, k), m, k, L.Q4)
    dw = ctx.upload_blob(L.Q4, m, k, ow.blob())
    buf = torch.full((m + 64,), 7.0, dtype=torch.bfloat16, device=ctx.device)
    x = bf16_t(O.f32_to_bf16(rng.normal(0, 1, size=k).astype(np.float32)), ctx.device)
    ctx.linear(dw, x, y=buf[32:32 + m])
    assert bool((buf[:32] == 7.0).all()) and bool((buf[32 + m:] == 7.0).all())


def test_kernels_are_deterministic(ctx):
    rng = np.random.default_rng(6)
    m, k = 3072, 1024
    ow = O.quantize(rand_w(rng, m, k), m, k, L.Q4)
    dw = ctx.upload_blob(L.Q4, m, k, ow.blob())
    x = bf16_t(O.f32_to_bf16(rng.normal(0, 1, size=k).astype(np.float32)), ctx.device)
    ys = [ctx.linear(dw, x).clone() for _ in range(5)]
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    nh, nkv, hd, pos = 16, 8, 128, 777
    q = torch.randn(nh * hd, device=ctx.device).to(torch.bfloat16)
    kc = torch.randn(pos + 1, nkv * hd, device=ctx.device).to(torch.bfloat16)
    vc = torch.randn(pos + 1, nkv * hd, device=ctx.device).to(torch.bfloat16)
    outs = [ctx.attn_decode(q, kc, vc, pos, nh, nkv, hd).clone() for _ in range(8)]
    assert all(torch.equal(outs[0], o) for o in outs[1:]), "split-KV merge order must not depend on arrival order"


def test_decode_to_the_last_cache_position():
    """max_seq - 1 is a legal position; max_seq is refused; position 0 (single key) works"""
    cfg = dict(synth.CONFIGS["tiny"], max_seq=70)
    raw = synth.raw_weights_numpy(cfg, 8, w_std=0.1)
    gm = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    om = oracle_model(cfg, raw, L.Q4, L.BF16)
    prompt = prompt_ids(cfg, 60, seed=1)
    ref = om.generate(prompt.tolist(), 10)        # positions 0..68, last fed position = 68 = max_seq - 2 ... then one more
    assert gm.generate(prompt, 10) == ref
    nxt_g, _ = gm.forward(ref[-1], 69)            # the very last row of the cache
    nxt_o, _, _ = om.decode(ref[-1], 69)
    assert nxt_g == nxt_o
    with pytest.raises(L.KFError):
        gm.forward(1, 70)
    with pytest.raises(L.KFError):
        gm.forward(cfg["vocab"], 0)
    with pytest.raises(L.KFError):
        gm.generate(prompt, 12)                   # 60 + 12 - 1 > 70 positions
    gm.close()


def test_null_and_misaligned_arguments(ctx):
    hip = ctx.hip
    x = torch.zeros(256, dtype=torch.bfloat16, device=ctx.device)
    assert hip.kf_swiglu(ctx.h, None, x.data_ptr(), x.data_ptr(), 256) == -20
    assert hip.kf_add(ctx.h, x.data_ptr(), x.data_ptr(), x.data_ptr(), 0) == -20
    assert hip.kf_attn_decode(ctx.h, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, None, 2, 1, 128, 128, None) == -20
    assert hip.kf_attn_decode(ctx.h, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, None, 3, 2, 128, 256, ctx._ws(4, 128).data_ptr()) == -20
    assert hip.kf_attn_decode(ctx.h, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, None, 2, 1, 96, 96, ctx._ws(4, 128).data_ptr()) == -20
    assert hip.kf_sync(None) == -20 and hip.kf_destroy(None) == 0
    assert hip.kf_tp_reduce(ctx.h, None, 2, 8, None, x.data_ptr()) == -20
    rng = np.random.default_rng(1)
    ow = O.quantize(rand_w(rng, 16, 256), 16, 256, L.Q4)
    dw = ctx.upload_blob(L.Q4, 16, 256, ow.blob())
    d = dw.desc()
    d.lGroup = 96      # group size that does not divide the tensor
    y = torch.zeros(16, dtype=torch.bfloat16, device=ctx.device)
    assert hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 0, None) == -701
    d = dw.desc()
    d.gama = None
    assert hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 0, None) == -701
    assert hip.kf_linear(ctx.h, C.byref(dw.desc()), x.data_ptr() + 2, y.data_ptr(), None, 1, 1.0, 0.0, 0, None) == -2000
    assert hip.kf_linear(ctx.h, C.byref(dw.desc()), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, L.KF_EPI_RESIDUAL, None) == -20


def test_tp_reduce_rank_order(ctx):
    rng = np.random.default_rng(2)
    R, n = 8, 5120
    p = rng.normal(0, 1, size=(R, n)).astype(np.float32)
    res = O.f32_to_bf16(rng.normal(0, 1, size=n).astype(np.float32))
    tot = p[0].copy()
    for r in range(1, R):
        tot = (tot + p[r]).astype(np.float32)
    ref = O.add(res, O.f32_to_bf16(tot))
    out = torch.zeros(n, dtype=torch.bfloat16, device=ctx.device)
    pd = torch.from_numpy(p).to(ctx.device)
    res_d = bf16_t(res, ctx.device)   # named: a temporary is freed as soon as data_ptr() returns
    assert ctx.hip.kf_tp_reduce(ctx.h, pd.data_ptr(), R, n, res_d.data_ptr(), out.data_ptr()) == 0
    assert np.array_equal(u16(out), ref)
    y = torch.zeros(64, dtype=torch.float32, device=ctx.device)
    ow = O.quantize(rand_w(rng, 64, 512), 64, 512, L.Q4)
    dw = ctx.upload_blob(L.Q4, 64, 512, ow.blob())
    x = O.f32_to_bf16(rng.normal(0, 1, size=512).astype(np.float32))
    x_d, desc_ = bf16_t(x, ctx.device), dw.desc()
    assert ctx.hip.kf_linear_f32(ctx.h, C.byref(desc_), x_d.data_ptr(), y.data_ptr()) == 0
    ref32 = O.linear_f32(ow, x)
    assert np.abs(y.cpu().numpy() - ref32).max() <= 1e-5 * np.abs(ref32).max() + 1e-6
