"""Row-codebook 4-bit storage (KF_QUANT_ROW_LUT; GeQuant::RT_NormalF, CU_Q42X_NF4 / CU_Q42X_lut, CU_embed_forw_q4 / _nf4) on the GPU,
through the C ABI, against the CPU oracle.

Bar: bit-exact for the quantiser bytes and tables, the dequantised matrix and embedding rows (byte / index work); mat-vecs within 1 bf16 ulp of
the oracle's and within 2^-8 of the fp64 product's scale (fp32 sums in a different order); whole decode steps and prefill of a model stored this
way: greedy ids identical, logits within 2^-6 of max|logit|.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import bf16_t, u16, ulp_diff_bf16
from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth

pytestmark = pytest.mark.gpu
MISMATCH_FRAC = 2e-3
LOGIT_TOL = 2.0 ** -6


def rand_w(O, rng, m, k, std=0.02):
    return O.f32_to_bf16(rng.normal(0, std, size=(m, k)).astype(np.float32))


@pytest.mark.parametrize("shape,std", [((64, 256), 0.02), ((8, 3072), 1.0), ((104, 32), 0.3), ((2048, 1024), 0.02)])
def test_quantizer_bytes_and_tables_match_oracle(ctx, O, shape, std):
    rng = np.random.default_rng(21)
    m, k = shape
    w = rand_w(O, rng, m, k, std)
    w.reshape(m, k)[3] = 0                       # an all-zero row: scale falls back to 1
    ow = O.quantize_nf4(w, m, k)
    dw = ctx.quantize_nf4(bf16_t(w, ctx.device).view(m, k))
    blob = dw.blob.cpu().numpy()
    assert np.array_equal(blob[:dw.szData], ow.data), "nibble stream differs"
    assert np.array_equal(u16(dw.lut()), ow.lut), "row tables differ"
    assert np.array_equal(blob, ow.blob()), "data||gama blob differs"
    assert np.array_equal(u16(ctx.dequant(dw)), O.dequant(ow))


def test_dequant_and_embed_bit_exact(ctx, O):
    rng = np.random.default_rng(22)
    m, k = 200, 512
    ow = O.quantize_nf4(rand_w(O, rng, m, k), m, k)
    dw = ctx.upload_lut_blob(m, k, ow.blob())
    ref = O.dequant(ow)
    assert np.array_equal(u16(ctx.dequant(dw)), ref)
    for tok in (0, 7, m - 1):
        assert np.array_equal(u16(ctx.embed(dw, tok)), ref[tok])
        assert np.array_equal(u16(ctx.embed(dw, tok)), O.embed(ow, tok))
    # token batch (ids from device memory; an id outside the table reads row 0, as every embed kernel here does) and the decode-state form
    ids = torch.tensor([5, 199, 0, 42, 100000], dtype=torch.int32, device=ctx.device)
    out = torch.empty(5, k, dtype=torch.bfloat16, device=ctx.device)
    d = dw.desc()
    L.check(ctx.hip.kf_embed_batch(ctx.h, C.byref(d), C.c_void_p(ids.data_ptr()), 5, C.c_void_p(out.data_ptr())), "kf_embed_batch")
    assert np.array_equal(u16(out), ref[[5, 199, 0, 42, 0]])
    state = torch.tensor([17, 2], dtype=torch.int32, device=ctx.device)
    forced = torch.tensor([-1, -1, 33, -1], dtype=torch.int32, device=ctx.device)
    row = torch.empty(k, dtype=torch.bfloat16, device=ctx.device)
    L.check(ctx.hip.kf_embed_state(ctx.h, C.byref(d), C.c_void_p(state.data_ptr()), None, C.c_void_p(row.data_ptr())), "kf_embed_state")
    assert np.array_equal(u16(row), ref[17])
    L.check(ctx.hip.kf_embed_state(ctx.h, C.byref(d), C.c_void_p(state.data_ptr()), C.c_void_p(forced.data_ptr()), C.c_void_p(row.data_ptr())), "kf_embed_state")
    assert np.array_equal(u16(row), ref[33])


@pytest.mark.parametrize("shape", [(2048, 1024), (1024, 3072), (512, 128), (40, 3200), (24, 5120), (8, 32), (151936 // 8, 1024)])
def test_linear_vs_oracle(ctx, O, shape):
    rng = np.random.default_rng(hash(shape) & 0xFFFF)
    m, k = shape
    ow = O.quantize_nf4(rand_w(O, rng, m, k), m, k)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    dw = ctx.upload_lut_blob(m, k, ow.blob())
    y = u16(ctx.linear(dw, bf16_t(x, ctx.device)))
    ref = O.linear(ow, x)
    d = ulp_diff_bf16(y, ref)
    assert d.max() <= 1, "max ulp %d" % d.max()
    assert (d > 0).mean() <= MISMATCH_FRAC or m < 1000, "mismatch fraction %g" % (d > 0).mean()
    exact = O.bf16_to_f32(O.dequant(ow)).astype(np.float64) @ O.bf16_to_f32(x).astype(np.float64)
    assert np.abs(O.bf16_to_f32(y) - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 1e-6


def test_linear_epilogues_and_fused_forms(ctx, O):
    rng = np.random.default_rng(23)
    m, k = 256, 1024
    ow = O.quantize_nf4(rand_w(O, rng, m, k), m, k)
    ow2 = O.quantize_nf4(rand_w(O, rng, m, k), m, k)
    dw, dw2 = ctx.upload_lut_blob(m, k, ow.blob()), ctx.upload_lut_blob(m, k, ow2.blob())
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    b = O.f32_to_bf16(rng.normal(0, 0.1, size=m).astype(np.float32))
    y0 = O.f32_to_bf16(rng.normal(0, 0.5, size=m).astype(np.float32))
    res = O.f32_to_bf16(rng.normal(0, 0.5, size=m).astype(np.float32))
    nw = O.f32_to_bf16((1 + 0.1 * rng.normal(size=k)).astype(np.float32))
    xt = bf16_t(x, ctx.device)
    y = ctx.linear(dw, xt, bias=bf16_t(b, ctx.device), alpha=0.5, beta=2.0, y=bf16_t(y0, ctx.device).clone())
    assert ulp_diff_bf16(u16(y), O.linear(ow, x, bias=b, alpha=0.5, beta=2.0, y=y0)).max() <= 1
    y = ctx.linear(dw, xt, residual=bf16_t(res, ctx.device))
    assert ulp_diff_bf16(u16(y), O.add(res, O.linear(ow, x))).max() <= 1
    # RMSNorm prologue + two matrices in one launch; the paired SwiGLU form
    xn = O.rmsnorm(x, nw)
    ys = ctx.norm_linear(xt, bf16_t(nw, ctx.device), [dw, dw2])
    assert ulp_diff_bf16(u16(ys[0]), O.linear(ow, xn)).max() <= 1 and ulp_diff_bf16(u16(ys[1]), O.linear(ow2, xn)).max() <= 1
    act = ctx.norm_gateup_swiglu(xt, bf16_t(nw, ctx.device), dw, dw2)
    ref = O.swiglu(O.linear(ow, xn), O.linear(ow2, xn))
    assert np.abs(O.bf16_to_f32(u16(act)) - O.bf16_to_f32(ref)).max() <= 2.0 ** -7 * np.abs(O.bf16_to_f32(ref)).max()
    # LM head form: logits + the first maximum
    logits, am = ctx.lm_head(dw, xt)
    assert ulp_diff_bf16(u16(logits), O.linear(ow, x)).max() <= 1
    assert am == O.argmax_bf16(u16(logits))
    # mixing a row-LUT matrix with a Packed128 one in one launch is refused
    og = O.quantize(rand_w(O, rng, m, k), m, k, L.Q4)
    dg = ctx.upload_blob(L.Q4, m, k, og.blob())
    with pytest.raises(L.KFError):
        ctx.norm_linear(xt, None, [dw, dg])


@pytest.mark.parametrize("n_tok,m,k", [(3, 384, 512), (40, 384, 512), (300, 384, 512), (1100, 256, 1024), (64, 128, 1600), (2100, 128, 512)])
def test_linear_token_batches(ctx, O, n_tok, m, k):
    """nTok > 1: the MFMA tile kernels with the nibble stream unpacked in registers through the row's table (mat-vec loop below 8 rows, the
    hand-written tile kernels whatever the row count) -- checked against the exact fp64 product of the dequantised weights"""
    rng = np.random.default_rng(24)
    ow = O.quantize_nf4(rand_w(O, rng, m, k), m, k)
    dw = ctx.upload_lut_blob(m, k, ow.blob())
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(n_tok, k)).astype(np.float32))
    bias = O.f32_to_bf16(rng.normal(0, 0.1, size=m).astype(np.float32))
    xt, bt = bf16_t(x, ctx.device), bf16_t(bias, ctx.device)
    y = torch.zeros(n_tok, m, dtype=torch.bfloat16, device=ctx.device)
    d = dw.desc()
    ctx.linear_scratch(dw, n_tok)   # shapes the tile kernels do not cover dequantise into caller-owned scratch
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), C.c_void_p(xt.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(bt.data_ptr()), n_tok, 1.0, 0.0, 0, None), "kf_linear")
    exact = O.bf16_to_f32(x).astype(np.float64) @ O.bf16_to_f32(O.dequant(ow)).astype(np.float64).T + O.bf16_to_f32(bias).astype(np.float64)
    tol = 2.0 ** -8
    assert np.abs(O.bf16_to_f32(u16(y)) - exact).max() <= tol * np.abs(exact).max() + 1e-6


def test_rejects_malformed(ctx, O):
    rng = np.random.default_rng(25)
    ow = O.quantize_nf4(rand_w(O, rng, 16, 256), 16, 256)
    dw = ctx.upload_lut_blob(16, 256, ow.blob())
    x = torch.zeros(256, dtype=torch.bfloat16, device=ctx.device)
    y = torch.zeros(16, dtype=torch.bfloat16, device=ctx.device)
    d = dw.desc()
    d.type = L.T_SIGN                       # the row codebook exists for 4-bit only
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 0, None) == -701
    d = dw.desc()
    d.quant = 7
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 0, None) == -1000
    d = dw.desc()
    d.ne1 = 48                              # rows must be whole 16-byte blocks
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 0, None) == -2000
    d = dw.desc()
    d.gama = None
    assert ctx.hip.kf_dequant(ctx.h, C.byref(d), y.data_ptr()) == -701


@pytest.mark.parametrize("head_type", [L.BF16, L.NF4])
def test_model_decode_and_prefill_with_nf4_layers(O, head_type):
    """a whole Qwen3 (tiny) stored in the row-codebook form: teacher-forced decode steps, then a batched prefill, against the oracle"""
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 4321, w_std=0.02)
    gm = synth.build_from_raw(cfg, raw, L.NF4, head_type)
    om = oracle_model(cfg, raw, L.NF4, head_type)
    prompt = prompt_ids(cfg, 24)
    tok = int(prompt[0])
    for pos in range(24):
        g_next, g_logits = gm.forward(tok, pos)
        o_next, o_logits, _ = om.decode(tok, pos)
        gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
        assert np.abs(gl - ol).max() <= LOGIT_TOL * np.abs(ol).max(), "step %d" % pos
        assert g_next == O.argmax_bf16(g_logits) and g_next == o_next, "step %d: greedy id %d vs oracle %d" % (pos, g_next, o_next)
        tok = int(prompt[pos + 1]) if pos + 1 < len(prompt) else o_next
    ids_graph = gm.generate(prompt, 12, use_graph=True)
    ids_eager = gm.generate(prompt, 12, use_graph=False)
    assert ids_graph == ids_eager
    g_next, g_logits = gm.prefill(prompt)
    assert np.abs(O.bf16_to_f32(g_logits) - O.bf16_to_f32(o_logits)).max() <= LOGIT_TOL * np.abs(O.bf16_to_f32(o_logits)).max()
    assert g_next == o_next
    gm.close()


# ---- 3- / 2-bit row forms (CU_Q32X_NF3 / CU_Q32X_ / CU_Q22X_ / CU_Q22X_RTN): dequant-only storage, kf_linear = GetDataX + the bf16 product
@pytest.mark.parametrize("shape,std", [((64, 256), 0.02), ((9, 3072), 1.0), ((2048, 1024), 0.02)])
def test_nf3_quantizer_and_dequant_bit_exact(ctx, O, shape, std):
    rng = np.random.default_rng(31)
    m, k = shape
    w = rand_w(O, rng, m, k, std)
    w.reshape(m, k)[2] = 0
    ow = O.quantize_nf3(w, m, k)
    dw = ctx.quantize_nf4(bf16_t(w, ctx.device).view(m, k), bits=3)
    assert np.array_equal(dw.blob.cpu().numpy(), np.concatenate([ow.data, np.full(m + k, 0x3F80, np.uint16).view(np.uint8), ow.lut.reshape(-1).view(np.uint8)]))
    assert np.array_equal(u16(ctx.dequant(dw)), O.dequant(ow))


def _two_bit(O, rng, m, k):
    ids = rng.integers(0, 4, size=(m, k)).astype(np.uint8)
    data = np.packbits(((ids[..., None] >> np.array([1, 0])) & 1).astype(np.uint8).reshape(-1))
    lut = O.f32_to_bf16(rng.normal(0, 0.1, size=(m, 4)).astype(np.float32))
    zs = O.f32_to_bf16(np.stack([rng.normal(-0.1, 0.02, size=m), rng.uniform(0.03, 0.08, size=m)], axis=1).astype(np.float32))
    return O.LutWeight(m, k, data, lut, bits=2), O.LutWeight(m, k, data, zs, bits=2, rtn=True)


def _blob(ow, m, k):
    return np.concatenate([ow.data, np.full(m + k, 0x3F80, np.uint16).view(np.uint8), ow.lut.reshape(-1).view(np.uint8)])


def test_two_bit_row_forms_dequant_bit_exact(ctx, O):
    rng = np.random.default_rng(32)
    m, k = 40, 512
    q, r = _two_bit(O, rng, m, k)
    assert np.array_equal(u16(ctx.dequant(ctx.upload_lut_blob(m, k, _blob(q, m, k), bits=2))), O.dequant(q))
    assert np.array_equal(u16(ctx.dequant(ctx.upload_lut_blob(m, k, _blob(r, m, k), bits=2, rtn=True))), O.dequant(r))


@pytest.mark.parametrize("n_tok", [1, 5, 64])
def test_row_forms_linear_through_dequant(ctx, O, n_tok):
    rng = np.random.default_rng(33)
    m, k = 256, 512
    o3 = O.quantize_nf3(rand_w(O, rng, m, k), m, k)
    o2, o2r = _two_bit(O, rng, m, k)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(n_tok, k)).astype(np.float32))
    bias = O.f32_to_bf16(rng.normal(0, 0.1, size=m).astype(np.float32))
    xt, bt = bf16_t(x, ctx.device), bf16_t(bias, ctx.device)
    for ow, dw in ((o3, ctx.upload_lut_blob(m, k, _blob(o3, m, k), bits=3)), (o2, ctx.upload_lut_blob(m, k, _blob(o2, m, k), bits=2)),
                   (o2r, ctx.upload_lut_blob(m, k, _blob(o2r, m, k), bits=2, rtn=True))):
        y = torch.zeros(n_tok, m, dtype=torch.bfloat16, device=ctx.device)
        d = dw.desc()
        ctx.linear_scratch(dw, n_tok)
        L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), C.c_void_p(xt.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(bt.data_ptr()), n_tok, 1.0, 0.0, 0, None), "kf_linear")
        W = O.bf16_to_f32(O.dequant(ow)).astype(np.float64)
        exact = O.bf16_to_f32(x).astype(np.float64) @ W.T + O.bf16_to_f32(bias).astype(np.float64)
        assert np.abs(O.bf16_to_f32(u16(y)) - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 1e-6
        if n_tok == 1:   # the mat-vec over the dequantised copy is the bf16 mat-vec: within 1 ulp of the oracle's
            assert ulp_diff_bf16(u16(y)[0], O.linear(ow, x[0], bias=bias)).max() <= 1


def test_row_forms_refusals(ctx, O):
    rng = np.random.default_rng(34)
    m, k = 16, 64
    o3 = O.quantize_nf3(rand_w(O, rng, m, k), m, k)
    d3 = ctx.upload_lut_blob(m, k, _blob(o3, m, k), bits=3)
    out = torch.zeros(k, dtype=torch.bfloat16, device=ctx.device)
    d = d3.desc()
    assert ctx.hip.kf_embed(ctx.h, C.byref(d), 1, None, C.c_void_p(out.data_ptr())) == -1000      # TokenEmbed::cuInfer: Q3 -> assert(0)
    x = torch.zeros(k, dtype=torch.bfloat16, device=ctx.device)
    with pytest.raises(L.KFError):
        ctx.norm_linear(x, None, [d3])                                                             # no in-place mat-vec for the 3-bit stream
    d = d3.desc()
    d.quant = L.QUANT_ROW_RTN                                                                      # CU_Q32X_RTN is not restated (reads row 0's stream for every row)
    assert ctx.hip.kf_dequant(ctx.h, C.byref(d), C.c_void_p(torch.zeros(m * k, dtype=torch.bfloat16, device=ctx.device).data_ptr())) == -701
    q2, _ = _two_bit(O, rng, m, k)
    d = ctx.upload_lut_blob(m, k, _blob(q2, m, k), bits=2).desc()
    src = torch.zeros(m, k, dtype=torch.bfloat16, device=ctx.device)
    assert ctx.hip.kf_quantize(ctx.h, C.byref(d), C.c_void_p(src.data_ptr()), 0) == -701          # RT_NormalF asserts bits == 4 || 3


def test_golden_normal_float_fixture_on_gpu(ctx, O):
    """tests/golden/nf4_linear.npz (made by the oracle, committed): the device quantiser must reproduce its streams and tables, the mat-vec its outputs"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nf4_linear.npz"))
    m, k = int(g["m"]), int(g["k"])
    wt = bf16_t(g["w"], ctx.device).view(m, k)
    for bits in (4, 3):
        dw = ctx.quantize_nf4(wt, bits=bits)
        assert np.array_equal(dw.blob[:dw.szData].cpu().numpy(), g["packed%d" % bits])
        assert np.array_equal(u16(dw.lut()), g["lut%d" % bits])
        assert np.array_equal(u16(ctx.dequant(dw)).reshape(-1), g["dequant%d" % bits])
        y = torch.zeros(m, dtype=torch.bfloat16, device=ctx.device)
        xt = bf16_t(g["x"], ctx.device)
        d = dw.desc()
        ctx.linear_scratch(dw, 1)
        L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), C.c_void_p(xt.data_ptr()), C.c_void_p(y.data_ptr()), None, 1, 1.0, 0.0, 0, None), "kf_linear")
        assert ulp_diff_bf16(u16(y), g["y%d" % bits]).max() <= 1
