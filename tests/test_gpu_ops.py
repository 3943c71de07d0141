"""Per-operator parity: HIP kernels through the C ABI vs the CPU oracle on the same seeded inputs.

Bar: bit-exact for integer / byte / index work (pack layout, quantiser bytes, dequant, embedding gather, arg-max)
and for the ops whose fp arithmetic is order-independent by construction (RMSNorm with fp64 sum of squares,
q/k-norm + RoPE with the host-built table, SwiGLU with the fixed exp, residual add).  Mat-vec and attention
accumulate in fp32 in a different order than the oracle: outputs must agree to <= 1 bf16 ulp, with at most
MISMATCH_FRAC of the elements differing at all.
"""
import numpy as np
import pytest
import torch

from conftest import bf16_t, u16, ulp_diff_bf16, close_bf16
from koifish_amd import lib as L

pytestmark = pytest.mark.gpu
MISMATCH_FRAC = 2e-3

TYPES = [L.BF16, L.F8E5M2, L.Q4, L.T_SIGN, L.BOOL1]


def rand_w(rng, m, k, std=0.02):
    from oracle import oracle as O
    return O.f32_to_bf16(rng.normal(0, std, size=(m, k)).astype(np.float32))


def oracle_weight(O, w_u16, m, k, t, symmetric=False):
    return O.quantize(w_u16, m, k, t, symmetric=symmetric)


@pytest.mark.parametrize("t", [L.Q4, L.T_SIGN, L.BOOL1])
@pytest.mark.parametrize("shape", [(64, 256), (96, 1024), (8, 3072), (5, 384)])   # (5, 384): 15 groups -- the last wave of the 16-lanes-per-group 4-bit kernel is partly idle
def test_quantizer_bytes_match_oracle(ctx, O, t, shape):
    rng = np.random.default_rng(11)
    m, k = shape
    w = rand_w(rng, m, k)
    ow = oracle_weight(O, w, m, k, t)
    dw = ctx.quantize(bf16_t(w, ctx.device), t)
    blob = dw.blob.cpu().numpy()
    assert np.array_equal(blob[:dw.szData], ow.data.view(np.uint8)), "packed stream differs"
    z, s = dw.zero_step()
    assert np.array_equal(u16(z), ow.zero) and np.array_equal(u16(s), ow.step)


@pytest.mark.parametrize("t,lg", [(L.Q4, 32), (L.Q4, 96), (L.Q4, 256), (L.Q4, 1024), (L.T_SIGN, 64), (L.T_SIGN, 192), (L.BOOL1, 384)])
def test_quantizer_group_sizes(ctx, O, t, lg):
    """group lengths other than 128 (any multiple of a Packed128 block's 128 / bits elements): the lanes past a short group, and groups several wave passes long"""
    rng = np.random.default_rng(13 + lg)
    m, k = 24, 3072
    w = rand_w(rng, m, k)
    ow = O.quantize(w, m, k, t, lGroup=lg)
    dw = ctx.quantize(bf16_t(w, ctx.device), t, lGroup=lg)
    assert np.array_equal(dw.blob.cpu().numpy()[:dw.szData], ow.data.view(np.uint8)), "packed stream differs"
    z, s_ = dw.zero_step()
    assert np.array_equal(u16(z), ow.zero) and np.array_equal(u16(s_), ow.step)


def test_quantizer_symmetric_q4(ctx, O):
    rng = np.random.default_rng(12)
    w = rand_w(rng, 32, 512)
    ow = oracle_weight(O, w, 32, 512, L.Q4, symmetric=True)
    dw = ctx.quantize(bf16_t(w, ctx.device), L.Q4, symmetric=True)
    assert np.array_equal(dw.blob.cpu().numpy()[:dw.szData], ow.data.view(np.uint8))
    assert dw.qBias == ow.qBias == 8
    assert np.array_equal(u16(ctx.dequant(dw)), O.dequant(ow))


@pytest.mark.parametrize("t", TYPES)
def test_dequant_bit_exact(ctx, O, t):
    rng = np.random.default_rng(5)
    m, k = 48, 512
    w = rand_w(rng, m, k)
    ow = oracle_weight(O, w, m, k, t)
    dw = ctx.upload_blob(t, m, k, ow.blob())
    assert np.array_equal(u16(ctx.dequant(dw)), O.dequant(ow))


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("shape", [(2048, 1024), (1024, 2048), (1024, 3072), (512, 128), (40, 3200), (24, 5120)])
def test_linear_vs_oracle(ctx, O, t, shape):
    rng = np.random.default_rng(hash((t, shape)) & 0xFFFF)
    m, k = shape
    if k % 128:
        pytest.skip("group size 128 must divide K")
    w = rand_w(rng, m, k)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    ow = oracle_weight(O, w, m, k, t)
    dw = ctx.upload_blob(t, m, k, ow.blob())
    y = u16(ctx.linear(dw, bf16_t(x, ctx.device)))
    ref = O.linear(ow, x)
    d = ulp_diff_bf16(y, ref)
    assert d.max() <= 1, "max ulp %d" % d.max()
    assert (d > 0).mean() <= MISMATCH_FRAC, "mismatch fraction %g" % (d > 0).mean()
    # against the exact (fp64) product of the dequantised weights: a tolerance that does not depend on any order
    exact = O.bf16_to_f32(O.dequant(ow)).astype(np.float64) @ O.bf16_to_f32(x).astype(np.float64)
    assert np.abs(O.bf16_to_f32(y) - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 1e-6


def test_linear_alpha_beta_bias_residual(ctx, O):
    rng = np.random.default_rng(9)
    m, k = 256, 1024
    w = rand_w(rng, m, k)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    b = O.f32_to_bf16(rng.normal(0, 0.1, size=m).astype(np.float32))
    y0 = O.f32_to_bf16(rng.normal(0, 0.5, size=m).astype(np.float32))
    res = O.f32_to_bf16(rng.normal(0, 0.5, size=m).astype(np.float32))
    ow = oracle_weight(O, w, m, k, L.Q4)
    dw = ctx.upload_blob(L.Q4, m, k, ow.blob())
    y = ctx.linear(dw, bf16_t(x, ctx.device), bias=bf16_t(b, ctx.device), alpha=0.5, beta=2.0, y=bf16_t(y0, ctx.device).clone())
    ref = O.linear(ow, x, bias=b, alpha=0.5, beta=2.0, y=y0)
    assert ulp_diff_bf16(u16(y), ref).max() <= 1
    y = ctx.linear(dw, bf16_t(x, ctx.device), residual=bf16_t(res, ctx.device))
    ref = O.add(res, O.linear(ow, x))
    assert ulp_diff_bf16(u16(y), ref).max() <= 1


def test_linear_rejects_bad_args(ctx, O):
    import ctypes as C
    rng = np.random.default_rng(1)
    w = rand_w(rng, 16, 256)
    ow = oracle_weight(O, w, 16, 256, L.Q4)
    dw = ctx.upload_blob(L.Q4, 16, 256, ow.blob())
    x = torch.zeros(256, dtype=torch.bfloat16, device=ctx.device)
    y = torch.zeros(16, dtype=torch.bfloat16, device=ctx.device)
    d = dw.desc()
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 0, 1.0, 0.0, 0, None) == -20  # nTok < 1
    d.type = L.Q3
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 0, None) == -1000
    d = dw.desc()
    d.data = d.data + 2
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 0, None) == -2000
    assert b"aligned" in ctx.hip.kf_last_error()


@pytest.mark.parametrize("dim", [128, 1024, 5120])
def test_rmsnorm_bit_exact(ctx, O, dim):
    rng = np.random.default_rng(dim)
    x = O.f32_to_bf16(rng.normal(0, 2.0, size=(3, dim)).astype(np.float32))
    w = O.f32_to_bf16((1 + rng.normal(0, 0.1, size=dim)).astype(np.float32))
    y = ctx.rmsnorm(bf16_t(x, ctx.device), bf16_t(w, ctx.device), 1e-6)
    assert np.array_equal(u16(y), O.rmsnorm(x, w, 1e-6))


def test_rmsnorm_odd_dim_refused(ctx):
    x = torch.zeros(7, dtype=torch.bfloat16, device=ctx.device)
    assert ctx.hip.kf_rmsnorm(ctx.h, x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 7, 1e-6, None) == -2100


@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("pos", [0, 1, 77, 2047])
def test_qknorm_rope_bit_exact(ctx, O, hd, pos):
    rng = np.random.default_rng(pos + hd)
    nh, nkv = 6, 2
    q = O.f32_to_bf16(rng.normal(0, 1.0, size=nh * hd).astype(np.float32))
    k = O.f32_to_bf16(rng.normal(0, 1.0, size=nkv * hd).astype(np.float32))
    wq = O.f32_to_bf16((1 + rng.normal(0, 0.1, size=hd)).astype(np.float32))
    wk = O.f32_to_bf16((1 + rng.normal(0, 0.1, size=hd)).astype(np.float32))
    table = ctx.rope_table(2048, hd, 1e6)
    c, s = O.rope_table(pos, hd, 1e6)
    assert np.array_equal(table[pos, :, 0].cpu().numpy(), c) and np.array_equal(table[pos, :, 1].cpu().numpy(), s)
    qd, kd = bf16_t(q, ctx.device).clone(), bf16_t(k, ctx.device).clone()
    ctx.qknorm_rope(qd, kd, bf16_t(wq, ctx.device), bf16_t(wk, ctx.device), table, pos, nh, nkv, hd, 1e-6)
    assert np.array_equal(u16(qd), O.rope(O.headnorm(q, wq, nh, hd, 1e-6), nh, hd, pos, 1e6))
    assert np.array_equal(u16(kd), O.rope(O.headnorm(k, wk, nkv, hd, 1e-6), nkv, hd, pos, 1e6))


def test_swiglu_add_bit_exact(ctx, O):
    rng = np.random.default_rng(3)
    g = O.f32_to_bf16(np.concatenate([rng.normal(0, 3.0, size=3000), [-100.0, 100.0, 0.0, -0.0, 88.0, -88.0, 1e-30, -1e30]]).astype(np.float32))
    u = O.f32_to_bf16(rng.normal(0, 3.0, size=g.size).astype(np.float32))
    assert np.array_equal(u16(ctx.swiglu(bf16_t(g, ctx.device), bf16_t(u, ctx.device))), O.swiglu(g, u))
    assert np.array_equal(u16(ctx.add(bf16_t(g, ctx.device), bf16_t(u, ctx.device))), O.add(g, u))


@pytest.mark.parametrize("t", TYPES)
def test_embed_bit_exact(ctx, O, t):
    rng = np.random.default_rng(4)
    v, d = 300, 256
    w = rand_w(rng, v, d)
    ow = oracle_weight(O, w, v, d, t)
    dw = ctx.upload_blob(t, v, d, ow.blob())
    for tok in (0, 17, 299):
        assert np.array_equal(u16(ctx.embed(dw, tok)), O.embed(ow, tok))
    import ctypes as C
    dd = dw.desc()
    out = torch.zeros(d, dtype=torch.bfloat16, device=ctx.device)
    assert ctx.hip.kf_embed(ctx.h, C.byref(dd), 300, None, out.data_ptr()) == -20


def test_lm_head_argmax_first_max(ctx, O):
    rng = np.random.default_rng(8)
    v, d = 5000, 256
    w = rand_w(rng, v, d)
    w[4000] = w[123]  # exact duplicate rows: identical logits, the first index must win
    w[4500] = w[123]
    x = O.f32_to_bf16((O.bf16_to_f32(w[123]) * 50).astype(np.float32))  # make row 123 the maximum
    ow = oracle_weight(O, w, v, d, L.BF16)
    dw = ctx.upload_blob(L.BF16, v, d, ow.blob())
    logits, am = ctx.lm_head(dw, bf16_t(x, ctx.device))
    ref = O.linear(ow, x)
    assert ulp_diff_bf16(u16(logits), ref).max() <= 1
    assert am == O.argmax_bf16(u16(logits)) == 123


@pytest.mark.parametrize("cfg", [(16, 8, 128), (4, 2, 64), (8, 1, 128), (4, 4, 128)])
@pytest.mark.parametrize("pos", [0, 3, 63, 64, 200, 1500, 2047, 4095])
def test_attn_decode_vs_oracle(ctx, O, cfg, pos):
    nh, nkv, hd = cfg
    rng = np.random.default_rng(pos * 7 + nh)
    kvd = nkv * hd
    q = O.f32_to_bf16(rng.normal(0, 1.0, size=nh * hd).astype(np.float32))
    kc = O.f32_to_bf16(rng.normal(0, 1.0, size=(pos + 1, kvd)).astype(np.float32))
    vc = O.f32_to_bf16(rng.normal(0, 1.0, size=(pos + 1, kvd)).astype(np.float32))
    out = u16(ctx.attn_decode(bf16_t(q, ctx.device), bf16_t(kc, ctx.device), bf16_t(vc, ctx.device), pos, nh, nkv, hd))
    ref = O.attn_decode(q, kc, vc, pos, nh, nkv, hd, mode=O.ATTN_FUSED)
    d = ulp_diff_bf16(out, ref)
    assert close_bf16(out, ref).all() and (d > 0).mean() <= 0.02, (d.max(), (d > 0).mean())
    # against the reference's own (bf16 score / bf16 probability) rounding chain: stated tolerance 2^-6 of max|out|
    refc = O.bf16_to_f32(O.attn_decode(q, kc, vc, pos, nh, nkv, hd, mode=O.ATTN_REF))
    assert np.abs(O.bf16_to_f32(out) - refc).max() <= 2.0 ** -6 * np.abs(refc).max()


def test_attn_spiked_scores(ctx, O):
    """one key dominates (forces the running-max rescale inside a slice and across slices)"""
    nh, nkv, hd, pos = 4, 2, 128, 700
    rng = np.random.default_rng(77)
    q = O.f32_to_bf16(rng.normal(0, 1.0, size=nh * hd).astype(np.float32))
    kc = O.f32_to_bf16(rng.normal(0, 1.0, size=(pos + 1, nkv * hd)).astype(np.float32))
    vc = O.f32_to_bf16(rng.normal(0, 1.0, size=(pos + 1, nkv * hd)).astype(np.float32))
    qf = O.bf16_to_f32(q).reshape(nh, hd)
    kc = kc.copy()
    kc[555, :hd] = O.f32_to_bf16(qf[0] * 4.0)   # spike late in a slice for head 0/1's kv head
    kc[3, hd:] = O.f32_to_bf16(qf[2] * 6.0)     # spike early for the other kv head
    out = u16(ctx.attn_decode(bf16_t(q, ctx.device), bf16_t(kc, ctx.device), bf16_t(vc, ctx.device), pos, nh, nkv, hd))
    ref = O.attn_decode(q, kc, vc, pos, nh, nkv, hd, mode=O.ATTN_FUSED)
    assert ulp_diff_bf16(out, ref).max() <= 1


def test_attn_block_writes_key_and_matches(ctx, O):
    nh, nkv, hd, pos = 16, 8, 128, 130
    rng = np.random.default_rng(21)
    kvd = nkv * hd
    q = O.f32_to_bf16(rng.normal(0, 1.0, size=nh * hd).astype(np.float32))
    kraw = O.f32_to_bf16(rng.normal(0, 1.0, size=kvd).astype(np.float32))
    kc = O.f32_to_bf16(rng.normal(0, 1.0, size=(pos + 1, kvd)).astype(np.float32))
    vc = O.f32_to_bf16(rng.normal(0, 1.0, size=(pos + 1, kvd)).astype(np.float32))
    wq = O.f32_to_bf16((1 + rng.normal(0, 0.1, size=hd)).astype(np.float32))
    wk = O.f32_to_bf16((1 + rng.normal(0, 0.1, size=hd)).astype(np.float32))
    table = ctx.rope_table(256, hd, 1e6)
    kcd = bf16_t(kc, ctx.device).clone()
    out = u16(ctx.attn_block(bf16_t(q, ctx.device), bf16_t(kraw, ctx.device), kcd, bf16_t(vc, ctx.device), bf16_t(wq, ctx.device), bf16_t(wk, ctx.device),
                             table, pos, nh, nkv, hd))
    qq = O.rope(O.headnorm(q, wq, nh, hd), nh, hd, pos, 1e6)
    kk = O.rope(O.headnorm(kraw, wk, nkv, hd), nkv, hd, pos, 1e6)
    kc2 = kc.copy()
    kc2[pos] = kk
    assert np.array_equal(u16(kcd)[pos], kk), "normed+roped key row not written bit-exactly"
    assert np.array_equal(u16(kcd)[:pos], kc[:pos]), "other cache rows touched"
    ref = O.attn_decode(qq, kc2, vc, pos, nh, nkv, hd, mode=O.ATTN_FUSED)
    d = ulp_diff_bf16(out, ref)
    assert d.max() <= 1 and (d > 0).mean() <= 0.02


def test_fused_launches_equal_unfused(ctx, O):
    """kf_norm_linear / kf_norm_gateup_swiglu are the same arithmetic as rmsnorm + linear (+ swiglu): bit-identical."""
    rng = np.random.default_rng(31)
    dim, ffn = 1024, 3072
    x = bf16_t(O.f32_to_bf16(rng.normal(0, 1.0, size=dim).astype(np.float32)), ctx.device)
    nw = bf16_t(O.f32_to_bf16((1 + rng.normal(0, 0.1, size=dim)).astype(np.float32)), ctx.device)
    ws = [ctx.quantize(bf16_t(rand_w(rng, m, dim), ctx.device), L.Q4) for m in (2048, 1024, 1024)]
    ys = ctx.norm_linear(x, nw, ws)
    xn = ctx.rmsnorm(x, nw)
    for w, y in zip(ws, ys):
        assert torch.equal(y, ctx.linear(w, xn))
    g = ctx.quantize(bf16_t(rand_w(rng, ffn, dim), ctx.device), L.Q4)
    u = ctx.quantize(bf16_t(rand_w(rng, ffn, dim), ctx.device), L.Q4)
    act = ctx.norm_gateup_swiglu(x, nw, g, u)
    assert torch.equal(act, ctx.swiglu(ctx.linear(g, xn), ctx.linear(u, xn)))


def test_linear_batch_of_tokens(ctx, O):
    """SLP::Forw with nToken > 1: y[t] = W . x[t] for every token row"""
    import ctypes as C
    rng = np.random.default_rng(41)
    m, k, nt = 192, 512, 5
    w = rand_w(rng, m, k)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, k)).astype(np.float32))
    ow = oracle_weight(O, w, m, k, L.Q4)
    dw = ctx.upload_blob(L.Q4, m, k, ow.blob())
    xd = bf16_t(x, ctx.device)
    y = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device)
    d = dw.desc()
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), xd.data_ptr(), y.data_ptr(), None, nt, 1.0, 0.0, 0, None) == 0
    for t in range(nt):
        assert close_bf16(u16(y[t]), O.linear(ow, x[t])).all()


def _awq_case(O, rng, n_in, n_out):
    q = rng.integers(0, 16, size=(n_in, n_out))
    z = rng.integers(0, 16, size=(n_in // 128, n_out))
    s = rng.uniform(0.003, 0.012, size=(n_in // 128, n_out)).astype(np.float16)
    return O.AWQWeight(n_out, n_in, O.awq_pack(q), O.awq_pack(z), s)


@pytest.mark.parametrize("shape", [(1024, 2048), (2048, 1024), (3072, 1024), (256, 64)])
def test_awq_layout_dequant_and_linear(ctx, O, shape):
    """vendor AutoAWQ GEMM format (CU_Q42X_awq): dequant bit-exact, mat-vec within 1 ulp of the oracle"""
    from koifish_amd.runtime import AWQDevWeight
    n_in, n_out = shape
    rng = np.random.default_rng(n_in + n_out)
    ow = _awq_case(O, rng, n_in, n_out)
    dw = AWQDevWeight(n_out, n_in, torch.from_numpy(ow.data.view(np.int32)).to(ctx.device), torch.from_numpy(ow.qzeros.view(np.int32)).to(ctx.device),
                      torch.from_numpy(ow.scales.view(np.int16)).to(ctx.device))
    assert np.array_equal(u16(ctx.dequant(dw)), O.dequant_awq(ow))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=n_in).astype(np.float32))
    y = u16(ctx.linear(dw, bf16_t(x, ctx.device)))
    ref = O.linear(ow, x)
    assert close_bf16(y, ref).all() and (ulp_diff_bf16(y, ref) > 0).mean() <= 5e-3
    res = O.f32_to_bf16(rng.normal(0, 0.5, size=n_out).astype(np.float32))
    y2 = u16(ctx.linear(dw, bf16_t(x, ctx.device), residual=bf16_t(res, ctx.device)))
    assert close_bf16(y2, O.add(res, ref)).all()
    # the fused entry points refuse this layout instead of misreading it
    import ctypes as C
    d = dw.desc()
    wp = (C.c_void_p * 1)(C.addressof(d))
    yy = torch.zeros(n_out, dtype=torch.bfloat16, device=ctx.device)
    yp = (C.c_void_p * 1)(yy.data_ptr())
    assert ctx.hip.kf_norm_linear(ctx.h, bf16_t(x, ctx.device).data_ptr(), None, 1e-6, 1, wp, yp, None, 0, None) == -1000


def test_f8e5m2_matvec_every_byte_value(ctx, O):
    """the f8 mat-vec converts with gfx950's packed E5M2 -> fp32 instruction: every finite byte value (subnormals included) must come out as
    half(byte << 8) does in the oracle -- row r holds byte r in its first column and x = e0, so y[r] is that weight"""
    k = 32
    w = np.zeros((256, k), dtype=np.uint8)
    w[:, 0] = np.arange(256, dtype=np.uint8)
    w[:, 1:] = 0x3C                      # 1.0: multiplied by x = 0
    ow = O.QWeight(O.F8E5M2, 256, k, w.reshape(-1))
    dw = ctx.upload_blob(L.F8E5M2, 256, k, ow.blob())
    x = np.zeros(k, dtype=np.float32)
    x[0] = 1.0
    xb = O.f32_to_bf16(x)
    y = u16(ctx.linear(dw, bf16_t(xb, ctx.device)))
    ref = O.linear(ow, xb)
    finite = (np.arange(256) & 0x7C) != 0x7C
    assert np.array_equal(y[finite], ref[finite])
    nz = finite & (np.arange(256) != 0x80)   # byte 0x80 is -0: the dot product 0 + (-0 * 1) is +0 on both sides
    assert np.array_equal(y[nz], O.dequant(ow)[nz, 0])
    assert y[0x80] == 0x0000 and O.bf16_to_f32(y[0x01:0x02])[0] == 2.0 ** -16   # the smallest subnormal


def test_argmax_rows_state_copy_blocks_memset32(ctx, O):
    """the three small entries behind a batch of prompts (XcdReplicas::PrefillBatch):
      * kf_argmax_rows_state: the FIRST maximum over float(logits) of every row (sample_argmax, GoPT.cpp:602-612) -- ties, negative rows, a maximum in the last column -- and the
        decode-state update of the row's sequence: tokens_out[seq][pos] = id, state = {id, pos + 1}, the other words and sequences untouched;
      * kf_copy_blocks: block b of a source to table[b] + offset, nothing else written;
      * kf_memset32: words on the stream."""
    import ctypes as C
    hip, dev = ctx.hip, ctx.device
    rng = np.random.default_rng(21)
    n, n_rows, n_seq, stride = 5003, 5, 9, 40
    lg = O.f32_to_bf16(rng.normal(0, 2.0, (n_rows, n)).astype(np.float32))
    top = O.f32_to_bf16(np.array([9.0], dtype=np.float32))[0]
    lg[0, 17] = lg[0, 4000] = lg[0, 900] = top          # three equal maxima: the first wins
    lg[1, n - 1] = top                                   # the last column
    lg[2] = O.f32_to_bf16(-np.abs(rng.normal(1, 0.2, n)).astype(np.float32))   # all negative
    lg[3, 0] = top                                       # the first column
    seqs = np.array([7, 2, 0, 8, 4], dtype=np.int32)
    states = np.tile(np.array([[-5, 3, 11, 12]], dtype=np.int32), (n_seq, 1))
    states[:, 1] = np.arange(n_seq) + 3
    out = np.full((n_seq, stride), -1, dtype=np.int32)
    d_lg, d_seq, d_st, d_out = bf16_t(lg, dev), torch.from_numpy(seqs).to(dev), torch.from_numpy(states.copy()).to(dev), torch.from_numpy(out.copy()).to(dev)
    assert hip.kf_argmax_rows_state(ctx.h, d_lg.data_ptr(), n, n, n_rows, d_seq.data_ptr(), d_st.data_ptr(), d_out.data_ptr(), stride) == 0, hip.kf_last_error()
    ctx.sync()
    f = O.bf16_to_f32(lg)
    want = [int(np.argmax(f[r])) for r in range(n_rows)]   # numpy's argmax is the first maximum
    assert want[0] == 17 and want[1] == n - 1 and want[3] == 0
    g_st, g_out = d_st.cpu().numpy(), d_out.cpu().numpy()
    for r, s in enumerate(seqs):
        assert g_st[s].tolist() == [want[r], states[s, 1] + 1, 11, 12] and g_out[s, states[s, 1]] == want[r]
        g_out[s, states[s, 1]] = -1
    for s in set(range(n_seq)) - set(seqs.tolist()):
        assert g_st[s].tolist() == states[s].tolist()
    assert (g_out == -1).all()
    assert hip.kf_argmax_rows_state(ctx.h, d_lg.data_ptr(), n - 1, n, n_rows, d_seq.data_ptr(), d_st.data_ptr(), None, stride) != 0   # ld < n
    # scatter of blocks through a device table
    blk, nb = 4096 + 16, 6
    src = torch.from_numpy(rng.integers(0, 255, (nb, blk + 32), dtype=np.uint8)).to(dev)      # blocks blk + 32 bytes apart
    dst = torch.zeros(nb + 2, 2 * blk, dtype=torch.uint8, device=dev)
    order = [5, 0, 3, 1, 7, 2]
    table = torch.tensor([dst[o].data_ptr() for o in order], dtype=torch.int64, device=dev)
    assert hip.kf_copy_blocks(ctx.h, table.data_ptr(), 64, src.data_ptr(), blk + 32, blk, nb) == 0, hip.kf_last_error()
    ctx.sync()
    h_src, h_dst = src.cpu().numpy(), dst.cpu().numpy()
    for b, o in enumerate(order):
        assert np.array_equal(h_dst[o, 64:64 + blk], h_src[b, :blk]) and not h_dst[o, :64].any() and not h_dst[o, 64 + blk:].any()
    assert not h_dst[4].any() and not h_dst[6].any()
    assert hip.kf_copy_blocks(ctx.h, table.data_ptr(), 8, src.data_ptr(), blk + 32, blk, nb) != 0       # offset not a multiple of 16
    # words on the stream
    w = torch.zeros(10, dtype=torch.int32, device=dev)
    assert hip.kf_memset32(ctx.h, w.data_ptr() + 8, -7, 5) == 0, hip.kf_last_error()
    ctx.sync()
    assert w.cpu().numpy().tolist() == [0, 0, -7, -7, -7, -7, -7, 0, 0, 0]
