"""Token-batch linear (SLP::Forw with nToken > 1) on the MFMA tile kernel vs the oracle's token-serial mat-vec.

Tolerance: the tile kernel accumulates in fp32 like the mat-vec but in the MFMA's order, so results agree with the oracle to
<= 1 bf16 ulp (or 2^-10 of the row scale for elements near zero), and with the exact fp64 product of the dequantised weights
to 2^-8 of the largest output."""
import ctypes as C

import numpy as np
import pytest
import torch

from koifish_amd import lib as L
from tests.conftest import bf16_t, close_bf16, u16, ulp_diff_bf16

pytestmark = pytest.mark.gpu
TYPES = [L.BF16, L.F8E5M2, L.Q4, L.T_SIGN, L.BOOL1]


def _case(ctx, O, t, m, k, nt, seed, std=0.02):
    rng = np.random.default_rng(seed)
    w = O.f32_to_bf16(rng.normal(0, std, size=(m, k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, k)).astype(np.float32))
    ow = O.quantize(w, m, k, t)
    dw = ctx.upload_blob(t, m, k, ow.blob())
    return ow, dw, x


def _run(ctx, dw, x, nt, m, bias=None, alpha=1.0, beta=0.0, y0=None, residual=None):
    xd = bf16_t(x, ctx.device)
    y = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device) if y0 is None else bf16_t(y0, ctx.device).clone()
    d = dw.desc()
    rc = ctx.hip.kf_linear(ctx.h, C.byref(d), xd.data_ptr(), y.data_ptr(), bias.data_ptr() if bias is not None else None, nt, alpha, beta,
                           1 if residual is not None else 0, residual.data_ptr() if residual is not None else None)
    assert rc == 0, ctx.hip.kf_last_error()
    ctx.sync()
    return u16(y)


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("shape", [(2048, 1024, 128), (1024, 3072, 37), (96, 256, 8), (3072, 1024, 200), (40, 128, 129), (1000, 512, 64)])
def test_gemm_vs_oracle(ctx, O, t, shape):
    m, k, nt = shape
    ow, dw, x = _case(ctx, O, t, m, k, nt, hash((t, shape)) & 0xFFFF)
    y = _run(ctx, dw, x, nt, m)
    deq = O.bf16_to_f32(O.dequant(ow)).astype(np.float64)
    exact = O.bf16_to_f32(x).astype(np.float64) @ deq.T
    assert np.abs(O.bf16_to_f32(y) - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 1e-6
    for tt in sorted({0, 1, nt // 2, nt - 1}):
        ref = O.linear(ow, x[tt])
        assert close_bf16(y[tt], ref).all(), "token %d: max ulp %d" % (tt, ulp_diff_bf16(y[tt], ref).max())


def test_gemm_equals_matvec_loop_within_one_ulp(ctx, O, monkeypatch):
    """same call through the per-token mat-vec loop (KF_GEMM_MIN is read once per process: compare against ctx.linear per row)"""
    m, k, nt = 512, 1024, 48
    ow, dw, x = _case(ctx, O, L.Q4, m, k, nt, 77)
    y = _run(ctx, dw, x, nt, m)
    for tt in range(nt):
        yv = u16(ctx.linear(dw, bf16_t(x[tt], ctx.device)))
        assert close_bf16(y[tt], yv).all()


def test_gemm_epilogues(ctx, O):
    m, k, nt = 256, 1024, 40
    ow, dw, x = _case(ctx, O, L.Q4, m, k, nt, 5)
    rng = np.random.default_rng(6)
    b = O.f32_to_bf16(rng.normal(0, 0.1, size=m).astype(np.float32))
    y0 = O.f32_to_bf16(rng.normal(0, 0.5, size=(nt, m)).astype(np.float32))
    res = O.f32_to_bf16(rng.normal(0, 0.5, size=(nt, m)).astype(np.float32))
    y = _run(ctx, dw, x, nt, m, bias=bf16_t(b, ctx.device), alpha=0.5, beta=2.0, y0=y0)
    for tt in (0, 17, nt - 1):
        assert close_bf16(y[tt], O.linear(ow, x[tt], bias=b, alpha=0.5, beta=2.0, y=y0[tt])).all()
    y = _run(ctx, dw, x, nt, m, residual=bf16_t(res, ctx.device))
    for tt in (0, 17, nt - 1):
        assert close_bf16(y[tt], O.add(res[tt], O.linear(ow, x[tt]))).all()


def test_gemm_linearity_full_size(ctx, O):
    """size-independent property at a Qwen3-32B shard shape: W(x1 + x2) = W x1 + W x2 up to bf16 stores"""
    m, k, nt = 6400, 5120, 256
    rng = np.random.default_rng(3)
    w = torch.randn(m, k, device=ctx.device, dtype=torch.float32).mul_(0.02).to(torch.bfloat16)
    dw = ctx.quantize(w, L.Q4)
    x1 = O.f32_to_bf16(rng.integers(-8, 9, size=(nt, k)).astype(np.float32) / 8)
    x2 = O.f32_to_bf16(rng.integers(-8, 9, size=(nt, k)).astype(np.float32) / 8)
    xs = O.f32_to_bf16(O.bf16_to_f32(x1) + O.bf16_to_f32(x2))  # exact in bf16
    y1, y2, ysum = (O.bf16_to_f32(_run(ctx, dw, v, nt, m)) for v in (x1, x2, xs))
    scale = np.abs(ysum).max()
    assert np.abs(ysum - (y1 + y2)).max() <= 2.0 ** -6 * scale


@pytest.mark.parametrize("t", [L.Q4, L.BF16, L.BOOL1])
@pytest.mark.parametrize("nt", [8, 100, 300])
def test_multi_and_paired_launches_equal_separate_ones(ctx, O, t, nt):
    """kf_linear_multi (Q, K, V in one launch) and kf_gateup_swiglu_batch are the same arithmetic as kf_linear per matrix (+ kf_swiglu):
    bit-identical outputs"""
    k = 1024
    rng = np.random.default_rng(nt + t)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, k)).astype(np.float32))
    xd = bf16_t(x, ctx.device)
    ms = (2048, 1024, 1000)
    dws = []
    for i, m in enumerate(ms):
        w = O.f32_to_bf16(rng.normal(0, 0.02, size=(m, k)).astype(np.float32))
        dws.append(ctx.upload_blob(t, m, k, O.quantize(w, m, k, t).blob()))
    descs = [d.desc() for d in dws]
    ys = [torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device) for m in ms]
    wp = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
    yp = (C.c_void_p * 3)(*[y.data_ptr() for y in ys])
    assert ctx.hip.kf_linear_multi(ctx.h, 3, wp, xd.data_ptr(), yp, nt) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    for d, y, m in zip(dws, ys, ms):
        assert np.array_equal(u16(y), _run(ctx, d, x, nt, m))
    # gate/up pair
    m = 3072
    g = ctx.upload_blob(t, m, k, O.quantize(O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32)), m, k, t).blob())
    u = ctx.upload_blob(t, m, k, O.quantize(O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32)), m, k, t).blob())
    act = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device)
    tmp = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device)
    gd, ud = g.desc(), u.desc()
    assert ctx.hip.kf_gateup_swiglu_batch(ctx.h, C.byref(gd), C.byref(ud), xd.data_ptr(), act.data_ptr(), tmp.data_ptr(), nt) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    yg, yu = _run(ctx, g, x, nt, m), _run(ctx, u, x, nt, m)
    want = ctx.swiglu(bf16_t(yg, ctx.device).view(-1), bf16_t(yu, ctx.device).view(-1))
    assert np.array_equal(u16(act).reshape(-1), u16(want).reshape(-1))


@pytest.mark.parametrize("t", [L.Q4, L.F8E5M2])
@pytest.mark.parametrize("nt", [1100, 2048])
def test_stacked_large_batch_route_vs_exact(ctx, O, t, nt):
    """kf_linear_multi / kf_gateup_swiglu_batch at >= 1024 rows with the workspace kf_linear_multi_scratch_bytes asks for: Q | K | V (gate | up) are dequantised back
    to back and multiplied by ONE launch of the 256 x 256 bf16 tile kernel, each matrix's rows landing in its own output -- against the exact fp64 products of the
    dequantised weights (the tolerance of the other token-batch GEMM tests)"""
    k = 1024
    rng = np.random.default_rng(nt + t)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, k)).astype(np.float32))
    xd = bf16_t(x, ctx.device)
    ms = (2048, 1024, 1024)
    ows = [O.quantize(O.f32_to_bf16(rng.normal(0, 0.02, size=(m, k)).astype(np.float32)), m, k, t) for m in ms]
    dws = [ctx.upload_blob(t, m, k, ow.blob()) for m, ow in zip(ms, ows)]
    descs = [d.desc() for d in dws]
    wp = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
    need = ctx.hip.kf_linear_multi_scratch_bytes(3, wp, nt)
    assert need == sum(m * k * 2 for m in ms)
    assert ctx.hip.kf_linear_multi_scratch_bytes(3, wp, 512) == 0      # small batches: the in-register-unpack kernels, no workspace
    ctx.sync()
    ctx._lin_ws = torch.empty(max(need, 3072 * k * 4), dtype=torch.uint8, device=ctx.device)
    L.check(ctx.hip.kf_set_scratch(ctx.h, C.c_void_p(ctx._lin_ws.data_ptr()), C.c_size_t(ctx._lin_ws.numel())), "kf_set_scratch")
    ys = [torch.full((nt, m), 7.0, dtype=torch.bfloat16, device=ctx.device) for m in ms]
    yp = (C.c_void_p * 3)(*[y.data_ptr() for y in ys])
    assert ctx.hip.kf_linear_multi(ctx.h, 3, wp, xd.data_ptr(), yp, nt) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    for ow, y, m in zip(ows, ys, ms):
        ref = f(x) @ f(O.dequant(ow)).reshape(m, k).T
        assert np.abs(f(u16(y)) - ref).max() <= 2.0 ** -7 * np.abs(ref).max()
    m = 3072
    og, ou = (O.quantize(O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32)), m, k, t) for _ in range(2))
    g, u = ctx.upload_blob(t, m, k, og.blob()), ctx.upload_blob(t, m, k, ou.blob())
    act = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device)
    tmp = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device)
    gd, ud = g.desc(), u.desc()
    assert ctx.hip.kf_gateup_swiglu_batch(ctx.h, C.byref(gd), C.byref(ud), xd.data_ptr(), act.data_ptr(), tmp.data_ptr(), nt) == 0, ctx.hip.kf_last_error()
    ctx.sync()
    yg = f(x) @ f(O.dequant(og)).reshape(m, k).T
    yu = f(x) @ f(O.dequant(ou)).reshape(m, k).T
    ref = yg / (1.0 + np.exp(-yg)) * yu
    assert np.abs(f(u16(act)) - ref).max() <= 2.0 ** -6 * np.abs(ref).max()
    # (round 4: gate | up are dequantised interleaved and the SwiGLU expression runs in the tile GEMM's epilogue -- the `up` projection is never written: up_scratch stays untouched)
    assert not u16(tmp).any()


@pytest.mark.parametrize("t", [L.Q4, L.BF16, L.T_SIGN])
def test_gemm_many_token_tiles_vs_exact(ctx, O, t):
    """a batch of 17 token tiles on the staged kernel (ragged tail): exact fp64 product of the dequantised weights"""
    m, k, nt = 4096, 512, 2100
    rng = np.random.default_rng(17 + t)
    w = (torch.randn(m, k, device=ctx.device) * 0.02).to(torch.bfloat16)
    dw = ctx.quantize(w, t)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, k)).astype(np.float32))
    y = O.bf16_to_f32(_run(ctx, dw, x, nt, m))
    deq = ctx.dequant(dw).float().cpu().numpy().astype(np.float64)
    exact = O.bf16_to_f32(x).astype(np.float64) @ deq.T
    assert np.abs(y - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 1e-6
    # and the last rows as a batch of their own: <= 1 bf16 ulp apart
    y2 = _run(ctx, dw, x[-100:], 100, m)
    assert close_bf16(O.f32_to_bf16(y[-100:].astype(np.float32)), y2).all()


@pytest.mark.parametrize("m,k,nt", [(4096, 1024, 2048), (4100, 512, 2050), (8192, 64, 4096), (1024, 2048, 2048), (1032, 3072, 2000), (3072, 1024, 1500)])
def test_large_batch_bf16_tile_kernel(ctx, O, m, k, nt):
    """the global_load_lds tile kernels (kf_gemm3.hip: bf16 operands; 256 x 256 tiles from 128 tiles up, else 128 x 128 tiles on four waves -- the last three shapes)
    against the fp32 product of the same bf16 values: ragged last tiles in both dimensions, and the epilogue (alpha, beta, bias, residual) in gemm_epilogue's order"""
    g = torch.Generator(device=ctx.device)
    g.manual_seed(m + k + nt)
    W = (torch.randn(m, k, generator=g, device=ctx.device) * 0.05).to(torch.bfloat16)
    X = torch.randn(nt, k, generator=g, device=ctx.device).to(torch.bfloat16)
    dw = ctx.quantize(W, L.BF16)
    d = dw.desc()
    y = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device)
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), X.data_ptr(), y.data_ptr(), None, nt, 1.0, 0.0, 0, None) == 0, ctx.hip.kf_last_error()
    ref = X.float() @ W.float().t()
    scale = ref.abs().max().item()
    assert (y.float() - ref).abs().max().item() <= 2.0 ** -8 * scale
    bias = (torch.randn(m, generator=g, device=ctx.device) * 0.1).to(torch.bfloat16)
    res = torch.randn(nt, m, generator=g, device=ctx.device).to(torch.bfloat16)
    y0 = torch.randn(nt, m, generator=g, device=ctx.device).to(torch.bfloat16)
    y2 = y0.clone()
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), X.data_ptr(), y2.data_ptr(), bias.data_ptr(), nt, 0.5, 2.0, 1, res.data_ptr()) == 0, ctx.hip.kf_last_error()
    want = (res.float() + (0.5 * ref + 2.0 * y0.float() + bias.float()).to(torch.bfloat16).float()).to(torch.bfloat16)
    assert (y2.float() - want.float()).abs().max().item() <= 2.0 ** -6 * max(scale, want.float().abs().max().item())
    ctx.sync()


@pytest.mark.parametrize("m,k,nt", [(4096, 1024, 2048), (1024, 2048, 4096)])   # 128 big tiles; 32 big = 256 small tiles (the 128 x 128 form from 4096 rows at M 1024)
def test_training_size_batch_of_a_quantised_weight_takes_the_dequantise_then_tile_path(ctx, O, m, k, nt):
    """>= 2048 token rows with the caller's scratch set (kf_linear_scratch_bytes says how much): GetDataX into the scratch, then the bf16 tile kernel.  The product of the
    SAME dequantised values, so it agrees with the fused dequant-GEMM kernels to the fp32 summation order (<= 1 bf16 ulp of the row scale)."""
    rng = np.random.default_rng(5)
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(m, k)).astype(np.float32))
    dw = ctx.upload_blob(L.Q4, m, k, O.quantize(w, m, k, L.Q4).blob())
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, k)).astype(np.float32))
    d = dw.desc()
    assert ctx.hip.kf_linear_scratch_bytes(C.byref(d), nt) == m * k * 2 and ctx.hip.kf_linear_scratch_bytes(C.byref(d), 1024) == 0
    fused = _run(ctx, dw, x, nt, m)                       # no scratch handed over: the fused kernels
    assert ctx.linear_scratch(dw, nt) == m * k * 2
    tiled = _run(ctx, dw, x, nt, m)
    a, b = O.bf16_to_f32(fused), O.bf16_to_f32(tiled)
    assert np.abs(a - b).max() <= 2.0 ** -7 * np.abs(a).max() and (a != b).mean() < 0.2
    L.check(ctx.hip.kf_set_scratch(ctx.h, None, 0), "kf_set_scratch")
    ctx._lin_ws = None


@pytest.mark.parametrize("nt,seq", [(1100, 0), (2047, 0), (1280, 160)])
def test_fused_qkv_rope_epilogue_equals_the_two_launch_route(ctx, O, nt, seq):
    """kf_qkv_rope_batch at >= 1024 tokens: Q | K | V stacked in ONE tile-GEMM launch with q/k-norm + RoPE in its epilogue (a 128-row tile is one head: the token's 128
    squares meet through row swaps and LDS, a rotation pair's other element sits in the other wave at the same lane) against kf_linear_multi + kf_qknorm_rope_batch:
    same tile kernel, same prep_head_cs arithmetic -- every bit of q, k (rotated) and v equal; plus the exact fp64 reference within the batch-GEMM tolerance.
    seq > 0: kf_qkv_rope_seqs -- the rows are sequences of `seq` tokens back to back (a batch of prompts), positions restart in each -- against kf_linear_multi +
    kf_qknorm_rope_train."""
    k, hd, n_head, n_kv, pos0, eps = 1024, 128, 16, 8, (0 if seq else 3), 1e-6
    rng = np.random.default_rng(nt)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, k)).astype(np.float32))
    xd = bf16_t(x, ctx.device)
    ms = (n_head * hd, n_kv * hd, n_kv * hd)
    ows = [O.quantize(O.f32_to_bf16(rng.normal(0, 0.03, size=(m, k)).astype(np.float32)), m, k, L.Q4) for m in ms]
    dws = [ctx.upload_blob(L.Q4, m, k, ow.blob()) for m, ow in zip(ms, ows)]
    descs = [d.desc() for d in dws]
    wp = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
    need = ctx.hip.kf_linear_multi_scratch_bytes(3, wp, nt)
    assert need > 0
    ctx.sync()
    ctx._lin_ws = torch.empty(need, dtype=torch.uint8, device=ctx.device)
    L.check(ctx.hip.kf_set_scratch(ctx.h, C.c_void_p(ctx._lin_ws.data_ptr()), C.c_size_t(ctx._lin_ws.numel())), "kf_set_scratch")
    wq = bf16_t(O.f32_to_bf16((1 + rng.normal(0, 0.1, size=hd)).astype(np.float32)), ctx.device)
    wk = bf16_t(O.f32_to_bf16((1 + rng.normal(0, 0.1, size=hd)).astype(np.float32)), ctx.device)
    table = ctx.rope_table(pos0 + nt + 1, hd, 1e6)
    outs = {}
    for fused in (True, False):
        ys = [torch.full((nt, m), 7.0, dtype=torch.bfloat16, device=ctx.device) for m in ms]
        if fused and seq:
            L.check(ctx.hip.kf_qkv_rope_seqs(ctx.h, C.byref(descs[0]), C.byref(descs[1]), C.byref(descs[2]), xd.data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), nt, seq,
                                             wq.data_ptr(), wk.data_ptr(), table.data_ptr(), 0, n_head, n_kv, hd, eps), "kf_qkv_rope_seqs")
        elif fused:
            L.check(ctx.hip.kf_qkv_rope_batch(ctx.h, C.byref(descs[0]), C.byref(descs[1]), C.byref(descs[2]), xd.data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), nt,
                                              wq.data_ptr(), wk.data_ptr(), table.data_ptr(), pos0, n_head, n_kv, hd, eps), "kf_qkv_rope_batch")
        else:
            yp = (C.c_void_p * 3)(*[y.data_ptr() for y in ys])
            L.check(ctx.hip.kf_linear_multi(ctx.h, 3, wp, xd.data_ptr(), yp, nt), "kf_linear_multi")
            if seq:
                L.check(ctx.hip.kf_qknorm_rope_train(ctx.h, ys[0].data_ptr(), ys[1].data_ptr(), wq.data_ptr(), wk.data_ptr(), table.data_ptr(), nt, seq, ms[0], ms[1], n_head, n_kv, hd, eps,
                                                     None, None), "kf_qknorm_rope_train")
            else:
                L.check(ctx.hip.kf_qknorm_rope_batch(ctx.h, ys[0].data_ptr(), ys[1].data_ptr(), wq.data_ptr(), wk.data_ptr(), table.data_ptr(), pos0, nt, ms[0], ms[1], n_head, n_kv, hd, eps),
                        "kf_qknorm_rope_batch")
        ctx.sync()
        outs[fused] = [u16(y).copy() for y in ys]
    for a, b, name in zip(outs[True], outs[False], "qkv"):
        assert np.array_equal(a, b), "%s: %d of %d values differ between the fused epilogue and the two-launch route" % (name, int((a != b).sum()), a.size)
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    v_ref = f(x) @ f(O.dequant(ows[2])).reshape(ms[2], k).T
    assert np.abs(f(outs[True][2]) - v_ref).max() <= 2.0 ** -7 * np.abs(v_ref).max()
    assert np.abs(f(outs[True][0])).max() > 0.1   # q really went through the norm (values of order one)
    if seq:   # the positions restart: sequence 3's rows equal ... only if its inputs did; instead: a call with mismatched nTok / seq_len is refused
        assert ctx.hip.kf_qkv_rope_seqs(ctx.h, C.byref(descs[0]), C.byref(descs[1]), C.byref(descs[2]), xd.data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), nt, seq + 1,
                                        wq.data_ptr(), wk.data_ptr(), table.data_ptr(), 0, n_head, n_kv, hd, eps) != 0


@pytest.mark.parametrize("t", [L.Q4, L.T_SIGN])
def test_resident_dequantised_copies_give_the_scratch_routes_results(ctx, O, t):
    """kf_set_dequant_arena: the stacked Q | K | V and the interleaved gate | up routes find their bf16 copies in the arena from the second call on (no dequantise launch) and
    return what the per-call scratch route returns, bit for bit (same tile kernels, same operand values); kf_linear takes the bf16 tile kernels
    on the resident copy from 1024 rows -- against the exact fp64 product, with residual; the arena fills once, an exhausted arena falls back to the scratch, and
    setting it again forgets every copy"""
    k, nt = 1024, 1536
    rng = np.random.default_rng(77 + t)
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, k)).astype(np.float32))
    xd = bf16_t(x, ctx.device)
    ms = (2048, 1024, 1024)
    ows = [O.quantize(O.f32_to_bf16(rng.normal(0, 0.02, size=(m, k)).astype(np.float32)), m, k, t) for m in ms]
    dws = [ctx.upload_blob(t, m, k, ow.blob()) for m, ow in zip(ms, ows)]
    descs = [d.desc() for d in dws]
    wp = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
    m = 3072
    og, ou = (O.quantize(O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32)), m, k, t) for _ in range(2))
    g, u = ctx.upload_blob(t, m, k, og.blob()), ctx.upload_blob(t, m, k, ou.blob())
    gd, ud = g.desc(), u.desc()
    ctx.sync()
    ctx._lin_ws = torch.empty(2 * m * k * 2, dtype=torch.uint8, device=ctx.device)
    L.check(ctx.hip.kf_set_scratch(ctx.h, C.c_void_p(ctx._lin_ws.data_ptr()), C.c_size_t(ctx._lin_ws.numel())), "kf_set_scratch")

    def run():
        ys = [torch.full((nt, mm), 7.0, dtype=torch.bfloat16, device=ctx.device) for mm in ms]
        yp = (C.c_void_p * 3)(*[y.data_ptr() for y in ys])
        assert ctx.hip.kf_linear_multi(ctx.h, 3, wp, xd.data_ptr(), yp, nt) == 0, ctx.hip.kf_last_error()
        act = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device)
        tmp = torch.zeros(nt, m, dtype=torch.bfloat16, device=ctx.device)
        assert ctx.hip.kf_gateup_swiglu_batch(ctx.h, C.byref(gd), C.byref(ud), xd.data_ptr(), act.data_ptr(), tmp.data_ptr(), nt) == 0, ctx.hip.kf_last_error()
        ctx.sync()
        return [u16(y).copy() for y in ys] + [u16(act).copy()]

    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    try:
        base = run()  # no arena: dequantise into the scratch, every call
        total = sum(mm * k * 2 for mm in ms) + 2 * m * k * 2 + ms[1] * k * 2
        arena = torch.empty(total, dtype=torch.uint8, device=ctx.device)
        L.check(ctx.hip.kf_set_dequant_arena(ctx.h, C.c_void_p(arena.data_ptr()), C.c_size_t(total)), "kf_set_dequant_arena")
        assert ctx.hip.kf_dequant_arena_used(ctx.h) == 0
        first = run()   # fills
        used = ctx.hip.kf_dequant_arena_used(ctx.h)
        assert used == sum(mm * k * 2 for mm in ms) + 2 * m * k * 2
        ctx._lin_ws.fill_(0x5a)  # the scratch holds nothing the second call could use
        ctx.sync()
        second = run()  # finds
        assert ctx.hip.kf_dequant_arena_used(ctx.h) == used
        for a, b, c_ in zip(base, first, second):
            assert np.array_equal(a, b) and np.array_equal(a, c_)
        # kf_linear on a resident copy: K's matrix again, alone (its own key), with the residual epilogue
        res = O.f32_to_bf16(rng.normal(0, 1.0, size=(nt, ms[1])).astype(np.float32))
        y = bf16_t(res, ctx.device)
        for _ in range(2):
            y.copy_(bf16_t(res, ctx.device))
            assert ctx.hip.kf_linear(ctx.h, C.byref(descs[1]), xd.data_ptr(), y.data_ptr(), None, nt, 1.0, 0.0, L.KF_EPI_RESIDUAL, y.data_ptr()) == 0, ctx.hip.kf_last_error()
            ctx.sync()
            ref = f(x) @ f(O.dequant(ows[1])).reshape(ms[1], k).T
            out = f(u16(y)) - f(res)
            assert np.abs(out - ref).max() <= 2.0 ** -6 * max(np.abs(ref).max(), np.abs(f(res)).max())
        assert ctx.hip.kf_dequant_arena_used(ctx.h) == total  # now full
        # a full arena: other weights go through the scratch as before
        w2 = ctx.upload_blob(t, ms[0], k, ows[0].blob())
        d2 = w2.desc()
        y2 = torch.zeros(nt, ms[0], dtype=torch.bfloat16, device=ctx.device)
        assert ctx.hip.kf_linear(ctx.h, C.byref(d2), xd.data_ptr(), y2.data_ptr(), None, nt, 1.0, 0.0, 0, None) == 0, ctx.hip.kf_last_error()
        ctx.sync()
        ref = f(x) @ f(O.dequant(ows[0])).reshape(ms[0], k).T
        assert np.abs(f(u16(y2)) - ref).max() <= 2.0 ** -7 * np.abs(ref).max()
        assert ctx.hip.kf_dequant_arena_used(ctx.h) == total
        L.check(ctx.hip.kf_set_dequant_arena(ctx.h, C.c_void_p(arena.data_ptr()), C.c_size_t(total)), "kf_set_dequant_arena")
        assert ctx.hip.kf_dequant_arena_used(ctx.h) == 0  # forgotten
    finally:
        ctx.sync()
        L.check(ctx.hip.kf_set_dequant_arena(ctx.h, None, 0), "kf_set_dequant_arena")
