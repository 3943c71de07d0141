"""Linear layer backward (kf_linear_backward; SLP::Back, NeuronFuse.cu:495-547) through the C-ABI: input gradient, weight gradient and bias gradient
against fp64 products of the oracle's dequantised weights (the tolerance of the forward GEMM tests: 2^-8 of the output scale, which covers the bf16
store and any fp32 summation order), the bias gradient bit for bit against the oracle's fixed-order column sums; accumulation into existing
gradients; the shapes of the GPT-2 and Qwen3 layers."""
import ctypes as C

import numpy as np
import pytest
import torch

from koifish_amd import lib as L
from oracle import oracle as O
from tests.conftest import bf16_t, u16

pytestmark = pytest.mark.gpu


def _run(ctx, t, OC, IC, n, accumulate, with_bias=True, want_gw=True):
    rng = np.random.default_rng(OC * 7 + IC * 3 + n + (t if isinstance(t, int) else len(t) + ord(t[-1])))
    w = O.f32_to_bf16(rng.normal(0, 0.05, (OC, IC)).astype(np.float32))
    if t in ("nf4", "nf3"):      # row-codebook storage: GetDataX is the row-table lookup (kf_lut.hip), the GEMMs are the same
        bits = 4 if t == "nf4" else 3
        ow = O.quantize_nf4(w, OC, IC, bits=bits)
        dw = ctx.upload_lut_blob(OC, IC, ow.blob(), bits=bits)
    else:
        ow = O.quantize(w, OC, IC, t)
        dw = ctx.upload_blob(t, OC, IC, ow.blob())
    dIn = O.f32_to_bf16(rng.normal(0, 1.0, (n, OC)).astype(np.float32))
    inp = O.f32_to_bf16(rng.normal(0, 1.0, (n, IC)).astype(np.float32))
    delta0 = O.f32_to_bf16(rng.normal(0, 1.0, (n, IC)).astype(np.float32))
    gW0 = O.f32_to_bf16(rng.normal(0, 1.0, (OC, IC)).astype(np.float32))
    gb0 = O.f32_to_bf16(rng.normal(0, 1.0, OC).astype(np.float32))
    dev = ctx.device
    d_delta, d_gW, d_gb = bf16_t(delta0, dev), bf16_t(gW0, dev), bf16_t(gb0, dev)
    nb = ctx.hip.kf_linear_backward_scratch_bytes(OC, IC, n)
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    sp = (scratch.data_ptr() + 255) & ~255
    desc = dw.desc()
    d_dIn, d_inp = bf16_t(dIn, dev), bf16_t(inp, dev)   # named: a temporary would be freed (and its block reused) as soon as data_ptr() returns
    rc = ctx.hip.kf_linear_backward(ctx.h, C.byref(desc), d_dIn.data_ptr(), d_inp.data_ptr(), d_delta.data_ptr(), d_gW.data_ptr() if want_gw else None,
                                    d_gb.data_ptr() if with_bias else None, n, int(accumulate), sp)
    assert rc == 0, ctx.hip.kf_last_error()
    ctx.sync()
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    Wd = f(O.dequant(ow)).reshape(OC, IC)
    ref_delta = f(dIn) @ Wd + (f(delta0) if accumulate else 0.0)
    assert np.abs(f(u16(d_delta)) - ref_delta).max() <= 2.0 ** -8 * np.abs(ref_delta).max() + 1e-6
    if want_gw:
        ref_gW = f(dIn).T @ f(inp) + f(gW0)
        assert np.abs(f(u16(d_gW)) - ref_gW).max() <= 2.0 ** -8 * np.abs(ref_gW).max() + 1e-6
    else:
        assert np.array_equal(u16(d_gW), gW0)
    if with_bias:
        ref_gb = gb0.copy()
        O.colsum_add(dIn, ref_gb)
        assert np.array_equal(u16(d_gb), ref_gb)
    else:
        assert np.array_equal(u16(d_gb), gb0)


@pytest.mark.parametrize("t", [L.Q4, L.F8E5M2, L.BF16])
@pytest.mark.parametrize("shape", [(256, 128, 128), (1024, 3072, 128), (4800, 1600, 192), (1600, 6400, 320), (1024, 1024, 1024), (1600, 1600, 2048)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_linear_backward(ctx, t, shape, accumulate):
    OC, IC, n = shape
    if t == L.Q4 and (OC * IC) % 128:
        pytest.skip("group size")
    _run(ctx, t, OC, IC, n, accumulate)


@pytest.mark.parametrize("t", ["nf4", "nf3"])
@pytest.mark.parametrize("shape", [(256, 128, 128), (1024, 3072, 128), (1600, 1600, 2048)])
def test_linear_backward_row_codebook_weights(ctx, t, shape):
    OC, IC, n = shape
    _run(ctx, t, OC, IC, n, True)


def test_linear_forward_large_batch(ctx):
    """n >= 2048 rows (a training-size batch) through kf_linear: the hand-written tile kernels by default (KF_GEMM_LIB=1 would route it through dequantise +
    rocBLAS), against the oracle's rows"""
    rng = np.random.default_rng(4)
    OC, IC, n = 1600, 1600, 2048
    for t in (L.Q4, L.F8E5M2, L.BF16):
        w = O.f32_to_bf16(rng.normal(0, 0.05, (OC, IC)).astype(np.float32))
        ow = O.quantize(w, OC, IC, t)
        dw = ctx.upload_blob(t, OC, IC, ow.blob())
        x = O.f32_to_bf16(rng.normal(0, 1.0, (n, IC)).astype(np.float32))
        b = O.f32_to_bf16(rng.normal(0, 0.5, OC).astype(np.float32))
        res = O.f32_to_bf16(rng.normal(0, 1.0, (n, OC)).astype(np.float32))
        xd, bd, rd = bf16_t(x, ctx.device), bf16_t(b, ctx.device), bf16_t(res, ctx.device)
        y = torch.zeros(n, OC, dtype=torch.bfloat16, device=ctx.device)
        desc = dw.desc()
        assert ctx.hip.kf_linear(ctx.h, C.byref(desc), xd.data_ptr(), y.data_ptr(), bd.data_ptr(), n, 1.0, 0.0, 1, rd.data_ptr()) == 0, ctx.hip.kf_last_error()
        ctx.sync()
        f = lambda a: O.bf16_to_f32(a).astype(np.float64)
        ref = f(x) @ f(O.dequant(ow)).reshape(OC, IC).T + f(b) + f(res)
        assert np.abs(f(u16(y)) - ref).max() <= 2.0 ** -7 * np.abs(ref).max()
        rows = [0, 1, 777, n - 1]
        for r_ in rows:   # and against the oracle's own epilogue order on a few rows: within the double rounding of the two-pass epilogue
            o_row = O.add(res[r_], O.linear(ow, x[r_], bias=b))
            assert (np.abs(f(u16(y)[r_]) - f(o_row)) <= 2.0 ** -6 * np.abs(f(o_row)).max()).all()   # up to three bf16 roundings apart at the top of the range


@pytest.mark.parametrize("t", [L.Q4, L.BF16])
@pytest.mark.parametrize("shape", [(2048, 2048, 2048), (1600, 6400, 2048), (4800, 1600, 4096), (1600, 1600, 8192), (2048, 2056, 2048)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_linear_backward_large_k_major(ctx, t, shape, accumulate):
    """shapes with >= 64 tiles of 256 x 256: both GEMMs run the large tile kernel on K-MAJOR operands (W as it lies for the input gradient; inp and deltaIn
    as they lie for the weight gradient, contraction over the token rows) -- no transposes; ragged M (1600 = 6.25 tiles, 2056) included"""
    OC, IC, n = shape
    if t == L.Q4 and IC % 128:
        pytest.skip("group size")
    _run(ctx, t, OC, IC, n, accumulate)


def test_linear_backward_split_k_is_deterministic(ctx):
    """the split-K forms add a tile's partial sums in slot order, not in arrival order: two runs give the same bits (49-tile weight gradient = 5 pieces per
    tile; 133 tiles = owners + helpers on tails)"""
    rng = np.random.default_rng(77)
    for OC, IC, n in ((1600, 1600, 8192), (4800, 1600, 4096), (1024, 3072, 8192)):
        w = O.f32_to_bf16(rng.normal(0, 0.05, (OC, IC)).astype(np.float32))
        dw = ctx.upload_blob(L.BF16, OC, IC, O.quantize(w, OC, IC, L.BF16).blob())
        dIn = bf16_t(O.f32_to_bf16(rng.normal(0, 1.0, (n, OC)).astype(np.float32)), ctx.device)
        inp = bf16_t(O.f32_to_bf16(rng.normal(0, 1.0, (n, IC)).astype(np.float32)), ctx.device)
        scratch = torch.empty(ctx.hip.kf_linear_backward_scratch_bytes(OC, IC, n) + 256, dtype=torch.uint8, device=ctx.device)
        sp = (scratch.data_ptr() + 255) & ~255
        desc = dw.desc()
        outs = []
        for _ in range(3):
            delta = torch.zeros(n, IC, dtype=torch.bfloat16, device=ctx.device)
            gW = torch.zeros(OC, IC, dtype=torch.bfloat16, device=ctx.device)
            scratch.random_(0, 255)   # stale partials / flags from the run before must not matter
            assert ctx.hip.kf_linear_backward(ctx.h, C.byref(desc), dIn.data_ptr(), inp.data_ptr(), delta.data_ptr(), gW.data_ptr(), None, n, 0, sp) == 0, ctx.hip.kf_last_error()
            ctx.sync()
            outs.append((u16(delta).copy(), u16(gW).copy()))
        for d_, g_ in outs[1:]:
            assert np.array_equal(d_, outs[0][0]) and np.array_equal(g_, outs[0][1])
        # other operands through the same slots in between: a partial tile left by that launch (the slots are read with write-through-coherent loads, no
        # fence) must not reach the next one -- A, B, A gives A's bits again
        dIn2 = bf16_t(O.f32_to_bf16(rng.normal(0, 1.0, (n, OC)).astype(np.float32)), ctx.device)
        res = []
        for src in (dIn, dIn2, dIn):
            delta = torch.zeros(n, IC, dtype=torch.bfloat16, device=ctx.device)
            gW = torch.zeros(OC, IC, dtype=torch.bfloat16, device=ctx.device)
            assert ctx.hip.kf_linear_backward(ctx.h, C.byref(desc), src.data_ptr(), inp.data_ptr(), delta.data_ptr(), gW.data_ptr(), None, n, 0, sp) == 0, ctx.hip.kf_last_error()
            ctx.sync()
            res.append((u16(delta).copy(), u16(gW).copy()))
        assert np.array_equal(res[0][1], outs[0][1]) and np.array_equal(res[2][1], outs[0][1]) and not np.array_equal(res[1][1], outs[0][1])
        assert np.array_equal(res[2][0], outs[0][0])


def test_linear_backward_fixed_weight_and_no_bias(ctx):
    _run(ctx, L.Q4, 512, 1024, 96, False, with_bias=False, want_gw=False)   # isFixWeight: input gradient only; n need not be a multiple of 64 then


def test_linear_backward_rejects(ctx):
    w = O.f32_to_bf16(np.zeros((96, 128), np.float32))
    ow = O.quantize(w, 96, 128, L.BF16)
    dw = ctx.upload_blob(L.BF16, 96, 128, ow.blob())
    desc = dw.desc()
    z = torch.zeros(64 * 128, dtype=torch.bfloat16, device=ctx.device)
    scratch = torch.empty(1 << 20, dtype=torch.uint8, device=ctx.device)
    sp = (scratch.data_ptr() + 255) & ~255
    assert ctx.hip.kf_linear_backward(ctx.h, C.byref(desc), z.data_ptr(), z.data_ptr(), z.data_ptr(), None, None, 64, 0, sp) == -20   # OC = 96 is not a multiple of 64
