"""Fused classifier (kf_fused_classifier; fused_classifier.cuh:68-140) through the C-ABI against the oracle's thread-by-thread restatement:
losses, logit gradients and probabilities bit for bit; ignore-mask, loss-only mode, accumulation, GPT-2's V = 50257 / padded 50264, and -- at the
BASELINE config 3 size (8 x 1024 rows) -- the properties that do not need the oracle: each gradient row sums to ~0 and mean loss ~ ln V."""
import numpy as np
import pytest
import torch

from koifish_amd import lib as L
from oracle import oracle as O
from tests.conftest import bf16_t, u16

pytestmark = pytest.mark.gpu


def _case(rows, V, P, seed, scale=3.0):
    rng = np.random.default_rng(seed)
    lg = np.full((rows, P), 0x7fc0, np.uint16)  # NaN padding: must never be read
    lg[:, :V] = O.f32_to_bf16((rng.standard_normal((rows, V)) * scale).astype(np.float32))
    tg = rng.integers(0, V, rows).astype(np.int32)
    return lg, tg


@pytest.mark.parametrize("rows,V,P", [(6, 50257, 50264), (3, 1000, 1000), (4, 66, 72), (2, 9001, 9008), (3, 7, 8), (2, 8193, 8200), (1, 1, 8)])
@pytest.mark.parametrize("with_probs", [False, True])
def test_fused_classifier_bit_exact(ctx, rows, V, P, with_probs):
    lg, tg = _case(rows, V, P, rows * 7919 + V)
    dloss = 1.0 / (rows * 3)
    ld = bf16_t(lg, ctx.device)
    td = torch.from_numpy(tg).to(ctx.device)
    losses = torch.full((rows,), 0.5, dtype=torch.float32, device=ctx.device)
    probs = torch.zeros(rows, P, dtype=torch.bfloat16, device=ctx.device) if with_probs else None
    assert ctx.hip.kf_fused_classifier(ctx.h, ld.data_ptr(), losses.data_ptr(), probs.data_ptr() if with_probs else None, dloss, td.data_ptr(), rows, 1, V, P,
                                       None, 1) == 0
    ctx.sync()
    ref_l = np.full(rows, 0.5, np.float32)
    ref_g = lg.copy()
    ref_p = O.fused_classifier(ref_g, ref_l, tg, V, dloss=dloss, want_probs=with_probs)
    assert np.array_equal(losses.cpu().numpy(), ref_l)
    assert np.array_equal(u16(ld), ref_g)  # gradients over the logits, padding untouched
    if with_probs:
        assert np.array_equal(u16(probs)[:, :V], ref_p[:, :V])


def test_fused_classifier_mask_and_loss_only(ctx):
    rows, V, P = 8, 515, 520
    lg, tg = _case(rows, V, P, 99, scale=1.0)
    mask = np.array([0, 0x10000, 5, 0x10003, 0, 0, 0x10000, 0], np.int32)
    ld = bf16_t(lg, ctx.device)
    td, md = torch.from_numpy(tg).to(ctx.device), torch.from_numpy(mask).to(ctx.device)
    losses = torch.zeros(rows, dtype=torch.float32, device=ctx.device)
    assert ctx.hip.kf_fused_classifier(ctx.h, ld.data_ptr(), losses.data_ptr(), None, 1.0, td.data_ptr(), 2, 4, V, P, md.data_ptr(), 1) == 0
    ctx.sync()
    ref_l, ref_g = np.zeros(rows, np.float32), lg.copy()
    O.fused_classifier(ref_g, ref_l, tg, V, mask=mask)
    assert np.array_equal(losses.cpu().numpy(), ref_l) and np.array_equal(u16(ld), ref_g)
    # loss only: logits stay, and a second call accumulates
    ld2 = bf16_t(lg, ctx.device)
    for _ in range(2):
        assert ctx.hip.kf_fused_classifier(ctx.h, ld2.data_ptr(), losses.data_ptr(), None, 1.0, td.data_ptr(), 2, 4, V, P, md.data_ptr(), 0) == 0
    ctx.sync()
    assert np.array_equal(u16(ld2), lg)
    skip = (mask & 0x10000) != 0
    got = losses.cpu().numpy()
    assert np.all(got[skip] == 0) and np.allclose(got[~skip], 3 * ref_l[~skip], rtol=1e-6)


def test_fused_classifier_rejects_bad_shapes(ctx):
    ld = torch.zeros(4, 16, dtype=torch.bfloat16, device=ctx.device)
    losses = torch.zeros(4, dtype=torch.float32, device=ctx.device)
    td = torch.zeros(4, dtype=torch.int32, device=ctx.device)
    assert ctx.hip.kf_fused_classifier(ctx.h, ld.data_ptr(), losses.data_ptr(), None, 1.0, td.data_ptr(), 4, 1, 16, 12, None, 1) == -20  # KF_INVALID_ARGS: P < V
    assert ctx.hip.kf_fused_classifier(ctx.h, ld.data_ptr(), losses.data_ptr(), None, 1.0, td.data_ptr(), 4, 1, 10, 12, None, 1) == -2000  # KF_BLAS_UNALIGN: P % 8
    assert ctx.hip.kf_fused_classifier(ctx.h, None, losses.data_ptr(), None, 1.0, td.data_ptr(), 4, 1, 16, 16, None, 1) == -20


def test_fused_classifier_full_size_properties(ctx):
    """BASELINE config 3 size: 8 x 1024 rows of GPT-2's vocabulary (823 MB of logits)."""
    B, T, V, P = 8, 1024, 50257, 50264
    g = torch.Generator(device=ctx.device).manual_seed(5)
    lg = (torch.randn(B * T, P, generator=g, device=ctx.device) * 0.5).to(torch.bfloat16)
    tg = torch.randint(0, V, (B * T,), generator=g, device=ctx.device, dtype=torch.int32)
    losses = torch.zeros(B * T, dtype=torch.float32, device=ctx.device)
    ref = torch.nn.functional.cross_entropy(lg[:64, :V].float(), tg[:64].long(), reduction="none")
    assert ctx.hip.kf_fused_classifier(ctx.h, lg.data_ptr(), losses.data_ptr(), None, 1.0, tg.data_ptr(), B, T, V, P, None, 1) == 0
    ctx.sync()
    assert torch.allclose(losses[:64], ref, rtol=1e-5, atol=1e-5)
    assert abs(float(losses.mean()) - (np.log(V) + 0.125)) < 0.01  # E[logsumexp] of N(0, 0.25) logits = ln V + sigma^2 / 2
    rowsum = lg[:, :V].float().sum(dim=1)  # sum(prob) - 1 = 0 up to the bf16 rounding of 50257 tiny terms
    assert float(rowsum.abs().max()) < 2e-3
    picked = lg[torch.arange(B * T, device=ctx.device), tg.long()].float()
    assert bool((picked < -0.99).all())  # prob(target) - 1
