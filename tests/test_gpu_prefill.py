"""Batched prompt prefill (token batches on the MFMA tile kernels, Fish::Prefill) vs the CPU oracle's token-serial forward.

The oracle restates the reference, which prefills one token at a time (Fish::Chat, GoPT.cpp:1139-1146).  The batched path runs the
same per-token arithmetic with fp32 sums in MFMA order, so: logits of the last prompt token within 2^-6 of max|logit|, KV rows within
the same relative bound, greedy ids identical on the committed seeds."""
import numpy as np
import pytest

from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
LOGIT_TOL = 2.0 ** -6


def _pair(cfg_name, layer_type, head_type, n_prompt, seed=1234, w_std=0.02):
    cfg = synth.CONFIGS[cfg_name]
    raw = synth.raw_weights_numpy(cfg, seed, w_std=w_std)
    gm = synth.build_from_raw(cfg, raw, layer_type, head_type)
    om = oracle_model(cfg, raw, layer_type, head_type)
    return cfg, gm, om, prompt_ids(cfg, n_prompt)


def _oracle_prefill(om, prompt):
    nxt, logits = None, None
    for pos, tok in enumerate(prompt):
        nxt, logits, _ = om.decode(int(tok), pos)
    return nxt, logits


@pytest.mark.parametrize("cfg_name,layer_type,head_type,n", [("tiny", L.Q4, L.BF16, 40), ("tiny", L.BF16, L.BF16, 33), ("tiny", L.F8E5M2, L.BF16, 16),
                                                             ("tiny", L.T_SIGN, L.BF16, 40), ("tiny", L.BOOL1, L.BF16, 40), ("tiny", L.Q4, L.Q4, 9),
                                                             ("small", L.Q4, L.BF16, 130), ("small", L.Q4, L.BF16, 7)])
def test_prefill_logits_kv_and_next_id(cfg_name, layer_type, head_type, n):
    cfg, gm, om, prompt = _pair(cfg_name, layer_type, head_type, n)
    g_next, g_logits = gm.prefill(prompt)
    o_next, o_logits = _oracle_prefill(om, prompt)
    gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
    assert np.abs(gl - ol).max() <= LOGIT_TOL * np.abs(ol).max()
    assert g_next == O.argmax_bf16(g_logits)
    assert g_next == o_next
    gk, gv = gm.kv_to_host()
    ok, ov = om.kv()
    for g, o in ((gk, ok), (gv, ov)):
        for l in range(cfg["n_layer"]):
            a, b = O.bf16_to_f32(g[l, :n]), O.bf16_to_f32(o[l, :n])
            assert np.abs(a - b).max() <= LOGIT_TOL * np.abs(b).max(), "layer %d KV rows differ" % l
    gm.close()


@pytest.mark.parametrize("cfg_name", ["tiny", "small"])
def test_generate_with_batched_prefill_matches_oracle_and_serial(cfg_name):
    cfg, gm, om, prompt = _pair(cfg_name, L.Q4, L.BF16, 24, w_std=0.1)
    n_new = 24 if cfg_name == "tiny" else 12
    ref = om.generate(prompt.tolist(), n_new)
    serial = gm.generate(prompt, n_new, use_graph=True)
    gm.set_prefill_mode(1)
    batched = gm.generate(prompt, n_new, use_graph=True)
    batched_eager = gm.generate(prompt, n_new, use_graph=False)
    assert serial == ref
    assert batched == ref, "ids after a batched prefill differ from the oracle"
    assert batched_eager == batched
    gm.close()


def test_chunked_prefill_and_continuation():
    """prompt longer than the chunk: chunks see the cache rows of earlier chunks; a second prefill continues at pos0 > 0"""
    cfg, gm, om, prompt = _pair("tiny", L.Q4, L.BF16, 50)
    gm.set_prefill_mode(1, chunk=16)
    g_next, g_logits = gm.prefill(prompt)
    o_next, o_logits = _oracle_prefill(om, prompt)
    gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
    assert np.abs(gl - ol).max() <= LOGIT_TOL * np.abs(ol).max() and g_next == o_next
    # same prompt in two calls
    cfg, gm2, _, _ = _pair("tiny", L.Q4, L.BF16, 50)
    gm2.prefill(prompt[:21])
    n2, l2 = gm2.prefill(prompt[21:], pos0=21)
    assert n2 == g_next
    assert np.abs(O.bf16_to_f32(l2) - ol).max() <= LOGIT_TOL * np.abs(ol).max()
    gm.close()
    gm2.close()


def test_prefill_bad_args():
    cfg, gm, om, prompt = _pair("tiny", L.Q4, L.BF16, 8)
    import ctypes as C
    t = np.array([1, 2, cfg["vocab"]], dtype=np.int32)
    assert gm.host.kfh_prefill(gm.h, t.ctypes.data_as(C.c_void_p), 3, 0) == -20          # id outside the table
    assert gm.host.kfh_prefill(gm.h, t.ctypes.data_as(C.c_void_p), 2, cfg["max_seq"] - 1) == -20  # runs past the context
    assert gm.host.kfh_prefill(gm.h, t.ctypes.data_as(C.c_void_p), 0, 0) == -20
    gm.close()


@pytest.mark.parametrize("nh,nkv,hd", [(16, 8, 128), (4, 2, 64), (8, 1, 128), (4, 4, 64), (8, 2, 128)])
@pytest.mark.parametrize("pos0,n", [(0, 128), (0, 37), (100, 70), (0, 300)])
def test_attn_prefill_kernel_vs_oracle(ctx, nh, nkv, hd, pos0, n):
    """kf_attn_prefill (MFMA flash form): every token against the oracle's decode attention at its position.  bf16 scores as the
    reference stores them; the probabilities enter P.V as bf16 (the reference's CU_softmax_multihead stores them in bf16 too), so
    the bound is the one the decode kernel is held to against the reference's chain: 2^-6 of max|out| vs REF, and 2^-7 vs FUSED."""
    import ctypes as C
    import torch
    from tests.conftest import bf16_t, u16
    rng = np.random.default_rng(pos0 * 31 + n + nh)
    kvd, qd = nkv * hd, nh * hd
    tot = pos0 + n
    q = O.f32_to_bf16(rng.normal(0, 1.0, size=(n, qd)).astype(np.float32))
    kc = O.f32_to_bf16(rng.normal(0, 1.0, size=(tot, kvd)).astype(np.float32))
    vc = O.f32_to_bf16(rng.normal(0, 1.0, size=(tot, kvd)).astype(np.float32))
    qd_t, kc_t, vc_t = bf16_t(q, ctx.device), bf16_t(kc, ctx.device), bf16_t(vc, ctx.device)
    out_t = torch.zeros(n, qd, dtype=torch.bfloat16, device=ctx.device)
    rc = ctx.hip.kf_attn_prefill(ctx.h, qd_t.data_ptr(), kc_t.data_ptr(), vc_t.data_ptr(), out_t.data_ptr(), pos0, n, qd, nh, nkv, hd, kvd)
    assert rc == 0, ctx.hip.kf_last_error()
    ctx.sync()
    out = u16(out_t)
    _check_attn_rows(q, kc, vc, out, pos0, sorted({0, 1, n // 3, n // 2, n - 2, n - 1}), nh, nkv, hd)


def _check_attn_rows(q, kc, vc, out, pos0, rows, nh, nkv, hd):
    for t in rows:
        pos = pos0 + t
        ref = O.bf16_to_f32(O.attn_decode(q[t], kc[:pos + 1], vc[:pos + 1], pos, nh, nkv, hd, mode=O.ATTN_FUSED))
        refc = O.bf16_to_f32(O.attn_decode(q[t], kc[:pos + 1], vc[:pos + 1], pos, nh, nkv, hd, mode=O.ATTN_REF))
        got = O.bf16_to_f32(out[t])
        assert np.abs(got - ref).max() <= 2.0 ** -7 * np.abs(ref).max(), "token %d vs FUSED" % t
        assert np.abs(got - refc).max() <= 2.0 ** -6 * np.abs(refc).max(), "token %d vs REF" % t


@pytest.mark.parametrize("nh,nkv,hd", [(16, 8, 128), (8, 1, 128), (4, 4, 64), (8, 2, 64)])
@pytest.mark.parametrize("pos0,n", [(0, 1024), (0, 1100), (5, 1031), (0, 2047)])
def test_attn_prefill_long_prompt_form_vs_oracle(ctx, nh, nkv, hd, pos0, n):
    """The form kf_attn_prefill takes from 1024 tokens when there is about one workgroup per CU: every workgroup holds a half block of tokens from the FRONT of the prompt
    and one from the BACK (equal key walks under the causal mask), eight waves in two key halves merged at the end.  Rows at both ends of the prompt, at the half-block
    seams (multiples of 64 / GQ tokens), in the middle block of an odd count and in the ragged last block -- against the oracle's decode attention, as the test above."""
    import torch
    from tests.conftest import bf16_t, u16
    rng = np.random.default_rng(pos0 * 31 + n + nh)
    kvd, qd = nkv * hd, nh * hd
    tot = pos0 + n
    q = O.f32_to_bf16(rng.normal(0, 1.0, size=(n, qd)).astype(np.float32))
    kc = O.f32_to_bf16(rng.normal(0, 1.0, size=(tot, kvd)).astype(np.float32))
    vc = O.f32_to_bf16(rng.normal(0, 1.0, size=(tot, kvd)).astype(np.float32))
    qd_t, kc_t, vc_t = bf16_t(q, ctx.device), bf16_t(kc, ctx.device), bf16_t(vc, ctx.device)
    out_t = torch.full((n, qd), 3.0, dtype=torch.bfloat16, device=ctx.device)
    rc = ctx.hip.kf_attn_prefill(ctx.h, qd_t.data_ptr(), kc_t.data_ptr(), vc_t.data_ptr(), out_t.data_ptr(), pos0, n, qd, nh, nkv, hd, kvd)
    assert rc == 0, ctx.hip.kf_last_error()
    ctx.sync()
    out = u16(out_t)
    ht = 64 // (nh // nkv)  # tokens per half block
    nsb = (n + ht - 1) // ht
    rows = {0, 1, ht - 1, ht, n - 1, n - 2, (nsb // 2) * ht, (nsb // 2) * ht - 1, min(n - 1, (nsb // 2) * ht + ht - 1), (nsb - 1) * ht, (nsb - 1) * ht - 1, n // 3, 2 * n // 3}
    _check_attn_rows(q, kc, vc, out, pos0, sorted(t for t in rows if 0 <= t < n), nh, nkv, hd)
    # every row was written (the fill value 3.0 cannot survive: outputs are convex combinations of N(0, 1) values rounded to bf16 -- |.| = 3.0 exactly in all of a row has probability 0)
    assert not (out == 0x4040).all(axis=1).any()
