"""Qwen3-0.6B at full size (BASELINE.json configs[1] shapes) on the GPU: parity against the CPU oracle on a few steps (the
oracle needs ~0.2 s per token per 100 cores, so the sample is small), and size-independent properties over a long run."""
import numpy as np
import pytest

from koifish_amd import lib as L
from koifish_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
LOGIT_TOL = 2.0 ** -6


@pytest.fixture(scope="module")
def model():
    cfg = dict(synth.CONFIGS["qwen3-0.6b"])
    m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
    yield cfg, m
    m.close()


@pytest.mark.fast_order   # (against the oracle's dot16 order at 2^-6; the canonical order's own check is bit-exact: test_gpu_canonical.py)
def test_full_size_logits_and_ids_vs_oracle(model):
    cfg, m = model
    om = O.from_device_model(m)
    ids = np.random.default_rng(7).integers(0, cfg["vocab"], size=6)
    for pos, tok in enumerate(ids):
        g_next, g_logits = m.forward(int(tok), pos)
        o_next, o_logits, _ = om.decode(int(tok), pos)
        gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
        err = np.abs(gl - ol).max() / np.abs(ol).max()
        assert err <= LOGIT_TOL, "pos %d: logits off by %g of max" % (pos, err)
        assert g_next == O.argmax_bf16(g_logits)
        top2 = np.sort(ol)[-2:]
        if top2[1] - top2[0] > 2.0 ** -7 * np.abs(ol).max():   # two bf16 ulps of the logit scale
            assert g_next == o_next
    om.close()


def _decode_after_prefill(m, om, cfg, P, n_follow, seed):
    """Fills positions 0..P-1 with the batched prefill, hands the SAME KV rows to the oracle, then decodes positions P..P+n_follow-1 teacher-forced on
    both (hipGraph replay of the position's bucket: the persistent engine where it serves the bound, with its multi-slice attention from 256 keys
    up).  Returns per position (gpu id, oracle id, relative logit error, relative top-2 margin of the oracle's logits)."""
    toks = np.random.default_rng(seed).integers(0, cfg["vocab"], size=P + n_follow).astype(np.int32)
    m.prefill(toks[:P], want_logits=False)
    gk, gv = m.kv_to_host()
    ok, ov = om.kv()
    ok[:, :P] = gk[:, :P]
    ov[:, :P] = gv[:, :P]
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:P + n_follow] = toks
    m.set_forced(forced)
    m.set_state(int(toks[P - 1]), P - 1)   # the step at P reads forced[P]; the state only has to name a valid position below it
    out = []
    for p in range(P, P + n_follow):
        m.set_state(int(toks[p]), p)
        m.run_steps(p, 1, use_graph=True)
        m.sync()
        g_id, g_logits = int(m.tokens_out(p + 1)[p]), m.logits()
        o_id, o_logits, _ = om.decode(int(toks[p]), p)
        gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
        top2 = np.sort(ol)[-2:]
        out.append((g_id, o_id, float(np.abs(gl - ol).max() / np.abs(ol).max()), float((top2[1] - top2[0]) / np.abs(ol).max())))
        assert g_id == O.argmax_bf16(g_logits), "device arg-max is not the first maximum of its own logits"
    return out


TIE = 2.0 ** -7   # two bf16 ulps at the top of the logit range: GPU and oracle each round their own fp32 sum to bf16 once, so a top-2 margin inside it may flip


def test_full_size_every_bucket_logits_and_ids_vs_oracle(model):
    """BASELINE config 2 at full size across the whole sequence: 6 positions behind prefills of 127, 255, 1023 and 2042 tokens -- every hipGraph bucket
    (128, 256, 1024, 2048), multi-slice attention up to all 2048 keys, the persistent engine and its in-launch merge.  Logits within 1.25 x 2^-6 of the
    oracle's at every position; every greedy id whose top-2 margin (of the oracle's logits) exceeds two bf16 ulps of the logit scale must equal the
    oracle's.  The relative top-2 margin of 151 936 near-Gaussian logits does not depend on the weights' scale (mean ~ 1/24 of the maximum,
    exponentially distributed), so some positions always fall inside the window; the rule is kept from being vacuous by requiring that at least
    half of the 24 positions are decided by it."""
    cfg, m = model
    om = O.from_device_model(m)
    n, decided = 0, 0
    for P in (127, 255, 1023, 2042):
        for i, (g_id, o_id, err, margin) in enumerate(_decode_after_prefill(m, om, cfg, P, 6, seed=100 + P)):
            # v_dot2c order, 28 layers deep: measured 0.011-0.016 of the logit scale over these 24 positions (which KV rows the prompt left -- resident-copy or scratch
            # route -- moves single positions by +-0.002): the bound is 1.25 x 2^-6; the bit-exact statement is tests/test_gpu_canonical.py
            assert err <= 1.25 * LOGIT_TOL, "position %d: logits off by %g of max" % (P + i, err)
            n += 1
            if margin > TIE:
                decided += 1
                assert g_id == o_id, "position %d: greedy id %d vs oracle %d at margin %g of max" % (P + i, g_id, o_id, margin)
    om.close()
    assert m.engine_steps() > 0
    m.engine_check()
    assert decided >= n // 2, "only %d of %d positions have a top-2 margin above two bf16 ulps" % (decided, n)


def test_full_size_graph_equals_eager_and_is_deterministic(model):
    cfg, m = model
    prompt = np.random.default_rng(3).integers(0, cfg["vocab"], size=100)
    a = m.generate(prompt, 200, use_graph=True)      # crosses the 64 / 128 / 256 buckets
    b = m.generate(prompt, 200, use_graph=True)
    c = m.generate(prompt, 40, use_graph=False)
    assert a == b, "two identical runs differ: a kernel is not deterministic"
    assert a[:40] == c, "hipGraph replay and eager launches disagree"


def test_full_size_linearity_and_dequant_roundtrip(model):
    """properties that need no CPU run: W.(2x) == 2*(W.x) exactly in bf16 (power-of-two scaling commutes with every rounding),
    and every dequantised value lies on its group's 16-level bf16-stepwise grid."""
    import torch
    cfg, m = model
    ctx = m._ctx
    w = m.weights[(3, 4)]                            # layer 3 gate_proj, 3072 x 1024, Q4
    x = torch.randn(cfg["dim"], device=ctx.device).to(torch.bfloat16)
    y1 = ctx.linear(w, x)
    y2 = ctx.linear(w, (x.float() * 2).to(torch.bfloat16))
    assert torch.equal((y1.float() * 2).to(torch.bfloat16), y2)
    deq = ctx.dequant(w).view(-1, 128)               # every 128-element group sits on its own <= 16-level grid {bf16(bf16(step*q) - zero)}
    z, s = w.zero_step()
    lut = (s.float()[:, None] * torch.arange(16, device=ctx.device)[None, :]).to(torch.bfloat16).float() - z.float()[:, None]
    lut = lut.to(torch.bfloat16)
    hit = (deq[:, :, None] == lut[:, None, :]).any(-1)
    assert bool(hit.all()), "a dequantised value is off its group's grid"
    # the LM head's device arg-max equals torch's first arg-max over the logits it wrote
    head = m.weights[(-1, 1)]
    logits, am = ctx.lm_head(head, x)
    lf = logits.float()
    assert am == int((lf == lf.max()).nonzero()[0])


def test_full_size_batched_prefill_vs_token_serial(model):
    """128-token prompt: token batches (MFMA) vs the token-serial decode path on the same device model -- logits of the last
    prompt token within the oracle tolerance of each other, KV rows likewise, and the ids that follow agree unless the first
    differing step was a near-tie inside that tolerance."""
    cfg, m = model
    prompt = np.random.default_rng(11).integers(0, cfg["vocab"], size=128)
    m.set_prefill_mode(0)
    serial = m.generate(prompt, 24, use_graph=True)
    ks, vs = m.kv_to_host()
    ks, vs = ks[:, :128].copy(), vs[:, :128].copy()
    nxt, logits = m.prefill(prompt)
    kb, vb = m.kv_to_host()
    for a, b in ((kb[:, :128], ks), (vb[:, :128], vs)):
        fa, fb = O.bf16_to_f32(a), O.bf16_to_f32(b)
        for l in range(cfg["n_layer"]):
            # two fp32 summation orders, 28 layers deep on random weights: rounding differences of one bf16 ulp (2^-8) are amplified
            # layer by layer (measured: max 0.0002 of the row scale at layer 0, 0.018 at layer 27, rms 0.003); the bound is on the rms,
            # with a loose cap on the single worst element
            d = np.abs(fa[l] - fb[l])
            scale = np.abs(fb[l]).max()
            assert np.sqrt((d ** 2).mean()) <= 2.0 ** -8 * scale, "layer %d rms" % l
            assert d.max() <= 2.0 ** -5 * scale, "layer %d max" % l
    assert nxt == O.argmax_bf16(logits)
    m.set_prefill_mode(1)
    batched = m.generate(prompt, 24, use_graph=True)
    m.set_prefill_mode(0)
    if batched != serial:
        first = next(i for i in range(24) if batched[i] != serial[i])
        # replay the serial path up to the first difference and look at the margin there
        toks = list(prompt) + serial[:first]
        lg = None
        for pos, t in enumerate(toks):
            _, lg = m.forward(int(t), pos)
        fl = O.bf16_to_f32(lg)
        assert abs(fl[batched[first]] - fl[serial[first]]) <= 2 * LOGIT_TOL * np.abs(fl).max(), "ids diverge at step %d beyond a near-tie" % first


def test_full_size_long_prompt_on_resident_copies_vs_the_scratch_route(model):
    """A 1536-token prompt with the layers' bf16 copies kept resident (kf_set_dequant_arena through Fish::EnsureResident; the default) against the same prompt with
    the copies dequantised into the scratch per call: Q | K | V and gate | up run the SAME launches on the same operand values, so layer 0's K / V rows are equal bit for
    bit; o_proj / down_proj change kernel (bf16 tile GEMM in split-K pieces instead of the in-register unpack: another fp32 order), so deeper rows and the logits
    agree within the token-batch tolerance.  The copies are filled by the first prompt only; a second one gives the first one's bits."""
    cfg, m = model
    n = 1536
    prompt = np.random.default_rng(12).integers(0, cfg["vocab"], size=n)
    try:
        m.set_prefill_resident(False)
        nxt0, lg0 = m.prefill(prompt)
        assert m.resident_bytes() == 0
        k0, v0 = m.kv_to_host()
        k0, v0 = k0[:, :n].copy(), v0[:, :n].copy()
        m.set_prefill_resident(True)
        nxt1, lg1 = m.prefill(prompt)
        filled = m.resident_bytes()
        per_layer = 2 * sum(w.ne0 * w.ne1 for (layer, slot), w in m.weights.items() if layer == 0)
        assert filled == per_layer * cfg["n_layer"]
        k1, v1 = m.kv_to_host()
        k1, v1 = k1[:, :n].copy(), v1[:, :n].copy()
        nxt2, lg2 = m.prefill(prompt)
        assert m.resident_bytes() == filled
        k2, v2 = m.kv_to_host()
        assert nxt2 == nxt1 and np.array_equal(lg1, lg2) and np.array_equal(k1, k2[:, :n]) and np.array_equal(v1, v2[:, :n])
        assert np.array_equal(k0[0], k1[0]) and np.array_equal(v0[0], v1[0])
        for a, b in ((k1, k0), (v1, v0)):
            fa, fb = O.bf16_to_f32(a), O.bf16_to_f32(b)
            for l in range(cfg["n_layer"]):
                d = np.abs(fa[l] - fb[l])
                scale = np.abs(fb[l]).max()
                assert np.sqrt((d ** 2).mean()) <= 2.0 ** -8 * scale, "layer %d rms" % l
                assert d.max() <= 2.0 ** -5 * scale, "layer %d max" % l
        f0, f1 = O.bf16_to_f32(lg0), O.bf16_to_f32(lg1)
        assert np.abs(f0 - f1).max() <= 2.0 ** -6 * np.abs(f0).max()
        assert nxt1 == O.argmax_bf16(lg1)
        if nxt1 != nxt0:  # a near-tie inside the tolerance
            assert abs(f0[nxt0] - f0[nxt1]) <= 2 * 2.0 ** -6 * np.abs(f0).max()
    finally:
        m.set_prefill_resident(True)


def test_resident_copy_routes_on_32b_shaped_layers():
    """Two layers of the Qwen3-32B shape (5120 wide, 64 / 8 heads, ffn 25600: the 256 x 256 and 192 x 256 tiles, GQA 8 in the prompt attention): a 1100-token prompt on the
    resident bf16 copies against the same prompt with a dequantise per call -- logits within the token-batch tolerance, the greedy id equal unless a near-tie, the copies'
    size = 2 bytes per layer weight."""
    cfg = dict(synth.CONFIGS["qwen3-32b"])
    cfg["n_layer"] = 2
    m = synth.build_on_gpu(cfg, seed=99, layer_type=L.Q4, head_type=L.BF16)
    try:
        prompt = np.random.default_rng(13).integers(0, cfg["vocab"], size=1100)
        m.set_prefill_resident(False)
        nxt0, lg0 = m.prefill(prompt)
        assert m.resident_bytes() == 0
        m.set_prefill_resident(True)
        nxt1, lg1 = m.prefill(prompt)
        assert m.resident_bytes() == 2 * sum(w.ne0 * w.ne1 for (layer, slot), w in m.weights.items() if layer >= 0)
        nxt2, lg2 = m.prefill(prompt)
        assert nxt2 == nxt1 and np.array_equal(lg1, lg2)
        f0, f1 = O.bf16_to_f32(lg0), O.bf16_to_f32(lg1)
        assert np.abs(f0 - f1).max() <= 2.0 ** -6 * np.abs(f0).max()
        if nxt1 != nxt0:
            assert abs(f0[nxt0] - f0[nxt1]) <= 2 * 2.0 ** -6 * np.abs(f0).max()
        assert nxt1 == O.argmax_bf16(lg1)
    finally:
        m.close()
