"""Sparse ("EOE" / hot-neuron) forward, BASELINE config 5: D_matmul_sparse (src/Utils/GST_float.cpp:306-318) with CS_Picker's hot[] array
(src/Manifold/SparseNeuron.cpp:20-29, T_hot = 0.2): a row is computed only when hot[i] == 1, a cold row is 0 (+ bias).  The GPU turns the mask into
a row list on the device (kf_hot_rows) and walks it, so cold rows cost no HBM traffic."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import bf16_t, u16, ulp_diff_bf16, close_bf16
from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def hot_mask(n, frac=0.2, seed=5):
    """a seeded mask with T_hot = 0.2 of the FFN rows hot"""
    rng = np.random.default_rng(seed)
    hot = np.zeros(n, dtype=np.int32)
    hot[rng.permutation(n)[: max(int(n * frac), 16)]] = 1
    return hot


@pytest.mark.parametrize("n,frac", [(3072, 0.2), (1000, 0.5), (4096, 0.0), (2049, 1.0)])
def test_hot_rows_list(ctx, n, frac):
    hot = hot_mask(n, frac, seed=n) if frac > 0 else np.zeros(n, dtype=np.int32)
    hot[::7] *= 3 if frac < 1.0 else 1     # values other than 1 are cold (hot[i] == 1 is the test)
    d_hot = torch.from_numpy(hot).to(ctx.device)
    rows = torch.full((n + 1,), -1, dtype=torch.int32, device=ctx.device)
    L.check(ctx.hip.kf_hot_rows(ctx.h, d_hot.data_ptr(), n, rows.data_ptr(), rows.data_ptr() + 4 * n), "kf_hot_rows")
    got = rows.cpu().numpy()
    ref = np.nonzero(hot == 1)[0]
    assert got[n] == ref.size and np.array_equal(got[: ref.size], ref)


@pytest.mark.parametrize("type_", [L.Q4, L.BOOL1, L.T_SIGN, L.BF16])
@pytest.mark.parametrize("shape", [(3072, 1024), (512, 256)])
def test_linear_masked_vs_oracle_and_dense(ctx, type_, shape):
    m, k = shape
    rng = np.random.default_rng(m + type_)
    w = O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    bias = O.f32_to_bf16(rng.normal(0, 0.1, size=m).astype(np.float32))
    ow = O.quantize(w, m, k, type_)
    dw = ctx.upload_blob(type_, m, k, ow.blob())
    hot = hot_mask(m)
    d_hot = torch.from_numpy(hot).to(ctx.device)
    rows = torch.zeros(m + 1, dtype=torch.int32, device=ctx.device)
    L.check(ctx.hip.kf_hot_rows(ctx.h, d_hot.data_ptr(), m, rows.data_ptr(), rows.data_ptr() + 4 * m), "kf_hot_rows")
    n_hot = int(rows[m].item())
    assert n_hot == int((hot == 1).sum())
    xt, bt = bf16_t(x, ctx.device), bf16_t(bias, ctx.device)
    for b in (None, bt):
        y = torch.full((m,), 7.0, dtype=torch.bfloat16, device=ctx.device)
        d = dw.desc()
        L.check(ctx.hip.kf_linear_masked(ctx.h, C.byref(d), xt.data_ptr(), y.data_ptr(), None if b is None else b.data_ptr(), rows.data_ptr(), n_hot), "kf_linear_masked")
        got = u16(y)
        ref = O.linear_masked(ow, x, hot, None if b is None else bias)
        assert close_bf16(got, ref).all() and (ulp_diff_bf16(got, ref) > 0).mean() <= 0.01
        dense = u16(ctx.linear(dw, xt, bias=b))
        assert np.array_equal(got[hot == 1], dense[hot == 1]), "a hot row must carry every bit of the dense product's row"
        cold = got[hot != 1]
        assert np.array_equal(cold, (bias if b is not None else np.zeros(m, dtype=np.uint16))[hot != 1]), "a cold row is 0 (+ bias)"


@pytest.mark.parametrize("layer_type", [L.BOOL1, L.Q4])
def test_sparse_decode_vs_oracle(layer_type):
    """whole decode steps with the seed-5, 20 % mask on every layer's FFN rows: logits within the step tolerance of the oracle's sparse forward,
    greedy ids equal, fused == per-kernel path bit for bit, and a dense model differs (the mask is really applied)"""
    cfg = synth.CONFIGS["small"]
    raw = synth.raw_weights_numpy(cfg, 77, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, layer_type, L.BF16)
    om = oracle_model(cfg, raw, layer_type, L.BF16)
    for l in range(cfg["n_layer"]):
        hot = hot_mask(cfg["ffn"], 0.2, seed=5 + l)
        m.set_hot(l, hot)
        om.set_hot(l, hot)
    prompt = prompt_ids(cfg, 12)
    tok = int(prompt[0])
    fused = []
    for pos in range(20):
        g_next, g_logits = m.forward(tok, pos)
        o_next, o_logits, _ = om.decode(tok, pos)
        gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
        assert np.abs(gl - ol).max() <= 2.0 ** -6 * np.abs(ol).max(), "step %d" % pos
        assert g_next == o_next, "step %d: greedy id %d vs oracle %d" % (pos, g_next, o_next)
        fused.append(g_logits.copy())
        tok = int(prompt[pos + 1]) if pos + 1 < len(prompt) else o_next
    ids_graph = m.generate(prompt, 16, use_graph=True)
    assert ids_graph == om.generate(prompt.tolist(), 16)
    assert m.engine_steps() > 0, "the persistent engine serves the sparse forward (hot bits per workgroup, cold rows published as zeros)"
    m.set_fuse_level(0)
    tok = int(prompt[0])
    for pos in range(6):
        nxt, lg = m.forward(tok, pos)
        assert np.array_equal(lg, fused[pos]), "per-kernel and fused sparse paths differ at step %d" % pos
        tok = int(prompt[pos + 1])
    m.set_fuse_level(1)
    for l in range(cfg["n_layer"]):
        m.set_hot(l, None)
    _, dense_logits = m.forward(int(prompt[0]), 0)
    assert not np.array_equal(dense_logits, fused[0])
    m.close()


def test_full_size_one_bit_sparse_step():
    """Qwen3-0.6B shapes, 1-bit YinYang layers, 20 % hot FFN rows: 4 decode steps against the oracle's sparse forward on the same device weights"""
    cfg = dict(synth.CONFIGS["qwen3-0.6b"])
    m = synth.build_on_gpu(cfg, seed=55, layer_type=L.BOOL1, head_type=L.BF16)
    om = O.from_device_model(m)
    for l in range(cfg["n_layer"]):
        hot = hot_mask(cfg["ffn"], 0.2, seed=5 + l)
        m.set_hot(l, hot)
        om.set_hot(l, hot)
    ids = np.random.default_rng(9).integers(0, cfg["vocab"], size=4)
    for pos, tok in enumerate(ids):
        g_next, g_logits = m.forward(int(tok), pos)
        o_next, o_logits, _ = om.decode(int(tok), pos)
        gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
        assert np.abs(gl - ol).max() <= 2.0 ** -6 * np.abs(ol).max(), "pos %d" % pos
        top2 = np.sort(ol)[-2:]
        if top2[1] - top2[0] > 2.0 ** -7 * np.abs(ol).max():
            assert g_next == o_next
    sparse_bytes = m.step_bytes(100)
    for l in range(cfg["n_layer"]):
        m.set_hot(l, None)
    assert sparse_bytes < m.step_bytes(100), "cold rows must not be counted (nor read)"
    om.close()
    m.close()


@pytest.mark.parametrize("layer_type", [L.BOOL1, L.T_SIGN, L.Q4])
@pytest.mark.parametrize("hot_frac", [0.2, 1.0])
def test_sparse_engine_equals_per_layer_launches_and_the_oracle_bit_for_bit(layer_type, hot_frac):
    _sparse_engine_case(dict(synth.CONFIGS["small"], max_seq=320), layer_type, hot_frac)


def test_sparse_engine_on_the_1p7b_shape():
    """the same through the engine's Qwen3-1.7B instantiation (two layers, vocab 4096): its gate | up blocks are all dequantised behind the hand-off (MvPhase::AH = 0), the hot
    bits mask them there"""
    _sparse_engine_case(dict(synth.CONFIGS["qwen3-1.7b"], n_layer=2, vocab=4096, max_seq=320), L.Q4, 0.2)


def _sparse_engine_case(cfg, layer_type, hot_frac):
    """The engine's sparse / 1-bit / 2-bit forms (round 4): 1-bit and ternary PackedQ layers through the LDS selector tables inside the persistent launch, CS_Picker's hot[] as per-workgroup hot bits
    (cold gate / up rows never read, zeros published by the owning workgroup -- no cold-fill launch).  Canonical order, teacher-forced steps across the single- and multi-slice
    attention forms: logits, ids and K / V rows equal the per-layer masked launches' AND the oracle's sparse forward, bit for bit."""
    raw = synth.raw_weights_numpy(cfg, 31, w_std=0.1 if cfg["dim"] <= 1024 else 0.05)
    n = 230
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:n] = prompt_ids(cfg, n, seed=13)
    masks = [hot_mask(cfg["ffn"], hot_frac, seed=5 + l) if hot_frac < 1.0 else None for l in range(cfg["n_layer"])]
    res = {}
    for engine in (True, False):
        m = synth.build_from_raw(cfg, raw, layer_type, L.BF16)
        m.set_canonical(True)
        m.set_engine(engine)
        for l, hm in enumerate(masks):
            if hm is not None:
                m.set_hot(l, hm)
        m.set_forced(forced)
        m.set_state(int(forced[0]), 0)
        logits = []
        for p0 in (0, 37, 190, 200):   # runs of steps (several per launch through the engine) ending where logits are compared
            p1 = {0: 37, 37: 190, 190: 200, 200: n}[p0]
            m.run_steps(p0, p1 - p0, use_graph=True)
            m.sync()
            logits.append(m.logits().copy())
        m.engine_check()
        assert (m.engine_steps() > 0) == engine, m.engine_why()
        k, v = m.kv_to_host()
        res[engine] = (m.tokens_out(n).tolist(), logits, k[:, :n].copy(), v[:, :n].copy())
        m.close()
    assert res[True][0] == res[False][0]
    for a, b in zip(res[True][1], res[False][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[True][2], res[False][2]) and np.array_equal(res[True][3], res[False][3])
    O.set_order(O.ORDER_CANON)
    try:
        om = oracle_model(cfg, raw, layer_type, L.BF16, attn_mode=O.ATTN_CANON)
        for l, hm in enumerate(masks):
            if hm is not None:
                om.set_hot(l, hm)
        o_ids, o_logits = [], {}
        for p in range(n):
            nxt, lg, _ = om.decode(int(forced[p]), p, want_logits=(p + 1 in (37, 190, 200, n)))
            o_ids.append(int(nxt))
            if lg is not None:
                o_logits[p + 1] = lg
        assert res[True][0] == o_ids
        for lg, end in zip(res[True][1], (37, 190, 200, n)):
            assert np.array_equal(lg, o_logits[end]), "logits of position %d" % (end - 1)
        ok, ov = om.kv()
        assert np.array_equal(res[True][2], ok[:, :n]) and np.array_equal(res[True][3], ov[:, :n])
        om.close()
    finally:
        O.set_order(O.ORDER_DOT16)


def test_full_size_one_bit_sparse_engine_equals_per_layer_launches():
    """BASELINE config 5 at full size through the engine: Qwen3-0.6B shapes, 1-bit layers, 20 % hot FFN rows -- 300 teacher-forced steps (position buckets 64 / 128 / 256 / 512,
    one and several attention slices), engine against per-layer launches: ids, last logits and K / V rows bit for bit, in both summation orders."""
    cfg = dict(synth.CONFIGS["qwen3-0.6b"])
    n = 300
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:n] = np.random.default_rng(21).integers(0, cfg["vocab"], size=n)
    m = synth.build_on_gpu(cfg, seed=55, layer_type=L.BOOL1, head_type=L.BF16)
    for l in range(cfg["n_layer"]):
        m.set_hot(l, hot_mask(cfg["ffn"], 0.2, seed=5 + l))
    m.set_forced(forced)
    for canonical in (True, False):
        m.set_canonical(canonical)
        res = {}
        for engine in (True, False):
            m.set_engine(engine)
            steps0 = max(m.engine_steps(), 0)
            m.set_state(int(forced[0]), 0)
            m.run_steps(0, n, use_graph=True)
            m.sync()
            m.engine_check()
            assert (max(m.engine_steps(), 0) - steps0 > 0) == engine, m.engine_why()
            k, v = m.kv_to_host()
            res[engine] = (m.tokens_out(n).tolist(), m.logits().copy(), k[:, :n].copy(), v[:, :n].copy())
        assert res[True][0] == res[False][0]
        if canonical:   # in the v_dot2c order the engine's fp32 attention sums are its own (tolerances, tests/test_gpu_engine.py)
            assert np.array_equal(res[True][1], res[False][1])
            assert np.array_equal(res[True][2], res[False][2]) and np.array_equal(res[True][3], res[False][3])
    m.close()
