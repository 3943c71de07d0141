"""Bit-exact parity in the canonical order (round 3).  The reference's GEMM order is cuBLASLt's and unspecified (gemm.cu:126), and the kernels'
v_dot2c_f32_bf16 has no bit-exact host model -- so kernels and oracle share ONE order built from single fused multiply-adds (oracle/kf_oracle.c
sections 4c and 6 "CANON"; koifish_amd/csrc/kf_gemv_blocks.h dotp<true>, kf_attn_common.h): per-lane chains + a balanced tree for every mat-vec, an
exact power-of-two softmax with fp64 sums for the decode attention.  Here: every output bit of the mat-vec entries, of the decode attention and of whole
decode steps (logits, greedy ids, KV rows) equals the oracle's -- through the per-layer launches AND the persistent engine, at toy sizes and at the
benchmark's own size (Qwen3-0.6B, 2 k context).  The canonical order is a switch (kf_set_canonical: it costs 4-5 % on the 0.6B step, ~28 % on 32B-sized
mat-vecs); the default forms (v_dot2c_f32_bf16, fp32 softmax) are held to the tolerances of the other test files."""
import numpy as np
import pytest
import torch

from conftest import bf16_t, u16
from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture()
def canon():
    """the oracle in the canonical order; the device side is switched per context / model (kf_set_canonical: the default is the v_dot2c / fp32 forms)"""
    O.set_order(O.ORDER_CANON)
    yield
    O.set_order(O.ORDER_DOT16)


@pytest.fixture()
def cctx(ctx):
    ctx.set_canonical(True)
    yield ctx
    ctx.set_canonical(False)


@pytest.mark.parametrize("type_name", ["Q4", "BF16", "F8E5M2", "T_SIGN", "BOOL1"])
@pytest.mark.parametrize("m,k", [(1024, 1024), (1024, 3072), (4096, 2048), (96, 256), (1536, 5120)])
def test_matvec_is_bit_exact(cctx, canon, type_name, m, k):
    """kf_linear (one token) in the canonical order: EVERY output equals the oracle's, whatever the storage (the same lanes per row, per-lane chain and tree)"""
    ctx = cctx
    t = getattr(L, type_name)
    rng = np.random.default_rng(m * 7 + k)
    w = O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    ow = O.quantize(w, m, k, t)
    dw = ctx.upload_blob(t, m, k, ow.blob())
    y = u16(ctx.linear(dw, bf16_t(x, ctx.device)))
    ref = O.linear(ow, x)
    assert np.array_equal(y, ref), "%s %dx%d: %d of %d outputs differ" % (type_name, m, k, int((y != ref).sum()), m)


@pytest.mark.parametrize("m,k", [(96, 25600), (160, 16384), (2, 25600), (2048, 25600), (5120, 25600)])
def test_long_4bit_rows_through_two_fp32_windows_are_bit_exact(cctx, canon, m, k):
    """4-bit rows too long for fp32 activations in 48 KiB of LDS (the 25600-wide down_proj of Qwen3-32B): half the block columns staged at a time, the lane's chains
    continued over the second window (gemv_kernel, XF2) -- every output equals the oracle's, and equals the form with bf16 activations widened per product (knob off)"""
    import ctypes as C
    ctx = cctx
    rng = np.random.default_rng(m + k)
    w = O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    ow = O.quantize(w, m, k, L.Q4)
    dw = ctx.upload_blob(L.Q4, m, k, ow.blob())
    y = u16(ctx.linear(dw, bf16_t(x, ctx.device)))
    ref = O.linear(ow, x)
    assert np.array_equal(y, ref), "%dx%d: %d of %d outputs differ" % (m, k, int((y != ref).sum()), m)
    ctx.hip.kfdbg_set_knob.argtypes = [C.c_char_p, C.c_long]
    assert ctx.hip.kfdbg_set_knob(b"gemv_xf2", 0) == 0
    try:
        y0 = u16(ctx.linear(dw, bf16_t(x, ctx.device)))
    finally:
        ctx.hip.kfdbg_set_knob(b"gemv_xf2", 1)
    assert np.array_equal(y0, ref)


def test_fused_entries_are_bit_exact(cctx, canon):
    """fused RMSNorm + Q|K|V (rows of the launch = all three matrices), paired gate / up + SwiGLU, LM head + arg-max"""
    ctx = cctx
    rng = np.random.default_rng(5)
    k = 1024
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    nw = O.f32_to_bf16((1 + rng.normal(0, 0.1, size=k)).astype(np.float32))
    xn = O.rmsnorm(x, nw, 1e-6)
    ms = [2048, 1024, 1024]
    ows = [O.quantize(O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32)), m, k, L.Q4) for m in ms]
    dws = [ctx.upload_blob(L.Q4, m, k, ow.blob()) for m, ow in zip(ms, ows)]
    ys = ctx.norm_linear(bf16_t(x, ctx.device), bf16_t(nw, ctx.device), dws, 1e-6)
    with O.canonical(rows=sum(ms)):
        for y, ow in zip(ys, ows):
            assert np.array_equal(u16(y), O.linear(ow, xn))
    f = 3072
    og, ou = (O.quantize(O.f32_to_bf16(rng.normal(0, 0.05, size=(f, k)).astype(np.float32)), f, k, L.Q4) for _ in range(2))
    dg, du = ctx.upload_blob(L.Q4, f, k, og.blob()), ctx.upload_blob(L.Q4, f, k, ou.blob())
    act = u16(ctx.norm_gateup_swiglu(bf16_t(x, ctx.device), bf16_t(nw, ctx.device), dg, du, 1e-6))
    with O.canonical(rows=f):
        assert np.array_equal(act, O.swiglu(O.linear(og, xn), O.linear(ou, xn)))
    v = 20000
    oh = O.quantize(O.f32_to_bf16(rng.normal(0, 0.05, size=(v, k)).astype(np.float32)), v, k, L.BF16)
    dh = ctx.upload_blob(L.BF16, v, k, oh.blob())
    logits, am = ctx.lm_head(dh, bf16_t(xn, ctx.device))
    ref = O.linear(oh, xn)
    assert np.array_equal(u16(logits), ref) and am == O.argmax_bf16(ref)


@pytest.mark.parametrize("n_head,n_kv,hd,pos", [(4, 2, 128, 0), (4, 2, 128, 63), (4, 2, 128, 191), (4, 2, 128, 300), (16, 8, 128, 2047), (8, 8, 64, 700), (8, 1, 128, 1029), (4, 1, 64, 4095), (32, 8, 128, 2047), (64, 8, 128, 4095)])
def test_decode_attention_is_bit_exact(cctx, n_head, n_kv, hd, pos):
    """kf_attn_decode against kfo_attn_decode mode CANON: one slice, several slices with the in-kernel merge, both head sizes, every GQA group size"""
    ctx = cctx
    rng = np.random.default_rng(pos + hd)
    kvd = n_kv * hd
    q = O.f32_to_bf16(rng.normal(0, 1.0, size=n_head * hd).astype(np.float32))
    kc = O.f32_to_bf16(rng.normal(0, 1.0, size=(pos + 1, kvd)).astype(np.float32))
    vc = O.f32_to_bf16(rng.normal(0, 1.0, size=(pos + 1, kvd)).astype(np.float32))
    out = u16(ctx.attn_decode(bf16_t(q, ctx.device), bf16_t(kc, ctx.device), bf16_t(vc, ctx.device), pos, n_head, n_kv, hd))
    ref = O.attn_decode(q, kc, vc, pos, n_head, n_kv, hd, mode=O.ATTN_CANON)
    assert np.array_equal(out, ref), "%d of %d outputs differ" % (int((out != ref).sum()), out.size)


def _steps(m, om, cfg, n, forced, use_graph):
    m.set_forced(forced)
    m.set_state(int(forced[0]), 0)
    for p in range(n):
        m.run_steps(p, 1, use_graph=use_graph)
        m.sync()
        g_id, g_logits = int(m.tokens_out(p + 1)[p]), m.logits()
        o_id, o_logits, _ = om.decode(int(forced[p]), p)
        assert np.array_equal(g_logits, o_logits), "position %d: %d logits differ" % (p, int((g_logits != o_logits).sum()))
        assert g_id == o_id, "position %d: id %d vs %d" % (p, g_id, o_id)


@pytest.mark.parametrize("cfg_name,max_seq,n_steps", [("tiny", 96, 96), ("small", 320, 300), ("tiny", 700, 260)])
@pytest.mark.parametrize("engine", [True, False])
def test_whole_steps_are_bit_exact(canon, cfg_name, max_seq, n_steps, engine):
    """teacher-forced decode through the persistent engine and through the per-layer launches: logits, ids and KV rows bit for bit at every position
    (single-slice and multi-slice attention, hipGraph replay)"""
    cfg = dict(synth.CONFIGS[cfg_name], max_seq=max_seq)
    raw = synth.raw_weights_numpy(cfg, 4321, w_std=0.1)
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:n_steps] = prompt_ids(cfg, n_steps, seed=11)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_engine(engine)
    m.set_canonical(True)
    om = oracle_model(cfg, raw, L.Q4, L.BF16, attn_mode=O.ATTN_CANON)
    _steps(m, om, cfg, n_steps, forced, use_graph=True)
    assert (m.engine_steps() > 0) == engine
    m.engine_check()
    gk, gv = m.kv_to_host()
    ok, ov = om.kv()
    assert np.array_equal(gk[:, :n_steps], ok[:, :n_steps]) and np.array_equal(gv[:, :n_steps], ov[:, :n_steps])
    om.close()
    m.close()


def test_free_running_ids_equal_the_oracles(canon):
    """greedy generation (no teacher forcing): one differing logit bit anywhere would eventually change an id"""
    cfg = dict(synth.CONFIGS["small"], max_seq=256)
    raw = synth.raw_weights_numpy(cfg, 99, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    om = oracle_model(cfg, raw, L.Q4, L.BF16, attn_mode=O.ATTN_CANON)
    m.set_prefill_mode(0)   # token-serial prompt: the batched (MFMA) prefill has its own summation order
    prompt = prompt_ids(cfg, 12)
    assert m.generate(prompt, 200, use_graph=True) == om.generate(prompt.tolist(), 200)
    om.close()
    m.close()


def test_full_size_steps_are_bit_exact(canon):
    """BASELINE config 2 at full size: behind batched prefills of 127 / 1023 / 2042 tokens (the SAME KV rows handed to the oracle) the next positions' logits
    and ids equal the oracle's bit for bit -- every bucket, attention over up to 2048 keys in 32 slices, the engine with its in-launch head and pick."""
    cfg = dict(synth.CONFIGS["qwen3-0.6b"])
    m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
    m.set_canonical(True)
    om = O.from_device_model(m, attn_mode=O.ATTN_CANON)
    om.prepare_fast()
    for P, n_follow in ((127, 3), (1023, 2), (2042, 5)):
        toks = np.random.default_rng(100 + P).integers(0, cfg["vocab"], size=P + n_follow).astype(np.int32)
        m.prefill(toks[:P], want_logits=False)
        gk, gv = m.kv_to_host()
        ok, ov = om.kv()
        ok[:, :P] = gk[:, :P]
        ov[:, :P] = gv[:, :P]
        forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
        forced[:P + n_follow] = toks
        m.set_forced(forced)
        for p in range(P, P + n_follow):
            m.set_state(int(toks[p]), p)
            m.run_steps(p, 1, use_graph=True)
            m.sync()
            g_id, g_logits = int(m.tokens_out(p + 1)[p]), m.logits()
            o_id, o_logits, _ = om.decode(int(toks[p]), p)
            assert np.array_equal(g_logits, o_logits), "position %d: %d of %d logits differ" % (p, int((g_logits != o_logits).sum()), g_logits.size)
            assert g_id == o_id
        gk, gv = m.kv_to_host()
        assert np.array_equal(gk[:, P:P + n_follow], om.kv()[0][:, P:P + n_follow])
    assert m.engine_steps() > 0
    m.engine_check()
    om.close()
    m.close()


def test_library_default_is_the_canonical_order(canon):
    """kf_init hands out a context in the canonical order (round 4: the bit-exact order is the default and the one bench.py times): no switch is touched here, and whole
    decode steps -- persistent engine, in-launch head and pick -- equal the oracle bit for bit."""
    from koifish_amd import runtime
    saved, runtime.DEFAULT_CANONICAL = runtime.DEFAULT_CANONICAL, None   # conftest starts this suite's contexts in the v_dot2c order: take the library as it is
    try:
        c = runtime.Context(0)
        assert c.hip.kf_get_canonical(c.h) == 1
        c.close()
        cfg = dict(synth.CONFIGS["small"], max_seq=320)
        raw = synth.raw_weights_numpy(cfg, 777, w_std=0.1)
        n = 40
        forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
        forced[:n] = prompt_ids(cfg, n, seed=3)
        m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    finally:
        runtime.DEFAULT_CANONICAL = saved
    om = oracle_model(cfg, raw, L.Q4, L.BF16, attn_mode=O.ATTN_CANON)
    _steps(m, om, cfg, n, forced, use_graph=True)
    assert m.engine_steps() > 0
    m.engine_check()
    om.close()
    m.close()


def test_full_size_sixteen_step_launch_is_bit_exact(canon):
    """The form bench.py times: ONE launch of 16 decode steps at positions 2028 .. 2043 of Qwen3-0.6B (kf_engine_steps_head: the picked id reaches the next step's
    embedding read as a tagged granule), free running behind a 2028-token prefill -- the 16 greedy ids, the last step's logits and the 16 new KV rows equal the oracle's."""
    cfg = dict(synth.CONFIGS["qwen3-0.6b"])
    m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16, head_std=0.1)
    m.set_canonical(True)
    om = O.from_device_model(m, attn_mode=O.ATTN_CANON)
    om.prepare_fast()
    P, n = 2028, 16
    toks = np.random.default_rng(2028).integers(0, cfg["vocab"], size=P + 1).astype(np.int32)
    m.prefill(toks[:P], want_logits=False)
    gk, gv = m.kv_to_host()
    ok, ov = om.kv()
    ok[:, :P] = gk[:, :P]
    ov[:, :P] = gv[:, :P]
    m.set_forced(np.full(cfg["max_seq"], -1, dtype=np.int32))
    m.set_state(int(toks[P]), P)
    steps0 = m.engine_steps()
    m.run_steps(P, n, use_graph=True)   # one position bucket: one launch of 16 steps
    m.sync()
    m.engine_check()
    assert m.engine_steps() - steps0 == n
    g_ids = m.tokens_out(P + n)[P:P + n].tolist()
    tok, o_ids, o_logits = int(toks[P]), [], None
    for p in range(P, P + n):
        o_id, o_logits, _ = om.decode(tok, p)
        o_ids.append(int(o_id))
        tok = int(o_id)
    assert g_ids == o_ids
    assert np.array_equal(m.logits(), o_logits)
    gk, gv = m.kv_to_host()
    assert np.array_equal(gk[:, P:P + n], om.kv()[0][:, P:P + n]) and np.array_equal(gv[:, P:P + n], om.kv()[1][:, P:P + n])
    om.close()
    m.close()


def test_qwen3_1p7b_shaped_engine_steps_equal_the_oracle(canon):
    """Three layers of the Qwen3-1.7B shape (vocab 4096 so that the oracle steps in milliseconds) through the engine's 2048-wide instantiation: teacher-forced positions
    0..40 and, behind a token-serial stretch, 300..305 -- logits, ids and KV rows equal the oracle's bit for bit."""
    cfg = dict(synth.CONFIGS["qwen3-1.7b"], n_layer=3, vocab=4096, max_seq=320)
    raw = synth.raw_weights_numpy(cfg, 1717, w_std=0.05)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    om = oracle_model(cfg, raw, L.Q4, L.BF16, attn_mode=O.ATTN_CANON)
    toks = prompt_ids(cfg, 320, seed=5)
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:320] = toks
    m.set_forced(forced)
    m.set_state(int(toks[0]), 0)
    for p in range(306):
        m.run_steps(p, 1, use_graph=True)
        m.sync()
        o_id, o_logits, _ = om.decode(int(toks[p]), p)
        if p <= 40 or p >= 300:
            g_logits = m.logits()
            assert np.array_equal(g_logits, o_logits), "position %d: %d logits differ" % (p, int((g_logits != o_logits).sum()))
            assert int(m.tokens_out(p + 1)[p]) == o_id
    gk, gv = m.kv_to_host()
    assert np.array_equal(gk[:, :306], om.kv()[0][:, :306]) and np.array_equal(gv[:, :306], om.kv()[1][:, :306])
    assert m.engine_steps() > 0, m.engine_why()
    m.engine_check()
    om.close()
    m.close()
