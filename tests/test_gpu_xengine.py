"""Eight independent decoders on one GPU, one per XCD (kf_xengine_*, koifish::XcdReplicas; round 5).  The reference decodes ONE sequence per process
(Fish::Chat, GoPT.cpp:1139-1180); the replicas here share a model's weights and nothing else.  Parity: EVERY sequence's greedy ids, logits and K / V rows equal what the
oracle gives for that sequence alone, bit for bit (canonical order) -- different prompts per sequence, sequences standing at different positions, several steps per launch --
and, at the benchmark's own size (Qwen3-0.6B, 2 k context), what the single-sequence persistent engine (itself proven against the oracle in test_gpu_canonical.py) and the
oracle produce."""
import numpy as np
import pytest

from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd.runtime import XcdReplicas
from oracle import oracle as O

pytestmark = pytest.mark.gpu

GQA4_SHAPES = {  # cases/tutorial/history.md:4-6 -- the models the reference lists beside 0.6B / 1.7B / 32B
    "qwen3-4b": dict(dim=2560, n_layer=3, n_head=32, n_kv=8, head_dim=128, ffn=9728, vocab=4096, max_seq=160, theta=1e6, tied=True),
    "qwen3-8b": dict(dim=4096, n_layer=3, n_head=32, n_kv=8, head_dim=128, ffn=12288, vocab=4096, max_seq=160, theta=1e6, tied=True),
}


@pytest.fixture()
def canon():
    O.set_order(O.ORDER_CANON)
    yield
    O.set_order(O.ORDER_DOT16)


_OM = {}


def _oracle_run(cfg, raw, forced, n, pos0=0, layer_type=L.Q4):
    """the oracle alone on one sequence: teacher-forced where forced >= 0, free running elsewhere -> ids per position, last logits, K / V.  One oracle model per (cfg, weights)
    serves every sequence of a test (a sequence starts at position 0 and rewrites the cache rows as it goes): quantising the weights once instead of once per sequence"""
    key = (layer_type, raw["embed"][:2, :32].tobytes(), raw["layers"][0]["q"][:2, :32].tobytes(), tuple(sorted((k, str(v)) for k, v in cfg.items())))
    if key not in _OM:
        for o in _OM.values():
            o.close()
        _OM.clear()
        _OM[key] = oracle_model(cfg, raw, layer_type, L.BF16, attn_mode=O.ATTN_CANON)
    om = _OM[key]
    ids, logits, tok = [], None, int(forced[0])
    for p in range(pos0 + n):
        if forced[p] >= 0:
            tok = int(forced[p])
        o_id, logits, _ = om.decode(tok, p)
        ids.append(int(o_id))
        tok = int(o_id)
    k, v = om.kv()
    k, v = k.copy(), v.copy()
    return ids, np.array(logits, copy=True), k, v


SLOW = pytest.mark.slow


@pytest.mark.parametrize("cfg_name,max_seq,n_steps,n_seq,spl,form,check", [
    ("tiny", 96, 90, 8, 7, 0, None), ("small", 320, 150, 8, 32, 0, (0, 2, 5, 7)), ("tiny", 700, 300, 3, 16, 0, None), ("tiny", 700, 200, 11, 9, 0, None), ("tiny", 700, 130, 27, 9, 0, (0, 4, 9, 13, 18, 22, 26)),
    ("small", 320, 150, 16, 32, 0, (0, 3, 6, 9, 12, 15)), ("small", 320, 150, 32, 32, 0, (0, 5, 10, 15, 17, 20, 27, 30)), ("tiny", 700, 130, 11, 9, 1, None),
    pytest.param("small", 320, 150, 8, 32, 0, None, marks=SLOW), pytest.param("tiny", 700, 130, 27, 9, 0, None, marks=SLOW), pytest.param("small", 320, 150, 16, 32, 0, None, marks=SLOW), pytest.param("small", 320, 150, 32, 32, 0, None, marks=SLOW), pytest.param("small", 320, 150, 16, 32, 1, None, marks=SLOW)])
def test_every_sequence_equals_the_oracle(canon, cfg_name, max_seq, n_steps, n_seq, spl, form, check):
    """different prompts per sequence (the first 12 .. 40 ids forced, then free running), several steps per launch: ids at every position, the last logits and all K / V rows
    of every sequence against the oracle run on that sequence alone; n_seq < 8 leaves XCDs idle; n_seq > 8 (round 6): every decoder multiplies each unpacked block against 2
    (<= 16) or 4 (<= 32) sequences' activations (11, 27: some decoders carry one sequence fewer); form 1: the round-5 form of 9 .. 16 sequences, two decoders per XCD.
    check: the sequences compared with the oracle (None: all; the default run checks a spread over the XCDs and over a decoder's places on the 3-layer shape, --kf-slow all)"""
    cfg = dict(synth.CONFIGS[cfg_name], max_seq=max_seq)
    raw = synth.raw_weights_numpy(cfg, 4321, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    xr = XcdReplicas(m, n_seq)
    if form:
        xr.variant(-2, 1)
    xr.set_steps_per_launch(spl)
    forced = []
    for s in range(n_seq):
        f = np.full(max_seq, -1, dtype=np.int32)
        npr = 12 + 4 * (s % 16)
        f[:npr] = prompt_ids(cfg, npr, seed=100 + s)
        forced.append(f)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    xr.run_steps(n_steps)
    m.sync()
    xr.check()
    for s in (range(n_seq) if check is None else check):
        o_ids, o_logits, ok, ov = _oracle_run(cfg, raw, forced[s], n_steps)
        g_ids = xr.tokens_out(s, n_steps).tolist()
        assert g_ids == o_ids, "sequence %d: first differing position %d" % (s, next(i for i, (a, b) in enumerate(zip(g_ids, o_ids)) if a != b))
        assert xr.state(s) == (o_ids[-1], n_steps)
        g_logits = xr.logits(s)
        assert np.array_equal(g_logits, o_logits), "sequence %d: %d logits differ" % (s, int((g_logits != o_logits).sum()))
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :n_steps], ok[:, :n_steps]) and np.array_equal(gv[:, :n_steps], ov[:, :n_steps]), "sequence %d: K / V rows" % s
    xr.close()
    m.close()


@pytest.mark.parametrize("layer_type,cfg_name,max_seq,n_steps,n_seq,check", [
    (L.BOOL1, "tiny", 200, 150, 8, None), (L.BOOL1, "small", 320, 100, 16, (0, 5, 9, 14)), (L.BOOL1, "small", 320, 100, 32, (0, 7, 13, 18, 24, 31)), (L.BOOL1, "tiny", 200, 150, 27, (1, 6, 10, 15, 19, 24)),
    (L.T_SIGN, "tiny", 200, 150, 8, None), (L.T_SIGN, "small", 320, 100, 32, (0, 7, 13, 18, 24, 31)), (L.T_SIGN, "tiny", 200, 100, 11, None)])
def test_one_bit_layers_every_sequence_equals_the_oracle(canon, layer_type, cfg_name, max_seq, n_steps, n_seq, check):
    """Round 6: the XCD-confined engines on 1-bit PackedQ layers (YinYang, groups of 128: BASELINE config 5's storage) -- a lane takes one dword of a 128-element block, the
    block's pair words come out of the LDS selector table once per decoder and every sequence takes its canonical chain pair over them; one, two and four sequences per
    decoder: ids at every position, the last logits and all K / V rows against the oracle run on the sequence alone."""
    cfg = dict(synth.CONFIGS[cfg_name], max_seq=max_seq)
    raw = synth.raw_weights_numpy(cfg, 2468, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, layer_type, L.BF16)   # (T_SIGN: 2-bit ternary, a lane takes one 8-byte half of a 64-element block)
    m.set_canonical(True)
    xr = XcdReplicas(m, n_seq)
    xr.set_steps_per_launch(13)
    forced = []
    for s in range(n_seq):
        f = np.full(max_seq, -1, dtype=np.int32)
        npr = 10 + 3 * (s % 16)
        f[:npr] = prompt_ids(cfg, npr, seed=500 + s)
        forced.append(f)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    xr.run_steps(n_steps)
    m.sync()
    xr.check()
    for s in (range(n_seq) if check is None else check):
        o_ids, o_logits, ok, ov = _oracle_run(cfg, raw, forced[s], n_steps, layer_type=layer_type)
        g_ids = xr.tokens_out(s, n_steps).tolist()
        assert g_ids == o_ids, "sequence %d: first differing position %d" % (s, next(i for i, (a, b) in enumerate(zip(g_ids, o_ids)) if a != b))
        assert np.array_equal(xr.logits(s), o_logits), "sequence %d: logits" % s
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :n_steps], ok[:, :n_steps]) and np.array_equal(gv[:, :n_steps], ov[:, :n_steps]), "sequence %d: K / V rows" % s
    xr.close()
    m.close()


@pytest.mark.parametrize("layer_type,n_seq", [(L.BOOL1, 32), (L.BOOL1, 8), (L.Q4, 16)])
def test_sparse_forward_every_sequence_equals_the_oracle(canon, layer_type, n_seq):
    """BASELINE config 5's forward through the XCD-confined engines (round 6): 1-bit (and 4-bit) layers with a hot-row mask on every layer's FFN (CS_Picker's hot[], 20 % hot,
    seeded per layer; D_matmul_sparse: a cold gate / up row contributes nothing) -- ids, last logits and K / V rows of the checked sequences against the oracle with the same
    masks; the mask set AFTER the replicas object was built (its next use re-creates the engine: Fish::weights_gen)."""
    cfg = dict(synth.CONFIGS["small"], max_seq=160)
    raw = synth.raw_weights_numpy(cfg, 97, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, layer_type, L.BF16)
    m.set_canonical(True)
    xr = XcdReplicas(m, n_seq)
    hots = {}
    for l in range(cfg["n_layer"]):
        hot = np.zeros(cfg["ffn"], dtype=np.int32)
        hot[np.random.default_rng(5 + l).permutation(cfg["ffn"])[: max(int(cfg["ffn"] * 0.2), 16)]] = 1
        m.set_hot(l, hot)
        hots[l] = hot
    n_steps = 70
    forced = []
    for s in range(n_seq):
        f = np.full(cfg["max_seq"], -1, dtype=np.int32)
        f[:9 + s % 7] = prompt_ids(cfg, 9 + s % 7, seed=800 + s)
        forced.append(f)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    xr.run_steps(n_steps)
    m.sync()
    xr.check()
    om = oracle_model(cfg, raw, layer_type, L.BF16, attn_mode=O.ATTN_CANON)
    for l, hot in hots.items():
        om.set_hot(l, hot)
    for s in sorted({0, 3, n_seq // 2 + 1, n_seq - 1}):
        tok, o_ids, o_logits = int(forced[s][0]), [], None
        for p in range(n_steps):
            if forced[s][p] >= 0:
                tok = int(forced[s][p])
            tok, o_logits, _ = om.decode(tok, p)
            o_ids.append(tok)
        assert xr.tokens_out(s, n_steps).tolist() == o_ids, "sequence %d" % s
        assert np.array_equal(xr.logits(s), o_logits), "sequence %d: logits" % s
        ok, ov = om.kv()
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :n_steps], ok[:, :n_steps]) and np.array_equal(gv[:, :n_steps], ov[:, :n_steps]), "sequence %d: K / V rows" % s
    om.close()
    xr.close()
    m.close()


@pytest.mark.parametrize("n_seq", [8, 32])
def test_sequences_at_different_positions(canon, n_seq):
    """the sequences of one launch need not be in step: all eight are first decoded from position 0 (so that every cache holds its sequence's history), then each is put back
    to a start of its own -- 0, 5, 63, 64, 65, 130, 257, 300: on both sides of the 64-key slice boundaries -- and all advance together; the slices of a sequence's attention
    follow ITS position, and the rows past a start are rewritten with the same values"""
    cfg = dict(synth.CONFIGS["tiny"], max_seq=400)
    raw = synth.raw_weights_numpy(cfg, 99, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    together = 40   # (n_seq 32: the four sequences of a decoder stand at four different positions -- their slices, RoPE pairs and cache rows are their own)
    starts = [0, 5, 63, 64, 65, 130, 257, 300] if n_seq == 8 else [(37 * s + 11 * (s // 8)) % 301 for s in range(32)]
    starts[:4] = [0, 5, 63, 64]
    checked = set(range(8)) if n_seq == 8 else {0, 1, 2, 3, 9, 12, 17, 22, 26, 31}   # (the 32-sequence run: a spread over the XCDs and over a decoder's four places)
    xr = XcdReplicas(m, n_seq)
    forced, o = [], []
    for s in range(n_seq):
        f = np.full(400, -1, dtype=np.int32)
        f[:starts[s] + 3] = prompt_ids(cfg, starts[s] + 3, seed=7 + s)
        forced.append(f)
        o.append(_oracle_run(cfg, raw, f, together, pos0=starts[s]) if s in checked else None)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    xr.run_steps(max(starts))
    m.sync()
    for s in range(n_seq):
        xr.set_state(s, int(forced[s][starts[s]]), starts[s])
    xr.run_steps(together)
    m.sync()
    xr.check()
    for s in sorted(checked):
        o_ids, o_logits, ok, ov = o[s]
        n = starts[s] + together
        assert xr.state(s)[1] == n
        g_ids = xr.tokens_out(s, n).tolist()
        assert g_ids[starts[s]:n] == o_ids[starts[s]:n], "sequence %d (start %d)" % (s, starts[s])
        assert np.array_equal(xr.logits(s), o_logits), "sequence %d" % s
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :n], ok[:, :n]) and np.array_equal(gv[:, :n], ov[:, :n])
    xr.close()
    m.close()


@pytest.mark.parametrize("n_seq", [8, 32, pytest.param(16, marks=SLOW)])
def test_full_size_eight_sequences(canon, n_seq):
    """(n_seq 32: the form bench.py's `xcd_replicas` headlines -- four sequences per decoder; 16: two.)
    Qwen3-0.6B at the benchmark's positions: eight sequences with different histories (each prefilled with its own 2028-token prompt through the batched prefill, whose
    K / V rows are copied into the sequence's cache) decode 2028 .. 2043 in one 16-step launch.  Sequence by sequence: the 16 ids, the last logits and the 16 new K / V rows
    equal the single-sequence engine's on the same history; sequences 0 and 5 also equal the oracle's."""
    import torch
    cfg = dict(synth.CONFIGS["qwen3-0.6b"])
    m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16, head_std=0.1)
    m.set_canonical(True)
    P, n = 2028, 16
    osel = (0, 5) if n_seq <= 16 else (0, 21)   # the sequences also compared with the oracle itself (21: the third of its decoder's four)
    kvd = cfg["n_kv"] * cfg["head_dim"]
    xr = XcdReplicas(m, n_seq)
    ref = []
    hip, host = m.hip, m.host
    import ctypes as C
    ctx = C.c_void_p(host.kfh_ctx(m.h))
    free = np.full(cfg["max_seq"], -1, dtype=np.int32)
    for s in range(n_seq):
        toks = np.random.default_rng(3000 + s).integers(0, cfg["vocab"], size=P + 1).astype(np.int32)
        m.prefill(toks[:P], want_logits=False)
        m.sync()
        # the sequence's history: this model's K / V rows 0 .. P-1 into the replica's cache (device to device, layer by layer: same layout [layer][pos][kv_dim])
        nbytes = cfg["n_layer"] * cfg["max_seq"] * kvd * 2
        L.check(hip.kf_d2d(ctx, C.c_void_p(host.kfh_xr_kcache(xr.h, s)), C.c_void_p(host.kfh_kcache(m.h)), C.c_size_t(nbytes)), "kf_d2d")
        L.check(hip.kf_d2d(ctx, C.c_void_p(host.kfh_xr_vcache(xr.h, s)), C.c_void_p(host.kfh_vcache(m.h)), C.c_size_t(nbytes)), "kf_d2d")
        m.sync()
        hist = m.kv_to_host() if s in osel else None
        # the single-sequence engine on the same history
        m.set_forced(free)
        m.set_state(int(toks[P]), P)
        m.run_steps(P, n, use_graph=True)
        m.sync()
        m.engine_check()
        gk, gv = m.kv_to_host()
        ref.append((m.tokens_out(P + n)[P:P + n].tolist(), m.logits().copy(), gk[:, P:P + n].copy(), gv[:, P:P + n].copy(), toks, hist))
        xr.set_forced(s, free)
        xr.set_state(s, int(toks[P]), P)
    assert m.engine_steps() > 0, m.engine_why()
    xr.set_steps_per_launch(n)
    xr.run_steps(n)
    m.sync()
    xr.check()
    for s in range(n_seq):
        ids, logits, rk, rv, toks, hist = ref[s]
        assert xr.tokens_out(s, P + n)[P:P + n].tolist() == ids, "sequence %d: ids" % s
        assert np.array_equal(xr.logits(s), logits), "sequence %d: logits" % s
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, P:P + n], rk) and np.array_equal(gv[:, P:P + n], rv), "sequence %d: K / V rows" % s
    # two of them against the oracle itself
    om = O.from_device_model(m, attn_mode=O.ATTN_CANON)
    om.prepare_fast()
    for s in osel:
        ids, logits, rk, rv, toks, hist = ref[s]
        ok, ov = om.kv()
        ok[:, :P] = hist[0][:, :P]
        ov[:, :P] = hist[1][:, :P]
        tok, o_ids, o_logits = int(toks[P]), [], None
        for p in range(P, P + n):
            o_id, o_logits, _ = om.decode(tok, p)
            o_ids.append(int(o_id))
            tok = int(o_id)
        assert xr.tokens_out(s, P + n)[P:P + n].tolist() == o_ids, "sequence %d vs the oracle: ids" % s
        assert np.array_equal(xr.logits(s), o_logits), "sequence %d vs the oracle: logits" % s
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, P:P + n], om.kv()[0][:, P:P + n]) and np.array_equal(gv[:, P:P + n], om.kv()[1][:, P:P + n])
    om.close()
    xr.close()
    m.close()


@pytest.mark.parametrize("n_seq", [8, 16, 32])
def test_a_finished_or_parked_sequence_does_not_stop_the_others(canon, n_seq):
    """the sequences are independent, so they end at different times.  One sequence stands near the end of its cache: the launch that would leave the cache skips
    THAT sequence (status 64 in its own state word, no shared error), the others decode on, bit for bit the oracle's; a parked sequence is skipped the same way and goes on
    where it stood once it is un-parked."""
    S = 96
    cfg = dict(synth.CONFIGS["tiny"], max_seq=S)
    raw = synth.raw_weights_numpy(cfg, 77, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    xr = XcdReplicas(m, n_seq)
    xr.set_steps_per_launch(10)
    late, parked = 3, n_seq - 2     # sequence `late` is first decoded to position 80 of 96
    forced = []
    for s in range(n_seq):
        f = np.full(S, -1, dtype=np.int32)
        f[:10] = prompt_ids(cfg, 10, seed=300 + s)
        forced.append(f)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    for s in range(n_seq):
        if s != late:
            xr.park(s)
    xr.run_steps(80)                # only `late` moves
    for s in range(n_seq):
        xr.park(s, s == parked)     # everyone back except `parked`
    xr.run_steps(10)                # late: 80 -> 90; the others 0 -> 10
    xr.run_steps(10)                # late would reach 100 > 96: skipped, status 64; the others 10 -> 20
    xr.park(parked, False)
    xr.run_steps(10)                # parked: 0 -> 10; late still skipped; the others 20 -> 30
    m.sync()
    xr.check()                      # no shared error: nothing timed out, nothing was poisoned
    st = xr.status(late)
    assert st[1] == 90 and st[3] == 64, st
    o_ids, _, _, _ = _oracle_run(cfg, raw, forced[late], 90)
    assert xr.tokens_out(late, 90).tolist() == o_ids
    for s in range(n_seq):
        if s == late:
            continue
        n = 10 if s == parked else 30
        assert xr.status(s)[1:] == [n, 0, 0], (s, xr.status(s))
        o_ids, o_logits, ok, ov = _oracle_run(cfg, raw, forced[s], n)
        assert xr.tokens_out(s, n).tolist() == o_ids, "sequence %d" % s
        assert np.array_equal(xr.logits(s), o_logits), "sequence %d" % s
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :n], ok[:, :n]) and np.array_equal(gv[:, :n], ov[:, :n]), "sequence %d" % s
    xr.set_state(late, int(xr.tokens_out(late, 90)[84]), 85)   # re-aimed inside its cache: the status word clears and it decodes again
    xr.run_steps(5)
    m.sync()
    xr.check()
    assert xr.status(late)[1:] == [90, 0, 0]
    xr.close()
    m.close()


@pytest.mark.parametrize("name", ["qwen3-4b", "tiny"])
def test_a_weight_update_reaches_an_existing_replicas_object(canon, name):
    """kfh_weights_changed / a weight set again on the Fish must reach every XcdReplicas built on it -- the GQA-4 forms hold a fused COPY of q | k | v made
    at create time, every form holds the weights' addresses.  The next use re-creates the engine on the weights as they are now; caches and states stay.  Never stale ids."""
    cfg = dict(GQA4_SHAPES[name], n_layer=2, vocab=1024, max_seq=64) if name in GQA4_SHAPES else dict(synth.CONFIGS[name], max_seq=64)
    raw = synth.raw_weights_numpy(cfg, 5, w_std=0.05 if name in GQA4_SHAPES else 0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    n_seq = 8
    xr = XcdReplicas(m, n_seq)
    forced = []
    for s in range(n_seq):
        f = np.full(cfg["max_seq"], -1, dtype=np.int32)
        f[:8] = prompt_ids(cfg, 8, seed=40 + s)
        forced.append(f)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    xr.run_steps(12)
    m.sync()
    xr.check()
    raw2 = synth.raw_weights_numpy(cfg, 6, w_std=0.05 if name in GQA4_SHAPES else 0.1)   # "a training step": every layer matrix replaced, same shapes
    m2 = synth.build_from_raw(cfg, raw2, L.Q4, L.BF16)
    for key, w in m2.weights.items():
        m.set_weight(key[0], key[1], w)
    for key, nw in m2._norms.items():
        m.set_norm(key[0], key[1], nw)
    if cfg.get("tied", True):
        m.tie_head()
    for s in range(n_seq):
        xr.set_state(s, int(forced[s][0]), 0)
    xr.run_steps(12)                # through the PRE-EXISTING object
    m.sync()
    xr.check()
    for s in (0, 5):
        o_ids, o_logits, ok, ov = _oracle_run(cfg, raw2, forced[s], 12)
        assert xr.tokens_out(s, 12).tolist() == o_ids, "sequence %d decoded stale weights" % s
        assert np.array_equal(xr.logits(s), o_logits)
    xr.close()
    m.close()
    m2.close()


def test_refusals():
    """what the XCD-confined engines do not serve is refused with a reason, never silently routed elsewhere: other shapes, other storages, more than eight sequences, the
    v_dot2c order"""
    cfg = dict(synth.CONFIGS["tiny"], max_seq=96)
    raw = synth.raw_weights_numpy(cfg, 1, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.BF16, L.BF16)
    with pytest.raises(L.KFError) as e:
        XcdReplicas(m, 8)
    assert "4-bit" in str(e.value)
    m.close()
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    with pytest.raises(L.KFError):
        XcdReplicas(m, 33)
    m.set_canonical(False)
    xr = XcdReplicas(m, 2)
    xr.set_state(0, 1, 0), xr.set_state(1, 2, 0)
    with pytest.raises(L.KFError) as e:
        xr.run_steps(1)
    assert "canonical" in str(e.value)
    xr.close()
    m.close()
    cfg4 = dict(GQA4_SHAPES["qwen3-4b"], n_layer=1, vocab=512)    # a GQA-4 shape: one decoder per XCD only
    m = synth.build_from_raw(cfg4, synth.raw_weights_numpy(cfg4, 3, w_std=0.05), L.Q4, L.BF16)
    with pytest.raises(L.KFError) as e:
        XcdReplicas(m, 9)
    assert "at most 8 sequences" in str(e.value)
    m.close()
    cfg2 = dict(synth.CONFIGS["qwen3-1.7b"], n_layer=2, vocab=4096, max_seq=128, ffn=4096)   # no such shape
    m = synth.build_from_raw(cfg2, synth.raw_weights_numpy(cfg2, 2, w_std=0.05), L.Q4, L.BF16)
    with pytest.raises(L.KFError) as e:
        XcdReplicas(m, 8)
    assert "not instantiated" in str(e.value)
    m.close()


@pytest.mark.parametrize("n_seq", [8, 16])
def test_qwen3_1p7b_shape_equals_the_oracle(canon, n_seq):
    """three layers of the Qwen3-1.7B shape (dim 2048, ffn 6144; vocab 4096 so that the oracle steps in milliseconds) through the XCD-confined engines: eight sequences (one
    decoder per XCD) and sixteen (two per XCD: the attention sums inside the second activation buffer, so that two workgroups fit a CU's LDS at full depth), ids at every
    position, last logits and K / V rows against the oracle"""
    cfg = dict(synth.CONFIGS["qwen3-1.7b"], n_layer=3, vocab=4096, max_seq=160)
    raw = synth.raw_weights_numpy(cfg, 1717, w_std=0.05)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    n_steps = 72
    xr = XcdReplicas(m, n_seq)
    forced = []
    for s in range(n_seq):
        f = np.full(160, -1, dtype=np.int32)
        f[:10 + s] = prompt_ids(cfg, 10 + s, seed=40 + s)
        forced.append(f)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    xr.run_steps(n_steps)
    m.sync()
    xr.check()
    for s in (0, n_seq - 5):   # two decoders; with 16 sequences the second place of a decoder
        o_ids, o_logits, ok, ov = _oracle_run(cfg, raw, forced[s], n_steps)
        assert xr.tokens_out(s, n_steps).tolist() == o_ids, "sequence %d" % s
        assert np.array_equal(xr.logits(s), o_logits)
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :n_steps], ok[:, :n_steps]) and np.array_equal(gv[:, :n_steps], ov[:, :n_steps])
    xr.close()
    m.close()




@pytest.mark.parametrize("name,variant,check", [("qwen3-4b", None, (0, 5)), ("qwen3-8b", None, (0, 5)), pytest.param("qwen3-4b", (8, 8), (0, 2, 5, 7), marks=SLOW),
                                                pytest.param("qwen3-8b", None, (0, 2, 5, 7), marks=SLOW)])
def test_gqa4_shapes_equal_the_per_layer_launches_and_the_oracle(canon, name, variant, check):
    """three layers of the Qwen3-4B / Qwen3-8B shapes (32 query heads on 8 kv-heads: four query heads per key tile; 24 of the 32 workgroups own q | k | v rows; the 9728- /
    12288-wide SwiGLU vector staged in pieces; 8B: the attention sums inside the second activation buffer) through the XCD-confined engines: eight sequences, ids at every
    position, last logits and K / V rows against the oracle; sequence 0 also through the per-layer launches of the same library.  variant: the 8-wave form (256 registers,
    contiguous row runs per wave) instead of the default 12-wave one; check: the sequences compared with the oracle (two by default, four with --kf-slow: the oracle at
    these widths is most of the test's time)"""
    cfg = dict(GQA4_SHAPES[name])
    raw = synth.raw_weights_numpy(cfg, 4040, w_std=0.04)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    n_seq, n_steps = 8, 64
    xr = XcdReplicas(m, n_seq)
    forced = []
    for s in range(n_seq):
        f = np.full(160, -1, dtype=np.int32)
        f[:9 + 2 * s] = prompt_ids(cfg, 9 + 2 * s, seed=70 + s)
        forced.append(f)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    xr.set_steps_per_launch(16)
    if variant:
        xr.variant(*variant)
    xr.run_steps(n_steps)
    m.sync()
    xr.check()
    for s in check:
        o_ids, o_logits, ok, ov = _oracle_run(cfg, raw, forced[s], n_steps)
        assert xr.tokens_out(s, n_steps).tolist() == o_ids, "sequence %d" % s
        assert np.array_equal(xr.logits(s), o_logits)
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :n_steps], ok[:, :n_steps]) and np.array_equal(gv[:, :n_steps], ov[:, :n_steps])
        if s == 0:
            ids0, logits0 = o_ids, o_logits
    xr.close()
    # the per-layer launches (no persistent engine serves these shapes): the same ids and logits
    m.set_forced(forced[0])
    m.set_state(int(forced[0][0]), 0)
    for p in range(n_steps):
        m.run_steps(p, 1, use_graph=False)
    m.sync()
    assert m.tokens_out(n_steps)[:n_steps].tolist() == ids0
    assert np.array_equal(m.logits(), logits0)
    m.close()


def test_eight_query_heads_on_one_kv_head(canon):
    """8 query heads on ONE kv-head (the head geometry of a TP = 8 rank of Qwen3-32B) through the XCD-confined engines: the heads in two groups of four, each group's
    workgroups over the same 16 key slices; 20 of the 32 workgroups own q | k | v rows (8 + 1 + 1 parts).  Eight sequences against the oracle: ids, last logits, K / V rows"""
    cfg = dict(dim=256, n_layer=3, n_head=8, n_kv=1, head_dim=64, ffn=512, vocab=512, max_seq=700, theta=1e6, tied=True)
    raw = synth.raw_weights_numpy(cfg, 8181, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    n_seq, n_steps = 8, 400
    xr = XcdReplicas(m, n_seq)
    forced = []
    for s in range(n_seq):
        f = np.full(700, -1, dtype=np.int32)
        f[:11 + 3 * s] = prompt_ids(cfg, 11 + 3 * s, seed=300 + s)
        forced.append(f)
        xr.set_forced(s, f)
        xr.set_state(s, int(f[0]), 0)
    xr.set_steps_per_launch(25)
    xr.run_steps(n_steps)
    m.sync()
    xr.check()
    for s in (0, 4, 7):
        o_ids, o_logits, ok, ov = _oracle_run(cfg, raw, forced[s], n_steps)
        assert xr.tokens_out(s, n_steps).tolist() == o_ids, "sequence %d" % s
        assert np.array_equal(xr.logits(s), o_logits)
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :n_steps], ok[:, :n_steps]) and np.array_equal(gv[:, :n_steps], ov[:, :n_steps])
    xr.close()
    m.close()


def test_prefill_then_decode_per_sequence(canon):
    """prefill + decode for the replicas: every sequence's prompt (different lengths: 40 ... 250 tokens, the longer ones on the tile-GEMM routes) goes through the model's own
    batched prefill (XcdReplicas.prefill -> Fish::Prefill), its K / V rows into the sequence's cache; then all sequences decode together.  Against the model ALONE doing the same
    for that sequence (prefill, then the single-sequence decode in the canonical order): the prompt's K / V rows, every generated id, the last logits and the generated
    positions' K / V rows bit for bit."""
    cfg = dict(synth.CONFIGS["small"], max_seq=320)
    raw = synth.raw_weights_numpy(cfg, 777, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    n_seq, n_new = 8, 40
    xr = XcdReplicas(m, n_seq)
    prompts = [prompt_ids(cfg, 40 + 30 * s, seed=900 + s) for s in range(n_seq)]
    for s in range(n_seq):
        xr.prefill(s, prompts[s])
    xr.set_steps_per_launch(8)
    xr.run_steps(n_new)
    m.sync()
    xr.check()
    for s in (0, 3, 7):
        P = len(prompts[s])
        nxt, _ = m.prefill(prompts[s], want_logits=False)
        m.set_forced(np.full(cfg["max_seq"], -1, dtype=np.int32))
        m.run_steps(P, n_new, use_graph=False)
        m.sync()
        ref_ids = m.tokens_out(P + n_new)
        got = xr.tokens_out(s, P + n_new)
        assert got[P - 1:].tolist() == ref_ids[P - 1:].tolist(), "sequence %d" % s
        assert np.array_equal(xr.logits(s), m.logits())
        rk, rv = m.kv_to_host()
        gk, gv = xr.kv_to_host(s)
        assert np.array_equal(gk[:, :P + n_new], rk[:, :P + n_new]) and np.array_equal(gv[:, :P + n_new], rv[:, :P + n_new])
    xr.close()
    m.close()


@pytest.mark.parametrize("n_seq,n_req", [(8, 21), (32, 45)])
def test_a_queue_of_prompts_through_the_slots(canon, n_seq, n_req):
    """XcdReplicas.chat: Fish::Chat's rounds over a prompt list (GoPT.cpp:1111-1180) with n_seq rounds in flight -- ragged prompts (2 ... 40 tokens, one standing five rows
    before the cache's end), a free slot refilled from the queue while the others decode on.  Every answer equals the model ALONE generating on that prompt (same batched
    prefill, the single-sequence decode); with an EOS id the answers are those cut behind their first EOS, and the ids a sequence decoded past it are dropped; with the
    reference's sampler every answer equals the model alone sampling under the request's seed."""
    cfg = dict(synth.CONFIGS["small"], max_seq=96)
    raw = synth.raw_weights_numpy(cfg, 4242, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    rng = np.random.default_rng(5)
    prompts = [prompt_ids(cfg, int(rng.integers(2, 41)), seed=300 + r) for r in range(n_req)]
    prompts[3] = prompt_ids(cfg, cfg["max_seq"] - 5, seed=77)   # room for 5 rows behind the prompt: 6 ids
    max_new = 12
    m.set_prefill_mode(1)   # the batched prefill XcdReplicas.prefill uses
    ref = []
    for p in prompts:
        ref.append(m.generate(p, min(max_new, cfg["max_seq"] - len(p) + 1), use_graph=False))
    assert len(ref[3]) == 6
    xr = XcdReplicas(m, n_seq)
    xr.set_steps_per_launch(5)
    got, st = xr.chat(prompts, max_new)
    assert got == ref
    assert st["prefills"] == n_req and st["dropped"] == 0 and st["launches"] >= 3
    # an EOS id: the most frequent id behind the answers' second position
    ids, cnt = np.unique(np.concatenate([np.array(a[2:]) for a in ref]), return_counts=True)
    eos = int(ids[cnt.argmax()])
    cut = [a[:a.index(eos) + 1] if eos in a else a for a in ref]
    assert any(len(c) < len(a) for c, a in zip(cut, ref))
    got, st = xr.chat(prompts, max_new, eos=eos)
    assert got == cut
    assert st["prefills"] == n_req
    # a limit of its own per request: the answers are the common run's, cut; slots free up at scattered steps and are refilled while the others decode on
    each = [1 + (5 * r) % max_new for r in range(n_req)]
    got, st = xr.chat(prompts, max_new, max_new_each=each)
    assert got == [a[:k] for a, k in zip(ref, each)]
    assert st["prefills"] == n_req and st["dropped"] == 0
    with pytest.raises(Exception):
        xr.chat(prompts[:2], 4, max_new_each=[4, 5])
    # the reference's sampler (GeneratOnPrompt::Sample: temperature, top-k, top-p, xorshift coin) per slot: request r draws with seed + r
    sub = list(range(0, n_req, 2))
    xr.set_sampler(temperature=0.8, top_p=0.9, top_k=40, seed=1000)
    got, st = xr.chat([prompts[r] for r in sub], max_new)
    for i, r in enumerate(sub):
        m.set_sampler(temperature=0.8, top_p=0.9, top_k=40, seed=1000 + i)
        want = m.generate(prompts[r], min(max_new, cfg["max_seq"] - len(prompts[r]) + 1), use_graph=False)
        assert got[i] == want, "request %d (sampled)" % i
    assert any(g != ref[r] for g, r in zip(got, sub)), "the sampler drew the greedy ids everywhere"
    m.set_sampler()
    xr.set_sampler()
    got, _ = xr.chat(prompts[:5], max_new)
    assert got == ref[:5]
    # prompts prefilled together when several slots are free (set_prefill_batch): one token batch per refill -- the batch sums in MFMA order over more rows, so the bar is
    # tests/test_gpu_prefill.py's (ids identical on the committed seeds), not bits
    xr.set_prefill_batch(8)
    got, st = xr.chat(prompts, max_new)
    assert st["prefills"] == n_req and [len(a) for a in got] == [len(a) for a in ref]
    same = sum(a == b for a, b in zip(got, ref))
    assert same >= 0.8 * n_req, "%d of %d answers equal the one-by-one prefill's" % (same, n_req)   # a near-tie may flip under the other summation order (w_std 0.1: spiky logits)
    again, _ = xr.chat(prompts, max_new)
    assert again == got                                                                             # the same batches, the same bits
    xr.set_prefill_batch(1)
    # the object is as before the queue: all slots free running
    for s in range(n_seq):
        assert xr.status(s)[2] == 0
    xr.close()
    m.close()


@pytest.mark.parametrize("lens,slots", [((5, 40, 17, 64, 33, 8), (3, 0, 7, 5, 1, 6)), ((40,) * 8, tuple(range(8))), ((100, 90, 3), (2, 6, 4))])
def test_prefill_batch_vs_the_oracle(lens, slots):
    """XcdReplicas.prefill_batch: several prompts as ONE token batch (ragged lengths padded to the longest; 8 x 40 = 320 rows takes the large-batch tile routes) -- a
    prompt's rows attend to that prompt only, positions restart per prompt, every prompt's K / V rows land in ITS slot's cache.  Against the oracle's token-serial forward
    of each prompt alone, at the bar of tests/test_gpu_prefill.py (token batches sum in MFMA order): K / V rows and the last logits within 2^-6 of scale, the picked id equal;
    then all slots decode together, teacher-forced along the oracle's continuation, and the logits six steps behind every prompt meet the same bar.  Slots that took no
    prompt keep their state."""
    cfg = dict(synth.CONFIGS["small"], max_seq=128)
    raw = synth.raw_weights_numpy(cfg, 1234, w_std=0.1)
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_canonical(True)
    xr = XcdReplicas(m, 8)
    idle = [s for s in range(8) if s not in slots]
    for s in idle:
        xr.set_state(s, 7, 0)
        xr.park(s)
    prompts = [prompt_ids(cfg, n, seed=40 + i) for i, n in enumerate(lens)]
    xr.prefill_batch(slots, prompts)
    m.sync()
    n_new = 6
    TOL = 2.0 ** -6
    om = oracle_model(cfg, raw, L.Q4, L.BF16)   # the reference's own order, as tests/test_gpu_prefill.py
    want_ids = {}
    for p, s in zip(prompts, slots):
        n = len(p)
        nxt = lg = None
        for pos, tok in enumerate(p):
            nxt, lg, _ = om.decode(int(tok), pos)
        ok, ov = om.kv()
        gk, gv = xr.kv_to_host(s)
        for g, o in ((gk, ok), (gv, ov)):
            for l in range(cfg["n_layer"]):
                a, b = O.bf16_to_f32(g[l, :n]), O.bf16_to_f32(o[l, :n])
                assert np.abs(a - b).max() <= TOL * np.abs(b).max(), "slot %d layer %d K / V rows" % (s, l)
        gl, ol = O.bf16_to_f32(xr.logits(s)), O.bf16_to_f32(lg)
        assert np.abs(gl - ol).max() <= TOL * np.abs(ol).max(), "slot %d logits" % s
        assert xr.state(s) == (nxt, n), "slot %d: state %s, the oracle picks %d" % (s, xr.state(s), nxt)
        assert int(xr.tokens_out(s, n)[n - 1]) == nxt
        f = np.full(cfg["max_seq"], -1, dtype=np.int32)   # the continuation teacher-forced along the oracle's own ids: the logits behind it read every prompt row of the slot
        tok = nxt
        for k in range(n_new):
            f[n + k] = tok
            tok, lg, _ = om.decode(int(tok), n + k)
        xr.set_forced(s, f)
        want_ids[s] = np.array(lg, copy=True)
    om.close()
    xr.run_steps(n_new)
    m.sync()
    xr.check()
    for p, s in zip(prompts, slots):
        gl, ol = O.bf16_to_f32(xr.logits(s)), O.bf16_to_f32(want_ids[s])
        assert np.abs(gl - ol).max() <= TOL * np.abs(ol).max(), "slot %d: logits %d steps behind the prompt" % (s, n_new)
        assert xr.state(s)[1] == len(p) + n_new
    for s in idle:
        assert xr.status(s)[:3] == [7, 0, 1]
    with pytest.raises(Exception):
        xr.prefill_batch((1, 1), prompts[:2])          # one slot twice
    with pytest.raises(Exception):
        xr.prefill_batch((0,), [prompt_ids(cfg, cfg["max_seq"], seed=1)])   # no row left behind the prompt
    xr.close()
    m.close()
