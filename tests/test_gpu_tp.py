"""Tensor-parallel decode on the GPU: all R ranks of a plan run on ONE MI355X in lock-step (koifish_amd.tp.VirtualTP: the same
per-rank kernels -- row-shard mat-vecs, local-head attention, fp32 column-shard partials -- and the same rank-ordered combine as
the multi-process driver), checked against the oracle's tensor-parallel emulation (kfo_qwen3_set_tp)."""
import numpy as np
import pytest
import torch

from conftest import u16
from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd import tp as TP
from koifish_amd.runtime import Context
from oracle import oracle as O

pytestmark = pytest.mark.gpu
LOGIT_TOL = 2.0 ** -6


def full_weights_on_gpu(ctx, cfg, raw, layer_type=L.Q4, head_type=L.BF16):
    w, norms = {}, {}
    w[(-1, 0)] = ctx.quantize(synth._bf16_t(raw["embed"], ctx.device), head_type)
    w[(-1, 1)] = w[(-1, 0)]
    norms[(-1, 0)] = synth._bf16_t(raw["final_norm"], ctx.device)
    for li, lw in enumerate(raw["layers"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = ctx.quantize(synth._bf16_t(lw[s], ctx.device), layer_type)
        for si, s in enumerate(synth.NORMS):
            norms[(li, si)] = synth._bf16_t(lw[s], ctx.device)
    return w, norms


@pytest.mark.parametrize("cfg_name,world", [("tiny", 2), ("small", 2), ("small", 8)])
def test_virtual_tp_matches_oracle_tp(ctx, cfg_name, world):
    cfg = dict(synth.CONFIGS[cfg_name])
    raw = synth.raw_weights_numpy(cfg, 31, w_std=0.1)
    w, norms = full_weights_on_gpu(ctx, cfg, raw)
    vt = TP.VirtualTP(cfg, w, norms, world, ctx)
    om = oracle_model(cfg, raw, L.Q4, L.BF16, tp=world)
    prompt = prompt_ids(cfg, 10, seed=2)
    tok = int(prompt[0])
    for pos in range(18):
        g_next = vt.step(tok, pos)
        o_next, o_logits, _ = om.decode(tok, pos)
        gl, ol = O.bf16_to_f32(u16(vt.logits())), O.bf16_to_f32(o_logits)
        assert np.abs(gl - ol).max() <= LOGIT_TOL * np.abs(ol).max(), "pos %d" % pos
        assert g_next == o_next, "pos %d: %d vs %d" % (pos, g_next, o_next)
        tok = int(prompt[pos + 1]) if pos + 1 < len(prompt) else o_next
    # the sharded model is the same model: TP=R ids equal the unsharded GPU path's ids on this seed
    gm = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    assert vt.generate(prompt, 8) == gm.generate(prompt, 8)
    gm.close()


def test_qwen3_32b_shapes_tp8_slice(ctx):
    """Qwen3-32B shapes (dim 5120, 64/8 heads, ffn 25600) cut to 2 layers and an 8192-row vocabulary so that the CPU oracle can
    follow ('oracle on a slice'): TP=8 shards = 8 q-heads + 1 kv-head + 3200 ffn rows per rank."""
    cfg = dict(synth.CONFIGS["qwen3-32b"], n_layer=2, vocab=8192, max_seq=48, tied=True)
    g = torch.Generator(device=ctx.device)
    g.manual_seed(5)

    def mat(r, c):
        return (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * 0.05).to(torch.bfloat16)

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)
    w, norms = {}, {}
    w[(-1, 0)] = ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    w[(-1, 1)] = w[(-1, 0)]
    norms[(-1, 0)] = nrm(cfg["dim"])
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = ctx.quantize(mat(*synth.SHAPES[s](cfg)), L.Q4)
        norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(128), nrm(128)

    class Dev:
        pass
    m = Dev()
    m.cfg, m.weights, m._norms = cfg, w, norms
    om = O.from_device_model(m)
    import ctypes as C
    O.lib().kfo_qwen3_set_tp(om.h, 8)
    vt = TP.VirtualTP(cfg, w, norms, 8, ctx)
    assert (vt.plan.n_head_l, vt.plan.n_kv_l, vt.plan.ffn_l) == (8, 1, 3200)
    ids = np.random.default_rng(9).integers(0, cfg["vocab"], size=5)
    for pos, tok in enumerate(ids):
        g_next = vt.step(int(tok), pos)
        o_next, o_logits, _ = om.decode(int(tok), pos)
        gl, ol = O.bf16_to_f32(u16(vt.logits())), O.bf16_to_f32(o_logits)
        assert np.abs(gl - ol).max() <= LOGIT_TOL * np.abs(ol).max(), "pos %d" % pos
        top2 = np.sort(ol)[-2:]
        if top2[1] - top2[0] > 2 * LOGIT_TOL * np.abs(ol).max():
            assert g_next == o_next
    om.close()


# ---------------------------------------------------------------------------------------------- the C++ host's TP step (Fish::TPPhase + kf_tp_*)
@pytest.mark.fast_order   # (the Python-stepped ranks launch q, k, v one by one: in the canonical order their lanes per row are those of THEIR launches, not of the fused one)
@pytest.mark.parametrize("cfg_name,world,use_graph", [("tiny", 2, False), ("small", 2, True), ("small", 8, True)])
def test_native_tp_equals_python_stepped_tp_bit_for_bit(ctx, cfg_name, world, use_graph):
    """The step in the C++ host with the exchange done by kernels (push into every rank's receive area, rank-ordered sum, arg-max pairs the
    same way) against the Python-stepped VirtualTP with its in-process gathers: the same kernels and the same summation order, so logits and
    ids are equal bit for bit -- eager and as one hipGraph per bucket -- and they follow the oracle's TP emulation."""
    cfg = dict(synth.CONFIGS[cfg_name])
    raw = synth.raw_weights_numpy(cfg, 31, w_std=0.1)
    w, norms = full_weights_on_gpu(ctx, cfg, raw)
    vt = TP.VirtualTP(cfg, w, norms, world, ctx)
    nt = TP.NativeTP(cfg, w, norms, world, ctx)
    prompt = prompt_ids(cfg, 10, seed=2)
    tok = int(prompt[0])
    for pos in range(14):
        v_next = vt.step(tok, pos)
        n_next = nt.step(tok, pos, use_graph=use_graph)
        assert n_next == v_next, "pos %d" % pos
        assert np.array_equal(u16(vt.logits()), nt.logits()), "pos %d" % pos
        tok = int(prompt[pos + 1]) if pos + 1 < len(prompt) else v_next
    om = oracle_model(cfg, raw, L.Q4, L.BF16, tp=world)
    ids = nt.generate(prompt, 20, use_graph=True)
    assert ids == vt.generate(prompt, 20)
    assert ids == om.generate(prompt.tolist(), 20)
    # a second run from position 0 reuses the same receive areas: tags never repeat, stale granules are never taken for fresh ones
    assert nt.generate(prompt, 20, use_graph=True) == ids
    assert nt.generate(prompt, 20, use_graph=False) == ids
    nt.close()


def test_native_tp8_on_32b_shaped_slice_long_context(ctx):
    """Qwen3-32B shapes (2 layers, 8192-row vocabulary), TP=8, context 4096: ids at positions {128, 1024, 4095} after teacher-forced runs equal the
    Python-stepped path's; every graph bucket up to 4096 is captured on the way."""
    cfg = dict(synth.CONFIGS["qwen3-32b"], n_layer=2, vocab=8192, max_seq=4096, tied=True)
    g = torch.Generator(device=ctx.device)
    g.manual_seed(5)

    def mat(r, c):
        return (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * 0.05).to(torch.bfloat16)

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)
    w, norms = {}, {}
    w[(-1, 0)] = ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    w[(-1, 1)] = w[(-1, 0)]
    norms[(-1, 0)] = nrm(cfg["dim"])
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = ctx.quantize(mat(*synth.SHAPES[s](cfg)), L.Q4)
        norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(128), nrm(128)
    nt = TP.NativeTP(cfg, w, norms, 8, ctx)
    forced = np.random.default_rng(11).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
    nt.set_forced(forced)
    nt.set_state(int(forced[0]), 0)
    nt.run_steps(0, 4096, use_graph=True)
    nt.check()
    toks = [m.tokens_out(4096) for m in nt.ranks]
    for t in toks[1:]:
        assert np.array_equal(t, toks[0])                                  # every rank picked the same ids all the way
    # the Python-stepped path on the same KV state: copy rank caches and replay single positions
    vt = TP.VirtualTP(cfg, w, norms, 8, ctx)
    for r, (a, b) in enumerate(zip(nt.ranks, vt.ranks)):
        n = cfg["n_layer"] * cfg["max_seq"] * vt.plan.kvd_l * 2
        cx = ctx.h
        import ctypes as C
        L.check(ctx.hip.kf_d2d(cx, C.c_void_p(b.kc.data_ptr()), C.c_void_p(a.host.kfh_kcache(a.h)), C.c_size_t(n)), "d2d")
        L.check(ctx.hip.kf_d2d(cx, C.c_void_p(b.vc.data_ptr()), C.c_void_p(a.host.kfh_vcache(a.h)), C.c_size_t(n)), "d2d")
    ctx.sync()
    for pos in (128, 1024, 4095):
        assert vt.step(int(forced[pos]), pos) == int(toks[0][pos]), "pos %d" % pos
    nt.close()


def test_native_tp8_32b_slice_at_depth_vs_the_oracle_bit_for_bit(ctx):
    """BASELINE config 4 against the ORACLE at depth and at long context (missing 3): a 4-layer Qwen3-32B-shaped slice, TP = 8,
    the native step (kernel-side exchange) run teacher-forced over all 4096 positions; the ranks' KV caches -- rank r holds kv-head r -- are copied into the
    oracle's cache, and at positions 128, 1024 and 4095 the oracle's tensor-parallel emulation (kfo_qwen3_set_tp: row shards as they are, column shards as fp32
    partials summed in rank order) in the canonical summation order must give the ranks' logits and id BIT FOR BIT -- the mat-vec shards (1280-row Q|K|V launch,
    3200-row gate|up, 1024- and 3200-column o_proj / down_proj shards), the canonical attention over up to 4096 keys, the rank-ordered sums, the sharded head."""
    import ctypes as C
    cfg = dict(synth.CONFIGS["qwen3-32b"], n_layer=4, vocab=8192, max_seq=4096, tied=True)
    g = torch.Generator(device=ctx.device)
    g.manual_seed(7)

    def mat(r, c):
        return (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * 0.05).to(torch.bfloat16)

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)
    w, norms = {}, {}
    w[(-1, 0)] = ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    w[(-1, 1)] = w[(-1, 0)]
    norms[(-1, 0)] = nrm(cfg["dim"])
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = ctx.quantize(mat(*synth.SHAPES[s](cfg)), L.Q4)
        norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(128), nrm(128)
    nt = TP.NativeTP(cfg, w, norms, 8, ctx)
    for rk in nt.ranks:
        rk.set_canonical(True)
    forced = np.random.default_rng(13).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
    nt.set_forced(forced)
    nt.set_state(int(forced[0]), 0)
    nt.run_steps(0, 4096, use_graph=True)
    nt.check()

    class Dev:
        pass
    m = Dev()
    m.cfg, m.weights, m._norms = cfg, w, norms
    om = O.from_device_model(m, attn_mode=O.ATTN_CANON)
    O.lib().kfo_qwen3_set_tp(om.h, 8)
    om.prepare_fast()
    ok, ov = om.kv()                                   # [n_layer, max_seq, 8 * 128]
    kvd_l = nt.plan.kvd_l
    assert kvd_l == 128
    n = cfg["n_layer"] * cfg["max_seq"] * kvd_l
    for r, a in enumerate(nt.ranks):
        hk, hv = np.zeros(n, dtype=np.uint16), np.zeros(n, dtype=np.uint16)
        L.check(ctx.hip.kf_d2h(ctx.h, hk.ctypes.data_as(C.c_void_p), C.c_void_p(a.host.kfh_kcache(a.h)), C.c_size_t(n * 2)), "d2h")
        L.check(ctx.hip.kf_d2h(ctx.h, hv.ctypes.data_as(C.c_void_p), C.c_void_p(a.host.kfh_vcache(a.h)), C.c_size_t(n * 2)), "d2h")
        ok[:, :, r * kvd_l:(r + 1) * kvd_l] = hk.reshape(cfg["n_layer"], cfg["max_seq"], kvd_l)
        ov[:, :, r * kvd_l:(r + 1) * kvd_l] = hv.reshape(cfg["n_layer"], cfg["max_seq"], kvd_l)
    O.set_order(O.ORDER_CANON)
    try:
        for pos in (128, 1024, 4095):
            g_id = nt.step(int(forced[pos]), pos, use_graph=True)
            g_logits = nt.logits()
            o_id, o_logits, _ = om.decode(int(forced[pos]), pos)
            assert np.array_equal(g_logits, o_logits), "position %d: %d of %d logits differ" % (pos, int((g_logits != o_logits).sum()), g_logits.size)
            assert g_id == o_id, "position %d" % pos
    finally:
        O.set_order(O.ORDER_DOT16)
    om.close()
    nt.close()


@pytest.mark.parametrize("n_layer", [8, pytest.param(64, marks=pytest.mark.slow)])
def test_native_tp8_full_depth_qwen3_32b_vs_the_oracle(ctx, n_layer):
    """(Default run: 8 of the 64 layers with the full head; --kf-slow: all 64 -- 3.6 minutes.  Round 6, VERDICT r05 item 5b: the SAME weights first through the one-launch
    form -- the eight ranks as the eight XCDs of one launch, koifish::XcdTP, with the 18992-row vocabulary shards of the real head -- then through the per-launch ranks; both
    against the oracle.)
    north_star's second target at FULL DEPTH (missing 5): all 64 layers of the Qwen3-32B shape (dim 5120, 64 / 8 heads of 128, ffn 25600, the 151936-row
    vocabulary: 16.6 GB of 4-bit layers + a 1.56 GB head), tensor parallel TP = 8 as eight virtual ranks on this GPU (the ranks' kernels and the kernel-side exchange; no
    xGMI), decoding a 4-token prompt and then 5 free-running greedy ids -- against the oracle's tensor-parallel emulation (row shards as they are, column shards as fp32
    partials summed in rank order) in the canonical order: every id and the last position's 151936 logits bit for bit."""
    from koifish_amd.runtime import XcdTP
    cfg = dict(synth.CONFIGS["qwen3-32b"], max_seq=64, n_layer=n_layer)
    g = torch.Generator(device=ctx.device)
    g.manual_seed(32)

    def mat(r, c, std=0.05):
        return (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * std).to(torch.bfloat16)

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)
    w, norms = {}, {}
    w[(-1, 0)] = ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    w[(-1, 1)] = ctx.quantize(mat(cfg["vocab"], cfg["dim"], std=0.1), L.BF16)   # untied head (the 32B card)
    norms[(-1, 0)] = nrm(cfg["dim"])
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = ctx.quantize(mat(*synth.SHAPES[s](cfg), std=0.03), L.Q4)
        norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(128), nrm(128)
    nt = TP.NativeTP(cfg, w, norms, 8, ctx)
    for rk in nt.ranks:
        rk.set_canonical(True)
    n_prompt, n_new = 4, 5
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:n_prompt] = np.random.default_rng(64).integers(0, cfg["vocab"], size=n_prompt)
    n = n_prompt + n_new - 1
    xt = XcdTP(nt)                  # the one-launch form first, on the ranks' shards as they are
    xt.set_forced(forced)
    xt.set_state(int(forced[0]), 0)
    xt.set_steps_per_launch(4)
    xt.run_steps(n)
    ctx.sync()
    xt.check()
    x_ids, x_logits = xt.tokens_out(n).tolist(), xt.logits()
    xt.close()
    nt.set_forced(forced)
    nt.set_state(int(forced[0]), 0)
    nt.run_steps(0, n, use_graph=True)
    nt.check()
    toks = [m.tokens_out(n) for m in nt.ranks]
    for t in toks[1:]:
        assert np.array_equal(t, toks[0])
    g_logits = nt.logits()

    class Dev:
        pass
    m = Dev()
    m.cfg, m.weights, m._norms = cfg, w, norms
    om = O.from_device_model(m, attn_mode=O.ATTN_CANON)
    O.lib().kfo_qwen3_set_tp(om.h, 8)
    om.prepare_fast()
    O.set_order(O.ORDER_CANON)
    try:
        tok, o_ids, o_logits = int(forced[0]), [], None
        for pos in range(n):
            if forced[pos] >= 0:
                tok = int(forced[pos])
            o_id, o_logits, _ = om.decode(tok, pos)
            o_ids.append(int(o_id))
            tok = int(o_id)
    finally:
        O.set_order(O.ORDER_DOT16)
    assert toks[0].tolist() == o_ids, (toks[0].tolist(), o_ids)
    assert np.array_equal(g_logits, o_logits), "%d of %d logits differ" % (int((g_logits != o_logits).sum()), g_logits.size)
    assert x_ids == o_ids, ("one-launch form", x_ids, o_ids)
    assert np.array_equal(x_logits, o_logits), "one-launch form: %d of %d logits differ" % (int((x_logits != o_logits).sum()), x_logits.size)
    om.close()
    nt.close()


@pytest.mark.parametrize("name,n", [("qwen3-32b", 36), ("qwen3-8b", 32), ("qwen3-4b", 32), pytest.param("qwen3-32b", 96, marks=pytest.mark.slow)])
def test_tp8_ranks_as_the_eight_xcds_of_one_launch_vs_the_oracle(ctx, name, n):
    """(default run: 36 positions, --kf-slow: 96.  qwen3-8b, round 6: ONE sequence of a GQA-4 model decoded by the eight XCDs as the eight TP ranks -- 4 query heads on 1 kv-head,
    ffn 1536 per rank: the same kernel, another rank shape)
    The TP = 8 ranks of a Qwen3-32B-shaped model as the eight XCDs of ONE launch (kf_xengine_create_tp, koifish::XcdTP; round 5): rank r's 32 workgroups stream rank r's
    shards, q | k | v / attention / gate | up inside the XCD, the o_proj / down_proj partials exchanged between the XCDs inside the kernel and summed in rank order, the head
    in vocabulary shards with a cross-XCD pick, several tokens per launch.  A 3-layer slice decoded from position 0 -- 24 forced ids, then free running -- against the
    oracle's tensor-parallel emulation in the canonical order: every id, the last logits and every K / V row bit for bit; and the same ids from the per-launch rank step
    (NativeTP, the kernels an 8-GPU node runs)."""
    from koifish_amd.runtime import XcdTP
    cfg = dict(synth.CONFIGS[name], n_layer=3, vocab=8192, max_seq=320, tied=True)
    ffn_real = cfg["ffn"]
    if name == "qwen3-4b":   # 9728 = 76 groups of 128 do not split into eight whole-group column shards: the FFN padded to 80 groups (zero rows of gate / up, zero columns of down_proj)
        cfg["ffn"] = 10240
    g = torch.Generator(device=ctx.device)
    g.manual_seed(77)

    def mat(r, c, std=0.05):
        return (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * std).to(torch.bfloat16)

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)
    w, norms = {}, {}
    w[(-1, 0)] = ctx.quantize(mat(cfg["vocab"], cfg["dim"], std=0.1), L.BF16)
    w[(-1, 1)] = w[(-1, 0)]
    norms[(-1, 0)] = nrm(cfg["dim"])
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(synth.SLOTS):
            t = mat(*synth.SHAPES[s](cfg))
            if s in ("gate", "up"):
                t[ffn_real:] = 0
            if s == "down":
                t[:, ffn_real:] = 0
            w[(li, si)] = ctx.quantize(t, L.Q4)
        norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(128), nrm(128)
    nt = TP.NativeTP(cfg, w, norms, 8, ctx)
    for rk in nt.ranks:
        rk.set_canonical(True)
    n_prompt = 24
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:n_prompt] = np.random.default_rng(5).integers(0, cfg["vocab"], size=n_prompt)
    xt = XcdTP(nt)
    xt.set_forced(forced)
    xt.set_state(int(forced[0]), 0)
    xt.set_steps_per_launch(16)
    xt.run_steps(n)
    ctx.sync()
    xt.check()
    ids = xt.tokens_out(n)
    g_logits = xt.logits()
    gk, gv = xt.kv_to_host()                            # [rank][layer][pos][128]
    # the per-launch rank step on the same ids
    nt.set_forced(forced)
    nt.set_state(int(forced[0]), 0)
    nt.run_steps(0, n, use_graph=True)
    nt.check()
    assert nt.ranks[0].tokens_out(n).tolist() == ids.tolist()
    assert np.array_equal(nt.logits(), g_logits)

    class Dev:
        pass
    m = Dev()
    m.cfg, m.weights, m._norms = cfg, w, norms
    om = O.from_device_model(m, attn_mode=O.ATTN_CANON)
    O.lib().kfo_qwen3_set_tp(om.h, 8)
    om.prepare_fast()
    O.set_order(O.ORDER_CANON)
    try:
        tok, o_ids, o_logits = int(forced[0]), [], None
        for pos in range(n):
            if forced[pos] >= 0:
                tok = int(forced[pos])
            o_id, o_logits, _ = om.decode(tok, pos)
            o_ids.append(int(o_id))
            tok = int(o_id)
    finally:
        O.set_order(O.ORDER_DOT16)
    assert ids.tolist() == o_ids
    assert np.array_equal(g_logits, o_logits), "%d of %d logits differ" % (int((g_logits != o_logits).sum()), g_logits.size)
    ok, ov = om.kv()                                    # [layer][pos][8 * 128]
    for r in range(8):
        assert np.array_equal(gk[r][:, :n], ok[:, :n, r * 128:(r + 1) * 128]) and np.array_equal(gv[r][:, :n], ov[:, :n, r * 128:(r + 1) * 128]), "rank %d" % r
    om.close()
    xt.close()
    nt.close()


def test_tp_over_the_xcds_refuses_what_it_does_not_serve(ctx):
    """koifish::XcdTP / kf_xengine_create_tp serves the TP = 8 ranks of the Qwen3-32B shape; other rank shapes and other rank counts are refused with the reason"""
    from koifish_amd.runtime import XcdTP
    cfg = dict(synth.CONFIGS["small"])
    raw = synth.raw_weights_numpy(cfg, 31, w_std=0.1)
    w, norms = full_weights_on_gpu(ctx, cfg, raw)
    for world, what in ((8, "not instantiated"), (2, "8 ranks")):
        nt = TP.NativeTP(cfg, w, norms, world, ctx)
        for rk in nt.ranks:
            rk.set_canonical(True)
        with pytest.raises(L.KFError) as e:
            XcdTP(nt)
        assert what in str(e.value), str(e.value)
        nt.close()


@pytest.mark.parametrize("S", [1024, pytest.param(4096, marks=pytest.mark.slow)])
def test_tp_over_the_xcds_at_long_context_equals_the_per_launch_rank_step(ctx, S):
    """(default run: the first 1024 positions; --kf-slow: all 4096)
    the one-launch TP engine over all 4096 positions of a 2-layer Qwen3-32B-shaped slice, teacher-forced (16 key slices x two head groups per rank: up to 256 keys per
    workgroup), against the per-launch rank step on the same ids: every greedy id, the last logits and the ranks' K / V rows bit for bit (the per-launch step itself is held
    to the oracle at positions 128 / 1024 / 4095 by test_native_tp8_32b_slice_at_depth_vs_the_oracle_bit_for_bit)"""
    import ctypes as C
    from koifish_amd.runtime import XcdTP
    cfg = dict(synth.CONFIGS["qwen3-32b"], n_layer=2, vocab=8192, max_seq=S, tied=True)
    g = torch.Generator(device=ctx.device)
    g.manual_seed(9)

    def mat(r, c):
        return (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * 0.05).to(torch.bfloat16)

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)
    w, norms = {}, {}
    w[(-1, 0)] = ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    w[(-1, 1)] = w[(-1, 0)]
    norms[(-1, 0)] = nrm(cfg["dim"])
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = ctx.quantize(mat(*synth.SHAPES[s](cfg)), L.Q4)
        norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(128), nrm(128)
    nt = TP.NativeTP(cfg, w, norms, 8, ctx)
    for rk in nt.ranks:
        rk.set_canonical(True)
    forced = np.random.default_rng(21).integers(0, cfg["vocab"], size=S).astype(np.int32)
    xt = XcdTP(nt)
    xt.set_forced(forced)
    xt.set_state(int(forced[0]), 0)
    xt.set_steps_per_launch(32)
    xt.run_steps(S)
    ctx.sync()
    xt.check()
    ids, g_logits = xt.tokens_out(S), xt.logits()
    gk, gv = xt.kv_to_host()
    nt.set_forced(forced)
    nt.set_state(int(forced[0]), 0)
    nt.run_steps(0, S, use_graph=True)
    nt.check()
    assert np.array_equal(nt.ranks[0].tokens_out(S), ids)
    assert np.array_equal(nt.logits(), g_logits)
    n = cfg["n_layer"] * S * 128
    for r, a in enumerate(nt.ranks):
        hk, hv = np.zeros(n, dtype=np.uint16), np.zeros(n, dtype=np.uint16)
        L.check(ctx.hip.kf_d2h(ctx.h, hk.ctypes.data_as(C.c_void_p), C.c_void_p(a.host.kfh_kcache(a.h)), C.c_size_t(n * 2)), "d2h")
        L.check(ctx.hip.kf_d2h(ctx.h, hv.ctypes.data_as(C.c_void_p), C.c_void_p(a.host.kfh_vcache(a.h)), C.c_size_t(n * 2)), "d2h")
        assert np.array_equal(gk[r].reshape(-1), hk) and np.array_equal(gv[r].reshape(-1), hv), "rank %d" % r
    xt.close()
    nt.close()
