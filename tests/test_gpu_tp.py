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
