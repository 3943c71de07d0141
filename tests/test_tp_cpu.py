"""Tensor-parallel host logic on the CPU (no GPU): the shard planner re-blobs `data||gama` weights correctly, and a world_size-2
`gloo` run of the distributed driver (koifish_amd.tp.DistributedTP) with the per-rank compute done by the CPU oracle reproduces the
oracle's own tensor-parallel emulation (kfo_qwen3_set_tp: rank-ordered fp32 partial sums) -- ids and logits bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd.runtime import DevWeight
from koifish_amd import tp as TP
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def dev_from_oracle(ow):
    """oracle QWeight -> DevWeight on a CPU torch tensor (the planner only does torch indexing)"""
    return DevWeight(ow.type, ow.ne0, ow.ne1, torch.from_numpy(ow.blob().copy()), ow.lGroup)


def oracle_from_dev(dw):
    blob = dw.blob.numpy()
    data = blob[:dw.szData]
    if dw.type == L.BF16:
        return O.QWeight(O.BF16, dw.ne0, dw.ne1, data.view(np.uint16))
    if dw.type == L.F8E5M2:
        return O.QWeight(O.F8E5M2, dw.ne0, dw.ne1, data)
    g = blob[dw.szData:].view(np.uint16)
    z0 = dw.ne0 + dw.ne1
    return O.QWeight(dw.type, dw.ne0, dw.ne1, data, g[z0:z0 + dw.nGroup].copy(), g[z0 + dw.nGroup:z0 + 2 * dw.nGroup].copy(), dw.lGroup, dw.qBias)


@pytest.mark.parametrize("t", [L.Q4, L.BF16, L.F8E5M2, L.T_SIGN, L.BOOL1])
def test_shards_dequantise_to_slices_of_the_full_weight(t):
    rng = np.random.default_rng(t)
    m, k = 64, 512
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(m, k)).astype(np.float32))
    ow = O.quantize(w, m, k, t)
    full = O.dequant(ow)
    dw = dev_from_oracle(ow)
    rs = TP.shard_rows(dw, 16, 48)
    assert (rs.ne0, rs.ne1) == (32, k)
    assert np.array_equal(O.dequant(oracle_from_dev(rs)), full[16:48])
    cs = TP.shard_cols(dw, 128, 384)
    assert (cs.ne0, cs.ne1) == (m, 256)
    assert np.array_equal(O.dequant(oracle_from_dev(cs)), full[:, 128:384])


def test_plan_rejects_bad_splits():
    cfg = dict(synth.CONFIGS["small"])
    TP.TPPlan(cfg, 8)
    with pytest.raises(ValueError):
        TP.TPPlan(cfg, 3)
    with pytest.raises(ValueError):
        TP.TPPlan(dict(cfg, ffn=3072 + 64 * 8), 8)     # shard not a multiple of the 128 group
    p = TP.TPPlan(dict(synth.CONFIGS["qwen3-32b"]), 8)
    assert (p.n_head_l, p.n_kv_l, p.ffn_l, p.vocab_l) == (8, 1, 3200, 18992)


class OracleRank:
    """koifish_amd.tp.TPRank's interface with the arithmetic done by the CPU oracle (test only)."""

    def __init__(self, plan, rank, weights, norms):
        self.p, self.rank, self.w, self.norms, self.device = plan, rank, weights, norms, torch.device("cpu")
        c = plan.cfg
        self.kc = np.zeros((c["n_layer"], c["max_seq"], plan.kvd_l), dtype=np.uint16)
        self.vc = np.zeros_like(self.kc)
        self.x = None
        self.logits = None

    def embed(self, token):
        self.x = O.embed(self.w[(-1, 0)], token)

    def attn_partial(self, layer, pos):
        p, c = self.p, self.p.cfg
        xb = O.rmsnorm(self.x, self.norms[(layer, 0)])
        q = O.linear(self.w[(layer, 0)], xb)
        k = O.linear(self.w[(layer, 1)], xb)
        self.vc[layer, pos] = O.linear(self.w[(layer, 2)], xb)
        q = O.rope(O.headnorm(q, self.norms[(layer, 2)], p.n_head_l, p.hd), p.n_head_l, p.hd, pos, c["theta"])
        self.kc[layer, pos] = O.rope(O.headnorm(k, self.norms[(layer, 3)], p.n_kv_l, p.hd), p.n_kv_l, p.hd, pos, c["theta"])
        att = O.attn_decode(q, self.kc[layer], self.vc[layer], pos, p.n_head_l, p.n_kv_l, p.hd, mode=O.ATTN_FUSED)
        return torch.from_numpy(O.linear_f32(self.w[(layer, 3)], att))

    def ffn_partial(self, layer):
        xb = O.rmsnorm(self.x, self.norms[(layer, 1)])
        act = O.swiglu(O.linear(self.w[(layer, 4)], xb), O.linear(self.w[(layer, 5)], xb))
        return torch.from_numpy(O.linear_f32(self.w[(layer, 6)], act))

    def combine(self, gathered):
        g = gathered.numpy()
        tot = g[0].copy()
        for r in range(1, g.shape[0]):
            tot = (tot + g[r]).astype(np.float32)
        self.x = O.add(self.x, O.f32_to_bf16(tot))

    def head_local(self):
        xn = O.rmsnorm(self.x, self.norms[(-1, 0)])
        self.logits = O.linear(self.w[(-1, 1)], xn)
        i = O.argmax_bf16(self.logits)
        return float(O.bf16_to_f32(self.logits[i:i + 1])[0]), i + self.p.head_rows(self.rank)[0]


def build_oracle_rank(cfg, raw, world, rank):
    plan = TP.TPPlan(cfg, world)

    def q(a, t):
        return O.quantize(a, a.shape[0], a.shape[1], t)
    emb = q(raw["embed"], L.BF16)
    w = {(-1, 0): emb, (-1, 1): oracle_from_dev(TP.shard_rows(dev_from_oracle(emb), *plan.head_rows(rank)))}
    norms = {(-1, 0): raw["final_norm"]}
    for li, lw in enumerate(raw["layers"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = oracle_from_dev(plan.shard(s, dev_from_oracle(q(lw[s], L.Q4)), rank))
        for si, s in enumerate(synth.NORMS):
            norms[(li, si)] = lw[s]
    return plan, OracleRank(plan, rank, w, norms)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, HERE)
    cfg = dict(synth.CONFIGS["tiny"])
    raw = synth.raw_weights_numpy(cfg, 77, w_std=0.1)
    plan, r = build_oracle_rank(cfg, raw, world, rank)
    drv = TP.DistributedTP(r)
    prompt = np.random.default_rng(5).integers(0, cfg["vocab"], size=6)
    ids, pos, nxt = [], 0, None
    for t in prompt:
        nxt = drv.step(int(t), pos)
        pos += 1
    for _ in range(8):
        ids.append(nxt)
        nxt = drv.step(nxt, pos)
        pos += 1
    logits = torch.from_numpy(r.logits.astype(np.int32))
    parts = [torch.zeros_like(logits) for _ in range(world)]
    dist.all_gather(parts, logits)
    if rank == 0:
        ret["ids"] = ids
        ret["logits"] = torch.cat(parts).numpy().astype(np.uint16)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_matches_oracle_tp():
    import torch.multiprocessing as mp
    from helpers import oracle_model
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    cfg = dict(synth.CONFIGS["tiny"])
    raw = synth.raw_weights_numpy(cfg, 77, w_std=0.1)
    om = oracle_model(cfg, raw, L.Q4, L.BF16, tp=2)
    prompt = np.random.default_rng(5).integers(0, cfg["vocab"], size=6)
    ref = om.generate(prompt.tolist(), 8)
    assert ret["ids"] == ref
    assert len(set(ref)) > 3
    # logits of the last step, bit for bit (same fp32 partials, same rank order)
    om2 = oracle_model(cfg, raw, L.Q4, L.BF16, tp=2)
    seq = prompt.tolist() + ref
    lg = None
    for pos, t in enumerate(seq):
        _, lg, _ = om2.decode(int(t), pos)
    assert np.array_equal(ret["logits"], lg)
