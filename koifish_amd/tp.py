"""Tensor-parallel decode for models that do not fit (or should not sit on) one GPU: Qwen3-32B, TP = 8 over xGMI.

The reference is single-GPU (`cudaSetDevice(0)`, src/Device/CUDA/QKV.cu:503); this is the one place where the path shards
naturally: Megatron-style
    q / k / v / gate / up : split by OUTPUT rows  (whole heads; ffn rows)      -> no exchange
    o_proj / down_proj    : split by INPUT columns (multiples of the 128 group) -> fp32 partial row dots per rank,
                            all-gather, summed in rank order 0..R-1, one bf16 store, + residual (kf_tp_reduce)
    embed / LM head       : embed replicated (one row per token), head split by vocab rows, (max, index) pairs gathered,
                            first maximum wins
Attention is local to the kv-heads a rank owns.  The summation order is fixed so that every rank and the CPU oracle
(kfo_qwen3_set_tp) produce the same bits.  One process per GPU; the collective is torch.distributed (backend "nccl" = RCCL
over xGMI): two all-gathers of a dim-vector of fp32 per layer are latency-bound messages (20 KB per rank at dim 5120).

`NativeTP` / `NativeRank` run the step in the C++ host (Fish::TPPhase, kf_host.cpp) with the exchange done by kernels: kf_linear_f32_push writes
each fp32 partial row into every rank's receive area over the peer mapping, kf_tp_reduce_recv sums the slots in rank order, the (max, index)
pairs of the vocabulary shards travel the same way -- no host round trip, no collective call, one hipGraph per position bucket.
`DistributedTP` (two torch.distributed all-gathers per layer, stepped from Python) stays as the RCCL baseline beside it.

`shard_rows` / `shard_cols` re-blob a `data || gama` weight (GTensor.cpp:456-510 layout) for one rank with torch indexing only,
so they work on CPU tensors (planner tests, gloo) and on GPU tensors alike.  `VirtualTP` runs all R ranks of a plan on ONE GPU
(lock-step, gathers done in process): it executes exactly the per-rank kernels and the rank-ordered combine, which is how the TP
arithmetic is tested on a 1-GPU box.
"""
import ctypes as C

import torch

from . import lib as L
from .runtime import Context, DevWeight, _ptr, SLOTS, NORMS

ROW_SPLIT = ("q", "k", "v", "gate", "up")
COL_SPLIT = ("o", "down")


def _meta(w):
    return dict(type=w.type, ne0=w.ne0, ne1=w.ne1, lGroup=w.lGroup)


def shard_rows(w, r0, r1):
    """rows [r0, r1) of a DevWeight-like (blob torch.uint8, type, ne0, ne1, lGroup) -> DevWeight of shape [r1-r0, ne1]"""
    bits = L.BITS[w.type]
    K = w.ne1
    rb = K * bits // 8
    data = w.blob[: w.szData].view(w.ne0, rb)[r0:r1].reshape(-1)
    if bits >= 8:
        return DevWeight(w.type, r1 - r0, K, data.contiguous(), w.lGroup)
    gpr = K // w.lGroup  # groups per row (rows start on group boundaries: K % lGroup == 0)
    g = w.blob[w.szData:].view(torch.int16)
    z0 = w.ne0 + w.ne1
    zero = g[z0: z0 + w.nGroup].view(w.ne0, gpr)[r0:r1].reshape(-1)
    step = g[z0 + w.nGroup: z0 + 2 * w.nGroup].view(w.ne0, gpr)[r0:r1].reshape(-1)
    ones = torch.full(((r1 - r0) + K,), 0x3F80, dtype=torch.int16, device=w.blob.device)
    blob = torch.cat([data, torch.cat([ones, zero, step]).view(torch.uint8)]).contiguous()
    return DevWeight(w.type, r1 - r0, K, blob, w.lGroup)


def shard_cols(w, c0, c1):
    """columns [c0, c1) (multiples of lGroup) -> DevWeight of shape [ne0, c1-c0]"""
    bits = L.BITS[w.type]
    assert c0 % 128 == 0 and c1 % 128 == 0
    rb = w.ne1 * bits // 8
    data = w.blob[: w.szData].view(w.ne0, rb)[:, c0 * bits // 8: c1 * bits // 8].reshape(-1)
    if bits >= 8:
        return DevWeight(w.type, w.ne0, c1 - c0, data.contiguous(), w.lGroup)
    assert c0 % w.lGroup == 0 and c1 % w.lGroup == 0
    gpr = w.ne1 // w.lGroup
    g = w.blob[w.szData:].view(torch.int16)
    z0 = w.ne0 + w.ne1
    zero = g[z0: z0 + w.nGroup].view(w.ne0, gpr)[:, c0 // w.lGroup: c1 // w.lGroup].reshape(-1)
    step = g[z0 + w.nGroup: z0 + 2 * w.nGroup].view(w.ne0, gpr)[:, c0 // w.lGroup: c1 // w.lGroup].reshape(-1)
    ones = torch.full((w.ne0 + (c1 - c0),), 0x3F80, dtype=torch.int16, device=w.blob.device)
    blob = torch.cat([data.contiguous(), torch.cat([ones, zero, step]).view(torch.uint8)]).contiguous()
    return DevWeight(w.type, w.ne0, c1 - c0, blob, w.lGroup)


class TPPlan:
    """Which slice of every tensor rank r of R owns."""

    def __init__(self, cfg, world):
        self.cfg, self.R = dict(cfg), world
        c = cfg
        if c["n_kv"] % world or c["n_head"] % world or c["ffn"] % world or c["vocab"] % world:
            raise ValueError("TP=%d does not divide heads/ffn/vocab of this config" % world)
        if (c["ffn"] // world) % 128 or (c["n_head"] // world * c["head_dim"]) % 128:
            raise ValueError("column shards must be multiples of the 128-element quantisation group")
        self.hd = c["head_dim"]
        self.n_head_l, self.n_kv_l = c["n_head"] // world, c["n_kv"] // world
        self.qd_l, self.kvd_l, self.ffn_l, self.vocab_l = self.n_head_l * self.hd, self.n_kv_l * self.hd, c["ffn"] // world, c["vocab"] // world

    def rows(self, slot, r):
        n = {"q": self.qd_l, "k": self.kvd_l, "v": self.kvd_l, "gate": self.ffn_l, "up": self.ffn_l}[slot]
        return r * n, (r + 1) * n

    def cols(self, slot, r):
        n = {"o": self.qd_l, "down": self.ffn_l}[slot]
        return r * n, (r + 1) * n

    def shard(self, slot, w, r):
        return shard_rows(w, *self.rows(slot, r)) if slot in ROW_SPLIT else shard_cols(w, *self.cols(slot, r))

    def head_rows(self, r):
        return r * self.vocab_l, (r + 1) * self.vocab_l


class TPRank:
    """One rank's share of the decode step on one GPU: shard weights, local KV cache, the per-phase kernels."""

    def __init__(self, plan, rank, ctx, weights, norms):
        """weights: {(layer, slot_index) | (-1, 0 embed) | (-1, 1 head): DevWeight SHARD (embed is the full table)}, norms as Qwen3._norms"""
        self.p, self.rank, self.ctx, self.w, self.norms = plan, rank, ctx, weights, norms
        self.device = ctx.device
        c, dev = plan.cfg, ctx.device
        self.kc = torch.zeros(c["n_layer"], c["max_seq"], plan.kvd_l, dtype=torch.bfloat16, device=dev)
        self.vc = torch.zeros_like(self.kc)
        self.table = ctx.rope_table(c["max_seq"], plan.hd, c.get("theta", 1e6))
        self.x = torch.zeros(c["dim"], dtype=torch.bfloat16, device=dev)
        self.q = torch.zeros(plan.qd_l, dtype=torch.bfloat16, device=dev)
        self.kraw = torch.zeros(plan.kvd_l, dtype=torch.bfloat16, device=dev)
        self.att = torch.zeros(plan.qd_l, dtype=torch.bfloat16, device=dev)
        self.act = torch.zeros(plan.ffn_l, dtype=torch.bfloat16, device=dev)
        self.partial = torch.zeros(c["dim"], dtype=torch.float32, device=dev)
        self.logits = torch.zeros(plan.vocab_l, dtype=torch.bfloat16, device=dev)
        self.eps, self.qk_eps = c.get("rms_eps", 1e-6), c.get("qk_eps", 1e-6)
        self.ws = torch.zeros(ctx.hip.kf_attn_scratch_bytes(plan.n_head_l, plan.hd), dtype=torch.uint8, device=dev)

    def _chk(self, rc, what):
        L.check(rc, what)

    def embed(self, token):
        self.x = self.ctx.embed(self.w[(-1, 0)], token)

    def attn_partial(self, layer, pos):
        """[norm + q,k,v rows of this rank] -> [q/k-norm + RoPE + attention over the local kv-heads] -> fp32 partial of o_proj"""
        ctx, p, hip = self.ctx, self.p, self.ctx.hip
        ws = [self.w[(layer, i)] for i in (0, 1, 2)]
        descs = [w.desc() for w in ws]
        wp = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
        ys = (C.c_void_p * 3)(self.q.data_ptr(), self.kraw.data_ptr(), self.vc[layer].data_ptr())
        strides = (C.c_int64 * 3)(0, 0, p.kvd_l)
        self._chk(hip.kf_norm_linear(ctx.h, _ptr(self.x), _ptr(self.norms[(layer, 0)]), self.eps, 3, wp, ys, strides, pos, None), "kf_norm_linear")
        self._chk(hip.kf_attn_block(ctx.h, _ptr(self.q), _ptr(self.kraw), _ptr(self.kc[layer]), _ptr(self.vc[layer]), _ptr(self.att),
                                    _ptr(self.norms[(layer, 2)]), _ptr(self.norms[(layer, 3)]), _ptr(self.table), pos, None, p.n_head_l, p.n_kv_l, p.hd,
                                    p.kvd_l, self.qk_eps, _ptr(self.ws)), "kf_attn_block")
        d = self.w[(layer, 3)].desc()
        self._chk(hip.kf_linear_f32(ctx.h, C.byref(d), _ptr(self.att), _ptr(self.partial)), "kf_linear_f32")
        return self.partial

    def ffn_partial(self, layer):
        ctx, hip = self.ctx, self.ctx.hip
        dg, du, dd = self.w[(layer, 4)].desc(), self.w[(layer, 5)].desc(), self.w[(layer, 6)].desc()
        self._chk(hip.kf_norm_gateup_swiglu(ctx.h, _ptr(self.x), _ptr(self.norms[(layer, 1)]), self.eps, C.byref(dg), C.byref(du), _ptr(self.act)), "gateup")
        self._chk(hip.kf_linear_f32(ctx.h, C.byref(dd), _ptr(self.act), _ptr(self.partial)), "kf_linear_f32")
        return self.partial

    def combine(self, gathered):
        """gathered: fp32 [R, dim] in rank order -> x = bf16(x + bf16(sum_r))"""
        self._chk(self.ctx.hip.kf_tp_reduce(self.ctx.h, _ptr(gathered), gathered.shape[0], gathered.shape[1], _ptr(self.x), _ptr(self.x)), "kf_tp_reduce")

    def head_local(self):
        """(max logit, GLOBAL index of its first occurrence) over this rank's vocab rows"""
        ctx = self.ctx
        xn = ctx.rmsnorm(self.x, self.norms[(-1, 0)], self.eps)
        w = self.w[(-1, 1)]
        am = torch.zeros(1, dtype=torch.int32, device=ctx.device)
        d = w.desc()
        self._chk(ctx.hip.kf_lm_head(ctx.h, C.byref(d), _ptr(xn), _ptr(self.logits), _ptr(am), _ptr(ctx._head_ws)), "kf_lm_head")
        i = int(am.item())
        return float(self.logits[i].float().item()), i + self.p.head_rows(self.rank)[0]


def pick_first_max(pairs):
    """pairs: [(value, global index)] in rank order; first maximum (sample_argmax, GoPT.cpp:602-612: lowest index among equals)"""
    best = None
    for v, i in pairs:
        if best is None or v > best[0] or (v == best[0] and i < best[1]):
            best = (v, i)
    return best[1]


def build_ranks_from_full(cfg, full_weights, norms, world, ctx, ranks=None):
    """full_weights: {(layer, slot_index): DevWeight} of the unsharded model (as koifish_amd.runtime.Qwen3.weights)."""
    plan = TPPlan(cfg, world)
    out = []
    for r in (range(world) if ranks is None else ranks):
        w = {(-1, 0): full_weights[(-1, 0)], (-1, 1): shard_rows(full_weights[(-1, 1)], *plan.head_rows(r))}
        for li in range(cfg["n_layer"]):
            for si, s in enumerate(SLOTS):
                w[(li, si)] = plan.shard(s, full_weights[(li, si)], r)
        out.append(TPRank(plan, r, ctx, w, norms))
    return plan, out


class VirtualTP:
    """All R ranks of a plan on one GPU, lock-step; the all-gather is a torch.stack.  Same kernels, same combine order."""

    def __init__(self, cfg, full_weights, norms, world, ctx):
        self.cfg, self.ctx = cfg, ctx
        self.plan, self.ranks = build_ranks_from_full(cfg, full_weights, norms, world, ctx)

    def step(self, token, pos):
        for r in self.ranks:
            r.embed(token)
        for layer in range(self.cfg["n_layer"]):
            g = torch.stack([r.attn_partial(layer, pos).clone() for r in self.ranks])
            for r in self.ranks:
                r.combine(g)
            g = torch.stack([r.ffn_partial(layer).clone() for r in self.ranks])
            for r in self.ranks:
                r.combine(g)
        return pick_first_max([r.head_local() for r in self.ranks])

    def logits(self):
        return torch.cat([r.logits for r in self.ranks])

    def generate(self, prompt, n_new):
        pos, nxt = 0, None
        for t in prompt:
            nxt = self.step(int(t), pos)
            pos += 1
        out = []
        for _ in range(n_new):
            out.append(nxt)
            nxt = self.step(nxt, pos)
            pos += 1
        return out


class DistributedTP:
    """One rank per process / GPU; collectives through torch.distributed (RCCL).  `rank_obj` is this process's TPRank."""

    def __init__(self, rank_obj, group=None):
        import torch.distributed as dist
        self.r, self.dist, self.group = rank_obj, dist, group
        self.world = dist.get_world_size(group)
        dim = rank_obj.p.cfg["dim"]
        self.dim = dim
        self.gathered = torch.zeros(self.world * dim, dtype=torch.float32, device=rank_obj.device)   # flat: what gloo and nccl both accept
        self.pair = torch.zeros(2, dtype=torch.float64, device=rank_obj.device)
        self.pairs = torch.zeros(self.world * 2, dtype=torch.float64, device=rank_obj.device)

    def _gather(self, partial):
        self.dist.all_gather_into_tensor(self.gathered, partial.contiguous().view(-1), group=self.group)
        return self.gathered.view(self.world, self.dim)

    def step(self, token, pos):
        r = self.r
        r.embed(token)
        for layer in range(r.p.cfg["n_layer"]):
            r.combine(self._gather(r.attn_partial(layer, pos)))
            r.combine(self._gather(r.ffn_partial(layer)))
        v, i = r.head_local()
        self.pair[0], self.pair[1] = v, float(i)
        self.dist.all_gather_into_tensor(self.pairs, self.pair, group=self.group)
        pp = self.pairs.view(self.world, 2).cpu().tolist()
        return pick_first_max([(a, int(b)) for a, b in pp])


def local_cfg(cfg, plan):
    """the card of one rank's Fish: local head / ffn / vocab counts, full embedding width"""
    return dict(cfg, n_head=plan.n_head_l, n_kv=plan.n_kv_l, ffn=plan.ffn_l, vocab=plan.vocab_l)


def build_native_rank(cfg, plan, rank, shards, norms, device=0):
    """shards as TPRank takes them -> a Qwen3 (C++ Fish) holding this rank's slice, TP state initialised (peers still to be set)"""
    from .runtime import Qwen3
    m = Qwen3(local_cfg(cfg, plan), device)
    m.set_weight(-1, 0, shards[(-1, 0)])          # replicated embedding table (rows of the FULL vocabulary)
    m.set_weight(-1, 1, shards[(-1, 1)])          # vocabulary shard of the head
    m.set_norm(-1, 0, norms[(-1, 0)])
    for li in range(cfg["n_layer"]):
        for si in range(len(SLOTS)):
            m.set_weight(li, si, shards[(li, si)])
        for si in range(len(NORMS)):
            m.set_norm(li, si, norms[(li, si)])
    L.check(m.host.kfh_tp_init(m.h, rank, plan.R, plan.head_rows(rank)[0]), "kfh_tp_init")
    return m


class NativeTP:
    """All R ranks of a plan in ONE process on one GPU, stepped by the C++ host in lock-step on one stream (phase by phase across the ranks, so a
    reduce never queues ahead of the pushes it waits for).  Same kernels, same exchange kernels, same graphs as one-rank-per-GPU; the peers' receive
    areas are simply local pointers."""

    def __init__(self, cfg, full_weights, norms, world, ctx):
        self.cfg, self.ctx, self.world = cfg, ctx, world
        self.plan = TPPlan(cfg, world)
        self.ranks = []
        for r in range(world):
            w = {(-1, 0): full_weights[(-1, 0)], (-1, 1): shard_rows(full_weights[(-1, 1)], *self.plan.head_rows(r))}
            for li in range(cfg["n_layer"]):
                for si, s in enumerate(SLOTS):
                    w[(li, si)] = self.plan.shard(s, full_weights[(li, si)], r)
            self.ranks.append(build_native_rank(cfg, self.plan, r, w, norms, ctx.device.index or 0))
        host = self.ranks[0].host
        for a in self.ranks:
            for r, b in enumerate(self.ranks):
                L.check(host.kfh_tp_set_peer(a.h, r, C.c_void_p(host.kfh_tp_area(b.h))), "kfh_tp_set_peer")
        self._hs = (C.c_void_p * world)(*[m.h for m in self.ranks])
        self.host = host

    def set_forced(self, forced):
        for m in self.ranks:
            m.set_forced(forced)

    def set_state(self, token, pos):
        for m in self.ranks:
            m.set_state(token, pos)

    def run_steps(self, pos, n, use_graph=True):
        L.check(self.host.kfh_tp_group_run(self._hs, self.world, int(pos), int(n), int(use_graph)), "kfh_tp_group_run")

    def check(self):
        for m in self.ranks:
            L.check(self.host.kfh_tp_check(m.h), "kfh_tp_check (a poll timed out)")

    def step(self, token, pos, use_graph=False):
        """one token through all ranks; returns the greedy id every rank agreed on"""
        self.set_state(token, pos)
        self.run_steps(pos, 1, use_graph)
        self.check()
        ids = {int(m.tokens_out(pos + 1)[pos]) for m in self.ranks}
        assert len(ids) == 1, ids
        return ids.pop()

    def logits(self):
        """the vocabulary shards of the last step's logits, concatenated in rank order (bf16 bit patterns)"""
        import numpy as np
        return np.concatenate([m.logits() for m in self.ranks])

    def generate(self, prompt, n_new, use_graph=True):
        import numpy as np
        n_prompt = len(prompt)
        forced = np.full(self.cfg["max_seq"], -1, dtype=np.int32)
        forced[:n_prompt] = prompt
        self.set_forced(forced)
        self.set_state(int(prompt[0]), 0)
        self.run_steps(0, n_prompt + n_new - 1, use_graph)
        self.check()
        toks = self.ranks[0].tokens_out(n_prompt + n_new - 1)
        return [int(t) for t in toks[n_prompt - 1:]]

    def close(self):
        for m in self.ranks:
            m.close()


class NativeRank:
    """One rank per process / GPU: the receive areas are exchanged as IPC handles through torch.distributed once, after that the step is the C++
    host's graph (Fish::EnqueueStepTP) with no collective call."""

    def __init__(self, cfg, plan, rank, shards, norms, device, group=None):
        import torch.distributed as dist
        self.m = build_native_rank(cfg, plan, rank, shards, norms, device)
        self.cfg, self.plan, self.rank, self.world = cfg, plan, rank, plan.R
        host = self.m.host
        h = (C.c_ubyte * 64)()
        L.check(host.kfh_tp_export(self.m.h, h), "kfh_tp_export")
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(h), group=group)
        for r, hb in enumerate(handles):
            if r == rank:
                continue
            buf = (C.c_ubyte * 64).from_buffer_copy(hb)
            L.check(host.kfh_tp_open_peer(self.m.h, r, buf), "kfh_tp_open_peer(%d)" % r)
        dist.barrier(group=group)

    def set_forced(self, forced):
        self.m.set_forced(forced)

    def set_state(self, token, pos):
        self.m.set_state(token, pos)

    def run_steps(self, pos, n, use_graph=True):
        self.m.run_steps(pos, n, use_graph)

    def check(self):
        L.check(self.m.host.kfh_tp_check(self.m.h), "kfh_tp_check (a poll timed out)")
