"""Synthetic Qwen3-shaped models (there is no network for checkpoints): shapes of SURVEY.md section 8 and the
weight recipe of section 8d -- weights ~ N(0, 0.02) rounded to bf16, norm weights 1 + N(0, 0.01).

numpy (PCG64) generation is used where a CPU checker must see the very same weights (tests, smoke); the
full-size benchmark model is drawn on the GPU with torch's generator.  Quantisation is always done by the
product's device quantiser (kf_quantize).
"""
import numpy as np
import torch

from . import lib as L
from .runtime import Qwen3, Context

CONFIGS = {
    # Qwen3-0.6B: cases/qwen3/qwen3_0.6B.json "transformer" block; tied embeddings
    "qwen3-0.6b": dict(dim=1024, n_layer=28, n_head=16, n_kv=8, head_dim=128, ffn=3072, vocab=151936, max_seq=2048, theta=1e6, tied=True),
    # Qwen3-1.7B: the 0.6B head geometry (16 / 8 heads of 128) on a 2048-wide stream, ffn 6144; tied embeddings
    "qwen3-1.7b": dict(dim=2048, n_layer=28, n_head=16, n_kv=8, head_dim=128, ffn=6144, vocab=151936, max_seq=2048, theta=1e6, tied=True),
    # Qwen3-4B / Qwen3-8B (cases/tutorial/history.md:4-6): 32 query heads on 8 kv-heads (GQA-4)
    "qwen3-4b": dict(dim=2560, n_layer=36, n_head=32, n_kv=8, head_dim=128, ffn=9728, vocab=151936, max_seq=2048, theta=1e6, tied=True),
    "qwen3-8b": dict(dim=4096, n_layer=36, n_head=32, n_kv=8, head_dim=128, ffn=12288, vocab=151936, max_seq=2048, theta=1e6, tied=False),
    # Qwen3-32B
    "qwen3-32b": dict(dim=5120, n_layer=64, n_head=64, n_kv=8, head_dim=128, ffn=25600, vocab=151936, max_seq=4096, theta=1e6, tied=False),
    # small shapes for parity tests (same structure: GQA group 2, head_dim 128 / 64)
    "tiny": dict(dim=256, n_layer=2, n_head=4, n_kv=2, head_dim=64, ffn=512, vocab=512, max_seq=96, theta=1e6, tied=True),
    "small": dict(dim=1024, n_layer=3, n_head=16, n_kv=8, head_dim=128, ffn=3072, vocab=4096, max_seq=160, theta=1e6, tied=True),
}

SHAPES = {  # slot -> (rows, cols) as functions of cfg
    "q": lambda c: (c["n_head"] * c["head_dim"], c["dim"]),
    "k": lambda c: (c["n_kv"] * c["head_dim"], c["dim"]),
    "v": lambda c: (c["n_kv"] * c["head_dim"], c["dim"]),
    "o": lambda c: (c["dim"], c["n_head"] * c["head_dim"]),
    "gate": lambda c: (c["ffn"], c["dim"]),
    "up": lambda c: (c["ffn"], c["dim"]),
    "down": lambda c: (c["dim"], c["ffn"]),
}
SLOTS = ("q", "k", "v", "o", "gate", "up", "down")
NORMS = ("norm_in", "norm_post", "qn", "kn")


def f32_to_bf16_np(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def raw_weights_numpy(cfg, seed=1234, w_std=0.02):
    """bf16 bit patterns (uint16) of every tensor, numpy PCG64."""
    rng = np.random.default_rng(seed)

    def mat(r, c):
        return f32_to_bf16_np(rng.normal(0.0, w_std, size=(r, c)).astype(np.float32))

    def nrm(n):
        return f32_to_bf16_np((1.0 + rng.normal(0.0, 0.01, size=n)).astype(np.float32))

    out = {"embed": mat(cfg["vocab"], cfg["dim"]), "final_norm": nrm(cfg["dim"]), "layers": []}
    if not cfg.get("tied", True):
        out["head"] = mat(cfg["vocab"], cfg["dim"])
    for _ in range(cfg["n_layer"]):
        lw = {s: mat(*SHAPES[s](cfg)) for s in SLOTS}
        lw["norm_in"], lw["norm_post"] = nrm(cfg["dim"]), nrm(cfg["dim"])
        lw["qn"], lw["kn"] = nrm(cfg["head_dim"]), nrm(cfg["head_dim"])
        out["layers"].append(lw)
    return out


def _bf16_t(a_u16, device):
    return torch.from_numpy(np.ascontiguousarray(a_u16).view(np.int16)).to(device).view(torch.bfloat16)


def build_from_raw(cfg, raw, layer_type=L.Q4, head_type=L.BF16, device=0, lGroup=128):
    """Uploads raw bf16 weights, quantises them on the GPU with kf_quantize and wires the host-side Fish."""
    ctx = Context(device)
    m = Qwen3(cfg, device)
    m._ctx = ctx
    emb = ctx.quantize(_bf16_t(raw["embed"], ctx.device), head_type, lGroup)
    m.set_weight(-1, 0, emb)
    if cfg.get("tied", True):
        m.tie_head()
    else:
        m.set_weight(-1, 1, ctx.quantize(_bf16_t(raw["head"], ctx.device), head_type, lGroup))
    m.set_norm(-1, 0, _bf16_t(raw["final_norm"], ctx.device))
    for li, lw in enumerate(raw["layers"]):
        for si, s in enumerate(SLOTS):
            m.set_weight(li, si, ctx.quantize(_bf16_t(lw[s], ctx.device), layer_type, lGroup))
        for si, s in enumerate(NORMS):
            m.set_norm(li, si, _bf16_t(lw[s], ctx.device))
    ctx.sync()
    return m


def build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16, device=0, lGroup=128, w_std=0.02, own_stream=False, head_std=None):
    """Full-size synthetic model drawn on the GPU (torch generator), quantised by kf_quantize.  head_std: std of the embedding / LM-head rows
    (default w_std); 0.1 gives peaked logits whose top-2 margin is far above one bf16 ulp (id-parity fixtures)."""
    ctx = Context(device)
    m = Qwen3(cfg, device, own_stream=own_stream)
    m._ctx = ctx
    g = torch.Generator(device=ctx.device)
    g.manual_seed(seed)

    def mat(r, c, std=None):
        return (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * (w_std if std is None else std)).to(torch.bfloat16)

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)

    m.set_weight(-1, 0, ctx.quantize(mat(cfg["vocab"], cfg["dim"], head_std), head_type, lGroup))
    if cfg.get("tied", True):
        m.tie_head()
    else:
        m.set_weight(-1, 1, ctx.quantize(mat(cfg["vocab"], cfg["dim"], head_std), head_type, lGroup))
    m.set_norm(-1, 0, nrm(cfg["dim"]))
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(SLOTS):
            m.set_weight(li, si, ctx.quantize(mat(*SHAPES[s](cfg)), layer_type, lGroup))
        m.set_norm(li, 0, nrm(cfg["dim"]))
        m.set_norm(li, 1, nrm(cfg["dim"]))
        m.set_norm(li, 2, nrm(cfg["head_dim"]))
        m.set_norm(li, 3, nrm(cfg["head_dim"]))
    ctx.sync()
    return m
