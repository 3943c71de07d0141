"""Shared test helpers: the same synthetic weights handed to the CPU oracle and to the HIP path."""
import numpy as np

from koifish_amd import lib as L
from koifish_amd import synth
from oracle import oracle as O


def oracle_model(cfg, raw, layer_type=L.Q4, head_type=L.BF16, attn_mode=O.ATTN_FUSED, tp=1):
    """Quantises `raw` with the oracle's own quantiser (GeQuant::RTN_x restatement) and builds the CPU decoder."""
    def q(a, t):
        if t == L.NF4:
            return O.quantize_nf4(a, a.shape[0], a.shape[1])
        return O.quantize(a, a.shape[0], a.shape[1], t)
    w = {"embed": q(raw["embed"], head_type), "final_norm": raw["final_norm"], "layers": []}
    w["head"] = w["embed"] if cfg.get("tied", True) else q(raw["head"], head_type)
    for lw in raw["layers"]:
        d = {s: q(lw[s], layer_type) for s in synth.SLOTS}
        for s in synth.NORMS:
            d[s] = lw[s]
        w["layers"].append(d)
    return O.Qwen3Oracle(cfg, w, attn_mode=attn_mode, tp=tp)


def prompt_ids(cfg, n, seed=7):
    return np.random.default_rng(seed).integers(0, cfg["vocab"], size=n).astype(np.int32)
