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


def ids_agree_up_to_a_near_tie(om, prompt, got, ref, ulps=2):
    """Free-running greedy ids in the DEFAULT order (fp32 sums, within the logit tolerance of the oracle): equal to the oracle's `ref`, or first different at a step
    where the oracle's own two best logits lie within `ulps` bf16 ulps of each other -- there the fp32 summation order legitimately decides the id (the canonical order,
    kf_set_canonical, is the mode in which ids are equal by construction).  `om`: a fresh oracle model of the same weights; returns (ok, message)."""
    if list(got) == list(ref):
        return True, "equal"
    i = next(k for k, (a, b) in enumerate(zip(got, ref)) if a != b)
    seq = list(prompt) + list(ref)
    logits = None
    for p in range(len(prompt) + i):          # teacher-forced along the oracle's ids up to the step that produced ref[i]
        _, logits, _ = om.decode(int(seq[p]), p)
    f = O.bf16_to_f32(logits)
    top = np.sort(f)[-2:]
    margin = float(top[1] - top[0])
    tol = ulps * 2.0 ** -7 * max(abs(float(top[1])), 1e-9)   # a bf16 ulp is 2^-8 .. 2^-7 of the value
    return margin <= tol, "first difference at generated id %d (%d vs %d): the oracle's top-2 margin there is %g, %d bf16 ulps are %g" % (i, got[i], ref[i], margin, ulps, tol)
