"""1.7B shape, v_dot2c order: which half of a layer lets the engine and the per-layer launches drift apart?  Models with down_proj = 0 (no FFN contribution) or o_proj = 0 (no attention contribution)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import synth, lib as L
cfg = dict(synth.CONFIGS["qwen3-1.7b"]); cfg["n_layer"] = 6; cfg["vocab"] = 4096
toks = np.random.default_rng(7).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
for zero in (None, "down", "o"):
    raw = synth.raw_weights_numpy(cfg, 1234)
    if zero:
        for lw in raw["layers"]:
            key = [k for k in lw if k.startswith(zero)][0]
            lw[key][:] = 0
    m = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    m.set_forced(toks); m.set_engine_autotune(0); m.set_canonical(0)
    out = {}
    for eng in (True, False):
        m.set_engine(eng)
        lg = []
        for p in (0, 1, 2, 3):
            m.set_state(int(toks[p]), p); m.run_steps(p, 1, True); m.sync()
            lg.append(m.logits().copy())
        out[eng] = lg
    print("zeroed:", zero, " differing logits per position:", [int((a != b).sum()) for a, b in zip(out[True], out[False])], "engine steps", m.engine_steps(), flush=True)
    m.close()
