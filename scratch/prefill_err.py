import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L, synth
from oracle import oracle as O
for cfg_name, lt, n in (("tiny", L.BOOL1, 40), ("tiny", L.Q4, 40), ("tiny", L.T_SIGN, 40), ("small", L.Q4, 130)):
    cfg = synth.CONFIGS[cfg_name]
    raw = synth.raw_weights_numpy(cfg, 1234, w_std=0.02)
    gm = synth.build_from_raw(cfg, raw, lt, L.BF16)
    om = oracle_model(cfg, raw, lt, L.BF16)
    prompt = prompt_ids(cfg, n)
    g_next, g_logits = gm.prefill(prompt)
    for pos, tok in enumerate(prompt):
        o_next, o_logits, _ = om.decode(int(tok), pos)
    gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
    gk, gv = gm.kv_to_host(); ok, ov = om.kv()
    errs = []
    for g, o in ((gk, ok), (gv, ov)):
        for l in range(cfg["n_layer"]):
            a, b = O.bf16_to_f32(g[l, :n]), O.bf16_to_f32(o[l, :n])
            errs.append(np.abs(a - b).max() / np.abs(b).max())
    print(cfg_name, lt, "logit err %.5f" % (np.abs(gl - ol).max() / np.abs(ol).max()), "kv errs", ["%.4f" % e for e in errs], g_next == o_next, flush=True)
