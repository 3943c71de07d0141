"""1.7B shape, teacher-forced: logits of the engine against the per-layer launches, per summation order and number of layers (where a difference enters)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import synth, lib as L
from oracle import oracle as O
for nl in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2", "28"])]:
    cfg = dict(synth.CONFIGS["qwen3-1.7b"]); cfg["n_layer"] = nl
    m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
    toks = np.random.default_rng(7).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
    m.set_forced(toks); m.set_engine_autotune(0)
    for canon in (1, 0):
        m.set_canonical(canon)
        out = {}
        for eng in (True, False):
            m.set_engine(eng)
            lg = []
            for p in (0, 1, 2, 70, 300):
                if p >= 70: 
                    pass
                m.set_state(int(toks[p]), p)
                m.run_steps(p, 1, True); m.sync()
                lg.append(m.logits().copy())
            out[eng] = lg
        d = [(int((a != b).sum()), float(np.abs(O.bf16_to_f32(a) - O.bf16_to_f32(b)).max())) for a, b in zip(out[True], out[False])]
        print("layers=%d canonical=%d: (differing logits, max abs diff) per position:" % (nl, canon), d, " engine steps", m.engine_steps(), flush=True)
    m.close()
