"""Soak of the persistent engine: R full greedy generations (positions 128..2047, runs of steps per launch) of the 0.6B model, default and canonical order alternating;
every generation must reproduce the first one's ids of its order bit for bit and leave the engine's error word clear."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = synth.CONFIGS["qwen3-0.6b"]
m = synth.build_on_gpu(cfg, seed=1234)
prompt = np.random.default_rng(3).integers(0, cfg["vocab"], size=128).astype(np.int32)
ref = {}
t0 = time.time()
for r in range(R):
    canon = r & 1
    m.set_canonical(bool(canon))
    ids = m.generate(prompt, cfg["max_seq"] - 128 - 1, use_graph=True)
    m.engine_check()
    if canon not in ref:
        ref[canon] = ids
        print("order %d: %d ids, %d distinct" % (canon, len(ids), len(set(ids))), flush=True)
    assert ids == ref[canon], "generation %d (order %d) differs from the first at index %d" % (r, canon, next(i for i, (a, b) in enumerate(zip(ids, ref[canon])) if a != b))
print("soak ok: %d generations, %d engine steps, %.1f s" % (R, m.engine_steps(), time.time() - t0))
