"""decode ms/step at positions 1920..2040: one-node hipGraph replay per step vs eager launches of the same kernel (the step is ONE launch either way)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth
cfg = synth.CONFIGS["qwen3-0.6b"]
m = synth.build_on_gpu(cfg, seed=1234)
forced = np.random.default_rng(7).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
m.set_forced(forced)
pos0, n = 1900, 140
for use_graph in (True, False, True, False):
    m.set_state(int(forced[pos0]), pos0)
    m.run_steps(pos0, 8, use_graph); m.sync()
    m.set_state(int(forced[pos0 + 8]), pos0 + 8)
    t0 = time.perf_counter()
    m.run_steps(pos0 + 8, n, use_graph); m.sync()
    dt = (time.perf_counter() - t0) / n * 1e3
    m.engine_check()
    print("graph=%s  %.4f ms/step (%.0f tok/s)" % (use_graph, dt, 1.0 / dt * 1e3), flush=True)
