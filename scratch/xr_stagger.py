import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from koifish_amd import lib as L, synth
from koifish_amd.runtime import XcdReplicas
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16); m.set_canonical(True)
xr = XcdReplicas(m, 16)
pos0, steps = 2028, 20
for var in (8, 86, 81, 82, 88):
    xr.variant(var, 0)
    stag = var
    best = 1e9
    for rep in range(3):
        for s in range(16): xr.set_state(s, 1 + s, pos0)
        m.sync(); t0 = time.perf_counter(); xr.run_steps(steps); m.sync(); best = min(best, time.perf_counter() - t0)
    xr.check()
    print("variant %4d: %.3f ms per step of 16 sequences, %.1f tokens/s" % (stag, best * 1e3 / steps, 16 * steps / best), flush=True)
