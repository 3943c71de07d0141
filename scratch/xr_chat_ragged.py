"""The request queue under ragged load (XcdReplicas.chat on Qwen3-0.6B 4-bit): prompts of 16 .. 512 tokens, a limit of 16 .. 256 new ids of its own per request (the
slots free up at scattered steps and are refilled while the others decode on), 32 slots, waiting prompts one by one / prefilled together.  Prints requests/s, generated tokens/s, the launches.
  python scratch/xr_chat_ragged.py [n_req=128]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd.runtime import XcdReplicas

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
m.set_canonical(True)
rng = np.random.default_rng(3)
prompts = [rng.integers(0, cfg["vocab"], size=int(rng.integers(16, 513))).astype(np.int32) for _ in range(n_req)]
xr = XcdReplicas(m, 32)
xr.set_steps_per_launch(16)
each = rng.integers(16, 257, size=n_req).astype(np.int32)
for pb in (1, 8):
    xr.set_prefill_batch(pb)
    xr.chat(prompts[:32], 32)
    m.sync()
    t0 = time.perf_counter()
    got, st = xr.chat(prompts, 256, max_new_each=each)
    dt = time.perf_counter() - t0
    lens = np.array([len(g) for g in got])
    print("prefill batch %d: %d requests (prompts 16..512, mean %.0f; answers of %d..%d ids, mean %.0f) in %.3f s: %.1f requests/s, %.1f generated tokens/s, %.1f prompt+generated tokens/s  %s" % (
        pb, n_req, np.mean([len(p) for p in prompts]), lens.min(), lens.max(), lens.mean(), dt, n_req / dt, lens.sum() / dt, (lens.sum() + sum(len(p) for p in prompts)) / dt, st), flush=True)
xr.close()
m.close()
