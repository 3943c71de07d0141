"""A queue of prompts through the batched XCD decoders (XcdReplicas.chat) on Qwen3-0.6B 4-bit: requests/s and tokens/s, prefill included.
  python scratch/xr_chat.py [n_req=96] [prompt=128] [new=128]     env: NSEQ="32,16,8"  SPL=32"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd.runtime import XcdReplicas

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 96
n_prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n_new = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cfg = dict(synth.CONFIGS[os.environ.get("CONFIG", "qwen3-0.6b")])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
m.set_canonical(True)
rng = np.random.default_rng(11)
prompts = [rng.integers(0, cfg["vocab"], size=n_prompt).astype(np.int32) for _ in range(n_req)]
for n_seq in [int(x) for x in os.environ.get("NSEQ", "32").split(",")]:
    xr = XcdReplicas(m, n_seq)
    xr.set_steps_per_launch(int(os.environ.get("SPL", "32")))
    for pb in [int(x) for x in os.environ.get("PB", "1").split(",")]:
      xr.set_prefill_batch(min(pb, n_seq))
      if pb > 1:   # the batch prefill alone
        xr.prefill_batch(list(range(min(pb, n_seq))), prompts[:min(pb, n_seq)])
        m.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            xr.prefill_batch(list(range(min(pb, n_seq))), prompts[:min(pb, n_seq)])
        m.sync()
        dt = (time.perf_counter() - t0) / 5
        print("   prefill_batch of %d x %d tokens: %.3f ms = %.3f ms per prompt" % (min(pb, n_seq), n_prompt, dt * 1e3, dt * 1e3 / min(pb, n_seq)), flush=True)
      xr.chat(prompts[:n_seq], 8)
      best, st = 1e9, None
      for rep in range(2):
        m.sync()
        t0 = time.perf_counter()
        got, st = xr.chat(prompts, n_new)
        best = min(best, time.perf_counter() - t0)
      assert all(len(g) == n_new for g in got)
      print("prefill batch %d  n_seq %d: %d requests (%d-token prompt, %d new) in %.3f s: %.1f requests/s, %.1f generated tokens/s, %.1f prompt+generated tokens/s  %s" % (
        pb, n_seq, n_req, n_prompt, n_new, best, n_req / best, n_req * n_new / best, n_req * (n_prompt + n_new) / best, st), flush=True)
    xr.close()
m.close()
