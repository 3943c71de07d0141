"""Prefill timing: token batches (MFMA) vs token-serial graph replay, Qwen3-0.6B Q4, prompts of 128 / 512 / 2047 tokens."""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth

cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
rng = np.random.default_rng(5)

for n in (32, 128, 512, 1024, 2047):
    p = rng.integers(0, cfg["vocab"], size=n).astype(np.int32)
    for _ in range(3):   # eager, capture, first replay
        m.prefill(p, want_logits=False)
    m.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        m.prefill(p, want_logits=False)
    m.sync()
    tb = (time.perf_counter() - t0) / 10
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32); forced[:n] = p
    m.set_forced(forced); m.set_state(int(p[0]), 0)
    m.run_steps(0, n); m.sync()
    m.set_state(int(p[0]), 0)
    t0 = time.perf_counter()
    m.run_steps(0, n); m.sync()
    ts = time.perf_counter() - t0
    print("n=%d batched %.3f ms (%.0f tok/s)  token-serial %.3f ms (%.0f tok/s)  x%.1f" % (n, tb * 1e3, n / tb, ts * 1e3, n / ts, ts / tb), flush=True)
