"""Prefill timing: token batches (MFMA) vs token-serial graph replay, Qwen3-0.6B Q4, prompts of 128 / 512 / 2047 tokens."""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth

cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
rng = np.random.default_rng(5)
import _knobs
_knobs.apply(m.hip)   # KF_RESIDENT_MIN / KF_G3_FIRST ... (scratch/_knobs.py)
SIZES = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else (32, 128, 512, 1024, 2047)
SERIAL = len(sys.argv) <= 2

for n in SIZES:
    p = rng.integers(0, cfg["vocab"], size=n).astype(np.int32)
    for _ in range(3):   # eager, capture, first replay
        m.prefill(p, want_logits=False)
    m.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        m.prefill(p, want_logits=False)
    m.sync()
    tb = (time.perf_counter() - t0) / 10
    if not SERIAL:
        print("n=%d batched %.3f ms (%.0f tok/s)" % (n, tb * 1e3, n / tb), flush=True)
        continue
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32); forced[:n] = p
    m.set_forced(forced); m.set_state(int(p[0]), 0)
    m.run_steps(0, n); m.sync()
    m.set_state(int(p[0]), 0)
    t0 = time.perf_counter()
    m.run_steps(0, n); m.sync()
    ts = time.perf_counter() - t0
    print("n=%d batched %.3f ms (%.0f tok/s)  token-serial %.3f ms (%.0f tok/s)  x%.1f" % (n, tb * 1e3, n / tb, ts * 1e3, n / ts, ts / tb), flush=True)
