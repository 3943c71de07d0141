"""Batched prefill only (for rocprofv3): python3 prefill_prof.py N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
p = np.random.default_rng(5).integers(0, cfg["vocab"], size=n).astype(np.int32)
for _ in range(4):
    m.prefill(p, want_logits=False)
m.sync()
