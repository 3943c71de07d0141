"""Prefill of a prompt that fills the context (MAX_SEQ - 1 tokens, default 2047) with different token-batch (chunk) sizes: python prefill_chunk.py [chunk ...]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
cfg["max_seq"] = int(os.environ.get("MAX_SEQ", "2048"))
p = np.random.default_rng(5).integers(0, cfg["vocab"], size=cfg["max_seq"] - 1).astype(np.int32)
for ch in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
    m.set_prefill_mode(1, ch)
    m.prefill(p, want_logits=False); m.sync()
    t0 = time.perf_counter()
    for _ in range(3): m.prefill(p, want_logits=False)
    m.sync()
    print("chunk %d: %.3f ms" % (ch, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    del m
