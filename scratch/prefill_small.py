"""128 / 256-token prompts through the resident-copy tile-GEMM routes (knob resident_min) against the in-register direct kernels: ms per prompt.  python3 scratch/prefill_small.py"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
m.set_prefill_resident(True, 96 << 30)
hip = m.hip
hip.kfdbg_set_knob.argtypes = [C.c_char_p, C.c_long]
long_p = np.random.default_rng(5).integers(0, cfg["vocab"], size=2047).astype(np.int32)
m.prefill(long_p, want_logits=False)   # fills the resident copies
m.sync()
print("resident bytes", m.resident_bytes())
for n in (128, 256, 64):
    p = np.random.default_rng(6).integers(0, cfg["vocab"], size=n).astype(np.int32)
    for rmin in (320, 64):
        assert hip.kfdbg_set_knob(b"resident_min", rmin) == 0
        ids = []
        for _ in range(3):
            nxt, _ = m.prefill(p, want_logits=False)
        m.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            nxt, _ = m.prefill(p, want_logits=False)
        m.sync()
        print("tokens %4d resident_min %3d: %.3f ms per prompt, next id %d" % (n, rmin, (time.perf_counter() - t0) * 100, nxt), flush=True)
