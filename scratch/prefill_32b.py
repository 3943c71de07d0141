"""Qwen3-32B-shaped model (4-bit, synthetic) on one GPU: a 2047-token prompt with and without the resident bf16 copies (62 GB of the 288)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth
cfg = dict(synth.CONFIGS["qwen3-32b"])
n_layer = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n_layer"]
cfg["n_layer"] = n_layer
t0 = time.perf_counter()
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
print("built %d layers in %.1f s" % (n_layer, time.perf_counter() - t0), flush=True)
rng = np.random.default_rng(5)
for n in (512, 2047):
    p = rng.integers(0, cfg["vocab"], size=n).astype(np.int32)
    for on in (False, True):
        m.set_prefill_resident(on)
        m.sync(); t0 = time.perf_counter()
        nxt, lg = m.prefill(p); m.sync()
        first = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for _ in range(3):
            m.prefill(p, want_logits=False)
        m.sync()
        tb = (time.perf_counter() - t0) / 3 * 1e3
        print("n=%d resident=%d: %.2f ms (%.0f tok/s; first call %.1f ms, %.2f GB resident) next id %d" % (n, on, tb, n / tb * 1e3, first, m.resident_bytes() / 1e9, nxt), flush=True)
