"""Long-prompt prefill with and without the resident bf16 copies (kf_set_dequant_arena): Qwen3-0.6B Q4, 1024 / 2047 tokens; then a kernel-trace friendly loop of 3 prompts."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth

cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
rng = np.random.default_rng(5)
sizes = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1024, 2047]
modes = [bool(int(a)) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [False, True]
for n in sizes:
    p = rng.integers(0, cfg["vocab"], size=n).astype(np.int32)
    for on in modes:
        m.set_prefill_resident(on)
        m.sync(); t0 = time.perf_counter()
        m.prefill(p, want_logits=False); m.sync()
        first = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for _ in range(5):
            m.prefill(p, want_logits=False)
        m.sync()
        tb = (time.perf_counter() - t0) / 5 * 1e3
        print("n=%d resident=%d: %.3f ms (first call %.3f ms, %.1f MB resident)" % (n, on, tb, first, m.resident_bytes() / 1e6), flush=True)
