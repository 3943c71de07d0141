"""The persistent decode engine alone (kf::engine_kernel: all 28 layers of a Qwen3-0.6B 4-bit decode step in one launch) at one position, launched
eagerly with a 320 MB streaming pass between launches so that the layer weights come from HBM as in the real step.  The stand-alone target of
the `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes behind profiles/r02_pmc_engine.json (--pmc with bench.py itself -- hipGraph + torch --
crashes the profiler on this pool).  usage: ub_engine.py [position, default 1087 = the mean position of bench.py's timed region]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L
from koifish_amd import synth
pos = int(sys.argv[1]) if len(sys.argv) > 1 else 1087
cfg = synth.CONFIGS["qwen3-0.6b"]
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
ctx = m._ctx
rng = np.random.default_rng(0)
toks = rng.integers(0, cfg["vocab"], pos).astype(np.int32)
m.prefill(toks, 0, want_logits=False)          # KV rows 0..pos-1
m.set_state(int(toks[-1]), pos)
assert m.engine_only(1), "engine does not serve this model"
m.sync()
flush = torch.empty(320 << 20, dtype=torch.uint8, device=ctx.device)
reps = int(os.environ.get("REPS", "20"))
e0, e1 = ctx.event(), ctx.event()
ms = 0.0
for r in range(reps):
    flush.add_(1)
    ctx.record(e0)
    m.engine_only(1)
    ctx.record(e1)
    m.sync()
    ms += ctx.elapsed_ms(e0, e1)
m.engine_check()
kvd = cfg["n_kv"] * cfg["head_dim"]
wbytes = sum(w.algorithmic_bytes() for (layer, slot), w in m.weights.items() if layer >= 0) + cfg["n_layer"] * (2 * cfg["dim"] + 2 * cfg["head_dim"]) * 2
nbytes = wbytes + 2 * cfg["n_layer"] * (pos + 1) * kvd * 2 + 2 * cfg["n_layer"] * kvd * 2
print("position %d: %d launches, %.1f us per launch, algorithmic bytes per launch %d (%.1f GB/s)" % (pos, reps, ms / reps * 1e3, nbytes, nbytes / (ms / reps * 1e-3) / 1e9))
