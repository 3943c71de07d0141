"""The persistent decode engine at one position (kf::engine_kernel: embedding row + 28 layers + final norm + LM head + greedy pick of a Qwen3-0.6B 4-bit decode
step in ONE launch), launched eagerly.  Consecutive launches stream 545 MB + the K/V rows each, so nothing comes from a cache.  The stand-alone target of the
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes behind profiles/r03_pmc_engine.json (--pmc with bench.py itself -- hipGraph + torch -- crashes the
profiler on this pool).  usage: ub_engine.py [position, default 2037 = the mean position of the driver's timed region (--steps 20 --warmup 5)]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L
from koifish_amd import synth
pos = int(sys.argv[1]) if len(sys.argv) > 1 else 2037
cfg = synth.CONFIGS["qwen3-0.6b"]
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
ctx = m._ctx
rng = np.random.default_rng(0)
toks = rng.integers(0, cfg["vocab"], pos + 1).astype(np.int32)
m.prefill(toks[:pos], 0, want_logits=False)          # KV rows 0..pos-1
forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
forced[:pos + 1] = toks
m.set_forced(forced)
m.set_state(int(toks[pos]), pos)
m.run_steps(pos, 1, False)
m.sync()
assert m.engine_steps() > 0, "engine does not serve this model"
reps = int(os.environ.get("REPS", "20"))
e0, e1 = ctx.event(), ctx.event()
ms = 0.0
for r in range(reps):
    m.set_state(int(toks[pos]), pos)
    ctx.record(e0)
    m.run_steps(pos, 1, False)
    ctx.record(e1)
    m.sync()
    ms += ctx.elapsed_ms(e0, e1)
m.engine_check()
nbytes = m.step_bytes(pos)
print("position %d: %d launches, %.1f us per launch, algorithmic bytes per launch %d (%.1f GB/s)" % (pos, reps, ms / reps * 1e3, nbytes, nbytes / (ms / reps * 1e-3) / 1e9))
