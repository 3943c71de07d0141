"""Qwen3-1.7B shape (4-bit, synthetic): decode through the persistent engine against the per-layer launches -- ids equal, ms/step of each, both summation orders."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import synth, lib as L
cfg = dict(synth.CONFIGS["qwen3-1.7b"])
if len(sys.argv) > 1: cfg["n_layer"] = int(sys.argv[1])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
forced = np.random.default_rng(7).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
forced[1960:] = -1   # free-running from 1960
m.set_forced(forced)
m.set_engine_autotune(0)
for canon in (1, 0):
    m.set_canonical(canon)
    res = {}
    for eng in (True, False):
        m.set_engine(eng)
        p0 = 1900
        m.set_state(int(forced[p0]), p0)
        m.run_steps(p0, 40, True); m.sync()
        t0 = time.perf_counter()
        m.run_steps(p0 + 40, 80, True); m.sync()
        dt = (time.perf_counter() - t0) / 80 * 1e3
        if eng: m.engine_check()
        ids = np.asarray(m.tokens_out(cfg["max_seq"])[p0 + 1:p0 + 120]).copy()
        res[eng] = (dt, ids, m.engine_steps() if eng else 0, m.engine_why())
    print("canonical=%d: engine %.4f ms/step (%d engine steps; why='%s'), per-layer launches %.4f ms/step, ids equal: %s" % (canon, res[True][0], res[True][2], res[True][3], res[False][0], np.array_equal(res[True][1], res[False][1])), flush=True)
