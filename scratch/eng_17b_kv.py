import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import synth, lib as L
cfg = dict(synth.CONFIGS["qwen3-1.7b"]); cfg["n_layer"] = 4
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
toks = np.random.default_rng(7).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
import _knobs; _knobs.apply(m.hip)
m.set_forced(toks); m.set_engine_autotune(0); m.set_canonical(0)
kv = {}
for eng in (True, False):
    m.set_engine(eng)
    for p in (0, 1):
        m.set_state(int(toks[p]), p); m.run_steps(p, 1, True); m.sync()
    k, v = m.kv_to_host()
    kv[eng] = (k[:, :2].copy(), v[:, :2].copy(), m.logits().copy())
for l in range(cfg["n_layer"]):
    for p in (0, 1):
        print("layer %d pos %d: K rows differ in %d elements, V in %d" % (l, p, int((kv[True][0][l, p] != kv[False][0][l, p]).sum()), int((kv[True][1][l, p] != kv[False][1][l, p]).sum())))
print("logits differ:", int((kv[True][2] != kv[False][2]).sum()))
