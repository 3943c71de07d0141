import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, synth
from oracle import oracle as O
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
prompt = np.random.default_rng(11).integers(0, cfg["vocab"], size=128)
m.set_prefill_mode(0)
serial = m.generate(prompt, 24, use_graph=True)
ks, vs = m.kv_to_host(); ks, vs = ks[:, :128].copy(), vs[:, :128].copy()
nxt, logits = m.prefill(prompt)
kb, vb = m.kv_to_host()
for l in range(cfg["n_layer"]):
    a, b = O.bf16_to_f32(kb[l, :128]), O.bf16_to_f32(ks[l])
    c, d = O.bf16_to_f32(vb[l, :128]), O.bf16_to_f32(vs[l])
    e = np.abs(a - b); f = np.abs(c - d)
    print(l, "K rel %.4f (rms %.5f) at tok %d | V rel %.4f (rms %.5f)" % (e.max() / np.abs(b).max(), np.sqrt((e**2).mean()) / np.abs(b).max(), np.unravel_index(e.argmax(), e.shape)[0], f.max() / np.abs(d).max(), np.sqrt((f**2).mean()) / np.abs(d).max()))
m.set_prefill_mode(1)
batched = m.generate(prompt, 24, use_graph=True)
print(serial); print(batched)
