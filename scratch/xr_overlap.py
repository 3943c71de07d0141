"""How fast is the two-workgroups-per-CU instantiation (128 registers) when a decoder has its XCD to itself, and when it shares it?  n_seq = 9: XCD 0 carries sequences 0 and 8,
XCDs 1 .. 7 one sequence each -- layer period of sequence 5 (alone on XCD 5) against sequence 0 (shared XCD)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from koifish_amd import lib as L, synth
from koifish_amd.runtime import XcdReplicas
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16); m.set_canonical(True)
nl = cfg["n_layer"]
for n_seq, seq in ((9, 5), (9, 0), (9, 8), (16, 5)):
    xr = XcdReplicas(m, n_seq)
    xr.set_steps_per_launch(2)
    xr.stamps(seq, 5, 2, nl)
    for s in range(n_seq): xr.set_state(s, 1 + s, 2028)
    xr.run_steps(2); m.sync(); xr.check()
    st = xr.stamps(0, 0, -2, nl).astype(np.int64)
    print("n_seq %d, sequence %d: layer period %.1f us (step 1)" % (n_seq, seq, (st[1, nl - 1, 0] - st[1, 1, 0]) / 100.0 / (nl - 2)), flush=True)
    xr.close()
