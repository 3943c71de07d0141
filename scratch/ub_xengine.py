"""The XCD-confined engines alone (the `--pmc` / kernel-trace target behind profiles/r05_pmc_xengine*.json): eight sequences of Qwen3-0.6B 4-bit, a few launches of STEPS steps at
position POS.   python scratch/ub_xengine.py [pos=2037] [steps=4] [launches=3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd.runtime import XcdReplicas

pos = int(sys.argv[1]) if len(sys.argv) > 1 else 2037
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
m.set_canonical(True)
n_seq = int(os.environ.get("NSEQ", "8"))
xr = XcdReplicas(m, n_seq)
xr.set_steps_per_launch(steps)
for it in range(launches):
    for s in range(n_seq):
        xr.set_state(s, 1 + s, pos)
    xr.run_steps(steps)
    m.sync()
xr.check()
bytes_step = sum(m.step_bytes(pos + i) for i in range(steps))
print("xengine: %d sequences x %d steps per launch at positions %d..%d, algorithmic bytes per launch %d (one sequence: %d)" % (n_seq, steps, pos, pos + steps - 1, n_seq * bytes_step, bytes_step))
xr.close()
m.close()
