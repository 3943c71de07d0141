"""distribution of the decode-step logit error against the oracle after prefills of several lengths / token seeds (the quantity tests/test_gpu_full_size.py bounds by 2^-6)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from koifish_amd import lib as L, synth
from oracle import oracle as O
from tests.test_gpu_full_size import _decode_after_prefill
cfg = dict(synth.CONFIGS["qwen3-0.6b"])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
om = O.from_device_model(m)
for P in (127, 255, 1023):
    for seed in (100 + P, 7, 8):
        errs = [e for (_, _, e, _) in _decode_after_prefill(m, om, cfg, P, 3, seed=seed)]
        print(P, seed, " ".join("%.4f" % e for e in errs), flush=True)
