"""timing of the XCD-confined engine forms on Qwen3-0.6B at positions 2028..2047 (synthetic K / V rows in front): n_seq x form (batched NB = 2 / 4, the round-5 two-decoders form, wave / depth variants)"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from koifish_amd import lib as L, synth
from koifish_amd.runtime import XcdReplicas

cfg = dict(synth.CONFIGS["qwen3-0.6b"])
S = cfg["max_seq"]
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
m.set_canonical(True)
forms = [(8, None), (16, None), (16, (8, 8)), (16, (12, 8)), (32, None), (32, (8, 8)), (24, None)]
if len(sys.argv) > 1:
    forms = [eval(a) for a in sys.argv[1:]]
K, W = 20, 5
first = S - K
bytes_tok = float(np.mean([m.step_bytes(p) for p in range(first, S)]))
for n_seq, var in forms:
    xr = XcdReplicas(m, n_seq)
    if var:
        xr.variant(*var)
    kvb = cfg["n_layer"] * S * cfg["n_kv"] * cfg["head_dim"] * 2
    bench._fill_kv_synthetic(m.hip, C.c_void_p(m.host.kfh_ctx(m.h)), [(f(xr.h, s), kvb) for s in range(n_seq) for f in (xr.host.kfh_xr_kcache, xr.host.kfh_xr_vcache)])
    best = None
    for rep in range(3):
        for s in range(n_seq):
            xr.set_state(s, 1 + s, first - W)
        xr.run_steps(W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xr.run_steps(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    xr.check()
    tps = n_seq * K / best
    print("n_seq %2d variant %-10s  %.3f ms per step of all  %8.1f tok/s  frac_vs_single_seq_roofline %.4f" % (n_seq, var, best * 1e3 / K, tps, bytes_tok * tps / 8e12), flush=True)
    xr.close()
