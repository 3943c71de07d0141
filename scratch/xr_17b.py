import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from koifish_amd import lib as L, synth
from koifish_amd.runtime import XcdReplicas
cfg = dict(synth.CONFIGS[os.environ.get("CONFIG", "qwen3-1.7b")])
S = cfg["max_seq"]
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
m.set_canonical(True)
K, W = 20, 5
first = S - K
for n_seq, var in ([(8, None), (16, None), (16, (-2, 1))] if len(sys.argv) < 2 else [eval(x) for x in sys.argv[1:]]):
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
    print("n_seq %2d variant %-10s  %.3f ms per step of all  %8.1f tok/s" % (n_seq, var, best * 1e3 / K, n_seq * K / best), flush=True)
    xr.close()
