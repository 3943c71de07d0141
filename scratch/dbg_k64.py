import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
from oracle import oracle as O
ctx = R.Context(0)
for (m, k, n, t) in ((64, 192, 32, L.BF16), (64, 192, 32, L.Q4), (64, 1600, 32, L.Q4), (4800, 1600, 96, L.Q4)):
    rng = np.random.default_rng(1)
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(m, k)).astype(np.float32)); x = O.f32_to_bf16(rng.normal(0, 1.0, size=(n, k)).astype(np.float32))
    ow = O.quantize(w, m, k, t); dw = ctx.upload_blob(t, m, k, ow.blob())
    xd = torch.from_numpy(x.view(np.int16)).to(ctx.device); y = torch.zeros(n, m, dtype=torch.bfloat16, device=ctx.device)
    d = dw.desc()
    print("launch", m, k, n, t, flush=True)
    rc = ctx.hip.kf_linear(ctx.h, C.byref(d), xd.data_ptr(), y.data_ptr(), None, n, 1.0, 0.0, 0, None)
    ctx.sync()
    exact = O.bf16_to_f32(x).astype(np.float64) @ O.bf16_to_f32(O.dequant(ow)).astype(np.float64).T
    got = y.float().cpu().numpy()
    print("  rc", rc, "err", np.abs(got - exact).max() / np.abs(exact).max(), flush=True)
