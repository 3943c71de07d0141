import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
from oracle import oracle as O
ctx = R.Context(0)
n = 1600 * 6400
rng = np.random.default_rng(n)
p = O.f32_to_bf16(rng.normal(0, 0.05, n).astype(np.float32)); g = O.f32_to_bf16(rng.normal(0, 0.01, n).astype(np.float32))
m = O.f32_to_bf16(rng.normal(0, 0.005, n).astype(np.float32)); v = O.f32_to_bf16(np.abs(rng.normal(0, 1e-4, n)).astype(np.float32))
hp = dict(lr=3e-4, beta1=0.9, beta2=0.95, b1c=float(np.float32(1 - 0.9 ** 3)), b2c=float(np.float32(1 - 0.95 ** 3)), eps=1e-8, wd=0.1, grad_scale=0.25, seed=4242)
t = lambda a: torch.from_numpy(a.view(np.int16).copy()).to(ctx.device)
dp, dg, dm, dv = t(p), t(g), t(m), t(v)
p0, g0, m0, v0 = p.copy(), g.copy(), m.copy(), v.copy()
O.adamw(p, g, m, v, **hp)
ctx.hip.kf_adamw(ctx.h, dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), n, L.BF16, hp["lr"], hp["beta1"], hp["beta2"], hp["b1c"], hp["b2c"], hp["eps"], hp["wd"], hp["grad_scale"], hp["seed"], None)
ctx.sync()
b = lambda x: x.cpu().numpy().view(np.uint16)
for name, got, want in (("p", b(dp), p), ("m", b(dm), m), ("v", b(dv), v)):
    bad = np.nonzero(got != want)[0]
    print(name, "mismatches", bad.size, "first", bad[:5], "thread/block of first", (bad[:3] // 8) % 512, (bad[:3] // 8) // 512)
    for i in bad[:3]:
        print("   idx", i, "got %04x want %04x" % (got[i], want[i]), "p0 %04x g0 %04x m0 %04x v0 %04x" % (p0[i], g0[i], m0[i], v0[i]))
