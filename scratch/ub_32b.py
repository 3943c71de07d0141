"""The four mat-vec launches of a Qwen3-32B decode layer, timed alone (cold weights: 6 rotating copies), for KF_GEMV_WAVES / KF_GEMV_G sweeps."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
import _knobs
ctx = R.Context(0); dev = ctx.device
_knobs.apply(ctx.hip)   # KF_GEMV_WAVES / KF_Q4_PERM ... (scratch/_knobs.py)
dim, qd, kvd, ffn = 5120, 8192, 1024, 25600
NS = 6
def mk(m, k): return [ctx.quantize((torch.randn(m, k, device=dev) * 0.02).to(torch.bfloat16), L.Q4) for _ in range(NS)]
wq, wk, wv, wo, wg, wu, wd = mk(qd, dim), mk(kvd, dim), mk(kvd, dim), mk(dim, qd), mk(ffn, dim), mk(ffn, dim), mk(dim, ffn)
x = torch.randn(dim, device=dev).to(torch.bfloat16); nw = torch.ones(dim, device=dev, dtype=torch.bfloat16)
xq = torch.randn(qd, device=dev).to(torch.bfloat16); xf = torch.randn(ffn, device=dev).to(torch.bfloat16)
yq, yk, yv = (torch.zeros(n, device=dev, dtype=torch.bfloat16) for n in (qd, kvd, kvd))
yo = torch.zeros(dim, device=dev, dtype=torch.bfloat16); act = torch.zeros(ffn, device=dev, dtype=torch.bfloat16)
def qkv(i):
    ds = [w[i % NS].desc() for w in (wq, wk, wv)]
    wp = (C.c_void_p * 3)(*[C.addressof(d) for d in ds]); yp = (C.c_void_p * 3)(yq.data_ptr(), yk.data_ptr(), yv.data_ptr())
    L.check(ctx.hip.kf_norm_linear(ctx.h, x.data_ptr(), nw.data_ptr(), 1e-6, 3, wp, yp, None, 0, None), "qkv")
def oproj(i):
    d = wo[i % NS].desc(); L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), xq.data_ptr(), yo.data_ptr(), None, 1, 1.0, 0.0, 1, x.data_ptr()), "o")
def gateup(i):
    g, u = wg[i % NS].desc(), wu[i % NS].desc()
    L.check(ctx.hip.kf_norm_gateup_swiglu(ctx.h, x.data_ptr(), nw.data_ptr(), 1e-6, C.byref(g), C.byref(u), act.data_ptr()), "gu")
def down(i):
    d = wd[i % NS].desc(); L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), xf.data_ptr(), yo.data_ptr(), None, 1, 1.0, 0.0, 1, x.data_ptr()), "d")
tot = 0.0
for name, f, mb in (("qkv 10240x5120", qkv, 10240 * 5120 * 0.53125), ("o 5120x8192", oproj, 5120 * 8192 * 0.53125), ("gate/up 2x25600x5120", gateup, 2 * 25600 * 5120 * 0.53125),
                    ("down 5120x25600", down, 5120 * 25600 * 0.53125)):
    for i in range(NS): f(i)
    e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for i in range(30): f(i)
    ctx.record(e1); us = ctx.elapsed_ms(e0, e1) * 1e3 / 30; tot += us
    print("%-24s %7.1f us  %6.0f GB/s" % (name, us, mb / us / 1e3), flush=True)
print("layer total %.1f us (x64 = %.2f ms)" % (tot, tot * 64 / 1e3))
