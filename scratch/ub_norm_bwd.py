"""kf_norm_backward on BASELINE config 3's activation size (8 x 1024 rows x 1600) and on 8192 x 5120; kf_gelu_backward on 8192 x 6400."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
ctx = Context(0); dev = ctx.device
def t(fn, reps=10):
    for _ in range(2): fn()
    ctx.sync(); e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for _ in range(reps): fn()
    ctx.record(e1); return ctx.elapsed_ms(e0, e1) / reps
for rows, C in ((8192, 1600), (8192, 5120), (8192, 1024)):
    for ln in (1, 0):
        x = torch.randn(rows, C, device=dev).to(torch.bfloat16); do = torch.randn_like(x); di = torch.randn_like(x)
        w = torch.ones(C, device=dev, dtype=torch.bfloat16); dw = torch.zeros_like(w); db = torch.zeros_like(w)
        mean = x.float().mean(1); rstd = 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5)
        sc = torch.empty(ctx.hip.kf_norm_backward_scratch_bytes(rows, C, ln) // 8 + 1, dtype=torch.float64, device=dev)
        ms = t(lambda: ctx.hip.kf_norm_backward(ctx.h, di.data_ptr(), dw.data_ptr(), db.data_ptr() if ln else None, do.data_ptr(), x.data_ptr(), w.data_ptr(),
                                               mean.data_ptr() if ln else None, rstd.data_ptr(), rows, C, sc.data_ptr()))
        print("norm backward %s %d x %d: %.3f ms  %.0f GB/s (4 x rows x C x 2 B)" % ("LN " if ln else "RMS", rows, C, ms, 4 * rows * C * 2 / ms / 1e6))
n = 8192 * 6400
x = torch.randn(n, device=dev).to(torch.bfloat16); d = torch.randn_like(x)
ms = t(lambda: ctx.hip.kf_gelu_backward(ctx.h, d.data_ptr(), x.data_ptr(), n))
print("gelu backward %d: %.3f ms  %.0f GB/s (3 x n x 2 B)" % (n, ms, 3 * n * 2 / ms / 1e6))
g = torch.randn_like(x); dg = torch.empty_like(x)
ms = t(lambda: ctx.hip.kf_swiglu_backward(ctx.h, d.data_ptr(), dg.data_ptr(), g.data_ptr(), x.data_ptr(), n))
print("swiglu backward %d: %.3f ms  %.0f GB/s (5 x n x 2 B)" % (n, ms, 5 * n * 2 / ms / 1e6))
