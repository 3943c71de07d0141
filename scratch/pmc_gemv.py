"""One large 4-bit mat-vec for counter passes (25600 x 5120, cold weights: 6 copies)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
ctx = R.Context(0); dev = ctx.device
M, K = 25600, 5120
ws = [ctx.quantize((torch.randn(M, K, device=dev) * 0.02).to(torch.bfloat16), L.Q4) for _ in range(6)]
x = torch.randn(K, device=dev).to(torch.bfloat16); y = torch.zeros(M, device=dev, dtype=torch.bfloat16)
for i in range(12):
    d = ws[i % 6].desc()
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 0, None), "lin")
ctx.sync()
