"""One staged-GEMM shape for counter passes: python3 pmc_gemm.py [type] (q4|bf16)"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
ctx = R.Context(0)
t = {"q4": L.Q4, "bf16": L.BF16}[sys.argv[1] if len(sys.argv) > 1 else "q4"]
m, k, n = 6400, 5120, 4096
dw = ctx.quantize((torch.randn(m, k, device=ctx.device) * 0.02).to(torch.bfloat16), t)
x = torch.randn(n, k, device=ctx.device).to(torch.bfloat16); y = torch.zeros(n, m, device=ctx.device, dtype=torch.bfloat16)
d = dw.desc()
for _ in range(3):
    assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, n, 1.0, 0.0, 0, None) == 0
ctx.sync()
