"""Driver of the instrumented gemm_direct_kernel (scratch/dbg_gemm_patch.py): the four GEMMs of a 0.6B layer on a 128-token batch, cold weights."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = Context(0); dev = ctx.device
def rw(m, k): return (torch.randn(m, k, device=dev) * 0.02).to(torch.bfloat16)
big = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
for (m, k) in [(4096, 1024), (1024, 2048), (1024, 3072)]:
    w = ctx.quantize(rw(m, k), L.Q4); d = w.desc()
    x = torch.randn(n, k, device=dev).to(torch.bfloat16); y = torch.zeros(n, m, device=dev, dtype=torch.bfloat16)
    for _ in range(4):
        if not os.environ.get("WARM"):
            big.add_(1); torch.cuda.synchronize()   # push the weights out of the caches
        L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), _ptr(x), _ptr(y), None, n, 1.0, 0.0, 0, None))
        ctx.sync()
