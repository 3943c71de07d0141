"""Stand-alone `rocprofv3 --pmc` target: the 1-bit (or ternary) mat-vec on a Qwen3-32B FFN shape (25600 x 5120), 20 launches.  usage: pmc_lowbit.py [1bit|ternary]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
kind = sys.argv[1] if len(sys.argv) > 1 else "1bit"
ctx = Context(0); dev = ctx.device
m, k = 25600, 5120
W = (torch.randn(m, k, device=dev) * 0.02).to(torch.bfloat16)
w = ctx.quantize(W, L.BOOL1 if kind == "1bit" else L.T_SIGN)
x = torch.randn(k, device=dev).to(torch.bfloat16); y = torch.zeros(m, dtype=torch.bfloat16, device=dev)
d = w.desc()
for _ in range(20):
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), _ptr(x), _ptr(y), None, 1, 1.0, 0.0, 0, None))
ctx.sync()
