"""kf_linear_backward on one shape (bf16 weight): python ub_lbw_one.py OC IC n  -- for per-kernel traces of the two backward GEMMs"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
OC, IC, n = [int(v) for v in sys.argv[1:4]]
dw = ctx.quantize((torch.randn(OC, IC, device=dev) * 0.02).to(torch.bfloat16), L.BF16)
dIn = torch.randn(n, OC, device=dev).to(torch.bfloat16); inp = torch.randn(n, IC, device=dev).to(torch.bfloat16)
delta = torch.zeros(n, IC, device=dev, dtype=torch.bfloat16); gW = torch.zeros(OC, IC, device=dev, dtype=torch.bfloat16)
sc = torch.empty(ctx.hip.kf_linear_backward_scratch_bytes(OC, IC, n) + 256, dtype=torch.uint8, device=dev); sp = (sc.data_ptr() + 255) & ~255
d = dw.desc()
def run(): L.check(ctx.hip.kf_linear_backward(ctx.h, C.byref(d), dIn.data_ptr(), inp.data_ptr(), delta.data_ptr(), gW.data_ptr(), None, n, 0, sp), "bwd")
for _ in range(2): run()
ctx.sync(); e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
for _ in range(5): run()
ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / 5
print("OC %d IC %d n %d: %.3f ms  %.0f TFLOP/s" % (OC, IC, n, ms, 4.0 * n * OC * IC / ms / 1e9))
