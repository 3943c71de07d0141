"""kf_linear_backward on the four GEMM shapes of a GPT2-1558M block, 8 x 1024 tokens: input gradient + weight gradient + bias gradient."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
n = 8192
tot = 0.0
for name, OC, IC, t in (("qkv", 4800, 1600, L.F8E5M2), ("proj", 1600, 1600, L.F8E5M2), ("fc", 6400, 1600, L.Q4), ("proj2", 1600, 6400, L.Q4)):
    dw = ctx.quantize((torch.randn(OC, IC, device=dev) * 0.02).to(torch.bfloat16), t)
    dIn = torch.randn(n, OC, device=dev).to(torch.bfloat16); inp = torch.randn(n, IC, device=dev).to(torch.bfloat16)
    delta = torch.zeros(n, IC, device=dev, dtype=torch.bfloat16); gW = torch.zeros(OC, IC, device=dev, dtype=torch.bfloat16); gb = torch.zeros(OC, device=dev, dtype=torch.bfloat16)
    sc = torch.empty(ctx.hip.kf_linear_backward_scratch_bytes(OC, IC, n) + 256, dtype=torch.uint8, device=dev); sp = (sc.data_ptr() + 255) & ~255
    d = dw.desc()
    def run(): L.check(ctx.hip.kf_linear_backward(ctx.h, C.byref(d), dIn.data_ptr(), inp.data_ptr(), delta.data_ptr(), gW.data_ptr(), gb.data_ptr(), n, 0, sp), "bwd")
    for _ in range(2): run()
    ctx.sync(); e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for _ in range(5): run()
    ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / 5; tot += ms
    print("%-6s OC %d IC %d: %.3f ms  %.0f TFLOP/s (4 n OC IC flops)" % (name, OC, IC, ms, 4.0 * n * OC * IC / ms / 1e9))
print("block total %.2f ms (x48 = %.0f ms)" % (tot, tot * 48))
