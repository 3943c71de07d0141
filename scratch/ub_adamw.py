"""kf_adamw roofline: GPT2-1558M-sized parameter vector, bf16 and fp32 moments."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
ctx = R.Context(0); dev = ctx.device
n = 1_558_000_000 // 8 * 8
for mv, name, bpp in ((L.BF16, "bf16 moments", 16), (L.F32, "fp32 moments", 24)):
    p = (torch.randn(n, device=dev) * 0.02).to(torch.bfloat16); g = (torch.randn(n, device=dev) * 0.01).to(torch.bfloat16)
    m = torch.zeros(n, device=dev, dtype=torch.bfloat16 if mv == L.BF16 else torch.float32); v = torch.zeros_like(m)
    def run(s): L.check(ctx.hip.kf_adamw(ctx.h, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, mv, 3e-4, 0.9, 0.95, 0.1, 0.05, 1e-8, 0.1, 1.0, s, None), "adamw")
    for s in range(3): run(s)
    e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for s in range(10): run(10 + s)
    ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / 10
    print("AdamW %s: %d params %.3f ms  %.0f GB/s (%d B/param) = %.1f%% of 8 TB/s" % (name, n, ms, n * bpp / ms / 1e6, bpp, n * bpp / ms / 1e6 / 80), flush=True)
    del p, g, m, v
