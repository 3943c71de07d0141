"""kf_embed_backward at BASELINE config 3's size: 8 x 1024 positions, C 1600, GPT-2 vocabulary (random ids: ~7500 distinct tokens)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
ctx = Context(0); dev = ctx.device
B, T, C, V = 8, 1024, 1600, 50257
dout = torch.randn(B * T, C, device=dev).to(torch.bfloat16); tok = torch.randint(0, V, (B * T,), device=dev, dtype=torch.int32)
dwte = torch.zeros(V, C, device=dev, dtype=torch.bfloat16); dwpe = torch.zeros(T, C, device=dev, dtype=torch.bfloat16)
run = lambda: ctx.hip.kf_embed_backward(ctx.h, dwte.data_ptr(), C, dwpe.data_ptr(), dout.data_ptr(), tok.data_ptr(), B, T, C, V)
for _ in range(2): run()
ctx.sync(); e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
for _ in range(5): run()
ctx.record(e1); print("embed backward 8 x 1024 x 1600: %.3f ms" % (ctx.elapsed_ms(e0, e1) / 5))
# natural-text-like ids (Zipf): a few tokens own many positions
tok2 = torch.clamp((torch.rand(B * T, device=dev) ** 6 * V).to(torch.int32), 0, V - 1)
run2 = lambda: ctx.hip.kf_embed_backward(ctx.h, dwte.data_ptr(), C, dwpe.data_ptr(), dout.data_ptr(), tok2.data_ptr(), B, T, C, V)
for _ in range(2): run2()
ctx.sync(); ctx.record(e0)
for _ in range(5): run2()
ctx.record(e1); print("embed backward, skewed ids (%d distinct): %.3f ms" % (int(torch.unique(tok2).numel()), ctx.elapsed_ms(e0, e1) / 5))
