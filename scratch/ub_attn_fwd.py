"""kf_attn_prefill_batch at config 3's size (8 x 1024 tokens, 25 heads x 64) and at a Qwen3-0.6B prompt (2048 tokens, 16 / 8 heads x 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
for B, T, H, KV, hd in ((8, 1024, 25, 25, 64), (1, 2048, 16, 8, 128)):
    C = H * hd; Ck = KV * hd
    q = torch.randn(B * T, C, device=dev).to(torch.bfloat16); k = torch.randn(B * T, Ck, device=dev).to(torch.bfloat16); v = torch.randn(B * T, Ck, device=dev).to(torch.bfloat16)
    o = torch.zeros(B * T, C, device=dev, dtype=torch.bfloat16)
    def run(): L.check(ctx.hip.kf_attn_prefill_batch(ctx.h, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), T, C, H, KV, hd, Ck, B), "attn")
    for _ in range(2): run()
    ctx.sync(); e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for _ in range(5): run()
    ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / 5
    fl = 4.0 * B * H * (T * T / 2) * hd
    print("attention forward B %d T %d heads %d/%d x %d: %.1f us  %.0f TFLOP/s (causal flops)" % (B, T, H, KV, hd, ms * 1e3, fl / ms / 1e9))
