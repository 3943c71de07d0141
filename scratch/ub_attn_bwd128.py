"""kf_attn_backward at the Qwen3-0.6B training size: 8 sequences x 1024 tokens, 16 / 8 heads x 128."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
ctx = Context(0); dev = ctx.device
B, T, H, KV, hd = 8, 1024, 16, 8, 128
Cq, Ck = H * hd, KV * hd; W = Cq + 2 * Ck
qkv = torch.randn(B * T, W, device=dev).to(torch.bfloat16); o = torch.randn(B * T, Cq, device=dev).to(torch.bfloat16); dO = torch.randn_like(o)
dqkv = torch.zeros_like(qkv); sc = torch.zeros(ctx.hip.kf_attn_backward_scratch_bytes(T, H, B) // 4 + 1, dtype=torch.float32, device=dev)
def run():
    assert ctx.hip.kf_attn_backward(ctx.h, qkv[:, :Cq].data_ptr(), qkv[:, Cq:].data_ptr(), qkv[:, Cq + Ck:].data_ptr(), W, o.data_ptr(), dO.data_ptr(), Cq,
                                    dqkv[:, :Cq].data_ptr(), dqkv[:, Cq:].data_ptr(), dqkv[:, Cq + Ck:].data_ptr(), W, T, H, KV, hd, B, sc.data_ptr()) == 0
for _ in range(2): run()
ctx.sync(); e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
for _ in range(3): run()
ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / 3
print("attention backward 8 x 1024 x 16/8 x 128: %.3f ms (x28 layers = %.1f ms)" % (ms, ms * 28))
