"""kf_attn_backward at BASELINE config 3's size: 8 sequences x 1024 tokens, 25 heads x 64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
ctx = Context(0); dev = ctx.device
B, T, H, hd = 8, 1024, 25, 64
C = H * hd
qkv = torch.randn(B * T, 3 * C, device=dev).to(torch.bfloat16); o = torch.randn(B * T, C, device=dev).to(torch.bfloat16); dO = torch.randn_like(o)
dqkv = torch.zeros_like(qkv); sc = torch.zeros(ctx.hip.kf_attn_backward_scratch_bytes(T, H, B) // 4 + 1, dtype=torch.float32, device=dev)
def run():   # the 8 sequences in one call
    assert ctx.hip.kf_attn_backward(ctx.h, qkv[:, :C].data_ptr(), qkv[:, C:2 * C].data_ptr(), qkv[:, 2 * C:].data_ptr(), 3 * C, o.data_ptr(), dO.data_ptr(), C,
                                    dqkv[:, :C].data_ptr(), dqkv[:, C:2 * C].data_ptr(), dqkv[:, 2 * C:].data_ptr(), 3 * C, T, H, H, hd, B, sc.data_ptr()) == 0
for _ in range(2): run()
ctx.sync(); e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
for _ in range(3): run()
ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / 3
flops = B * H * (T * T / 2) * hd * 2 * 8   # 8 hd MACs per (query, key) pair over the two kernels
print("attention backward 8 x 1024 x 25 x 64: %.2f ms  (%.0f TFLOP/s of fp32 VALU work; x48 layers = %.0f ms)" % (ms, flops / ms / 1e9, ms * 48))
