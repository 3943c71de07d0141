"""Attention forward (batched) and backward at config-3 size, three launches each, for rocprofv3 counter passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
ctx = Context(0); dev = ctx.device
B, T, H, hd = 8, 1024, 25, 64
C = H * hd
qkv = torch.randn(B * T, 3 * C, device=dev).to(torch.bfloat16); qc = qkv[:, :C].contiguous(); o = torch.zeros(B * T, C, device=dev, dtype=torch.bfloat16); dO = torch.randn_like(o)
dqkv = torch.zeros_like(qkv); sc = torch.zeros(ctx.hip.kf_attn_backward_scratch_bytes(T, H, B) // 4 + 1, dtype=torch.float32, device=dev)
for _ in range(3):
    assert ctx.hip.kf_attn_prefill_batch(ctx.h, qc.data_ptr(), qkv[:, C:].data_ptr(), qkv[:, 2 * C:].data_ptr(), o.data_ptr(), T, C, H, H, hd, 3 * C, B) == 0
    assert ctx.hip.kf_attn_backward(ctx.h, qkv[:, :C].data_ptr(), qkv[:, C:2 * C].data_ptr(), qkv[:, 2 * C:].data_ptr(), 3 * C, o.data_ptr(), dO.data_ptr(), C,
                                    dqkv[:, :C].data_ptr(), dqkv[:, C:2 * C].data_ptr(), dqkv[:, 2 * C:].data_ptr(), 3 * C, T, H, H, hd, B, sc.data_ptr()) == 0
ctx.sync()
