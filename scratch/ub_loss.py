import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
ctx = Context(0); dev = ctx.device
B, T, V, P = 8, 1024, 50257, 50264
lg0 = (torch.randn(B * T, P, device=dev) * 0.5).to(torch.bfloat16)
tg = torch.randint(0, V, (B * T,), device=dev, dtype=torch.int32)
losses = torch.zeros(B * T, dtype=torch.float32, device=dev)
for wd in (1, 0):
    lg = lg0.clone()
    for _ in range(2): ctx.hip.kf_fused_classifier(ctx.h, lg.data_ptr(), losses.data_ptr(), None, 1.0, tg.data_ptr(), B, T, V, P, None, wd)
    ctx.sync(); e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    n = 5
    for _ in range(n): ctx.hip.kf_fused_classifier(ctx.h, lg.data_ptr(), losses.data_ptr(), None, 1.0, tg.data_ptr(), B, T, V, P, None, wd)
    ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / n
    byt = B * T * V * 2 * (2 if wd else 1)
    print(f"fused_classifier 8x1024x50257 write_dlogits={wd}: {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s algorithmic")
