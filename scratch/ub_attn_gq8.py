"""Canonical decode attention of a GQA-8 model (64 query / 8 kv heads of 128) at 2047 / 4095 keys: KF_ATTN_GQ_SPLIT = 2 or 4 workgroups per (kv-head, slice)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
import _knobs
ctx = Context(0); dev = ctx.device
_knobs.apply(ctx.hip)
nh, nkv, hd, S = int(os.environ.get('NH', '64')), 8, 128, 4096
kc = torch.randn(S, nkv * hd, device=dev).to(torch.bfloat16); vc = torch.randn(S, nkv * hd, device=dev).to(torch.bfloat16)
q = torch.randn(nh * hd, device=dev).to(torch.bfloat16); kraw = torch.randn(nkv * hd + nkv * hd, device=dev).to(torch.bfloat16)
qn = torch.ones(hd, device=dev, dtype=torch.bfloat16); table = ctx.rope_table(S, hd, 1e6)
ws = ctx._ws(nh, hd); out = torch.zeros(nh * hd, dtype=torch.bfloat16, device=dev)
dp = torch.zeros(1, dtype=torch.int32, device=dev)
for canon in (1, 0):
    L.check(ctx.hip.kf_set_canonical(ctx.h, canon))
    for pos in (2047, 4095):
        dp[0] = pos
        f = lambda: L.check(ctx.hip.kf_attn_block(ctx.h, _ptr(q), _ptr(kraw), _ptr(kc), _ptr(vc), _ptr(out), _ptr(qn), _ptr(qn), _ptr(table), pos, _ptr(dp), nh, nkv, hd, nkv * hd, 1e-6, _ptr(ws)))
        for _ in range(5): f()
        ctx.sync()
        e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
        for _ in range(50): f()
        ctx.record(e1); ctx.sync()
        print("canonical=%d pos=%d: %.2f us  digest %d" % (canon, pos, ctx.elapsed_ms(e0, e1) * 1e3 / 50, int(out.view(torch.int16).to(torch.int64).sum().item())), flush=True)
