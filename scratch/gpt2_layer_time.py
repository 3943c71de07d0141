"""One GPT2-1558M transformer block (n_embd 1600, 25 heads, ffn 6400), hybrid f8e5m2 / 4-bit weights, B x T = 8 x 1024 tokens: forward time."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
ctx = R.Context(0); dev = ctx.device
Cn, H, T, B = 1600, 25, 1024, 8
hd = Cn // H; N = B * T
mk = lambda m, k, t: ctx.quantize((torch.randn(m, k, device=dev) * 0.02).to(torch.bfloat16), t)
wqkv, wproj, wfc, wproj2 = mk(3 * Cn, Cn, L.F8E5M2), mk(Cn, Cn, L.F8E5M2), mk(4 * Cn, Cn, L.Q4), mk(Cn, 4 * Cn, L.Q4)
bias = lambda n: torch.zeros(n, device=dev, dtype=torch.bfloat16)
bq, bp, bf, bp2 = bias(3 * Cn), bias(Cn), bias(4 * Cn), bias(Cn)
lnw, lnb = torch.ones(Cn, device=dev, dtype=torch.bfloat16), bias(Cn)
x = torch.randn(N, Cn, device=dev).to(torch.bfloat16)
h1 = torch.empty_like(x); qkv = torch.empty(N, 3 * Cn, device=dev, dtype=torch.bfloat16); qc = torch.empty_like(x); att = torch.empty_like(x); x2 = torch.empty_like(x)
h2 = torch.empty_like(x); f = torch.empty(N, 4 * Cn, device=dev, dtype=torch.bfloat16); g = torch.empty_like(f); out = torch.empty_like(x)
def lin(w, xin, y, n, b, res=None):
    d = w.desc()
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), xin.data_ptr(), y.data_ptr(), b.data_ptr(), n, 1.0, 0.0, 1 if res is not None else 0, res.data_ptr() if res is not None else None), "lin")
def block():
    L.check(ctx.hip.kf_layernorm(ctx.h, x.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), h1.data_ptr(), N, Cn, 1e-5, None, None), "ln")
    lin(wqkv, h1, qkv, N, bq)
    qc.copy_(qkv[:, :Cn])
    for b in range(B):
        s = slice(b * T, (b + 1) * T)
        L.check(ctx.hip.kf_attn_prefill(ctx.h, qc[s].data_ptr(), qkv[s, Cn:].data_ptr(), qkv[s, 2 * Cn:].data_ptr(), att[s].data_ptr(), 0, T, Cn, H, H, hd, 3 * Cn), "attn")
    lin(wproj, att, x2, N, bp, x)
    L.check(ctx.hip.kf_layernorm(ctx.h, x2.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), h2.data_ptr(), N, Cn, 1e-5, None, None), "ln")
    lin(wfc, h2, f, N, bf)
    L.check(ctx.hip.kf_gelu(ctx.h, f.data_ptr(), g.data_ptr(), f.numel()), "gelu")
    lin(wproj2, g, out, N, bp2, x2)
for _ in range(2): block()
ctx.sync()
e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
for _ in range(5): block()
ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / 5
flops = 2.0 * N * (3 * Cn * Cn + Cn * Cn + 8 * Cn * Cn) + 4.0 * B * T * T * Cn / 2
print("GPT2-1558M block forward, %d tokens: %.2f ms  (%.0f TFLOP/s; x48 layers = %.0f ms => %.0f tok/s forward)" % (N, ms, flops / ms / 1e9, ms * 48, N / (ms * 48) * 1e3))
