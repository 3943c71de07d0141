"""BASELINE config 3 as one measured step: GPT2-1558M (n_embd 1600, 48 layers, 25 heads, ffn 6400, vocab 50257 padded to 50264), hybrid storage
(attention matrices f8e5m2, MLP matrices RTN 4-bit, tied bf16 wte), batch 8 x 1024 random ids: forward through all 48 blocks + final LayerNorm +
LM-head GEMM (in 8 chunks of 1024 rows like Head4Token, NeuronFuse.cu:895-925) + fused classifier, then one AdamW step over 1.558 G parameters
(bf16 moments).  One set of block weights per layer (48 x 30.7 M parameters in 4-bit / f8 = 1 GB).  Prints the time of each part."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
ctx = R.Context(0); dev = ctx.device
Cn, H, T, B, NL, V, Vp = 1600, 25, 1024, 8, 48, 50257, 50264
hd = Cn // H; N = B * T
mk = lambda m, k, t: ctx.quantize((torch.randn(m, k, device=dev) * 0.02).to(torch.bfloat16), t)
layers = [(mk(3 * Cn, Cn, L.F8E5M2), mk(Cn, Cn, L.F8E5M2), mk(4 * Cn, Cn, L.Q4), mk(Cn, 4 * Cn, L.Q4)) for _ in range(NL)]
bias = lambda n: torch.zeros(n, device=dev, dtype=torch.bfloat16)
bq, bp, bf, bp2 = bias(3 * Cn), bias(Cn), bias(4 * Cn), bias(Cn)
lnw, lnb = torch.ones(Cn, device=dev, dtype=torch.bfloat16), bias(Cn)
wte_t = torch.zeros(Vp, Cn, device=dev, dtype=torch.bfloat16); wte_t[:V] = (torch.randn(V, Cn, device=dev) * 0.02).to(torch.bfloat16)
wte = ctx.quantize(wte_t, L.BF16)
wpe = (torch.randn(T, Cn, device=dev) * 0.01).to(torch.bfloat16)
ids = torch.randint(0, V, (N,), device=dev, dtype=torch.int32); tgt = torch.randint(0, V, (N,), device=dev, dtype=torch.int32)
x = torch.empty(N, Cn, device=dev, dtype=torch.bfloat16); xb = torch.empty_like(x)
h1 = torch.empty_like(x); qkv = torch.empty(N, 3 * Cn, device=dev, dtype=torch.bfloat16); qc = torch.empty_like(x); att = torch.empty_like(x); x2 = torch.empty_like(x)
f = torch.empty(N, 4 * Cn, device=dev, dtype=torch.bfloat16); g = torch.empty_like(f)
logits = torch.empty(T, Vp, device=dev, dtype=torch.bfloat16); losses = torch.zeros(N, device=dev, dtype=torch.float32)
def lin(w, xin, y, n, b, res=None):
    d = w.desc()
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), xin.data_ptr(), y.data_ptr(), b.data_ptr() if b is not None else None, n, 1.0, 0.0, 1 if res is not None else 0, res.data_ptr() if res is not None else None), "lin")
def embed():
    x.copy_(wte_t[ids.long()]); x.add_(wpe.repeat(B, 1))   # token + position rows (torch: plumbing, not timed as a kernel of ours)
def block(l, xin, xout):
    wqkv, wproj, wfc, wproj2 = layers[l]
    L.check(ctx.hip.kf_layernorm(ctx.h, xin.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), h1.data_ptr(), N, Cn, 1e-5, None, None), "ln")
    lin(wqkv, h1, qkv, N, bq)
    qc.copy_(qkv[:, :Cn])
    for b in range(B):
        s = slice(b * T, (b + 1) * T)
        L.check(ctx.hip.kf_attn_prefill(ctx.h, qc[s].data_ptr(), qkv[s, Cn:].data_ptr(), qkv[s, 2 * Cn:].data_ptr(), att[s].data_ptr(), 0, T, Cn, H, H, hd, 3 * Cn), "attn")
    lin(wproj, att, x2, N, bp, xin)
    L.check(ctx.hip.kf_layernorm(ctx.h, x2.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), h1.data_ptr(), N, Cn, 1e-5, None, None), "ln")
    lin(wfc, h1, f, N, bf)
    L.check(ctx.hip.kf_gelu(ctx.h, f.data_ptr(), g.data_ptr(), f.numel()), "gelu")
    lin(wproj2, g, xout, N, bp2, x2)
def head_loss():
    L.check(ctx.hip.kf_layernorm(ctx.h, x.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), h1.data_ptr(), N, Cn, 1e-5, None, None), "lnf")
    for b in range(B):   # Head4Token: one batch row (T tokens) of logits at a time, reusing the buffer
        lin(wte, h1[b * T:(b + 1) * T], logits, T, None)
        L.check(ctx.hip.kf_fused_classifier(ctx.h, logits.data_ptr(), losses[b * T:].data_ptr(), None, 1.0 / N, tgt[b * T:].data_ptr(), 1, T, V, Vp, None, 1), "cls")
def forward():
    embed()
    a, b2 = x, xb
    for l in range(NL):
        block(l, a, b2); a, b2 = b2, a
    if a is not x: x.copy_(a)
    head_loss()
def timed(fn, reps=3):
    fn(); ctx.sync()
    e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for _ in range(reps): fn()
    ctx.record(e1); return ctx.elapsed_ms(e0, e1) / reps
t_blocks = timed(lambda: [block(l, x, xb) for l in range(NL)])
losses.zero_(); t_head = timed(head_loss)
losses.zero_(); forward(); ctx.sync()
print("mean loss %.4f (ln V = %.4f)" % (float(losses.mean()), float(torch.log(torch.tensor(float(V))))))
del layers
npar = 1_558_000_000 // 8 * 8
p = (torch.randn(npar, device=dev) * 0.02).to(torch.bfloat16); gr = (torch.randn(npar, device=dev) * 0.01).to(torch.bfloat16)
m1 = torch.zeros(npar, device=dev, dtype=torch.bfloat16); m2 = torch.zeros_like(m1)
t_adam = timed(lambda: L.check(ctx.hip.kf_adamw(ctx.h, p.data_ptr(), gr.data_ptr(), m1.data_ptr(), m2.data_ptr(), npar, L.BF16, 3e-4, 0.9, 0.95, 0.1, 0.05, 1e-8, 0.1, 1.0, 7, None), "adamw"))
flops = NL * (2.0 * N * 12 * Cn * Cn + 4.0 * B * T * T * Cn / 2) + 2.0 * N * Vp * Cn
tot = t_blocks + t_head + t_adam
print("GPT2-1558M, 8 x 1024 tokens: 48 blocks %.1f ms, final LN + head GEMM + loss %.1f ms, AdamW %.2f ms; forward %.1f ms = %.0f TFLOP/s, %.0f tok/s; fwd + Adam step %.1f ms"
      % (t_blocks, t_head, t_adam, t_blocks + t_head, flops / (t_blocks + t_head) / 1e9, N / (t_blocks + t_head) * 1e3, tot))
