"""BASELINE config 3 as one whole training step on one MI355X: GPT2-1558M (n_embd 1600, 48 layers, 25 heads, ffn 6400, vocab 50257 padded to 50304),
hybrid storage (attention matrices f8e5m2, MLP matrices RTN 4-bit, tied bf16 wte), batch 8 x 1024 random ids.  Forward with every activation kept,
loss, backward through every operator of the ABI (gradients of the quantised layers = bf16 gradients of their dequantised weights), AdamW over
1.558 G parameters.  Prints the time of each part.  (Gradient buffers of the 48 blocks are shared between layers: the optimiser part is timed on
its own full-size vector, as in scratch/ub_adamw.py.)"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
ctx = R.Context(0); dev = ctx.device
Cn, H, T, B, NL, V, Vp = 1600, 25, 1024, 8, int(os.environ.get("NL", "48")), 50257, 50304
hd = Cn // H; N = B * T
bf = torch.bfloat16
mk = lambda m, k, t: ctx.quantize((torch.randn(m, k, device=dev) * 0.02).to(bf), t)
layers = [(mk(3 * Cn, Cn, L.F8E5M2), mk(Cn, Cn, L.F8E5M2), mk(4 * Cn, Cn, L.Q4), mk(Cn, 4 * Cn, L.Q4)) for _ in range(NL)]
z = lambda *s, dt=bf: torch.zeros(*s, device=dev, dtype=dt)
bq, bp, bfc, bp2 = z(3 * Cn), z(Cn), z(4 * Cn), z(Cn)
lnw, lnb = torch.ones(Cn, device=dev, dtype=bf), z(Cn)
wte_t = z(Vp, Cn); wte_t[:V] = (torch.randn(V, Cn, device=dev) * 0.02).to(bf)
wte = ctx.quantize(wte_t, L.BF16)
wpe = (torch.randn(T, Cn, device=dev) * 0.01).to(bf)
ids = torch.randint(0, V, (N,), device=dev, dtype=torch.int32); tgt = torch.randint(0, V, (N,), device=dev, dtype=torch.int32)
# activations kept for the backward pass
A = [dict(x=z(N, Cn), h1=z(N, Cn), m1=z(N, dt=torch.float32), r1=z(N, dt=torch.float32), qkv=z(N, 3 * Cn), att=z(N, Cn), x2=z(N, Cn), h2=z(N, Cn), m2=z(N, dt=torch.float32),
          r2=z(N, dt=torch.float32), f=z(N, 4 * Cn), g=z(N, 4 * Cn)) for _ in range(NL)]
xf, hf, mf, rf = z(N, Cn), z(N, Cn), z(N, dt=torch.float32), z(N, dt=torch.float32)
qc = z(N, Cn); logits = z(N, Vp); losses = z(N, dt=torch.float32)
for _ws in layers[0]: ctx.linear_scratch(_ws, N)   # kernels never allocate: the dequantise-then-multiply workspace of training-size batches (kf_linear_scratch_bytes)
def lin(w, xin, y, n, b, res=None):
    d = w.desc()
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), xin.data_ptr(), y.data_ptr(), b.data_ptr() if b is not None else None, n, 1.0, 0.0, 1 if res is not None else 0, res.data_ptr() if res is not None else None), "lin")
def ln(x, y, m, r): L.check(ctx.hip.kf_layernorm(ctx.h, x.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), y.data_ptr(), N, Cn, 1e-5, m.data_ptr(), r.data_ptr()), "ln")
def forward():
    A[0]["x"].copy_(wte_t[ids.long()]); A[0]["x"].add_(wpe.repeat(B, 1))
    for l in range(NL):
        a = A[l]; wqkv, wproj, wfc, wproj2 = layers[l]
        ln(a["x"], a["h1"], a["m1"], a["r1"])
        lin(wqkv, a["h1"], a["qkv"], N, bq)
        qc.copy_(a["qkv"][:, :Cn])
        L.check(ctx.hip.kf_attn_prefill_batch(ctx.h, qc.data_ptr(), a["qkv"][:, Cn:].data_ptr(), a["qkv"][:, 2 * Cn:].data_ptr(), a["att"].data_ptr(), T, Cn, H, H, hd, 3 * Cn, B), "attn")
        lin(wproj, a["att"], a["x2"], N, bp, a["x"])
        ln(a["x2"], a["h2"], a["m2"], a["r2"])
        lin(wfc, a["h2"], a["f"], N, bfc)
        L.check(ctx.hip.kf_gelu(ctx.h, a["f"].data_ptr(), a["g"].data_ptr(), a["f"].numel()), "gelu")
        lin(wproj2, a["g"], A[l + 1]["x"] if l + 1 < NL else xf, N, bp2, a["x2"])
    ln(xf, hf, mf, rf)
    lin(wte, hf, logits, N, None)
    losses.zero_()
    L.check(ctx.hip.kf_fused_classifier(ctx.h, logits.data_ptr(), losses.data_ptr(), None, 1.0 / N, tgt.data_ptr(), B, T, V, Vp, None, 1), "cls")
# gradient buffers (shared between the layers) and scratch
dx, dh, dqkv, datt, d4 = z(N, Cn), z(N, Cn), z(N, 3 * Cn), z(N, Cn), z(N, 4 * Cn)
gW = {"qkv": z(3 * Cn, Cn), "proj": z(Cn, Cn), "fc": z(4 * Cn, Cn), "proj2": z(Cn, 4 * Cn)}
gB = {"qkv": z(3 * Cn), "proj": z(Cn), "fc": z(4 * Cn), "proj2": z(Cn)}
g_lnw, g_lnb, g_wte, g_wpe = z(Cn), z(Cn), z(Vp, Cn), z(T, Cn)
sc_lin = torch.empty(max(ctx.hip.kf_linear_backward_scratch_bytes(oc, ic, N) for oc, ic in ((3 * Cn, Cn), (Cn, Cn), (4 * Cn, Cn), (Cn, 4 * Cn), (Vp, Cn))) + 256, dtype=torch.uint8, device=dev)
sp_lin = (sc_lin.data_ptr() + 255) & ~255
sc_ln = torch.empty(ctx.hip.kf_norm_backward_scratch_bytes(N, Cn, 1) // 8 + 1, dtype=torch.float64, device=dev)
sc_at = torch.empty(ctx.hip.kf_attn_backward_scratch_bytes(T, H, B) // 4 + 1, dtype=torch.float32, device=dev)
def lin_bwd(w, dIn, inp, delta, gw, gb, acc=0):
    d = w.desc()
    L.check(ctx.hip.kf_linear_backward(ctx.h, C.byref(d), dIn.data_ptr(), inp.data_ptr(), delta.data_ptr(), gw.data_ptr(), gb.data_ptr() if gb is not None else None, N, acc, sp_lin), "lin_bwd")
def ln_bwd(dxx, dout, inp, m, r): L.check(ctx.hip.kf_norm_backward(ctx.h, dxx.data_ptr(), g_lnw.data_ptr(), g_lnb.data_ptr(), dout.data_ptr(), inp.data_ptr(), lnw.data_ptr(), m.data_ptr(), r.data_ptr(), N, Cn, sc_ln.data_ptr()), "ln_bwd")
T_ = {}
def backward():
    logits[:, V:].zero_()
    lin_bwd(wte, logits, hf, dh, g_wte, None)
    dx.zero_()
    ln_bwd(dx, dh, xf, mf, rf)
    for l in reversed(range(NL)):
        a = A[l]; wqkv, wproj, wfc, wproj2 = layers[l]
        lin_bwd(wproj2, dx, a["g"], d4, gW["proj2"], gB["proj2"])
        L.check(ctx.hip.kf_gelu_backward(ctx.h, d4.data_ptr(), a["f"].data_ptr(), d4.numel()), "gelu_bwd")
        lin_bwd(wfc, d4, a["h2"], dh, gW["fc"], gB["fc"])
        ln_bwd(dx, dh, a["x2"], a["m2"], a["r2"])
        lin_bwd(wproj, dx, a["att"], datt, gW["proj"], gB["proj"])
        L.check(ctx.hip.kf_attn_backward(ctx.h, a["qkv"][:, :Cn].data_ptr(), a["qkv"][:, Cn:2 * Cn].data_ptr(), a["qkv"][:, 2 * Cn:].data_ptr(), 3 * Cn, a["att"].data_ptr(), datt.data_ptr(), Cn,
                                         dqkv[:, :Cn].data_ptr(), dqkv[:, Cn:2 * Cn].data_ptr(), dqkv[:, 2 * Cn:].data_ptr(), 3 * Cn, T, H, H, hd, B, sc_at.data_ptr()), "attn_bwd")
        lin_bwd(wqkv, dqkv, a["h1"], dh, gW["qkv"], gB["qkv"])
        ln_bwd(dx, dh, a["x"], a["m1"], a["r1"])
    L.check(ctx.hip.kf_embed_backward(ctx.h, g_wte.data_ptr(), Cn, g_wpe.data_ptr(), dx.data_ptr(), ids.data_ptr(), B, T, Cn, Vp), "embed_bwd")
def timed(fn, reps=2):
    fn(); ctx.sync()
    e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for _ in range(reps): fn()
    ctx.record(e1); return ctx.elapsed_ms(e0, e1) / reps
t_f = timed(forward)
forward(); ctx.sync(); print("mean loss %.4f" % float(losses.mean()))
t_b = timed(backward)
del A, layers, logits
npar = 1_558_000_000 // 8 * 8
p = (torch.randn(npar, device=dev) * 0.02).to(bf); gr = (torch.randn(npar, device=dev) * 0.01).to(bf); m1 = z(npar); m2 = z(npar)
t_a = timed(lambda: L.check(ctx.hip.kf_adamw(ctx.h, p.data_ptr(), gr.data_ptr(), m1.data_ptr(), m2.data_ptr(), npar, L.BF16, 3e-4, 0.9, 0.95, 0.1, 0.05, 1e-8, 0.1, 1.0, 7, None), "adamw"))
print("GPT2-1558M (%d layers), 8 x 1024 tokens: forward + loss %.1f ms, backward %.1f ms, AdamW %.2f ms => training step %.1f ms = %.0f tok/s" % (NL, t_f, t_b, t_a, t_f + t_b + t_a, N / (t_f + t_b + t_a) * 1e3))
