"""A whole training step of Qwen3-0.6B on one MI355X through the ABI: dim 1024, 28 layers, 16 / 8 heads x 128, ffn 3072, vocabulary 151936 (tied bf16
table), 4-bit PackedQ layer weights, batch 8 x 1024 random ids.  Forward with every activation kept, loss, backward through every operator
(RMSNorm, Q / K / V, q/k-norm + RoPE, GQA attention, o_proj, gate / up / SwiGLU / down, head, embedding), AdamW over 0.6 G parameters."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R
ctx = R.Context(0); dev = ctx.device
dim, H, KV, hd, ffn, NL, V, T, B, theta, eps = 1024, 16, 8, 128, 3072, int(os.environ.get("NL", "28")), 151936, 1024, 8, 1e6, 1e-6
N, Cq, Ck = B * T, H * hd, KV * hd
W_ = Cq + 2 * Ck
bf = torch.bfloat16
z = lambda *s, dt=bf: torch.zeros(*s, device=dev, dtype=dt)
mk = lambda m, k: ctx.quantize((torch.randn(m, k, device=dev) * 0.02).to(bf), L.Q4)
shapes = {"q": (Cq, dim), "k": (Ck, dim), "v": (Ck, dim), "o": (dim, Cq), "gate": (ffn, dim), "up": (ffn, dim), "down": (dim, ffn)}
layers = [{k: mk(*s) for k, s in shapes.items()} for _ in range(NL)]
n1 = torch.ones(dim, device=dev, dtype=bf); nh = torch.ones(hd, device=dev, dtype=bf)
wte_t = (torch.randn(V, dim, device=dev) * 0.02).to(bf); wte = ctx.quantize(wte_t, L.BF16)
ids = torch.randint(0, V, (N,), device=dev, dtype=torch.int32); tgt = torch.randint(0, V, (N,), device=dev, dtype=torch.int32)
table = ctx.rope_table(T, hd, theta)
f32 = lambda n: z(n, dt=torch.float32)
A = [dict(x=z(N, dim), h1=z(N, dim), r1=f32(N), raw=z(N, W_), rq=f32(N * H), rk=f32(N * KV), qkv=z(N, W_), att=z(N, Cq), x2=z(N, dim), h2=z(N, dim), r2=f32(N), gate=z(N, ffn), up=z(N, ffn),
          act=z(N, ffn)) for _ in range(NL)]
xf, hf, rf = z(N, dim), z(N, dim), f32(N)
qc, tq, tk = z(N, Cq), z(N * H, hd), z(N * KV, hd)
logits = z(N, V); losses = f32(N)
for _w in layers[0].values(): ctx.linear_scratch(_w, N)   # kernels never allocate: the dequantise-then-multiply workspace of training-size batches (kf_linear_scratch_bytes)
def lin(w, xin, y, res=None):
    d = w.desc()
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), xin.data_ptr(), y.data_ptr(), None, N, 1.0, 0.0, 1 if res is not None else 0, res.data_ptr() if res is not None else None), "lin")
def rms(x, w, y, r, rows, d_): L.check(ctx.hip.kf_rmsnorm(ctx.h, x.data_ptr(), w.data_ptr(), y.data_ptr(), rows, d_, eps, r.data_ptr()), "rms")
tmp_q, tmp_k, tmp_v = z(N, Cq), z(N, Ck), z(N, Ck)
def forward():
    A[0]["x"].copy_(wte_t[ids.long()])
    for l in range(NL):
        a, w = A[l], layers[l]
        rms(a["x"], n1, a["h1"], a["r1"], N, dim)
        lin(w["q"], a["h1"], tmp_q); lin(w["k"], a["h1"], tmp_k); lin(w["v"], a["h1"], tmp_v)
        a["qkv"][:, :Cq] = tmp_q; a["qkv"][:, Cq:Cq + Ck] = tmp_k; a["qkv"][:, Cq + Ck:] = tmp_v
        a["raw"].copy_(a["qkv"])
        L.check(ctx.hip.kf_qknorm_rope_train(ctx.h, a["qkv"][:, :Cq].data_ptr(), a["qkv"][:, Cq:].data_ptr(), nh.data_ptr(), nh.data_ptr(), table.data_ptr(), N, T, W_, W_, H, KV, hd, eps,
                                             a["rq"].data_ptr(), a["rk"].data_ptr()), "rope")
        qc.copy_(a["qkv"][:, :Cq])
        L.check(ctx.hip.kf_attn_prefill_batch(ctx.h, qc.data_ptr(), a["qkv"][:, Cq:].data_ptr(), a["qkv"][:, Cq + Ck:].data_ptr(), a["att"].data_ptr(), T, Cq, H, KV, hd, W_, B), "attn")
        lin(w["o"], a["att"], a["x2"], a["x"])
        rms(a["x2"], n1, a["h2"], a["r2"], N, dim)
        lin(w["gate"], a["h2"], a["gate"]); lin(w["up"], a["h2"], a["up"])
        L.check(ctx.hip.kf_swiglu(ctx.h, a["gate"].data_ptr(), a["up"].data_ptr(), a["act"].data_ptr(), N * ffn), "swiglu")
        lin(w["down"], a["act"], A[l + 1]["x"] if l + 1 < NL else xf, a["x2"])
    rms(xf, n1, hf, rf, N, dim)
    lin(wte, hf, logits)
    losses.zero_()
    L.check(ctx.hip.kf_fused_classifier(ctx.h, logits.data_ptr(), losses.data_ptr(), None, 1.0 / N, tgt.data_ptr(), B, T, V, V, None, 1), "cls")
dx, dh, dact, dgate, datt, dqkv = z(N, dim), z(N, dim), z(N, ffn), z(N, ffn), z(N, Cq), z(N, W_)
dq_post, dk_post, dq_raw, dk_raw, dvc = z(N * H, hd), z(N * KV, hd), z(N * H, hd), z(N * KV, hd), z(N, Ck)
gW = {k: z(*s) for k, s in shapes.items()}; g_n, g_h, g_wte = z(dim), z(hd), z(V, dim)
sc_lin = torch.empty(max(ctx.hip.kf_linear_backward_scratch_bytes(oc, ic, N) for oc, ic in list(shapes.values()) + [(V, dim)]) + 256, dtype=torch.uint8, device=dev)
sp_lin = (sc_lin.data_ptr() + 255) & ~255
sc_n = torch.empty(ctx.hip.kf_norm_backward_scratch_bytes(N * H, hd, 0) // 8 + ctx.hip.kf_norm_backward_scratch_bytes(N, dim, 0) // 8 + 2, dtype=torch.float64, device=dev)
sc_at = torch.empty(ctx.hip.kf_attn_backward_scratch_bytes(T, H, B) // 4 + 1, dtype=torch.float32, device=dev)
def lin_bwd(w, dIn, inp, delta, gw, acc=0):
    d = w.desc()
    L.check(ctx.hip.kf_linear_backward(ctx.h, C.byref(d), dIn.data_ptr(), inp.data_ptr(), delta.data_ptr(), gw.data_ptr(), None, N, acc, sp_lin), "lin_bwd")
def rms_bwd(dxx, dout, inp, w, r, gw, rows, d_): L.check(ctx.hip.kf_norm_backward(ctx.h, dxx.data_ptr(), gw.data_ptr(), None, dout.data_ptr(), inp.data_ptr(), w.data_ptr(), None, r.data_ptr(), rows, d_, sc_n.data_ptr()), "rms_bwd")
def backward():
    lin_bwd(wte, logits, hf, dh, g_wte)
    dx.zero_(); rms_bwd(dx, dh, xf, n1, rf, g_n, N, dim)
    for l in reversed(range(NL)):
        a, w = A[l], layers[l]
        lin_bwd(w["down"], dx, a["act"], dact, gW["down"])
        L.check(ctx.hip.kf_swiglu_backward(ctx.h, dact.data_ptr(), dgate.data_ptr(), a["gate"].data_ptr(), a["up"].data_ptr(), N * ffn), "swiglu_bwd")
        lin_bwd(w["up"], dact, a["h2"], dh, gW["up"]); lin_bwd(w["gate"], dgate, a["h2"], dh, gW["gate"], 1)
        rms_bwd(dx, dh, a["x2"], n1, a["r2"], g_n, N, dim)
        lin_bwd(w["o"], dx, a["att"], datt, gW["o"])
        q = a["qkv"]
        L.check(ctx.hip.kf_attn_backward(ctx.h, q[:, :Cq].data_ptr(), q[:, Cq:].data_ptr(), q[:, Cq + Ck:].data_ptr(), W_, a["att"].data_ptr(), datt.data_ptr(), Cq,
                                         dqkv[:, :Cq].data_ptr(), dqkv[:, Cq:].data_ptr(), dqkv[:, Cq + Ck:].data_ptr(), W_, T, H, KV, hd, B, sc_at.data_ptr()), "attn_bwd")
        L.check(ctx.hip.kf_rope_backward(ctx.h, dqkv[:, :Cq].data_ptr(), table.data_ptr(), 0, N, T, W_, H, hd), "rope_bwd")
        L.check(ctx.hip.kf_rope_backward(ctx.h, dqkv[:, Cq:].data_ptr(), table.data_ptr(), 0, N, T, W_, KV, hd), "rope_bwd")
        dq_post.copy_(dqkv[:, :Cq].reshape(N * H, hd)); dk_post.copy_(dqkv[:, Cq:Cq + Ck].reshape(N * KV, hd)); dvc.copy_(dqkv[:, Cq + Ck:])
        tmp_q.copy_(a["raw"][:, :Cq]); tmp_k.copy_(a["raw"][:, Cq:Cq + Ck])
        dq_raw.zero_(); dk_raw.zero_()
        rms_bwd(dq_raw, dq_post, tmp_q.view(N * H, hd), nh, a["rq"], g_h, N * H, hd); rms_bwd(dk_raw, dk_post, tmp_k.view(N * KV, hd), nh, a["rk"], g_h, N * KV, hd)
        lin_bwd(w["q"], dq_raw.view(N, Cq), a["h1"], dh, gW["q"]); lin_bwd(w["k"], dk_raw.view(N, Ck), a["h1"], dh, gW["k"], 1); lin_bwd(w["v"], dvc, a["h1"], dh, gW["v"], 1)
        rms_bwd(dx, dh, a["x"], n1, a["r1"], g_n, N, dim)
    L.check(ctx.hip.kf_embed_backward(ctx.h, g_wte.data_ptr(), dim, None, dx.data_ptr(), ids.data_ptr(), B, T, dim, V), "embed_bwd")
def timed(fn, reps=2):
    fn(); ctx.sync()
    e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for _ in range(reps): fn()
    ctx.record(e1); return ctx.elapsed_ms(e0, e1) / reps
t_f = timed(forward); forward(); ctx.sync(); print("mean loss %.4f" % float(losses.mean()))
t_b = timed(backward)
del A, logits, g_wte
npar = (NL * sum(a * b for a, b in shapes.values()) + V * dim) // 8 * 8
p = (torch.randn(npar, device=dev) * 0.02).to(bf); gr = (torch.randn(npar, device=dev) * 0.01).to(bf); m1 = z(npar); m2 = z(npar)
t_a = timed(lambda: L.check(ctx.hip.kf_adamw(ctx.h, p.data_ptr(), gr.data_ptr(), m1.data_ptr(), m2.data_ptr(), npar, L.BF16, 3e-4, 0.9, 0.95, 0.1, 0.05, 1e-8, 0.1, 1.0, 7, None), "adamw"))
print("Qwen3-0.6B (%d layers), 8 x 1024 tokens: forward + loss %.1f ms, backward %.1f ms, AdamW %.2f ms => training step %.1f ms = %.0f tok/s" % (NL, t_f, t_b, t_a, t_f + t_b + t_a, N / (t_f + t_b + t_a) * 1e3))
