"""One WHOLE training step of the hybrid-precision GPT-2 of BASELINE config 3, every operator through the C ABI -- the host-side order of the reference's step:

  forward  (TokenEmbed, 48 x [LayerNorm, SLP qkv, causal attention, SLP proj + residual, LayerNorm, SLP fc, GELU, SLP proj2 + residual], LayerNorm, tied head)
           on the QUANTISED blobs (attention matrices f8e5m2, MLP matrices 4-bit PackedQ): SLP::Forw, NeuronFuse.cu:305-381
  loss     fused classifier (cross entropy + logit gradient in place), NeuronFuse.cu / kf_loss.hip
  backward SLP::Back (NeuronFuse.cu:495-563), LayerNormal / GELU / attention / embedding backward -- weight gradients into PER-TENSOR buffers
  update   CU_adamw_ (Optimizer.cu:135-160: seeded stochastic rounding) on the model's OWN bf16 master weights and bf16 moments
  requant  CU_XtoQ128_ / Float2T<f8e5> (T.cu:105-175) of every updated matrix back into its blob, which the next forward reads

The step keeps every activation (no recomputation).  Python here only sequences ABI calls and owns the device buffers (torch tensors); embedding gather / add and the
zero fills are torch ops.  Used by tests/test_gpu_train_step.py (a 2-layer toy, two consecutive steps against the oracle) and by bench.py's config3 leg (full size)."""
import ctypes as C

import torch

from . import lib as L

MATS = ("qkv", "proj", "fc", "proj2")


class GPT2Step:
    def __init__(self, ctx, C_, H, NL, V, Vp, B, T, types=None, seed=0, w_std=0.02, masters=None):
        """masters: optional dict of host-provided bf16 torch tensors (tests hand the same numbers to the reference): 'wte' [Vp, C], 'wpe' [T, C], 'lnf' (w, b), 'blocks':
        list of dicts {mat: (W [out, in], b [out])} + 'ln' (w1, b1, w2, b2).  Otherwise N(0, w_std) draws on the device."""
        self.ctx, self.C, self.H, self.NL, self.V, self.Vp, self.B, self.T = ctx, C_, H, NL, V, Vp, B, T
        self.hd, self.N = C_ // H, B * T
        self.types = dict(qkv=L.F8E5M2, proj=L.F8E5M2, fc=L.Q4, proj2=L.Q4) if types is None else dict(types)
        dev, bf = ctx.device, torch.bfloat16
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        z = lambda *s, dt=bf: torch.zeros(*s, device=dev, dtype=dt)
        rnd = lambda *s, std=w_std: (torch.randn(*s, device=dev, generator=g) * std).to(bf)
        shapes = dict(qkv=(3 * C_, C_), proj=(C_, C_), fc=(4 * C_, C_), proj2=(C_, 4 * C_))
        self.params = []   # every trained tensor: dict(name, p (bf16 master), g (bf16 gradient), m, v (bf16 moments), wd (bool), blob (DevWeight or None), type)
        self.blocks = []

        def reg(name, p, wd, type_=None):
            e = dict(name=name, p=p.contiguous(), g=torch.zeros_like(p), m=torch.zeros_like(p), v=torch.zeros_like(p), wd=wd, blob=None, type=type_)
            if type_ is not None:
                e["blob"] = ctx.quantize(e["p"], type_)
                if type_ == L.BF16:
                    e["p"] = e["blob"].blob.view(bf).view(p.shape)   # a bf16 "blob" IS the master (the tied head): updated in place, nothing to re-quantise
            self.params.append(e)
            return e
        for l in range(NL):
            mb = masters["blocks"][l] if masters else None
            blk = {}
            for k in MATS:
                W = mb[k][0].to(dev) if mb else rnd(*shapes[k])
                b = mb[k][1].to(dev) if mb else z(shapes[k][0])
                blk[k] = reg("h%d.%s.w" % (l, k), W, True, self.types[k])
                blk[k + "_b"] = reg("h%d.%s.b" % (l, k), b, False)
            ln = [t.to(dev) for t in mb["ln"]] if mb else [torch.ones(C_, device=dev, dtype=bf), z(C_), torch.ones(C_, device=dev, dtype=bf), z(C_)]
            for i, nm in enumerate(("ln1.w", "ln1.b", "ln2.w", "ln2.b")):
                blk[nm] = reg("h%d.%s" % (l, nm), ln[i], False)
            self.blocks.append(blk)
        wte = masters["wte"].to(dev) if masters else torch.cat([rnd(V, C_), z(Vp - V, C_)])
        self.wte = reg("wte", wte, True, L.BF16)
        self.wpe = reg("wpe", masters["wpe"].to(dev) if masters else rnd(T, C_, std=w_std / 2), False)
        lnf = [t.to(dev) for t in masters["lnf"]] if masters else [torch.ones(C_, device=dev, dtype=bf), z(C_)]
        self.lnf_w, self.lnf_b = reg("lnf.w", lnf[0], False), reg("lnf.b", lnf[1], False)
        # activations of one step, all kept
        N = self.N
        self.A = [dict(x=z(N, C_), h1=z(N, C_), m1=z(N, dt=torch.float32), r1=z(N, dt=torch.float32), qkv=z(N, 3 * C_), att=z(N, C_), x2=z(N, C_), h2=z(N, C_),
                       m2=z(N, dt=torch.float32), r2=z(N, dt=torch.float32), f=z(N, 4 * C_), g=z(N, 4 * C_)) for _ in range(NL)]
        self.xf, self.hf, self.mf, self.rf = z(N, C_), z(N, C_), z(N, dt=torch.float32), z(N, dt=torch.float32)
        self.qc, self.logits, self.losses = z(N, C_), z(N, Vp), z(N, dt=torch.float32)
        self.dx, self.dh, self.dqkv, self.datt, self.d4 = z(N, C_), z(N, C_), z(N, 3 * C_), z(N, C_), z(N, 4 * C_)
        hip = ctx.hip
        for k in MATS:
            ctx.linear_scratch(self.blocks[0][k]["blob"], N)
        nb = max(hip.kf_linear_backward_scratch_bytes(oc, ic, N) for oc, ic in list(shapes.values()) + [(Vp, C_)])
        self._sc_lin = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
        self._sp_lin = (self._sc_lin.data_ptr() + 255) & ~255
        self._sc_ln = torch.empty(hip.kf_norm_backward_scratch_bytes(N, C_, 1) // 8 + 1, dtype=torch.float64, device=dev)
        self._sc_at = torch.empty(hip.kf_attn_backward_scratch_bytes(T, H, B) // 4 + 1, dtype=torch.float32, device=dev)
        self.t = 0   # optimizer steps taken

    # ---- operators
    def _lin(self, e, xin, y, bias, res=None):
        d = e["blob"].desc()
        L.check(self.ctx.hip.kf_linear(self.ctx.h, C.byref(d), xin.data_ptr(), y.data_ptr(), bias["p"].data_ptr() if bias is not None else None, self.N, 1.0, 0.0,
                                       1 if res is not None else 0, res.data_ptr() if res is not None else None), "kf_linear")

    def _ln(self, x, w, b, y, m_, r_):
        L.check(self.ctx.hip.kf_layernorm(self.ctx.h, x.data_ptr(), w["p"].data_ptr(), b["p"].data_ptr(), y.data_ptr(), self.N, self.C, 1e-5, m_.data_ptr(), r_.data_ptr()), "kf_layernorm")

    def forward(self, ids, tgt):
        """ids, tgt: int32 [B * T] on the device.  Leaves the per-row losses in self.losses and the logit gradients (of the MEAN loss) in self.logits."""
        ctx, hip, C_, NL, B, T = self.ctx, self.ctx.hip, self.C, self.NL, self.B, self.T
        A = self.A
        A[0]["x"].copy_(self.wte["p"][ids.long()])
        A[0]["x"].add_(self.wpe["p"].repeat(B, 1))
        for l in range(NL):
            a, b = A[l], self.blocks[l]
            self._ln(a["x"], b["ln1.w"], b["ln1.b"], a["h1"], a["m1"], a["r1"])
            self._lin(b["qkv"], a["h1"], a["qkv"], b["qkv_b"])
            self.qc.copy_(a["qkv"][:, :C_])
            L.check(hip.kf_attn_prefill_batch(ctx.h, self.qc.data_ptr(), a["qkv"][:, C_:].data_ptr(), a["qkv"][:, 2 * C_:].data_ptr(), a["att"].data_ptr(), T, C_, self.H, self.H, self.hd,
                                              3 * C_, B), "kf_attn_prefill_batch")
            self._lin(b["proj"], a["att"], a["x2"], b["proj_b"], a["x"])
            self._ln(a["x2"], b["ln2.w"], b["ln2.b"], a["h2"], a["m2"], a["r2"])
            self._lin(b["fc"], a["h2"], a["f"], b["fc_b"])
            L.check(hip.kf_gelu(ctx.h, a["f"].data_ptr(), a["g"].data_ptr(), a["f"].numel()), "kf_gelu")
            self._lin(b["proj2"], a["g"], A[l + 1]["x"] if l + 1 < NL else self.xf, b["proj2_b"], a["x2"])
        self._ln(self.xf, self.lnf_w, self.lnf_b, self.hf, self.mf, self.rf)
        self._lin(self.wte, self.hf, self.logits, None)
        self.losses.zero_()
        L.check(hip.kf_fused_classifier(ctx.h, self.logits.data_ptr(), self.losses.data_ptr(), None, 1.0 / self.N, tgt.data_ptr(), B, T, self.V, self.Vp, None, 1), "kf_fused_classifier")
        self._ids = ids

    def _lin_bwd(self, e, dIn, inp, delta, bias):
        d = e["blob"].desc()
        L.check(self.ctx.hip.kf_linear_backward(self.ctx.h, C.byref(d), dIn.data_ptr(), inp.data_ptr(), delta.data_ptr(), e["g"].data_ptr(), bias["g"].data_ptr() if bias is not None else None,
                                                self.N, 0, self._sp_lin), "kf_linear_backward")

    def _ln_bwd(self, dxx, dout, inp, w, b, m_, r_):
        # kf_norm_backward ADDS into dweight / dbias (one shared LayerNorm in the operator tests): the per-tensor gradients are zero here (kf_adamw zeroes what it consumed)
        L.check(self.ctx.hip.kf_norm_backward(self.ctx.h, dxx.data_ptr(), w["g"].data_ptr(), b["g"].data_ptr(), dout.data_ptr(), inp.data_ptr(), w["p"].data_ptr(), m_.data_ptr(), r_.data_ptr(),
                                              self.N, self.C, self._sc_ln.data_ptr()), "kf_norm_backward")

    def backward(self):
        ctx, hip, C_, NL, B, T = self.ctx, self.ctx.hip, self.C, self.NL, self.B, self.T
        A, dx, dh, dqkv, datt, d4 = self.A, self.dx, self.dh, self.dqkv, self.datt, self.d4
        self.logits[:, self.V:].zero_()
        self._lin_bwd(self.wte, self.logits, self.hf, dh, None)
        dx.zero_()
        self._ln_bwd(dx, dh, self.xf, self.lnf_w, self.lnf_b, self.mf, self.rf)
        for l in reversed(range(NL)):
            a, b = A[l], self.blocks[l]
            self._lin_bwd(b["proj2"], dx, a["g"], d4, b["proj2_b"])
            L.check(hip.kf_gelu_backward(ctx.h, d4.data_ptr(), a["f"].data_ptr(), d4.numel()), "kf_gelu_backward")
            self._lin_bwd(b["fc"], d4, a["h2"], dh, b["fc_b"])
            self._ln_bwd(dx, dh, a["x2"], b["ln2.w"], b["ln2.b"], a["m2"], a["r2"])
            self._lin_bwd(b["proj"], dx, a["att"], datt, b["proj_b"])
            L.check(hip.kf_attn_backward(ctx.h, a["qkv"][:, :C_].data_ptr(), a["qkv"][:, C_:2 * C_].data_ptr(), a["qkv"][:, 2 * C_:].data_ptr(), 3 * C_, a["att"].data_ptr(), datt.data_ptr(), C_,
                                         dqkv[:, :C_].data_ptr(), dqkv[:, C_:2 * C_].data_ptr(), dqkv[:, 2 * C_:].data_ptr(), 3 * C_, T, self.H, self.H, self.hd, B, self._sc_at.data_ptr()),
                    "kf_attn_backward")
            self._lin_bwd(b["qkv"], dqkv, a["h1"], dh, b["qkv_b"])
            self._ln_bwd(dx, dh, a["x"], b["ln1.w"], b["ln1.b"], a["m1"], a["r1"])
        L.check(hip.kf_embed_backward(ctx.h, self.wte["g"].data_ptr(), C_, self.wpe["g"].data_ptr(), dx.data_ptr(), self._ids.data_ptr(), B, T, C_, self.Vp), "kf_embed_backward")

    def update(self, lr=3e-4, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.1, seed=1234):
        """AdamW on every tensor (its own master, moments and gradient; seeded stochastic rounding: seed + the tensor's index, as one seed per launch in the reference), then
        the re-quantisation of every quantised matrix from its updated master.  kf_adamw zeroes the gradients it has consumed."""
        ctx, hip = self.ctx, self.ctx.hip
        self.t += 1
        b1c, b2c = 1.0 - beta1 ** self.t, 1.0 - beta2 ** self.t
        for i, e in enumerate(self.params):
            n = e["p"].numel()
            assert n % 8 == 0
            L.check(hip.kf_adamw(ctx.h, e["p"].data_ptr(), e["g"].data_ptr(), e["m"].data_ptr(), e["v"].data_ptr(), n, L.BF16, lr, beta1, beta2, b1c, b2c, eps, wd if e["wd"] else 0.0, 1.0,
                                 (seed + 7919 * self.t + i) & 0xFFFFFFFF, None), "kf_adamw")
            if e["type"] is not None and e["type"] != L.BF16:
                d = e["blob"].desc()
                L.check(hip.kf_quantize(ctx.h, C.byref(d), e["p"].data_ptr(), 0), "kf_quantize")

    def step(self, ids, tgt, **hp):
        self.forward(ids, tgt)
        self.backward()
        self.update(**hp)

    def n_params(self):
        return sum(e["p"].numel() for e in self.params)
