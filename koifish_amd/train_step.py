"""One WHOLE training step of the hybrid-precision GPT-2 of BASELINE config 3, every operator through the C ABI -- the host-side order of the reference's step:

  forward  (TokenEmbed, 48 x [LayerNorm, SLP qkv, causal attention, SLP proj + residual, LayerNorm, SLP fc, GELU, SLP proj2 + residual], LayerNorm, tied head)
           on the QUANTISED blobs (attention matrices f8e5m2, MLP matrices 4-bit PackedQ): SLP::Forw, NeuronFuse.cu:305-381
  loss     fused classifier (cross entropy + logit gradient in place), NeuronFuse.cu / kf_loss.hip
  backward SLP::Back (NeuronFuse.cu:495-563), LayerNormal / GELU / attention / embedding backward -- weight gradients into PER-TENSOR buffers
  update   CU_adamw_ (Optimizer.cu:135-160: seeded stochastic rounding) on the model's OWN bf16 master weights and bf16 moments
  requant  CU_XtoQ128_ / Float2T<f8e5> (T.cu:105-175) of every updated matrix back into its blob, which the next forward reads

The step keeps every activation (no recomputation).  Python here OWNS the device buffers (torch tensors: setup, outside any timed region) and registers them with
koifish::GPT2Trainer (koifish_amd/host/kf_train.cpp, libkf_host.so), which sequences the step: forward / backward / update are one C call each, step() is ONE call,
and nothing in them is a torch op (the embedding gather + add is kf_embed_pos, the attention reads q out of the fused rows, the zero fills are kf_memset / kf_memset2d).
Used by tests/test_gpu_train_step.py (a 2-layer toy, two consecutive steps against the oracle) and by bench.py's config3 leg (full size)."""
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

        # the step's sequencer: koifish::GPT2Trainer of libkf_host.so (koifish_amd/host/kf_train.cpp) over the buffers above -- every tensor registered once
        host = ctx.host
        self.h = host.kfh_gpt2_create(ctx.h, C_, H, NL, V, Vp, B, T)
        if not self.h:
            raise RuntimeError("kfh_gpt2_create refused the shape")
        assert host.kfh_gpt2_n_params(self.h) == len(self.params)
        for i, e in enumerate(self.params):
            d = e["blob"].desc() if e["blob"] is not None else None
            L.check(host.kfh_gpt2_set_param(self.h, i, e["p"].data_ptr(), e["g"].data_ptr(), e["m"].data_ptr(), e["v"].data_ptr(), e["p"].numel(), int(e["wd"]),
                                            C.byref(d) if d is not None else None, int(e["type"] is not None and e["type"] != L.BF16)), "kfh_gpt2_set_param")
        for l, a in enumerate(self.A):
            arr = (C.c_void_p * 12)(*[a[k].data_ptr() for k in ("x", "h1", "m1", "r1", "qkv", "att", "x2", "h2", "m2", "r2", "f", "g")])
            L.check(host.kfh_gpt2_set_block_acts(self.h, l, arr), "kfh_gpt2_set_block_acts")
        arr = (C.c_void_p * 14)(*([t_.data_ptr() for t_ in (self.xf, self.hf, self.mf, self.rf, self.logits, self.losses, self.dx, self.dh, self.dqkv, self.datt, self.d4)]
                                  + [self._sp_lin, self._sc_ln.data_ptr(), self._sc_at.data_ptr()]))
        L.check(host.kfh_gpt2_set_buffers(self.h, arr), "kfh_gpt2_set_buffers")

    def close(self):
        if getattr(self, "h", None):
            self.ctx.host.kfh_gpt2_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def t(self):
        """optimizer steps taken"""
        return int(self.ctx.host.kfh_gpt2_steps_taken(self.h))

    # ---- the step: sequenced by the host library; Python passes two device pointers and the hyper-parameters
    def forward(self, ids, tgt):
        """ids, tgt: int32 [B * T] on the device.  Leaves the per-row losses in self.losses and the logit gradients (of the MEAN loss) in self.logits."""
        self._ids = ids   # kept alive: the backward reads them
        L.check(self.ctx.host.kfh_gpt2_forward(self.h, ids.data_ptr(), tgt.data_ptr()), "kfh_gpt2_forward")

    def backward(self):
        L.check(self.ctx.host.kfh_gpt2_backward(self.h), "kfh_gpt2_backward")

    def update(self, lr=3e-4, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.1, seed=1234):
        """AdamW on every tensor (its own master, moments and gradient; seeded stochastic rounding: seed + 7919 t + the tensor's index, as one seed per launch in the
        reference), then the re-quantisation of every quantised matrix from its updated master.  kf_adamw zeroes the gradients it has consumed."""
        L.check(self.ctx.host.kfh_gpt2_update(self.h, lr, beta1, beta2, eps, wd, seed & 0xFFFFFFFF), "kfh_gpt2_update")

    def step(self, ids, tgt, lr=3e-4, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.1, seed=1234):
        """forward + loss, backward, update + re-quantisation: ONE call into the host library"""
        self._ids = ids
        L.check(self.ctx.host.kfh_gpt2_step(self.h, ids.data_ptr(), tgt.data_ptr(), lr, beta1, beta2, eps, wd, seed & 0xFFFFFFFF), "kfh_gpt2_step")

    def n_params(self):
        return sum(e["p"].numel() for e in self.params)
