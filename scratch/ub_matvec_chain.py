"""The 84 gemv_kernel<2,1,0> launches of one Qwen3-0.6B decode step ([norm+QKV], [o_proj+residual], [down_proj+residual] x 28 layers, each layer
with its own synthetic 4-bit weights), launched eagerly, with an LM-head-sized streaming pass between repetitions so that the weights come from
HBM as in the real step.  The stand-alone target of the `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE` passes behind profiles/r01_pmc_matvec.json
(--pmc with bench.py itself crashes the profiler on this pool)."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
dim, ffn, nL = 1024, 3072, 28
rw = lambda m, k: (torch.randn(m, k, device=dev) * 0.02).to(torch.bfloat16)
x = torch.randn(dim, device=dev).to(torch.bfloat16); nw = torch.ones(dim, device=dev, dtype=torch.bfloat16)
y = torch.zeros(dim, dtype=torch.bfloat16, device=dev)
att = torch.randn(2048, device=dev).to(torch.bfloat16); act = torch.randn(ffn, device=dev).to(torch.bfloat16)
keep, launches, nbytes = [], [], 0
for l in range(nL):
    qkv = [ctx.quantize(rw(m, dim), L.Q4) for m in (2048, 1024, 1024)]
    wo, wd = ctx.quantize(rw(dim, 2048), L.Q4), ctx.quantize(rw(dim, ffn), L.Q4)
    descs = [w.desc() for w in qkv]; outs = [torch.zeros(w.ne0, dtype=torch.bfloat16, device=dev) for w in qkv]
    wp = (C.c_void_p * 3)(*[C.addressof(d) for d in descs]); yp = (C.c_void_p * 3)(*[o.data_ptr() for o in outs])
    do, dd = wo.desc(), wd.desc()
    keep += [qkv, wo, wd, descs, outs, wp, yp, do, dd]
    launches.append(lambda wp=wp, yp=yp: L.check(ctx.hip.kf_norm_linear(ctx.h, _ptr(x), _ptr(nw), 1e-6, 3, wp, yp, None, 0, None)))
    launches.append(lambda do=do: L.check(ctx.hip.kf_linear(ctx.h, C.byref(do), _ptr(att), _ptr(y), None, 1, 1.0, 0.0, 1, _ptr(x))))
    launches.append(lambda dd=dd: L.check(ctx.hip.kf_linear(ctx.h, C.byref(dd), _ptr(act), _ptr(y), None, 1, 1.0, 0.0, 1, _ptr(x))))
    nbytes += sum(w.algorithmic_bytes() for w in qkv) + wo.algorithmic_bytes() + wd.algorithmic_bytes() + 2 * (dim + 2048 + ffn) + 2 * (4096 + 2 * dim) + 4 * dim
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
reps = int(os.environ.get("REPS", "10"))
for r in range(reps):
    flush.add_(1)
    for f in launches: f()
ctx.sync()
print("launches per repetition %d, algorithmic bytes per launch %.0f" % (len(launches), nbytes / len(launches)))
