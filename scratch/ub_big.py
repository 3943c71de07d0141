import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0,'.')
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx=Context(0); dev=ctx.device
def bench(name, M, K, t, reps=30, nsets=6):
    ws=[ctx.quantize((torch.randn(M,K,device=dev)*0.02).to(torch.bfloat16), t) for _ in range(nsets)]
    x=torch.randn(K,device=dev).to(torch.bfloat16); y=torch.zeros(M,dtype=torch.bfloat16,device=dev)
    ds=[w.desc() for w in ws]
    def f(i): L.check(ctx.hip.kf_linear(ctx.h,C.byref(ds[i%nsets]),_ptr(x),_ptr(y),None,1,1.0,0.0,0,None))
    for i in range(nsets): f(i)
    e0,e1=ctx.event(),ctx.event(); ctx.record(e0)
    for i in range(reps): f(i)
    ctx.record(e1); us=ctx.elapsed_ms(e0,e1)*1e3/reps
    b=ws[0].algorithmic_bytes()
    print(f"{name:34s} {M}x{K}: {us:8.1f} us  {b/us/1e3:7.0f} GB/s  ({b/1e6:.0f} MB, waves={os.environ.get('KF_GEMV_WAVES')})")
for t,n in ((L.Q4,"q4"),(L.BF16,"bf16"),(L.F8E5M2,"f8"),(L.T_SIGN,"ternary"),(L.BOOL1,"1bit")):
    bench(n, 25600, 5120, t)
bench("q4 down", 5120, 25600, L.Q4)
bench("q4 qkv-ish", 8192, 5120, L.Q4)
