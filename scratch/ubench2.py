import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0,'.')
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx=Context(0); dev=ctx.device
def rw(m,k): return (torch.randn(m,k,device=dev)*0.02).to(torch.bfloat16)
def timeit(name, fns, reps=20):
    for f in fns[:3]: f()
    L.check(ctx.hip.kf_graph_begin(ctx.h))
    for f in fns: f()
    g=C.c_void_p(); L.check(ctx.hip.kf_graph_end(ctx.h, C.byref(g)))
    ctx.hip.kf_graph_launch(ctx.h,g); ctx.sync()
    e0,e1=ctx.event(),ctx.event(); ctx.record(e0)
    for _ in range(reps): ctx.hip.kf_graph_launch(ctx.h,g)
    ctx.record(e1)
    us=ctx.elapsed_ms(e0,e1)*1e3/(len(fns)*reps)
    print(f"{name:60s} {us:8.2f} us"); return us
dim=1024
x=torch.randn(dim,device=dev).to(torch.bfloat16); nw=torch.ones(dim,device=dev,dtype=torch.bfloat16)
def mk(nsets, shape=(4096,dim), norm=True):
    fns=[]; keep=[]
    for i in range(nsets):
        w=ctx.quantize(rw(*shape),L.Q4); y=torch.zeros(shape[0],dtype=torch.bfloat16,device=dev)
        d=w.desc(); wp=(C.c_void_p*1)(C.addressof(d)); yp=(C.c_void_p*1)(y.data_ptr())
        keep.append((w,y,d,wp,yp))
        fns.append((lambda wp=wp,yp=yp: L.check(ctx.hip.kf_norm_linear(ctx.h,_ptr(x),_ptr(nw) if norm else None,1e-6,1,wp,yp,None,0,None))))
    fns[0]._keep=keep
    return fns
for n in (1,28,300):
    f=mk(n); timeit(f"norm+linear 4096x1024 q4, {n} weight sets ({n*2.2:.0f} MB)", f if n>1 else f*50)
for n in (1,300):
    f=mk(n,norm=False); timeit(f"linear (no norm) 4096x1024 q4, {n} sets", f if n>1 else f*50)
# dependent chain: y of kernel i feeds x of kernel i+1 (1024x1024)
ws=[ctx.quantize(rw(1024,1024),L.Q4) for _ in range(50)]
bufs=[torch.randn(1024,device=dev).to(torch.bfloat16) for _ in range(51)]
fns=[]
descs=[w.desc() for w in ws]
for i in range(50):
    fns.append(lambda i=i: L.check(ctx.hip.kf_linear(ctx.h,C.byref(descs[i]),_ptr(bufs[i]),_ptr(bufs[i+1]),None,1,1.0,0.0,0,None)))
timeit("dependent chain linear 1024x1024 (x from previous kernel)", fns)
fns=[]
for i in range(50):
    fns.append(lambda i=i: L.check(ctx.hip.kf_linear(ctx.h,C.byref(descs[i]),_ptr(bufs[0]),_ptr(bufs[i+1]),None,1,1.0,0.0,0,None)))
timeit("independent linear 1024x1024 (same x)", fns)
