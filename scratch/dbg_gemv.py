import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx=Context(0); dev=ctx.device
def rw(m,k): return (torch.randn(m,k,device=dev)*0.02).to(torch.bfloat16)
dim,ffn=1024,3072
x=torch.randn(dim,device=dev).to(torch.bfloat16); nw=torch.ones(dim,device=dev,dtype=torch.bfloat16)
wq=ctx.quantize(rw(2048,dim),L.Q4); wk=ctx.quantize(rw(1024,dim),L.Q4); wv=ctx.quantize(rw(1024,dim),L.Q4)
wo=ctx.quantize(rw(dim,2048),L.Q4); wg=ctx.quantize(rw(ffn,dim),L.Q4); wu=ctx.quantize(rw(ffn,dim),L.Q4); wd=ctx.quantize(rw(dim,ffn),L.Q4)
ys=[torch.zeros(w.ne0,dtype=torch.bfloat16,device=dev) for w in (wq,wk,wv)]
descs=[w.desc() for w in (wq,wk,wv)]
wp=(C.c_void_p*3)(*[C.addressof(d) for d in descs]); yp=(C.c_void_p*3)(*[y.data_ptr() for y in ys])
act=torch.zeros(ffn,dtype=torch.bfloat16,device=dev); dg,du=wg.desc(),wu.desc()
y=torch.zeros(dim,dtype=torch.bfloat16,device=dev); dd=wd.desc(); do=wo.desc(); att=torch.randn(2048,device=dev).to(torch.bfloat16)
big=torch.zeros(64<<20,dtype=torch.uint8,device=dev)
for _ in range(4):
    big.add_(1)  # push the weights out of the caches
    L.check(ctx.hip.kf_norm_linear(ctx.h,_ptr(x),_ptr(nw),1e-6,3,wp,yp,None,0,None))
for _ in range(4):
    big.add_(1)
    L.check(ctx.hip.kf_norm_gateup_swiglu(ctx.h,_ptr(x),_ptr(nw),1e-6,C.byref(dg),C.byref(du),_ptr(act)))
for _ in range(4):
    big.add_(1)
    L.check(ctx.hip.kf_linear(ctx.h,C.byref(dd),_ptr(act),_ptr(y),None,1,1.0,0.0,1,_ptr(x)))
for _ in range(4):
    big.add_(1)
    L.check(ctx.hip.kf_linear(ctx.h,C.byref(do),_ptr(att),_ptr(y),None,1,1.0,0.0,1,_ptr(x)))
ctx.sync()
