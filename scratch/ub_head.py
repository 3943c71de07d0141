import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0,'.')
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx=Context(0); dev=ctx.device
V,D=151936,1024
w=ctx.quantize((torch.randn(V,D,device=dev)*0.02).to(torch.bfloat16), L.BF16)
x=torch.randn(D,device=dev).to(torch.bfloat16); nw=torch.ones(D,device=dev,dtype=torch.bfloat16)
logits=torch.empty(V,dtype=torch.bfloat16,device=dev); st=torch.zeros(4,dtype=torch.int32,device=dev); d=w.desc()
def f(): L.check(ctx.hip.kf_norm_lm_head(ctx.h,_ptr(x),_ptr(nw),1e-6,C.byref(d),_ptr(logits),_ptr(st),None,_ptr(ctx._head_ws)))
for _ in range(5): f()
e0,e1=ctx.event(),ctx.event(); ctx.record(e0)
for _ in range(100): f()
ctx.record(e1); us=ctx.elapsed_ms(e0,e1)*10
print(f"head waves={os.environ.get('KF_GEMV_WAVES')} G={os.environ.get('KF_GEMV_G')}: {us:.1f} us  {V*D*2/us/1e3:.0f} GB/s")
