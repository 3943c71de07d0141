import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx=Context(0); dev=ctx.device
def timeit(name, fn, n=50, reps=20):
    for _ in range(3): fn()
    L.check(ctx.hip.kf_graph_begin(ctx.h))
    for _ in range(n): fn()
    g=C.c_void_p(); L.check(ctx.hip.kf_graph_end(ctx.h, C.byref(g)))
    ctx.hip.kf_graph_launch(ctx.h,g); ctx.sync()
    e0,e1=ctx.event(),ctx.event(); ctx.record(e0)
    for _ in range(reps): ctx.hip.kf_graph_launch(ctx.h,g)
    ctx.record(e1)
    us=ctx.elapsed_ms(e0,e1)*1e3/(n*reps)
    print(f"{name:48s} {us:8.2f} us"); return us
nh,nkv,hd=16,8,128; S=2048
kc=torch.randn(S,nkv*hd,device=dev).to(torch.bfloat16); vc=torch.randn(S,nkv*hd,device=dev).to(torch.bfloat16)
q=torch.randn(nh*hd,device=dev).to(torch.bfloat16); kraw=torch.randn(nkv*hd,device=dev).to(torch.bfloat16)
qn=torch.ones(hd,device=dev,dtype=torch.bfloat16); table=ctx.rope_table(S,hd,1e6)
ws=ctx._ws(nh,hd); out=torch.zeros(nh*hd,dtype=torch.bfloat16,device=dev)
dp=torch.zeros(1,dtype=torch.int32,device=dev)
for pos in (255,383,511,767,1023,1535,2047):
    dp[0]=pos
    timeit(f"attn_block pos={pos}", lambda: L.check(ctx.hip.kf_attn_block(ctx.h,_ptr(q),_ptr(kraw),_ptr(kc),_ptr(vc),_ptr(out),_ptr(qn),_ptr(qn),_ptr(table),pos,_ptr(dp),nh,nkv,hd,nkv*hd,1e-6,_ptr(ws))))
