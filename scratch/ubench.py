import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0,'.')
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx=Context(0)
dev=ctx.device
def rw(m,k): return (torch.randn(m,k,device=dev)*0.02).to(torch.bfloat16)
def timeit(name, fn, n=50, reps=20, nbytes=0):
    for _ in range(3): fn()
    L.check(ctx.hip.kf_graph_begin(ctx.h))
    for _ in range(n): fn()
    g=C.c_void_p()
    L.check(ctx.hip.kf_graph_end(ctx.h, C.byref(g)))
    ctx.hip.kf_graph_launch(ctx.h,g); ctx.sync()
    e0,e1=ctx.event(),ctx.event()
    ctx.record(e0)
    for _ in range(reps): ctx.hip.kf_graph_launch(ctx.h,g)
    ctx.record(e1)
    us=ctx.elapsed_ms(e0,e1)*1e3/(n*reps)
    print(f"{name:48s} {us:8.2f} us" + (f"  {nbytes/us/1e3:8.1f} GB/s" if nbytes else ""))
    return us
dim,ffn=1024,3072
x=torch.randn(dim,device=dev).to(torch.bfloat16); nw=torch.ones(dim,device=dev,dtype=torch.bfloat16)
wq=ctx.quantize(rw(2048,dim),L.Q4); wk=ctx.quantize(rw(1024,dim),L.Q4); wv=ctx.quantize(rw(1024,dim),L.Q4)
wo=ctx.quantize(rw(dim,2048),L.Q4); wg=ctx.quantize(rw(ffn,dim),L.Q4); wu=ctx.quantize(rw(ffn,dim),L.Q4); wd=ctx.quantize(rw(dim,ffn),L.Q4)
wsmall=ctx.quantize(rw(64,dim),L.Q4)
ys=[torch.zeros(w.ne0,dtype=torch.bfloat16,device=dev) for w in (wq,wk,wv)]
def norm_linear(ws, ys, norm=True):
    descs=[w.desc() for w in ws]
    wp=(C.c_void_p*len(ws))(*[C.addressof(d) for d in descs]); yp=(C.c_void_p*len(ws))(*[y.data_ptr() for y in ys])
    def f(): L.check(ctx.hip.kf_norm_linear(ctx.h,_ptr(x),_ptr(nw) if norm else None,1e-6,len(ws),wp,yp,None,0,None))
    f._keep=(descs,wp,yp)
    return f
timeit("norm+QKV (4096x1024 q4)", norm_linear([wq,wk,wv],ys), nbytes=sum(w.algorithmic_bytes() for w in (wq,wk,wv)))
timeit("QKV no norm", norm_linear([wq,wk,wv],ys,False))
timeit("norm+linear 64x1024 (fixed cost)", norm_linear([wsmall],[torch.zeros(64,dtype=torch.bfloat16,device=dev)]))
timeit("linear 64x1024 no norm", norm_linear([wsmall],[torch.zeros(64,dtype=torch.bfloat16,device=dev)],False))
act=torch.zeros(ffn,dtype=torch.bfloat16,device=dev); dg,du=wg.desc(),wu.desc()
timeit("norm+gate/up+swiglu (2x3072x1024)", lambda: L.check(ctx.hip.kf_norm_gateup_swiglu(ctx.h,_ptr(x),_ptr(nw),1e-6,C.byref(dg),C.byref(du),_ptr(act))), nbytes=wg.algorithmic_bytes()*2)
y=torch.zeros(dim,dtype=torch.bfloat16,device=dev); dd=wd.desc(); do=wo.desc(); att=torch.randn(2048,device=dev).to(torch.bfloat16)
timeit("down+res (1024x3072)", lambda: L.check(ctx.hip.kf_linear(ctx.h,C.byref(dd),_ptr(act),_ptr(y),None,1,1.0,0.0,1,_ptr(x))), nbytes=wd.algorithmic_bytes())
timeit("o_proj+res (1024x2048)", lambda: L.check(ctx.hip.kf_linear(ctx.h,C.byref(do),_ptr(att),_ptr(y),None,1,1.0,0.0,1,_ptr(x))), nbytes=wo.algorithmic_bytes())
# attention
nh,nkv,hd=16,8,128
S=2048
kc=torch.randn(S,nkv*hd,device=dev).to(torch.bfloat16); vc=torch.randn(S,nkv*hd,device=dev).to(torch.bfloat16)
q=torch.randn(nh*hd,device=dev).to(torch.bfloat16); kraw=torch.randn(nkv*hd,device=dev).to(torch.bfloat16)
qn=torch.ones(hd,device=dev,dtype=torch.bfloat16); table=ctx.rope_table(S,hd,1e6)
ws=ctx._ws(nh,hd); out=torch.zeros(nh*hd,dtype=torch.bfloat16,device=dev)
for pos in (127,1023,2047):
    kvb=2*(pos+1)*nkv*hd*2
    timeit(f"attn_block pos={pos}", lambda: L.check(ctx.hip.kf_attn_block(ctx.h,_ptr(q),_ptr(kraw),_ptr(kc),_ptr(vc),_ptr(out),_ptr(qn),_ptr(qn),_ptr(table),pos,None,nh,nkv,hd,nkv*hd,1e-6,_ptr(ws))), nbytes=kvb)
# trivial kernels
st=torch.zeros(4,dtype=torch.int32,device=dev)
timeit("set_state (1 thread)", lambda: ctx.hip.kf_set_state(ctx.h,_ptr(st),1,2))
a=torch.zeros(1024,dtype=torch.bfloat16,device=dev)
timeit("add 1024", lambda: ctx.hip.kf_add(ctx.h,_ptr(a),_ptr(a),_ptr(a),1024))
timeit("rmsnorm 1024", lambda: ctx.hip.kf_rmsnorm(ctx.h,_ptr(x),_ptr(nw),_ptr(a),1,1024,1e-6,None))
