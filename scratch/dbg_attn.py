import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx=Context(0); dev=ctx.device
nh,nkv,hd=16,8,128; S=2048
kc=torch.randn(S,nkv*hd,device=dev).to(torch.bfloat16); vc=torch.randn(S,nkv*hd,device=dev).to(torch.bfloat16)
q=torch.randn(nh*hd,device=dev).to(torch.bfloat16); kraw=torch.randn(nkv*hd,device=dev).to(torch.bfloat16)
qn=torch.ones(hd,device=dev,dtype=torch.bfloat16); table=ctx.rope_table(S,hd,1e6)
ws=ctx._ws(nh,hd); out=torch.zeros(nh*hd,dtype=torch.bfloat16,device=dev)
dp=torch.zeros(1,dtype=torch.int32,device=dev)
for pos in (100, 383, 2047):
    dp[0]=pos
    for _ in range(8):
        L.check(ctx.hip.kf_attn_block(ctx.h,_ptr(q),_ptr(kraw),_ptr(kc),_ptr(vc),_ptr(out),_ptr(qn),_ptr(qn),_ptr(table),pos,_ptr(dp),nh,nkv,hd,nkv*hd,1e-6,_ptr(ws)))
