import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from koifish_amd.runtime import Context
from koifish_amd import lib as L
from oracle import oracle as O
from conftest import bf16_t, u16
ctx=Context(0)
rng=np.random.default_rng(11)
w=O.f32_to_bf16(rng.normal(0,0.02,size=(64,256)).astype(np.float32))
for t in (L.T_SIGN, L.BOOL1):
    ow=O.quantize(w,64,256,t)
    dw=ctx.quantize(bf16_t(w,ctx.device),t)
    z,s=dw.zero_step()
    print(t,"gpu step",s[:4].float().cpu().numpy(),"oracle step",O.bf16_to_f32(ow.step[:4]))
    print("gpu bytes",dw.blob[:16].cpu().numpy(),"oracle",ow.data[:16])
