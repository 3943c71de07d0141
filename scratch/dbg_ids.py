import sys, numpy as np, torch
sys.path.insert(0,'.')
from koifish_amd import lib as L, synth
from koifish_amd.runtime import stream
from oracle import oracle as O
stream(0)
cfg=dict(synth.CONFIGS["qwen3-0.6b"])
m=synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
S=cfg["max_seq"]
forced=np.full(S,-1,dtype=np.int32); forced[:128]=np.random.default_rng(7).integers(0,cfg["vocab"],size=128)
m.set_forced(forced); m.set_state(int(forced[0]),0); m.run_steps(0,160,True); m.sync()
ids=m.tokens_out(S)
print("gpu ids 120..160:", ids[120:160])
om=O.from_device_model(m)
gk,gv=m.kv_to_host(); ok,ov=om.kv(); ok[:,:128]=gk[:,:128]; ov[:,:128]=gv[:,:128]
tok=int(ids[127])
for p in range(128,140):
    nxt,lg,_=om.decode(tok,p)
    l=O.bf16_to_f32(lg); top=np.argsort(l)[-3:][::-1]
    print(p,"oracle",nxt,"gpu",ids[p],"top3",top,l[top], "gpu-tok logit", l[ids[p]])
    tok=int(ids[p])
