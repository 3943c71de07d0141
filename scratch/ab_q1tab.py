"""A/B inside one box of the 1-bit / 2-bit mat-vecs through the LDS selector tables (KF_Q1_TAB / KF_Q2_TAB = 1, default) against the per-weight
arithmetic forms (0): output digests (must be equal) and bench.py --layers 1bit / ternary rates.  Usage: ab_q1tab.py [1bit|ternary]"""
import os, subprocess, sys, json
KIND = sys.argv[1] if len(sys.argv) > 1 else "1bit"
KNOB = "KF_Q1_TAB" if KIND == "1bit" else "KF_Q2_TAB"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from koifish_amd.runtime import Context
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(5)
h = hashlib.sha256()
for (m, k) in [(4096, 1024), (1024, 3072), (1000, 2048), (40, 3200 if False else 3072), (8192, 5120), (151936, 1024)]:
    W = (torch.randn(m, k, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    x = torch.randn(k, device=dev, generator=g).to(torch.bfloat16)
    w = ctx.quantize(W, L.BOOL1 if %r == '1bit' else L.T_SIGN)
    y = ctx.linear(w, x)
    a = ctx.norm_gateup_swiglu(x, torch.ones(k, device=dev, dtype=torch.bfloat16), w, w)
    lg, am = ctx.lm_head(w, x)
    ctx.sync()
    for t in (y, a, lg):
        h.update(t.view(torch.int16).cpu().numpy().tobytes())
    h.update(str(am).encode())
print("DIGEST", h.hexdigest())
''' % (root, KIND)
for v in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{KNOB: v}), capture_output=True, text=True)
    print("%s=%s" % (KNOB, v), r.stdout.strip()[-80:], r.stderr.strip()[-300:] if r.returncode else "")
for rep in range(2):
    for v in ("0", "1"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-seconds", "0", "--streams", "0", "--layers", KIND], env=dict(os.environ, **{KNOB: v}),
                             capture_output=True, text=True).stdout.strip().splitlines()[-1]
        j = json.loads(out)
        print("%s=%s: %.1f tok/s, %.4f ms/step" % (KNOB, v, j["value"], j["ms_per_step"]))
