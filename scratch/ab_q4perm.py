"""A/B of the register-table 4-bit dot (KF_Q4_PERM=1) against the arithmetic form (0) inside one process pair on one box: bit-identity of a
kf_linear output on three shapes, then bench.py decode rates."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, ctypes as C, torch, hashlib
sys.path.insert(0, %r)
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(5)
h = hashlib.sha256()
for (m, k) in [(4096, 1024), (1024, 3072), (1024, 2048), (25600, 5120), (151936, 1024)]:
    W = (torch.randn(m, k, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    x = torch.randn(k, device=dev, generator=g).to(torch.bfloat16)
    w = ctx.quantize(W, L.Q4)
    y = ctx.linear(w, x)
    a = ctx.norm_gateup_swiglu(x, torch.ones(k, device=dev, dtype=torch.bfloat16), w, w)
    ctx.sync()
    h.update(y.view(torch.int16).cpu().numpy().tobytes()); h.update(a.view(torch.int16).cpu().numpy().tobytes())
print(h.hexdigest())
''' % root
for v in ("0", "1"):
    env = dict(os.environ, KF_Q4_PERM=v)
    print("KF_Q4_PERM=%s digest" % v, subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip())
for rep in range(2):
    for v in ("0", "1", "-1"):
        env = dict(os.environ, KF_Q4_PERM=v)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-seconds", "0", "--streams", "0"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        j = json.loads(out)
        print("KF_Q4_PERM=%s: %.1f tok/s, %.4f ms/step, matvec %.2f us" % (v, j["value"], j["ms_per_step"], j["roofline"].get("us_per_launch", 0)))
