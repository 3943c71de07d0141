"""Large 4-bit mat-vec shapes of Qwen3-32B, timed alone with cold weights (6 rotating copies): KF_GEMV_WAVES / KF_GEMV_G / KF_GEMV_STREAM sweeps."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
def bench(name, M, K, t, reps=30, nsets=6):
    ws = [ctx.quantize((torch.randn(M, K, device=dev) * 0.02).to(torch.bfloat16), t) for _ in range(nsets)]
    x = torch.randn(K, device=dev).to(torch.bfloat16); y = torch.zeros(M, dtype=torch.bfloat16, device=dev)
    ds = [w.desc() for w in ws]
    def f(i): L.check(ctx.hip.kf_linear(ctx.h, C.byref(ds[i % nsets]), _ptr(x), _ptr(y), None, 1, 1.0, 0.0, 0, None))
    for i in range(nsets): f(i)
    e0, e1 = ctx.event(), ctx.event(); ctx.record(e0)
    for i in range(reps): f(i)
    ctx.record(e1); us = ctx.elapsed_ms(e0, e1) * 1e3 / reps
    b = ws[0].algorithmic_bytes()
    print(f"{name:12s} {M}x{K}: {us:8.1f} us  {b/us/1e3:7.0f} GB/s  (waves={os.environ.get('KF_GEMV_WAVES')} stream={os.environ.get('KF_GEMV_STREAM')})", flush=True)
T = {"q4": L.Q4, "ternary": L.T_SIGN, "1bit": L.BOOL1, "bf16": L.BF16, "f8": L.F8E5M2}
for n in (sys.argv[1:] or ["q4"]):
    bench(n, 25600, 5120, T[n])
    bench(n + " down", 5120, 25600, T[n])
    bench(n + " qkv", 8192, 5120, T[n])
