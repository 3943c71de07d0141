"""Row-codebook (NF4) 4-bit storage against the Packed128 RTN form on the shapes of the Qwen3 decode step: the LM head (151936 x 1024), the 0.6B layer
launches, a 32B-sized projection; plus quantiser / dequant rates.  Each timing: HIP events around REPS launches with a 320 MB flush between them."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context, _ptr
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
rw = lambda m, k: (torch.randn(m, k, device=dev) * 0.02).to(torch.bfloat16)
flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
REPS = 20

def timeit(f, flush_each=True):
    f(); ctx.sync()
    tot = 0.0
    for _ in range(REPS):
        if flush_each: flush.add_(1)
        a, b = ctx.event(), ctx.event()
        ctx.record(a); f(); ctx.record(b)
        tot += ctx.elapsed_ms(a, b)
    return tot / REPS * 1e3  # us

for (m, k) in [(151936, 1024), (4096, 1024), (1024, 3072), (25600, 5120), (8192, 5120)]:
    W = rw(m, k)
    x = torch.randn(k, device=dev).to(torch.bfloat16)
    y = torch.zeros(m, dtype=torch.bfloat16, device=dev)
    row = []
    for name, w in (("q4 rtn", ctx.quantize(W, L.Q4)), ("nf4 lut", ctx.quantize_nf4(W))):
        d = w.desc()
        us = timeit(lambda: L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), _ptr(x), _ptr(y), None, 1, 1.0, 0.0, 0, None)))
        row.append("%s %.1f us (%.0f GB/s)" % (name, us, w.algorithmic_bytes() / us / 1e3))
    print("mat-vec %6d x %5d: %s" % (m, k, "; ".join(row)))
W = rw(151936, 1024)
us = timeit(lambda: ctx.quantize_nf4(W), False); print("NF4 quantise 151936 x 1024: %.0f us (%.0f GB/s of bf16 read + nibbles written)" % (us, W.numel() * 2.5 / us / 1e3))
w = ctx.quantize_nf4(W); out = torch.empty_like(W); d = w.desc()
us = timeit(lambda: L.check(ctx.hip.kf_dequant(ctx.h, C.byref(d), _ptr(out)))); print("NF4 dequant  151936 x 1024: %.0f us (%.0f GB/s of nibbles read + bf16 written)" % (us, W.numel() * 2.5 / us / 1e3))
