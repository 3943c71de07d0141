"""Child process of tests/test_gpu_q4_variants.py: the 4-bit mat-vec in the form argv[1] selects (kfdbg_set_knob("q4_perm", 0 | 1)),
plain / fused-norm / SwiGLU-pair / arg-max entries, against the oracle.  Exits non-zero on the first mismatch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from koifish_amd import lib as L  # noqa: E402
from koifish_amd.runtime import Context  # noqa: E402
from oracle import oracle as O  # noqa: E402


def ulp(a, b):
    def key(u):
        u = u.astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - (u & 0x7fff), u + 0x8000)   # monotone integer key of a bf16 bit pattern
    return np.abs(key(np.asarray(a)) - key(np.asarray(b)))


def main():
    import ctypes as C
    ctx = Context(0)
    ctx.hip.kfdbg_set_knob.argtypes = [C.c_char_p, C.c_long]
    assert ctx.hip.kfdbg_set_knob(b"q4_perm", int(sys.argv[1]) if len(sys.argv) > 1 else 1) == 0
    dev = ctx.device
    rng = np.random.default_rng(77)
    bf = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(dev).view(torch.bfloat16)
    u16 = lambda t: t.detach().cpu().view(torch.int16).numpy().view(np.uint16)
    for (m, k) in [(1024, 1024), (768, 3072), (4096, 2048), (12800, 5120), (96, 256)]:
        w = O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32))
        x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
        ow = O.quantize(w, m, k, L.Q4)
        dw = ctx.upload_blob(L.Q4, m, k, ow.blob())
        y = u16(ctx.linear(dw, bf(x)))
        ref = O.linear(ow, x)
        d = ulp(y, ref)
        exact = O.bf16_to_f32(O.dequant(ow)).astype(np.float64) @ O.bf16_to_f32(x).astype(np.float64)
        yf, rf = O.bf16_to_f32(y), O.bf16_to_f32(ref)
        close = (d <= 1) | (np.abs(yf - rf) <= 2.0 ** -10 * np.abs(rf).max())   # tests/conftest.py close_bf16: 1 ulp, or far below the sum's own fp32 noise
        ok = bool(close.all()) and np.abs(yf - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 1e-6
        print("linear %dx%d: max ulp %d, mismatching %.4f" % (m, k, d.max(), (d > 0).mean()), flush=True)
        if not ok:
            return 1
    # fused RMSNorm + SwiGLU pair, and the arg-max head form, on a 4-bit matrix
    m, k = 3072, 1024
    wg, wu = (O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32)) for _ in range(2))
    nw = O.f32_to_bf16((1 + rng.normal(0, 0.1, size=k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    og, ou = O.quantize(wg, m, k, L.Q4), O.quantize(wu, m, k, L.Q4)
    dg, du = ctx.upload_blob(L.Q4, m, k, og.blob()), ctx.upload_blob(L.Q4, m, k, ou.blob())
    act = u16(ctx.norm_gateup_swiglu(bf(x), bf(nw), dg, du, 1e-6))
    xn = O.rmsnorm(x, nw, 1e-6)
    ref = O.swiglu(O.linear(og, xn), O.linear(ou, xn))
    d = ulp(act, ref)
    print("norm+gate/up+swiglu: max ulp %d" % d.max(), flush=True)
    if d.max() > 2:   # a 1-ulp difference in either projection can move the product by 2
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
