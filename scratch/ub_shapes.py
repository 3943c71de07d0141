"""4-bit mat-vec launches timed alone in the canonical order, cold weights (rotating copies), per knob setting inside ONE process:
    python scratch/ub_shapes.py [tp|big|all]     tp: the TP = 8 rank shards of Qwen3-32B (the launches of Fish::TPPhase); big: the whole matrices on one GPU
Knob sweeps: gemv_waves (0 = the launcher's rule), gemv_xf2 (the two-window fp32 staging of the 25600-wide rows)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L
from koifish_amd.runtime import Context, _ptr

ctx = Context(0)
dev = ctx.device
ctx.set_canonical(True)
ctx.hip.kfdbg_set_knob.argtypes = [C.c_char_p, C.c_long]


def knob(name, v):
    assert ctx.hip.kfdbg_set_knob(name.encode(), int(v)) == 0, name


def bench(name, M, K, settings, reps=200):
    nsets = max(2, min(8, int(600e6 / (M * K * 0.53))))
    ws = [ctx.quantize((torch.randn(M, K, device=dev) * 0.02).to(torch.bfloat16), L.Q4) for _ in range(nsets)]
    x = torch.randn(K, device=dev).to(torch.bfloat16)
    y = torch.zeros(M, dtype=torch.bfloat16, device=dev)
    ds = [w.desc() for w in ws]

    def f(i):
        L.check(ctx.hip.kf_linear(ctx.h, C.byref(ds[i % nsets]), _ptr(x), _ptr(y), None, 1, 1.0, 0.0, 0, None))
    b = ws[0].algorithmic_bytes()
    out = []
    for st in settings:
        for k, v in st.items():
            knob(k, v)
        for i in range(nsets):
            f(i)
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for i in range(reps):
            f(i)
        ctx.record(e1)
        us = ctx.elapsed_ms(e0, e1) * 1e3 / reps
        out.append("%s %.1f us (%.0f GB/s)" % (",".join("%s=%s" % kv for kv in st.items()), us, b / us / 1e3))
    print("%-22s %6d x %-6d %5.1f MB: %s" % (name, M, K, b / 1e6, "   ".join(out)), flush=True)
    knob("gemv_waves", 0)
    knob("gemv_xf2", 1)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
W = [{"gemv_waves": w} for w in (0, 1024, 2048, 4096, 8192)]
if which in ("tp", "all"):
    bench("o_proj shard", 5120, 1024, W)
    bench("down_proj shard", 5120, 3200, W)
    bench("q rows shard", 1280, 5120, W)
    bench("gate rows shard", 3200, 5120, W)
if which in ("big", "all"):
    bench("down_proj 32B", 5120, 25600, [{"gemv_xf2": 1}, {"gemv_xf2": 0}], reps=60)
    bench("gate_proj 32B", 25600, 5120, [{"gemv_waves": 0}], reps=60)
    bench("o_proj 32B", 5120, 8192, [{"gemv_waves": 0}], reps=60)
