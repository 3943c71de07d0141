"""kf_linear with nTok > 1: per-shape time (hip events around 50 back-to-back launches)."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L, runtime as R

ctx = R.Context(0)
import _knobs
_knobs.apply(ctx.hip)   # KF_G3_TILES / KF_G3_FIRST ... (scratch/_knobs.py)
shapes = [(2048, 1024), (1024, 1024), (1024, 2048), (3072, 1024), (1024, 3072), (6400, 5120), (5120, 3200)]
if os.environ.get('UB_SHAPES'): shapes = [tuple(int(v) for v in sh.split('x')) for sh in os.environ['UB_SHAPES'].split(',')]
ns = [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256, 1024]
for (m, k) in shapes:
    w = torch.randn(m, k, device=ctx.device).mul_(0.02).to(torch.bfloat16)
    dw = ctx.quantize(w, {"q4": L.Q4, "bf16": L.BF16, "f8": L.F8E5M2}[os.environ.get("UB_TYPE", "q4")])
    d = dw.desc()
    for n in ns:
        x = torch.randn(n, k, device=ctx.device).to(torch.bfloat16)
        y = torch.zeros(n, m, device=ctx.device, dtype=torch.bfloat16)
        def run():
            assert ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y.data_ptr(), None, n, 1.0, 0.0, 0, None) == 0
        for _ in range(5): run()
        ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print("M=%5d K=%5d n=%5d  %8.1f us  %7.1f TFLOP/s  %6.2f Tw/s" % (m, k, n, us, 2.0 * m * k * n / us / 1e6, m * k * ((n + 127) // 128) / us / 1e6), flush=True)
