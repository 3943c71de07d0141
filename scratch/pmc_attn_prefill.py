"""Prompt attention (kf_attn_prefill) at the long-prompt shape of the bench -- 2047 tokens, 16 query / 8 kv heads of 128 -- three launches, for rocprofv3 passes; prints the time of 20 more."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd.runtime import Context
ctx = Context(0); dev = ctx.device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2047
H, KV, hd = 16, 8, 128
q = torch.randn(n, H * hd, device=dev).to(torch.bfloat16); k = torch.randn(n, KV * hd, device=dev).to(torch.bfloat16); v = torch.randn(n, KV * hd, device=dev).to(torch.bfloat16)
o = torch.zeros_like(q)
def run():
    assert ctx.hip.kf_attn_prefill(ctx.h, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 0, n, H * hd, H, KV, hd, KV * hd) == 0, ctx.hip.kf_last_error()
for _ in range(3): run()
ctx.sync()
if len(sys.argv) > 2:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.ExternalStream(ctx.stream_ptr()) if hasattr(ctx, "stream_ptr") else None
    import time
    t0 = time.perf_counter()
    for _ in range(50): run()
    ctx.sync()
    print("n=%d: %.1f us per launch (wall, 50 back to back)" % (n, (time.perf_counter() - t0) * 1e6 / 50))
if os.environ.get("KF_LIB_DIR", "").endswith("apstamp"):
    import ctypes as C
    buf = (C.c_ulonglong * 8)()
    assert ctx.hip.kfdbg_ap_stamps(buf) == 0
    names = ["loads issued + barrier wait", "S MFMAs issued", "scores + max", "rescale", "exp + pack", "P.V issued", "tile stored"]
    ns = int(buf[7])
    print("stamps of a long-walk wave (s_memtime ticks of 10 ns summed over %d steps; per step in ns):" % ns)
    for k, nm in enumerate(names): print("  %-28s %8d  %7.1f ns/step" % (nm, buf[k], buf[k] * 10.0 / ns))
    print("  total %.1f us" % (sum(buf[:7]) * 10.0 / 1e3))
