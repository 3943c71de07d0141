"""Per-phase wall-clock stamps of one workgroup of the persistent decode engine (the diagnostic instantiation of the kernel): where a layer's time goes.
   STAMP_WG=77 [ENG_DELAY=16,8,12,16,16,16] python scratch/eng_stamps.py [pos]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import lib as L
from koifish_amd import synth

pos = int(sys.argv[1]) if len(sys.argv) > 1 else 2040
cfg = synth.CONFIGS["qwen3-0.6b"]
m = synth.build_on_gpu(cfg, seed=1234)
forced = np.random.default_rng(7).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
m.set_forced(forced)
if os.environ.get("CANON"):
    m.set_canonical(int(os.environ["CANON"]))
if os.environ.get("ENG_DELAY"):
    d = (C.c_int * 6)(*[int(v) for v in os.environ["ENG_DELAY"].split(",")])
    assert m.host.kfh_engine_set_delays(m.h, d) == 0
assert m.host.kfh_engine_stamps_enable(m.h, int(os.environ.get("STAMP_WG", "77"))) == 0
m.set_state(int(forced[pos - 4]), pos - 4)
m.run_steps(pos - 4, 4, True)
m.sync()
m.engine_check()
nl = cfg["n_layer"]
buf = (C.c_uint64 * (nl * 32))()
m.host.kfh_engine_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
n = m.host.kfh_engine_stamps(m.h, buf, nl * 32)
a = np.frombuffer(buf, dtype=np.uint64).reshape(nl, 2, 16).astype(np.int64)
t0 = a[8, 0, 0]
names_p = ["P1 poll start", "P1 staged", "qkv staged", "attn barriers done", "merge done", "P4 staged", "P5 staged", "P6 staged"]
names_c = ["P1 barrier", "P1 stores issued", "qkv barrier", "attention done", "P4 barrier", "P4 done", "P5 barrier", "P5 done", "P6 barrier", "P6 done"]
for l in range(8, 8 + int(os.environ.get('STAMP_LAYERS', '1'))):
    print("layer %d  (us from layer 8's start)" % l)
    print("  poller :", "  ".join("%s %+.2f" % (names_p[k], (a[l, 0, k] - t0) / 100.0) for k in range(8)))
    print("  wave 0 :", "  ".join("%s %+.2f" % (names_c[k], (a[l, 1, k] - t0) / 100.0) for k in range(10)))
sw = a[4:, 0, 8]
print("sweeps per poll (mean): P1 %.1f  P4 %.1f  P5 %.1f  P6 %.1f" % tuple(((sw >> (16 * i)) & 0xffff).mean() for i in range(4)))
for l in range(8, 12):
    print('  layer %d merge: barriers done %+.2f, first sweep issued %+.2f, sweeps %d, good %+.2f, merge done %+.2f | wave0: attention done (stores issued) %+.2f, stores acknowledged %+.2f' % (
        l, (a[l, 0, 3] - a[l, 0, 0]) / 100.0, (a[l, 0, 9] - a[l, 0, 0]) / 100.0, a[l, 0, 10], (a[l, 0, 11] - a[l, 0, 0]) / 100.0, (a[l, 0, 4] - a[l, 0, 0]) / 100.0,
        (a[l, 1, 3] - a[l, 0, 0]) / 100.0, (a[l, 1, 10] - a[l, 0, 0]) / 100.0))
for l in range(8, 10):
    print('  layer %d attention (wave 0): raw heads staged %+.2f, heads prepared %+.2f, keys summed %+.2f, wave sums in LDS %+.2f, barrier %+.2f, partial stores issued %+.2f' % (
        l, *[(a[l, 1, k] - a[l, 0, 0]) / 100.0 for k in (2, 11, 12, 13, 14, 3)]))
d = (a[1:, 0, 0] - a[:-1, 0, 0]) / 100.0
print("layer period (us): mean %.2f  min %.2f  max %.2f" % (d[2:].mean(), d[2:].min(), d[2:].max()))
# per-phase durations averaged over layers 4..27 (poller view)
seg = np.diff(a[4:, 0, :8], axis=1) / 100.0
print("poller segments, mean us:", "  ".join("%s %.2f" % (names_p[k + 1], seg[:, k].mean()) for k in range(7)))
segc = np.diff(a[4:, 1, :10], axis=1) / 100.0
print("wave-0 segments, mean us:", "  ".join("%s %.2f" % (names_c[k + 1], segc[:, k].mean()) for k in range(9)))
