"""Decode ms/step of the 0.6B engine per position range, canonical and v_dot2c order, autotune on/off: python scratch/eng_positions.py
(whole-sequence behaviour: the bench's `value` is timed at 2028..2047 only)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from koifish_amd import synth
cfg = synth.CONFIGS["qwen3-0.6b"]
m = synth.build_on_gpu(cfg, seed=1234)
forced = np.random.default_rng(7).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
m.set_forced(forced)
for canon in (1, 0):
    for tune in (0, 2):
        m.set_canonical(canon)
        m.set_engine(False); m.set_engine(True)   # a fresh engine: no delays carried over
        m.set_engine_autotune(tune)
        row = []
        for p0 in (128, 256, 512, 1024, 1536, 1984):
            m.set_state(int(forced[p0]), p0)
            m.run_steps(p0, 16, True); m.sync()
            m.set_state(int(forced[p0 + 16]), p0 + 16)
            t0 = time.perf_counter()
            m.run_steps(p0 + 16, 48, True); m.sync()
            row.append((time.perf_counter() - t0) / 48 * 1e3)
            m.engine_check()
        st = m.engine_stats(2000)
        print("canonical=%d autotune=%d  ms/step at 144.. 272.. 528.. 1040.. 1552.. 2000..: %s   delays(2000)=%s sweeps/poll=%s" % (canon, tune, " ".join("%.3f" % v for v in row), st["delay"], st["sweeps_per_poll"]), flush=True)
