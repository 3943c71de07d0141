"""A/B of engine knobs inside ONE box: decode ms/step and engine-only us per launch at the long positions, per environment setting.
   python scratch/eng_ab.py "ENG_DELAY=0,0,0,0,0,0" "ENG_DELAY=12,4,8,10,12,12" ...   (each argument: space-separated NAME=VALUE pairs; '-' = no change)
Every setting runs in a child process (the knobs are read when the engine is built)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(%r))
from koifish_amd import lib as L
from koifish_amd import synth
cfg = synth.CONFIGS["qwen3-0.6b"]
m = synth.build_on_gpu(cfg, seed=1234)
forced = np.random.default_rng(7).integers(0, cfg["vocab"], size=cfg["max_seq"]).astype(np.int32)
m.set_forced(forced)
if os.environ.get("ENG_DELAY"):
    d = (C.c_int * 6)(*[int(v) for v in os.environ["ENG_DELAY"].split(",")])
    assert m.host.kfh_engine_set_delays(m.h, d) == 0
if os.environ.get("CANON"):
    m.set_canonical(int(os.environ["CANON"]))
for kv in os.environ.get("KF_KNOBS", "").split():
    k, v = kv.split("=")
    m.hip.kfdbg_set_knob.argtypes = [C.c_char_p, C.c_long]
    assert m.hip.kfdbg_set_knob(k.encode(), int(v)) == 0
if os.environ.get("TUNE"):
    m.set_engine_autotune(int(os.environ["TUNE"]))
pos0 = 1900
m.set_state(int(forced[pos0]), pos0)
m.run_steps(pos0, 20, True)   # warm: graph of the bucket
m.sync()
n = 120
m.set_state(int(forced[pos0 + 20]), pos0 + 20)
t0 = time.perf_counter()
m.run_steps(pos0 + 20, n, True)
m.sync()
dt = (time.perf_counter() - t0) / n * 1e3
m.engine_check()
ids = m.tokens_out(cfg["max_seq"])[pos0 + 20:pos0 + 20 + n]
dig = int((np.asarray(ids).astype(np.int64) * np.arange(1, n + 1)).sum())
# engine alone at position 2040 (events on the launch stream, warm)
m.set_state(int(forced[2040]), 2040)
m.engine_only(3)
m.sync()
e0, e1 = m._ctx.event(), m._ctx.event()
m._ctx.record(e0)
m.engine_only(50)
m._ctx.record(e1)
m.sync()
us = m._ctx.elapsed_ms(e0, e1) / 50 * 1e3
m.engine_check()
st = m.engine_stats(2040)
print("RESULT %%.4f %%.1f %%d %%s" %% (dt, us, dig, ("delays=%%s tuned=%%d sweeps/poll=%%s" %% (",".join(map(str, st["delay"])), st["tuned"], ",".join("%%.2f" %% v for v in st["sweeps_per_poll"]))).replace(" ", "_")))
""" % (HERE,)


def main():
    sets = sys.argv[1:] or ["-"]
    for s in sets:
        env = dict(os.environ)
        if s != "-":
            for kv in s.split():
                k, v = kv.split("=", 1)
                env[k] = v
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        res = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        if not res:
            print("%-60s FAILED: %s" % (s, (out.stderr or out.stdout)[-400:]))
            continue
        dt, us, dig, st = res[-1].split()[1:]
        print("%-60s %s ms/step (%.0f tok/s)  engine-only %s us  ids digest %s  %s" % (s, dt, 1e3 / float(dt), us, dig, st), flush=True)


if __name__ == "__main__":
    main()
