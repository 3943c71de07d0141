"""Eight XCD-confined decoders (kf_xengine_*) on Qwen3-0.6B 4-bit: aggregate tokens/s at a position range, per variant (waves per workgroup x ring depth), n_seq sweep,
and the per-phase stamps of one workgroup.
  python scratch/xr_time.py [pos0=2028] [steps=20]      env: VARIANTS="9x8,13x6,16x4,9x12"  NSEQ="8,4,1"  STAMPS=1  DEAL="0,8,11,16" (xe_deal's weight, 0 = default)  CONFIG=qwen3-4b"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd.runtime import XcdReplicas

pos0 = int(sys.argv[1]) if len(sys.argv) > 1 else 2028
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = dict(synth.CONFIGS[os.environ.get("CONFIG", "qwen3-0.6b")])
m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
m.set_canonical(True)
S = cfg["max_seq"]
variants = [tuple(int(x) for x in v.split("x")) for v in os.environ.get("VARIANTS", "9x8").split(",")]
for n_seq in [int(x) for x in os.environ.get("NSEQ", "8").split(",")]:
    xr = XcdReplicas(m, n_seq)
    for s in range(n_seq):
        f = np.full(S, -1, dtype=np.int32)
        f[:128] = np.random.default_rng(7 + s).integers(0, cfg["vocab"], size=128)
        xr.set_forced(s, f)
    bytes_step = m.step_bytes(pos0 + steps // 2)
    for (nwv, depth, deal) in [(v[0], v[1], int(d)) for v in variants for d in os.environ.get("DEAL", "0").split(",")]:
        xr.variant(nwv, depth)
        xr.variant(-1, deal)
        for s in range(n_seq):   # the K / V rows below pos0 hold whatever earlier runs left (zeros at first): the arithmetic does not care
            xr.set_state(s, 1 + s, pos0 - 4)
        xr.run_steps(4)
        m.sync()
        xr.check()
        best = 1e9
        for rep in range(3):
            for s in range(n_seq):
                xr.set_state(s, 1 + s, pos0)
            m.sync()
            t0 = time.perf_counter()
            xr.run_steps(steps)
            m.sync()
            best = min(best, time.perf_counter() - t0)
        xr.check()
        tps = n_seq * steps / best
        print("n_seq %d  waves %2d depth %2d deal %2d  positions %d..%d: %.3f ms per step (all sequences), %.1f tokens/s aggregate, %.1f per sequence, %.1f GB/s = %.3f of 8 TB/s" % (
            n_seq, nwv, depth, deal, pos0, pos0 + steps - 1, best * 1e3 / steps, tps, tps / n_seq, bytes_step * tps / 1e9, bytes_step * tps / 8e12), flush=True)
    if os.environ.get("STAMPS") and n_seq in (8, 16, 32):
        xr.variant(-1, int(os.environ.get("STAMP_DEAL", "0")))
        nl = cfg["n_layer"]
        xr.stamps(int(os.environ.get("STAMP_SEQ", "3")), int(os.environ.get("STAMP_WG", "5")), 2, nl)
        for s in range(n_seq):
            xr.set_state(s, 1 + s, pos0)
        xr.set_steps_per_launch(2)
        xr.run_steps(2)
        m.sync()
        st = xr.stamps(0, 0, -2, nl).astype(np.int64)
        names = {0: "poll x", 1: "x staged", 2: "qkv staged", 3: "sums in LDS", 4: "partials seen", 5: "ao stored", 6: "ao staged", 7: "xB staged", 8: "act staged",
                 16: "P1 go", 17: "P1 pub", 24: "attn done", 18: "P4 go", 19: "P4 pub", 20: "P5 go", 21: "P5 pub", 22: "P6 go", 23: "P6 pub",
                 9: "own qkv out", 10: "own xB out", 11: "own act out", 12: "own x out"}
        order = [0, 12, 1, 16, 17, 9, 2, 3, 24, 5, 6, 18, 19, 10, 7, 20, 21, 11, 8, 22, 23]
        for step in (1,):
            acc = np.zeros(len(order))
            cnt = 0
            for l in range(2, nl - 1):
                t = st[step, l]
                base = t[0]
                if base == 0:
                    continue
                acc += np.array([(t[k] - base) / 100.0 for k in order])
                cnt += 1
            nxt = np.mean([(st[step, l + 1, 0] - st[step, l, 0]) / 100.0 for l in range(2, nl - 2)])
            print("stamps of sequence / workgroup (us from the layer's first poll, mean over layers; layer period %.2f us):" % nxt)
            for k, v in zip(order, acc / max(cnt, 1)):
                print("   %-14s %7.2f" % (names[k], v))
            print("   sweeps of the act poll: %.2f" % np.mean([st[step, l, 13] for l in range(2, nl - 1)]))
            print("   shader clock: %.0f MHz" % (100.0 * (st[step, nl - 2, 31] - st[step, 2, 31]) / (st[step, nl - 2, 0] - st[step, 2, 0])))
            hw = [int(st[step, 5, k]) for k in (14, 15, 25, 26, 27, 28, 29, 30)]
            print("   compute waves 0..7: SIMD", [(h >> 4) & 3 for h in hw], "CU", [(h >> 8) & 15 for h in hw], "wave slot", [h & 15 for h in hw])
            for q, nm in enumerate(("q|k|v", "o_proj", "gate|up", "down")):
                go = {0: 16, 1: 18, 2: 20, 3: 22}[q]
                fin = np.mean([[(st[step, l, 32 + 8 * q + w] - st[step, l, go]) / 100.0 for w in range(8)] for l in range(2, nl - 1)], axis=0)
                print("   %-8s per-wave finish (us after the phase's barrier):" % nm, " ".join("%.2f" % v for v in fin))
    xr.close()
m.close()
