"""The TP = 8 ranks of Qwen3-32B as the eight XCDs of one launch (koifish::XcdTP): ms per step at a position range and the per-phase stamps of one workgroup of one rank.
    python scratch/xtp_time.py [layers=16] [pos0=4000] [steps=16]     env: STAMPS=1 STAMP_RANK=3 STAMP_WG=5"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd import tp as TP
from koifish_amd.runtime import Context, XcdTP

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pos0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg = dict(synth.CONFIGS[os.environ.get("CONFIG", "qwen3-32b")], n_layer=layers)
ffn_real = cfg["ffn"]
if cfg["ffn"] % 1024:   # Qwen3-4B: the FFN padded to whole groups per rank (zero rows / columns)
    cfg["ffn"] = (cfg["ffn"] + 1023) // 1024 * 1024
ctx = Context(0)
g = torch.Generator(device=ctx.device)
g.manual_seed(1)


def mat(r, c):
    t = (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * 0.02).to(torch.bfloat16)
    if r == cfg["ffn"] and ffn_real < r:
        t[ffn_real:] = 0
    if c == cfg["ffn"] and ffn_real < c:
        t[:, ffn_real:] = 0
    return t


def nrm(n):
    return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)


w, norms = {}, {}
w[(-1, 0)] = ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
w[(-1, 1)] = w[(-1, 0)]
norms[(-1, 0)] = nrm(cfg["dim"])
for li in range(layers):
    for si, s in enumerate(synth.SLOTS):
        w[(li, si)] = ctx.quantize(mat(*synth.SHAPES[s](cfg)), L.Q4)
    norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(128), nrm(128)
nt = TP.NativeTP(cfg, w, norms, 8, ctx)
for rk in nt.ranks:
    rk.set_canonical(True)
xt = XcdTP(nt)
forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
forced[:16] = 5
xt.set_forced(forced)
xt.set_steps_per_launch(steps)
if os.environ.get("VARIANT"):   # a library built with -DXE_TP_VARIANTS: "12x8", "12x4", "12x62" (two key tiles per batch), "8x8"
    xt.variant(*[int(v) for v in os.environ["VARIANT"].split("x")])
best = 1e9
for rep in range(3):
    xt.set_state(7, pos0)          # the K / V rows below pos0 are zeros: the arithmetic does not care
    ctx.sync()
    t0 = time.perf_counter()
    xt.run_steps(steps)
    ctx.sync()
    best = min(best, time.perf_counter() - t0)
xt.check()
weights = sum(x.algorithmic_bytes() for k, x in w.items() if k[0] >= 0) + w[(-1, 1)].algorithmic_bytes()
kv = 2 * layers * (pos0 + steps / 2) * cfg["n_kv"] * cfg["head_dim"] * 2
print("%d layers, positions %d..%d: %.3f ms per step = %.1f us per layer (head included), %.1f GB/s = %.3f of 8 TB/s" % (
    layers, pos0, pos0 + steps - 1, best * 1e3 / steps, best * 1e6 / steps / layers, (weights + kv) / (best / steps) / 1e9, (weights + kv) / (best / steps) / 8e12), flush=True)
if os.environ.get("STAMPS"):
    xt.stamps(int(os.environ.get("STAMP_RANK", "3")), int(os.environ.get("STAMP_WG", "5")), 2, layers)
    xt.set_state(7, pos0)
    xt.set_steps_per_launch(2)
    xt.run_steps(2)
    ctx.sync()
    st = xt.stamps(0, 0, -2, layers).astype(np.int64)
    names = {0: "poll x", 1: "x staged", 2: "qkv staged", 3: "sums in LDS", 4: "partials seen", 5: "ao stored", 6: "ao staged", 7: "xB staged", 8: "act staged",
             16: "P1 go", 17: "P1 pub", 24: "attn done", 18: "P4 go", 19: "P4 pub", 20: "P5 go", 21: "P5 pub", 22: "P6 go", 23: "P6 pub",
             9: "own qkv out", 10: "own xB out", 11: "own act out", 12: "own x out", 31: "x rows summed", 13: "xB rows summed"}
    order = [0, 12, 31, 1, 16, 17, 9, 2, 3, 24, 4, 5, 6, 18, 19, 10, 13, 7, 20, 21, 11, 8, 22, 23]
    acc = np.zeros(len(order))
    cnt = 0
    for l in range(2, layers - 1):
        t = st[1, l]
        if t[0] == 0:
            continue
        acc += np.array([(t[k] - t[0]) / 100.0 for k in order])
        cnt += 1
    per = np.mean([(st[1, l + 1, 0] - st[1, l, 0]) / 100.0 for l in range(2, layers - 2)])
    print("stamps (us from the layer's first poll, mean over layers; layer period %.2f us):" % per)
    for k, v in zip(order, acc / max(cnt, 1)):
        print("   %-14s %7.2f" % (names[k], v))
    for q, nm in enumerate(("q|k|v", "o_proj", "gate|up", "down")):
        go = {0: 16, 1: 18, 2: 20, 3: 22}[q]
        fin = np.mean([[(st[1, l, 32 + 8 * q + w] - st[1, l, go]) / 100.0 for w in range(8)] for l in range(2, layers - 1)], axis=0)
        print("   %-8s per-wave finish (us after the phase's barrier):" % nm, " ".join("%.2f" % v for v in fin))
xt.close()
nt.close()
