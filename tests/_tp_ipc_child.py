"""One rank of a 2-process tensor-parallel run on ONE GPU (both ranks on cuda:0): the receive areas travel as IPC handles, the exchange runs
through the mapped peer memory exactly as it would across xGMI.  Launched by tests/test_gpu_tp_ipc.py via torch.distributed.run (gloo)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from koifish_amd import lib as L          # noqa: E402
from koifish_amd import synth            # noqa: E402
from koifish_amd import tp as TP         # noqa: E402
from koifish_amd.runtime import Context   # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = Context(0)
    cfg = dict(synth.CONFIGS["small"])
    raw = synth.raw_weights_numpy(cfg, 31, w_std=0.1)
    plan = TP.TPPlan(cfg, world)
    w, norms = {}, {}
    emb = ctx.quantize(synth._bf16_t(raw["embed"], ctx.device), L.BF16)
    w[(-1, 0)] = emb
    w[(-1, 1)] = TP.shard_rows(emb, *plan.head_rows(rank))
    norms[(-1, 0)] = synth._bf16_t(raw["final_norm"], ctx.device)
    for li, lw in enumerate(raw["layers"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = plan.shard(s, ctx.quantize(synth._bf16_t(lw[s], ctx.device), L.Q4), rank)
        for si, s in enumerate(synth.NORMS):
            norms[(li, si)] = synth._bf16_t(lw[s], ctx.device)
    nr = TP.NativeRank(cfg, plan, rank, w, norms, 0)
    prompt = np.random.default_rng(2).integers(0, cfg["vocab"], size=10).astype(np.int32)
    n_new = 20
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:len(prompt)] = prompt
    res = {}
    for mode, use_graph in (("eager", False), ("graph", True)):
        nr.set_forced(forced)
        nr.set_state(int(prompt[0]), 0)
        nr.run_steps(0, len(prompt) + n_new - 1, use_graph)
        nr.check()
        res[mode] = [int(t) for t in nr.m.tokens_out(len(prompt) + n_new - 1)[len(prompt) - 1:]]
        dist.barrier()
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
