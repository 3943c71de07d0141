#!/usr/bin/env python3
"""bench_tp.py -- Qwen3-32B 4-bit tensor-parallel decode (BASELINE.json configs[3]), one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench_tp.py --steps 64

Every rank draws each full tensor on its own GPU from the same seed (so no 64 GB host blob and no broadcast), quantises it with
kf_quantize and keeps only its shard (koifish_amd.tp.TPPlan).  Per token: 2 all-gathers of a 5120-vector of fp32 per layer + one
gather of (max, index) pairs, torch.distributed backend "nccl" (= RCCL over xGMI).  This driver is Python-stepped (correctness and
sharding first); bench.py remains the contract benchmark.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--layers", type=int, default=0, help="cut the model to this many layers (0 = all 64)")
    ap.add_argument("--config", default="qwen3-32b")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    from koifish_amd import lib as L
    from koifish_amd import synth
    from koifish_amd import tp as TP
    from koifish_amd.runtime import Context, stream

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    stream(local)
    ctx = Context(local)
    cfg = dict(synth.CONFIGS[args.config])
    if args.layers:
        cfg["n_layer"] = args.layers
    cfg["max_seq"] = max(256, args.prompt + args.steps + args.warmup + 1)
    plan = TP.TPPlan(cfg, world)
    g = torch.Generator(device=ctx.device)
    g.manual_seed(1234)   # same seed on every rank: identical full tensors, each rank keeps its slice

    def mat(r, c):
        return (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * 0.02).to(torch.bfloat16)

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)

    w, norms = {}, {}
    emb = ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    w[(-1, 0)] = emb
    head = emb if cfg.get("tied", True) else ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    w[(-1, 1)] = TP.shard_rows(head, *plan.head_rows(rank))
    norms[(-1, 0)] = nrm(cfg["dim"])
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(synth.SLOTS):
            w[(li, si)] = plan.shard(s, ctx.quantize(mat(*synth.SHAPES[s](cfg)), L.Q4), rank)
        norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(cfg["head_dim"]), nrm(cfg["head_dim"])
    torch.cuda.empty_cache()
    drv = TP.DistributedTP(TP.TPRank(plan, rank, ctx, w, norms))
    prompt = np.random.default_rng(7).integers(0, cfg["vocab"], size=args.prompt)
    pos, nxt = 0, None
    for t in prompt:                       # token-serial prefill (Fish::Chat, GoPT.cpp:1139-1146)
        nxt = drv.step(int(t), pos)
        pos += 1
    for _ in range(args.warmup):
        nxt = drv.step(nxt, pos)
        pos += 1
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    ids = []
    for _ in range(args.steps):
        nxt = drv.step(nxt, pos)
        ids.append(nxt)
        pos += 1
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        bytes_rank = sum(x.algorithmic_bytes() for k, x in w.items() if k != (-1, 0))
        print(json.dumps({"metric": "tokens/s, Qwen3-32B-shaped 4-bit decode, tensor parallel", "value": round(args.steps / float(tt.item()), 3), "unit": "tokens/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(tt.item()) * 1e3 / args.steps, 3),
                          "config": {"workload": "%s, %d layers, TP=%d, prompt %d" % (args.config, cfg["n_layer"], world, args.prompt),
                                     "weight_bytes_per_rank": int(bytes_rank), "driver": "python-stepped, torch.distributed all_gather (RCCL)"},
                          "greedy_ids_head": ids[:8]}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
