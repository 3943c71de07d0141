#!/usr/bin/env python3
"""bench.py -- tokens/s of the Qwen3-0.6B 4-bit PackedQ decode path on MI355X (BASELINE.json configs[1]).

A "step" is one decode token: embed -> 28 x (norm+QKV, q/k-norm+RoPE+attention, o_proj+residual, norm+gate/up+SwiGLU,
down+residual) -> final norm + LM head + greedy pick, replayed from hipGraphs with token and position held on the
device (no host round trip per token).  Workload: synthetic weights of the 0.6B architecture (N(0,0.02), RTN 4-bit g128
layers, bf16 tied LM head), a 2048-position sequence: positions [0, 2048-W-K) are set-up, the next W are warm-up, the
last K are timed (default W=128 = the prompt, K=1920 = decode to position 2047).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run; independent replicas,
                                                       no data-path collective -- "scaling": "weak")
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the CPU-baseline legs time an OpenMP mat-vec: one thread per core, pinned, so that the number is reproducible (set before libgomp starts)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

try:   # the CPUs this process may use, before an OpenMP runtime pins its initial thread to one core (OMP_PROC_BIND): child processes inherit the mask of the thread that spawns them
    _ALL_CPUS = os.sched_getaffinity(0)
except Exception:
    _ALL_CPUS = None

_T_START = time.perf_counter()
_LAPS = []


def _lap(name):
    """wall-clock bookkeeping of the run's phases (detail file only)"""
    _LAPS.append((name, round(time.perf_counter() - _T_START, 1)))


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy achieves


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1920)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--config", default="qwen3-0.6b")
    ap.add_argument("--head", default="bf16", choices=["bf16", "q4", "nf4"])
    ap.add_argument("--layers", default="q4", choices=["q4", "bf16", "f8", "ternary", "1bit", "nf4"],
                    help="weight type of the transformer layers (default: the metric's 4-bit PackedQ; the others are side measurements)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-fp16-steps", type=int, default=64, help="BASELINE config 1 beside the line (never `value`): Qwen3-0.6B with IEEE-half weights decoded on the host cores -- a 128-token "
                    "prompt fed token by token, then this many greedy steps (0 = skip)")
    ap.add_argument("--sparse", type=float, default=0.0, help="BASELINE config 5 side measurement: this fraction of every layer's FFN rows hot (seeded mask, seed 5 + layer), the rest skipped (D_matmul_sparse); use with --layers 1bit")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--engine", type=int, default=-1, help="1: the layer loop as one persistent launch (kf_engine_*); 0: five launches per layer; -1: the library default")
    ap.add_argument("--autotune", type=int, default=2, help="passes of kf_engine_tune (self-calibrated first-sweep delays of the engine's hand-offs) per position bucket; 0 = the built-in delays")
    ap.add_argument("--streams", type=int, default=0, help="side measurement after the timed region: this many INDEPENDENT decoders (own weights, own "
                    "KV cache, own HIP stream) running concurrently on the GPU over the same positions; 0/1 = skip.  Never part of `value`.")
    ap.add_argument("--xcd-replicas", type=int, default=32, help="side object beside the line (never `value`): this many INDEPENDENT sequences decoded by one launch, one decoder per XCD with "
                    "1 (<= 8 sequences), 2 (<= 16) or 4 (<= 32) sequences each (kf_xengine_*: every unpacked 4-bit block multiplied against all of a decoder's sequences), over the same timed positions; 0 = skip")
    ap.add_argument("--tp-exchange", default="p2p", choices=["p2p", "rccl"], help="--config qwen3-32b --gpus N > 1 runs tensor parallel TP = N (BASELINE config 4): "
                    "p2p = the C++ host's graph with kernel-side exchange over peer-mapped receive areas; rccl = the Python-stepped baseline with two "
                    "torch.distributed all-gathers per layer")
    ap.add_argument("--tp-virtual", type=int, default=0, help="side measurement on ONE GPU: this many TP ranks of qwen3-32b in one process, lock-step on one stream "
                    "(the per-rank kernels and the exchange kernels of TP = R, serialised: R x the work of one rank's GPU, no xGMI)")
    # test hooks (tests/test_gpu_bench_ranks.py: two ranks sharing ONE GPU over gloo exercise the multi-process path on a 1-GPU box; a cut TP model); the launcher of a bare
    # `--gpus N` hands them to its ranks on the command line -- the product bench reads no environment variable of its own
    ap.add_argument("--x-backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--x-device", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--x-tp-layers", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--x-tp-vocab", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--x-no-tp-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--tp-xcd", type=int, default=0, help="with --tp-virtual 8: 1 = the eight ranks as the eight XCDs of ONE launch (kf_xengine_create_tp) instead of the per-launch rank step; 2 = both, one after the other")
    ap.add_argument("--tp-layers", type=int, default=0, help="with --tp-virtual: this many of the model's layers (0 = all): bounds the side leg's wall time")
    ap.add_argument("--lean-cpu", type=float, default=0.0, help="with --lean: also the CPU-baseline leg (parity passes + a timed sample of this many seconds) of the model being run")
    ap.add_argument("--lean-prefill", type=int, default=0, help="with --lean: also a prompt of this many tokens through Fish::Prefill (prefill_rate)")
    ap.add_argument("--lean-xcd", type=int, default=0, help="with --lean: also the XCD-confined engines on this many independent sequences of the model being run (xcd_replicas)")
    ap.add_argument("--jump", action="store_true", help="side legs only (never the run behind `value`): no token-serial run-up to the timed window -- the K / V rows of the positions in front of it "
                    "are synthetic N(0, 1) bf16 rows written once (a step's time depends on how many rows it reads, not on their values), the decode state is set to the window's first position")
    ap.add_argument("--lean", action="store_true", help="only the timed decode and step_roofline (what the side legs run in their child processes)")
    ap.add_argument("--leg", default="", choices=["", "config3", "config4cpu"], help="run ONE side leg and print its JSON (child processes of the main run)")
    ap.add_argument("--side-legs", default="config3,config5,config4,qwen3_1p7b,qwen3_1p7b_launches,qwen3_4b", help="side objects beside the line, each measured in a child process after the main measurements "
                    "(never `value`): config3 = GPT2-1558M operator path of a training step (sum of separately timed forward+loss, backward, AdamW phases; no parameter update), config5 = 1-bit layers + 20 %% hot FFN rows, config4 = Qwen3-32B on ONE GPU; '' = none")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` run bare: THIS process becomes the launcher of N rank processes -- before torch is imported or anything touches a GPU (a process
        # that has initialised HIP must never exec or fork GPU work) -- and relays rank 0's single JSON line
        sys.exit(launch_ranks(args))
    if args.leg == "config3":
        print(json.dumps(config3_train_step()))
        return
    if args.leg == "config4cpu":
        print(json.dumps(config4_slice_cpu()))
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    from koifish_amd import lib as L
    from koifish_amd import synth
    from koifish_amd.runtime import stream

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = args.x_backend   # --x-backend / --x-device: test hooks only
    dev = args.x_device if args.x_device >= 0 else (local if world > 1 else 0)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev)
    stream(dev)

    cfg = dict(synth.CONFIGS[args.config])
    S = cfg["max_seq"]
    K, W = args.steps, args.warmup
    if K < 1 or W < 0:
        raise SystemExit("steps >= 1, warmup >= 0")
    if (args.config == "qwen3-32b" and world > 1) or args.tp_virtual > 1:   # (--tp-virtual 8 --tp-xcd 1 also serves ONE sequence of Qwen3-4B / 8B: the eight TP ranks as the eight XCDs)
        return tp_main(args, cfg, rank, world, dev)
    head_type = {"bf16": L.BF16, "q4": L.Q4, "nf4": L.NF4}[args.head]
    layer_type = {"q4": L.Q4, "bf16": L.BF16, "f8": L.F8E5M2, "ternary": L.T_SIGN, "1bit": L.BOOL1, "nf4": L.NF4}[args.layers]
    m = synth.build_on_gpu(cfg, seed=1234 + rank, layer_type=layer_type, head_type=head_type, device=dev)
    _lap("model built")
    m.set_prefill_resident(True, 96 << 30)   # opt-in (library default: off): this run never changes a weight in place, so long prompts may keep bf16 copies of the layer matrices in HBM
    ctx = m._ctx
    hots = {}
    if args.sparse > 0.0:
        for l in range(cfg["n_layer"]):
            hot = np.zeros(cfg["ffn"], dtype=np.int32)
            hot[np.random.default_rng(5 + l).permutation(cfg["ffn"])[: max(int(cfg["ffn"] * args.sparse), 16)]] = 1
            m.set_hot(l, hot)
            hots[l] = hot
    m.set_engine_autotune(args.autotune)   # kf_engine_tune once per position bucket, at the first multi-step launch inside it (set-up span: never inside the timed region)
    if args.engine >= 0:
        m.set_engine(bool(args.engine))
    elif args.x_device >= 0 and world > 1:
        m.set_engine(False)  # test hook: several ranks share one GPU; the persistent engine needs the CUs to itself
    use_graph = not args.no_graph

    # synthetic prompt: 128 ids, then free-running greedy decode
    n_prompt = 128
    forced = np.full(S, -1, dtype=np.int32)
    forced[:n_prompt] = np.random.default_rng(7 + rank).integers(0, cfg["vocab"], size=n_prompt)
    m.set_forced(forced)

    def run_span(p0, n):
        """n consecutive decode steps starting at position p0, wrapping to a fresh sequence at the end of the cache"""
        done = 0
        while done < n:
            p = (p0 + done) % S
            c = min(n - done, S - p)
            if p == 0:
                m.set_state(int(forced[0]), 0)
            m.run_steps(p, c, use_graph)
            done += c
        return (p0 + n) % S

    # lay the timed window at the end of the sequence: [setup | warmup W | timed K] ends at position S-1
    span = (W + K) % S if (W + K) % S else min(W + K, S)
    start = (S - span) % S
    # one-time set-up of every position bucket the timed window enters AFTER its first step (the host tunes the engine's hand-off delays -- ~0.2 s of launches -- and captures
    # graphs at the first multi-step launch inside a bucket; the warm-up covers the bucket it ends in): a few steps at each such boundary now, on a cache of zeros
    first_timed = (start + W) % S
    for b in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
        if b < S - 16 and any((first_timed + i) % S == b for i in range(K)):
            m.set_state(int(forced[0]), b)
            m.run_steps(b, 16, use_graph)
    m.set_state(int(forced[0]), 0)
    if args.jump and start > 0:
        import ctypes as C
        kvb = cfg["n_layer"] * S * cfg["n_kv"] * cfg["head_dim"] * 2
        _fill_kv_synthetic(m.hip, C.c_void_p(m.host.kfh_ctx(m.h)), [(m.host.kfh_kcache(m.h), kvb), (m.host.kfh_vcache(m.h), kvb)])
        m.set_state(1, start)
        pos = start
    else:
        pos = run_span(0, start) if start else 0
    pos = run_span(pos, W)
    timed_positions = [(pos + i) % S for i in range(K)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    e0, e1 = ctx.event(), ctx.event()
    _lap("run-up + warm-up")
    barrier()
    t0 = time.perf_counter()
    ctx.record(e0)
    run_span(pos, K)
    ctx.record(e1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dt = t1 - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    dev_ms = ctx.elapsed_ms(e0, e1)
    ms_per_step = dt * 1e3 / K
    value = world * K / dt

    out = None
    if rank == 0:
        bytes_steps = [m.step_bytes(p) for p in timed_positions]
        mean_bytes = float(np.mean(bytes_steps))
        out = {
            "metric": "tokens/sec/GPU Qwen3 4-bit prefill+decode; achieved HBM GB/s vs peak",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"q4": "u4", "bf16": "bf16", "f8": "f8e5m2", "ternary": "u2", "1bit": "u1", "nf4": "u4 (NF4)"}[args.layers] + " weights x bf16, fp32 accumulate",
            "data": "synthetic",
            "config": {"workload": "%s %s greedy decode, 1xMI355X per replica, seq=%d: prompt 128, timed positions %d..%d"
                                   % ({"qwen3-0.6b": "Qwen3-0.6B", "qwen3-32b": "Qwen3-32B", "qwen3-1.7b": "Qwen3-1.7B", "qwen3-4b": "Qwen3-4B", "qwen3-8b": "Qwen3-8B"}.get(args.config, args.config), {"q4": "4-bit PackedQ", "bf16": "bf16", "f8": "f8e5m2", "ternary": "2-bit ternary PackedQ", "1bit": "1-bit PackedQ", "nf4": "4-bit NF4 row-codebook"}[args.layers],
                                      S, timed_positions[0], timed_positions[-1]) + ("; K / V rows in front of the window synthetic (--jump)" if args.jump else ""),
                       "lm_head": args.head, "sparse_ffn_rows_hot": args.sparse if args.sparse > 0 else None, "replicas": world,
                       "hipgraph": bool(use_graph and m.num_graphs() > 0),  # a step that is ONE launch (the engine with head and pick) is launched directly: a one-node graph only adds replay cost
                       "device_ms_per_step": round(dev_ms / K, 5)},
            "step_roofline": {"bound": "hbm", "bytes_per_step": int(mean_bytes), "achieved": round(mean_bytes * (value / world) / 1e9, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(mean_bytes * (value / world) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        # `value` was timed in the library's default summation order: the CANONICAL one (kf_abi.h kf_set_canonical; round 4), every logit, greedy id and KV row bit-exact against
        # the CPU oracle (cpu_baseline.parity_timed_order).  Beside it, never `value`: the same positions in the v_dot2c / fp32 order (kf_set_canonical(ctx, 0)).
        out["config"]["summation_order"] = "canonical (the library default): two v_pk_fma_f32 chains per lane + tree, exact power-of-two softmax with fp64 sums -- bit-exact against the CPU oracle"
        ids_timed_run = m.tokens_out(S).copy()   # the greedy ids of the timed (canonical) run at every position: what sequence 0 of the xcd_replicas leg must reproduce
        try:
            ids0 = m.tokens_out(S)

            def rewind():   # the device state back at the first timed position (token = what the run picked before it)
                m.set_state(int(ids0[pos - 1]) if pos > 0 else int(forced[0]), pos)
            m.set_canonical(False)
            rewind()
            run_span(pos, min(K, 64))   # graphs of the bucket re-captured
            rewind()
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            run_span(pos, K)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - tf0
            out["fast_order_mode"] = {"tokens_per_s": round(K / dtf, 2), "ms_per_step": round(dtf * 1e3 / K, 5),
                                      "note": "kf_set_canonical(ctx, 0): products by v_dot2c_f32_bf16, fp32 softmax -- fewer vector instructions, <= 1 bf16 ulp per mat-vec output from the oracle, greedy ids "
                                              "may flip at near-ties (cpu_baseline.parity_fast_order classifies them); same positions; never `value`"}
        except Exception as e:
            out["fast_order_mode"] = {"error": repr(e)[:200]}
        finally:
            m.set_canonical(True)
            m.set_state(int(m.tokens_out(S)[pos - 1]) if pos > 0 else int(forced[0]), pos)
            run_span(pos, K)   # the ids, logits and KV rows the legs below read are those of the timed (canonical) run again
            torch.cuda.synchronize()
        try:   # sweeps per poll of the engine's six hand-offs over everything run so far, and the delays in use in the timed bucket (kf_engine_stats)
            if m.engine_steps() > 0:
                st = m.engine_stats(timed_positions[-1])
                out["engine_handoffs"] = dict(st, order="x (P1), q|k|v (P2), slice partials (P3), ao (P4), xB (P5), act (P6)", delays_unit="s_sleep(1) trips behind the own publish",
                                              autotune_passes=args.autotune)
        except Exception as e:
            out["engine_handoffs"] = {"error": repr(e)[:160]}
        if args.lean:
            m.engine_check()
            out["config"]["decode_path"] = "persistent engine, one launch per token" if m.engine_steps() > 0 else "per-layer launches: 5 per layer (%s)" % (m.engine_why() or "engine off")
            if args.lean_cpu > 0 and world == 1:
                try:   # the oracle with the same weights (and the same hot-row masks) on this host's cores: parity passes + a bounded timed sample
                    out["cpu_baseline"] = cpu_baseline(m, cfg, forced, args.lean_cpu, min_steps=16, max_steps=96, canon_steps=24, hots=hots)
                except Exception as e:
                    out["cpu_baseline"] = {"error": repr(e)[:300]}
            if args.lean_xcd > 0 and world == 1:
                try:
                    out["xcd_replicas"] = xcd_replicas(m, cfg, forced, timed_positions, W, args.lean_xcd, ids_timed_run, jump=args.jump)
                except Exception as e:
                    out["xcd_replicas"] = {"error": repr(e)[:300]}
            if args.lean_prefill > 0 and world == 1:
                try:
                    lp = np.random.default_rng(7).integers(0, cfg["vocab"], size=min(args.lean_prefill, S - 1)).astype(np.int32)
                    out["prefill"] = prefill_rate(m, lp, ms_per_step, reps=2, bound="mfma: 5-25 k row matrices on 256 x 256 bf16 tiles (resident copies) + flash attention")
                except Exception as e:
                    out["prefill"] = {"error": repr(e)[:200]}
            print(json.dumps(out))
            return
        _lap("timed + fast order")
        out["prefill"] = prefill_rate(m, forced[:n_prompt], ms_per_step)
        if world == 1 and args.streams > 1 and args.config == "qwen3-0.6b":
            try:
                out["concurrent_streams"] = concurrent_streams(cfg, layer_type, head_type, dev, args.streams, forced, n_prompt, mean_bytes)
            except Exception as e:   # a side measurement must never cost the bench line
                out["concurrent_streams"] = {"error": repr(e)[:200]}
        head_rl = kernel_roofline(m, ctx, cfg)
        try:   # the time-dominant kernel of the step; the LM head (the byte-dominant launch) is reported beside it
            eng = engine_roofline(m, ctx, cfg, forced, timed_positions) if m.engine_steps() > 0 else None
            out["roofline"] = eng if eng else (matvec_roofline(m, ctx, cfg, head_rl) if args.layers == "q4" else head_rl)
            out["config"]["decode_path"] = "persistent engine: kf::engine_kernel (embedding row + all layers + final norm + LM head + greedy pick) = ONE launch per token, runs of up to 16 tokens of a position bucket in one launch" if eng else \
                "per-layer launches: 5 per layer"
        except Exception as e:
            out["roofline"] = head_rl
            out["roofline_error"] = repr(e)[:200]
        m.engine_check()
        _lap("prefill + rooflines")
        out["roofline_lm_head"] = head_rl
        out["cpu_baseline"] = cpu_baseline(m, cfg, forced, args.cpu_seconds, hots=hots) if (world == 1 and args.cpu_seconds > 0) else None
        _lap("cpu_baseline")
        if args.config == "qwen3-0.6b":   # last (it overwrites the KV rows and ids the checks above read): the prompt half through a prompt that fills the context, one token batch
            try:
                long_prompt = np.random.default_rng(7).integers(0, cfg["vocab"], size=S - 1).astype(np.int32)
                out["prefill"]["long_prompt"] = prefill_rate(m, long_prompt, ms_per_step, reps=3,
                                                             bound="mfma (token-batch GEMMs with in-register 4-bit unpack over all rows of the prompt + flash attention); far from the peak at this "
                                                                   "model size: the per-layer products are 4-13 GFLOP each")
            except Exception as e:   # a side measurement must never cost the bench line
                out["prefill"]["long_prompt"] = {"error": repr(e)[:200]}
        if world == 1 and args.cpu_seconds > 0 and args.cpu_fp16_steps > 0 and args.config == "qwen3-0.6b":
            try:
                out["cpu_baseline_fp16"] = cpu_fp16_decode(cfg, ctx.device, args.cpu_fp16_steps)
            except Exception as e:   # a side measurement must never cost the bench line
                out["cpu_baseline_fp16"] = {"error": repr(e)[:200]}
        _lap("long prompt + cpu fp16")
        if args.config == "qwen3-0.6b" and args.layers == "q4" and args.sparse == 0.0 and args.xcd_replicas > 0:
            try:   # eight independent decoders, one per XCD, sharing this model's weights (kf_xengine_*): the aggregate beside the single-sequence `value`
                out["xcd_replicas"] = xcd_replicas(m, cfg, forced, timed_positions, W, args.xcd_replicas, ids_timed_run, queue=True)
                keys = ("streams", "batch", "decoders_per_xcd", "tokens_per_s", "per_stream_tokens_per_s", "ms_per_step_all_streams", "frac_vs_single_sequence_roofline", "hbm_frac_batch", "parity", "skipped")
                if args.xcd_replicas > 16 and "error" not in out["xcd_replicas"]:   # two sequences per decoder beside the four
                    e16 = xcd_replicas(m, cfg, forced, timed_positions, W, 16, ids_timed_run)
                    out["xcd_replicas"]["two_per_xcd"] = {k: e16.get(k) for k in keys}
                if args.xcd_replicas > 8 and "error" not in out["xcd_replicas"]:   # the one-sequence-per-XCD form beside it (what VERDICT r04 asked for by name)
                    e8 = xcd_replicas(m, cfg, forced, timed_positions, W, 8, ids_timed_run)
                    out["xcd_replicas"]["one_per_xcd"] = {k: e8.get(k) for k in keys}
            except Exception as e:   # a side measurement must never cost the bench line
                out["xcd_replicas"] = {"error": repr(e)[:300]}
        _lap("xcd_replicas")
        if world == 1 and args.config == "qwen3-0.6b" and args.layers == "q4" and args.sparse == 0.0:
            m.close()
            del m
            torch.cuda.empty_cache()
            try:
                out.update(side_legs([l for l in args.side_legs.split(",") if l]))
            except Exception as e:   # side objects must never cost the bench line
                out["side_legs_error"] = repr(e)[:300]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        _lap("side legs")
        out["wall_s"] = round(time.perf_counter() - _T_START, 1)
        out["wall_laps_s"] = _LAPS
        emit(out)


def _fill_kv_synthetic(hip, ctx_h, ptrs_bytes, seed=11):
    """--jump: N(0, 1) bf16 rows into K / V caches (device pointers + byte counts) from one 16 MB host block copied at consecutive offsets (it repeats every 8192 rows of 1024)"""
    import ctypes as C
    import numpy as np
    from koifish_amd import lib as L
    blk = (np.random.default_rng(seed).standard_normal(8 << 20, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
    for ptr, nbytes in ptrs_bytes:
        off = 0
        while off < nbytes:
            n = min(blk.nbytes, nbytes - off)
            L.check(hip.kf_h2d(ctx_h, C.c_void_p(ptr + off), blk.ctypes.data_as(C.c_void_p), C.c_size_t(n)), "kf_h2d")
            off += n


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(n, argv, timeout_s):
    """n fresh rank processes of this script under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1): (return code, rank 0's JSON line or None, tail of stderr)"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py")] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
    except subprocess.TimeoutExpired as e:
        return 124, None, "timeout after %d s: %s" % (timeout_s, (e.stderr or b"")[-300:] if isinstance(e.stderr, (bytes, str)) else "")
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (lines[-1] if lines else None), r.stderr[-2000:]


def launch_ranks(args):
    """The parent of a bare `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment).  It never imports torch and never touches a GPU.  Two child jobs, each N
    fresh rank processes: (1) the command line as given -- for the default config the 0.6B decode as N independent replicas (`value`, "scaling": "weak"); (2) when (1) was the
    default config, BASELINE config 4 -- Qwen3-32B tensor parallel TP = N over the kernel-side exchange ("scaling": "strong") -- whose line rides inside (1)'s as
    `config4_tp`, so that ONE driver command per N yields both curves north_star asks for.  A failure of (1) is this process's exit code; a failure of (2) costs only its object."""
    argv = [a for a in sys.argv[1:]]
    rc, line, err = _run_ranks(args.gpus, argv, 3000)
    if rc != 0 or line is None:
        sys.stderr.write(err + "\n")
        return rc if rc != 0 else 1
    out = json.loads(line)   # rank 0's COMPACT line (its detail file: bench_detail.json)
    if args.config == "qwen3-0.6b" and not args.x_no_tp_leg:
        tp_argv = ["--gpus", str(args.gpus), "--config", "qwen3-32b", "--steps", "64", "--warmup", "16", "--tp-exchange", args.tp_exchange, "--x-backend", args.x_backend,
                   "--x-device", str(args.x_device), "--x-tp-layers", str(args.x_tp_layers), "--x-tp-vocab", str(args.x_tp_vocab)]
        rc2, line2, err2 = _run_ranks(args.gpus, tp_argv, 1500)
        side = out.setdefault("side", {})
        if rc2 == 0 and line2:
            d = json.loads(line2)   # tp_main's compact line (detail: bench_detail_tp.json)
            side["config4_tp"] = {"tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "n_gpus": d["n_gpus"], "scaling": d["scaling"], "tp": d["config"].get("tp"),
                                  "exchange": d["config"].get("exchange"), "ranks_in_process_group": d["config"].get("ranks_in_process_group"), "layers": d["config"].get("layers"),
                                  "decode_path": d["config"].get("decode_path"), "frac": (d.get("roofline") or {}).get("frac"), "detail": d.get("detail")}
        else:
            side["config4_tp"] = {"error": (err2 or "")[-200:], "returncode": rc2}
    print(json.dumps(out, separators=(",", ":")), flush=True)
    return 0


def _child(argv, timeout_s):
    """one side leg in a child process (its own HIP context and memory; a failure or a timeout costs only that object): the last JSON line it prints"""
    import subprocess
    t0 = time.perf_counter()
    try:
        def unpin():   # the parent's OpenMP runtime bound this thread to one core (2 hardware threads): the child's CPU legs would see 2 "cores"
            if _ALL_CPUS:
                os.sched_setaffinity(0, _ALL_CPUS)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout_s, preexec_fn=unpin)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or r.stdout)[-300:]}
        d = json.loads(lines[-1])
        d["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        return d
    except Exception as e:   # a side measurement must never cost the bench line
        return {"error": repr(e)[:300]}


def _tp_legs(layers=64):
    """BASELINE config 4's rank step without an 8-GPU node, ONE child process for both forms (the shards are built once): (1) the 8 ranks of TP = 8 (8 q-heads + 1 kv-head + 3200 FFN
    rows + 18992 vocabulary rows each) in lock-step on one stream with the kernel-side exchange through local pointers -- every rank's kernels and the exchange kernels run,
    serialised; no xGMI; (2) the SAME ranks as the eight XCDs of ONE launch (kf_xengine_create_tp).  All 64 layers; K / V rows in front of the timed window are synthetic (--jump)."""
    d = _child(["--config", "qwen3-32b", "--tp-virtual", "8", "--tp-xcd", "2", "--tp-layers", str(layers), "--steps", "24", "--warmup", "8", "--jump"], 600)
    if "error" in d:
        return d, d
    ms8 = d["ms_per_step"]   # 8 ranks serialised
    xcd = d.pop("xcd", {"error": "missing"})
    return ({"workload": d["config"]["workload"], "layers_run": layers, "layers_of_model": 64, "ms_per_step_8_ranks_serialised": ms8, "ms_per_rank_step": round(ms8 / 8, 4),
             "bytes_per_step_per_rank": d["roofline"]["bytes_per_step_per_rank"], "achieved_GBs_per_rank_kernel_time": d["roofline"]["achieved"], "frac": d["roofline"]["frac"],
             "decode_path": d["config"]["decode_path"], "summation_order": "canonical (library default)",
             "note": "what ONE rank's GPU would spend per token at TP = 8 if the exchange cost nothing more than here (ms_per_rank_step, all %d layers); no scaling curve has been measured on hardware" % layers,
             "leg_wall_s": d.get("leg_wall_s")}, xcd)


def side_legs(which):
    """The other single-GPU configurations of BASELINE.json beside the line (never `value`), each re-derivable from the profile named in it."""
    out = {}
    if "config3" in which:
        out["config3_train_step"] = _child(["--leg", "config3"], 420)
    if "config5" in which:
        d = _child(["--layers", "1bit", "--sparse", "0.2", "--steps", "512", "--warmup", "64", "--lean", "--lean-cpu", "6", "--lean-xcd", "32"], 420)
        out["config5_sparse_1bit"] = d if "error" in d else {
            "workload": "Qwen3-0.6B, 1-bit PackedQ layers (YinYang), 20 %% of every FFN's rows hot (D_matmul_sparse: cold rows cost no HBM), bf16 head; positions %s" % d["config"]["workload"].split("timed positions ")[-1],
            "tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "bytes_per_step": d["step_roofline"]["bytes_per_step"], "frac": d["step_roofline"]["frac"],
            "fast_order_tokens_per_s": d.get("fast_order_mode", {}).get("tokens_per_s"),
            "summation_order": d["config"].get("summation_order"), "cpu_baseline": d.get("cpu_baseline"), "engine_handoffs": d.get("engine_handoffs"),
            "xcd_replicas": d.get("xcd_replicas"),   # round 6: the same model (1-bit layers, the same hot-row masks) through the batched XCD decoders, 32 independent sequences
            "decode_path": d["config"]["decode_path"], "profile": "profiles/r04_config5_sparse_1bit_kernel_stats.csv", "leg_wall_s": d.get("leg_wall_s")}
    if "qwen3_1p7b" in which:   # not a BASELINE configuration: the second model shape the persistent engine is instantiated for (round 4)
        d = _child(["--config", "qwen3-1.7b", "--steps", "64", "--warmup", "16", "--lean", "--lean-xcd", "16", "--jump"], 420)
        e = _child(["--config", "qwen3-1.7b", "--steps", "64", "--warmup", "16", "--lean", "--engine", "0", "--jump"], 420) if "qwen3_1p7b_launches" in which else {}
        out["qwen3_1p7b_shape"] = d if "error" in d else {
            "workload": "Qwen3-1.7B shape (dim 2048, 16 / 8 heads of 128, ffn 6144), 4-bit PackedQ greedy decode: %s" % d["config"]["workload"].split("seq=")[-1],
            "tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "bytes_per_step": d["step_roofline"]["bytes_per_step"], "frac": d["step_roofline"]["frac"],
            "fast_order_tokens_per_s": d.get("fast_order_mode", {}).get("tokens_per_s"), "decode_path": d["config"]["decode_path"],
            "per_layer_launches_tokens_per_s": e.get("value"), "per_layer_launches_ms_per_step": e.get("ms_per_step"), "xcd_replicas": d.get("xcd_replicas"), "leg_wall_s": d.get("leg_wall_s")}
    for nm, title in (("qwen3_4b", "Qwen3-4B shape (dim 2560, 32 / 8 heads of 128, ffn 9728, 36 layers)"), ("qwen3_8b", "Qwen3-8B shape (dim 4096, 32 / 8 heads of 128, ffn 12288, 36 layers)")):
        if nm in which:   # not BASELINE configurations: the GQA-4 models the reference lists as supported (cases/tutorial/history.md:4-6); round 5: served by the XCD-confined engines
            d = _child(["--config", nm.replace("_", "-"), "--steps", "64", "--warmup", "16", "--lean", "--lean-xcd", "8", "--jump"], 600)
            # ONE sequence through the eight XCDs as the eight tensor-parallel ranks of the model (round 6: kf_xengine_create_tp serves the TP = 8 ranks of Qwen3-8B and of Qwen3-4B with
            # its FFN padded to whole groups per rank); the bits are those of the TP = 8 rank step (oracle: tests/test_gpu_tp.py), not of the unsplit model
            t = _child(["--config", nm.replace("_", "-"), "--tp-virtual", "8", "--tp-xcd", "1", "--steps", "24", "--warmup", "8", "--jump"], 600)
            out[nm + "_shape"] = d if "error" in d else {
                "one_sequence_tp_over_xcds": t if "error" in t else {k: t.get(k) for k in ("workload", "tokens_per_s", "ms_per_step", "bytes_per_step", "frac", "summation_order", "parity", "leg_wall_s")},
                "workload": "%s, 4-bit PackedQ greedy decode: %s" % (title, d["config"]["workload"].split("seq=")[-1]),
                "tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "bytes_per_step": d["step_roofline"]["bytes_per_step"], "frac": d["step_roofline"]["frac"],
                "decode_path": d["config"]["decode_path"], "xcd_replicas": d.get("xcd_replicas"), "leg_wall_s": d.get("leg_wall_s")}
    if "config4" in which:
        d = _child(["--config", "qwen3-32b", "--steps", "48", "--warmup", "16", "--lean", "--lean-prefill", "2047", "--jump"], 600)
        tpv, tpx = _tp_legs()
        out["config4_one_gpu"] = d if "error" in d else {
            "workload": "Qwen3-32B 4-bit PackedQ greedy decode on ONE MI355X (the reference shards it over 8 GPUs for memory): %s" % d["config"]["workload"].split("seq=")[-1],
            "tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "bytes_per_step": d["step_roofline"]["bytes_per_step"], "frac": d["step_roofline"]["frac"],
            "fast_order_tokens_per_s": d.get("fast_order_mode", {}).get("tokens_per_s"),
            "decode_path": d["config"]["decode_path"], "profile": "profiles/r04_config4_one_gpu_kernel_stats.csv", "leg_wall_s": d.get("leg_wall_s"),
            "prefill_2047_tokens": {k: (d.get("prefill") or {}).get(k) for k in ("ms", "tokens_per_s", "first_call_ms", "resident_copy_bytes", "roofline", "error") if (d.get("prefill") or {}).get(k) is not None},
            "cpu_baseline_4_layer_slice": _child(["--leg", "config4cpu"], 420),
            "tp8_virtual_ranks": tpv, "tp8_ranks_as_xcds": tpx,
            "note": "TP = 8 over xGMI needs an 8-GPU node: bench.py --config qwen3-32b --gpus 8 (no scaling curve has been measured on hardware)"}
    return out


def config3_train_step():
    """BASELINE config 3: ONE whole training step of the hybrid-precision GPT2-1558M on one MI355X, in ONE timed region (koifish_amd/train_step.py, the loop
    tests/test_gpu_train_step.py::test_gpt2_two_consecutive_steps_with_update checks against the oracle at toy size): n_embd 1600, 48 layers, 25 heads, ffn 6400, vocab 50257
    padded to 50304; attention matrices f8e5m2, MLP matrices RTN 4-bit, tied bf16 wte; batch 8 x 1024 random ids.  forward (every activation kept) + fused classifier loss ->
    backward through every operator into PER-TENSOR gradient buffers -> kf_adamw on the model's own bf16 masters and moments (seeded stochastic rounding) -> kf_quantize of every
    matrix back into the blob the next forward reads.  The step is sequenced by the host library (koifish::GPT2Trainer, koifish_amd/host/kf_train.cpp): the timed region holds
    three C calls per step and no torch op (embedding gather + add = kf_embed_pos, q read out of the fused rows, zero fills = kf_memset / kf_memset2d).
    Synthetic data and weights: not comparable one-to-one with the reference's end-to-end training throughput (48.8 k tokens/s on an RTX 4090,
    cases/gpt2/1558M_F8_B80/F8_B80.info:2928-2951: a real run with a data loader); it says what the step's kernel path sustains."""
    import torch
    from koifish_amd import runtime as R
    from koifish_amd.train_step import GPT2Step
    ctx = R.Context(0)
    dev = ctx.device
    Cn, H, T, B, NL, V, Vp = 1600, 25, 1024, 8, 48, 50257, 50304
    N, hd = B * T, Cn // H
    st = GPT2Step(ctx, Cn, H, NL, V, Vp, B, T, seed=3)
    ids = torch.randint(0, V, (N,), device=dev, dtype=torch.int32)
    tgt = torch.randint(0, V, (N,), device=dev, dtype=torch.int32)
    hp = dict(lr=3e-4, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.1, seed=7)
    loss_hist = []
    st.step(ids, tgt, **hp)   # warm-up step (first-launch costs, scratch sizing)
    ctx.sync()
    loss_hist.append(float(st.losses.mean()))
    probe = st.blocks[0]["fc"]["p"].view(torch.int16)[:4096].clone()
    reps = 2
    ev = [ctx.event() for _ in range(4 * reps + 1)]
    ctx.record(ev[0])
    for r in range(reps):   # ONE region: whole steps back to back; the events inside only split the report
        st.forward(ids, tgt)
        ctx.record(ev[4 * r + 1])
        st.backward()
        ctx.record(ev[4 * r + 2])
        st.update(**hp)
        ctx.record(ev[4 * r + 3])
        ctx.record(ev[4 * r + 4])
    ctx.sync()
    ms = ctx.elapsed_ms(ev[0], ev[4 * reps]) / reps
    t_f = sum(ctx.elapsed_ms(ev[4 * r], ev[4 * r + 1]) for r in range(reps)) / reps
    t_b = sum(ctx.elapsed_ms(ev[4 * r + 1], ev[4 * r + 2]) for r in range(reps)) / reps
    t_a = sum(ctx.elapsed_ms(ev[4 * r + 2], ev[4 * r + 3]) for r in range(reps)) / reps
    st.forward(ids, tgt)
    ctx.sync()
    loss_hist.append(float(st.losses.mean()))
    loss = loss_hist[-1]
    moved = bool((st.blocks[0]["fc"]["p"].view(torch.int16)[:4096] != probe).any())
    n_par = st.n_params()
    cpu_leg = None
    try:   # "CPU fwd of 1 layer x 1 batch row only (extrapolated; stated as such)": plain fp32 torch on this host's cores, the same operator sequence
        torch.set_num_threads(_physical_cores())   # one thread per CORE (the default is one per hardware thread), as the other CPU legs
        xc = torch.randn(T, Cn)
        wq_, wp_, wf_, wp2_ = (torch.randn(3 * Cn, Cn) * 0.02, torch.randn(Cn, Cn) * 0.02, torch.randn(4 * Cn, Cn) * 0.02, torch.randn(Cn, 4 * Cn) * 0.02)

        def cpu_layer():
            h = torch.nn.functional.layer_norm(xc, (Cn,))
            q, k, v = (h @ wq_.t()).view(T, 3, H, hd).permute(1, 2, 0, 3)
            a_ = torch.nn.functional.scaled_dot_product_attention(q[None], k[None], v[None], is_causal=True)[0].permute(1, 0, 2).reshape(T, Cn)
            x2 = xc + a_ @ wp_.t()
            h2 = torch.nn.functional.layer_norm(x2, (Cn,))
            return x2 + torch.nn.functional.gelu(h2 @ wf_.t()) @ wp2_.t()
        cpu_layer()
        tc = []
        for _ in range(5):
            t1 = time.perf_counter()
            cpu_layer()
            tc.append(time.perf_counter() - t1)
        lay_ms = float(sorted(tc)[len(tc) // 2]) * 1e3
        cpu_leg = {"value": round(lay_ms, 2), "unit": "ms per layer forward of ONE batch row (1024 tokens)", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": "one GPT2-1558M layer (LayerNorm, QKV, causal attention, proj, LayerNorm, fc, GELU, proj2), fp32 torch on the host cores, median of 5",
                   "extrapolated_step_ms": round(lay_ms * NL * B * 3, 0), "extrapolation": "x 48 layers x 8 rows x 3 (forward + backward ~ 2 x forward); the head, the loss and AdamW not included: a lower bound, stated as such"}
    except Exception as e:
        cpu_leg = {"error": repr(e)[:200]}
    # flops of the step: 2 x (block matrices 12 C^2 x 48 + head V C) per token forward + causal attention (QK^T and PV, half the square), x 3 for forward + backward
    w_el = NL * 12 * Cn * Cn + Vp * Cn
    fwd = 2.0 * N * w_el + NL * 4.0 * Cn * (T * (T + 1) / 2) * B
    flops = 3.0 * fwd
    return {"workload": "GPT2-1558M (48 layers, n_embd 1600, 25 heads, ffn 6400, vocab 50257), hybrid f8e5m2 / 4-bit blocks, tied bf16 head, 8 x 1024 random tokens, every activation kept: "
                        "ONE timed region per step -- forward + loss, backward into per-tensor gradient buffers, AdamW on the model's own bf16 masters and moments, re-quantisation of "
                        "every matrix into the blob the next forward reads; sequenced in C++ (koifish::GPT2Trainer in libkf_host.so): no torch op in the timed region",
            "host_loop": "C++ (libkf_host.so: kfh_gpt2_forward / _backward / _update; kfh_gpt2_step is the three in one call)",
            "params_updated": moved, "one_timed_region": True, "parameters": int(n_par), "steps_timed": reps,
            "ms": round(ms, 2), "tokens_per_s": round(N / ms * 1e3, 1), "forward_loss_ms": round(t_f, 2), "backward_ms": round(t_b, 2), "adamw_requantise_ms": round(t_a, 2),
            "loss_after_1_and_%d_steps_on_one_batch" % (1 + reps): [round(v, 4) for v in loss_hist],
            "cpu_baseline": cpu_leg, "mean_loss": round(loss, 4), "flops": int(flops), "achieved_TFLOPs": round(flops / (ms * 1e-3) / 1e12, 1), "mfma_peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS,
            "mfma_frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "parity": "tests/test_gpu_train_step.py::test_gpt2_two_consecutive_steps_with_update: the same loop at toy size -- losses vs fp64, AdamW and re-quantisation vs the oracle bit for bit",
            "reference": "48.8 k tokens/s END-TO-END training on an RTX 4090 (cases/gpt2/1558M_F8_B80/F8_B80.info:2928-2951, BASELINE.md): a real run with data loading; here synthetic "
                         "data, the step's kernel path", "profile": "profiles/r05_config3_train_step_kernel_stats.csv"}


MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak of MI355X (MI355X_MICROARCH.md); the prefill GEMMs multiply bf16 fragments unpacked from 4-bit tiles


def tp_main(args, cfg, rank, world, dev):
    """BASELINE config 4: Qwen3-32B 4-bit greedy decode, tensor parallel over the GPUs of the node (TP = world; one process per GPU), or -- a
    side measurement on one GPU -- R ranks of one process in lock-step (--tp-virtual R).  Every rank draws each full tensor on its own GPU from the
    same seed and keeps its shard (koifish_amd.tp.TPPlan: q/k/v/gate/up by rows, o_proj/down_proj by columns, head by vocabulary rows).  The
    timed region is K decode steps ending at the last position of the 4096-token context, after a token-serial run-up, with teacher-forced
    ids for the first 128 positions.  `value` = tokens/s of the ONE sequence the node decodes (strong scaling: the model is split, not copied)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from koifish_amd import lib as L
    from koifish_amd import synth
    from koifish_amd import tp as TP
    from koifish_amd.runtime import Context
    virtual = world == 1
    R = args.tp_virtual if virtual else world
    if args.x_tp_layers > 0:   # test hook only (tests/test_gpu_bench_ranks.py): a cut model; the line says so
        cfg = dict(cfg, n_layer=args.x_tp_layers, vocab=args.x_tp_vocab if args.x_tp_vocab > 0 else cfg["vocab"])
    elif args.tp_layers > 0:                 # the side leg of the default run: a cut of the layer stack bounds its wall time; the line says so
        cfg = dict(cfg, n_layer=args.tp_layers)
    ctx = Context(dev)
    ffn_real = cfg["ffn"]
    if cfg["ffn"] % (128 * R) != 0:   # Qwen3-4B: 9728 = 76 groups of 128 do not split into R whole-group column shards -- the FFN padded with zero rows of gate / up and zero columns of down_proj
        cfg = dict(cfg, ffn=(cfg["ffn"] + 128 * R - 1) // (128 * R) * (128 * R))
    plan = TP.TPPlan(cfg, R)
    g = torch.Generator(device=ctx.device)
    g.manual_seed(1234)

    def mat(r, c):
        t = (torch.randn(r, c, generator=g, device=ctx.device, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        if r == cfg["ffn"] and ffn_real < r:
            t[ffn_real:] = 0
        if c == cfg["ffn"] and ffn_real < c:
            t[:, ffn_real:] = 0
        return t

    def nrm(n):
        return (1.0 + 0.01 * torch.randn(n, generator=g, device=ctx.device, dtype=torch.float32)).to(torch.bfloat16)

    ranks = range(R) if virtual else [rank]
    shards = {r: {} for r in ranks}
    norms = {}
    emb = ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    head = emb if cfg.get("tied", True) else ctx.quantize(mat(cfg["vocab"], cfg["dim"]), L.BF16)
    for r in ranks:
        shards[r][(-1, 0)] = emb
        shards[r][(-1, 1)] = TP.shard_rows(head, *plan.head_rows(r))
    norms[(-1, 0)] = nrm(cfg["dim"])
    for li in range(cfg["n_layer"]):
        for si, s in enumerate(synth.SLOTS):
            full = ctx.quantize(mat(*synth.SHAPES[s](cfg)), L.Q4)
            for r in ranks:
                shards[r][(li, si)] = plan.shard(s, full, r)
            del full
        norms[(li, 0)], norms[(li, 1)], norms[(li, 2)], norms[(li, 3)] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(cfg["head_dim"]), nrm(cfg["head_dim"])
    if head is not emb:
        del head
    torch.cuda.empty_cache()
    S, K, W = cfg["max_seq"], args.steps, args.warmup
    if W + K > S:
        raise SystemExit("warmup + steps must fit the %d-token context of the TP run" % S)
    forced = np.full(S, -1, dtype=np.int32)
    forced[:128] = np.random.default_rng(7).integers(0, cfg["vocab"], size=128)
    use_graph = not args.no_graph
    bytes_rank = sum(x.algorithmic_bytes() for k, x in shards[ranks[0]].items() if k != (-1, 0)) + cfg["dim"] * 2
    if args.tp_exchange == "rccl" and not virtual:
        drv = TP.DistributedTP(TP.TPRank(plan, rank, ctx, shards[rank], norms))
        tok, state = int(forced[0]), {"pos": 0}

        def run(p0, n):
            nonlocal tok
            for p in range(p0, p0 + n):
                nxt = drv.step(tok, p)
                tok = int(forced[p + 1]) if p + 1 < S and forced[p + 1] >= 0 else nxt
        check = lambda: None
        path = "Python-stepped, 2 torch.distributed all_gather_into_tensor (RCCL) per layer + 1 per token, host pick"
    else:
        if virtual:
            class _G:   # NativeTP wants the full weights; here the shards exist already
                pass
            nt = TP.NativeTP.__new__(TP.NativeTP)
            nt.cfg, nt.ctx, nt.world, nt.plan = cfg, ctx, R, plan
            nt.ranks = [TP.build_native_rank(cfg, plan, r, shards[r], norms, dev) for r in range(R)]
            nt.host = nt.ranks[0].host
            import ctypes as C
            for a in nt.ranks:
                for r, b in enumerate(nt.ranks):
                    L.check(nt.host.kfh_tp_set_peer(a.h, r, C.c_void_p(nt.host.kfh_tp_area(b.h))), "kfh_tp_set_peer")
            nt._hs = (C.c_void_p * R)(*[m.h for m in nt.ranks])
            drv = nt
            if args.tp_xcd == 1:
                print(json.dumps(xcd_tp_leg(nt, cfg, ctx, forced, S, K, W, shards, plan, args.jump)))
                return
        else:
            drv = TP.NativeRank(cfg, plan, rank, shards[rank], norms, dev)
        drv.set_forced(forced)
        drv.set_state(int(forced[0]), 0)
        run = lambda p0, n: drv.run_steps(p0, n, use_graph)
        check = drv.check
        path = "C++ host, one hipGraph per position bucket: per layer 2 x [mat-vec whose epilogue stores fp32 partials into every rank's peer-mapped receive area, " \
               "rank-ordered sum kernel], arg-max pairs the same way; no collective call, no host round trip"
    start = S - (W + K)
    if args.jump and virtual:   # side leg: synthetic K / V rows in front of the window, no token-serial run-up
        import ctypes as C
        kvb = cfg["n_layer"] * S * plan.kvd_l * 2
        for rk in drv.ranks:
            _fill_kv_synthetic(rk.hip, C.c_void_p(rk.host.kfh_ctx(rk.h)), [(rk.host.kfh_kcache(rk.h), kvb), (rk.host.kfh_vcache(rk.h), kvb)])
        drv.set_state(1, start)
    else:
        run(0, start)
    run(start, W)
    torch.cuda.synchronize()
    if not virtual:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    ctx.record(e0)
    run(start + W, K)
    ctx.record(e1)
    torch.cuda.synchronize()
    if not virtual:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    check()
    if not virtual:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        ms = dt * 1e3 / K
        kvd_l = plan.kvd_l
        mean_pos = S - K / 2.0
        step_bytes = bytes_rank + 2 * cfg["n_layer"] * mean_pos * kvd_l * 2
        ach = step_bytes / (ms * 1e-3) / 1e9 * (R if virtual else 1)
        out = {"metric": "tokens/sec/GPU Qwen3 4-bit prefill+decode; achieved HBM GB/s vs peak", "value": round(K / dt, 3), "unit": "tokens/s", "n_gpus": world, "steps": K,
               "warmup": W, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u4 weights x bf16, fp32 accumulate", "data": "synthetic",
               "config": {"workload": "Qwen3-32B 4-bit PackedQ greedy decode, tensor parallel TP=%d%s, context %d: timed positions %d..%d" % (
                   R, " (all ranks on ONE GPU, lock-step: a side measurement, not a scaling point)" if virtual else " over %d MI355X" % world, S, S - K, S - 1),
                   "layers": cfg["n_layer"], "vocab": cfg["vocab"], "tp": R, "exchange": args.tp_exchange if not virtual else "p2p (local pointers)", "decode_path": path, "hipgraph": use_graph,
                   "ranks_in_process_group": (dist.get_world_size() if dist.is_initialized() else 1), "process_group_backend": (dist.get_backend() if dist.is_initialized() else None),
                   "device_ms_per_step": round(ctx.elapsed_ms(e0, e1) / K, 4), "weight_bytes_per_rank": int(bytes_rank)},
               "roofline": {"bound": "hbm", "kernel": "one rank's decode step (its weight shards + its KV heads)", "bytes_per_step_per_rank": int(step_bytes),
                            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                            "note": "per GPU; %s" % ("R ranks share this GPU, so the per-rank rate is the step's bytes x R / time" if virtual else "every GPU streams its own shard")},
               "scaling_curve": "none measured on hardware yet: this container's GPU boxes have one MI355X; the driver's N = 2, 4, 8 runs of this command produce it"}
        if virtual:
            if args.tp_xcd == 2:   # the same ranks, the same shards: the one-launch form (the eight ranks as the eight XCDs) timed in this process too
                try:
                    out["xcd"] = xcd_tp_leg(drv, cfg, ctx, forced, S, K, W, shards, plan, args.jump)
                except Exception as e:
                    out["xcd"] = {"error": repr(e)[:300]}
            print(json.dumps(out))   # a side leg's child: the parent takes what it needs
        else:
            emit(out, "bench_detail_tp.json")
    if not virtual and dist.is_initialized():
        dist.destroy_process_group()


def xcd_tp_leg(nt, cfg, ctx, forced, S, K, W, shards, plan, jump=False):
    """Qwen3-32B on ONE MI355X as tensor parallel over the XCDs: rank r of the TP = 8 plan on XCD r, all of them in ONE launch (koifish::XcdTP, kf_xengine_create_tp) -- the
    o_proj / down_proj partials exchanged between the XCDs inside the kernel.  Timed: the last K positions of the context (after a token-serial run-up by this engine itself,
    or -- jump -- on synthetic K / V rows).  Bits: those of the TP = 8 rank step (tests/test_gpu_tp.py::test_tp8_ranks_as_the_eight_xcds_of_one_launch_vs_the_oracle); here the
    first 24 ids from position 0 are compared with the per-launch rank step's."""
    import numpy as np
    import torch
    from koifish_amd.runtime import XcdTP
    for rk in nt.ranks:
        rk.set_canonical(True)
    xt = XcdTP(nt)
    xt.set_forced(forced)
    xt.set_state(int(forced[0]), 0)
    start = S - (W + K)
    if jump:
        import ctypes as C
        xt.run_steps(24)
        xt.check()
        ids_x = xt.tokens_out(24)
        kvb = len(nt.ranks) * cfg["n_layer"] * S * plan.kvd_l * 2
        r0 = nt.ranks[0]
        _fill_kv_synthetic(r0.hip, C.c_void_p(r0.host.kfh_ctx(r0.h)), [(xt.host.kfh_xtp_kcache(xt.h), kvb), (xt.host.kfh_xtp_vcache(xt.h), kvb)])
        xt.set_forced(np.full(S, -1, dtype=np.int32))
        xt.set_state(1, start)
    else:
        xt.run_steps(start)
    xt.run_steps(W)
    torch.cuda.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    ctx.record(e0)
    xt.run_steps(K)
    ctx.record(e1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    xt.check()
    if not jump:
        ids_x = xt.tokens_out(S)
    nt.set_forced(forced)
    nt.set_state(int(forced[0]), 0)
    nt.run_steps(0, 24, True)
    nt.check()
    same = bool(np.array_equal(nt.ranks[0].tokens_out(24), ids_x[:24]))
    ms = dt * 1e3 / K
    mean_pos = S - K / 2.0
    weights = sum(x.algorithmic_bytes() for r in shards for k, x in shards[r].items() if k != (-1, 0)) + cfg["dim"] * 2
    step_bytes = weights + 2 * cfg["n_layer"] * mean_pos * cfg["n_kv"] * cfg["head_dim"] * 2
    ach = step_bytes / (ms * 1e-3) / 1e9
    res = {
        "workload": "%s (dim %d, ffn %d) 4-bit PackedQ greedy decode of ONE sequence on ONE MI355X, the TP = 8 ranks as the eight XCDs of one launch, context %d: timed positions %d..%d%s" % (
            {5120: "Qwen3-32B", 4096: "Qwen3-8B", 2560: "Qwen3-4B (FFN padded to 10240 with zero rows)"}.get(cfg["dim"], "model"), cfg["dim"], cfg["ffn"], S, S - K, S - 1, "; K / V rows in front of the window synthetic" if jump else ""),
        "tokens_per_s": round(K / dt, 2), "ms_per_step": round(ms, 4), "device_ms_per_step": round(ctx.elapsed_ms(e0, e1) / K, 4), "steps": K, "layers": cfg["n_layer"], "vocab": cfg["vocab"],
        "bytes_per_step": int(step_bytes), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
        "kernel": "kf::xengine_kernel<XCfg<..., TP>> (koifish_amd/csrc/kf_xengine.hip): per layer four hand-offs inside each XCD + two exchanges between them",
        "summation_order": "canonical, tensor parallel TP = 8 (column shards as fp32 partials summed in rank order: the bits an 8-GPU node computes)",
        "parity": {"first_24_ids_equal_per_launch_rank_step": same, "oracle": "tests/test_gpu_tp.py::test_tp8_ranks_as_the_eight_xcds_of_one_launch_vs_the_oracle (ids, logits, K / V rows bit for bit)"}}
    xt.close()
    return res


def prefill_rate(m, prompt, decode_ms_per_step, reps=5, bound=None):
    """The prompt half of the metric, reported beside the decode rate (never inside `value`): the 128-token prompt through
    Fish::Prefill (token batches on the MFMA tile kernels) -- wall time of `reps` calls after one warm-up, each ending with the
    head + pick of the first generated token.  The reference prefills token by token through the decode path (GoPT.cpp:1139-1146),
    which here costs one decode step per prompt token."""
    import time
    m.sync()
    t0 = time.perf_counter()
    m.prefill(prompt, want_logits=False)  # the warm-up; the first prompt of >= 320 tokens also fills the resident bf16 copies (kf_set_dequant_arena), reported as first_call_ms
    m.sync()
    first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        m.prefill(prompt, want_logits=False)
    m.sync()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    n = int(len(prompt))
    cfg = m.cfg
    # both roofs of the prefill: flops = 2 x tokens x (layer matrix elements) + causal attention (QK^T and PV over t <= s) + the head mat-vec of the last token;
    # bytes = every packed weight once (layers + the embedding rows touched + the LM head) + K/V rows written and read once
    w_elems = sum(w.ne0 * w.ne1 for (layer, slot), w in m.weights.items() if layer >= 0)
    flops = 2.0 * n * w_elems + cfg["n_layer"] * 4.0 * cfg["n_head"] * cfg["head_dim"] * (n * (n + 1) / 2) + 2.0 * cfg["vocab"] * cfg["dim"]
    kvd = cfg["n_kv"] * cfg["head_dim"]
    nbytes = sum(w.algorithmic_bytes() for (layer, slot), w in m.weights.items() if layer >= 0) + m.weights[(-1, 1)].algorithmic_bytes() + n * cfg["dim"] * 2 \
        + cfg["n_layer"] * n * kvd * 2 * 2 * 2
    tf, gbs = flops / (ms * 1e-3) / 1e12, nbytes / (ms * 1e-3) / 1e9
    return {"prompt_tokens": n, "ms": round(ms, 3), "tokens_per_s": round(n / ms * 1e3, 1), "mode": "token batches: < 320 rows MFMA 32x32x16 bf16 on 4-bit tiles unpacked in registers; >= 320 rows the 256x256 / 128x128 bf16 tile kernels (MFMA 16x16x32; SwiGLU and q/k-norm + RoPE in their epilogues) on RESIDENT bf16 copies of the layer matrices (dequantised by the first such prompt, kept in HBM); flash attention tile",
            "first_call_ms": round(first_ms, 3), "resident_copy_bytes": m.resident_bytes(),
            "token_serial_ms": round(decode_ms_per_step * n, 3),
            "roofline": {"flops": int(flops), "bytes": int(nbytes), "achieved_TFLOPs": round(tf, 2), "mfma_peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS, "mfma_frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                         "achieved_GBs": round(gbs, 1), "hbm_peak_GBs": HBM_PEAK_GBS, "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                         "bound": bound or "neither: %d dependent launches of ~10 us each (8 per layer); at this size the prompt is bound by launch + first-load latency" % (8 * cfg["n_layer"] + 3)}}


def concurrent_streams(cfg, layer_type, head_type, dev, S, forced, n_prompt, mean_bytes):
    """What the chip sustains once launch latency is overlapped: S independent single-stream decoders -- each with its OWN copy of the weights
    (so nothing is shared through the caches: S x 545 MB >> the 256 MB Infinity Cache), its own KV cache and its own HIP stream -- decode the
    same positions 128..2047 concurrently; one host thread keeps every stream's queue of step graphs filled.  Aggregate tokens/s and the HBM
    fraction it implies.  The single-stream number above stays the benchmark's `value`; this shows how much of the gap to the HBM roofline is
    per-launch latency rather than kernel throughput."""
    import time
    import numpy as np
    import torch
    from koifish_amd import synth
    S_len = cfg["max_seq"]
    models = []
    for i in range(S):
        mm = synth.build_on_gpu(cfg, seed=4321 + i, layer_type=layer_type, head_type=head_type, device=dev, own_stream=True)
        mm.set_engine(False)  # the persistent engine needs the CUs to itself; concurrent decoders use the per-layer launches
        f = forced.copy()
        f[:n_prompt] = np.random.default_rng(100 + i).integers(0, cfg["vocab"], size=n_prompt)
        mm.set_forced(f)
        mm.set_state(int(f[0]), 0)
        models.append(mm)
    torch.cuda.synchronize()
    chunk = 32
    for p in range(0, n_prompt, chunk):          # the prompts (untimed; also captures the graphs of the first buckets)
        for mm in models:
            mm.run_steps(p, min(chunk, n_prompt - p), True)
    torch.cuda.synchronize()
    K = S_len - n_prompt
    # one host thread per decoder: enqueueing a 145-node step graph costs the host ~0.4 ms, so a single thread tops out near 2.2 k graphs/s
    # (ctypes drops the GIL during the call; each thread drives its own stream)
    import threading
    threads = [threading.Thread(target=lambda mm=mm: mm.run_steps(n_prompt, K, True)) for mm in models]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tps = S * K / dt
    for mm in models:
        mm.close()
    return {"streams": S, "steps_per_stream": K, "tokens_per_s": round(tps, 1), "ms_per_step_per_stream": round(dt * 1e3 / K, 4),
            "hbm_achieved_GBs": round(mean_bytes * tps / 1e9, 1), "hbm_frac": round(mean_bytes * tps / 1e9 / HBM_PEAK_GBS, 4),
            "note": "independent decoders, separate weight copies and HIP streams, positions %d..%d each" % (n_prompt, S_len - 1)}


def xcd_replicas(m, cfg, forced, timed_positions, warmup, n_seq, ids_main, jump=False, queue=False):
    """The chip's aggregate rate on INDEPENDENT sequences (never `value`, which stays the single-sequence rate): n_seq decoders inside one launch, one per XCD (32 workgroups
    each, every hand-off in that XCD's L2: koifish_amd/csrc/kf_xengine.hip), sharing this model's weights; own K / V cache, state, prompt and logits per sequence.  The
    reference decodes one sequence per process (GoPT.cpp:1139-1180) and scales a small model with more processes -- these are the processes, moved inside the package.
    Every sequence is decoded token by token from position 0 through its own 128-token prompt to the end of the context by the replicas' engine itself; the timed span is the
    positions of `value`.  Sequence 0 carries the main run's prompt: its ids at EVERY position must equal the single-sequence engine's (whose ids and logits the cpu_baseline
    leg checks against the oracle bit for bit); tests/test_gpu_xengine.py holds the per-sequence oracle parity."""
    import numpy as np
    import torch
    from koifish_amd.runtime import XcdReplicas
    S = cfg["max_seq"]
    K = len(timed_positions)
    first = timed_positions[0]
    traffic, traffic_src, valu = None, None, None
    try:   # counter passes of scratch/gpu_r05_profile.sh / gpu_r06_pmc.sh (not re-collected by this run: --pmc around bench.py itself crashes the profiler on this pool)
        if cfg["dim"] != 1024:
            raise KeyError("the passes were taken on Qwen3-0.6B")
        ns = 32 if n_seq > 16 else (16 if n_seq > 8 else 8)
        pf = "r06_pmc_xengine_%d.json" % ns if ns > 8 else "r05_pmc_xengine_8.json"
        pj = json.load(open(os.path.join(ROOT, "profiles", pf)))
        traffic = int(pj["hbm_bytes_per_launch"] / (ns * 4))
        traffic_src = "profiles/%s: FETCH_SIZE / WRITE_SIZE passes of scratch/ub_xengine.py (4 steps per launch at positions 2037..2040), per sequence and step" % pf
        valu = int(json.load(open(os.path.join(ROOT, "profiles", pf.replace(".json", "_sq.json"))))["valu_instructions_per_sequence_step"])
    except Exception:
        pass
    if first + K != S or first - warmup < 1:
        return {"skipped": "the timed window does not end at the last position of the context"}
    xr = XcdReplicas(m, n_seq)
    try:
        for s in range(n_seq):
            f = forced.copy()
            if s > 0:
                f[:128] = np.random.default_rng(200 + s).integers(0, cfg["vocab"], size=128)
            xr.set_forced(s, f)
            xr.set_state(s, int(f[0]), 0)
        if jump:   # side shapes: synthetic K / V rows in front of the window instead of a token-serial run-up (the 0.6B leg of the default run keeps the run-up: its sequence 0 is compared id for id)
            import ctypes as C
            kvb = cfg["n_layer"] * S * cfg["n_kv"] * cfg["head_dim"] * 2
            _fill_kv_synthetic(m.hip, C.c_void_p(m.host.kfh_ctx(m.h)), [(f(xr.h, s), kvb) for s in range(n_seq) for f in (xr.host.kfh_xr_kcache, xr.host.kfh_xr_vcache)])
            for s in range(n_seq):
                xr.set_state(s, 1 + s, first - warmup)
        else:
            xr.run_steps(first - warmup)     # set-up: every sequence's own history
        xr.run_steps(warmup)
        torch.cuda.synchronize()
        ctx = m._ctx
        e0, e1 = ctx.event(), ctx.event()
        t0 = time.perf_counter()
        ctx.record(e0)
        xr.run_steps(K)
        ctx.record(e1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        xr.check()
        dev_ms = ctx.elapsed_ms(e0, e1)
        ids0 = xr.tokens_out(0, S)
        same = None if jump else bool(np.array_equal(ids0, ids_main))
        distinct = len({tuple(xr.tokens_out(s, S)[128:160].tolist()) for s in range(n_seq)})
        tps = n_seq * K / dt
        bytes_tok = float(np.mean([m.step_bytes(p) for p in timed_positions]))
        prompt_leg = None
        try:   # the prompt half for the replicas (after every check above: it overwrites the sequences' rows): each sequence's 128-token prompt through the model's batched prefill, then 32 tokens of all
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s in range(n_seq):
                pr = forced[:128] if s == 0 else np.random.default_rng(200 + s).integers(0, cfg["vocab"], size=128).astype(np.int32)
                xr.prefill(s, pr)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for s in range(n_seq):
                xr.set_forced(s, np.full(S, -1, dtype=np.int32))
            xr.run_steps(32)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            xr.check()
            prompt_leg = {"prompt_tokens": 128, "prefill_ms_per_sequence": round((t2 - t1) * 1e3 / n_seq, 3), "then_32_tokens_of_every_sequence_ms": round((t3 - t2) * 1e3, 3),
                          "prompt_plus_32_tokens_per_s": round(n_seq * (128 + 32) / (t3 - t1), 1),
                          "parity": "tests/test_gpu_xengine.py::test_prefill_then_decode_per_sequence (rows, ids, logits == the model alone doing prefill + decode)"}
        except Exception as e:
            prompt_leg = {"error": repr(e)[:200]}
        chat_leg = None
        try:   # (the primary leg only) a queue of requests through the slots (XcdReplicas.chat: Fish::Chat's rounds over a prompt list, n_seq in flight): 3 x n_seq requests, 128-token prompt, 128 new ids each
            if not queue:
                raise KeyError("not asked for")
            n_req, rng_q = 3 * n_seq, np.random.default_rng(31)
            prompts_q = [rng_q.integers(0, cfg["vocab"], size=128).astype(np.int32) for _ in range(n_req)]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ans, qst = xr.chat(prompts_q, 128)
            t2 = time.perf_counter()
            chat_leg = {"requests": n_req, "prompt_tokens": 128, "new_tokens": 128, "seconds": round(t2 - t1, 4), "requests_per_s": round(n_req / (t2 - t1), 1),
                        "generated_tokens_per_s": round(sum(len(a) for a in ans) / (t2 - t1), 1), "launches": qst["launches"], "prefills": qst["prefills"],
                        "note": "prefill of every request included; positions 128..255 (short context: a step moves fewer K / V rows than at the 2 k window the headline is quoted on)",
                        "parity": "tests/test_gpu_xengine.py::test_a_queue_of_prompts_through_the_slots (every answer == the model alone on that prompt; EOS cut)"}
            # the same queue with the waiting prompts prefilled TOGETHER (XcdReplicas::PrefillBatch: one token batch of 16 x 128 rows on the tile kernels per refill)
            pb = min(16, n_seq)
            xr.set_prefill_batch(pb)
            xr.prefill_batch(list(range(pb)), prompts_q[:pb])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                xr.prefill_batch(list(range(pb)), prompts_q[:pb])
            torch.cuda.synchronize()
            t_pb = (time.perf_counter() - t1) / 3
            t1 = time.perf_counter()
            ans, qst = xr.chat(prompts_q, 128)
            t2 = time.perf_counter()
            xr.set_prefill_batch(1)
            chat_leg["prefill_batch"] = {"prompts_per_batch": pb, "batch_ms": round(t_pb * 1e3, 3), "ms_per_prompt": round(t_pb * 1e3 / pb, 3), "seconds": round(t2 - t1, 4),
                                         "requests_per_s": round(n_req / (t2 - t1), 1), "generated_tokens_per_s": round(sum(len(a) for a in ans) / (t2 - t1), 1),
                                         "parity": "tests/test_gpu_xengine.py::test_prefill_batch_vs_the_oracle (K / V rows, logits within 2^-6 of scale: token batches sum in MFMA order)"}
        except KeyError:
            chat_leg = None
        except Exception as e:
            chat_leg = {"error": repr(e)[:200]}
        # two readings, neither of them `roofline.frac`: (1) aggregate tokens/s against the rate ONE sequence's algorithmic bytes allow at the HBM peak (what north_star's 0.70 is quoted
        # on: 0.70 <-> 7.2 k tokens/s here); (2) HBM utilisation in BATCH-AWARE bytes -- the decoders share the layer weights and the head (counted once per step of all
        # sequences), each adds only its own K / V rows and vectors
        kv_tok = float(np.mean([2.0 * cfg["n_layer"] * (p + 1.5) * cfg["n_kv"] * cfg["head_dim"] * 2 for p in timed_positions]))
        bytes_batch = bytes_tok + (n_seq - 1) * kv_tok
        return {"streams": n_seq, "batch": getattr(xr, "batch", 1), "decoders_per_xcd": getattr(xr, "decoders_per_xcd", 2 if n_seq > 8 else 1),
                "layout": "32 workgroups of one launch per decoder (one decoder per XCD), every hand-off inside that XCD's L2; a decoder's sequences share every unpacked 4-bit block",
                "tokens_per_s": round(tps, 1), "per_stream_tokens_per_s": round(tps / n_seq, 1),
                "ms_per_step_all_streams": round(dt * 1e3 / K, 4), "device_ms_per_step": round(dev_ms / K, 4), "steps": K, "positions": "%d..%d" % (first, S - 1),
                "bytes_per_token_one_sequence": int(bytes_tok), "single_sequence_roofline_tokens_per_s": round(HBM_PEAK_GBS * 1e9 / bytes_tok, 1),
                "frac_vs_single_sequence_roofline": round(bytes_tok * tps / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_per_step_batch_aware": int(bytes_batch), "hbm_GBs_batch": round(bytes_batch * (tps / n_seq) / 1e9, 1), "hbm_frac_batch": round(bytes_batch * (tps / n_seq) / 1e9 / HBM_PEAK_GBS, 4),
                "aggregate_of_independent_sequences": True,
                "note": "frac_vs_single_sequence_roofline = ONE sequence's algorithmic bytes x aggregate tokens/s / 8 TB/s: a RATE ratio, not HBM utilisation (the decoders share the weights and the "
                        "head: L2 / memory-side-cache traffic); hbm_frac_batch counts the shared bytes once per step of all sequences; never `value`",
                "traffic_per_sequence_step": traffic, "traffic_source": traffic_src, "valu_insts_per_sequence_step": valu, "prefill_then_decode": prompt_leg, "request_queue": chat_leg,
                "summation_order": "canonical (the only order the XCD-confined engines run)", "kernel": "kf::xengine_kernel (koifish_amd/csrc/kf_xengine.hip)",
                "parity": {"sequence_0_ids_equal_single_sequence_engine": same, "positions_compared": int(S), "distinct_continuations": distinct,
                           "per_sequence_oracle_parity": "tests/test_gpu_xengine.py (ids, logits, K / V rows of every sequence, bit for bit)"}}
    finally:
        xr.close()


def kernel_roofline(m, ctx, cfg, reps=200):
    """The dominant kernel of the step by bytes and by time: the LM-head mat-vec (kf::gemv_kernel<BF16, ., ARGMAX>, 311 MB of
    bf16 rows read once).  Timed alone with HIP events on the launch stream; achieved = algorithmic bytes / mean duration."""
    import ctypes as C
    import torch
    from koifish_amd import lib as L
    head = m.weights[(-1, 1)]
    x = torch.randn(cfg["dim"], device=ctx.device).to(torch.bfloat16)
    nw = torch.ones(cfg["dim"], device=ctx.device, dtype=torch.bfloat16)
    logits = torch.empty(head.ne0, dtype=torch.bfloat16, device=ctx.device)
    state = torch.zeros(4, dtype=torch.int32, device=ctx.device)
    d = head.desc()

    def launch():   # the head mat-vec alone (no arg-max finish launch): what rocprofv3 lists as kf::gemv_kernel<0, 4, 2>
        L.check(ctx.hip.kf_lm_head(ctx.h, C.byref(d), x.data_ptr(), logits.data_ptr(), None, ctx._head_ws.data_ptr()), "kf_lm_head")
    for _ in range(10):
        launch()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        launch()
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1) / reps
    nbytes = head.algorithmic_bytes() + cfg["dim"] * 2 + head.ne0 * 2  # weights + x + logits written
    ach = nbytes / (ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this process): FETCH_SIZE x 2
    # (gfx950 correction) + WRITE_SIZE, collected by `rocprofv3 --pmc ... -- python3 scratch/ub_head.py`
    traffic = None
    if (head.ne0, head.ne1, head.type) == (151936, 1024, L.BF16):   # the PMC passes were taken on this shape
        try:
            traffic = int(json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_lm_head.json")))["hbm_bytes_per_launch"])
        except Exception:
            pass
    return {"bound": "hbm", "kernel": "kf::gemv_kernel<%d, 4, 2> = LM head %dx%d mat-vec + per-workgroup arg-max partials"
                                      % ({L.BF16: 0, L.F8E5M2: 1, L.Q4: 2}.get(head.type, 0), head.ne0, head.ne1), "achieved": round(ach, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "bytes_per_launch": int(nbytes),
            "us_per_launch": round(ms * 1e3, 2)}


def engine_roofline(m, ctx, cfg, forced, timed_positions, reps=6):
    """The kernel that IS the decode step when the persistent engine serves the model: kf::engine_kernel, ONE launch per token -- the embedding row, all layers
    (RMSNorm, Q/K/V, q/k-norm + RoPE + attention, o_proj, gate/up + SwiGLU, down_proj), the final norm, the LM head and the greedy pick.  Timed with HIP events
    on the launch stream around single eager launches at positions spread over the timed region (the KV cache still holds the rows of the run; consecutive
    launches stream 545 MB + KV each, so nothing comes from a cache).  achieved = algorithmic bytes of a launch (packed weights + zero / step
    of every layer matrix, the norm vectors, the bf16 head, K/V rows 0..pos read and row pos written) / mean duration.  traffic = HBM bytes per launch from the
    rocprofv3 --pmc passes committed under profiles/ (collected on the same launch at the position named there; rocprofv3 cannot run inside this process)."""
    pos_list = sorted(set(timed_positions[:: max(1, len(timed_positions) // 8)]))
    tot_ms, tot_bytes, n = 0.0, 0.0, 0
    for p in pos_list:
        if p + 1 >= cfg["max_seq"]:
            continue
        m.set_state(int(forced[p]) if forced[p] >= 0 else 1, p)
        m.run_steps(p, 1, False)   # warm: the eager path of this position
        for r in range(reps):
            m.set_state(int(forced[p]) if forced[p] >= 0 else 1, p)
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            m.run_steps(p, 1, False)
            ctx.record(e1)
            m.sync()
            tot_ms += ctx.elapsed_ms(e0, e1)
            tot_bytes += m.step_bytes(p)
            n += 1
    if n == 0 or m.engine_steps() <= 0:
        return None
    ms, nbytes = tot_ms / n, tot_bytes / n
    ach = nbytes / (ms * 1e-3) / 1e9
    # what the timed region launches: runs of up to 16 steps in ONE engine_kernel launch (kf_engine_steps_head).  In a rocprofv3 --kernel-trace of this command the kernel's
    # MinNs is a single-step launch and its MaxNs a 16-step run; AverageNs mixes both.
    run = None
    try:
        p0 = timed_positions[0]
        R = min(16, len(timed_positions), cfg["max_seq"] - 1 - p0)
        if R >= 2:
            rms = 0.0
            for r in range(4):
                m.set_state(int(forced[p0]) if forced[p0] >= 0 else 1, p0)
                e0, e1 = ctx.event(), ctx.event()
                ctx.record(e0)
                m.run_steps(p0, R, True)
                ctx.record(e1)
                m.sync()
                if r > 0:
                    rms += ctx.elapsed_ms(e0, e1) / 3
            rbytes = float(sum(m.step_bytes(p0 + i) for i in range(R)))
            rach = rbytes / (rms * 1e-3) / 1e9
            run = {"steps_per_launch": R, "positions": [p0, p0 + R - 1], "us_per_launch": round(rms * 1e3, 1), "us_per_step": round(rms * 1e3 / R, 2), "bytes_per_launch": int(rbytes),
                   "achieved": round(rach, 1), "frac": round(rach / HBM_PEAK_GBS, 4)}
    except Exception as e:
        run = {"error": repr(e)[:160]}
    traffic, traffic_src = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_engine.json")))
        traffic = int(pj["hbm_bytes_per_launch"])
        traffic_src = "profiles/r04_pmc_engine.json: counter passes of scratch/ub_engine.py at position %d (%d algorithmic bytes there); not re-collected by this run" % (
            int(pj.get("position", -1)), int(pj["algorithmic_bytes_per_launch"]))
    except Exception:
        pass
    return {"bound": "hbm", "kernel": "kf::engine_kernel = one decode step in one persistent launch: embedding row + %d layers + final norm + LM head + greedy pick (256 workgroups, "
                                      "hand-offs through tagged granules)" % cfg["n_layer"],
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "bytes_per_launch": int(nbytes), "us_per_launch": round(ms * 1e3, 2), "launches": 1, "positions": pos_list, "run_of_steps": run,
            "note": "latency-bound: %d layers x 6 dependent hand-offs (4 cross the XCDs: ~1.3 us each with the first sweep timed behind the own publish, scratch/ub_handoff3.hip) + "
                    "~5 us of phase arithmetic per layer; the head's %.0f MB stream at ~6 TB/s inside the same launch.  Timed here as single-step launches; the timed region of `value` "
                    "launches runs of up to 16 steps (kf_engine_steps_head), which saves the launch boundary per step" % (cfg["n_layer"], m.weights[(-1, 1)].algorithmic_bytes() / 1e6)}


def matvec_roofline(m, ctx, cfg, head_rl, reps=20):
    """The kernel that takes most of the step's time (47 % in profiles/r01g; 49 % in r01l): kf::gemv_kernel<5, 1, 0> (<2, 1, 0> before the register-table
    form of the same arithmetic became the default; KF_Q4_PERM=0 still selects it), the 4-bit mat-vec behind
    [RMSNorm + Q/K/V], [o_proj + residual] and [down_proj + residual] -- 3 launches per layer, 84 per token.  All 84 launches with the
    decode step's own weights and arguments are captured in one graph (a dependent chain, like the step) and replayed; an untimed LM-head
    launch between replays pushes the layer weights out of L2 / Infinity Cache as the real step does.  achieved = algorithmic bytes of the
    84 launches / their summed duration (HIP events on the launch stream); `us_per_launch` is what rocprofv3 lists as the kernel's average."""
    import ctypes as C
    import torch
    from koifish_amd import lib as L
    dim, nL = cfg["dim"], cfg["n_layer"]
    dev = ctx.device
    x = torch.randn(dim, device=dev).to(torch.bfloat16)
    y = torch.zeros(dim, dtype=torch.bfloat16, device=dev)
    keep, launches, nbytes = [], [], 0
    for l in range(nL):
        ws = [m.weights[(l, s)] for s in range(7)]   # q k v o gate up down
        qkv = ws[:3]
        descs = [w.desc() for w in qkv]
        outs = [torch.zeros(w.ne0, dtype=torch.bfloat16, device=dev) for w in qkv]
        wp = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
        yp = (C.c_void_p * 3)(*[o.data_ptr() for o in outs])
        nw = m._norms[(l, 0)]
        att = torch.randn(ws[3].ne1, device=dev).to(torch.bfloat16)
        act = torch.randn(ws[6].ne1, device=dev).to(torch.bfloat16)
        do, dd = ws[3].desc(), ws[6].desc()
        keep += [descs, outs, wp, yp, att, act, do, dd]
        launches.append(lambda wp=wp, yp=yp, nw=nw: L.check(ctx.hip.kf_norm_linear(ctx.h, x.data_ptr(), nw.data_ptr(), 1e-6, 3, wp, yp, None, 0, None), "kf_norm_linear"))
        launches.append(lambda do=do, att=att: L.check(ctx.hip.kf_linear(ctx.h, C.byref(do), att.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 1, x.data_ptr()), "kf_linear"))
        launches.append(lambda dd=dd, act=act: L.check(ctx.hip.kf_linear(ctx.h, C.byref(dd), act.data_ptr(), y.data_ptr(), None, 1, 1.0, 0.0, 1, x.data_ptr()), "kf_linear"))
        nbytes += sum(w.algorithmic_bytes() for w in qkv) + ws[3].algorithmic_bytes() + ws[6].algorithmic_bytes()
        nbytes += 2 * (dim + ws[3].ne1 + ws[6].ne1) + 2 * (sum(w.ne0 for w in qkv) + 2 * dim) + 2 * 2 * dim   # x in, y out, residual in
    for f in launches:
        f()
    L.check(ctx.hip.kf_graph_begin(ctx.h), "graph_begin")
    for f in launches:
        f()
    g = C.c_void_p()
    L.check(ctx.hip.kf_graph_end(ctx.h, C.byref(g)), "graph_end")
    head = m.weights[(-1, 1)]
    hd = head.desc()
    hx = torch.randn(dim, device=dev).to(torch.bfloat16)
    logits = torch.empty(head.ne0, dtype=torch.bfloat16, device=dev)

    def flush():
        L.check(ctx.hip.kf_lm_head(ctx.h, C.byref(hd), hx.data_ptr(), logits.data_ptr(), None, ctx._head_ws.data_ptr()), "kf_lm_head")
    ms = 0.0
    for r in range(reps + 2):
        flush()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        L.check(ctx.hip.kf_graph_launch(ctx.h, g), "graph_launch")
        ctx.record(e1)
        ctx.sync()
        if r >= 2:
            ms += ctx.elapsed_ms(e0, e1)
    ms /= reps
    n = len(launches)
    ach = nbytes / (ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed under profiles/ (taken on the 0.6B shapes by scratch/ub_matvec_chain.py; rocprofv3 --pmc
    # cannot run inside this process): FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, averaged over the three launch shapes
    traffic = None
    if (dim, nL, m.weights[(0, 6)].ne1) == (1024, 28, 3072):
        try:
            traffic = int(json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_matvec.json")))["hbm_bytes_per_launch"])
        except Exception:
            pass
    return {"bound": "hbm", "kernel": "kf::gemv_kernel<5, 1, 0> = 4-bit mat-vec of [norm+QKV], [o_proj+residual], [down_proj+residual]: %d launches per token, "
                                      "the largest share of the step's time (LM head: roofline_lm_head)" % n,
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
            "bytes_per_launch": int(nbytes / n), "us_per_launch": round(ms * 1e3 / n, 2), "launches": n,
            "note": "latency-bound: %.1f MB per launch is %.2f us at the HBM peak; the in-kernel time split is in DESIGN.md section 6" % (nbytes / n / 1e6, nbytes / n / HBM_PEAK_GBS / 1e3)}


def cpu_baseline(m, cfg, forced, budget_s, min_steps=64, max_steps=256, canon_steps=48, hots=None):
    """The CPU oracle (a port: no runnable CPU forward exists in the reference) decoding the SAME 4-bit model on this host's cores: weights and the KV rows of the
    128-token prompt are copied from the GPU model, then it decodes from position 128, teacher-forced on the GPU's ids so that both decode one sequence.
    (1) PARITY, once per summation order: `canon_steps` + 1 free-running steps.  In the order `value` is timed in -- the canonical one kernels and oracle share (oracle/kf_oracle.c
    sections 4c, 6 CANON; the library default) -- every greedy id AND every logit must equal the GPU's bit for bit (parity_timed_order; its mismatches must be 0).  In the
    v_dot2c / fp32 order the ids are compared too and a mismatch is classified (near-tie / inside twice the logit tolerance / beyond): parity_fast_order.  (2) TIMING: at least 64 steps (more while the
    budget lasts) with the mat-vec in the reference's own CPU idiom (two 8-lane AVX2 accumulators over 16 consecutive elements, rows over OpenMP threads:
    dotprod_fp16 / D_matvec, GST_float.cpp:75-101, 293-304) on a bf16 copy of the dequantised weights (what GetDataX produces), one pinned thread per core;
    `value` = 1 / median step time, with the 10th / 90th percentile beside it."""
    import numpy as np
    from oracle import oracle as O

    _pick_threads()
    om = O.from_device_model(m, attn_mode=O.ATTN_CANON)
    for l, hot in (hots or {}).items():   # the sparse forward's masks (D_matmul_sparse): the same rows hot on both sides
        om.set_hot(l, hot)
    prep = om.prepare_fast()
    p0 = 128
    gk, gv = m.kv_to_host()
    ok, ov = om.kv()
    ok[:, :p0] = gk[:, :p0]
    ov[:, :p0] = gv[:, :p0]
    # ---- (1) parity passes: the GPU decodes canon_steps + 1 positions from p0, free running, on the KV rows the oracle was just given (the rows of earlier runs behind p0
    # were produced from other prompt rows: the batched prefill above rewrote rows 0..p0-1); the oracle (canonical order) is teacher-forced on the GPU's ids.
    #   timed order (canonical): every id and every logit of the last position must be equal -- mismatches are parity failures;
    #   fast order (v_dot2c / fp32): ids may differ at near-ties; a mismatch is classified by the oracle's logit of the GPU's pick against the oracle's maximum.
    tok0 = int(m.tokens_out(cfg["max_seq"])[p0 - 1])
    O.set_order(O.ORDER_CANON)
    parity = {}
    try:
        for name, canonical in (("timed_order", True), ("fast_order", False)):
            m.set_canonical(canonical)
            m.set_state(tok0, p0)
            m.run_steps(p0, canon_steps + 1, True)
            m.sync()
            gpu_ids = m.tokens_out(cfg["max_seq"]).copy()
            g_logits = m.logits().copy()   # of position p0 + canon_steps
            same, near_tie, in_tol, logits_equal, within_tol = 0, 0, 0, None, None
            tok = tok0
            for i in range(canon_steps + 1):
                nxt, lg, _ = om.decode(tok, p0 + i, want_logits=(i == canon_steps or not canonical))
                g = int(gpu_ids[p0 + i])
                if nxt == g:
                    same += 1
                elif lg is not None:
                    # the oracle's logit of the GPU's pick against its own maximum: a difference of <= 2 bf16 ulps of the maximum (2^-7 relative: each side rounds its own fp32
                    # sum to bf16 once) is a tie inside the stated tolerance
                    lf = O.bf16_to_f32(lg)
                    gap = float(lf[nxt] - lf[g]) / max(abs(float(lf[nxt])), 1e-30)
                    near_tie += int(gap <= 2.0 ** -7)
                    in_tol += int(2.0 ** -7 < gap <= 2.0 ** -5)   # inside twice the stated logit tolerance (each side may be 2^-6 of the scale off)
                tok = g   # teacher-forced on the GPU's ids: both decode one sequence
                if i == canon_steps:
                    logits_equal = int((g_logits == lg).sum())
                    a, b = O.bf16_to_f32(g_logits), O.bf16_to_f32(lg)
                    within_tol = int((np.abs(a - b) <= 2.0 ** -6 * np.abs(b).max()).sum())
            n_c = canon_steps + 1
            parity[name] = {"summation_order": "canonical: kf_set_canonical(ctx, 1), the library default and the order `value` is timed in" if canonical else "kf_set_canonical(ctx, 0): v_dot2c_f32_bf16 / fp32 softmax",
                            "positions": [p0, p0 + canon_steps], "greedy_ids_compared": n_c, "greedy_ids_equal_oracle": same,
                            "mismatches_that_are_ties_within_2_bf16_ulps": near_tie, "mismatches_inside_twice_the_logit_tolerance": in_tol, "mismatches_beyond_tolerance": n_c - same - near_tie - in_tol,
                            "logits_compared": int(g_logits.size), "logits_equal_bit_for_bit": logits_equal, "logits_within_2^-6_of_scale": within_tol}
    finally:
        O.set_order(O.ORDER_DOT16)
        m.set_canonical(True)
    pt = parity.get("timed_order", {})
    same, n_c, logits_equal, n_logits = pt.get("greedy_ids_equal_oracle", 0), pt.get("greedy_ids_compared", 0), pt.get("logits_equal_bit_for_bit", 0), pt.get("logits_compared", 0)
    # ---- (2) timing pass, the reference's CPU dot-product order
    tok, n = tok0, 0
    t0 = time.perf_counter()
    steps = []
    while True:
        t1 = time.perf_counter()
        om.decode(tok, p0 + n, want_logits=False)
        steps.append(time.perf_counter() - t1)
        tok = int(gpu_ids[p0 + n])
        n += 1
        if n >= max_steps or p0 + n >= cfg["max_seq"] - 1 or (n >= min_steps and time.perf_counter() - t0 > budget_s):
            break
    sp = _spread(steps[2:] if n > 8 else steps)
    om.close()
    return {"value": round(1e3 / sp["median_ms"], 3), "unit": "tokens/s", "cores": O.num_threads(), "kind": "port",
            "sample": "%d decode steps at positions %d..%d of the same model; AVX2 two-accumulator dot on a bf16 dequantised copy (%d MB), OpenMP rows, "
                      "threads pinned one per core" % (n, p0, p0 + n - 1, max(prep, 0) // 2 ** 20), "step_ms": sp,
            "parity_pass": "%d free-running greedy steps at positions %d..%d per summation order, oracle teacher-forced on the GPU's ids; the timed order must be equal bit for bit "
                           "(ids and all %d logits of position %d)" % (n_c, p0, p0 + n_c - 1, n_logits, p0 + canon_steps),
            "parity_timed_order": parity.get("timed_order"), "parity_fast_order": parity.get("fast_order"),
            "greedy_ids_equal_gpu": same, "greedy_ids_compared": n_c, "logits_equal_bit_for_bit": logits_equal, "logits_compared": n_logits,
            "mismatches_beyond_tolerance": n_c - same}


def config4_slice_cpu(n_layer=4, n_steps=24):
    """SURVEY section 8d, config 4's CPU column: the oracle cannot hold Qwen3-32B (18 GB, minutes per token), so a 4-LAYER SLICE of the same shapes (dim 5120, 64 / 8 heads of 128,
    ffn 25600, vocab 151936, untied bf16 head; RTN 4-bit g128 layers) is decoded on the GPU (canonical order) and by the oracle on this host's cores from the same device weights:
    greedy ids and the last logits must be equal bit for bit; the oracle's AVX2 pass is timed.  Reported for the slice only; 64 / 4 of the layer time is what a full model would cost."""
    import numpy as np
    from koifish_amd import lib as L, synth
    from oracle import oracle as O
    cfg = dict(synth.CONFIGS["qwen3-32b"], n_layer=n_layer, max_seq=512)
    m = synth.build_on_gpu(cfg, seed=1234, layer_type=L.Q4, head_type=L.BF16)
    _pick_threads()
    om = O.from_device_model(m, attn_mode=O.ATTN_CANON)
    prep = om.prepare_fast()
    forced = np.full(cfg["max_seq"], -1, dtype=np.int32)
    forced[:128] = np.random.default_rng(7).integers(0, cfg["vocab"], size=128)
    m.set_forced(forced)
    m.set_state(int(forced[0]), 0)
    m.run_steps(0, 128 + n_steps, True)
    m.sync()
    gpu_ids = m.tokens_out(cfg["max_seq"])
    g_logits = m.logits().copy()
    O.set_order(O.ORDER_CANON)
    same, lg = 0, None
    try:
        tok = int(forced[0])
        for p in range(128 + n_steps):   # canonical order, teacher-forced on the GPU's ids: the prompt, then the free-running stretch
            nxt, lg, _ = om.decode(tok, p, want_logits=(p == 128 + n_steps - 1))
            if p >= 127:
                same += int(nxt == int(gpu_ids[p]))
            tok = int(forced[p + 1]) if p + 1 < 128 else int(gpu_ids[p])
    finally:
        O.set_order(O.ORDER_DOT16)
    steps = []
    tok = int(gpu_ids[127])
    for i in range(n_steps):   # timed: the reference's CPU dot-product idiom on the bf16 dequantised copy, positions 128 ..
        t1 = time.perf_counter()
        om.decode(tok, 128 + i, want_logits=False)
        steps.append(time.perf_counter() - t1)
        tok = int(gpu_ids[128 + i])
    sp = _spread(steps[2:])
    wbytes = sum(w.algorithmic_bytes() for (layer, slot), w in m.weights.items() if not (layer == -1 and slot == 0))
    om.close()
    m.close()
    return {"value": round(1e3 / sp["median_ms"], 3), "unit": "tokens/s of the %d-layer slice" % n_layer, "cores": O.num_threads(), "kind": "port",
            "sample": "%d decode steps at positions 128..%d of a %d-layer Qwen3-32B-shaped model (4-bit layers, bf16 head; %d MB dequantised copy); a full 64-layer model would take ~%.0f ms per token on these cores "
                      "(layers x 16 + head)" % (n_steps, 127 + n_steps, n_layer, max(prep, 0) // 2 ** 20, sp["median_ms"] * 16), "step_ms": sp,
            "weight_bytes_4bit": int(wbytes), "greedy_ids_compared": n_steps + 1, "greedy_ids_equal_gpu": same, "logits_equal_bit_for_bit": int((g_logits == lg).sum()), "logits_compared": int(g_logits.size)}


def _pick_threads():
    """the OpenMP thread count this host sustains on the decode's mat-vecs: probed on a 3072 x 1024 and a 151936 x 1024 product (the FFN and LM-head shapes)
    over candidate counts -- a container may expose more logical CPUs than it can use, and the second socket costs more in remote memory and barriers
    than it adds.  Set before the weights are first touched, so that pages land next to the threads that read them."""
    from oracle import oracle as O
    if "OMP_NUM_THREADS" in os.environ:
        return O.num_threads()
    top = O.num_threads()
    try:   # a cgroup CPU quota below the visible CPU count: more busy threads than the quota are throttled in 100 ms periods (spikes of ~90 ms per step)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            top = max(1, min(top, int(int(q) / int(per))))
    except Exception:
        pass
    best, best_t = top, None
    for c in sorted(set([c for c in (8, 16, 24, 32, 48, 64, 96, 128, 192, 256) if c <= top] + [top])):
        O.set_num_threads(c)
        t = 28 * 7 * O.bench_matvec(3072, 1024, 8) + O.bench_matvec(151936, 1024, 3)
        if best_t is None or t < best_t:
            best, best_t = c, t
    O.set_num_threads(best)
    return best


def _physical_cores():
    """distinct (package, core) pairs among the CPUs this process may run on: hardware threads are not cores"""
    try:
        allowed = _ALL_CPUS or os.sched_getaffinity(0)
        seen = set()
        for c in allowed:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        n = max(1, len(seen))
    except Exception:
        n = max(1, (os.cpu_count() or 2) // 2)
    try:   # a cgroup CPU quota below the core count: more busy threads than the quota are throttled (what _pick_threads honours for the OpenMP legs)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def _spread(step_s):
    import numpy as np
    a = np.sort(np.asarray(step_s))
    return {"median_ms": round(float(np.median(a)) * 1e3, 3), "p10_ms": round(float(a[int(0.1 * (a.size - 1))]) * 1e3, 3), "p90_ms": round(float(a[int(0.9 * (a.size - 1))]) * 1e3, 3)}


def cpu_fp16_decode(cfg, device, n_new, n_prompt=128):
    """BASELINE config 1 (plumbing, no GPU work in the timed part): Qwen3-0.6B with IEEE-half weights -- N(0, 0.02) rows drawn once (seed 1234) and rounded
    to half, norms 1 + N(0, 0.01) -- decoded by the CPU oracle on this host's cores: the 128-token prompt (ids randint, seed 7) is fed token by token
    as the reference's chat loop does (GoPT.cpp:1139-1146), then n_new greedy steps.  Mat-vec = dotprod_fp16 restated (_mm256_cvtph_ps, two 8-lane
    accumulators, GST_float.cpp:75-101) over OpenMP rows (D_matvec).  1.19 GB of weights per token."""
    import numpy as np
    import torch
    from koifish_amd import synth
    from oracle import oracle as O

    _pick_threads()
    g = torch.Generator(device=device)
    g.manual_seed(1234)

    def half(r, c):   # drawn on the GPU only because 596 M normal samples take the host ~10 s; the decode below touches no GPU
        return O.QWeight(O.F16, r, c, (torch.randn(r, c, generator=g, device=device, dtype=torch.float32) * 0.02).to(torch.float16).cpu().numpy().view(np.uint16))

    def nrm(n):
        return O.f32_to_bf16((1.0 + 0.01 * torch.randn(n, generator=g, device=device, dtype=torch.float32)).cpu().numpy())

    w = {"embed": half(cfg["vocab"], cfg["dim"]), "final_norm": nrm(cfg["dim"]), "layers": []}
    w["head"] = w["embed"]
    for _ in range(cfg["n_layer"]):
        d = {s: half(*synth.SHAPES[s](cfg)) for s in synth.SLOTS}
        d["norm_in"], d["norm_post"], d["qn"], d["kn"] = nrm(cfg["dim"]), nrm(cfg["dim"]), nrm(cfg["head_dim"]), nrm(cfg["head_dim"])
        w["layers"].append(d)
    om = O.Qwen3Oracle(cfg, w)
    om.prepare_fast()
    prompt = np.random.default_rng(7).integers(0, cfg["vocab"], size=n_prompt)
    t0 = time.perf_counter()
    nxt = 0
    for p, t in enumerate(prompt):
        nxt, _, _ = om.decode(int(t), p, want_logits=False)
    t_prompt = time.perf_counter() - t0
    steps, ids = [], []
    for i in range(n_new):
        t1 = time.perf_counter()
        ids.append(nxt)
        nxt, _, _ = om.decode(int(nxt), n_prompt + i, want_logits=False)
        steps.append(time.perf_counter() - t1)
    sp = _spread(steps)
    nbytes = sum(x.data.nbytes for d in w["layers"] for x in (d[s] for s in synth.SLOTS)) + w["head"].data.nbytes
    om.close()
    return {"value": round(1e3 / sp["median_ms"], 3), "unit": "tokens/s", "cores": O.num_threads(), "kind": "port",
            "workload": "Qwen3-0.6B fp16 greedy decode, 128-token prompt, CPU inference path (BASELINE.json configs[0])",
            "sample": "%d-token prompt fed token by token (%.1f tokens/s), then %d greedy steps" % (n_prompt, n_prompt / t_prompt, n_new), "step_ms": sp,
            "weight_bytes_per_token": int(nbytes), "host_GBs": round(nbytes / (sp["median_ms"] * 1e-3) / 1e9, 1), "distinct_ids": len(set(ids))}


# ---------------------------------------------------------------------------------------------------------------------------------------------------------
# The ONE line on stdout is a compact record (target <= 4 KB, hard limit 8 KB: tests/test_bench_line.py): metric / value / config / roofline / cpu_baseline and one
# short numeric object per side leg.  Everything else this run measured -- prose notes, workloads, nested parity tables -- goes to bench_detail.json next to this
# file (and to gpurun_out/ when that directory exists, so that it travels back from a GPU box); the line names the file.
LINE_LIMIT = 8192
DETAIL_FILE = "bench_detail.json"


def _g(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def _keep(d, keys):
    """the named keys of d that exist and are not None; a leg that failed keeps only a short error"""
    if not isinstance(d, dict):
        return None
    if "error" in d:
        return {"error": str(d["error"])[:120]}
    return {k: d[k] for k in keys if d.get(k) is not None}


def _cpu_short(cb):
    """a cpu_baseline object as {value, unit, cores, kind, sample (short), ids_equal [equal, compared], logits_equal [equal, compared]}"""
    if not isinstance(cb, dict):
        return None
    if "error" in cb:
        return {"error": str(cb["error"])[:120]}
    o = {k: cb[k] for k in ("value", "unit", "cores", "kind") if k in cb}
    if "sample" in cb:
        o["sample"] = str(cb["sample"]).split(";")[0][:110]
    if cb.get("greedy_ids_compared") is not None:
        o["ids_equal"] = [cb.get("greedy_ids_equal_gpu"), cb.get("greedy_ids_compared")]
    if cb.get("logits_compared") is not None:
        o["logits_equal"] = [cb.get("logits_equal_bit_for_bit"), cb.get("logits_compared")]
    if cb.get("extrapolated_step_ms") is not None:
        o["extrapolated_step_ms"] = cb["extrapolated_step_ms"]
    return o


def _parity_ok(cb):
    """True when a cpu_baseline object's timed-order parity pass found every id and every logit equal"""
    if not isinstance(cb, dict) or cb.get("greedy_ids_compared") in (None, 0):
        return None
    return bool(cb.get("greedy_ids_equal_gpu") == cb.get("greedy_ids_compared") and cb.get("logits_equal_bit_for_bit") == cb.get("logits_compared"))


def _xcd_short(x):
    if not isinstance(x, dict):
        return None
    if x.get("error") or x.get("skipped"):
        return {"error": str(x.get("error") or x.get("skipped"))[:120]}
    o = _keep(x, ("streams", "batch", "tokens_per_s", "ms_per_step_all_streams", "frac_vs_single_sequence_roofline", "hbm_frac_batch", "valu_insts_per_sequence_step", "traffic_per_sequence_step"))
    p = x.get("parity") or {}
    v = p.get("sequence_0_ids_equal_single_sequence_engine") if p else None
    o["parity"] = None if v is None else bool(v)
    q = x.get("request_queue")
    if isinstance(q, dict) and "generated_tokens_per_s" in q:
        o["request_queue"] = _keep(q, ("requests", "prompt_tokens", "new_tokens", "requests_per_s", "generated_tokens_per_s"))
        if isinstance(q.get("prefill_batch"), dict):
            o["request_queue"]["prefill_batch"] = _keep(q["prefill_batch"], ("prompts_per_batch", "ms_per_prompt", "generated_tokens_per_s"))
    return o


def compact_line(out, detail=DETAIL_FILE):
    """the compact record of a full result dict (what main() / tp_main() assembled)"""
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    c["config"] = {k: v for k, v in (("workload", cfg.get("workload")), ("order", "canonical: bit-exact vs the CPU oracle" if "summation_order" in cfg else None),
                                     ("decode_path", str(cfg.get("decode_path", "")).split(":")[0][:80] or None), ("lm_head", cfg.get("lm_head")), ("replicas", cfg.get("replicas")),
                                     ("tp", cfg.get("tp")), ("exchange", cfg.get("exchange")), ("layers", cfg.get("layers")), ("ranks_in_process_group", cfg.get("ranks_in_process_group")),
                                     ("device_ms_per_step", cfg.get("device_ms_per_step"))) if v is not None}
    rl = out.get("roofline")
    if isinstance(rl, dict):
        r = {k: rl[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "bytes_per_step_per_rank", "us_per_launch", "launches") if k in rl}
        r.setdefault("traffic", None)
        r["kernel"] = str(rl.get("kernel", "")).split(" = ")[0].split(" (")[0][:60]
        if rl.get("traffic_source"):
            r["traffic_source"] = str(rl["traffic_source"]).split(":")[0]
        if isinstance(rl.get("run_of_steps"), dict) and "frac" in rl["run_of_steps"]:
            r["run_of_16"] = {k: rl["run_of_steps"][k] for k in ("us_per_step", "frac") if k in rl["run_of_steps"]}
        c["roofline"] = r
    if isinstance(out.get("step_roofline"), dict):
        c["step_roofline"] = {k: out["step_roofline"][k] for k in ("bytes_per_step", "achieved", "peak", "unit", "frac") if k in out["step_roofline"]}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = _cpu_short(cb)
        fo = cb.get("parity_fast_order") or {}
        if fo:
            c["cpu_baseline"]["fast_order_ids_equal"] = [fo.get("greedy_ids_equal_oracle"), fo.get("greedy_ids_compared")]
    else:
        c["cpu_baseline"] = None
    side = {}
    if isinstance(out.get("fast_order_mode"), dict):
        side["fast_order"] = _keep(out["fast_order_mode"], ("tokens_per_s", "ms_per_step"))
    if isinstance(out.get("roofline_lm_head"), dict):
        side["lm_head_kernel"] = _keep(out["roofline_lm_head"], ("frac", "us_per_launch", "traffic"))
    pf = out.get("prefill")
    if isinstance(pf, dict):
        def one(p):
            if not isinstance(p, dict) or "error" in p:
                return _keep(p, ())
            return {k: v for k, v in (("tokens", p.get("prompt_tokens")), ("ms", p.get("ms")), ("mfma_frac", _g(p, "roofline", "mfma_frac")), ("hbm_frac", _g(p, "roofline", "hbm_frac")),
                                      ("path", p.get("path"))) if v is not None}
        side["prefill"] = [x for x in (one(pf), one(pf.get("long_prompt")) if pf.get("long_prompt") else None) if x]
    if isinstance(out.get("cpu_baseline_fp16"), dict):   # BASELINE config 1
        side["config1_cpu_fp16"] = _keep(out["cpu_baseline_fp16"], ("value", "unit", "cores", "host_GBs"))
    x = out.get("xcd_replicas")
    if isinstance(x, dict):
        side["xcd_replicas"] = _xcd_short(x)
        for k in ("one_per_xcd", "two_per_xcd", "batched"):
            if isinstance(x.get(k), dict) and side["xcd_replicas"] is not None and "error" not in side["xcd_replicas"]:
                side["xcd_replicas"][k] = _keep(_xcd_short(x[k]), ("streams", "batch", "tokens_per_s", "frac_vs_single_sequence_roofline", "hbm_frac_batch", "parity", "error"))
        if isinstance(x.get("prefill_then_decode"), dict) and "error" not in side["xcd_replicas"]:
            side["xcd_replicas"]["prompt_plus_32_tokens_per_s"] = x["prefill_then_decode"].get("prompt_plus_32_tokens_per_s")
    t = out.get("config3_train_step")
    if isinstance(t, dict):
        side["config3_train_step"] = _keep(t, ("ms", "tokens_per_s", "mfma_frac", "params_updated", "one_timed_region", "host_loop"))
        if "error" not in t:
            side["config3_train_step"]["cpu_baseline"] = _cpu_short(t.get("cpu_baseline"))
    t = out.get("config5_sparse_1bit")
    if isinstance(t, dict):
        side["config5_sparse_1bit"] = _keep(t, ("tokens_per_s", "ms_per_step", "frac"))
        if "error" not in t:
            side["config5_sparse_1bit"]["parity"] = _parity_ok(t.get("cpu_baseline"))
            side["config5_sparse_1bit"]["cpu_tokens_per_s"] = _g(t, "cpu_baseline", "value")
            if isinstance(t.get("xcd_replicas"), dict):
                side["config5_sparse_1bit"]["xcd_replicas"] = _keep(_xcd_short(t["xcd_replicas"]), ("streams", "batch", "tokens_per_s", "frac_vs_single_sequence_roofline", "hbm_frac_batch", "parity", "error"))
    for nm in ("qwen3_1p7b_shape", "qwen3_4b_shape", "qwen3_8b_shape"):
        t = out.get(nm)
        if isinstance(t, dict):
            side[nm] = _keep(t, ("tokens_per_s", "frac", "per_layer_launches_tokens_per_s", "one_sequence_path"))
            if "error" not in t and isinstance(t.get("one_sequence_tp_over_xcds"), dict):
                o1 = t["one_sequence_tp_over_xcds"]
                side[nm]["one_seq_tp_over_xcds"] = _keep(o1, ("tokens_per_s", "frac"))
                if "error" not in o1:
                    side[nm]["one_seq_tp_over_xcds"]["parity"] = _g(o1, "parity", "first_24_ids_equal_per_launch_rank_step")
            if "error" not in t and isinstance(t.get("xcd_replicas"), dict):
                side[nm]["xcd_replicas"] = _keep(_xcd_short(t["xcd_replicas"]), ("streams", "batch", "tokens_per_s", "frac_vs_single_sequence_roofline", "hbm_frac_batch", "parity", "error"))
    t = out.get("config4_one_gpu")
    if isinstance(t, dict):
        o = _keep(t, ("tokens_per_s", "ms_per_step", "frac"))
        if "error" not in t:
            o["prefill_2047_ms"] = _g(t, "prefill_2047_tokens", "ms")
            o["prefill_2047_mfma_frac"] = _g(t, "prefill_2047_tokens", "roofline", "mfma_frac")
            cs = t.get("cpu_baseline_4_layer_slice")
            o["cpu_4_layer_slice"] = {"value": _g(cs, "value"), "cores": _g(cs, "cores"), "parity": _parity_ok(cs)} if isinstance(cs, dict) and "error" not in cs else _keep(cs, ())
            o["tp8_rank_step_per_launch"] = _keep(t.get("tp8_virtual_ranks"), ("layers_run", "ms_per_rank_step", "frac"))
            o["tp8_rank_engine"] = _keep(t.get("tp8_rank_engine"), ("layers_run", "ms_per_rank_step", "frac", "xcds_per_rank", "parity"))
            xx = t.get("tp8_ranks_as_xcds")
            o["tp8_ranks_as_xcds"] = _keep(xx, ("tokens_per_s", "ms_per_step", "frac"))
            if isinstance(xx, dict) and "error" not in xx:
                o["tp8_ranks_as_xcds"]["parity"] = _g(xx, "parity", "first_24_ids_equal_per_launch_rank_step")
            o["multi_gpu"] = "no scaling curve measured on hardware"
        side["config4_one_gpu"] = {k: v for k, v in o.items() if v is not None}
    t = out.get("config4_tp")
    if isinstance(t, dict):
        side["config4_tp"] = _keep(t, ("tokens_per_s", "ms_per_step", "n_gpus", "scaling", "tp", "exchange", "ranks_in_process_group", "layers"))
        if "error" not in t:
            side["config4_tp"]["frac"] = _g(t, "roofline", "frac")
    for k in ("side_legs_error", "roofline_error"):
        if out.get(k):
            side[k] = str(out[k])[:120]
    if side:
        c["side"] = side
    if out.get("scaling_curve"):
        c["scaling_curve"] = "none measured on hardware"
    if out.get("wall_s") is not None:
        c["wall_s"] = out["wall_s"]
    c["detail"] = detail
    return c


def emit(out, detail=DETAIL_FILE):
    """write the full result beside this file, print the compact line (the only thing on stdout)"""
    txt = json.dumps(out, indent=1, allow_nan=False, default=str)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, detail), "w") as f:
                    f.write(txt + "\n")
        except Exception:
            pass
    line = json.dumps(compact_line(out, detail), allow_nan=False, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:   # never hand the driver a line it cannot parse: drop the side objects, keep the contract's fields
        c = compact_line(out, detail)
        c["side"] = {"dropped": "line would be %d bytes: see %s" % (len(line), detail)}
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    print(line, flush=True)


if __name__ == "__main__":
    main()
