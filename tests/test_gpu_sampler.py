"""kf_sample (GeneratOnPrompt::Sample on the device) vs the oracle: token ids and rng state bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _dev_sample(ctx, lg_t, n, k, temp, top_p, rng_t, tok_t):
    return ctx.hip.kf_sample(ctx.h, lg_t.data_ptr(), n, k, temp, top_p, rng_t.data_ptr(), tok_t.data_ptr(), None, None, None, 0)


@pytest.mark.parametrize("n,k,temp,top_p", [(151936, 50, 0.6, 0.95), (4096, 8, 1.0, 0.9), (512, 2, 0.3, 0.5), (4096, 1024, 1.5, 1.0), (2100, 1000, 0.8, 0.99),
                                            (151936, 40, 2.0, 1e-6)])
def test_sample_ids_and_rng_match_oracle(ctx, n, k, temp, top_p):
    rng = np.random.default_rng(n + k)
    for trial in range(6):
        coarse = trial % 2 == 1          # coarse logits: many exact ties among bf16 values
        lg = O.f32_to_bf16(rng.normal(0, 0.4 if coarse else 3.0, size=n).astype(np.float32))
        seed = int(rng.integers(1, 2 ** 62))
        st = np.array([seed], dtype=np.uint64)
        lg_t = torch.from_numpy(lg.view(np.int16)).to(ctx.device)
        rng_t = torch.from_numpy(st.view(np.int64).copy()).to(ctx.device)
        tok_t = torch.zeros(1, dtype=torch.int32, device=ctx.device)
        for draw in range(8):
            want = O.sample(lg, k, temp, top_p, st)
            assert _dev_sample(ctx, lg_t, n, k, temp, top_p, rng_t, tok_t) == 0, ctx.hip.kf_last_error()
            ctx.sync()
            assert int(tok_t.item()) == want, "trial %d draw %d" % (trial, draw)
            assert int(rng_t.cpu().numpy().view(np.uint64)[0]) == int(st[0])


def test_sample_bad_args(ctx):
    lg = torch.zeros(1024, dtype=torch.bfloat16, device=ctx.device)
    r = torch.ones(1, dtype=torch.int64, device=ctx.device)
    t = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    assert _dev_sample(ctx, lg, 1024, 1, 1.0, 0.9, r, t) == -20       # greedy branch: not a kf_sample case
    assert _dev_sample(ctx, lg, 1024, 512, 1.0, 0.9, r, t) == -20     # k >= n/2
    assert _dev_sample(ctx, lg, 1024, 50, 0.0, 0.9, r, t) == -20
    assert _dev_sample(ctx, lg, 1024, 50, 1.0, 0.0, r, t) == -20
    assert ctx.hip.kf_sample(ctx.h, lg.data_ptr(), 1024, 50, 1.0, 0.9, None, t.data_ptr(), None, None, None, 0) == -20
    assert b"top_k" in ctx.hip.kf_last_error() or b"null" in ctx.hip.kf_last_error()


@pytest.mark.parametrize("prefill_mode", [0, 1])
def test_generate_with_sampler(prefill_mode):
    """Sampling amplifies the <= 1 ulp logit differences between the device and the oracle (a coin next to a CDF edge flips the token and
    everything after it), so whole sequences are compared between device paths, the oracle's sequence only on a prefix, and the pick
    itself is checked exactly by handing the DEVICE's logits of every step to the oracle's sampler."""
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 1234, w_std=0.1)
    gm = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    om = oracle_model(cfg, raw, L.Q4, L.BF16)
    prompt = prompt_ids(cfg, 12)
    samp = dict(top_k=20, temperature=0.9, top_p=0.9, seed=2024)
    ref = om.generate(prompt.tolist(), 30, sampler=samp)
    greedy = om.generate(prompt.tolist(), 30)
    assert ref != greedy, "fixture: sampling should leave the greedy path"
    gm.set_prefill_mode(prefill_mode)
    gm.set_sampler(**samp)
    a = gm.generate(prompt, 30, use_graph=True)
    gm.set_sampler(**samp)                       # reseed
    b = gm.generate(prompt, 30, use_graph=False)
    assert b == a, "graph replay and eager launches draw different tokens"
    assert a[:12] == ref[:12], "sampled ids leave the oracle's within the first steps"
    gm.set_sampler(temperature=0.0)
    assert gm.generate(prompt, 30, use_graph=True) == greedy
    gm.close()


def test_pick_is_exact_on_device_logits():
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 1234, w_std=0.1)
    gm = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    samp = dict(top_k=20, temperature=0.9, top_p=0.9, seed=77)
    gm.set_sampler(**samp)
    st = np.array([77], dtype=np.uint64)
    tok = 5
    for pos in range(40):
        nxt, logits = gm.forward(tok, pos)       # eager step: head -> kf_sample -> state update; one coin per step
        want = O.sample(logits, samp["top_k"], samp["temperature"], samp["top_p"], st)
        assert nxt == want, "step %d" % pos
        tok = nxt
    gm.close()


@pytest.mark.parametrize("n,k,temp,top_p", [(151936, 50, 0.7, 0.95), (4096, 1024, 1.2, 1.0), (2100, 9, 0.5, 0.8), (512, 2, 1.0, 0.9)])
def test_sample_true_topk_matches_oracle(ctx, n, k, temp, top_p):
    """kf_sample_topk: radix select of the k largest logits (ties towards the lower index), then the same pick"""
    rng = np.random.default_rng(n * 3 + k)
    for trial in range(6):
        coarse = trial % 2 == 1
        lg = O.f32_to_bf16(rng.normal(0, 0.4 if coarse else 3.0, size=n).astype(np.float32))
        if trial == 4:
            lg[:] = lg[0]                      # every logit equal: the candidates are tokens 0..k-1
        seed = int(rng.integers(1, 2 ** 62))
        st = np.array([seed], dtype=np.uint64)
        lg_t = torch.from_numpy(lg.view(np.int16)).to(ctx.device)
        rng_t = torch.from_numpy(st.view(np.int64).copy()).to(ctx.device)
        tok_t = torch.zeros(1, dtype=torch.int32, device=ctx.device)
        for draw in range(6):
            want = O.sample(lg, k, temp, top_p, st, true_topk=True)
            assert ctx.hip.kf_sample_topk(ctx.h, lg_t.data_ptr(), n, k, temp, top_p, rng_t.data_ptr(), tok_t.data_ptr(), None, None, None, 0) == 0
            ctx.sync()
            assert int(tok_t.item()) == want, "trial %d draw %d" % (trial, draw)


def test_generate_true_topk_pick_on_device_logits():
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 1234, w_std=0.1)
    gm = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    samp = dict(top_k=12, temperature=0.8, top_p=0.9, seed=5)
    gm.set_sampler(true_topk=True, **samp)
    st = np.array([5], dtype=np.uint64)
    tok = 3
    for pos in range(30):
        nxt, logits = gm.forward(tok, pos)
        assert nxt == O.sample(logits, samp["top_k"], samp["temperature"], samp["top_p"], st, true_topk=True), "step %d" % pos
        tok = nxt
    gm.close()
