"""The oracle's restatement of GeneratOnPrompt::Sample (GoPT.cpp:594-790) against literal re-enactments of the reference's steps."""
import heapq

import numpy as np

from oracle import oracle as O


def _bf16_logits(rng, n, coarse=False):
    x = rng.normal(0, 3.0 if not coarse else 0.5, size=n).astype(np.float32)
    return O.f32_to_bf16(x)


def _select_literal(vals, k):
    """TOPK_heap::Select with its std::priority_queue<int> (ordered by INDEX, default std::less) acted out with heapq"""
    heap = []  # max-heap on index via negation
    for i in range(len(vals)):
        if len(heap) < k:
            heapq.heappush(heap, -i)
        elif vals[i] > vals[-heap[0]]:
            heapq.heappop(heap)
            heapq.heappush(heap, -i)
    picks = []
    while heap:
        picks.append(-heapq.heappop(heap))
    return picks


def _xorshift_f32(state):
    m = (1 << 64) - 1
    state ^= state >> 12
    state = (state ^ (state << 25)) & m
    state ^= state >> 27
    u = ((state * 0x2545F4914F6CDD1D) & m) >> 32
    return state, np.float32(u >> 8) / np.float32(16777216.0)


def test_candidate_set_is_what_the_reference_heap_keeps():
    rng = np.random.default_rng(0)
    for n, k in ((512, 50), (4096, 8), (1000, 2), (300, 149)):
        lg = _bf16_logits(rng, n)
        vals = O.bf16_to_f32(lg)
        st = np.array([7], dtype=np.uint64)
        tok, picks, probs, npick = O.sample(lg, k, 0.8, 0.95, st, want_detail=True)
        lit = _select_literal(vals, k)
        assert sorted(lit) == sorted(picks.tolist())
        assert set(range(k - 1)) <= set(picks.tolist()), "indices 0..k-2 always survive the reference's heap"
        # descending by logit; equal logits keep the extraction order (descending index, newest first)
        pv = vals[picks]
        assert (np.diff(pv) <= 0).all()
        order = {p: i for i, p in enumerate(lit)}
        for a, b in zip(picks[:-1], picks[1:]):
            if vals[a] == vals[b]:
                assert order[int(a)] < order[int(b)]
        assert abs(probs.sum() - 1.0) < 1e-5 and 1 <= npick <= k and tok in picks[:npick]


def test_rng_known_answers_and_coin_walk():
    import ctypes as C
    fn = O.lib().kfo_random_f32
    fn.restype, fn.argtypes = C.c_float, [C.c_void_p]
    st = np.array([42], dtype=np.uint64)
    s = 42
    for _ in range(5):
        s, f = _xorshift_f32(s)
        assert np.float32(fn(O._p(st))) == f and int(st[0]) == s
    # the C routine through kfo_sample: with top_p tiny only the best candidate is kept, whatever the coin
    rng = np.random.default_rng(1)
    lg = _bf16_logits(rng, 2048)
    st = np.array([42], dtype=np.uint64)
    tok, picks, probs, npick = O.sample(lg, 50, 0.7, 1e-6, st, want_detail=True)
    assert npick == 1 and tok == picks[0]
    s, _ = _xorshift_f32(42)
    assert int(st[0]) == s, "one xorshift64* step per sampled token"
    # coin walk re-enacted in numpy
    st = np.array([123456789], dtype=np.uint64)
    tok, picks, probs, npick = O.sample(lg, 50, 1.3, 0.9, st, want_detail=True)
    _, coin = _xorshift_f32(123456789)
    ps = np.float32(0)
    for j in range(npick):
        ps = np.float32(ps + probs[j])
    coin = np.float32(coin * ps)
    cdf, want = np.float32(0), picks[npick - 1]
    for j in range(npick):
        cdf = np.float32(cdf + probs[j])
        if coin < cdf:
            want = picks[j]
            break
    assert tok == want


def test_top_p_one_keeps_all_and_bad_args():
    rng = np.random.default_rng(2)
    lg = _bf16_logits(rng, 1024)
    st = np.array([5], dtype=np.uint64)
    _, _, _, npick = O.sample(lg, 40, 1.0, 1.0, st, want_detail=True)
    assert npick == 40
    assert O.sample(lg, 1, 1.0, 0.9, st) == -1          # top_k == 1 is the greedy branch
    assert O.sample(lg, 512, 1.0, 0.9, st) == -1        # assert(nPick < dim/2) in TOPK_heap::Select
    assert O.sample(lg, 40, 0.0, 0.9, st) == -1         # temperature == 0 is the greedy branch


def test_empirical_frequencies_follow_the_probabilities():
    rng = np.random.default_rng(3)
    lg = _bf16_logits(rng, 600)
    st = np.array([99], dtype=np.uint64)
    _, picks, probs, npick = O.sample(lg, 6, 1.0, 1.0, np.array([1], dtype=np.uint64), want_detail=True)
    counts = {}
    draws = 20000
    for _ in range(draws):
        t = O.sample(lg, 6, 1.0, 1.0, st)
        counts[t] = counts.get(t, 0) + 1
    for p, pr in zip(picks, probs):
        assert abs(counts.get(int(p), 0) / draws - pr) < 0.02


def test_true_topk_candidates():
    rng = np.random.default_rng(8)
    for n, k, coarse in ((4096, 50, False), (4096, 50, True), (600, 7, True)):
        lg = _bf16_logits(rng, n, coarse)
        vals = O.bf16_to_f32(lg)
        tok, picks, probs, npick = O.sample(lg, k, 0.9, 0.95, np.array([3], dtype=np.uint64), want_detail=True, true_topk=True)
        want = sorted(range(n), key=lambda i: (-vals[i], i))[:k]          # k largest, ties to the lower index, ordered the same way
        assert picks.tolist() == want
        assert abs(probs.sum() - 1.0) < 1e-5 and tok in picks[:npick]
