"""CPU: the oracle's restatement of D_matmul_sparse (src/Utils/GST_float.cpp:306-318) -- a row is computed only when hot[i] == 1, a cold row is
0 (+ bias) -- against its own dense product and a numpy statement of the same rule."""
import numpy as np

from koifish_amd import lib as L
from oracle import oracle as O


def test_linear_masked_is_the_dense_row_or_zero():
    rng = np.random.default_rng(3)
    m, k = 384, 256
    w = O.f32_to_bf16(rng.normal(0, 0.05, size=(m, k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=k).astype(np.float32))
    bias = O.f32_to_bf16(rng.normal(0, 0.1, size=m).astype(np.float32))
    hot = (rng.random(m) < 0.2).astype(np.int32)
    hot[::11] = 2   # only the value 1 is hot
    for t in (L.Q4, L.BOOL1, L.BF16):
        ow = O.quantize(w, m, k, t)
        dense, dense_b = O.linear(ow, x), O.linear(ow, x, bias=bias)
        y, yb = O.linear_masked(ow, x, hot), O.linear_masked(ow, x, hot, bias)
        assert np.array_equal(y[hot == 1], dense[hot == 1]) and np.array_equal(yb[hot == 1], dense_b[hot == 1])
        assert not y[hot != 1].any() and np.array_equal(yb[hot != 1], bias[hot != 1])


def test_model_with_hot_masks_differs_and_dense_mask_is_identity():
    from koifish_amd import synth
    from helpers import oracle_model
    cfg = dict(dim=256, n_layer=2, n_head=4, n_kv=2, head_dim=64, ffn=512, vocab=512, max_seq=32, theta=1e6, tied=True)
    raw = synth.raw_weights_numpy(cfg, 1, w_std=0.1)
    a, b = oracle_model(cfg, raw, L.BOOL1, L.BF16), oracle_model(cfg, raw, L.BOOL1, L.BF16)
    for l in range(cfg["n_layer"]):
        b.set_hot(l, np.ones(cfg["ffn"], dtype=np.int32))
    _, la, _ = a.decode(5, 0)
    _, lb, _ = b.decode(5, 0)
    assert np.array_equal(la, lb), "an all-hot mask is the dense forward"
    hot = np.zeros(cfg["ffn"], dtype=np.int32)
    hot[::5] = 1
    for l in range(cfg["n_layer"]):
        b.set_hot(l, hot)
    _, lc, _ = b.decode(5, 0)
    assert not np.array_equal(la, lc)
