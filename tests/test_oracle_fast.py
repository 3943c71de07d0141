"""CPU: the oracle's vectorised mat-vec (AVX2 + F16C, the structure of the reference's dotprod_fp16 / dotprod_fp8, src/Utils/GST_float.cpp:75-130)
against its scalar 16-lane form: the same values in the same order, so every bit must be equal; and IEEE-half weights (BASELINE config 1)."""
import numpy as np

from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from oracle import oracle as O

CFG = dict(dim=256, n_layer=2, n_head=4, n_kv=2, head_dim=64, ffn=512, vocab=512, max_seq=48, theta=1e6, tied=True)


def _decode_all(om, toks):
    return [om.decode(int(t), p)[1].copy() for p, t in enumerate(toks)]


def test_vectorised_path_is_bit_identical_to_the_scalar_path():
    raw = synth.raw_weights_numpy(CFG, 3, w_std=0.1)
    toks = prompt_ids(CFG, 12)
    for lt, ht in ((L.Q4, L.BF16), (L.BOOL1, L.Q4), (L.F8E5M2, L.BF16), (O.F16, O.F16)):
        slow, fast = oracle_model(CFG, raw, lt, ht), oracle_model(CFG, raw, lt, ht)
        assert fast.prepare_fast() >= 0
        a, b = _decode_all(slow, toks), _decode_all(fast, toks)
        for p in range(len(toks)):
            assert np.array_equal(a[p], b[p]), "layers %d head %d: logits differ at position %d" % (lt, ht, p)
        ka, va = slow.kv()
        kb, vb = fast.kv()
        assert np.array_equal(ka[:, :len(toks)], kb[:, :len(toks)]) and np.array_equal(va[:, :len(toks)], vb[:, :len(toks)])


def test_vectorised_path_with_hot_masks():
    raw = synth.raw_weights_numpy(CFG, 4, w_std=0.1)
    slow, fast = oracle_model(CFG, raw, L.BOOL1, L.BF16), oracle_model(CFG, raw, L.BOOL1, L.BF16)
    fast.prepare_fast()
    hot = np.zeros(CFG["ffn"], dtype=np.int32)
    hot[::5] = 1
    for l in range(CFG["n_layer"]):
        slow.set_hot(l, hot), fast.set_hot(l, hot)
    for p, t in enumerate(prompt_ids(CFG, 6)):
        assert np.array_equal(slow.decode(int(t), p)[1], fast.decode(int(t), p)[1])


def test_half_weights_are_the_rounded_bf16_weights():
    rng = np.random.default_rng(0)
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(64, 128)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1.0, size=128).astype(np.float32))
    ow = O.quantize(w, 64, 128, O.F16)
    assert ow.data.dtype == np.uint16
    # bf16 -> half is exact for these magnitudes (8 significant bits, exponents inside the half range): the products equal the bf16 model's
    assert np.array_equal(O.linear(ow, x), O.linear(O.quantize(w, 64, 128, L.BF16), x))
    # and a half subnormal / overflow behaves like _cvtss_sh: round to nearest even, overflow to infinity
    tiny = O.f32_to_bf16(np.array([[1e-9] * 128], dtype=np.float32))
    assert (O.quantize(tiny, 1, 128, O.F16).data == 0).all()
