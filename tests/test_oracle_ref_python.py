"""The oracle against outputs of the REFERENCE ITSELF: tests/golden/ref_python_vectors.npz holds inputs and outputs of four functions of the reference's
own Python files, executed in the build container by tests/golden/make_ref_python_vectors.py (AutoAWQ unpack / order / dequant of
src/Python/test_awq.py, the attention `ref_program` of src/Python/tile_wrapper/tl_qkv.py, the RMS `ref_program` of tile_wrapper/tl_norm.py, the product `ref_program` of tile_wrapper/tl_gemm.py).  These pin
SURVEY 8a rows a7 (bit order exactly, values to the bf16 rounding of the script's fp16 product), a13 (score scale, causal mask, which kv head a query head
reads, softmax, PV) a9 (epsilon inside the root, mean over the row) and a8 (y = x . W^T: the contracted index and the transposed operand) to the reference's statements rather than to our reading of its CUDA."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python_vectors.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_awq_bit_order_and_dequant_follow_the_reference_script(O, gold):
    qw, qz, sc = gold["awq_qweight"], gold["awq_qzeros"], gold["awq_scales"]
    n_in, n_out = qw.shape[0], qw.shape[1] * 8
    # the script's integers in natural column order, packed back by the oracle's AWQ order, are the checkpoint words: the nibble order is pinned exactly
    assert np.array_equal(O.awq_pack(gold["awq_iweight"].astype(np.uint32)), qw.view(np.uint32))
    assert np.array_equal(O.awq_pack(gold["awq_izeros"].astype(np.uint32)), qz.view(np.uint32))
    w = O.AWQWeight(n_out, n_in, qw.view(np.uint32), qz.view(np.uint32), sc.view(np.float16))
    got = O.bf16_to_f32(O.dequant_awq(w))
    ref = O.bf16_to_f32(gold["awq_dequant"])
    # (q - z) * s: the script rounds the product to fp16 and then to bf16, CU_Q42X_awq (the path, GeQuant.cpp:410) rounds once: at most one bf16 ulp apart, mostly equal
    assert np.abs(got - ref).max() <= 2.0 ** -7 * np.abs(ref).max()
    assert (got != ref).mean() < 0.08          # the double rounding (fp16, then bf16) moves a few per cent of the values by one ulp
    # and the exact integers times the exact scales agree with both
    exact = (gold["awq_iweight"].astype(np.float64) - np.repeat(gold["awq_izeros"].astype(np.float64), 128, axis=0)) * np.repeat(sc.view(np.float16).astype(np.float64), 128, axis=0)
    assert np.abs(got - exact).max() <= 2.0 ** -8 * np.abs(exact).max()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("mode", ["REF", "FUSED"])
def test_attention_follows_the_reference_program(O, gold, tag, mode):
    q, k, v, ref = gold["att_%s_q" % tag], gold["att_%s_k" % tag], gold["att_%s_v" % tag], gold["att_%s_out" % tag]
    T, HQ, D = q.shape
    HK = k.shape[1]
    kc, vc = np.ascontiguousarray(k.reshape(T, HK * D)), np.ascontiguousarray(v.reshape(T, HK * D))
    scale = np.abs(ref).max()
    for t in range(T):
        out = O.bf16_to_f32(O.attn_decode(np.ascontiguousarray(q[t].reshape(-1)), kc, vc, t, HQ, HK, D, mode=getattr(O, "ATTN_" + mode))).reshape(HQ, D)
        # bf16 scores (and, in REF mode, bf16 probabilities) against the program's fp32: 2^-6 of the output scale (SURVEY 8c's attention tolerance)
        assert np.abs(out - ref[t]).max() <= 2.0 ** -6 * scale, "t=%d" % t
    # a wrong head -> kv-head map (h % HK instead of h // groups) would be an O(1) error whenever the two differ
    if HQ // HK > 1 and HK > 1:
        wrong = ref[:, [(h % HK) * (HQ // HK) for h in range(HQ)], :]
        assert np.abs(wrong - ref).max() > 0.1 * scale


def test_rmsnorm_follows_the_reference_program(O, gold):
    x, ref = gold["rms_x"], gold["rms_out"]
    ones = np.full(x.shape[1], 0x3F80, dtype=np.uint16)
    got = O.bf16_to_f32(O.rmsnorm(x, ones, eps=1e-12))
    assert np.abs(got - ref).max() <= 2.0 ** -8 * np.abs(ref).max()          # one bf16 rounding of the output
    assert np.array_equal(O.f32_to_bf16(ref), O.rmsnorm(x, ones, eps=1e-12)) or (O.f32_to_bf16(ref) != O.rmsnorm(x, ones, eps=1e-12)).mean() < 0.01


@pytest.mark.parametrize("tag", ["a", "b"])
def test_linear_follows_the_reference_program(O, gold, tag):
    """SLP::Forw's product as the reference states it (tl_gemm.py:150-151, C = A @ B.T, checked there to rtol = atol = 1e-2): the oracle's mat-vec on bf16 weights, row
    by row of x -- which operand is transposed, which index is contracted -- within one bf16 rounding of the program's fp32 result (row a8)."""
    x, w, ref = gold["gemm_%s_x" % tag], gold["gemm_%s_w" % tag], gold["gemm_%s_out" % tag]
    M, K = w.shape
    ow = O.quantize(w, M, K, O.BF16)
    for t in range(x.shape[0]):
        got = O.bf16_to_f32(O.linear(ow, x[t]))
        assert np.abs(got - ref[t]).max() <= 2.0 ** -8 * np.abs(ref[t]).max() + 1e-6, "row %d" % t
        assert np.allclose(got, ref[t], rtol=1e-2, atol=1e-2)      # the reference's own bound
    # the transposed reading (x . W instead of x . W^T) is a different shape altogether; a swapped contraction index on a square slice would be an O(1) error
    sq = min(M, K)
    f = lambda a: O.bf16_to_f32(a).astype(np.float64)
    wrong = f(x[0][:sq]) @ f(w[:sq, :sq])
    right = f(x[0][:sq]) @ f(w[:sq, :sq]).T
    assert np.abs(wrong - right).max() > 0.1 * np.abs(right).max()
