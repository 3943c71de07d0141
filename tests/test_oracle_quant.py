"""GeQuant::RTN_x / YinYang and CU_Q128toX_ restatements: an independent numpy restatement must agree bit for bit,
and the documented error bounds of the quantiser must hold."""
import numpy as np
import pytest

from oracle import oracle as OO


def np_round_bf16(x):
    return OO.bf16_to_f32(OO.f32_to_bf16(x))


def np_rtn_x(w_u16, lGroup, bits, symmetric):
    """GeQuant.cpp:428-533 in numpy (float32 arithmetic, std::round = half away from zero)."""
    a = OO.bf16_to_f32(w_u16).reshape(-1, lGroup)
    qMin, qMax, qBias = OO.quant_range(bits, symmetric, False)
    vmax, vmin = a.max(1), a.min(1)
    if symmetric:
        step = (np.maximum(np.abs(vmax), np.abs(vmin)) / np.float32(qMax)).astype(np.float32)
        zero = np.zeros_like(step)
    else:
        step = ((vmax - vmin) / np.float32(qMax - qMin)).astype(np.float32)
        zero = (-vmin).astype(np.float32)
    t = ((a + zero[:, None]).astype(np.float32) / step[:, None]).astype(np.float32)
    q = np.where(t >= 0, np.floor(t + np.float32(0.5)), np.ceil(t - np.float32(0.5))).astype(np.int32)
    q = np.clip(q, qMin, qMax) + qBias
    return q, OO.f32_to_bf16(zero), OO.f32_to_bf16(step), qBias


def np_dequant(q, zero_u16, step_u16, qBias):
    """T.cu:274 with bf16 operators: bf16(bf16(step * bf16(q - qBias)) - zero)"""
    z, s = OO.bf16_to_f32(zero_u16)[:, None], OO.bf16_to_f32(step_u16)[:, None]
    t = np_round_bf16((s * (q - qBias).astype(np.float32)).astype(np.float32))
    return OO.f32_to_bf16((t - z).astype(np.float32))


@pytest.mark.parametrize("bits,symmetric", [(4, False), (4, True), (2, False)])
def test_rtn_matches_numpy_restatement(O, bits, symmetric):
    rng = np.random.default_rng(bits * 2 + symmetric)
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(64, 512)).astype(np.float32))
    t = O.Q4 if bits == 4 else O.Q2
    import ctypes as C
    nG = w.size // 128
    packed = np.zeros(w.size * bits // 8, dtype=np.uint8)
    zero = np.zeros(nG, dtype=np.uint16)
    step = np.zeros(nG, dtype=np.uint16)
    O.lib().kfo_rtn_x(w.ctypes.data_as(C.c_void_p), C.c_size_t(nG), 128, bits, int(symmetric), 0, packed.ctypes.data_as(C.c_void_p),
                      zero.ctypes.data_as(C.c_void_p), step.ctypes.data_as(C.c_void_p))
    q, z, s, qb = np_rtn_x(w, 128, bits, symmetric)
    assert np.array_equal(z, zero) and np.array_equal(s, step)
    assert np.array_equal(O.unpack(packed, bits).reshape(-1, 128), q)
    deq = O.dequant_q128(packed, zero, step, 128, bits, qb)
    assert np.array_equal(deq.reshape(-1, 128), np_dequant(q, z, s, qb))
    # quantisation error: at most half a step plus the bf16 representation of zero/step and of the result
    err = np.abs(O.bf16_to_f32(deq).reshape(-1, 128) - O.bf16_to_f32(w).reshape(-1, 128))
    stepf = O.bf16_to_f32(step)[:, None]
    rng_g = np.abs(O.bf16_to_f32(w).reshape(-1, 128)).max(1)[:, None]
    assert np.all(err <= 0.5 * stepf + 2.0 ** -6 * rng_g + 1e-7)


def test_yinyang_ternary_and_binary(O):
    rng = np.random.default_rng(5)
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(32, 256)).astype(np.float32))
    a = O.bf16_to_f32(w).reshape(-1, 128)
    ref_step = np.maximum(np.float32(1e-5), np.sqrt((np.where(a < 0, 0, a).astype(np.float32) ** 2).astype(np.float64).mean(1))).astype(np.float32)
    for t, levels in ((O.T_SIGN, {-1, 0, 1}), (O.BOOL1, {0, 1})):
        qw = O.quantize(w, 32, 256, t)
        assert np.array_equal(qw.step, O.f32_to_bf16(ref_step))
        assert not qw.zero.any()
        q = O.unpack(qw.data, qw.bits).reshape(-1, 128) - qw.qBias
        assert set(np.unique(q).tolist()) <= levels
        expect = np.clip(np.where(a / ref_step[:, None] >= 0, np.floor(a / ref_step[:, None] + 0.5), np.ceil(a / ref_step[:, None] - 0.5)), min(levels), max(levels))
        assert np.array_equal(q, expect.astype(np.int32))
        # dequant of a zero-`zero` group is exactly q*step
        d = O.bf16_to_f32(O.dequant(qw)).reshape(-1, 128)
        assert np.array_equal(d, (q * O.bf16_to_f32(qw.step)[:, None]).astype(np.float32))


def test_blob_layout_is_data_then_gama(O):
    """GTensor::gama_T (GTensor.cpp:456-510): [R_SCALE ne0][C_SCALE ne1][ZERO nGroup][STEP nGroup] bf16 after the packed data"""
    rng = np.random.default_rng(2)
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(16, 256)).astype(np.float32))
    qw = O.quantize(w, 16, 256, O.Q4)
    blob = qw.blob()
    szData = 16 * 256 // 2
    assert blob.size == szData + (16 + 256 + 2 * qw.nGroup) * 2     # szGama of GeQuant.cpp:518
    g = blob[szData:].view(np.uint16)
    assert np.array_equal(blob[:szData], qw.data)
    assert np.array_equal(g[16 + 256:16 + 256 + qw.nGroup], qw.zero)
    assert np.array_equal(g[16 + 256 + qw.nGroup:], qw.step)


def test_linear_dot_order_is_the_16_lane_order(O):
    """dotprod_fp16 (GST_float.cpp:75-101): two 8-lane accumulators, mul then add, folded 16->8->4->(0+1)+(2+3)"""
    rng = np.random.default_rng(3)
    k = 256
    w = O.f32_to_bf16(rng.normal(0, 1, size=(4, k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1, size=k).astype(np.float32))
    wf, xf = O.bf16_to_f32(w), O.bf16_to_f32(x)
    got = O.linear_f32(O.QWeight(O.BF16, 4, k, w), x)
    for r in range(4):
        s = np.zeros(16, dtype=np.float32)
        for j in range(0, k, 16):
            s = (xf[j:j + 16] * wf[r, j:j + 16]).astype(np.float32) + s
        s8 = s[:8] + s[8:]
        s4 = s8[:4] + s8[4:]
        ref = np.float32(np.float32(s4[0] + s4[1]) + np.float32(s4[2] + s4[3]))
        assert got[r] == ref


def test_awq_unpack_matches_the_reference_script_semantics(O):
    """src/Python/test_awq.py:32-66 (unpack_awq + reverse_awq_order) restated in numpy: shifts [0,4,..,28], columns re-ordered by
    AWQ_REVERSE_ORDER = [0,4,1,5,2,6,3,7]; packing uses AWQ_ORDER = [0,2,4,6,1,3,5,7]; dequant (q - z) * scale -> bf16
    (CU_Q42X_awq, quantizer.cu:131-156)."""
    rng = np.random.default_rng(17)
    n_in, n_out = 256, 64
    q = rng.integers(0, 16, size=(n_in, n_out))
    z = rng.integers(0, 16, size=(n_in // 128, n_out))
    s = rng.uniform(0.003, 0.012, size=(n_in // 128, n_out)).astype(np.float16)
    w = O.AWQWeight(n_out, n_in, O.awq_pack(q), O.awq_pack(z), s)
    shifts = np.arange(0, 32, 4)
    rev = np.arange(n_out).reshape(-1, 8)[:, [0, 4, 1, 5, 2, 6, 3, 7]].reshape(-1)
    iw = ((w.data[:, :, None] >> shifts[None, None, :]) & 0xF).reshape(n_in, -1)[:, rev]
    iz = ((w.qzeros[:, :, None] >> shifts[None, None, :]) & 0xF).reshape(n_in // 128, -1)[:, rev]
    assert np.array_equal(iw, q) and np.array_equal(iz, z)
    # a word whose elements are 0..7 packs to 0x75316420 (element k at nibble AWQ_REVERSE_ORDER[k])
    assert O.awq_pack(np.arange(8))[0] == 0x75316420
    ref = (q - np.repeat(z, 128, axis=0)).astype(np.float32) * np.repeat(s.astype(np.float32), 128, axis=0)
    assert np.array_equal(O.dequant_awq(w), O.f32_to_bf16(ref).reshape(n_in, n_out))
    x = O.f32_to_bf16(rng.normal(0, 1, size=n_in).astype(np.float32))
    exact = O.bf16_to_f32(O.dequant_awq(w)).astype(np.float64).T @ O.bf16_to_f32(x).astype(np.float64)
    assert np.abs(O.bf16_to_f32(O.linear(w, x)) - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 1e-6
