"""Row-codebook 4-bit storage (GeQuant::RT_NormalF / QUANT_MODE::RTNf): the oracle against the reference's own literals and
against an independent numpy restatement.

What the reference's tree pins for this storage: the NF4 / NF3 tables and the NF4 midpoints (src/g_float.hpp:543-569 -- written
here as data, the way the reference's headers spell them) and the BIT_SET_k / BIT_GET_k bit order (CLI_params.cpp:2177-2207,
covered in test_oracle_layout.py).  The quantiser arithmetic itself has no golden vector in the reference: the numpy restatement
below follows GeQuant.cpp:641-755 independently of oracle/kf_oracle.c and the two must agree bit for bit.
"""
import numpy as np
import pytest

from oracle import oracle as O

# NF4_LUT::table, NF4_LUT::mids, NF3_LUT::table as printed in src/g_float.hpp:543-569
NF4 = [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
       0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0]
NF4_MIDS = [-0.8480964004993438, -0.6106329262256622, -0.4599952697753906, -0.33967943489551544, -0.23460715460772705, -0.13791173315048218,
            -0.045525018125772475, 0.03979014977812767, 0.1202552504837513, 0.20352124667733002, 0.2920137718319893, 0.3893125355243683, 0.5016634166240692,
            0.6427869200706482, 0.8614784181118011]
NF3 = [-1.0, -0.5350227355957031, -0.2469314038753510, 0.0, 0.1833375245332718, 0.3819939494132996, 0.6229856610298157, 1.0]


def test_tables_are_the_reference_literals():
    t4, t3 = O.nf4_table(), O.nf3_table()
    assert np.array_equal(t4, np.array(NF4, dtype=np.float32))
    assert np.array_equal(t3, np.array(NF3, dtype=np.float32))
    assert np.all(np.diff(t4) > 0) and np.all(np.diff(t3) > 0)
    # the reference's own midpoint table is the mean of neighbouring entries: a second, independent literal for the same 16 numbers
    mids = (np.array(NF4[:-1], dtype=np.float64) + np.array(NF4[1:], dtype=np.float64)) / 2
    assert np.abs(mids - np.array(NF4_MIDS)).max() < 1e-6   # the header's 5th midpoint is 2.5e-7 off the mean of its neighbours


def numpy_nf4(w_u16, ne0, ne1):
    """GeQuant::_row_lut + Distri_PIPE::Prepare / X2NormalF, restated with numpy (fp32 arithmetic spelled with np.float32)."""
    w = O.bf16_to_f32(w_u16).reshape(ne0, ne1)
    table = np.array(NF4, dtype=np.float32)
    packed = np.zeros((ne0, ne1 // 2), dtype=np.uint8)
    lut = np.zeros((ne0, 16), dtype=np.uint16)
    for r in range(ne0):
        abs_max = max(abs(np.float32(w[r].min())), abs(np.float32(w[r].max())))
        scale = np.float32(1.0 / np.float64(abs_max)) if abs_max > 0 else np.float32(1.0)
        cb = (table / scale).astype(np.float32)
        lut[r] = O.f32_to_bf16(cb)
        dist = np.abs(w[r][:, None] - cb[None, :]).astype(np.float32)
        idx = np.argmin(dist, axis=1).astype(np.uint8)     # first minimum, as the strict `<` scan
        packed[r] = (idx[0::2] << 4) | idx[1::2]           # BIT_SET_k: even element in the high nibble
    return packed.reshape(-1), lut


@pytest.mark.parametrize("shape,std,seed", [((16, 64), 0.02, 1), ((8, 1024), 1.0, 2), ((33, 96), 0.3, 3)])
def test_quantiser_matches_numpy_restatement(shape, std, seed):
    rng = np.random.default_rng(seed)
    m, k = shape
    w = O.f32_to_bf16(rng.normal(0, std, size=(m, k)).astype(np.float32))
    q = O.quantize_nf4(w, m, k)
    packed, lut = numpy_nf4(w, m, k)
    assert np.array_equal(q.lut, lut)
    assert np.array_equal(q.data, packed)
    # dequant = table entry of the row
    nib = np.stack([packed.reshape(m, -1) >> 4, packed.reshape(m, -1) & 15], axis=-1).reshape(m, k)
    assert np.array_equal(O.dequant(q), np.take_along_axis(lut, nib.astype(np.int64), axis=1))


def test_bit_stream_is_bit_set_k():
    import ctypes as C
    rng = np.random.default_rng(5)
    w = O.f32_to_bf16(rng.normal(0, 1, size=(4, 32)).astype(np.float32))
    q = O.quantize_nf4(w, 4, 32)
    L = O.lib()
    ids = [L.kfo_bit_get_k(q.data.ctypes.data_as(C.c_void_p), C.c_size_t(i), 4) for i in range(4 * 32)]
    assert ids == [int(b >> 4) if i % 2 == 0 else int(b & 15) for i, b in enumerate(np.repeat(q.data, 2))]


def test_edges_zero_rows_extremes_and_error_bound():
    rng = np.random.default_rng(7)
    m, k = 6, 128
    wf = rng.normal(0, 0.05, size=(m, k)).astype(np.float32)
    wf[2] = 0.0                     # all-zero row: scale falls back to 1, every element takes entry 7 (0.0)
    wf[3, 5] = -0.75                # the extreme of a row is negative: it must come back exactly (entry 0 = -abs_max)
    wf[4, 9] = 0.5                  # ... or positive (entry 15)
    w = O.f32_to_bf16(wf)
    q, err = O.quantize_nf4(w, m, k, want_err=True)
    d = O.bf16_to_f32(O.dequant(q))
    x = O.bf16_to_f32(w).reshape(m, k)
    assert np.all(d[2] == 0.0) and np.all((q.data.reshape(m, -1)[2] == 0x77))
    assert d[3, 5] == -0.75 and d[4, 9] == 0.5
    assert np.array_equal(O.bf16_to_f32(q.lut[2]), O.bf16_to_f32(O.f32_to_bf16(np.array(NF4, dtype=np.float32))))
    # nearest-entry quantisation: the error of an element is at most half the widest gap of the row's table (+ the table's bf16 rounding)
    amax = np.abs(x).max(axis=1, keepdims=True)
    gap = np.diff(np.array(NF4)).max() / 2
    assert np.all(np.abs(d - x) <= gap * amax * (1 + 2.0 ** -7) + 1e-12)
    assert 0 < err < 0.05
    # blob layout: nibble stream, then bf16 [R ne0][C ne1][LUT ne0 x 16]
    blob = q.blob()
    assert blob.size == m * k // 2 + 2 * (m + k + 16 * m)
    assert np.array_equal(blob[m * k // 2 + 2 * (m + k):].view(np.uint16), q.lut.reshape(-1))


def test_linear_and_embed_read_the_table():
    rng = np.random.default_rng(9)
    m, k = 24, 256
    w = O.f32_to_bf16(rng.normal(0, 0.02, size=(m, k)).astype(np.float32))
    x = O.f32_to_bf16(rng.normal(0, 1, size=k).astype(np.float32))
    q = O.quantize_nf4(w, m, k)
    dq = O.QWeight(O.BF16, m, k, O.dequant(q))
    assert np.array_equal(O.linear(q, x), O.linear(dq, x))
    assert np.array_equal(O.embed(q, 5), O.dequant(q)[5])


def numpy_nf3(w_u16, ne0, ne1):
    """the 3-bit form: 8-entry table, 8 ids -> 3 bytes, most significant bit first"""
    w = O.bf16_to_f32(w_u16).reshape(ne0, ne1)
    table = np.array(NF3, dtype=np.float32)
    packed = np.zeros((ne0, ne1 // 8, 3), dtype=np.uint8)
    lut = np.zeros((ne0, 8), dtype=np.uint16)
    for r in range(ne0):
        abs_max = max(abs(np.float32(w[r].min())), abs(np.float32(w[r].max())))
        scale = np.float32(1.0 / np.float64(abs_max)) if abs_max > 0 else np.float32(1.0)
        cb = (table / scale).astype(np.float32)
        lut[r] = O.f32_to_bf16(cb)
        idx = np.argmin(np.abs(w[r][:, None] - cb[None, :]).astype(np.float32), axis=1).astype(np.uint32).reshape(-1, 8)
        v = np.zeros(ne1 // 8, dtype=np.uint32)
        for h in range(8):
            v = (v << 3) | idx[:, h]
        packed[r, :, 0], packed[r, :, 1], packed[r, :, 2] = (v >> 16) & 255, (v >> 8) & 255, v & 255
    return packed.reshape(-1), lut


def unpack_ids(data, ne0, ne1, bits):
    """MSB-first ids of a `bits`-wide stream, as CU_Q32X_* / CU_Q22X_* extract them (quantizer.cu:672-675, 727-731)"""
    b = np.unpackbits(np.asarray(data, dtype=np.uint8)).reshape(ne0 * ne1, bits)
    return (b * (1 << np.arange(bits - 1, -1, -1))).sum(axis=1).reshape(ne0, ne1)


@pytest.mark.parametrize("shape,std,seed", [((16, 64), 0.02, 11), ((5, 1024), 1.0, 12)])
def test_nf3_quantiser_matches_numpy_restatement(shape, std, seed):
    rng = np.random.default_rng(seed)
    m, k = shape
    w = O.f32_to_bf16(rng.normal(0, std, size=(m, k)).astype(np.float32))
    q = O.quantize_nf3(w, m, k)
    packed, lut = numpy_nf3(w, m, k)
    assert np.array_equal(q.lut, lut) and np.array_equal(q.data, packed)
    ids = unpack_ids(q.data, m, k, 3)
    assert np.array_equal(O.dequant(q), np.take_along_axis(lut, ids, axis=1))
    x = O.f32_to_bf16(rng.normal(0, 1, size=k).astype(np.float32))
    assert np.array_equal(O.linear(q, x), O.linear(O.QWeight(O.BF16, m, k, O.dequant(q)), x))


def test_two_bit_row_forms():
    """CU_Q22X_ (4-entry table per row) and CU_Q22X_RTN (zero + step * id per row, bf16 operators) over a raw 2-bit stream"""
    rng = np.random.default_rng(13)
    m, k = 6, 64
    ids = rng.integers(0, 4, size=(m, k)).astype(np.uint8)
    data = np.packbits(((ids[..., None] >> np.array([1, 0])) & 1).astype(np.uint8).reshape(-1))
    assert np.array_equal(unpack_ids(data, m, k, 2), ids)
    lut = O.f32_to_bf16(rng.normal(0, 0.1, size=(m, 4)).astype(np.float32))
    q = O.LutWeight(m, k, data, lut, bits=2)
    assert np.array_equal(O.dequant(q), np.take_along_axis(lut, ids.astype(np.int64), axis=1))
    zs = O.f32_to_bf16(np.stack([rng.normal(-0.1, 0.02, size=m), rng.uniform(0.03, 0.08, size=m)], axis=1).astype(np.float32))
    r = O.LutWeight(m, k, data, zs, bits=2, rtn=True)
    z, s = O.bf16_to_f32(zs[:, 0])[:, None], O.bf16_to_f32(zs[:, 1])[:, None]
    step_id = O.bf16_to_f32(O.f32_to_bf16((s * ids.astype(np.float32)).astype(np.float32)))
    assert np.array_equal(O.dequant(r), O.f32_to_bf16((z + step_id).astype(np.float32)))
