"""Pins the oracle's Packed128 layout against what the reference's own tree states:
the PACK_/UNPACK_ macros of src/PackedQ.hpp:99-239 and BIT_SET_k/BIT_GET_k of src/Utils/CLI_params.cpp:2177-2207.
The macro semantics are restated here a SECOND time, independently, in numpy (bit arithmetic on Python ints);
oracle/kf_oracle.c must agree with this restatement on random and on hand-built blocks."""
import numpy as np
import pytest


def np_pack(q, bits):
    """element i < per/2 -> high >> (64 - bits*(i+1)); i >= per/2 -> low likewise; memory = low (LE) then high (LE)."""
    per = 128 // bits
    q = np.asarray(q).reshape(-1, per)
    out = np.zeros((q.shape[0], 16), dtype=np.uint8)
    for b in range(q.shape[0]):
        high = low = 0
        for i in range(per // 2):
            sh = 64 - bits * (i + 1)
            high |= (int(q[b, i]) & ((1 << bits) - 1)) << sh
            low |= (int(q[b, i + per // 2]) & ((1 << bits) - 1)) << sh
        out[b, :8] = np.frombuffer(low.to_bytes(8, "little"), dtype=np.uint8)
        out[b, 8:] = np.frombuffer(high.to_bytes(8, "little"), dtype=np.uint8)
    return out.reshape(-1)


@pytest.mark.parametrize("bits", [4, 2, 1])
def test_pack_unpack_matches_independent_restatement(O, bits):
    rng = np.random.default_rng(bits)
    per = 128 // bits
    q = rng.integers(0, 1 << bits, size=per * 37).astype(np.int32)
    packed = O.pack(q, bits)
    assert np.array_equal(packed, np_pack(q, bits))
    assert np.array_equal(O.unpack(packed, bits), q)


def test_pack4_known_block(O):
    """element 0 is the TOP nibble of byte 15, element 16 the top nibble of byte 7 (PackedQ.hpp:104-136)."""
    q = np.arange(32, dtype=np.int32) % 16
    p = O.pack(q, 4)
    assert p[15] == (0 << 4 | 1) and p[14] == (2 << 4 | 3) and p[8] == (14 << 4 | 15)
    assert p[7] == (0 << 4 | 1) and p[0] == (14 << 4 | 15)
    one = np.zeros(32, dtype=np.int32)
    one[0] = 0xF
    assert O.pack(one, 4).tolist() == [0] * 15 + [0xF0]
    one[:] = 0
    one[31] = 0x9
    assert O.pack(one, 4).tolist() == [0x09] + [0] * 15


def test_pack2_pack1_known_blocks(O):
    q = np.zeros(64, dtype=np.int32)
    q[0] = 3
    assert O.pack(q, 2).tolist() == [0] * 15 + [0xC0]      # high >> 62
    q[:] = 0
    q[32] = 2
    assert O.pack(q, 2).tolist() == [0] * 7 + [0x80] + [0] * 8  # low >> 62
    b = np.zeros(128, dtype=np.int32)
    b[0] = 1
    assert O.pack(b, 1).tolist() == [0] * 15 + [0x80]      # high >> 63
    b[:] = 0
    b[127] = 1
    assert O.pack(b, 1).tolist() == [0x01] + [0] * 15       # low >> 0


@pytest.mark.parametrize("bits", [1, 2, 3, 4])
def test_bit_set_get_k(O, bits):
    import ctypes as C
    L = O.lib()
    rng = np.random.default_rng(bits + 10)
    n = 200
    lo, hi = (-1, 2) if bits == 2 else (0, (1 << bits) - 1)
    vals = rng.integers(lo, hi + 1, size=n)
    arr = np.zeros(n * bits // 8 + 8, dtype=np.uint8)
    for i, v in enumerate(vals):
        L.kfo_bit_set_k(arr.ctypes.data_as(C.c_void_p), C.c_size_t(i), int(v), bits)
    for i, v in enumerate(vals):
        assert L.kfo_bit_get_k(arr.ctypes.data_as(C.c_void_p), C.c_size_t(i), bits) == v
    # "LittleEndian!!! [5, 10, 9, 11, 12, 7, 9, 7] => 0x5A, 0x9B, 0xC7, 0x97" (CLI_params.cpp, comment above PrintQ_128)
    arr[:] = 0
    for i, v in enumerate([5, 10, 9, 11, 12, 7, 9, 7]):
        L.kfo_bit_set_k(arr.ctypes.data_as(C.c_void_p), C.c_size_t(i), v, 4)
    assert arr[:4].tolist() == [0x5A, 0x9B, 0xC7, 0x97]


def test_quant_ranges_follow_ctor(O):
    """GeQuant ctor, GeQuant.cpp:107-124"""
    assert O.quant_range(4, False, False) == (0, 15, 0)
    assert O.quant_range(4, True, False) == (-8, 7, 8)
    assert O.quant_range(2, False, False) == (0, 3, 0)
    assert O.quant_range(2, True, True) == (-1, 1, 1)    # ternary
    assert O.quant_range(1, False, True) == (0, 1, 0)    # 1-bit
