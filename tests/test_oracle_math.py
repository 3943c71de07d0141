"""Number formats of the oracle: bf16 RNE, f8e5m2 (top byte of a half), the fixed exp."""
import math

import numpy as np


def test_bf16_round_to_nearest_even(O):
    f = np.array([1.0, 1.00390625, 1.005859375, 1.01171875, -2.5, 3.4e38, 1e-40, 0.0, -0.0], dtype=np.float32)
    h = O.f32_to_bf16(f)
    assert h[0] == 0x3F80
    assert h[1] == 0x3F80          # 1 + 2^-8 is a tie -> even (0x3F80)
    assert h[2] == 0x3F81          # just above the tie
    assert h[3] == 0x3F82          # 1 + 3*2^-8: tie -> even (0x3F82)
    assert np.array_equal(O.bf16_to_f32(h[[0, 4]]), np.array([1.0, -2.5], dtype=np.float32))
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, 100000).astype(np.float32)
    back = O.bf16_to_f32(O.f32_to_bf16(x))
    assert np.all(np.abs(back - x) <= np.abs(x) * 2.0 ** -8)


def test_f8e5m2_is_the_high_byte_of_a_half(O):
    import ctypes as C
    L = O.lib()
    b = np.arange(256, dtype=np.uint8)
    out = np.zeros(256, dtype=np.uint16)
    L.kfo_f8e5m2_to_bf16(b.ctypes.data_as(C.c_void_p), C.c_size_t(256), out.ctypes.data_as(C.c_void_p))
    ref = (b.astype(np.uint16) << 8).view(np.float16).astype(np.float32)
    got = O.bf16_to_f32(out)
    fin = np.isfinite(ref)
    assert np.array_equal(got[fin], ref[fin])      # every e5m2 value is exact in bf16
    back = np.zeros(256, dtype=np.uint8)
    L.kfo_bf16_to_f8e5m2(out.ctypes.data_as(C.c_void_p), C.c_size_t(256), back.ctypes.data_as(C.c_void_p))
    assert np.array_equal(back[fin], b[fin])
    # Float2T<f8e5> truncates the half's low byte (g_float.hpp:433-443): 1.124 -> half 0x3C7F -> 0x3C
    v = O.f32_to_bf16(np.array([1.124], dtype=np.float32))
    one = np.zeros(1, dtype=np.uint8)
    L.kfo_bf16_to_f8e5m2(v.ctypes.data_as(C.c_void_p), C.c_size_t(1), one.ctypes.data_as(C.c_void_p))
    assert one[0] == 0x3C


def test_expf_within_2ulp_of_libm(O):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-87, 88, 20000), rng.normal(0, 3, 20000), [0.0, -0.0, 1.0, -1.0, 88.7, -87.3, -100.0, 100.0]]).astype(np.float32)
    got = O.expf(xs)
    for x, g in zip(xs[:3000].tolist() + xs[-8:].tolist(), got[:3000].tolist() + got[-8:].tolist()):
        ref = math.exp(x) if x < 88.72283 else float("inf")
        if x < -87.33654:
            assert g == 0.0
            continue
        if math.isinf(ref) or ref > 3.4028234e38:
            assert math.isinf(g)
            continue
        ulp = np.spacing(np.float32(ref))
        assert abs(g - ref) <= 2 * ulp, (x, g, ref)
    assert O.expf([0.0])[0] == 1.0
