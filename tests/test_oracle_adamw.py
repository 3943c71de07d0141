"""The oracle's restatement of CU_adamw_p (Optimizer.cu:393-442) + CU_Float2T<bf16> stochastic rounding (packedN.cuh:62-72) against an
independent numpy statement of the same formulas."""
import numpy as np

from oracle import oracle as O


def _noise(x, y, seed):
    m = 0xFFFFFFFF
    b = (x + 198491317 * y) & m
    b = (b * 0xd2a80a3f) & m
    b = (b + seed) & m
    b ^= b >> 9
    b = (b + 0xa884f197) & m
    b ^= b >> 11
    b = (b * 0x6C736F4B) & m
    b ^= b >> 13
    b = (b + 0xB79F3ABB) & m
    b ^= b >> 15
    b = (b * 0x1b56c4f5) & m
    b ^= b >> 17
    return b


def _stoch(a32, thr):
    u = np.float32(a32).view(np.uint32)
    u = np.where((u & 0xFFFF) > thr, u | 0xFFFF, u & ~np.uint32(0xFFFF)).astype(np.uint32)
    return O.f32_to_bf16(u.view(np.float32))


def test_adamw_matches_numpy_statement():
    rng = np.random.default_rng(0)
    n = 512 * 8 * 3 + 64
    p = O.f32_to_bf16(rng.normal(0, 0.05, n).astype(np.float32))
    g = O.f32_to_bf16(rng.normal(0, 0.01, n).astype(np.float32))
    m = O.f32_to_bf16(rng.normal(0, 0.005, n).astype(np.float32))
    v = O.f32_to_bf16(np.abs(rng.normal(0, 1e-4, n)).astype(np.float32))
    hp = dict(lr=np.float32(3e-4), beta1=np.float32(0.9), beta2=np.float32(0.95), b1c=np.float32(1 - 0.9 ** 7), b2c=np.float32(1 - 0.95 ** 7),
              eps=np.float32(1e-8), wd=np.float32(0.1), grad_scale=np.float32(0.5), seed=12345)
    p2, g2, m2, v2 = p.copy(), g.copy(), m.copy(), v.copy()
    assert O.adamw(p2, g2, m2, v2, **{k: (float(x) if k != "seed" else x) for k, x in hp.items()}) == 0
    # numpy, elementwise in fp32 (fma written with float64 intermediates: exact product, one rounding)
    f = lambda a: O.bf16_to_f32(a)
    def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    G = (hp["grad_scale"] * f(g)).astype(np.float32)
    M = fma(np.full(n, hp["beta1"], np.float32), f(m), fma(np.full(n, -hp["beta1"], np.float32), G, G))
    G2 = (G * G).astype(np.float32)
    V = fma(np.full(n, hp["beta2"], np.float32), f(v), fma(np.full(n, -hp["beta2"], np.float32), G2, G2))
    step = ((M / hp["b1c"]).astype(np.float32) / (np.sqrt((V / hp["b2c"]).astype(np.float32)).astype(np.float32) + hp["eps"]).astype(np.float32)).astype(np.float32)
    old = f(p)
    P = ((old - (hp["lr"] * hp["wd"]).astype(np.float32) * old).astype(np.float32) - (hp["lr"] * step).astype(np.float32)).astype(np.float32)
    t = np.arange(n) // 8
    thr = np.array([_noise(int(tt % 512), int(tt // 512) * 512, hp["seed"]) & 0xFFFF for tt in t[::8]], dtype=np.uint32).repeat(8)
    # fma double rounding can differ from a true fma in rare cases: allow the numpy statement 1 ulp of fp32 slack by comparing the bf16 results loosely
    for got, want in ((p2, _stoch(P, thr)), (m2, _stoch(M, thr)), (v2, _stoch(V, thr))):
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
    assert not g2.any()


def test_adamw_stochastic_rounding_is_unbiased_and_fp32_moments():
    n = 512 * 8 * 8
    p = np.full(n, O.f32_to_bf16(np.float32([1.0]))[0], dtype=np.uint16)
    g = np.full(n, O.f32_to_bf16(np.float32([1.0]))[0], dtype=np.uint16)
    m = np.zeros(n, dtype=np.float32)
    v = np.zeros(n, dtype=np.float32)
    # lr tiny: p moves by far less than a bf16 ulp (2^-7 at 1.0); round-to-nearest would leave every parameter at 1.0
    assert O.adamw(p, g, m, v, 1e-3, 0.9, 0.999, 0.1, 0.001, 1e-8, 0.0, 1.0, 7) == 0
    vals = O.bf16_to_f32(p)
    moved = (vals != 1.0).mean()
    expect = 1e-3 / 2.0 ** -8          # below 1.0 the bf16 spacing is 2^-8: P(round down) = delta / spacing
    assert 0.5 * expect < moved < 1.5 * expect, (moved, expect)
    assert np.allclose(m, 0.1, rtol=1e-6) and np.allclose(v, 0.001, rtol=1e-5)


def test_adamw_nonfinite_thread_stores_nothing():
    n = 64
    p = O.f32_to_bf16(np.linspace(-1, 1, n).astype(np.float32))
    p[19] = 0x7FC0   # NaN in thread 2
    g = O.f32_to_bf16(np.full(n, 0.1, np.float32))
    m, v = np.zeros(n, np.uint16), np.zeros(n, np.uint16)
    p0 = p.copy()
    assert O.adamw(p, g, m, v, 1e-2, 0.9, 0.999, 0.1, 0.001, 1e-8, 0.0, 1.0, 7) == -1
    assert np.array_equal(p[16:24], p0[16:24]) and g[16:24].all() and not m[16:24].any()
    assert not g[:16].any() and not np.array_equal(p[:16], p0[:16])
