"""kf_adamw (CU_adamw_p on the device, reference launch geometry, seeded stochastic rounding) vs the oracle: bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

from koifish_amd import lib as L
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _t16(a, dev):
    return torch.from_numpy(a.view(np.int16).copy()).to(dev)


def _back(t):
    return t.cpu().numpy().view(np.uint16)


def _run(ctx, p, g, m, v, hp, status=None):
    mv = L.BF16 if m.dtype == torch.int16 else L.F32
    return ctx.hip.kf_adamw(ctx.h, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), mv, hp["lr"], hp["beta1"], hp["beta2"], hp["b1c"], hp["b2c"],
                            hp["eps"], hp["wd"], hp["grad_scale"], hp["seed"], status.data_ptr() if status is not None else None)


@pytest.mark.parametrize("n", [8, 4096, 512 * 8 * 5 + 72, 1600 * 6400])
@pytest.mark.parametrize("mv_bf16", [True, False])
def test_adamw_bit_exact(ctx, n, mv_bf16):
    rng = np.random.default_rng(n)
    p = O.f32_to_bf16(rng.normal(0, 0.05, n).astype(np.float32))
    g = O.f32_to_bf16(rng.normal(0, 0.01, n).astype(np.float32))
    m32 = rng.normal(0, 0.005, n).astype(np.float32)
    v32 = np.abs(rng.normal(0, 1e-4, n)).astype(np.float32)
    hp = dict(lr=3e-4, beta1=0.9, beta2=0.95, b1c=float(np.float32(1 - 0.9 ** 3)), b2c=float(np.float32(1 - 0.95 ** 3)), eps=1e-8, wd=0.1, grad_scale=0.25, seed=4242)
    if mv_bf16:
        m, v = O.f32_to_bf16(m32), O.f32_to_bf16(v32)
        dm, dv = _t16(m, ctx.device), _t16(v, ctx.device)
    else:
        m, v = m32.copy(), v32.copy()
        dm, dv = torch.from_numpy(m.copy()).to(ctx.device), torch.from_numpy(v.copy()).to(ctx.device)
    dp, dg = _t16(p, ctx.device), _t16(g, ctx.device)
    for step in range(2):          # two steps: the second starts from stochastic-rounded state, with a new seed
        hp["seed"] += step
        if step:
            g = O.f32_to_bf16(rng.normal(0, 0.01, n).astype(np.float32))
            dg = _t16(g, ctx.device)
        assert O.adamw(p, g, m, v, **hp) == 0
        assert _run(ctx, dp, dg, dm, dv, hp) == 0, ctx.hip.kf_last_error()
        ctx.sync()
        assert np.array_equal(_back(dp), p), "params differ at step %d" % step
        assert not _back(dg).any()
        if mv_bf16:
            assert np.array_equal(_back(dm), m) and np.array_equal(_back(dv), v)
        else:
            assert np.array_equal(dm.cpu().numpy(), m) and np.array_equal(dv.cpu().numpy(), v)


def test_adamw_nonfinite_and_bad_args(ctx):
    n = 64
    p = O.f32_to_bf16(np.linspace(-1, 1, n).astype(np.float32))
    p[19] = 0x7FC0
    g = O.f32_to_bf16(np.full(n, 0.1, np.float32))
    m, v = np.zeros(n, np.uint16), np.zeros(n, np.uint16)
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, b1c=0.1, b2c=0.001, eps=1e-8, wd=0.0, grad_scale=1.0, seed=7)
    dp, dg, dm, dv = (_t16(a, ctx.device) for a in (p, g, m, v))
    st = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    assert O.adamw(p, g, m, v, **hp) == -1
    assert _run(ctx, dp, dg, dm, dv, hp, st) == 0
    ctx.sync()
    assert int(st.item()) == -5100                                  # KOIFISH_ADAMW_MV
    assert np.array_equal(_back(dp), p) and np.array_equal(_back(dg), g)
    assert ctx.hip.kf_adamw(ctx.h, dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), 60, L.BF16, 1e-3, 0.9, 0.99, 1.0, 1.0, 1e-8, 0.0, 1.0, 1, None) == -20
    assert ctx.hip.kf_adamw(ctx.h, dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), 64, L.Q4, 1e-3, 0.9, 0.99, 1.0, 1.0, 1e-8, 0.0, 1.0, 1, None) == -1000
