import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--kf-slow", action="store_true", default=False, help="also run the tests marked `slow` (the long forms: full-depth TP, 4096-position runs, every sequence of the widest launches)")
    parser.addoption("--kf-shipped-order", action="store_true", default=False,
                     help="leave the library's own default summation order (canonical) in place for the whole suite instead of starting contexts in the v_dot2c order")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long form of a GPU test whose shortened variant runs by default (run with --kf-slow; the default `-m gpu` run stays inside the driver's step limit)")
    config.addinivalue_line("markers", "fast_order: written against the v_dot2c order on BOTH sides (skipped under --kf-shipped-order)")
    if config.getoption("--kf-shipped-order"):   # the whole suite on what ships by default (run: 630 pass, the fast_order ones skipped)
        return
    # The library's default summation order is the canonical one (round 4).  The tolerance tests of this suite were written against the v_dot2c order and keep
    # covering it: contexts and models built through koifish_amd.runtime start in that order here; every bit-exact test switches the canonical order on itself
    # (set_canonical(True)), and tests/test_gpu_canonical.py::test_library_default_is_the_canonical_order checks the untouched default.
    from koifish_amd import runtime
    runtime.DEFAULT_CANONICAL = False


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if not config.getoption("--kf-slow"):
        sk_slow = pytest.mark.skip(reason="long form: run with --kf-slow (its shortened variant runs by default)")
        for it in items:
            if "slow" in it.keywords:
                it.add_marker(sk_slow)
    if config.getoption("--kf-shipped-order"):
        sk = pytest.mark.skip(reason="compares two v_dot2c-order computations (or a v_dot2c launch with the oracle's dot16 order at its tolerance): not meaningful in the canonical order")
        for it in items:
            if "fast_order" in it.keywords:
                it.add_marker(sk)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (checker only)"""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx():
    from koifish_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def bf16_t(a_u16, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a_u16).view(np.int16)).to(device).view(torch.bfloat16)


def u16(t):
    """torch bf16 tensor -> numpy uint16 bit patterns"""
    import torch
    return t.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def ulp_diff_bf16(a_u16, b_u16):
    """distance in bf16 ulps between two bit-pattern arrays (sign-magnitude -> ordered ints)"""
    def key(x):
        x = x.astype(np.int32)
        return np.where(x & 0x8000, -(x & 0x7FFF), x & 0x7FFF)
    return np.abs(key(np.asarray(a_u16)) - key(np.asarray(b_u16)))


def close_bf16(a_u16, b_u16, rel_of_max=2.0 ** -10):
    """True where two bf16 arrays agree to <= 1 ulp, or -- for elements that are small against the vector's scale,
    where an ulp is far below the fp32 accumulation noise of the whole sum -- to rel_of_max * max|b|."""
    a = (np.asarray(a_u16).astype(np.uint32) << 16).view(np.float32)
    b = (np.asarray(b_u16).astype(np.uint32) << 16).view(np.float32)
    return (ulp_diff_bf16(a_u16, b_u16) <= 1) | (np.abs(a - b) <= rel_of_max * np.abs(b).max())
