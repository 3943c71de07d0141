import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--kf-slow", action="store_true", default=False, help="also run the tests marked `slow` (the long forms: full-depth TP, 4096-position runs, every sequence of the widest launches)")
    parser.addoption("--kf-shipped-order", action="store_true", default=False, help="(the default since round 6; kept as a no-op) the library's own default summation order (canonical) for the whole suite")
    parser.addoption("--kf-fast-order", action="store_true", default=False,
                     help="start every context / model of the suite in the v_dot2c order (kf_set_canonical(ctx, 0)) as rounds 4 - 5 did; the tests marked fast_order switch to it themselves either way")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long form of a GPU test whose shortened variant runs by default (run with --kf-slow; the default `-m gpu` run stays inside the driver's step limit)")
    config.addinivalue_line("markers", "fast_order: written against the v_dot2c order on BOTH sides: runs in that order whatever the suite's default (the fast_order_for_marked fixture)")
    # Round 6: the suite runs on what SHIPS -- the library's default summation order, the canonical one -- so that the driver's own `pytest -m gpu` is evidence for it.
    # The tolerance tests hold in it as they held in the v_dot2c order (their bound is 2^-6 of scale against the oracle's dot16 order); the two tests that compare two v_dot2c
    # computations are marked fast_order and switch themselves.  --kf-fast-order starts every context in the v_dot2c order instead (rounds 4 - 5).
    if config.getoption("--kf-fast-order"):
        from koifish_amd import runtime
        runtime.DEFAULT_CANONICAL = False


@pytest.fixture(autouse=True)
def fast_order_for_marked(request):
    """a test marked fast_order runs with contexts and models starting in the v_dot2c order (and the session context switched to it), then everything is put back"""
    if "fast_order" not in request.keywords or request.config.getoption("--kf-fast-order"):
        yield
        return
    from koifish_amd import runtime
    old = runtime.DEFAULT_CANONICAL
    runtime.DEFAULT_CANONICAL = False
    c = request.getfixturevalue("ctx") if "ctx" in request.fixturenames else None
    mdl = request.getfixturevalue("model") if "model" in request.fixturenames else None   # (a module-scoped (cfg, model) pair made before this fixture ran)
    if c is not None:
        c.set_canonical(False)
    if mdl is not None:
        mdl[1].set_canonical(False)
    yield
    runtime.DEFAULT_CANONICAL = old
    if c is not None:
        c.set_canonical(True)
    if mdl is not None:
        mdl[1].set_canonical(True)


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
