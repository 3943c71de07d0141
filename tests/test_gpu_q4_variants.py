"""The low-bit mat-vecs have a default form (4-bit: the register table inside the main kernel wherever a group is one lane quad; 2-bit / 1-bit: v_perm
selectors from an LDS table) and the per-weight arithmetic form each replaces, bit for bit.  The forms are switched inside one process through the
development hook kfdbg_set_knob (not part of the ABI; the product reads no environment variables)."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from koifish_amd import lib as L

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _knob(ctx, name, value):
    ctx.hip.kfdbg_set_knob.argtypes = [C.c_char_p, C.c_long]
    assert ctx.hip.kfdbg_set_knob(name.encode(), int(value)) == 0, name


@pytest.mark.parametrize("perm", [0, 1])
def test_q4_matvec_form(perm):
    """both forms of the 4-bit mat-vec meet the oracle's parity bar (plain / fused-norm / SwiGLU-pair entries); a child process so that a failure cannot leave the
    knob set for the rest of the run"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_q4_variant_child.py"), str(perm)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "q4_perm=%d\n%s\n%s" % (perm, r.stdout[-2000:], r.stderr[-2000:])


def _digest(ctx, type_name):
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    h = hashlib.sha256()
    for (m, k) in [(4096, 1024), (1024, 3072), (1000, 2048), (40, 3200 if type_name == "Q4" else 3072), (8192, 5120)]:
        W = (torch.randn(m, k, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        x = torch.randn(k, device=dev, generator=g).to(torch.bfloat16)
        w = ctx.quantize(W, getattr(L, type_name))
        y = ctx.linear(w, x)
        a = ctx.norm_gateup_swiglu(x, torch.ones(k, device=dev, dtype=torch.bfloat16), w, w)
        lg, am = ctx.lm_head(w, x)
        ctx.sync()
        for t in (y, a, lg):
            h.update(t.view(torch.int16).cpu().numpy().tobytes())
        h.update(str(am).encode())
    return h.hexdigest()


@pytest.mark.parametrize("knob,type_name", [("q4_perm", "Q4"), ("q2_tab", "T_SIGN"), ("q1_tab", "BOOL1")])
def test_table_form_is_bit_identical_to_the_arithmetic_form(ctx, knob, type_name):
    """the defaults -- 4-bit: register-table lookup (BlockDot<FMT_Q4P>); 2-bit / 1-bit: v_perm selectors from an LDS table indexed by a weight byte
    (BlockDot<FMT_Q2T> / <FMT_Q1T>) -- form the same weights, pair them and sum them exactly as the per-weight arithmetic forms do: every output bit of
    the plain, paired-SwiGLU and arg-max launches is the same"""
    try:
        _knob(ctx, knob, 0)
        d0 = _digest(ctx, type_name)
        _knob(ctx, knob, 1)
        d1 = _digest(ctx, type_name)
    finally:
        _knob(ctx, knob, 1)
    assert d0 == d1


def test_large_batch_keeps_the_in_place_residual(ctx):
    """a training-size batch (n >= 2048 rows) through kf_linear with a residual that aliases y (the in-place form of SelfAttention / FFN::cuFlow) gives
    residual + x.W^T, as the separate-output form does: the tile kernels read the residual per element before they store.  No vendor GEMM is
    involved any more: libkf_hip.so does not load rocBLAS."""
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    n, m, k = 2048, 1024, 1024
    W = (torch.randn(m, k, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    x = torch.randn(n, k, device=dev, generator=g).to(torch.bfloat16)
    res = torch.randn(n, m, device=dev, generator=g).to(torch.bfloat16)
    w = ctx.quantize(W, L.Q4)
    ctx.linear_scratch(w, n)
    d = w.desc()
    y_sep = torch.zeros(n, m, dtype=torch.bfloat16, device=dev)
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y_sep.data_ptr(), None, n, 1.0, 0.0, 1, res.data_ptr()), "kf_linear")
    y_inp = res.clone()
    L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y_inp.data_ptr(), None, n, 1.0, 0.0, 1, y_inp.data_ptr()), "kf_linear in place")
    ctx.sync()
    ref = res.float() + (x.float() @ ctx.dequant(w).float().T).to(torch.bfloat16).float()
    for name, y in (("separate", y_sep), ("in place", y_inp)):
        err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2.0 ** -6, (name, err)


def test_the_product_library_has_no_vendor_gemm_and_reads_no_environment():
    """libkf_hip.so neither loads rocBLAS nor looks at environment variables (the yardstick lives in scratch/, knobs behind kfdbg_set_knob)"""
    so = os.path.join(ROOT, "koifish_amd", "libkf_hip.so")
    blob = open(so, "rb").read()
    assert b"librocblas" not in blob and b"rocblas_gemm_ex" not in blob
    out = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    assert " getenv" not in out and "dlopen" not in out, out[-2000:]
