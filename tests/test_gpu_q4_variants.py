"""The 4-bit mat-vec has a default form (the register table inside the main kernel wherever a group is one lane quad) and the arithmetic form it
replaces (KF_Q4_PERM=0), bit for bit; the knob is read once per process.  Both meet the same parity bar; each runs in a child process with its knob set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("knobs", [{"KF_Q4_PERM": "0"}, {"KF_Q4_PERM": "1"}])
def test_q4_matvec_form(knobs):
    env = dict(os.environ)
    env.update(knobs)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_q4_variant_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "%s\n%s\n%s" % (knobs, r.stdout[-2000:], r.stderr[-2000:])


_DIGEST = r"""
import sys, hashlib, torch
sys.path.insert(0, %r)
from koifish_amd.runtime import Context
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(5)
h = hashlib.sha256()
for (m, k) in [(4096, 1024), (1024, 3072), (1000, 2048), (40, 3200 if sys.argv[1] == 'Q4' else 3072), (8192, 5120)]:
    W = (torch.randn(m, k, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    x = torch.randn(k, device=dev, generator=g).to(torch.bfloat16)
    w = ctx.quantize(W, getattr(L, sys.argv[1]))
    y = ctx.linear(w, x)
    a = ctx.norm_gateup_swiglu(x, torch.ones(k, device=dev, dtype=torch.bfloat16), w, w)
    lg, am = ctx.lm_head(w, x)
    ctx.sync()
    for t in (y, a, lg):
        h.update(t.view(torch.int16).cpu().numpy().tobytes())
    h.update(str(am).encode())
print("DIGEST", h.hexdigest())
""" % ROOT


@pytest.mark.parametrize("knob,type_name", [("KF_Q4_PERM", "Q4"), ("KF_Q2_TAB", "T_SIGN"), ("KF_Q1_TAB", "BOOL1")])
def test_table_form_is_bit_identical_to_the_arithmetic_form(knob, type_name):
    """the defaults -- 4-bit: register-table lookup (BlockDot<FMT_Q4P>); 2-bit / 1-bit: v_perm selectors from an LDS table indexed by a weight byte
    (BlockDot<FMT_Q2T> / <FMT_Q1T>) -- form the same weights, pair them and sum them exactly as the per-weight arithmetic forms do: every output bit of
    the plain, paired-SwiGLU and arg-max launches is the same"""
    digests = []
    for v in ("0", "1"):
        env = dict(os.environ, **{knob: v})
        r = subprocess.run([sys.executable, "-c", _DIGEST, type_name], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        digests.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert digests[0] == digests[1]


def test_vendor_gemm_is_opt_in_and_keeps_the_in_place_residual():
    """KF_GEMM_LIB=1 (opt-in; the default path is the hand-written tile kernels): large token batches may go through dequantise + rocBLAS, but a residual that
    aliases y (the in-place form of SelfAttention / FFN::cuFlow) must still give residual + x.W^T -- the library path re-reads the residual after y is
    written, so such calls stay on the tile kernels."""
    code = r"""
import sys, ctypes as C, torch
sys.path.insert(0, %r)
from koifish_amd.runtime import Context
from koifish_amd import lib as L
ctx = Context(0); dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(9)
n, m, k = 2048, 1024, 1024
W = (torch.randn(m, k, device=dev, generator=g) * 0.05).to(torch.bfloat16)
x = torch.randn(n, k, device=dev, generator=g).to(torch.bfloat16)
res = torch.randn(n, m, device=dev, generator=g).to(torch.bfloat16)
w = ctx.quantize(W, L.Q4)
ctx.linear_scratch(w, n)
ws = torch.empty(m * k * 2, dtype=torch.uint8, device=dev)          # the library path dequantises into caller-owned scratch
L.check(ctx.hip.kf_set_scratch(ctx.h, C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel())), "kf_set_scratch")
d = w.desc()
y_sep = torch.zeros(n, m, dtype=torch.bfloat16, device=dev)
L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y_sep.data_ptr(), None, n, 1.0, 0.0, 1, res.data_ptr()), "kf_linear")
y_inp = res.clone()
L.check(ctx.hip.kf_linear(ctx.h, C.byref(d), x.data_ptr(), y_inp.data_ptr(), None, n, 1.0, 0.0, 1, y_inp.data_ptr()), "kf_linear in place")
ctx.sync()
ref = res.float() + (x.float() @ ctx.dequant(w).float().T).to(torch.bfloat16).float()
for name, y in (("separate", y_sep), ("in place", y_inp)):
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    print(name, err)
    assert err < 2.0 ** -6, (name, err)
print("OK")
""" % ROOT
    for lib in ("1", "0"):
        env = dict(os.environ, KF_GEMM_LIB=lib)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, "KF_GEMM_LIB=%s\n%s\n%s" % (lib, r.stdout[-2000:], r.stderr[-2000:])
