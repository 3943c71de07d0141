"""The 4-bit mat-vec has a default form (the register table inside the main kernel wherever a group is one lane quad; KF_Q4_PERM=0 selects the
arithmetic form it replaces, bit for bit) and three opt-in forms (KF_Q4_LUT = 1 / 2 / 3: the lane-owns-a-group kernels of kf_gemv_lut.hip); the
knobs are read once per process.  Every form must meet the same parity bar; each runs in a child process with its knob set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("knobs", [{"KF_Q4_PERM": "0"}, {"KF_Q4_PERM": "1"}, {"KF_Q4_LUT": "1"}, {"KF_Q4_LUT": "2"}, {"KF_Q4_LUT": "3"}])
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


def test_attention_backward_first_version():
    """KF_ATTN_BWD=valu selects the first (VALU on LDS tiles, head_dim 64) attention backward of kf_attn_bwd.hip instead of the MFMA form: same tests"""
    env = dict(os.environ)
    env["KF_ATTN_BWD"] = "valu"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_gpt2_ops.py"), "-q", "-m", "gpu", "-x", "-k", "attn_backward_vs_oracle and 64"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


def test_large_batch_gemm_without_the_vendor_library():
    """KF_GEMM_LIB=0: token batches of >= 2048 rows stay on the hand-written dequant-GEMM kernels (forward) and on dequantise + transposes + those
    kernels (backward): the same tests as with the library"""
    env = dict(os.environ)
    env["KF_GEMM_LIB"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_linear_backward.py"), "-q", "-m", "gpu", "-x", "-k", "2048 or large_batch"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout
