"""The 4-bit mat-vec has one default form per launch size and three opt-in forms (environment knobs read once per process: KF_Q4_PERM = register
table inside the main kernel -- the default for launches of >= 0.5 M blocks --, KF_Q4_LUT = 1 / 2 / 3 the lane-owns-a-group kernels of
kf_gemv_lut.hip).  Every form must meet the same parity bar; each runs in a child process with its knob set."""
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
