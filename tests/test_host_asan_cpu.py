"""SURVEY section 5 (race detection / sanitizers): the host library's checkpoint readers (own JSON, msgpack and mmap code over untrusted files) under
AddressSanitizer + UBSan.  CPU only: builds koifish_amd/libkf_host_asan.so with g++ -fsanitize=address,undefined and re-runs the checkpoint tests
(tests/test_safetensors_cpu.py, tests/test_kun_cpu.py: well-formed and malformed files) in a child interpreter that has libasan preloaded."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_checkpoint_readers_under_asan_ubsan():
    sys.path.insert(0, ROOT)
    from koifish_amd import build as B
    so = B.build_host_asan()
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "libasan.so not found next to gcc"
    env = dict(os.environ, KF_HOST_LIB=so, LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_safetensors_cpu.py"),
                        os.path.join(ROOT, "tests", "test_kun_cpu.py")], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
