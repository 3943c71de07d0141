"""The drop-in boundary without a GPU: libkf_hip.so loads, exports every function include/kf_abi.h declares (and nothing is
declared that is not exported), fails cleanly when no device is present, and the host library above it loads too."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "kf_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kf_[a-z0-9_]+)\s*\(", src)))


def exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def test_header_and_library_agree():
    from koifish_amd import lib as L
    hip, host = L.load()
    decl = declared_functions()
    exp = exported(L.LIB_HIP)
    missing = [f for f in decl if f not in exp]
    assert not missing, "declared in kf_abi.h but not exported: %s" % missing
    extra = sorted(f for f in exp if f.startswith("kf_") and f not in decl)
    assert not extra, "exported but not declared in kf_abi.h: %s" % extra
    assert sorted(L.ABI_SYMBOLS) == decl, "koifish_amd/lib.py ABI_SYMBOLS out of date"
    for f in decl:
        assert hasattr(hip, f)


def test_no_torch_types_in_the_abi():
    src = open(os.path.join(ROOT, "include", "kf_abi.h")).read()
    assert "torch" not in src and "at::" not in src and "#include <hip" not in src


def test_init_without_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from koifish_amd import lib as L
    hip, host = L.load()
    h = C.c_void_p()
    rc = hip.kf_init(0, None, C.byref(h))
    assert rc == -1400 and not h.value          # KOIFISH_CUDA_CHECK
    assert b"no HIP device" in hip.kf_last_error()
    rcv = C.c_int(0)
    m = host.kfh_create(0, None, 256, 2, 4, 2, 64, 512, 512, 96, 1e-6, 1e-6, 1e6, C.byref(rcv))
    assert not m and rcv.value == -1400
    from koifish_amd.runtime import Context
    with pytest.raises(L.KFError):
        Context(0)


def test_product_does_not_import_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "koifish_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                # (comments may cite oracle/kf_oracle.c; importing, linking or loading it is what is forbidden)
                assert "import oracle" not in txt and "from oracle" not in txt and "libkf_oracle" not in txt and '#include "../../oracle' not in txt, os.path.join(dp, f)
    b = open(os.path.join(ROOT, "bench.py")).read()
    head, _, tail = b.partition("def cpu_baseline")
    assert "from oracle import" in tail and "from oracle import" not in head


def test_rope_table_host_matches_libm():
    import math
    import numpy as np
    from koifish_amd import lib as L
    hip, _ = L.load()
    t = np.zeros((5, 32, 2), dtype=np.float32)
    assert hip.kf_rope_table_host(t.ctypes.data_as(C.c_void_p), 5, 64, 1e6) == 0
    assert t[0, :, 0].tolist() == [1.0] * 32 and not t[0, :, 1].any()
    from oracle import oracle as O
    for p in range(5):
        c, s = O.rope_table(p, 64, 1e6)
        assert np.array_equal(t[p, :, 0], c) and np.array_equal(t[p, :, 1], s)
    assert hip.kf_rope_table_host(None, 5, 64, 1e6) == -20


def test_kf_weight_struct_matches_the_ctypes_mirror(tmp_path):
    """struct kf_weight as the C compiler lays it out (offsets of every field, total size) == koifish_amd.lib.Weight: a field added to the header
    without the Python mirror (or the other way round) fails here, not as silent garbage in a kernel argument"""
    from koifish_amd import lib as L
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "kf_abi.h")).read(), flags=re.S)
    body = re.search(r"typedef struct kf_weight \{(.*?)\} kf_weight;", hdr, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields += [n.strip().lstrip("*") for n in decl.split(" ", 1)[1].replace("void*", "").replace("kf_bf16*", "").split(",")] if "," in decl \
                else [decl.split()[-1].lstrip("*")]
    assert fields == [n for n, _ in L.Weight._fields_], (fields, [n for n, _ in L.Weight._fields_])
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "kf_abi.h"\nint main(void) {\n' +
                   "".join('  printf("%%zu\\n", offsetof(kf_weight, %s));\n' % f for f in fields) + '  printf("%zu\\n", sizeof(kf_weight));\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).decode().split()]
    assert nums[:-1] == [getattr(L.Weight, f).offset for f in fields]
    assert nums[-1] == C.sizeof(L.Weight)
