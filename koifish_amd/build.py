"""Builds the two in-tree shared libraries of koifish_amd:

  libkf_hip.so   hipcc --offload-arch=gfx950: the kernels + the C ABI of include/kf_abi.h
  libkf_host.so  g++: the C++ host mirror of the reference's neuron interface, above the ABI

Both are written next to this file so they travel with the source tree.  No JIT cache.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
HIP_SOURCES = ["kf_gemv.hip", "kf_gemv_canon.hip", "kf_gemm.hip", "kf_gemm2.hip", "kf_gemm3.hip", "kf_attn.hip", "kf_engine.hip", "kf_xengine.hip", "kf_xengine_q1.hip", "kf_attn_prefill.hip", "kf_ops.hip", "kf_lut.hip", "kf_loss.hip", "kf_norm_bwd.hip", "kf_linear_bwd.hip", "kf_embed_bwd.hip", "kf_attn_bwd_mfma.hip", "kf_awq.hip", "kf_tp.hip", "kf_abi.hip"]
HIP_DEPS = ["kf_device.h", "kf_kernels.h", "kf_gemm_common.h", "kf_gemv_blocks.h", "kf_attn_common.h", "kf_engine_common.h", "kf_xengine_kernel.h"]
LIB_HIP = os.path.join(HERE, "libkf_hip.so")
LIB_HOST = os.path.join(HERE, "libkf_host.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the kernels spell every fma they want; an implicit contraction would change the rounding
# points that the oracle pins (RoPE, kf_expf, RMSNorm).
# -fno-slp-vectorize: keeps the dequant chain in plain v_fma/v_sub (the packed-f32 forms the SLP pass picks cost more issue
# slots, need s_nop hazard padding and 40 % more registers on this kernel)
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
# MFMA results that the VALU works on next (attention scores, the epilogues of the token-batch GEMMs): with the default AGPR form of the MFMA destination the
# compiler moves every such value AGPR -> VGPR (v_accvgpr_read) and zero-fills accumulators through v_accvgpr_write -- 160 of the ~330 vector instructions of a
# key tile in the attention backward.  The VGPR form has none of them (and the files below stay inside 256 VGPRs where it matters).
HIP_FILE_FLAGS = {f: ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] for f in os.environ.get("KF_VGPR_FORM", "kf_attn_prefill.hip,kf_attn_bwd_mfma.hip,kf_gemm.hip").split(",") if f}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, d) for d in HIP_DEPS] + [os.path.join(HERE, "..", "include", "kf_abi.h")]
    objs, cmds = [], []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        extra = [os.path.join(CSRC, "kf_gemv.hip")] if os.path.basename(s) == "kf_gemv_canon.hip" else []   # it #includes kf_gemv.hip
        if force or _stale(o, [s] + extra + deps[len(srcs):]):
            cmds.append([HIPCC] + HIP_FLAGS + HIP_FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o])
    if cmds:   # independent translation units: a few at a time (the container has 8 CPUs; one hipcc peaks near 2 GB)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=int(os.environ.get("KF_BUILD_JOBS", "6"))) as ex:
            list(ex.map(run, cmds))
    if force or _stale(LIB_HIP, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_HIP] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_HIP


def build_host(force=False, verbose=False):
    src = os.path.join(HOST, "kf_host.cpp")
    src2 = os.path.join(HOST, "kf_safetensors.cpp")
    src3 = os.path.join(HOST, "kf_train.cpp")
    deps = [src, src2, src3, os.path.join(HOST, "kf_safetensors.hpp"), os.path.join(HOST, "kf_host.hpp"), os.path.join(HERE, "..", "include", "kf_abi.h"), LIB_HIP]
    if force or _stale(LIB_HOST, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", LIB_HOST, src, src2, src3, "-L" + HERE, "-lkf_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_HOST


def build_host_asan(verbose=False):
    """CPU-only sanitizer build of the host library (AddressSanitizer + UBSan): the JSON / msgpack / mmap parsers of kf_safetensors.cpp read untrusted
    files.  Written to libkf_host_asan.so; tests/test_host_asan_cpu.py runs the checkpoint tests under it (LD_PRELOAD of libasan, KF_HOST_LIB).
    Never for the GPU box: sanitizers run on the CPU build only."""
    out = os.path.join(HERE, "libkf_host_asan.so")
    src = os.path.join(HOST, "kf_host.cpp")
    src2 = os.path.join(HOST, "kf_safetensors.cpp")
    src3 = os.path.join(HOST, "kf_train.cpp")
    deps = [src, src2, src3, os.path.join(HOST, "kf_safetensors.hpp"), os.path.join(HOST, "kf_host.hpp"), os.path.join(HERE, "..", "include", "kf_abi.h"), LIB_HIP]
    if _stale(out, deps):
        cmd = ["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-o", out, src, src2, src3, "-L" + HERE, "-lkf_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    if "asan" in sys.argv:   # python -m koifish_amd.build asan
        print(build_host_asan(verbose=True))
    else:
        print(build_all(force="--force" in sys.argv, verbose=True))
