"""A/B builds of the device library: scratch/build_variant.py NAME file.hip[,file2.hip] -DENG_X=1 ... -> scratch/variants/NAME/{libkf_hip.so, libkf_host.so}
(the named sources recompiled with the extra flags, every other object taken from the in-tree build).  Run a script against it with KF_LIB_DIR=scratch/variants/NAME."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from koifish_amd import build as B  # noqa: E402


def main():
    name, files, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    B.build_all()
    out = os.path.join(HERE, "variants", name)
    os.makedirs(out, exist_ok=True)
    objs = []
    for s in B.HIP_SOURCES:
        o = os.path.join(B.CSRC, s[:-4] + ".o")
        if s in files:
            o = os.path.join(out, s[:-4] + ".o")
            subprocess.check_call([B.HIPCC] + B.HIP_FLAGS + B.HIP_FILE_FLAGS.get(s, []) + flags + ["-c", os.path.join(B.CSRC, s), "-o", o])
        objs.append(o)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libkf_hip.so")] + objs)
    for s in files:
        os.remove(os.path.join(out, s[:-4] + ".o"))
    shutil.copy(B.LIB_HOST, os.path.join(out, "libkf_host.so"))   # rpath $ORIGIN: it binds to the variant's libkf_hip.so
    print(out)


if __name__ == "__main__":
    main()
