"""koifish_amd -- MI355X (gfx950) implementation of Koifish's quantized transformer forward path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/kf_abi.h), host/ (C++ mirror of the
reference's neuron interface above the ABI) and thin ctypes plumbing.  PyTorch is used for device memory,
streams and torch.distributed only.  There is no CPU fallback: loading fails loudly when the HIP library is
missing, and nothing in this package imports oracle/.
"""
from .lib import load, KFError  # noqa: F401
