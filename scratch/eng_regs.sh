#!/bin/bash
# register / spill / scratch figures of every kernel of one HIP source (default: the persistent engine): scratch/eng_regs.sh [file.hip] [extra hipcc flags]
F=${1:-kf_engine.hip}; shift
O=$(mktemp -d /tmp/engregs.XXXXXX)
cd "$(dirname "$0")/../koifish_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function --cuda-device-only -c "$F" -o $O/b.o "$@" || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$O/b.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$O/d.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $O/d.o | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.agpr_count:/{a=$2} /\.vgpr_spill_count:/{sp=$2} /\.private_segment_fixed_size:/{s=$2} /\.wavefront_size:/{print n, "vgpr", v, "agpr", a, "spill", sp, "scratch", s}' | c++filt | cut -c1-220
rm -rf $O
