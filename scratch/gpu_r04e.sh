#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/dot2_single scratch/dbg/dot2_single.hip && /tmp/dot2_single > $O/dot2_single.txt 2>&1; cat $O/dot2_single.txt
timeout 900 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_full_size.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python scratch/prefill_time.py > $O/prefill_time.txt 2>&1; tail -6 $O/prefill_time.txt
