#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_canonical.py tests/test_gpu_tp.py tests/test_gpu_ops.py tests/test_gpu_prefill.py tests/test_gpu_gemm.py tests/test_gpu_full_size.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python scratch/prefill_time.py > $O/prefill_time.txt 2>&1; tail -3 $O/prefill_time.txt
timeout 900 python bench.py --lean --config qwen3-32b --steps 64 --warmup 16 > $O/c4.json 2> $O/c4.err; python3 -c "
import json; d=json.loads(open('$O/c4.json').read().strip().splitlines()[-1]); print('32B one gpu', d['value'], d['ms_per_step'], d['fast_order_mode'])"
