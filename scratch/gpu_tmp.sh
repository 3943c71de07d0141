R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python3 scratch/xtp_time.py 16 4000 16 2>&1 | grep -v amdgpu.ids | head -3
CONFIG=qwen3-4b VARIANTS="12x6" NSEQ="8" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
CONFIG=qwen3-8b VARIANTS="12x6" NSEQ="8" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_gpu_xengine.py tests/test_gpu_tp.py -x -q -k "gqa4 or eight_xcds or refus or eight_query" 2>&1 | tail -4
