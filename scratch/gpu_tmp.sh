R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
VARIANTS="8x4" NSEQ="16" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
VARIANTS="8x4" NSEQ="16" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
VARIANTS="12x6" NSEQ="8" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
