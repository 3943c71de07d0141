R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
VARIANTS="8x4,86x6,81x8,82x4,8x4" NSEQ="16" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
VARIANTS="8x4" NSEQ="16" DEAL="14,13,15" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
