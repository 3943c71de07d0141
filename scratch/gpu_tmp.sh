R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
STAMPS=1 timeout 900 python3 scratch/xtp_time.py 16 4000 16 2>&1 | grep -v amdgpu.ids | tail -40
