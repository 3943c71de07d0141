R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for V in "" 12x8 12x4 12x62 8x8; do VARIANT=$V timeout 600 python3 scratch/xtp_time.py 16 4000 16 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/variant [$V]: /"; done
