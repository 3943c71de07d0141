R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q --kf-shipped-order -k "not full_depth and not eight_xcds" > $O/pytest_shipped.log 2>&1; echo "rc=$?"; tail -40 $O/pytest_shipped.log | cut -c1-220
