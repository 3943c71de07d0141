R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_xengine.py tests/test_gpu_tp.py -x -q -k "tiny-96 or refusals or prefill_then or long_context or small-320-150-16" 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
