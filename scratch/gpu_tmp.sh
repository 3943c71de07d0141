R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_prefill.py tests/test_gpu_train_step.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -4
