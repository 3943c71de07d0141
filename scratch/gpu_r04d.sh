#!/bin/bash
# round 4, fourth GPU call: the driver's bench command end to end (all side legs), sparse tests again (fp32 activations in the canonical 1-bit engine form), stamps in both orders
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_engine.py tests/test_gpu_canonical.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"; tail -c 400 $O/bench_driver_flags.err; python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r04d/bench_driver_flags.json').read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "step_roofline", "fast_order_mode", "engine_handoffs"):
    print(k, d.get(k))
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "us_per_launch", "traffic")})
cb = d.get("cpu_baseline") or {}
print("cpu", cb.get("value"), cb.get("cores"), cb.get("parity_timed_order"), cb.get("parity_fast_order"))
for k in ("config3_train_step", "config5_sparse_1bit", "config4_one_gpu"):
    print(k, json.dumps(d.get(k))[:1500])
print("prefill", json.dumps(d.get("prefill"))[:600])
PY
CANON=1 STAMP_WG=77 STAMP_LAYERS=1 timeout 300 python scratch/eng_stamps.py 2040 > $O/stamps_canonical.txt 2>&1
CANON=0 STAMP_WG=77 STAMP_LAYERS=1 timeout 300 python scratch/eng_stamps.py 2040 > $O/stamps_fast.txt 2>&1
tail -3 $O/stamps_canonical.txt; tail -3 $O/stamps_fast.txt
