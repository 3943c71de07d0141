#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
( time timeout 1800 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2>&1 | grep real; python3 - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r04i/bench_driver_flags.json').read().strip().splitlines()[-1])
print("value", d.get("value"), d.get("ms_per_step"), d["fast_order_mode"]["tokens_per_s"], d["prefill"]["ms"], d["prefill"]["long_prompt"]["ms"])
for k in ("config3_train_step", "config5_sparse_1bit", "config4_one_gpu"):
    x = d.get(k); print(k, x.get("tokens_per_s"), x.get("ms_per_step") or x.get("ms"), x.get("fast_order_tokens_per_s"))
print(json.dumps(d["config4_one_gpu"].get("tp8_virtual_ranks"))[:1200])
PY
