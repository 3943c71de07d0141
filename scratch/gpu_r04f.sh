#!/bin/bash
# 1-bit canonical sub-chains (oracle, mat-vec kernels) + the engine's dword-per-lane 1-bit form; new engine API tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python bench.py --layers 1bit --sparse 0.2 --steps 512 --warmup 64 --lean > $O/config5.json 2> $O/config5.err; python3 -c "
import json; d=json.loads(open('$O/config5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'], d['fast_order_mode'], d['config']['decode_path'], d.get('engine_handoffs'))"
timeout 600 python bench.py --layers 1bit --steps 512 --warmup 64 --lean > $O/onebit_dense.json 2> $O/onebit_dense.err; python3 -c "
import json; d=json.loads(open('$O/onebit_dense.json').read().strip().splitlines()[-1]); print('1bit dense', d['value'], d['ms_per_step'], d['fast_order_mode'])"
timeout 600 python bench.py --config qwen3-32b --layers 1bit --steps 32 --warmup 8 --lean > $O/onebit_32b.json 2> $O/onebit_32b.err; python3 -c "
import json; d=json.loads(open('$O/onebit_32b.json').read().strip().splitlines()[-1]); print('1bit 32B one gpu', d['value'], d['ms_per_step'], d['fast_order_mode'])"
