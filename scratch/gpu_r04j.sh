#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --layers ternary --steps 512 --warmup 64 --lean > $O/ternary.json 2> $O/ternary.err; python3 -c "
import json; d=json.loads(open('$O/ternary.json').read().strip().splitlines()[-1]); print('ternary dense', d['value'], d['ms_per_step'], d['fast_order_mode'].get('tokens_per_s'), d['config']['decode_path'])"
