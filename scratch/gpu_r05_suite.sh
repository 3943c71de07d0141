#!/bin/bash
# the per-change GPU call of round 5: the whole -m gpu suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -18 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 2400 python3 bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err ) 2>&1 | grep real; tail -c 600 $O/bench_line.json
