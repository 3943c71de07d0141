#!/bin/bash
# the per-change GPU call of round 4: the whole -m gpu suite, smoke(), then whatever A/B the change needs (edit below)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python scratch/eng_ab.py "CANON=1 TUNE=2" "CANON=0 TUNE=2" 2>&1 | tail -2
