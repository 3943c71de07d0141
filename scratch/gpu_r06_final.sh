#!/bin/bash
# Round-6 final evidence: the default GPU suite, smoke, the default bench run, the kernel trace of the config-3 step.  Summaries land in gpurun_out/r06f/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -q --durations=12 ) > $O/suite.txt 2>&1; tail -3 $O/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2>&1; tail -4 $O/bench.txt | cut -c1-600
cp bench_detail.json $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 $R/bench.py --leg config3 > $O/c3_traced.log 2>&1; echo "trace rc=$?"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
find $O -name "*stats*.csv" | head
