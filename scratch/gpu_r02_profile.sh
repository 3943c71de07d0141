#!/bin/bash
# Round-2 evidence run on the GPU box: the .kun tests, a kernel-trace of the default bench, the two PMC passes of the engine, a plain bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kun.py -x -q > $O/kun_tests.log 2>&1; echo "kun tests rc=$?"; tail -5 $O/kun_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --cpu-seconds 0 --cpu-fp16-steps 0 > $O/bench_traced.log 2>&1; echo "trace rc=$?"; tail -2 $O/bench_traced.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/scratch/ub_engine.py > $O/pmc_f.log 2>&1; echo "pmc fetch rc=$?"; tail -2 $O/pmc_f.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/scratch/ub_engine.py > $O/pmc_w.log 2>&1; echo "pmc write rc=$?"; tail -2 $O/pmc_w.log
cd $R
timeout 600 python3 scratch/ub_engine.py > $O/ub_engine_plain.log 2>&1; tail -1 $O/ub_engine_plain.log
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cat $O/bench_default.json
find $O -name "*.csv" | head -20
# keep the merged output small: drop the per-dispatch traces, keep stats and counter files
find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
