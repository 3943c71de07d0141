#!/bin/bash
# Round-3 evidence run on the GPU box (one gpurun call): kernel-trace stats of the driver's bench command, the PMC passes of the engine kernel,
# kernel-trace stats of the side legs (config 3 / 5 / 4).  Summaries land in gpurun_out/r03/; the ones to keep are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cpu-fp16-steps 0 --side-legs "" > $O/bench_traced.log 2>&1; echo "trace rc=$?"; tail -c 400 $O/bench_traced.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_f.log 2>&1; echo "pmc fetch rc=$?"; tail -1 $O/pmc_f.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_w.log 2>&1; echo "pmc write rc=$?"; tail -1 $O/pmc_w.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
cd $R
timeout 300 python3 scratch/ub_engine.py 2037 > $O/ub_engine_plain.log 2>&1; tail -1 $O/ub_engine_plain.log
ALG=$(tail -1 $O/ub_engine_plain.log | sed 's/.*per launch \([0-9]*\) .*/\1/')
python3 scratch/pmc_engine_json.py $O 2037 $ALG $O/r03_pmc_engine.json > /dev/null
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 $R/bench.py --leg config3 > $O/config3.log 2>&1; echo "config3 rc=$?"; tail -c 300 $O/config3.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -- python3 $R/bench.py --layers 1bit --sparse 0.2 --steps 512 --warmup 64 --lean > $O/config5.log 2>&1; echo "config5 rc=$?"; tail -c 300 $O/config5.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --config qwen3-32b --steps 64 --warmup 16 --lean > $O/config4.log 2>&1; echo "config4 rc=$?"; tail -c 300 $O/config4.log
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; find $O -name "*stats*.csv" | head
