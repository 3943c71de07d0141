#!/bin/bash
# Round-3 evidence run, second half (after the dequantise-ahead engine and the small-batch GEMM changes): one gpurun call.  Summaries land in gpurun_out/r03b/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cpu-fp16-steps 0 --side-legs "" > $O/bench_traced.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_f.log 2>&1; echo "pmc fetch rc=$?"; tail -1 $O/pmc_f.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_w.log 2>&1; echo "pmc write rc=$?"; tail -1 $O/pmc_w.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pf128 -- python3 $R/scratch/prefill_prof.py 128 > $O/pf128.log 2>&1; echo "pf128 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pf2047 -- python3 $R/scratch/prefill_prof.py 2047 > $O/pf2047.log 2>&1; echo "pf2047 rc=$?"
cd $R
timeout 300 python3 scratch/ub_engine.py 2037 > $O/ub_engine_plain.log 2>&1; tail -1 $O/ub_engine_plain.log
ALG=$(tail -1 $O/ub_engine_plain.log | sed 's/.*per launch \([0-9]*\) .*/\1/')
python3 scratch/pmc_engine_json.py $O 2037 $ALG $O/r03b_pmc_engine.json > /dev/null
STAMP_WG=77 STAMP_LAYERS=1 python3 scratch/eng_stamps.py 2040 > $O/stamps_default.txt 2>&1
CANON=1 STAMP_WG=77 STAMP_LAYERS=1 python3 scratch/eng_stamps.py 2040 > $O/stamps_canonical.txt 2>&1
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.log 2>&1; tail -c 300 $O/bench_driver_flags.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; find $O -name "*stats*.csv" | head
