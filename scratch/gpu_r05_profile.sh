#!/bin/bash
# Round-5 evidence run: one gpurun call.  Summaries land in gpurun_out/r05p/ (copied into profiles/ by scratch/copy_profiles_r05.sh afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cpu-fp16-steps 0 --side-legs "" > $O/bench_traced.log 2>&1; echo "trace rc=$?"
for NS in 16 8; do
  NSEQ=$NS timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/x$NS/pmc_f -- python3 $R/scratch/ub_xengine.py 2037 4 3 > $O/x${NS}_f.log 2>&1; echo "pmc fetch $NS rc=$?"
  NSEQ=$NS timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/x$NS/pmc_w -- python3 $R/scratch/ub_xengine.py 2037 4 3 > $O/x${NS}_w.log 2>&1; echo "pmc write $NS rc=$?"
  NSEQ=$NS timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/x$NS/pmc_sq -- python3 $R/scratch/ub_xengine.py 2037 4 3 > $O/x${NS}_sq.log 2>&1; echo "pmc sq $NS rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 $R/bench.py --leg config3 > $O/c3.log 2>&1; echo "c3 rc=$?"
cd $R
for NS in 16 8; do
  NSEQ=$NS timeout 300 python3 scratch/ub_xengine.py 2037 4 3 > $O/ub_xengine_$NS.log 2>&1; tail -1 $O/ub_xengine_$NS.log
  ALG=$(tail -1 $O/ub_xengine_$NS.log | sed 's/.*bytes per launch \([0-9]*\) .*/\1/')
  python3 scratch/pmc_xengine_json.py $O/x$NS $NS 4 $ALG $O/r05_pmc_xengine_$NS > /dev/null
done
VARIANTS="12x6" NSEQ="8,16" STAMPS=1 timeout 400 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids > $O/xengine_stamps.txt
STAMPS=1 timeout 600 python3 scratch/xtp_time.py 16 4000 16 2>&1 | grep -v amdgpu.ids > $O/xtp_stamps.txt
for c in qwen3-1.7b qwen3-4b qwen3-8b; do CONFIG=$c VARIANTS="12x6" NSEQ="8" timeout 600 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids | sed "s/^/$c  /" >> $O/xengine_shapes.txt; done
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_xtp -- python3 $R/bench.py --config qwen3-32b --tp-virtual 8 --tp-xcd 1 --tp-layers 16 --steps 32 --warmup 8 > $O/xtp_traced.log 2>&1; echo "xtp trace rc=$?")
timeout 2400 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; tail -c 300 $O/bench_driver_flags.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O; find $O -name "*stats*.csv" | head -20
