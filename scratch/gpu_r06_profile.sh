#!/bin/bash
# Round-6 evidence run: one gpurun call.  Summaries land in gpurun_out/r06p/ (copied into profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cpu-fp16-steps 0 --side-legs "" > $O/bench_traced.log 2>&1; echo "trace rc=$?"
cd $R
TRAFFIC=1 NSEQS="32 16" bash scratch/gpu_r06_pmc.sh 2>&1 | tail -60
VARIANTS="12x2" NSEQ="8" STAMPS=1 timeout 400 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids > $O/xengine_stamps.txt
NSEQ="16,32" STAMPS=1 timeout 400 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids >> $O/xengine_stamps.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*stats*.csv" | head
