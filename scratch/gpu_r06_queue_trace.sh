#!/bin/bash
# kernel trace of the request queue (XcdReplicas::Chat, 96 requests of 128 + 128 tokens through 32 slots, 16 prompts prefilled together): summary into gpurun_out/r06q/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export NSEQ=32 PB=16
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/scratch/xr_chat.py 96 128 128 > $O/traced.log 2>&1; echo "trace rc=$?"
tail -3 $O/traced.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
find $O -name "*kernel_stats.csv" | head -2
