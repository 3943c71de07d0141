#!/bin/bash
# usage: pmc_run.sh <tag> <script> [args...]   (env passes through) -> gpurun_out/pmc_<tag>/{a,b}
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_$tag/a -- python3 $R/$@ > $R/gpurun_out/pmc_$tag.a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_$tag/b -- python3 $R/$@ > $R/gpurun_out/pmc_$tag.b.log 2>&1
find $R/gpurun_out/pmc_$tag -name "*kernel_trace.csv" -delete; find $R/gpurun_out/pmc_$tag -name "*agent_info.csv" -delete
