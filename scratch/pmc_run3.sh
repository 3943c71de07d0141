#!/bin/bash
# pmc_run.sh + a third pass with the matrix-pipe counters: pmc_run3.sh <tag> <script> [args...] -> gpurun_out/pmc_<tag>/{a,b,c}
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
bash $R/scratch/pmc_run.sh $tag "$@"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_$tag/c -- python3 $R/$@ > $R/gpurun_out/pmc_$tag.c.log 2>&1
find $R/gpurun_out/pmc_$tag -name "*kernel_trace.csv" -delete; find $R/gpurun_out/pmc_$tag -name "*agent_info.csv" -delete
python3 - $R/gpurun_out/pmc_$tag <<P
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(("void kf::", "kf::")): acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    print("   ", {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
P
