#!/bin/bash
# usage (on the GPU box): bash scratch/prof_stats.sh <tag> <script.py> [args]  -> top kernels by total time
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$tag -- python3 $R/$@ > $R/gpurun_out/st_$tag.log 2>&1
cd $R
python3 - $tag <<P
import csv,glob,sys
f=glob.glob("gpurun_out/st_%s/**/*kernel_stats.csv" % sys.argv[1],recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]: print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
P
