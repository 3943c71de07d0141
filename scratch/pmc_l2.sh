#!/bin/bash
# usage: pmc_l2.sh <tag> <script> [args...] -> L2 hit/miss + fetch size per kernel (gpurun_out/pmc_<tag>/l2)
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
# one small counter set per pass: a set the hardware cannot collect at once aborts rocprofv3 and leaves the child hanging until the timeout
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag/l2_${set%% *} -- python3 $R/$@ > $R/gpurun_out/pmc_$tag.l2.log 2>&1
done
cd $R
python3 - $tag <<P
import csv,glob,collections,os,sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob("gpurun_out/pmc_%s/l2_*"%sys.argv[1]):
    fs=sorted(glob.glob(d+"/**/*counter_collection.csv",recursive=True),key=os.path.getmtime)
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        if "kf::" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    c={a:sum(b)/len(b) for a,b in c.items()}
    if c.get("TCC_HIT_sum",0)+c.get("TCC_MISS_sum",0) < 1e5: continue
    print(k, "hit %.3g miss %.3g hit-rate %.3f FETCH_SIZE %.3g"%(c["TCC_HIT_sum"],c["TCC_MISS_sum"],c["TCC_HIT_sum"]/max(1,c["TCC_HIT_sum"]+c["TCC_MISS_sum"]),c.get("FETCH_SIZE",-1)))
P
