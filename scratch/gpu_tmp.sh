R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python3 scratch/xtp_time.py 16 4000 16 2>&1 | grep -v amdgpu.ids | head -1
CONFIG=qwen3-4b VARIANTS="12x6" NSEQ="8" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
O=$R/gpurun_out/r05x; rm -rf $O/pmc_f; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/scratch/xtp_time.py 16 4000 4 > $O/f.log 2>&1; echo "fetch rc=$?")
python3 - <<'PY'
import csv, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r05x")
v=[]
for f in glob.glob(O+"/pmc_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "xengine_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": v.append(float(r["Counter_Value"]))
print("fetch bytes per launch", sum(v)/len(v)*2048, len(v))
PY
find $O -name "*.csv" -size +1M -delete
timeout 1500 python -m pytest tests/test_gpu_xengine.py tests/test_gpu_tp.py -x -q -k "gqa4 or eight_xcds_of_one or long_context or eight_query" 2>&1 | tail -3
