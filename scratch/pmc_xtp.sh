#!/bin/bash
# (GPU box) counter passes around the TP-over-XCDs engine: FETCH_SIZE, WRITE_SIZE, SQ_* of kf::xengine_kernel<XCfg<..., TP>> -- 16 layers, 4 tokens per launch at 4 k keys
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/scratch/xtp_time.py 16 4000 4 > $O/f.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/scratch/xtp_time.py 16 4000 4 > $O/w.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/scratch/xtp_time.py 16 4000 4 > $O/sq.log 2>&1; echo "sq rc=$?"
cd $R
python3 - <<'PY'
import collections, csv, glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r05x")
def rows(sub):
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (O, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if "xengine_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
f, w, sq = rows("pmc_f"), rows("pmc_w"), rows("pmc_sq")
res = {"kernel": "kf::xengine_kernel<XCfg<..., TP>>: the eight TP ranks of a 16-layer Qwen3-32B-shaped model as the eight XCDs of one launch, 4 tokens per launch at positions 4000..4003",
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes) around scratch/xtp_time.py 16 4000 4; FETCH_SIZE x 1024 x 2 (gfx950), WRITE_SIZE x 1024",
       "fetch_bytes_per_launch": f.get("FETCH_SIZE", (0, 0))[0] * 2048, "write_bytes_per_launch": w.get("WRITE_SIZE", (0, 0))[0] * 1024, "launches": f.get("FETCH_SIZE", (0, 0))[1],
       "sq": {k: round(v[0], 1) for k, v in sorted(sq.items())}}
c = {k: v[0] for k, v in sq.items()}
if "SQ_BUSY_CYCLES" in c:
    res["valu_busy_fraction_of_simd"] = round(c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (c["SQ_BUSY_CYCLES"] / 32.0), 3)
    res["wave_cycles_waiting_fraction"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
json.dump(res, open(O + "/r05_pmc_xtp.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
