"""profiles/r04_pmc_engine_sq.json from the SQ counter pass of scratch/gpu_r04_profile.sh (rows of engine_kernel only).  usage: pmc_engine_sq_json.py <pmc dir> <out.json> <note>"""
import collections, csv, glob, json, sys
d, out, note = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "engine_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in acc.items()}
res = {"kernel": "engine_kernel (one decode step in one launch)", "method": note, "counters": {k: round(v, 1) for k, v in sorted(c.items())}, "launches_averaged": len(acc.get("SQ_WAVES", []))}
if "SQ_BUSY_CYCLES" in c:
    res["kernel_cycles"] = round(c["SQ_BUSY_CYCLES"] / 32.0)
    res["valu_busy_fraction_of_simd"] = round(c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (c["SQ_BUSY_CYCLES"] / 32.0), 3)
    res["wave_cycles_waiting_fraction"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    res["wave_cycles_issue_stalled_fraction"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
