"""compact json of the SQ / L2 counter passes (scratch/pmc_run.sh, scratch/pmc_l2.sh) per (kernel, grid): pmc_json.py <tag> <kernel substring> <out.json> <note>"""
import csv, glob, collections, json, os, sys
tag, sub, out, note = sys.argv[1:5]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob("gpurun_out/pmc_%s/*" % tag):
    fs = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        if sub in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][:110], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"note": note, "kernels": []}
for (name, wgs), c in acc.items():
    c = {k: sum(v) / len(v) for k, v in c.items()}
    e = {"kernel": name, "workgroups": wgs, "counters": {k: round(v, 1) for k, v in sorted(c.items())}}
    if "SQ_BUSY_CYCLES" in c:
        cyc = c["SQ_BUSY_CYCLES"] / 32.0
        e["kernel_cycles"] = round(cyc)
        e["valu_busy_fraction_of_simd"] = round(c.get("SQ_INSTS_VALU", 0) * 4.0 / 1024.0 / cyc, 3)
        e["wave_cycles_waiting_fraction"] = round(c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3)
    if "SQ_LDS_IDX_ACTIVE" in c:
        e["lds_bank_conflict_fraction_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c["SQ_LDS_IDX_ACTIVE"], 1), 3)
    if "TCC_HIT_sum" in c:
        e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0), 1), 3)
    res["kernels"].append(e)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
