"""rocprofv3 --pmc passes (scratch/pmc_run.sh) of one mat-vec kernel -> a small json with the derived figures DESIGN.md quotes.
usage: pmc_summary.py <pmc dir> <kernel substring> <weights per launch> <out.json> <note>"""
import collections, csv, glob, json, sys
d, sub, nweights, out, note = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
acc = collections.defaultdict(list)
name = None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_VGPR"] = [float(r["VGPR_Count"])]
c = {k: sum(v) / len(v) for k, v in acc.items()}
waves_per_simd = c["SQ_WAVES"] / 1024.0
res = {"kernel": name, "launches_averaged": len(acc["SQ_WAVES"]), "note": note, "counters": {k: round(v, 1) for k, v in sorted(c.items())},
       "valu_instructions_per_weight": round(c["SQ_INSTS_VALU"] * 64 / nweights, 2),
       # wall clock of the launch in SIMD cycles ~ SQ_BUSY_CYCLES / 32 shader engines; a wave-64 VALU instruction occupies its SIMD for 4 cycles
       "kernel_cycles": round(c["SQ_BUSY_CYCLES"] / 32.0), "valu_busy_fraction_of_simd": round(c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (c["SQ_BUSY_CYCLES"] / 32.0), 3),
       "wave_cycles_waiting_fraction": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3), "wave_cycles_issue_stalled_fraction": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
       "lds_bank_conflict_fraction_of_lds_cycles": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 3), "waves": c["SQ_WAVES"], "waves_per_simd": round(waves_per_simd, 2)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
