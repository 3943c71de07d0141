"""LDS / wait counters per (kernel, grid) of a scratch/pmc_run.sh collection: usage pmc_lds.py <tag> <kernel substring>"""
import csv,glob,collections,os,sys
tag,sub=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for ab in "ab":
    fs=sorted(glob.glob("gpurun_out/pmc_%s/%s/**/*counter_collection.csv"%(tag,ab),recursive=True),key=os.path.getmtime)
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        if sub in r["Kernel_Name"]:
            k=(r["Kernel_Name"][9:45], r["Grid_Size"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    c={a:sum(b)/len(b) for a,b in c.items()}
    print(k, "cycles %.0f"%(c["SQ_BUSY_CYCLES"]/32), "LDS conflict frac %.3f"%(c["SQ_LDS_BANK_CONFLICT"]/c["SQ_LDS_IDX_ACTIVE"]), "wait_lds %.2f"%(c["SQ_WAIT_INST_LDS"]/c["SQ_WAVE_CYCLES"]), "wait any %.2f"%(c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"]), "LDS insts %.0f"%c["SQ_INSTS_LDS"], "idx_active %.0f"%c["SQ_LDS_IDX_ACTIVE"], "valu %.0f"%c["SQ_INSTS_VALU"])
