"""Mean per-launch counters of the kernels whose name contains <substr> from rocprofv3 counter_collection csv files under a directory."""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
            acc[(r["Kernel_Name"][:60], "_VGPR")] = [float(r["VGPR_Count"])]
for k in sorted(acc):
    v = acc[k]
    print("%-62s %-24s %14.0f  (n=%d)" % (k[0], k[1], sum(v) / len(v), len(v)))
