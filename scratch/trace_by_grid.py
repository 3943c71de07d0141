"""average duration per (kernel, grid) of the newest kernel trace of a scratch/prof_stats.sh run: trace_by_grid.py <tag> <substring>"""
import csv,glob,collections,os,sys
f=sorted(glob.glob("gpurun_out/st_%s/**/*kernel_trace.csv"%sys.argv[1],recursive=True), key=os.path.getmtime)[-1]
d=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if sys.argv[2] in n:
        d.setdefault((n[:70], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])),[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items(): print(k, len(v), "%.1f us"%(sum(v)/len(v)/1e3))
