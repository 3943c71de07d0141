"""Average FETCH_SIZE / WRITE_SIZE per launch of one kernel from rocprofv3 --pmc counter_collection csv files -> json (gfx950: FETCH_SIZE counts
128-B requests as 64 B, so x 2; both counters are in KiB units).  usage: pmc_parse.py <kernel substring> <fetch.csv> <write.csv> <out.json> <algorithmic bytes> [target script]"""
import csv, json, sys
name, fcsv, wcsv, out, alg = sys.argv[1:6]
target = sys.argv[6] if len(sys.argv) > 6 else "scratch/ub_matvec_chain.py"
def avg(path, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if name in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)
f, nf = avg(fcsv, "FETCH_SIZE"); w, nw_ = avg(wcsv, "WRITE_SIZE")
d = {"kernel": name, "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around %s; FETCH_SIZE x 1024 x 2 (gfx950 correction), WRITE_SIZE x 1024; averages over %d / %d launches" % (target, nf, nw_),
     "fetch_bytes_per_launch": f * 1024 * 2, "write_bytes_per_launch": w * 1024, "hbm_bytes_per_launch": f * 2048 + w * 1024, "algorithmic_bytes_per_launch": float(alg)}
json.dump(d, open(out, "w"), indent=1); print(d)
