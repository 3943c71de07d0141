"""profiles/r05_pmc_xengine*.json from the counter passes of scratch/gpu_r05_profile.sh (rows of xengine_kernel only).
usage: pmc_xengine_json.py <dir with pmc_f / pmc_w / pmc_sq> <n_seq> <steps per launch> <algorithmic bytes per launch> <out prefix>"""
import collections, csv, glob, json, sys
d, n_seq, steps, alg, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5]


def rows(sub):
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if "xengine_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


f, w, sq = rows("pmc_f"), rows("pmc_w"), rows("pmc_sq")
fb = f.get("FETCH_SIZE", (0.0, 0))
wb = w.get("WRITE_SIZE", (0.0, 0))
res = {"kernel": "kf::xengine_kernel: %d independent sequences (%s per XCD), %d decode steps per launch (embedding row + 28 layers + final norm + LM head + pick per sequence and step)" % (
           n_seq, "two" if n_seq > 8 else "one", steps),
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around scratch/ub_xengine.py; FETCH_SIZE x 1024 x 2 (gfx950 counts a 128-byte request as 64: "
                 "MI355X_MICROARCH.md), WRITE_SIZE x 1024; averages over %d / %d launches" % (fb[1], wb[1]),
       "fetch_bytes_per_launch": fb[0] * 1024 * 2, "write_bytes_per_launch": wb[0] * 1024, "hbm_bytes_per_launch": fb[0] * 1024 * 2 + wb[0] * 1024,
       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((fb[0] * 1024 * 2 + wb[0] * 1024) / alg, 3) if alg else None,
       "note": "FETCH_SIZE counts what the XCDs' L2s request from the fabric: every decoder streams the layer weights and the head through its OWN XCD's L2 (4 MB against an 8.4 MB "
               "layer), so the counter sees the sum of the sequences' algorithmic bytes (+ the hand-off sweeps); whether the 256 MB memory-side cache or HBM answers a request that "
               "another XCD made a moment ago is below this counter"}
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1))
if sq:
    c = {k: v[0] for k, v in sq.items()}
    r2 = {"kernel": res["kernel"], "method": "rocprofv3 --kernel-trace --pmc SQ_* (one pass) around scratch/ub_xengine.py", "counters": {k: round(v, 1) for k, v in sorted(c.items())},
          "launches_averaged": sq.get("SQ_WAVES", (0, 0))[1]}
    if "SQ_BUSY_CYCLES" in c:
        r2["kernel_cycles"] = round(c["SQ_BUSY_CYCLES"] / 32.0)
        r2["valu_instructions_per_sequence_step"] = round(c["SQ_INSTS_VALU"] / (n_seq * steps))
        r2["valu_busy_fraction_of_simd"] = round(c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (c["SQ_BUSY_CYCLES"] / 32.0), 3)
        r2["wave_cycles_waiting_fraction"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        r2["wave_cycles_issue_stalled_fraction"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    json.dump(r2, open(out + "_sq.json", "w"), indent=1)
    print(json.dumps(r2, indent=1))
