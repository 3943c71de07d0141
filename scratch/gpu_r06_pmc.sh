#!/bin/bash
# Round-6 counter passes of the XCD-confined engines (batched forms): SQ instruction / wait counters and LDS counters, per NSEQ.  Summaries: gpurun_out/r06p/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for NS in ${NSEQS:-32 16}; do
  NSEQ=$NS timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/x$NS/pmc_sq -- python3 $R/scratch/ub_xengine.py 2037 4 3 > $O/x${NS}_sq.log 2>&1; echo "pmc sq $NS rc=$?"
  NSEQ=$NS timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $O/x$NS/pmc_lds -- python3 $R/scratch/ub_xengine.py 2037 4 3 > $O/x${NS}_lds.log 2>&1; echo "pmc lds $NS rc=$?"
  if [ -n "$TRAFFIC" ]; then
    NSEQ=$NS timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/x$NS/pmc_f -- python3 $R/scratch/ub_xengine.py 2037 4 3 > $O/x${NS}_f.log 2>&1; echo "pmc fetch $NS rc=$?"
    NSEQ=$NS timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/x$NS/pmc_w -- python3 $R/scratch/ub_xengine.py 2037 4 3 > $O/x${NS}_w.log 2>&1; echo "pmc write $NS rc=$?"
  fi
done
cd $R
for NS in ${NSEQS:-32 16}; do
  NSEQ=$NS timeout 300 python3 scratch/ub_xengine.py 2037 4 3 > $O/ub_xengine_$NS.log 2>&1; tail -1 $O/ub_xengine_$NS.log
  ALG=$(tail -1 $O/ub_xengine_$NS.log | sed 's/.*bytes per launch \([0-9]*\) .*/\1/')
  python3 scratch/pmc_xengine_json.py $O/x$NS $NS 4 $ALG $O/r06_pmc_xengine_$NS | tail -25
  python3 - <<PY
import collections, csv, glob
acc = collections.defaultdict(list)
for f in glob.glob("$O/x$NS/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "xengine_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in acc.items()}
print("LDS pass NSEQ=$NS:", {k: round(v) for k, v in sorted(c.items())})
if c:
    cyc = c["SQ_BUSY_CYCLES"] / 32.0
    print("  kernel cycles %.0f; LDS insts per sequence-step %.2f M; LDS idx active / (kernel cycles x 256 CUs) = %.3f; bank conflict fraction %.3f" % (
        cyc, c["SQ_INSTS_LDS"] / ($NS * 4) / 1e6, c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)))
PY
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
