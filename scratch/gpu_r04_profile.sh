#!/bin/bash
# Round-4 evidence run: one gpurun call.  Summaries land in gpurun_out/r04p/ (copied into profiles/ by hand afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cpu-fp16-steps 0 --side-legs "" > $O/bench_traced.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_f.log 2>&1; echo "pmc fetch rc=$?"; tail -1 $O/pmc_f.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_w.log 2>&1; echo "pmc write rc=$?"; tail -1 $O/pmc_w.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/scratch/ub_engine.py 2037 > $O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pf128 -- python3 $R/scratch/prefill_prof.py 128 > $O/pf128.log 2>&1; echo "pf128 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pf2047 -- python3 $R/scratch/prefill_prof.py 2047 > $O/pf2047.log 2>&1; echo "pf2047 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -- python3 $R/bench.py --lean --layers 1bit --sparse 0.2 --steps 512 --warmup 64 > $O/c5.log 2>&1; echo "c5 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 $R/bench.py --leg config3 > $O/c3.log 2>&1; echo "c3 rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --lean --config qwen3-32b --steps 64 --warmup 16 > $O/c4.log 2>&1; echo "c4 rc=$?"
cd $R
timeout 300 python3 scratch/ub_engine.py 2037 > $O/ub_engine_plain.log 2>&1; tail -1 $O/ub_engine_plain.log
ALG=$(tail -1 $O/ub_engine_plain.log | sed 's/.*per launch \([0-9]*\) .*/\1/')
python3 scratch/pmc_engine_json.py $O 2037 $ALG $O/r04_pmc_engine.json > /dev/null
python3 scratch/pmc_engine_sq_json.py $O/pmc_sq $O/r04_pmc_engine_sq.json "rocprofv3 --kernel-trace --pmc SQ_* (one pass) around scratch/ub_engine.py 2037, canonical order (the library default)" > /dev/null
STAMP_WG=77 STAMP_LAYERS=1 python3 scratch/eng_stamps.py 2040 > $O/stamps_canonical.txt 2>&1
CANON=0 STAMP_WG=77 STAMP_LAYERS=1 python3 scratch/eng_stamps.py 2040 > $O/stamps_fast.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_handoff3 scratch/ub_handoff3.hip 2>/dev/null && timeout 600 /tmp/ub_handoff3 > $O/handoff_floor.txt 2>&1
timeout 1500 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; tail -c 300 $O/bench_driver_flags.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O; find $O -name "*stats*.csv" | head -20
