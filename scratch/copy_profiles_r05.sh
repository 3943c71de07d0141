#!/bin/bash
# after scratch/gpu_r05_profile.sh: copy the summaries of gpurun_out/r05p/ into profiles/ under their tracked names; $1 = name of the bench line (e.g. r05a_bench_line.json)
O=gpurun_out/r05p; P=profiles
cp $(ls -t $(find $O/trace_bench -name "*kernel_stats.csv") | head -1) $P/r05_bench_kernel_stats.csv
cp $(ls -t $(find $O/trace_c3 -name "*kernel_stats.csv") | head -1) $P/r05_config3_train_step_kernel_stats.csv
for NS in 16 8; do
  python3 scratch/pmc_xengine_json.py $O/x$NS $NS 4 $(tail -1 $O/ub_xengine_$NS.log | sed 's/.*bytes per launch \([0-9]*\) .*/\1/') $P/r05_pmc_xengine_$NS > /dev/null
done
cp $O/xengine_stamps.txt $P/r05_xengine_stamps.txt
tail -1 $O/bench_driver_flags.json > $P/$1
[ -f $O/xtp_stamps.txt ] && cp $O/xtp_stamps.txt $P/r05_xtp_stamps.txt
[ -f $O/xengine_shapes.txt ] && cp $O/xengine_shapes.txt $P/r05_xengine_shapes.txt
X=$(ls -t $(find $O/trace_xtp -name "*kernel_stats.csv" 2>/dev/null) 2>/dev/null | head -1); [ -n "$X" ] && cp $X $P/r05_xtp_kernel_stats.csv
cp $O/r05_pmc_xengine_16_sq.json $O/r05_pmc_xengine_8_sq.json $P/ 2>/dev/null
