#!/bin/bash
# after scratch/gpu_r04_profile.sh: copy the summaries of gpurun_out/r04p/ into profiles/ under their tracked names; $1 = name of the bench line (e.g. r04c_bench_line.json)
O=gpurun_out/r04p; P=profiles
cp $(ls -t $(find $O/trace_bench -name "*kernel_stats.csv") | head -1) $P/r04_bench_kernel_stats.csv
cp $(ls -t $(find $O/trace_pf128 -name "*kernel_stats.csv") | head -1) $P/r04_prefill128_kernel_stats.csv
cp $(ls -t $(find $O/trace_pf2047 -name "*kernel_stats.csv") | head -1) $P/r04_prefill2047_kernel_stats.csv
cp $(ls -t $(find $O/trace_c3 -name "*kernel_stats.csv") | head -1) $P/r04_config3_train_step_kernel_stats.csv
cp $(ls -t $(find $O/trace_c4 -name "*kernel_stats.csv") | head -1) $P/r04_config4_one_gpu_kernel_stats.csv
cp $(ls -t $(find $O/trace_c5 -name "*kernel_stats.csv") | head -1) $P/r04_config5_sparse_1bit_kernel_stats.csv
cp $O/r04_pmc_engine.json $O/r04_pmc_engine_sq.json $P/
cp $O/stamps_canonical.txt $P/r04_engine_stamps_canonical.txt
cp $O/stamps_fast.txt $P/r04_engine_stamps_fast_order.txt
cp $O/handoff_floor.txt $P/r04_handoff_floor.txt
tail -1 $O/bench_driver_flags.json > $P/$1
