#!/bin/bash
# round 4, third GPU call: suite (1-bit + sparse engine forms), A/B of the tuner, config 5 through the engine, the hand-off floor microbenchmark
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 900 python scratch/eng_ab.py "CANON=1 TUNE=0" "CANON=0 TUNE=0" "CANON=1 TUNE=1" "CANON=0 TUNE=1" "CANON=1 TUNE=2" "CANON=0 TUNE=2" > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 600 python bench.py --layers 1bit --sparse 0.2 --steps 512 --warmup 64 --lean > $O/config5.json 2> $O/config5.err; tail -c 1500 $O/config5.json
timeout 600 python bench.py --layers 1bit --steps 512 --warmup 64 --lean > $O/onebit_dense.json 2> $O/onebit_dense.err; tail -c 600 $O/onebit_dense.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_handoff3 scratch/ub_handoff3.hip && timeout 600 /tmp/ub_handoff3 > $O/handoff_floor.txt 2>&1; grep "uncached" $O/handoff_floor.txt | grep "jitter   0" | head -20
