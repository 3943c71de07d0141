#!/bin/bash
# round 4, second GPU call: suite; self-calibrated hand-off delays (kf_engine_tune); K/V tiles of the poller requested early
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
V=$R/scratch/variants
timeout 1200 python scratch/eng_ab.py "KF_LIB_DIR=$V/nokvearly CANON=1" "KF_LIB_DIR=$V/nokvearly CANON=0" "CANON=1" "CANON=0" "CANON=1 TUNE=1" "CANON=0 TUNE=1" "CANON=1 TUNE=3" "CANON=0 TUNE=3" "KF_LIB_DIR=$V/prep CANON=1 TUNE=3" "KF_LIB_DIR=$V/prep CANON=0 TUNE=3" "KF_LIB_DIR=$V/nokvearly CANON=1 TUNE=3" "KF_LIB_DIR=$V/nokvearly CANON=0 TUNE=3" > $O/ab.txt 2>&1
cat $O/ab.txt
CANON=1 STAMP_WG=77 STAMP_LAYERS=1 timeout 300 python scratch/eng_stamps.py 2040 > $O/stamps_canonical.txt 2>&1
CANON=0 STAMP_WG=77 STAMP_LAYERS=1 timeout 300 python scratch/eng_stamps.py 2040 > $O/stamps_fast.txt 2>&1
tail -4 $O/stamps_canonical.txt
