#!/bin/bash
# round 4, first GPU call: the whole GPU suite on the new canonical order (two v_pk_fma_f32 chains per lane, default), then A/B of the engine's build knobs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
V=$R/scratch/variants
timeout 900 python scratch/eng_ab.py "KF_LIB_DIR=$V/base CANON=1" "KF_LIB_DIR=$V/base CANON=0" "KF_LIB_DIR=$V/kv CANON=1" "KF_LIB_DIR=$V/kv CANON=0" "CANON=1" "CANON=0" "KF_LIB_DIR=$V/widen3 CANON=1" "CANON=1" "CANON=0" > $O/ab.txt 2>&1
cat $O/ab.txt
CANON=1 STAMP_WG=77 STAMP_LAYERS=1 timeout 300 python scratch/eng_stamps.py 2040 > $O/stamps_canonical.txt 2>&1
CANON=0 STAMP_WG=77 STAMP_LAYERS=1 timeout 300 python scratch/eng_stamps.py 2040 > $O/stamps_fast.txt 2>&1
tail -4 $O/stamps_canonical.txt
