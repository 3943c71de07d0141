#!/bin/bash
# 1-bit / ternary / 4-bit large mat-vec timings + the bit-identity tests of the table forms
timeout 600 python3 -m pytest tests/test_gpu_q4_variants.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3
timeout 300 python3 scratch/ub_big_q4.py 1bit ternary q4 2>&1 | grep -v amdgpu
