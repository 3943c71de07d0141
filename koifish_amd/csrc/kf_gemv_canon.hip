// kf_gemv_canon.hip -- the mat-vec kernels of kf_gemv.hip in the CANONICAL summation order (oracle/kf_oracle.c section 4c): every pair of products is two
// v_fma_f32 (low element first) instead of one v_dot2c_f32_bf16, so that the host reproduces every output bit with fmaf.  Same kernels, geometry and launcher:
// the source is kf_gemv.hip, compiled a second time.
#define KF_GEMV_CANON 1
#include "kf_gemv.hip"
