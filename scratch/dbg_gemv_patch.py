"""Temporary instrumentation of kf_gemv.hip (NOT for commit): wall-clock + cycle stamps per workgroup, dumped by gemv_launch when
KF_GEMV_DBG is set (eager launches only).  git checkout the two files afterwards."""
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(root, 'koifish_amd/csrc/kf_kernels.h')
s = open(p).read()
s = s.replace("    float* amax_val;\n    int* amax_idx;\n};", "    float* amax_val;\n    int* amax_idx;\n    long long* dbg;\n};", 1)
open(p, 'w').write(s)
p = os.path.join(root, 'koifish_amd/csrc/kf_gemv.hip')
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
S = '''
#define STAMP(i) do { ts[i] = wall_clock64(); cs[i] = __builtin_readcyclecounter(); } while (0)
#define FLUSH() do { if (a.dbg && threadIdx.x == 0) for (int i_ = 0; i_ < 6; i_++) a.dbg[blockIdx.x * 8 + i_] = (i_ == 0 || !ts[i_]) ? ts[i_] : ((cs[i_] - cs[i_ - 1]) << 32) | (ts[i_] - ts[0]); } while (0)
'''
rep("template <int FMT, int G, int MODE>\n__global__ void __launch_bounds__(256) gemv_kernel(const GemvArgs a) {\n", S + "template <int FMT, int G, int MODE>\n__global__ void __launch_bounds__(256) gemv_kernel(const GemvArgs a) {\n    long long ts[6] = {0, 0, 0, 0, 0, 0}, cs[6] = {0, 0, 0, 0, 0, 0};\n    STAMP(0);\n")
rep("    const int pos = a.d_pos ? *a.d_pos : a.pos;\n\n    // ---- prologue: stage x", "    const int pos = a.d_pos ? *a.d_pos : a.pos;\n    STAMP(1);\n\n    // ---- prologue: stage x")
rep("        __syncthreads();\n    }\n\n    // ---- main: pipelined", "        __syncthreads();\n    }\n    STAMP(2);\n\n    // ---- main: pipelined")
rep("        cur = nxt;\n        if (++it == iters) it = 0, bi++;\n    }\n", "        if (k == 0) STAMP(3);\n        cur = nxt;\n        if (++it == iters) it = 0, bi++;\n    }\n    STAMP(4);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    STAMP(5);\n    FLUSH();\n")
rep("int gemv_launch(hipStream_t st, GemvLaunch& L) {\n    GemvArgs& a = L.args;\n", """int gemv_launch(hipStream_t st, GemvLaunch& L) {
    GemvArgs& a = L.args;
    static long long* dbg = nullptr;
    static int dbgon = -1;
    if (dbgon < 0) { dbgon = getenv("KF_GEMV_DBG") ? 1 : 0; if (dbgon) (void)hipMalloc(&dbg, 8 * 8 * 8192); }
    a.dbg = nullptr;
    if (dbgon) { a.dbg = dbg; (void)hipMemsetAsync(dbg, 0, 8 * 8 * 8192, st); }
""")
rep("    if (L.mode == GEMV_ARGMAX && blocks > KF_MAX_ARGMAX_PARTIALS) return KF_INTERNAL_ERR;\n", "    if (L.mode == GEMV_ARGMAX && blocks > KF_MAX_ARGMAX_PARTIALS) return KF_INTERNAL_ERR;\n    const int dbg_blocks = blocks;\n")
# dump after launch: find the end of the switch
i = s.index("    dim3 grid(blocks);\n    switch (fmt) {")
j = s.index("    }\n", s.index("default:", i)) + 6
s = s[:j] + """    if (a.dbg && dbg_blocks <= 8192) {
        (void)hipStreamSynchronize(st);
        static long long h[8 * 8192];
        (void)hipMemcpy(h, dbg, sizeof(long long) * 8 * dbg_blocks, hipMemcpyDeviceToHost);
        long long t0 = h[0], tl = 0;
        for (int w = 0; w < dbg_blocks; w++) { if (h[w * 8] < t0) t0 = h[w * 8]; }
        for (int w = 0; w < dbg_blocks; w++) { long long e = h[w * 8] - t0 + (h[w * 8 + 5] & 0xffffffff); if (e > tl) tl = e; }
        static int cnt = 0;
        if (++cnt % 4 == 0) {
            fprintf(stderr, "gemv dbg M=%d K=%d mode=%d G=%d blocks=%d iters=%d spw=%d: last end %lld (10 ns ticks)\\n", a.job[0].M, K, L.mode, G, dbg_blocks, a.iters, a.spw, tl);
            for (int w = 0; w < dbg_blocks; w += (dbg_blocks / 6 > 0 ? dbg_blocks / 6 : 1)) {
                fprintf(stderr, " wg%4d: %5lld", w, h[w * 8] - t0);
                for (int i = 1; i < 6; i++) fprintf(stderr, " %5lld(%5lldc)", h[w * 8 + i] & 0xffffffff, h[w * 8 + i] >> 32);
                fprintf(stderr, "\\n");
            }
        }
    }
""" + s[j:]
if "#include <stdio.h>" not in s:
    s = s.replace("#include <stdlib.h>", "#include <stdio.h>\n#include <stdlib.h>", 1)
open(p, 'w').write(s)
