"""Temporary instrumentation of kf_attn.hip (NOT for commit): wall-clock + cycle stamps per workgroup, dumped by attn_launch when
KF_ATTN_DBG is set.  Usage: python scratch/dbg_attn_patch.py; python koifish_amd/build.py; KF_ATTN_DBG=1 python scratch/dbg_attn.py;
git checkout koifish_amd/csrc/kf_attn.hip koifish_amd/csrc/kf_kernels.h"""
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(root, 'koifish_amd/csrc/kf_kernels.h')
s = open(p).read()
s = s.replace("    long long q_stride; /* elements between the q (and out) rows of consecutive tokens */\n};", "    long long q_stride; /* elements between the q (and out) rows of consecutive tokens */\n    long long* dbg;\n};", 1)
open(p, 'w').write(s)
p = os.path.join(root, 'koifish_amd/csrc/kf_attn.hip')
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
S = '''
#define STAMP(i) do { ts[i] = wall_clock64(); cs[i] = __builtin_readcyclecounter(); } while (0)
#define FLUSH() do { if (a.dbg && tid == 0) for (int i_ = 0; i_ < 10; i_++) a.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + i_] = (i_ == 0 || !ts[i_]) ? ts[i_] : ((cs[i_] - cs[i_ - 1]) << 32) | (ts[i_] - ts[0]); } while (0)
'''
rep("template <int GQ, int NW, int HD>\n__global__ void __launch_bounds__(NW * 64) attn_kernel", S + "template <int GQ, int NW, int HD>\n__global__ void __launch_bounds__(NW * 64) attn_kernel")
rep("    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n", "    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, cs[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};\n    STAMP(0);\n")
rep("    const int len = pos + 1;\n", "    const int len = pos + 1;\n    STAMP(1);\n")
rep("        __syncthreads();\n        if (own_new) {", "        __syncthreads();\n        STAMP(2);\n        if (own_new) {")
rep("            __syncthreads(); /* previous batch's readers of wmax are done */\n", "            STAMP(3);\n            __syncthreads(); /* previous batch's readers of wmax are done */\n")
rep("        // ---- sum the key groups", "        STAMP(4);\n        // ---- sum the key groups")
rep("    if (nsp == 1) return;\n", "    STAMP(5);\n    if (nsp == 1) { FLUSH(); return; }\n")
rep("    __syncthreads();\n    if (tid == 0) {\n        const int old", "    __syncthreads();\n    STAMP(6);\n    if (tid == 0) {\n        const int old")
rep("    if (!flag[0]) return;\n", "    STAMP(7);\n    if (!flag[0]) { FLUSH(); return; }\n")
rep("#pragma unroll\n    for (int e = 0; e < NV; e++) {\n        const int i = tid + e * NT;\n        const float Mx", "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    STAMP(8);\n#pragma unroll\n    for (int e = 0; e < NV; e++) {\n        const int i = tid + e * NT;\n        const float Mx")
rep("    if (tid == 0) __hip_atomic_store(counter, 0,", "    STAMP(9);\n    FLUSH();\n    if (tid == 0) __hip_atomic_store(counter, 0,")
rep("    dim3 grid(nsp, a.n_kv, a.n_tok);\n", """    dim3 grid(nsp, a.n_kv, a.n_tok);
    static long long* dbg = nullptr;
    static int dbgon = -1;
    if (dbgon < 0) { dbgon = getenv("KF_ATTN_DBG") ? 1 : 0; if (dbgon) (void)hipMalloc(&dbg, 16 * 8 * 4096); }
    a.dbg = nullptr;
    if (dbgon && a.n_tok == 1) { a.dbg = dbg; (void)hipMemsetAsync(dbg, 0, 16 * 8 * 4096, st); }
""")
rep("    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;\n}\n\nint qknorm_rope_launch", """    if (a.dbg) {
        (void)hipStreamSynchronize(st);
        static long long h[16 * 4096];
        const int nwg = nsp * a.n_kv;
        (void)hipMemcpy(h, dbg, sizeof(long long) * 16 * nwg, hipMemcpyDeviceToHost);
        long long t0 = h[0];
        for (int w = 0; w < nwg; w++) if (h[w * 16] < t0) t0 = h[w * 16];
        static int cnt = 0;
        if (++cnt % 8 == 0) {
            fprintf(stderr, "attn dbg pos=%d nsp=%d (10 ns ticks rel. first start)\\n", a.pos, nsp);
            int shown = 0;
            for (int w = 0; w < nwg; w++) {
                if (!(w < 2 || h[w * 16 + 9])) continue;
                if (++shown > 6) break;
                fprintf(stderr, " wg%3d: %5lld", w, h[w * 16] - t0);
                for (int i = 1; i < 10; i++) fprintf(stderr, " %5lld(%5lldc)", h[w * 16 + i] ? (h[w * 16 + i] & 0xffffffff) : -1, h[w * 16 + i] >> 32);
                fprintf(stderr, "\\n");
            }
        }
    }
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

int qknorm_rope_launch""")
if "#include <stdio.h>" not in s:
    s = s.replace("#include <stdlib.h>", "#include <stdio.h>\n#include <stdlib.h>", 1)
open(p, 'w').write(s)
