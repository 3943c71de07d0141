"""Temporary instrumentation of gemm_direct_kernel (kf_gemm.hip) -- NOT for commit: s_memrealtime stamps per workgroup (wave 0, and the last wave for the first two),
dumped by gm_launch after every 4th launch of a shape.  `git checkout koifish_amd/csrc` afterwards.
stamps: 0 kernel entry, 1 first group's loads issued, 2 k loop done (MFMAs issued), 3 barrier behind the partial sums, 4 partials added, 5 epilogue stores issued, 6 stores acknowledged"""
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(root, 'koifish_amd/csrc/kf_gemm_common.h')
s = open(p).read()
s = s.replace("    long long xldy[2];\n};", "    long long xldy[2];\n    unsigned long long* dbg;\n};", 1)
open(p, 'w').write(s)
p = os.path.join(root, 'koifish_amd/csrc/kf_gemm.hip')
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    const int r = lane & 31, h = lane >> 5;\n    GemmArgs a = a0;\n    int rb = blockIdx.x;",
    "    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    const int r = lane & 31, h = lane >> 5;\n    GemmArgs a = a0;\n    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n#define STAMP(i) ts[i] = __builtin_amdgcn_s_memrealtime()\n    STAMP(0);\n    int rb = blockIdx.x;")
rep("    if (u0 < nunit) gload(u0, gc);\n", "    if (u0 < nunit) gload(u0, gc);\n    STAMP(1);\n")
rep("        if (more) gc = gn;\n    }\n    // fixed-order sum over the k-slices", "        if (more) gc = gn;\n    }\n    STAMP(2);\n    if (a0.dbg && wave == NW - 1 && lane == 0) { a0.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 8] = ts[0]; a0.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 9] = ts[1]; a0.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 10] = ts[2]; }\n    // fixed-order sum over the k-slices")
rep("    __syncthreads();\n    if (wave > 0) return;\n#pragma unroll 1 /* one partial at a time", "    __syncthreads();\n    STAMP(3);\n    if (wave > 0) return;\n#pragma unroll 1 /* one partial at a time")
rep("    if (!PAIRED) {\n        gemm_epilogue<TB>(acc, a, tok0, row_base, r, h);\n    } else {", "    STAMP(4);\n    if (!PAIRED) {\n        gemm_epilogue<TB>(acc, a, tok0, row_base, r, h);\n    } else {")
# end of kernel: after the paired else-branch closes
i = s.index("static int gm_fmt_of(int type) {")
j = s.rindex("}\n", 0, i)
s = s[:j] + "    STAMP(5);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    STAMP(6);\n    if (a0.dbg && lane == 0) for (int i_ = 0; i_ < 7; i_++) a0.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + i_] = ts[i_];\n#undef STAMP\n" + s[j:]
rep("template <int FMT>\nstatic void gm_launch(const GemmArgs& a, int KS, dim3 grid, size_t smem, hipStream_t st) {\n",
"""template <int FMT>
static void gm_launch(const GemmArgs& a_in, int KS, dim3 grid, size_t smem, hipStream_t st) {
    GemmArgs a = a_in;
    static unsigned long long* dbg = nullptr;
    static int dbgon = -1;
    if (dbgon < 0) { dbgon = getenv("KF_GEMM_DBG") ? 1 : 0; if (dbgon) (void)hipMalloc(&dbg, 8 * 16 * 4096); }
    a.dbg = nullptr;
    const int nwg = (int)(grid.x * grid.y);
    if (dbgon && KS <= 0 && nwg <= 4096) { a.dbg = dbg; (void)hipMemsetAsync(dbg, 0, 8 * 16 * 4096, st); }
    struct Dump { const GemmArgs& a; int KS, nwg; hipStream_t st; unsigned long long* d;
        ~Dump() {
            if (!a.dbg) return;
            (void)hipStreamSynchronize(st);
            static unsigned long long hbuf[16 * 4096];
            (void)hipMemcpy(hbuf, d, sizeof(unsigned long long) * 16 * nwg, hipMemcpyDeviceToHost);
            static int cnt = 0;
            if (++cnt % 4) return;
            unsigned long long t0 = ~0ull, tl = 0;
            for (int w = 0; w < nwg; w++) { if (hbuf[w * 16] && hbuf[w * 16] < t0) t0 = hbuf[w * 16]; if (hbuf[w * 16 + 6] > tl) tl = hbuf[w * 16 + 6]; }
            fprintf(stderr, "gemm dbg M=%d K=%d n=%d KS=%d wgs=%d: first entry -> last store acknowledged %.2f us\\n", a.M, a.K, a.n, KS, nwg, (tl - t0) * 0.01);
            for (int w = 0; w < nwg; w += (nwg / 6 > 0 ? nwg / 6 : 1)) {
                fprintf(stderr, "  wg%4d wave0: entry %+6.2f", w, (hbuf[w * 16] - t0) * 0.01);
                for (int i = 1; i < 7; i++) fprintf(stderr, " s%d %+6.2f", i, (hbuf[w * 16 + i] - t0) * 0.01);
                fprintf(stderr, " | last wave: entry %+6.2f loads issued %+6.2f loop done %+6.2f\\n", (hbuf[w * 16 + 8] - t0) * 0.01, (hbuf[w * 16 + 9] - t0) * 0.01, (hbuf[w * 16 + 10] - t0) * 0.01);
            }
        }
    } dump{a, KS, nwg, st, dbg};
""")
if "#include <stdio.h>" not in s:
    s = s.replace("#include <stdlib.h>", "#include <stdio.h>\n#include <stdlib.h>", 1)
open(p, 'w').write(s)
print("patched")
