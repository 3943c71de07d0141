// ub_handoff4.hip -- round 3: a tag-free form of the all-to-all hand-off (2 bytes per bf16 element instead of a 4-byte {value, tag} granule).
// The sweep of ub_handoff3 is bound by bytes per CU (0.6 us + 43..75 ns per KB swept), so the bytes are halved: a vector element is plain bf16,
// "not yet written" is the bit pattern 0xFFFF (a NaN the producers never emit), and a producer puts the sentinel back into its own slots of a
// buffer two phases before that buffer is published again (4 buffers in rotation; in the engine the five vectors of a layer give the same slack).
// Variants:
//   LOADK 0 buffer_load sc1, 1 global_load sc0 sc1 (system), 2 plain volatile global load (uncached memory only)
//   POL   0 sweep at once and keep sweeping; 1 wait `delay_ns` after the own publish, then sweep; 2 poll ONE 16-byte piece of another XCD's
//         producer until it is valid, then sweep
//   NWV   waves per workgroup that share the sweep (each sweeps n / NWV elements), joined by a workgroup barrier
// Every phase every workgroup checks every element against the value its producer must have written.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ub_handoff4 scratch/ub_handoff4.hip && scratch/ub_handoff4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
// a 16-bit half equal to 0xFFFF anywhere in the dword?
__device__ __forceinline__ uint32_t has_sentinel(uint32_t v) { return (uint32_t)((v & 0xffffu) == 0xffffu) | (uint32_t)((v >> 16) == 0xffffu); }
__device__ __forceinline__ uint32_t bad4(u32x4 g) { return has_sentinel(g.x) | has_sentinel(g.y) | has_sentinel(g.z) | has_sentinel(g.w); }

struct Args {
    uint16_t* glob;  // [4][n] bf16 elements
    int* err;
    unsigned long long* log;
    int n, nphase, work_ns, nwg, delay_ns;
};

template <int LOADK>
__device__ __forceinline__ u32x4 ld16(const uint16_t* base, __amdgpu_buffer_rsrc_t rs, int byte_off) {
    if (LOADK == 0) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16 /* sc1 */));
    if (LOADK == 1) {
        u32x4 v;
        const char* p = reinterpret_cast<const char*>(base) + byte_off;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return *reinterpret_cast<const volatile u32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// value of element e of phase p (never 0xFFFF)
__device__ __forceinline__ uint16_t val_of(int e, int p) { return (uint16_t)(((e * 31 + p * 7) & 0x7fff) | 0x0001); }

// NLD: 16-byte loads per lane of ONE wave's share (n * 2 / NWV / 1024)
template <int NLD, int NWV, int LOADK, int POL>
__global__ void __launch_bounds__(NWV * 64) handoff_kernel(const Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wg = blockIdx.x, n = a.n;
    unsigned long long t0 = 0;
    const int per = n / a.nwg; /* elements per workgroup: a multiple of 8 */
    __shared__ int fail;
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    uint32_t nsweeps = 0;
    for (int p = 0; p < a.nphase; p++) {
        if ((p & 15) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break; /* a timed-out run ends quickly */
        uint16_t* const gbuf = a.glob + (size_t)(p & 3) * n;
        uint16_t* const rbuf = a.glob + (size_t)((p + 2) & 3) * n; /* published again two phases from now: every reader of its last content has finished */
        if (wave == 0) {
            if (4 * lane < per) { /* 8 bytes per lane: a workgroup owns per * 2 bytes (8 at n = 1024, 24 at n = 3072) */
                const u32x2 s = {0xffffffffu, 0xffffffffu};
                __builtin_amdgcn_raw_buffer_store_b64(s, rsrc(rbuf + wg * per + 4 * lane, 8), 0, 0, 16 /* sc1 */);
                const int e0 = wg * per + 4 * lane;
                u32x2 o;
                o.x = val_of(e0, p) | ((uint32_t)val_of(e0 + 1, p) << 16), o.y = val_of(e0 + 2, p) | ((uint32_t)val_of(e0 + 3, p) << 16);
                __builtin_amdgcn_raw_buffer_store_b64(o, rsrc(gbuf + e0, 8), 0, 0, 16 /* sc1 */);
            }
        }
        if (p == 4 && wg == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
        const __amdgpu_buffer_rsrc_t rs = rsrc(gbuf, (uint32_t)n * 2u);
        if (POL == 1 && a.delay_ns > 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime() + (unsigned long long)a.delay_ns / 10;
            while (__builtin_amdgcn_s_memrealtime() < t1) __builtin_amdgcn_s_sleep(1);
        }
        if (POL == 2) { /* one piece of the workgroup half the grid away (another XCD under round-robin placement) */
            const int src = ((wg + a.nwg / 2 + 1) % a.nwg) * per * 2;
            for (int spins = 0; spins < (1 << 16); spins++) {
                asm volatile("" ::: "memory");
                const u32x4 g = ld16<LOADK>(gbuf, rs, src & ~15);
                if (bad4(g) == 0) break;
            }
        }
        u32x4 g[NLD];
        const int w0 = wave * NLD * 1024; /* byte offset of this wave's share */
        int spins = 0;
        for (;; spins++) {
            asm volatile("" ::: "memory"); /* the poll loads are re-issued every pass */
            uint32_t bad = 0;
#pragma unroll
            for (int r = 0; r < NLD; r++) g[r] = ld16<LOADK>(gbuf, rs, w0 + (r * 64 + lane) * 16);
#pragma unroll
            for (int r = 0; r < NLD; r++) bad |= bad4(g[r]);
            if (__all(bad == 0)) break;
            if (spins > (1 << 12)) {
                if (lane == 0) atomicAdd(a.err, 1);
                break;
            }
        }
        if (wg == 0 && wave == 0) nsweeps += spins + 1;
        // check every element
        uint32_t wrong = 0;
#pragma unroll
        for (int r = 0; r < NLD; r++) {
            const int e0 = (w0 + (r * 64 + lane) * 16) / 2;
            const uint32_t d[4] = {g[r].x, g[r].y, g[r].z, g[r].w};
#pragma unroll
            for (int i = 0; i < 4; i++) wrong |= (uint32_t)((d[i] & 0xffffu) != val_of(e0 + 2 * i, p)) | (uint32_t)((d[i] >> 16) != val_of(e0 + 2 * i + 1, p));
        }
        if (wrong && fail == 0) fail = 1, atomicAdd(a.err + 2, 1);
        if (NWV > 1) __syncthreads();
        if (a.work_ns > 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime() + (unsigned long long)a.work_ns / 10;
            while (__builtin_amdgcn_s_memrealtime() < t1) __builtin_amdgcn_s_sleep(1);
        }
        if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the sentinel stores of this phase are acknowledged before the next publish */
    }
    if (wg == 0 && threadIdx.x == 0) a.log[0] = __builtin_amdgcn_s_memrealtime() - t0, a.log[1] = nsweeps;
}

template <int NLD, int NWV, int LOADK, int POL>
static void run(const char* what, uint16_t* buf, int* err, unsigned long long* log, hipStream_t st, int n, int nwg, int work, int delay) {
    const int nphase = 404, reps = 8;
    CK(hipMemset(err, 0, 64));
    double us = 0, ns = 0;
    for (int r = 0; r < reps; r++) {
        CK(hipMemset(buf, 0xff, 4 * 4096 * 2));
        Args a{buf, err, log, n, nphase, work, nwg, delay};
        hipLaunchKernelGGL((handoff_kernel<NLD, NWV, LOADK, POL>), dim3(nwg), dim3(NWV * 64), 0, st, a);
        CK(hipStreamSynchronize(st));
        unsigned long long t[2];
        CK(hipMemcpy(t, log, 16, hipMemcpyDeviceToHost));
        if (r >= 2) us += t[0] / 100.0 / (nphase - 4), ns += (double)t[1] / nphase;
    }
    int e[3];
    CK(hipMemcpy(e, err, 12, hipMemcpyDeviceToHost));
    printf("n %4d nwg %3d work %4d  %-34s waves %d load %d policy %d delay %4d: %.3f us per phase (hand-off alone %.3f), %.1f sweeps  timeouts %d wrong %d\n", n, nwg, work, what, NWV, LOADK, POL,
           delay, us / (reps - 2), us / (reps - 2) - work / 1000.0, ns / (reps - 2), e[0], e[2]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int* err;
    unsigned long long* log;
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&log, 64));
    uint16_t *coarse, *unc = nullptr;
    CK(hipMalloc(&coarse, 4 * 4096 * 2));
    if (hipExtMallocWithFlags((void**)&unc, 4 * 4096 * 2, hipDeviceMallocUncached) != hipSuccess) unc = nullptr;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int work : {0, 1000}) {
        // n = 1024 elements = 2 KB = 2 loads; n = 3072 = 6 KB = 6 loads
        run<2, 1, 0, 0>("coarse", coarse, err, log, st, 1024, 256, work, 0);
        if (!unc) continue;
        run<2, 1, 0, 0>("uncached", unc, err, log, st, 1024, 256, work, 0);
        run<2, 1, 1, 0>("uncached", unc, err, log, st, 1024, 256, work, 0);
        run<2, 1, 2, 0>("uncached", unc, err, log, st, 1024, 256, work, 0);
        run<1, 2, 0, 0>("uncached", unc, err, log, st, 1024, 256, work, 0);
        for (int d : {200, 400, 600}) run<2, 1, 0, 1>("uncached", unc, err, log, st, 1024, 256, work, d);
        run<2, 1, 0, 2>("uncached", unc, err, log, st, 1024, 256, work, 0);
        run<2, 1, 2, 2>("uncached", unc, err, log, st, 1024, 256, work, 0);
        run<6, 1, 0, 0>("coarse", coarse, err, log, st, 3072, 256, work, 0);
        run<6, 1, 0, 0>("uncached", unc, err, log, st, 3072, 256, work, 0);
        run<6, 1, 2, 0>("uncached", unc, err, log, st, 3072, 256, work, 0);
        run<3, 2, 0, 0>("uncached", unc, err, log, st, 3072, 256, work, 0);
        run<1, 6, 0, 0>("uncached", unc, err, log, st, 3072, 256, work, 0);
        for (int d : {200, 400, 600}) run<6, 1, 0, 1>("uncached", unc, err, log, st, 3072, 256, work, d);
        run<6, 1, 0, 2>("uncached", unc, err, log, st, 3072, 256, work, 0);
        run<1, 6, 0, 2>("uncached", unc, err, log, st, 3072, 256, work, 0);
        run<6, 1, 2, 2>("uncached", unc, err, log, st, 3072, 256, work, 0);
    }
    return 0;
}
