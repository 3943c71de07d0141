// Does the memory-side cache (Infinity Cache, 256 MB) keep a buffer that is read again?  Repeated reads of the same N MB by 2048 workgroups, plain and nontemporal loads:
// bandwidth above the HBM peak for N below the cache size = hits.  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_mall scratch/ub_mall.hip && /tmp/ub_mall
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void __launch_bounds__(256) rd_chunk(const u32x4* __restrict__ p, size_t n, unsigned* out) {
    const size_t nwaves = (size_t)gridDim.x * 4, per = n / nwaves / 64;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const u32x4* base = p + wave * per * 64 + lane;
    unsigned acc = 0;
    for (size_t k = 0; k + 8 <= per; k += 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = NT ? __builtin_nontemporal_load(base + (k + u) * 64) : base[(k + u) * 64];
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// one dword per 128-byte line (a prefetch: 1/8 of the bytes requested, every line touched)
__global__ void __launch_bounds__(256) touch(const unsigned* __restrict__ p, size_t nlines, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < nlines; i += stride) acc ^= p[i * 32];
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t maxb = (size_t)1024 << 20;
    u32x4* d; unsigned* o;
    hipMalloc(&d, maxb); hipMalloc(&o, 4); hipMemset(d, 1, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {32, 64, 128, 192, 256, 311, 512, 1024}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        for (int nt = 0; nt < 2; nt++) {
            for (int i = 0; i < 3; i++) { if (nt) hipLaunchKernelGGL(rd_chunk<1>, dim3(2048), dim3(256), 0, 0, d, n, o); else hipLaunchKernelGGL(rd_chunk<0>, dim3(2048), dim3(256), 0, 0, d, n, o); }
            hipEventRecord(e0);
            for (int i = 0; i < 20; i++) { if (nt) hipLaunchKernelGGL(rd_chunk<1>, dim3(2048), dim3(256), 0, 0, d, n, o); else hipLaunchKernelGGL(rd_chunk<0>, dim3(2048), dim3(256), 0, 0, d, n, o); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%5zu MB re-read, %s loads: %7.1f us per pass  %6.0f GB/s\n", mb, nt ? "nontemporal" : "plain      ", ms * 50, bytes / (ms / 20 * 1e-3) / 1e9);
        }
    }
    // a prefetch pass (one dword per line) of 160 MB, then ONE full nontemporal read of it behind 700 MB of other traffic or directly
    for (int between = 0; between < 2; between++) {
        const size_t bytes = (size_t)160 << 20, n = bytes / 16;
        float tot = 0;
        for (int rep = 0; rep < 10; rep++) {
            hipLaunchKernelGGL(rd_chunk<1>, dim3(2048), dim3(256), 0, 0, d + (((size_t)300 << 20) / 16), ((size_t)700 << 20) / 16, o); /* flush: 700 MB of other lines */
            hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, 0, (const unsigned*)d, bytes / 128, o);
            if (between) hipLaunchKernelGGL(rd_chunk<1>, dim3(2048), dim3(256), 0, 0, d + (((size_t)300 << 20) / 16), ((size_t)80 << 20) / 16, o); /* 80 MB of nontemporal traffic in between */
            hipEventRecord(e0);
            hipLaunchKernelGGL(rd_chunk<1>, dim3(2048), dim3(256), 0, 0, d, n, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
        }
        printf("160 MB touched line by line, %s, then read once (nontemporal): %7.1f us  %6.0f GB/s\n", between ? "80 MB of other nontemporal reads in between" : "read directly", tot * 100, bytes / (tot / 10 * 1e-3) / 1e9);
    }
    return 0;
}
