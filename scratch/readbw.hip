#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template<int NT, int U>
__global__ void __launch_bounds__(256) rd(const u32x4* __restrict__ p, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// contiguous per-wave chunks (like the gemv: each wave streams its own region)
template<int NT, int U>
__global__ void __launch_bounds__(256) rd_chunk(const u32x4* __restrict__ p, size_t n, unsigned* out) {
    const size_t nwaves = (size_t)gridDim.x * 4, per = n / nwaves / 64;  // u32x4 per lane
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const u32x4* base = p + wave * per * 64 + lane;
    unsigned acc = 0;
    for (size_t k = 0; k + U <= per; k += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(base + (k + u) * 64) : base[(k + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    size_t bytes = 311164928; size_t n = bytes / 16;
    u32x4* d; unsigned* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int blocks) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, n, o);
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, n, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s blocks %5d: %7.1f us  %6.0f GB/s\n", name, blocks, ms * 50, bytes / (ms / 20 * 1e-3) / 1e9);
    };
    for (int b : {1024, 2048, 4096, 8192}) {
        run("strided plain U4", rd<0, 4>, b);
        run("strided nt U4", rd<1, 4>, b);
        run("strided nt U8", rd<1, 8>, b);
        run("chunk nt U4", rd_chunk<1, 4>, b);
        run("chunk plain U4", rd_chunk<0, 4>, b);
        run("chunk nt U8", rd_chunk<1, 8>, b);
    }
    return 0;
}
