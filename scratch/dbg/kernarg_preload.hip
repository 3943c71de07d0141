// Does the kernarg SGPR preload (-mllvm -amdgpu-kernarg-preload-count=N) shorten the time to the first kernarg-dependent instruction?
// Build twice (with / without the flag) and compare the two stamps.  hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=8] ...
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(long long* out, const int* src, int mul) {
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    const int v = src[threadIdx.x] * mul;   // needs src (kernarg) -> load -> use
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("" ::"v"(v));
    const int u = __builtin_amdgcn_readfirstlane(mul);  // needs only the kernarg
    asm volatile("s_nop 0" ::"s"(u));
    const long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0, out[1] = t2 - t0, out[2] = w0, out[3] = v;
}
int main() {
    long long* out; int* src;
    hipMalloc(&out, 64); hipMalloc(&src, 4096); hipMemset(src, 0, 4096);
    hipGraph_t g; hipGraphExec_t ge; hipStream_t st; hipStreamCreate(&st);
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, st, out, src, i + 1);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 5; r++) {
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
        printf("cycles to first kernarg-dependent load result: %lld   (stamp2 %lld)\n", h[0], h[1]);
    }
    return 0;
}
