#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
__global__ void k(const float* x, const float* y, float* s32, float* s64, float* dv, float* rc, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s32[i] = sqrtf(x[i]);
    s64[i] = (float)sqrt((double)x[i]);
    dv[i] = x[i] / y[i];
    rc[i] = 1.0f / y[i];
}
int main() {
    const int n = 1 << 24;
    std::vector<float> x(n), y(n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> ue(-30.f, 30.f), um(1.f, 2.f);
    for (int i = 0; i < n; i++) x[i] = ldexpf(um(g), (int)ue(g)), y[i] = ldexpf(um(g), (int)ue(g));
    float *dx, *dy, *a, *b, *c, *d;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&d, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, a, b, c, d, n);
    std::vector<float> ha(n), hb(n), hc(n), hd(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), d, n * 4, hipMemcpyDeviceToHost);
    long e32 = 0, e64 = 0, ediv = 0, erc = 0;
    for (int i = 0; i < n; i++) {
        float s = sqrtf(x[i]), q = x[i] / y[i], r = 1.0f / y[i];
        e32 += memcmp(&s, &ha[i], 4) != 0; e64 += memcmp(&s, &hb[i], 4) != 0; ediv += memcmp(&q, &hc[i], 4) != 0; erc += memcmp(&r, &hd[i], 4) != 0;
    }
    printf("n=%d mismatches vs host: sqrtf %ld, (float)sqrt(double) %ld, x/y %ld, 1/y %ld\n", n, e32, e64, ediv, erc);
    return 0;
}
