#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
__global__ void k(float old, float g0, float m0, float v0, float* out) {
    const float lr = 3e-4f, beta1 = 0.9f, beta2 = 0.95f, b1c = 1 - 0.9f * 0.9f * 0.9f, b2c = 1 - 0.95f * 0.95f * 0.95f, eps = 1e-8f, wd = 0.1f, gs = 0.25f;
    float g = gs * g0;
    float m = fmaf(beta1, m0, fmaf(-beta1, g, g));
    float g2 = g * g;
    float v = fmaf(beta2, v0, fmaf(-beta2, g2, g2));
    float mh = __fdiv_rn(m, out[20]), vh = __fdiv_rn(v, out[21]);
    float sq = __fsqrt_rn(vh);
    float den = sq + eps;
    float step = __fdiv_rn(mh, den);
    float a = lr * wd * old, b = lr * step;
    float p = old - lr * wd * old - lr * step;
    out[0] = g, out[1] = m, out[2] = v, out[3] = mh, out[4] = vh, out[5] = sq, out[6] = den, out[7] = step, out[8] = a, out[9] = b, out[10] = p;
    out[11] = mh / den; out[12] = sqrtf(vh);
}
int main() {
    const uint16_t cases[2][4] = {{0x39e0, 0x3c72, 0x3bcd, 0x3815}, {0x3a08, 0x399e, 0x3c56, 0x38c8}};
    float* d; hipMalloc(&d, 128);
    for (int c = 0; c < 2; c++) {
        float h[32] = {0};
        float b1c, b2c;
        { volatile float x = 0.9f; float y = 1 - (float)pow(0.9, 3); (void)x; b1c = y; }
        { float y = 1 - (float)pow(0.95, 3); b2c = y; }
        h[20] = b1c, h[21] = b2c;
        hipMemcpy(d, h, 128, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, bf(cases[c][0]), bf(cases[c][1]), bf(cases[c][2]), bf(cases[c][3]), d);
        hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
        // host
        const float lr = 3e-4f, beta1 = 0.9f, beta2 = 0.95f, eps = 1e-8f, wd = 0.1f, gs = 0.25f;
        float old = bf(cases[c][0]), g = gs * bf(cases[c][1]);
        float m = fmaf(beta1, bf(cases[c][2]), fmaf(-beta1, g, g));
        float g2 = g * g;
        float v = fmaf(beta2, bf(cases[c][3]), fmaf(-beta2, g2, g2));
        float mh = m / b1c, vh = v / b2c, sq = sqrtf(vh), den = sq + eps, step = mh / den, a = lr * wd * old, b = lr * step, p = old - lr * wd * old - lr * step;
        float hh[11] = {g, m, v, mh, vh, sq, den, step, a, b, p};
        const char* nm[11] = {"g", "m", "v", "mh", "vh", "sq", "den", "step", "a", "b", "p"};
        { uint32_t x, y; memcpy(&x, &h[12], 4); float sh = sqrtf(hh[4]); memcpy(&y, &sh, 4); printf("case %d sqrtf() dev %08x host %08x\n", c, x, y); }
        for (int i = 0; i < 11; i++) { uint32_t x, y; memcpy(&x, &h[i], 4); memcpy(&y, &hh[i], 4); printf("case %d %-5s dev %08x host %08x %s\n", c, nm[i], x, y, x == y ? "" : "<<<<"); }
    }
    return 0;
}
