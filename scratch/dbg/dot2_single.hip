// Is v_dot2c_f32_bf16 with ONE of its two products zeroed an exact IEEE fused multiply-add?  D = acc + a.lo * b.lo + a.hi * 0  (and the hi-only twin).
// scratch/dbg/dot2_semantics.hip found no bit-exact host model for the two-product form; with one product identically +0 the candidate models (fused three-term sum, fma
// chain, products-first) coincide with fmaf(a, b, acc) -- unless the hardware truncates aligned addends.  If this holds on every operand, the canonical order's even / odd
// chains can run as two v_dot2c per weight pair on pre-masked activation pairs, with no bf16 -> fp32 conversions at all.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/dot2_single scratch/dbg/dot2_single.hip && /tmp/dot2_single
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, const float* c, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a[i]), __builtin_bit_cast(bf16x2_t, b[i]), c[i], false);
}
static float bf(uint32_t h) { uint32_t u = h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 23;
    std::vector<uint32_t> a(n), b(n);
    std::vector<float> c(n), out(n);
    srand(4242);
    auto rnd_bf = [&](int center, int spread) -> uint32_t { return ((rand() & 1) << 15) | ((uint32_t)(center - spread + rand() % (2 * spread + 1)) << 7) | (rand() & 127); };
    for (int i = 0; i < n; i++) {
        const int mode = i & 7, sp = 1 + ((i >> 3) & 31);   /* exponent spreads up to +-32 around 127 for the operands; the accumulator up to +-48 away */
        const uint32_t alo = rnd_bf(127, sp), ahi = rnd_bf(127, sp), bv = rnd_bf(127, sp);
        a[i] = alo | (ahi << 16);
        b[i] = (mode & 1) ? (bv << 16) : bv;               /* hi-only or lo-only activation */
        uint32_t cu = ((rand() & 1u) << 31) | ((uint32_t)(127 - 48 + rand() % 97) << 23) | ((((uint32_t)rand() << 8) ^ (uint32_t)rand()) & 0x7fffff);
        if (mode == 2) cu = 0;                              /* first link of a chain */
        if (mode == 4) cu &= 0xff800000u;                   /* power-of-two accumulator: exact ties become likely */
        if (mode == 6) {                                    /* accumulator = -(product) + tiny: cancellation */
            const float p = bf((mode & 1) ? ahi : alo) * bf(bv);
            float t = -p * (1.0f + (float)(rand() % 5 - 2) * 1.1920929e-7f);
            memcpy(&cu, &t, 4);
        }
        memcpy(&c[i], &cu, 4);
    }
    uint32_t *da, *db; float *dc, *dout;
    hipMalloc(&da, n * 4), hipMalloc(&db, n * 4), hipMalloc(&dc, n * 4), hipMalloc(&dout, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice), hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
    hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, shown = 0;
    long bad_mode[8] = {0};
    for (int i = 0; i < n; i++) {
        const int mode = i & 7;
        const float av = (mode & 1) ? bf(a[i] >> 16) : bf(a[i] & 0xffff), bv = (mode & 1) ? bf(b[i] >> 16) : bf(b[i] & 0xffff);
        const float ref = fmaf(av, bv, c[i]);
        if (memcmp(&ref, &out[i], 4) != 0) {
            bad++, bad_mode[mode]++;
            if (shown++ < 12) printf("  mismatch mode %d: a %a b %a c %a -> gpu %a fmaf %a\n", mode, av, bv, c[i], out[i], ref);
        }
    }
    printf("single-product v_dot2c_f32_bf16 vs fmaf: %ld mismatches of %d", bad, n);
    for (int m = 0; m < 8; m++) printf("  m%d:%ld", m, bad_mode[m]);
    printf("\n");
    return 0;
}
