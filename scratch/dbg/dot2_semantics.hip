// What does v_dot2c_f32_bf16 compute, bit for bit?  D = D + a.lo*b.lo + a.hi*b.hi -- but with which roundings?  Candidates, evaluated in fp64
// (bf16 x bf16 products are exact in fp32, their sum with the accumulator is exact in fp64 for operands within ~2^37 of each other):
//   0 fused       : one rounding of the exact three-term sum
//   1 lo then hi  : fmaf(a.hi, b.hi, fmaf(a.lo, b.lo, acc))
//   2 hi then lo  : fmaf(a.lo, b.lo, fmaf(a.hi, b.hi, acc))
//   3 products first: acc + round(p_lo + p_hi)   (two roundings)
//   4 round-toward-zero fused
//   hipcc --offload-arch=gfx950 -O2 -o scratch/dbg/dot2_semantics scratch/dbg/dot2_semantics.hip && scratch/dbg/dot2_semantics
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, const float* c, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a[i]), __builtin_bit_cast(bf16x2_t, b[i]), c[i], false);
}
static float bf(uint32_t h) { uint32_t u = h << 16; float f; memcpy(&f, &u, 4); return f; }
static float rtz(double x) { float f = (float)x; if ((double)f != x && fabs((double)f) > fabs(x)) f = nextafterf(f, 0.0f); return f; }
int main() {
    const int n = 1 << 22;
    std::vector<uint32_t> a(n), b(n);
    std::vector<float> c(n), out(n);
    srand(12345);
    auto rnd_bf = [&](int spread) -> uint32_t { /* sign, exponent 127 +- spread, 7 mantissa bits */
        return ((rand() & 1) << 15) | ((uint32_t)(127 - spread + rand() % (2 * spread + 1)) << 7) | (rand() & 127);
    };
    for (int i = 0; i < n; i++) {
        const int sp = 1 + (i & 7);
        a[i] = rnd_bf(sp) | (rnd_bf(sp) << 16), b[i] = rnd_bf(sp) | (rnd_bf(sp) << 16);
        uint32_t cu = ((rand() & 1u) << 31) | ((uint32_t)(127 - sp + rand() % (2 * sp + 1)) << 23) | (((uint32_t)rand() << 8 ^ rand()) & 0x7fffff);
        if ((i & 15) == 0) cu = 0; /* acc = 0: the first link of a chain */
        memcpy(&c[i], &cu, 4);
    }
    uint32_t *da, *db; float *dc, *dout;
    hipMalloc(&da, n * 4), hipMalloc(&db, n * 4), hipMalloc(&dc, n * 4), hipMalloc(&dout, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice), hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice), hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
    hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
    long hit[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const float al = bf(a[i] & 0xffff), ah = bf(a[i] >> 16), bl = bf(b[i] & 0xffff), bh = bf(b[i] >> 16);
        const double pl = (double)al * bl, ph = (double)ah * bh;
        const float cand[5] = {(float)((double)c[i] + pl + ph), fmaf(ah, bh, fmaf(al, bl, c[i])), fmaf(al, bl, fmaf(ah, bh, c[i])), c[i] + (float)(pl + ph), rtz((double)c[i] + pl + ph)};
        for (int j = 0; j < 5; j++) hit[j] += memcmp(&cand[j], &out[i], 4) == 0;
    }
    // alignment models: every addend (two exact products, the accumulator) is a sign-magnitude integer at its own exponent; all are aligned to the
    // largest exponent keeping G bits below the largest addend's unit in the last place (24-bit significand), the shifted-out bits dropped (trunc) or
    // OR-ed into a sticky bit; the aligned integers are added exactly and the sum rounded to nearest-even.
    {
        auto decomp = [](double x, long long& m, int& e) { /* x = m * 2^e with m a 53-bit integer (0 for x = 0) */
            if (x == 0) { m = 0, e = -2000; return; }
            int ex;
            const double fr = frexp(x, &ex);
            m = (long long)ldexp(fr, 53), e = ex - 53;
        };
        for (int sticky = 0; sticky < 2; sticky++)
            for (int G = 0; G <= 32; G++) {
                long ok = 0;
                for (int i = 0; i < n; i++) {
                    const float al = bf(a[i] & 0xffff), ah = bf(a[i] >> 16), bl = bf(b[i] & 0xffff), bh = bf(b[i] >> 16);
                    const double t[3] = {(double)al * bl, (double)ah * bh, (double)c[i]};
                    long long m[3];
                    int e[3], emax = -4000;
                    for (int j = 0; j < 3; j++) {
                        decomp(t[j], m[j], e[j]);
                        if (m[j] != 0) { int top = e[j] + 52; if (top > emax) emax = top; } /* exponent of the leading bit */
                    }
                    if (emax == -4000) { ok += out[i] == 0.0f; continue; }
                    const int lsb = emax - 23 - G; /* weight of the last kept bit */
                    __int128 sum = 0;
                    bool st = false;
                    for (int j = 0; j < 3; j++) {
                        if (m[j] == 0) continue;
                        const long long mag = m[j] < 0 ? -m[j] : m[j];
                        const int sh = lsb - e[j];
                        __int128 v;
                        if (sh <= 0) v = (__int128)mag << (-sh);
                        else if (sh >= 63) { v = 0; st = st || mag != 0; }
                        else { v = mag >> sh; st = st || (mag & ((1ll << sh) - 1)) != 0; }
                        sum += m[j] < 0 ? -v : v;
                    }
                    // value = sum * 2^lsb (+ sticky epsilon in the direction of the dropped bits: ignored unless sticky mode, where it only breaks ties)
                    double val = ldexp((double)sum, lsb); /* sum has < 2^64 magnitude but may exceed 53 bits: do the rounding by hand */
                    __int128 mag = sum < 0 ? -sum : sum;
                    float r;
                    if (mag == 0) r = 0.0f;
                    else {
                        int hb = 127;
                        while (!((mag >> hb) & 1)) hb--;
                        if (hb <= 23) r = (float)ldexp((double)(long long)mag, lsb);
                        else {
                            const int drop = hb - 23;
                            __int128 keep = mag >> drop;
                            const __int128 rem = mag & (((__int128)1 << drop) - 1), half = (__int128)1 << (drop - 1);
                            bool up = rem > half || (rem == half && ((keep & 1) || (sticky && st)));
                            if (sticky && st && rem == half) up = true;
                            if (up) keep += 1;
                            r = (float)ldexp((double)(long long)keep, lsb + drop);
                        }
                        if (sum < 0) r = -r;
                    }
                    (void)val;
                    ok += memcmp(&r, &out[i], 4) == 0;
                }
                if (ok > (long)(0.97 * n) || G % 8 == 0) printf("align G=%2d %s: matches %ld of %d (%.4f %%)\n", G, sticky ? "sticky" : "trunc ", ok, n, 100.0 * ok / n);
            }
    }
    // two-stage models: t = the two products added and kept to P significant bits (rne or trunc), then r = float(acc + t) (one IEEE rounding);
    // and the mirror: u = (acc + p_lo) kept to P bits, r = float(u + p_hi)
    {
        auto keepP = [](double x, int P, bool trunc) -> double {
            if (x == 0) return 0;
            int ex;
            const double fr = frexp(x, &ex); /* |fr| in [0.5, 1) */
            const double sc = ldexp(fr, P);
            const double q = trunc ? (sc < 0 ? ceil(sc) : floor(sc)) : nearbyint(sc);
            return ldexp(q, ex - P);
        };
        for (int form = 0; form < 3; form++)
            for (int tr = 0; tr < 2; tr++)
                for (int P = 24; P <= 50; P++) {
                    long ok = 0;
                    for (int i = 0; i < n; i += 4) {
                        const float al = bf(a[i] & 0xffff), ah = bf(a[i] >> 16), bl = bf(b[i] & 0xffff), bh = bf(b[i] >> 16);
                        const double pl = (double)al * bl, ph = (double)ah * bh;
                        float r;
                        if (form == 0) r = (float)((double)c[i] + keepP(pl + ph, P, tr));
                        else if (form == 1) r = (float)(keepP((double)c[i] + pl, P, tr) + ph);
                        else r = (float)(keepP((double)c[i] + ph, P, tr) + pl);
                        ok += memcmp(&r, &out[i], 4) == 0;
                    }
                    if (ok > (long)(0.95 * (n / 4)) || P == 24 || P == 32) printf("two-stage form %d %s P=%2d: %.4f %%\n", form, tr ? "trunc" : "rne  ", P, 100.0 * ok / (n / 4));
                }
    }
    const char* name[5] = {"fused (one rounding)", "fma lo then hi", "fma hi then lo", "acc + round(p_lo + p_hi)", "fused, toward zero"};
    for (int j = 0; j < 5; j++) printf("%-28s matches %ld of %d (%.4f %%)\n", name[j], hit[j], n, 100.0 * hit[j] / n);
    return 0;
}
