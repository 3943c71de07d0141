// ub_gemm3.hip -- stand-alone development bench of the 256 x 256 x 64 bf16 tile GEMM (kf_gemm3.hip): y[n, M] = x[n, K] . W[M, K]^T, both operands
// K-contiguous, staged global -> LDS with global_load_lds (16 B per lane, swizzled source address, lane-linear image), two LDS buffers, 8 waves as
// 2 (M) x 4 (tokens), mfma_f32_16x16x32_bf16.  Checks against a host fp64 product on a sample, prints TFLOP/s on random operands.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ub_gemm3 scratch/ub_gemm3.hip && scratch/ub_gemm3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2; /* 32 KB per A tile */

struct Args {
    const uint16_t* w;  // [M][K]
    const uint16_t* x;  // [n][K]
    uint16_t* y;        // [n][M]
    int M, n, K;
};

__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// one 32 KB operand tile (256 rows x 128 B): 32 wave instructions of 1 KB (8 rows); wave `wid` issues instructions wid*4 .. wid*4+3.
// LDS image: row r at r*128, 16-byte chunk c of the row stored at position c ^ ((r >> 1) & 7)  (conflict-free ds_read_b128 of 16 rows x one k chunk)
template <int NI>
__device__ __forceinline__ void stage_tile(const uint16_t* __restrict__ src, int row0, int nrows, int K, int k0, unsigned char* lds_tile, int wid, int lane) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int j = wid * NI + i;
        const int r = j * 8 + (lane >> 3), p = lane & 7, c = p ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;
        const uint16_t* g = src + (size_t)gr * K + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(lds_tile + j * 1024), 16, 0, 0);
    }
}

template <int BN, int NST>
__global__ void __launch_bounds__(512) gemm3_kernel(const Args a) {
    constexpr int NT = BN / 64, TB_BYTES = BN * BK * 2, STAGE = TILE_BYTES + TB_BYTES, NLD = 4 + BN / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    // XCD-aware block order: consecutive blocks of one XCD share the W row-tile
    const int nbx = (a.M + BM - 1) / BM, nby = (a.n + BN - 1) / BN, nwg = nbx * nby;
    int orig = blockIdx.x;
    const int q = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    const int bx = wg % nbx, by = wg / nbx;
    const int m0 = bx * BM, t0 = by * BN;
    const int nkt = a.K / BK;
    auto bufA = [&](int b) { return smem + (size_t)b * STAGE; };
    auto bufB = [&](int b) { return smem + (size_t)b * STAGE + TILE_BYTES; };

    f32x4 acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: NST - 1 k-tiles in flight
#pragma unroll
    for (int p = 0; p < NST - 1; p++)
        if (p < nkt) {
            stage_tile<4>(a.w, m0, a.M, a.K, p * BK, bufA(p), wid, lane);
            stage_tile<BN / 64>(a.x, t0, a.n, a.K, p * BK, bufB(p), wid, lane);
        }
    const int r16 = lane & 15, q4 = lane >> 4;
    int cur = 0, nxt = NST - 1;
    for (int kt = 0; kt < nkt; kt++) {
        if (kt + NST - 1 < nkt) {
            stage_tile<4>(a.w, m0, a.M, a.K, (kt + NST - 1) * BK, bufA(nxt), wid, lane);
            stage_tile<BN / 64>(a.x, t0, a.n, a.K, (kt + NST - 1) * BK, bufB(nxt), wid, lane);
            if (NST == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
        } else if (kt + NST - 2 < nkt && NST == 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 af[8], bfr[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int row = wn * (BN / 4) + nt * 16 + r16, c = kk * 4 + q4;
                bfr[nt] = *reinterpret_cast<const bf16x8*>(bufB(cur) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < 8; mt++) {
                const int row = wm * 128 + mt * 16 + r16, c = kk * 4 + q4;
                af[mt] = *reinterpret_cast<const bf16x8*>(bufA(cur) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < 8; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); /* everyone is done reading buf[cur]: a later iteration's stage may overwrite it */
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    // epilogue: lane holds rows m = 4 q4 + j of the 16 x 16 tile for token column r16
#pragma unroll
    for (int mt = 0; mt < 8; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int tok = t0 + wn * (BN / 4) + nt * 16 + r16, m = m0 + wm * 128 + mt * 16 + 4 * q4;
            if (tok < a.n && m + 3 < a.M) {
                const f32x4 v = acc[mt][nt];
                *reinterpret_cast<u32x2*>(a.y + (size_t)tok * a.M + m) = u32x2{(uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16), (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16)};
            }
        }
}

static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t hf2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <int BN, int NST>
static double run(const Args& a, int reps) {
    const size_t smem = (size_t)NST * (TILE_BYTES + BN * BK * 2);
    CK(hipFuncSetAttribute((const void*)gemm3_kernel<BN, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int grid = ((a.M + BM - 1) / BM) * ((a.n + BN - 1) / BN);
    hipLaunchKernelGGL((gemm3_kernel<BN, NST>), dim3(grid), dim3(512), smem, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((gemm3_kernel<BN, NST>), dim3(grid), dim3(512), smem, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}
int main() {
    const int shapes[][3] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {4800, 8192, 1600}, {6400, 8192, 1600}, {1600, 8192, 6400}, {1600, 8192, 1600}, {50304, 8192, 1600}, {1600, 4800, 8192}, {6144, 2048, 1024}, {3072, 8192, 1024}};
    for (auto& s : shapes) {
        const int M = s[0], n = s[1], K = s[2];
        std::vector<uint16_t> hw((size_t)M * K), hx((size_t)n * K);
        srand(1);
        for (auto& v : hw) v = hf2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
        for (auto& v : hx) v = hf2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
        uint16_t *dw, *dx, *dy;
        CK(hipMalloc(&dw, hw.size() * 2));
        CK(hipMalloc(&dx, hx.size() * 2));
        CK(hipMalloc(&dy, (size_t)n * M * 2));
        CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        Args a{dw, dx, dy, M, n, K};
        const double us[4] = {run<256, 2>(a, 20), run<128, 2>(a, 20), run<128, 3>(a, 20), 0};
        // correctness of the last variant run
        std::vector<uint16_t> hy((size_t)n * M);
        CK(hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0, scale = 0;
        for (int s2 = 0; s2 < 300; s2++) {
            const int t = (int)((unsigned)rand() % n), m = (int)((unsigned)rand() % (M / 4 * 4));
            double ref = 0;
            for (int k = 0; k < K; k++) ref += (double)bf2f(hw[(size_t)m * K + k]) * bf2f(hx[(size_t)t * K + k]);
            maxerr = fmax(maxerr, fabs(bf2f(hy[(size_t)t * M + m]) - ref)), scale = fmax(scale, fabs(ref));
        }
        const double fl = 2.0 * M * n * K / 1e6;
        printf("M %6d n %5d K %5d:  256x256/2: %7.1f us %6.0f TF | 256x128/2: %7.1f us %6.0f TF | 256x128/3: %7.1f us %6.0f TF   (%s)\n", M, n, K, us[0], fl / us[0], us[1],
               fl / us[1], us[2], fl / us[2], maxerr <= scale * 0.01 ? "ok" : "WRONG");
        CK(hipFree(dw));
        CK(hipFree(dx));
        CK(hipFree(dy));
    }
    return 0;
}
