// ub_overlap.hip -- what does an early-launched, resident consumer buy on a chain of short weight-streaming kernels?
//
// Stand-in for the 0.6B decode step: 28 x 5 dependent "phases" (M x K 4-bit-sized weight slabs, every output needs the whole
// input vector).  Each phase kernel requests its weights FIRST (16 B per lane, nt), then polls its input vector, which the
// producer's epilogue wrote as tagged granules {value16, tag16} with sc1 stores, then multiplies (integer arithmetic: exact and
// order-free, so every variant must produce the same digest) and writes its own tagged granules.
//
// One hipGraph of 140 kernel nodes, node i depending on node i-C: C = 1 is today's dependent chain (polls match at once),
// C = 2..4 lets the next C-1 kernels become resident and fetch weights while the producer still runs.
//
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ub_overlap scratch/ub_overlap.hip && scratch/ub_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct PhaseArgs {
    const u32x4* w;      // this phase's weight slab
    const uint32_t* xin; // K granules
    uint32_t* xout;      // M granules
    const int* d_epoch;
    int* d_err;
    unsigned long long* d_log;
    int M, K, rows_per_wg, idx, nphase, work, last;
};

__device__ __forceinline__ u32x4 ld_sc1_16(const uint32_t* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1_4(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }
// 256 threads; WG handles rows_per_wg rows; block b of the WG slab: row b / nblk, column block b % nblk (32 weights = 16 B)
template <int POLL>
__global__ void __launch_bounds__(256) phase_kernel(const PhaseArgs a) {
    __shared__ uint32_t xs[4096 / 2]; // K values as packed u16 pairs
    __shared__ uint32_t rowsum[16];
    const int tid = threadIdx.x;
    const bool logme = blockIdx.x == gridDim.x / 2 && tid == 0;
    if (logme) a.d_log[a.idx * 4] = rt();
    const int nblk = a.K >> 5, total = a.rows_per_wg * nblk;
    const u32x4* slab = a.w + (size_t)blockIdx.x * total;
    // weights first
    u32x4 w0 = u32x4{0, 0, 0, 0}, w1 = w0;
    const bool h0 = tid < total, h1 = tid + 256 < total;
    w0 = __builtin_nontemporal_load(slab + (h0 ? tid : 0));
    w1 = __builtin_nontemporal_load(slab + (h1 ? tid + 256 : 0));
    const int epoch = *a.d_epoch;
    const int max_spins = (*a.d_err != 0) ? 1 : (1 << 12); /* once a poll has timed out, later kernels do not wait again */
    const uint32_t tag_in = (uint32_t)(epoch * 256 + a.idx) & 0xffffu, tag_out = (uint32_t)(epoch * 256 + a.idx + 1) & 0xffffu;
    if (tid < 16) rowsum[tid] = 0;
    // poll the input vector: thread t takes granules 4t.. (+1024 per round)
    const int rounds = a.K >> 10;
    for (int r = 0; r < rounds; r++) {
        const uint32_t* src = a.xin + (size_t)(r * 256 + tid) * 4;
        u32x4 g;
        int spins = 0;
        for (;;) {
            if (POLL)
                g = ld_sc1_16(src);
            else
                g = *reinterpret_cast<const u32x4*>(src);
            const bool ok = ((g.x >> 16) == tag_in) & ((g.y >> 16) == tag_in) & ((g.z >> 16) == tag_in) & ((g.w >> 16) == tag_in);
            if (__all(ok)) break;
            if (++spins > max_spins) {
                if ((tid & 63) == 0) atomicAdd(a.d_err, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        xs[(r * 256 + tid) * 2] = (g.x & 0xffffu) | (g.y << 16);
        xs[(r * 256 + tid) * 2 + 1] = (g.z & 0xffffu) | (g.w << 16);
    }
    if (logme) a.d_log[a.idx * 4 + 1] = rt();
    __syncthreads();
    // multiply: weights nibble i of dword d against x[col*32 + d*8 + i]
    auto dot_block = [&](u32x4 w, int b) -> uint32_t {
        const int col = b % nblk;
        const uint32_t* xc = xs + col * 16;
        const uint32_t dw[4] = {w.x, w.y, w.z, w.w};
        uint32_t acc = 0;
        for (int rep = 0; rep < a.work; rep++) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t xp = xc[d * 4 + i];
                    acc += ((dw[d] >> (8 * i)) & 15u) * (xp & 0xffffu) + ((dw[d] >> (8 * i + 4)) & 15u) * (xp >> 16);
                }
            }
            acc += rep;
        }
        return acc;
    };
    if (h0) atomicAdd(&rowsum[tid / nblk], dot_block(w0, tid));
    if (h1) atomicAdd(&rowsum[(tid + 256) / nblk], dot_block(w1, tid + 256));
    __syncthreads();
    if (tid < a.rows_per_wg) {
        const int row = blockIdx.x * a.rows_per_wg + tid;
        uint32_t v = (rowsum[tid] * 2654435761u) >> 16;
        st_sc1_4(a.xout + row, (tag_out << 16) | (v & 0xffffu));
    }
    if (logme) a.d_log[a.idx * 4 + 2] = rt();
}

__global__ void seed_kernel(uint32_t* x, int n, const int* d_epoch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tag = (uint32_t)(*d_epoch * 256) & 0xffffu;
    if (i < n) st_sc1_4(x + i, (tag << 16) | ((i * 40503u + 7u) & 0xffffu));
}
__global__ void bump_kernel(int* d_epoch) { *d_epoch += 1; }
__global__ void fill_kernel(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
}


// ------------------------------------------------------------------------------------------------ persistent form
// ONE launch, 256 workgroups x 1024 threads (one per CU), looping over all phases.  Wave 15 only polls: it sweeps the phase's input
// granules (16 B per lane per load, sc1), checks the tags, stages the vector into LDS; waves 0..14 hold the phase's weight blocks
// (prefetched one phase ahead) and multiply.  Outputs leave as tagged granules.
struct PPhase {
    const u32x4* w;
    const uint32_t* xin;
    uint32_t* xout;
    int M, K, rows_per_wg, pad;
};
struct PArgs {
    const PPhase* ph;
    const int* d_epoch;
    int* d_err;
    unsigned long long* d_log; // [NP][4] stamps of workgroup `log_wg`
    int nphase, work, sleep, log_wg;
};


__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// one sweep of NLD x 1 KiB granule pieces (16 B per lane each), every load in flight before the first tag is looked at
template <int NLD>
__device__ __forceinline__ bool sweep(const uint32_t* xin, int lane, uint32_t tag, uint32_t* xs, int sleep, int* d_err) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(xin, NLD * 1024);
    u32x4 g[NLD];
    for (int spins = 0;; spins++) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < NLD; r++) g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (r * 64 + lane) * 16, 0, 16 /* sc1 */));
#pragma unroll
        for (int r = 0; r < NLD; r++) ok &= ((g[r].x >> 16) == tag) & ((g[r].y >> 16) == tag) & ((g[r].z >> 16) == tag) & ((g[r].w >> 16) == tag);
        if (__all(ok)) break;
        if (spins > (1 << 14)) {
            if (lane == 0) atomicAdd(d_err, 1);
            break;
        }
        if (sleep == 1) __builtin_amdgcn_s_sleep(1); else if (sleep == 2) __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int r = 0; r < NLD; r++) {
        xs[(r * 64 + lane) * 2] = (g[r].x & 0xffffu) | (g[r].y << 16);
        xs[(r * 64 + lane) * 2 + 1] = (g[r].z & 0xffffu) | (g[r].w << 16);
    }
    return true;
}

__global__ void __launch_bounds__(1024) persist_kernel(const PArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t xs[4096 / 2];
    __shared__ uint32_t rowsum[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool poller = wave == 15;
    const int epoch = *a.d_epoch;
    const bool logme = (int)blockIdx.x == a.log_wg;
    if (tid < 32) rowsum[tid] = 0;
    // prefetch phase 0 weights
    PPhase ph = a.ph[0];
    int nblk = ph.K >> 5, total = ph.rows_per_wg * nblk;
    u32x4 wcur = u32x4{0, 0, 0, 0};
    if (!poller) wcur = __builtin_nontemporal_load(ph.w + (size_t)blockIdx.x * total + (tid < total ? tid : 0));
    PPhase nxd = a.ph[1];
    __syncthreads();
    for (int p = 0; p < a.nphase; p++) {
        const uint32_t tag_in = (uint32_t)(epoch * 256 + p) & 0xffffu, tag_out = (uint32_t)(epoch * 256 + p + 1) & 0xffffu;
        const PPhase nx = nxd;
        nxd = a.ph[p + 2 < a.nphase ? p + 2 : p];
        u32x4 wnext = wcur;
        if (poller) {
            if (logme && lane == 0) a.d_log[p * 4 + 0] = rt();
            const int nld = ph.K >> 8; /* 256 granules per wave-load */
            if (nld == 4) sweep<4>(ph.xin, lane, tag_in, xs, a.sleep, a.d_err);
            else if (nld == 8) sweep<8>(ph.xin, lane, tag_in, xs, a.sleep, a.d_err);
            else if (nld == 12) sweep<12>(ph.xin, lane, tag_in, xs, a.sleep, a.d_err);
            else sweep<16>(ph.xin, lane, tag_in, xs, a.sleep, a.d_err);
            if (logme && lane == 0) a.d_log[p * 4 + 1] = rt();
        } else {
            // next phase's weights go out while this phase waits for its input
            const int nb2 = nx.K >> 5, tot2 = nx.rows_per_wg * nb2;
            if (p + 1 < a.nphase) wnext = __builtin_nontemporal_load(nx.w + (size_t)blockIdx.x * tot2 + (tid < tot2 ? tid : 0));
        }
        __syncthreads(); /* A: x staged */
        if (!poller && tid < total) {
            const int col = tid % nblk;
            const uint32_t* xc = xs + col * 16;
            const uint32_t dw[4] = {wcur.x, wcur.y, wcur.z, wcur.w};
            uint32_t acc = 0;
            for (int rep = 0; rep < a.work; rep++) {
#pragma unroll
                for (int d = 0; d < 4; d++) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t xp = xc[d * 4 + i];
                        acc += ((dw[d] >> (8 * i)) & 15u) * (xp & 0xffffu) + ((dw[d] >> (8 * i + 4)) & 15u) * (xp >> 16);
                    }
                }
                acc += rep;
            }
            atomicAdd(&rowsum[tid / nblk], acc);
        }
        __syncthreads(); /* B: row sums complete */
        if (tid < ph.rows_per_wg) {
            const int row = blockIdx.x * ph.rows_per_wg + tid;
            const uint32_t v = (rowsum[tid] * 2654435761u) >> 16;
            rowsum[tid] = 0;
            st_sc1_4(ph.xout + row, (tag_out << 16) | (v & 0xffffu));
            if (logme && tid == 0) a.d_log[p * 4 + 2] = rt();
        }
        ph = nx, nblk = ph.K >> 5, total = ph.rows_per_wg * nblk;
        wcur = wnext;
    }
}

struct Shape {
    int M, K, rows;
};

int main(int argc, char** argv) {
    const int L = 28;
    // per layer: [norm+QKV] 4096x1024, [attention stand-in] 2048x4096 (1/4 of it polled... here all), [o_proj] 1024x2048, [gate/up] 6144x1024, [down] 1024x3072
    const Shape layer[5] = {{4096, 1024, 8}, {2048, 4096, 2}, {1024, 2048, 4}, {6144, 1024, 8}, {1024, 3072, 4}};
    const int NP = L * 5;
    int work = argc > 1 ? atoi(argv[1]) : 2;
    int reps = argc > 2 ? atoi(argv[2]) : 50;
    size_t wbytes = 0;
    std::vector<size_t> woff(NP);
    for (int p = 0; p < NP; p++) {
        woff[p] = wbytes;
        wbytes += (size_t)layer[p % 5].M * layer[p % 5].K / 2;
    }
    wbytes += (size_t)L * (2048 * 1024 / 2);
    printf("phases %d, weight bytes %.1f MB, work %d\n", NP, wbytes / 1e6, work);
    uint32_t* dw;
    CK(hipMalloc(&dw, wbytes));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, dw, wbytes / 4);
    // hand-off buffers: one per edge type (6: input of phase j of a layer; index 5 == 0 of the next layer)
    uint32_t* xb[5];
    const int xsz[5] = {1024, 4096, 2048, 1024, 6144 /* only 3072 used as K of down: M of gate/up stand-in is 6144 */};
    for (int j = 0; j < 5; j++) {
        CK(hipMalloc(&xb[j], 8192 * 4));
        CK(hipMemset(xb[j], 0xff, 8192 * 4));
    }
    (void)xsz;
    int *d_epoch, *d_err;
    CK(hipMalloc(&d_epoch, 4));
    CK(hipMalloc(&d_err, 4));
    CK(hipMemset(d_err, 0, 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));

    unsigned long long* glog;
    CK(hipMalloc(&glog, 256 * 4 * 8));
    uint32_t digest_ref = 0;
    for (int variant = 0; variant < 6; variant++) {
        // variant: 0 = C1 plain loads (today), 1 = C1 sc1 polls, 2..4 = C = 2,3,4 polls, 5 = C=8
        const int C = variant <= 1 ? 1 : (variant == 5 ? 8 : variant);
        const bool poll = variant >= 1;
        int one = 1;
        CK(hipMemcpy(d_epoch, &one, 4, hipMemcpyHostToDevice));
        hipGraph_t g;
        CK(hipGraphCreate(&g, 0));
        std::vector<hipGraphNode_t> nodes(NP + 2);
        std::vector<PhaseArgs> pargs(NP);
        // seed node
        {
            hipKernelNodeParams kp = {};
            static uint32_t* x0;
            static int n0 = 1024;
            static const int* ep;
            x0 = xb[0], ep = d_epoch;
            static void* args[3];
            args[0] = &x0, args[1] = &n0, args[2] = &ep;
            kp.func = (void*)seed_kernel, kp.gridDim = dim3(4), kp.blockDim = dim3(256), kp.kernelParams = args;
            CK(hipGraphAddKernelNode(&nodes[0], g, nullptr, 0, &kp));
        }
        std::vector<void*> argp(NP);
        for (int p = 0; p < NP; p++) {
            const Shape& s = layer[p % 5];
            PhaseArgs& a = pargs[p];
            a.w = reinterpret_cast<const u32x4*>(reinterpret_cast<char*>(dw) + woff[p]);
            a.xin = xb[p % 5], a.xout = xb[(p + 1) % 5];
            a.d_epoch = d_epoch, a.d_err = d_err, a.d_log = glog;
            a.M = s.M, a.K = s.K, a.rows_per_wg = s.rows, a.idx = p, a.nphase = NP, a.work = work, a.last = p == NP - 1;
            hipKernelNodeParams kp = {};
            argp[p] = &a;
            kp.func = poll ? (void*)phase_kernel<1> : (void*)phase_kernel<0>;
            kp.gridDim = dim3(s.M / s.rows), kp.blockDim = dim3(256), kp.kernelParams = &argp[p];
            // dependencies: node p-C (same chain); the first C phases hang off the seed node
            hipGraphNode_t dep = p >= C ? nodes[1 + p - C] : nodes[0];
            CK(hipGraphAddKernelNode(&nodes[1 + p], g, &dep, 1, &kp));
        }
        {
            hipKernelNodeParams kp = {};
            static int* ep2;
            ep2 = d_epoch;
            static void* args[1];
            args[0] = &ep2;
            kp.func = (void*)bump_kernel, kp.gridDim = dim3(1), kp.blockDim = dim3(1), kp.kernelParams = args;
            std::vector<hipGraphNode_t> deps;
            for (int c = 0; c < C && c < NP; c++) deps.push_back(nodes[1 + NP - 1 - c]);
            CK(hipGraphAddKernelNode(&nodes[NP + 1], g, deps.data(), deps.size(), &kp));
        }
        hipGraphExec_t ge;
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; i++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        {
            int err0;
            CK(hipMemcpy(&err0, d_err, 4, hipMemcpyDeviceToHost));
            if (err0) {
                printf("variant %d: C=%d  polls timed out during warm-up (%d): chains not co-resident / ordered as assumed\n", variant, C, err0);
                CK(hipMemset(d_err, 0, 4));
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
                continue;
            }
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // digest of the final vector (values only)
        std::vector<uint32_t> out(1024);
        CK(hipMemcpy(out.data(), xb[NP % 5], 4096, hipMemcpyDeviceToHost));
        uint32_t dg = 0;
        for (int i = 0; i < 1024; i++) dg = dg * 31u + (out[i] & 0xffffu);
        int err;
        CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        // the digest depends on the number of replays only through nothing (x is reseeded each replay): all variants equal
        if (variant == 0) digest_ref = dg;
        printf("variant %d: C=%d poll=%d  %.1f us/replay  %.3f us/phase  digest %08x %s  timeouts %d\n", variant, C, (int)poll, ms * 1e3 / reps,
               ms * 1e3 / reps / NP, dg, dg == digest_ref ? "ok" : "MISMATCH", err);
        {
            std::vector<unsigned long long> lg(NP * 4);
            CK(hipMemcpy(lg.data(), glog, NP * 4 * 8, hipMemcpyDeviceToHost));
            for (int p = 60; p < 70; p++)
                printf("  phase %3d (M %4d K %4d): start +%.2f us, polled +%.2f, end +%.2f\n", p, layer[p % 5].M, layer[p % 5].K, (lg[p * 4] - lg[60 * 4]) / 100.0,
                       (lg[p * 4 + 1] - lg[60 * 4]) / 100.0, (lg[p * 4 + 2] - lg[60 * 4]) / 100.0);
        }
        fflush(stdout);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }

    // ---- persistent form: 6 phases per layer (QKV, attention, merge, o_proj, gate/up, down stand-ins)
    {
        const Shape pl[6] = {{4096, 1024, 16}, {2048, 2048, 8}, {2048, 1024, 8}, {1024, 2048, 4}, {6144, 1024, 24}, {1024, 3072, 4}};
        const int NPP = L * 6;
        std::vector<PPhase> hp(NPP);
        size_t off = 0;
        uint32_t* pb[6];
        for (int j = 0; j < 6; j++) {
            CK(hipMalloc(&pb[j], 8192 * 4));
            CK(hipMemset(pb[j], 0xff, 8192 * 4));
        }
        for (int p = 0; p < NPP; p++) {
            const Shape& sh = pl[p % 6];
            hp[p].w = reinterpret_cast<const u32x4*>(reinterpret_cast<char*>(dw) + off);
            off += (size_t)sh.M * sh.K / 2;
            hp[p].xin = pb[p % 6], hp[p].xout = pb[(p + 1) % 6];
            hp[p].M = sh.M, hp[p].K = sh.K, hp[p].rows_per_wg = sh.rows, hp[p].pad = 0;
        }
        if (off > wbytes) { printf("weight pool too small\n"); return 1; }
        PPhase* dph;
        CK(hipMalloc(&dph, sizeof(PPhase) * NPP));
        CK(hipMemcpy(dph, hp.data(), sizeof(PPhase) * NPP, hipMemcpyHostToDevice));
        unsigned long long* dlog;
        CK(hipMalloc(&dlog, NPP * 4 * 8));
        CK(hipMemset(dlog, 0, NPP * 4 * 8));
        for (int sl = 0; sl < 3; sl++) {
            int one = 1;
            CK(hipMemcpy(d_epoch, &one, 4, hipMemcpyHostToDevice));
            CK(hipMemset(d_err, 0, 4));
            PArgs pa;
            pa.ph = dph, pa.d_epoch = d_epoch, pa.d_err = d_err, pa.d_log = dlog, pa.nphase = NPP, pa.work = work, pa.sleep = sl, pa.log_wg = 77;
            auto once = [&]() {
                hipLaunchKernelGGL(seed_kernel, dim3(4), dim3(256), 0, st, pb[0], 1024, (const int*)d_epoch);
                hipLaunchKernelGGL(persist_kernel, dim3(256), dim3(1024), 0, st, pa);
                hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, st, d_epoch);
            };
            for (int i = 0; i < 3; i++) once();
            CK(hipStreamSynchronize(st));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; i++) once();
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            int err;
            CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
            std::vector<uint32_t> out(1024);
            CK(hipMemcpy(out.data(), pb[NPP % 6], 4096, hipMemcpyDeviceToHost));
            uint32_t dg = 0;
            for (int i = 0; i < 1024; i++) dg = dg * 31u + (out[i] & 0xffffu);
            printf("persistent sleep=%d: %.1f us/replay (3 launches)  %.3f us/phase over %d phases  digest %08x timeouts %d\n", sl, ms * 1e3 / reps, ms * 1e3 / reps / NPP, NPP, dg, err);
            std::vector<unsigned long long> lg(NPP * 4);
            CK(hipMemcpy(lg.data(), dlog, NPP * 4 * 8, hipMemcpyDeviceToHost));
            for (int p = 60; p < 72; p++)
                printf("  phase %3d (M %4d K %4d): poll start +%.2f us, polled +%.2f, stored +%.2f   (poll wait %.2f, compute+store %.2f)\n", p, hp[p].M, hp[p].K,
                       (lg[p * 4] - lg[60 * 4]) / 100.0, (lg[p * 4 + 1] - lg[60 * 4]) / 100.0, (lg[p * 4 + 2] - lg[60 * 4]) / 100.0,
                       (lg[p * 4 + 1] - lg[p * 4]) / 100.0, (lg[p * 4 + 2] - lg[p * 4 + 1]) / 100.0);
            fflush(stdout);
        }
    }
    return 0;
}
