// kf_attn_bwd_mfma.hip -- causal multi-head attention backward on MFMA (flash form), gfx950 / wave64.  Same contract as kf_attn_bwd.hip (its
// header has the mathematics and the reference), same two-launch split, the tile algebra of the forward kernel (kf_attn_prefill.hip):
//
//   dQ launch, workgroup = 128 query columns (32 per wave), key tiles of 32 staged in LDS (K and V row-major, K a second time with the row stride the transposing reads want):
//     pass A   S^T[key][col] = K . Q^T                      -> the column's log-sum-exp L (and D = dO . O from registers), kept for launch 2
//     pass B   S^T again, dP^T[key][col] = V . dO^T         A = K / V rows (ds_read_b128), B = Q / dO rows (registers, loaded once)
//              dS^T = exp(scale S^T - L) o (dP^T - D)       element-wise on the accumulators: a column lives in one lane pair
//              dQ^T[d][col] += K^T[d][key] . dS^T[key][col] the dS^T registers, packed to bf16, ARE the B operand (free contraction order)
//   dK/dV launch, workgroup = 128 key columns, query tiles of 32 from the diagonal down (Q and dO row-major, twice: row reads / transposing reads; L and D of the tile):
//              S[q][key] = Q . K^T, dP[q][key] = dO . V^T   A = Q / dO rows, B = K / V rows of the column (registers)
//              P = exp(scale S - L_q), dS = P o (dP - D_q)  L, D per accumulator ROW: four ds_read_b128 each
//              dV^T[d][key] += dO^T[d][q] . P[q][key],  dK^T[d][key] += Q^T[d][q] . dS[q][key]
// P and dS enter their products as single bf16 values (gradients are compared at 2^-7 of scale); everything else fp32.
#include "kf_kernels.h"

namespace kf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AB_T = 32;        /* tile of keys (dQ launch) / queries (dK/dV launch) */
constexpr int AB_PAD2 = 32;     /* row padding (elements, 64 B) of the second row-major copy that the transposing reads use: 4 rows of a 32-lane pass -> distinct 16-bank groups */
constexpr float AB_LOG2E = 1.44269502162933349609375f;

__device__ __forceinline__ f32x16 ab_zero() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ f32x16 ab_mfma(u32x4 A, u32x4 B, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), c, 0, 0, 0);
}
// stage a [32][HD] tile of rows (row index clamped to T - 1) row-major (padded rows) and, optionally, a second time with AB_PAD2 row padding (ab_tfrag's operand)
template <int HD, bool TRANSPOSE>
__device__ __forceinline__ void ab_stage(const uint16_t* __restrict__ src, long long ld, size_t hoff, int row0, int T, uint16_t* rows, uint16_t* tr) {
    constexpr int KS = HD + 8, CH = AB_T * HD / 8;
    for (int c = threadIdx.x; c < CH; c += 256) {
        const int row = c / (HD / 8), dc = c - row * (HD / 8);
        int rr = row0 + row;
        rr = rr < T ? rr : T - 1;
        const u32x4 x = *reinterpret_cast<const u32x4*>(src + (size_t)rr * ld + hoff + dc * 8);
        *reinterpret_cast<u32x4*>(rows + row * KS + dc * 8) = x;
        if (TRANSPOSE) *reinterpret_cast<u32x4*>(tr + row * (HD + AB_PAD2) + dc * 8) = x;
    }
}
// the same in two halves, so that a tile's global loads can be in flight while the previous tile is multiplied: registers <- global, LDS <- registers
template <int HD>
__device__ __forceinline__ void ab_load(const uint16_t* __restrict__ src, long long ld, size_t hoff, int row0, int T, u32x4* regs) {
    constexpr int CH = AB_T * HD / 8;
#pragma unroll
    for (int i = 0; i < CH / 256; i++) {
        const int c = threadIdx.x + 256 * i, row = c / (HD / 8), dc = c - row * (HD / 8);
        int rr = row0 + row;
        rr = rr < T ? rr : T - 1;
        regs[i] = *reinterpret_cast<const u32x4*>(src + (size_t)rr * ld + hoff + dc * 8);
    }
}
template <int HD, bool TRANSPOSE>
__device__ __forceinline__ void ab_store(const u32x4* regs, uint16_t* rows, uint16_t* tr) {
    constexpr int KS = HD + 8, CH = AB_T * HD / 8;
#pragma unroll
    for (int i = 0; i < CH / 256; i++) {
        const int c = threadIdx.x + 256 * i, row = c / (HD / 8), dc = c - row * (HD / 8);
        *reinterpret_cast<u32x4*>(rows + row * KS + dc * 8) = regs[i];
        if (TRANSPOSE) *reinterpret_cast<u32x4*>(tr + row * (HD + AB_PAD2) + dc * 8) = regs[i];
    }
}
// A fragment of X^T (row d = 32 db + (lane & 31), contraction slots j <-> tile rows 16 s2 + (j & 3) + 8 (j >> 2) + 4 h) read straight from the second row-major
// copy of the tile with two transposing ds_read_b64_tr_b16: lane l16 of a 16-lane group addresses tile row (l16 >> 2) of its 4 and d-piece 4 (l16 & 3) of the
// group's 16 d, and receives the 4 rows' values of d = l16 (scratch/dbg/ds_read_tr_probe.hip).  No transposed copy, no 16-bit scatter stores.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <int HD>
__device__ __forceinline__ u32x4 ab_tfrag(const uint16_t* rows2, int db, int s2, int lane) {
    constexpr int VS = HD + AB_PAD2;
    const uint16_t* p = rows2 + (16 * s2 + 4 * (lane >> 5) + ((lane & 15) >> 2)) * VS + db * 32 + (lane & 16) + 4 * (lane & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 8 * VS));
    return __builtin_bit_cast(u32x4, bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

template <int HD>
__global__ void __launch_bounds__(256) attn_bwd_dq_mfma_kernel(const uint16_t* q, const uint16_t* k, const uint16_t* v, long long ld_qkv, const uint16_t* o, const uint16_t* dO, long long ld_o,
                                                               uint16_t* dq, long long ld_d, float* Lbuf, float* Dbuf, int T, float scale, int gq, long long ld_kv) {
    constexpr int NS = HD / 16, NDB = HD / 32, KS = HD + 8;
    __shared__ __attribute__((aligned(16))) uint16_t ks[AB_T * KS], vs[AB_T * KS], kt[AB_T * (HD + AB_PAD2)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, tok0 = (gridDim.x - 1 - blockIdx.x) * 128; /* the longest columns first */
    const size_t hoff = (size_t)head * HD, hoff_kv = (size_t)(head / gq) * HD; /* GQA: gq query heads share a kv head */
    { /* sequence blockIdx.z: T rows further on in every tensor */
        const size_t ro = (size_t)blockIdx.z * T;
        q += ro * ld_qkv, k += ro * ld_kv, v += ro * ld_kv, o += ro * ld_o, dO += ro * ld_o, dq += ro * ld_d;
        Lbuf += (size_t)blockIdx.z * gridDim.y * T, Dbuf += (size_t)blockIdx.z * gridDim.y * T;
    }
    int tok = tok0 + wave * 32 + r;
    const bool col_ok = tok < T;
    if (!col_ok) tok = T - 1;
    int tok_last = tok0 + 127;
    tok_last = tok_last < T ? tok_last : T - 1;
    const int ntile = tok_last / AB_T + 1;

    u32x4 qf[NS], dof[NS];
    float Dp = 0.f;
    {
        const uint16_t* qrow = q + (size_t)tok * ld_qkv + hoff;
        const uint16_t* drow = dO + (size_t)tok * ld_o + hoff;
        const uint16_t* orow = o + (size_t)tok * ld_o + hoff;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            qf[s] = *reinterpret_cast<const u32x4*>(qrow + 16 * s + 8 * h);
            dof[s] = *reinterpret_cast<const u32x4*>(drow + 16 * s + 8 * h);
            const u32x4 of = *reinterpret_cast<const u32x4*>(orow + 16 * s + 8 * h);
            const uint32_t a[4] = {dof[s].x, dof[s].y, dof[s].z, dof[s].w}, b[4] = {of.x, of.y, of.z, of.w};
#pragma unroll
            for (int e = 0; e < 4; e++) Dp = fmaf(bf_lo(a[e]), bf_lo(b[e]), Dp), Dp = fmaf(bf_hi(a[e]), bf_hi(b[e]), Dp);
        }
    }
    const float D = Dp + __shfl_xor(Dp, 32, 64);

    // ---- pass A: log-sum-exp of every column
    float M = -__builtin_inff(), l = 0.f;
    for (int t = 0; t < ntile; t++) {
        __syncthreads();
        ab_stage<HD, false>(k, ld_kv, hoff_kv, t * AB_T, T, ks, nullptr);
        __syncthreads();
        f32x16 st = ab_zero();
#pragma unroll
        for (int s = 0; s < NS; s++) st = ab_mfma(*reinterpret_cast<const u32x4*>(ks + r * KS + 16 * s + 8 * h), qf[s], st);
        // scores in log2 units (scale * log2 e folded in): M, l are the running maximum / sum of 2^(s - M)
        float sc[16], mt = -__builtin_inff();
        const bool full = t * AB_T + AB_T - 1 <= tok0 + wave * 32;
        const float c1 = scale * AB_LOG2E;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int key = t * AB_T + 4 * h + (i & 3) + 8 * (i >> 2);
            sc[i] = st[i] * c1;
            if (!full) sc[i] = key <= tok ? sc[i] : -__builtin_inff();
            mt = fmaxf(mt, sc[i]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        if (mt > M) l *= __builtin_amdgcn_exp2f(M - mt), M = mt;
#pragma unroll
        for (int i = 0; i < 16; i++) l += __builtin_amdgcn_exp2f(sc[i] - M);
    }
    l += __shfl_xor(l, 32, 64);
    const float L = (M + __log2f(l)) * 0.693147182464599609375f; /* back to natural units: log-sum-exp of the scaled scores */
    if (col_ok && h == 0) Lbuf[(size_t)head * T + tok] = L, Dbuf[(size_t)head * T + tok] = D;

    // ---- pass B: dQ^T
    f32x16 acc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; db++) acc[db] = ab_zero();
    constexpr int CPT = AB_T * HD / 8 / 256;
    u32x4 kreg[CPT], vreg[CPT];
    __syncthreads(); /* pass A's last readers of ks are done */
    ab_load<HD>(k, ld_kv, hoff_kv, 0, T, kreg);
    ab_load<HD>(v, ld_kv, hoff_kv, 0, T, vreg);
    ab_store<HD, true>(kreg, ks, kt);
    ab_store<HD, false>(vreg, vs, nullptr);
    __syncthreads();
    for (int t = 0; t < ntile; t++) {
        const bool more = t + 1 < ntile;
        if (more) { /* the next tile's rows travel while this one is multiplied */
            ab_load<HD>(k, ld_kv, hoff_kv, (t + 1) * AB_T, T, kreg);
            ab_load<HD>(v, ld_kv, hoff_kv, (t + 1) * AB_T, T, vreg);
        }
        f32x16 st = ab_zero(), dpt = ab_zero();
#pragma unroll
        for (int s = 0; s < NS; s++) {
            st = ab_mfma(*reinterpret_cast<const u32x4*>(ks + r * KS + 16 * s + 8 * h), qf[s], st);
            dpt = ab_mfma(*reinterpret_cast<const u32x4*>(vs + r * KS + 16 * s + 8 * h), dof[s], dpt);
        }
        uint32_t dw[8];
        const bool full = t * AB_T + AB_T - 1 <= tok0 + wave * 32; /* every key of the tile precedes every column of this wave: no mask */
        const float c1 = scale * AB_LOG2E, L2 = L * AB_LOG2E;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float ds[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int ii = 2 * i + e, key = t * AB_T + 4 * h + (ii & 3) + 8 * (ii >> 2);
                float p = __builtin_amdgcn_exp2f(fmaf(st[ii], c1, -L2));
                if (!full) p = key <= tok ? p : 0.f;
                ds[e] = p * (dpt[ii] - D);
            }
            dw[i] = pack_bf16x2(ds[0], ds[1]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const u32x4 B = u32x4{dw[4 * s2], dw[4 * s2 + 1], dw[4 * s2 + 2], dw[4 * s2 + 3]};
#pragma unroll
            for (int db = 0; db < NDB; db++) acc[db] = ab_mfma(ab_tfrag<HD>(kt, db, s2, lane), B, acc[db]);
        }
        __syncthreads(); /* everyone has read this tile */
        if (more) {
            ab_store<HD, true>(kreg, ks, kt);
            ab_store<HD, false>(vreg, vs, nullptr);
        }
        __syncthreads();
    }
    if (!col_ok) return;
    uint16_t* out = dq + (size_t)tok * ld_d + hoff;
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int d = 32 * db + 8 * g + 4 * h;
            *reinterpret_cast<u32x2*>(out + d) = u32x2{pack_bf16x2(acc[db][4 * g] * scale, acc[db][4 * g + 1] * scale), pack_bf16x2(acc[db][4 * g + 2] * scale, acc[db][4 * g + 3] * scale)};
        }
}

template <int HD>
__global__ void __launch_bounds__(256) attn_bwd_dkv_mfma_kernel(const uint16_t* q, const uint16_t* k, const uint16_t* v, long long ld_qkv, const uint16_t* dO, long long ld_o, uint16_t* dk,
                                                                uint16_t* dv, long long ld_d, const float* Lbuf, const float* Dbuf, int T, float scale, int gq, long long ld_kv, long long ld_dkv) {
    constexpr int NS = HD / 16, NDB = HD / 32, KS = HD + 8;
    __shared__ __attribute__((aligned(16))) uint16_t qs[AB_T * KS], os[AB_T * KS], qt[AB_T * (HD + AB_PAD2)], ot[AB_T * (HD + AB_PAD2)];
    __shared__ __attribute__((aligned(16))) float Ls[AB_T], Ds[AB_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int kvh = blockIdx.y, key0 = blockIdx.x * 128; /* blockIdx.y = kv head; the first key columns see the most query tiles: already first */
    const size_t hoff = (size_t)kvh * HD;
    {
        const size_t ro = (size_t)blockIdx.z * T;
        q += ro * ld_qkv, k += ro * ld_kv, v += ro * ld_kv, dO += ro * ld_o, dk += ro * ld_dkv, dv += ro * ld_dkv;
        Lbuf += (size_t)blockIdx.z * gridDim.y * gq * T, Dbuf += (size_t)blockIdx.z * gridDim.y * gq * T;
    }
    int key = key0 + wave * 32 + r;
    const bool key_ok = key < T;
    if (!key_ok) key = T - 1;
    const int wave_key0 = key0 + wave * 32;

    u32x4 kf_[NS], vf[NS];
    {
        const uint16_t* krow = k + (size_t)key * ld_kv + hoff;
        const uint16_t* vrow = v + (size_t)key * ld_kv + hoff;
#pragma unroll
        for (int s = 0; s < NS; s++) kf_[s] = *reinterpret_cast<const u32x4*>(krow + 16 * s + 8 * h), vf[s] = *reinterpret_cast<const u32x4*>(vrow + 16 * s + 8 * h);
    }
    f32x16 dka[NDB], dva[NDB];
#pragma unroll
    for (int db = 0; db < NDB; db++) dka[db] = ab_zero(), dva[db] = ab_zero();
    const int ntq = (T + AB_T - 1) / AB_T;
    constexpr int CPT = AB_T * HD / 8 / 256;
    u32x4 qreg[CPT], oreg[CPT];
    float lreg = 0.f, dreg = 0.f;
    const int tg0 = (key0 / AB_T) * gq, tg1 = ntq * gq;
    auto fetch = [&](int tg) { /* (query tile, query head of the group) pair tg: rows of Q and dO, L and D of the tile's queries */
        const int t = tg / gq, head = kvh * gq + (tg - t * gq);
        ab_load<HD>(q, ld_qkv, (size_t)head * HD, t * AB_T, T, qreg);
        ab_load<HD>(dO, ld_o, (size_t)head * HD, t * AB_T, T, oreg);
        if (tid < AB_T) {
            const int qi = t * AB_T + tid;
            lreg = qi < T ? Lbuf[(size_t)head * T + qi] * AB_LOG2E : 0.f, dreg = qi < T ? Dbuf[(size_t)head * T + qi] : 0.f;
        }
    };
    auto put = [&]() {
        ab_store<HD, true>(qreg, qs, qt);
        ab_store<HD, true>(oreg, os, ot);
        if (tid < AB_T) Ls[tid] = lreg, Ds[tid] = dreg;
    };
    if (tg0 < tg1) fetch(tg0);
    put();
    __syncthreads();
    for (int tg = tg0; tg < tg1; tg++) {
        const int t = tg / gq;
        const bool more = tg + 1 < tg1;
        if (more) fetch(tg + 1);
        if (t * AB_T + AB_T - 1 >= wave_key0) { /* else: every query of the tile precedes this wave's keys */
        f32x16 st = ab_zero(), dpt = ab_zero();
#pragma unroll
        for (int s = 0; s < NS; s++) {
            st = ab_mfma(*reinterpret_cast<const u32x4*>(qs + r * KS + 16 * s + 8 * h), kf_[s], st);
            dpt = ab_mfma(*reinterpret_cast<const u32x4*>(os + r * KS + 16 * s + 8 * h), vf[s], dpt);
        }
        uint32_t pw[8], dw[8];
        /* every query of the tile follows every key of this wave and lies inside the sequence: no mask */
        const bool full = t * AB_T >= wave_key0 + 31 && t * AB_T + AB_T <= T && wave_key0 + 31 < T;
        const float c1 = scale * AB_LOG2E;
#pragma unroll
        for (int g = 0; g < 4; g++) { /* accumulator rows 8 g + 4 h + {0..3}: one 16-byte read of L and of D */
            const f32x4 Lq = *reinterpret_cast<const f32x4*>(Ls + 8 * g + 4 * h), Dq = *reinterpret_cast<const f32x4*>(Ds + 8 * g + 4 * h);
            float p[4], ds[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int qi = t * AB_T + 8 * g + 4 * h + e;
                p[e] = __builtin_amdgcn_exp2f(fmaf(st[4 * g + e], c1, -Lq[e])); /* Ls holds L * log2 e */
                if (!full) p[e] = (key_ok && qi < T && qi >= key) ? p[e] : 0.f;
                ds[e] = p[e] * (dpt[4 * g + e] - Dq[e]);
            }
            pw[2 * g] = pack_bf16x2(p[0], p[1]), pw[2 * g + 1] = pack_bf16x2(p[2], p[3]);
            dw[2 * g] = pack_bf16x2(ds[0], ds[1]), dw[2 * g + 1] = pack_bf16x2(ds[2], ds[3]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const u32x4 BP = u32x4{pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]}, BS = u32x4{dw[4 * s2], dw[4 * s2 + 1], dw[4 * s2 + 2], dw[4 * s2 + 3]};
#pragma unroll
            for (int db = 0; db < NDB; db++) {
                dva[db] = ab_mfma(ab_tfrag<HD>(ot, db, s2, lane), BP, dva[db]);
                dka[db] = ab_mfma(ab_tfrag<HD>(qt, db, s2, lane), BS, dka[db]);
            }
        }
        } /* wave has work on this tile */
        __syncthreads();
        if (more) put();
        __syncthreads();
    }
    if (!key_ok) return;
    uint16_t* okp = dk + (size_t)key * ld_dkv + hoff;
    uint16_t* ovp = dv + (size_t)key * ld_dkv + hoff;
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int d = 32 * db + 8 * g + 4 * h;
            *reinterpret_cast<u32x2*>(okp + d) = u32x2{pack_bf16x2(dka[db][4 * g] * scale, dka[db][4 * g + 1] * scale), pack_bf16x2(dka[db][4 * g + 2] * scale, dka[db][4 * g + 3] * scale)};
            *reinterpret_cast<u32x2*>(ovp + d) = u32x2{pack_bf16x2(dva[db][4 * g], dva[db][4 * g + 1]), pack_bf16x2(dva[db][4 * g + 2], dva[db][4 * g + 3])};
        }
}

// KF_OK launched; 1 = shape not covered by the MFMA form
int attn_backward_mfma_launch(hipStream_t st, const uint16_t* q, const uint16_t* k, const uint16_t* v, long long ld_qkv, const uint16_t* o, const uint16_t* dO, long long ld_o,
                              uint16_t* dq, uint16_t* dk, uint16_t* dv, long long ld_d, int T, int n_head, int hd, int n_seq, float* scratch, int n_kv, long long ld_kv, long long ld_dkv) {
    if ((hd != 64 && hd != 128) || n_kv < 1 || n_head % n_kv != 0) return 1;
    const int gq = n_head / n_kv;
    const float scale = 1.0f / sqrtf((float)hd);
    float* Lb = scratch;
    float* Db = scratch + (size_t)n_seq * n_head * T;
    const dim3 grid((T + 127) / 128, n_head, n_seq), grid_kv((T + 127) / 128, n_kv, n_seq);
    if (hd == 64) {
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<64>), grid, dim3(256), 0, st, q, k, v, ld_qkv, o, dO, ld_o, dq, ld_d, Lb, Db, T, scale, gq, ld_kv);
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<64>), grid_kv, dim3(256), 0, st, q, k, v, ld_qkv, dO, ld_o, dk, dv, ld_d, Lb, Db, T, scale, gq, ld_kv, ld_dkv);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<128>), grid, dim3(256), 0, st, q, k, v, ld_qkv, o, dO, ld_o, dq, ld_d, Lb, Db, T, scale, gq, ld_kv);
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<128>), grid_kv, dim3(256), 0, st, q, k, v, ld_qkv, dO, ld_o, dk, dv, ld_d, Lb, Db, T, scale, gq, ld_kv, ld_dkv);
    }
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
