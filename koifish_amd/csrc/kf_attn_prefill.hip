// kf_attn_prefill.hip -- causal GQA attention for a batch of prompt tokens on MFMA (flash form), gfx950 / wave64.
//
// Reference: none on the inference side (Fish::Chat prefills token by token through attention_qk_kernel / CU_softmax_multihead /
// attention_v_kernel, QKV.cu:669-673); its batched forward is cuDNN SDPA on the training side (QKV.cu:130-315), removed here.
// Same arithmetic contract as the decode kernel (kf_attn.hip): score = bf16(dot / sqrt(hd)) -- the store the reference makes --, fp32
// softmax statistics with exp as v_exp_f32(x * log2 e), out = (sum_t p_t v_t) / (sum_t p_t) with one bf16 store; the probabilities
// enter the P.V product as a bf16 pair (high part + remainder, ~16 significant bits; the reference keeps them in plain bf16,
// operator.cuh:251-277), the sum of p stays fp32.
//
// Workgroup = one kv-head x TQ = 128/GQ consecutive tokens: 128 "columns" (token, query head of the group), 32 per wave.
// Per 32-key tile, staged once in LDS for the 4 waves (K row-major with 16 bytes of row padding, V row-major too with 64 bytes of padding):
//   S^T[key][col] = sum_d K[key][d] Q[col][d]      v_mfma_f32_32x32x16_bf16, A = K rows (ds_read_b128), B = Q rows (registers, loaded once)
//   -> a column's 32 scores sit in ONE lane pair (lane l and l^32, 16 registers each): max / exp / sum need no cross-lane traffic
//      beyond one lane-pair exchange of the tile maximum;
//   O^T[d][col]  += sum_key V^T[d][key] P^T[key][col]   the S^T accumulator registers, packed to bf16, ARE the B operand (the contraction
//      index may be visited in any order: slot (step s, half h, j) <-> key 16 s + (j & 3) + 8 (j >> 2) + 4 h on both operands), A = V^T
//      fragments read straight from the row-major V tile with two transposing ds_read_b64_tr_b16 (4 keys x 16 d per 16 lanes; the 192- / 320-byte row
//      stride puts the 4 rows of a 32-lane pass on different banks) -- no transposed copy, no 16-bit scatter stores.
// Online softmax: running maximum per column, O and l rescaled when it moves.  HBM/L2 bytes: K and V once per 128 columns.
#include <type_traits>

#include "kf_kernels.h"

namespace kf {
#ifdef AP_STAMP /* scratch/build_variant.py ... -DAP_STAMP: cycles per segment of the key walk, summed by wave 2 of workgroup (0, 0) -- a long walk of the even key half */
__device__ unsigned long long g_ap_stamp[8];
#define AP_T(k)                                                          \
    do {                                                                 \
        if (stamping) {                                                  \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            acc_t[k] += now_ - t_last, t_last = now_;                    \
        }                                                                \
    } while (0)
#else
#define AP_T(k)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AP_VPAD = 32;      /* V row padding, elements (64 B): rows 4 apart in a transposing read land on distinct 16-bank groups */

struct AttnPrefillArgs {
    const uint16_t* q;
    const uint16_t* kcache;
    const uint16_t* vcache;
    uint16_t* out;
    int pos0, n_tok, n_kv, kv_stride;
    int n_seq; /* sequences of n_tok rows each, back to back in q / out and in the K / V rows (blockIdx.z); 1 for prompt prefill */
    long long q_stride;
    long long out_stride; /* row stride of out (a training step reads q out of the fused [rows, 3C] buffer and writes a dense [rows, C] out) */
    float rden;
};

__device__ __forceinline__ float ap_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr bool PSPLIT = true; /* probabilities enter P.V as bf16 high + bf16 remainder (two MFMAs) */

// KH = 2: eight waves; waves 0-3 walk the even key tiles, waves 4-7 the odd ones (each half with its own pair of LDS tile buffers), and the two partial softmaxes
// (O, running maximum, sum) of a column are merged through LDS at the end.  A workgroup's time is the walk of its longest column over its keys: halved -- which is
// what a prompt of a few thousand tokens needs, where there is about one workgroup per CU and the launch lasts as long as the last query block.
// (Halving the columns per workgroup instead -- 64 columns, twice the workgroups -- was measured too: 72.8 vs 67.3 us at 2047 tokens; a workgroup's time is its
// key walk, not its column count, and the second workgroup of a heavy block does not land on an idle CU.  Cutting the late blocks' KEY ranges into pieces for other
// workgroups -- partial (O, max, sum) through memory, owners dispatched last and merging in key order -- was built and measured as well: 70.7 us in the 2047-token
// prefill against 67 us without, so it is not in the tree; the launch behaves throughput-bound at ~250 TFLOP/s of causal flops rather than tail-bound.)
template <int HD, int GQ, int KH, int KT>
__global__ void __launch_bounds__(256 * KH) attn_prefill_kernel(const AttnPrefillArgs a) {
    constexpr int NSUB = KT / 32; /* 32-key sub-tiles per staged tile */
    constexpr int NS = HD / 16;   /* MFMA steps over d for S^T */
    constexpr int NDB = HD / 32;  /* 32-row blocks of O^T */
    constexpr int KS = HD + 8;    /* padded K row, elements */
    constexpr int TQ = 128 / GQ;  /* tokens per workgroup */
    constexpr int KCH = KT * HD / 8; /* 16-byte chunks per K (or V) tile */
    constexpr int CPT = KCH / 256;      /* chunks per thread */
    static_assert(KCH % 256 == 0, "tile chunks must divide among 256 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int VS = HD + AP_VPAD; /* V row stride, elements */
    constexpr int HALF_EL = 2 * KT * KS + 2 * KT * VS; /* one key-half's buffers, elements */
    const int kh = KH == 2 ? (int)(threadIdx.x >> 8) : 0; /* which key tiles this wave walks: t = kh, kh + KH, ... */
    uint16_t* ks = reinterpret_cast<uint16_t*>(smem_raw) + (size_t)kh * HALF_EL;  // 2 x [KT][KS]
    uint16_t* vt = ks + 2 * KT * KS;                                           // 2 x [KT][VS]

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // workgroup -> (kv-head g, block bx): consecutive workgroups go to consecutive XCDs, so g = id mod n_kv puts all the workgroups of a kv-head on the same XCD(s) -- its K / V
    // rows (1 MB at 2047 tokens) then stay in that XCD's 4 MB L2 instead of every XCD streaming all heads' rows through the fabric (199 MB per launch at 2047 tokens)
    const int wg_id = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
    const int g = wg_id % a.n_kv, bx = wg_id / a.n_kv;
    const size_t seq_row = (size_t)blockIdx.z * a.n_tok;
    // Which tokens a wave's 32 columns are.  KH = 1: the workgroup is TQ consecutive tokens, the blocks with the most keys dispatched first.  KH = 2 (about one workgroup per
    // CU: the launch lasts as long as its heaviest workgroup, and under the causal mask the last query block walks twice the keys of the average one -- measured at 2047 tokens:
    // matrix + vector pipe time of the average SIMD 25 us, of the last block's 57 us, launch 65 us): the workgroup is TWO half blocks of TQ / 2 tokens, number x from the front
    // (two waves per key half) and number x from the back (the other two), so every workgroup walks the same number of (tile, wave) pairs; the tiles are staged for all eight waves up to the
    // later half block's last key, a wave multiplies only the tiles its own columns attend to.
    constexpr bool PAIR = KH == 2;
    constexpr int HT = TQ / 2;
    int sb_first, sb_tokens, col_in; /* first token and token count of this wave's (half) block, the wave's first column inside it */
    bool half_ok = true;
    int kmax;                        /* last key any column of this workgroup needs */
    if constexpr (PAIR) {
        // (SIMD = wave % 4 holds wave w of the even key half and wave w of the odd one: the back half block -- the long walk -- sits on waves 2-3 of the even half and on
        // waves 0-1 of the odd half, so that every SIMD runs one long and one short walk; with both long walks on SIMDs 2-3 the launch took what it took unpaired.)
        const int nsb = (a.n_tok + HT - 1) / HT, lo = bx, hi = nsb - 1 - bx, half = ((wave >> 1) ^ kh) & 1;
        sb_first = (half ? hi : lo) * HT, sb_tokens = HT, col_in = (wave & 1) * 32;
        half_ok = half || lo != hi; /* an odd count's middle block belongs to the back half's waves alone */
        int last = hi * HT + HT - 1;
        kmax = a.pos0 + (last < a.n_tok - 1 ? last : a.n_tok - 1);
    } else {
        sb_first = ((int)gridDim.x - 1 - bx) * TQ, sb_tokens = TQ, col_in = wave * 32;
        int last = sb_first + TQ - 1;
        kmax = a.pos0 + (last < a.n_tok - 1 ? last : a.n_tok - 1);
    }
    const int colh = col_in + r;
    const int tq = colh / GQ, hq = colh - tq * GQ;
    int tok = sb_first + tq;
    const bool col_ok = half_ok && tok < a.n_tok;
    if (tok >= a.n_tok) tok = a.n_tok - 1;
    const int pos_q = a.pos0 + tok; /* last key this column attends to */
    const int ntile = kmax / KT + 1;
    int ntile_w = ntile; /* tiles this wave multiplies */
    if constexpr (PAIR) {
        int last = sb_first + sb_tokens - 1;
        last = last < a.n_tok - 1 ? last : a.n_tok - 1;
        ntile_w = half_ok ? (a.pos0 + last) / KT + 1 : 0;
    }
    const int wave_tok0 = sb_first + col_in / GQ; /* the first token among this wave's columns */
    const bool block_whole = sb_first + sb_tokens <= a.n_tok;

    // Q fragments: B operand of S^T, lane (col, h) holds Q[col][16 s + 8 h .. + 8]
    u32x4 qf[NS];
    {
        const uint16_t* qrow = a.q + (seq_row + tok) * a.q_stride + (size_t)(g * GQ + hq) * HD;
#pragma unroll
        for (int s = 0; s < NS; s++) qf[s] = *reinterpret_cast<const u32x4*>(qrow + 16 * s + 8 * h);
    }

    // K / V tiles travel global -> registers -> LDS.  Round 4: TWO register sets, a tile's loads are issued two steps before it is multiplied (the step that follows the issue
    // stores the OTHER set): with one set the loads of step j + 1 had only step j's arithmetic (~0.4 us) to land in, and a step took what the load took (~2 us from the
    // fabric at one workgroup per CU: 67 us for 32 steps at 2047 tokens).
    // (KH = 1 -- more workgroups than CUs, two resident per CU hide each other's loads -- keeps one set: a second one takes it past 256 registers.)
    constexpr int NSET = (KH == 2 && KT == 32) ? 2 : 1; /* 64-key tiles: one set of 8 + 8 registers; a step is twice as long */
    u32x4 kr[NSET][CPT], vr[NSET][CPT];
    auto tload = [&](int t, auto set_c) {
        constexpr int S = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = tid + 256 * i, key = c / (HD / 8), dc = c - key * (HD / 8);
            int kk = t * KT + key;
            if (kk > kmax) kk = kmax; /* rows past the last needed key are masked below; never read past it */
            const size_t off = (seq_row + kk) * a.kv_stride + (size_t)g * HD + dc * 8;
            kr[S][i] = *reinterpret_cast<const u32x4*>(a.kcache + off);
            vr[S][i] = *reinterpret_cast<const u32x4*>(a.vcache + off);
        }
    };
    auto tstore = [&](int buf, auto set_c) {
        constexpr int S = decltype(set_c)::value;
        uint16_t* kd = ks + (size_t)buf * KT * KS;
        uint16_t* vd = vt + (size_t)buf * KT * VS;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = tid + 256 * i, key = c / (HD / 8), dc = c - key * (HD / 8);
            *reinterpret_cast<u32x4*>(kd + key * KS + dc * 8) = kr[S][i];
            *reinterpret_cast<u32x4*>(vd + key * VS + dc * 8) = vr[S][i];
        }
    };

    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int i = 0; i < 16; i++) o[db][i] = 0.f;
    float M = -__builtin_inff(), l = 0.f;
    const float LOG2E = 1.44269502162933349609375f;

#ifdef AP_STAMP
    const bool stamping = bx == 0 && g == 0 && blockIdx.z == 0 && threadIdx.x == 128;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
#endif
    const int nstep = (ntile + KH - 1) / KH; /* both halves make the same number of steps (and barriers); the odd half may find its last one empty */
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    tload(kh < ntile ? kh : ntile - 1, S0{});
    if constexpr (NSET == 2) tload(kh + KH < ntile ? kh + KH : ntile - 1, S1{});
    tstore(0, S0{});
    __syncthreads();
    auto step = [&](int j, auto cur_c) { /* cur = j & 1: the LDS buffer of tile j, and the register set that takes tile j + 2 while set 1 - cur (tile j + 1) goes to LDS */
        constexpr int cur = decltype(cur_c)::value;
        using SC = std::integral_constant<int, NSET == 2 ? cur : 0>;
        using SN = std::integral_constant<int, NSET == 2 ? 1 - cur : 0>;
        const int t = kh + KH * j;
        const bool more = t + KH < ntile;
        if constexpr (NSET == 2) {
            if (t + 2 * KH < ntile) tload(t + 2 * KH, SC{});
        } else {
            if (more) tload(t + KH, SC{});
        }
        AP_T(0); /* loads issued (+ the barrier wait of the step before) */
        if (t < ntile_w) {
        const uint16_t* kb = ks + (size_t)cur * KT * KS;
        const uint16_t* vb = vt + (size_t)cur * KT * VS;
        // ---- S^T: NSUB sub-tiles of 32 keys, each its own accumulator -- their MFMA chains are independent and interleave (a lone chain of 8 dependent MFMAs leaves the
        // matrix pipe idle half of the time: with one long-walk wave per SIMD nothing else fills it)
        f32x16 st[NSUB];
#pragma unroll
        for (int u = 0; u < NSUB; u++)
#pragma unroll
            for (int i = 0; i < 16; i++) st[u][i] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int u = 0; u < NSUB; u++) {
                const u32x4 A = *reinterpret_cast<const u32x4*>(kb + (u * 32 + r) * KS + 16 * s + 8 * h);
                st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, qf[s]), st[u], 0, 0, 0);
            }
        AP_T(1); /* S MFMAs issued */
        // ---- scores (kept in the accumulator registers): bf16 store, causal mask, tile maximum of this column
        float mt = -__builtin_inff();
        // every key of the tile precedes every column of this wave (its first token is wave_tok0): no mask -- all tiles but the last few
        const bool full = t * KT + KT - 1 <= a.pos0 + wave_tok0 && block_whole;
        if (full) {
#pragma unroll
            for (int u = 0; u < NSUB; u++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    st[u][i] = round_bf16(st[u][i] * a.rden);
                    mt = fmaxf(mt, st[u][i]);
                }
        } else {
#pragma unroll
            for (int u = 0; u < NSUB; u++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int key = t * KT + u * 32 + 4 * h + (i & 3) + 8 * (i >> 2);
                    const float v = round_bf16(st[u][i] * a.rden);
                    st[u][i] = key <= pos_q ? v : -__builtin_inff();
                    mt = fmaxf(mt, st[u][i]);
                }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        AP_T(2); /* scores + maximum */
        if (mt > M) { /* the column's running maximum moves: rescale what has been accumulated */
            const float alpha = ap_exp2((M - mt) * LOG2E); /* M = -inf on the first tile: 0 */
            l *= alpha;
#pragma unroll
            for (int db = 0; db < NDB; db++)
#pragma unroll
                for (int i = 0; i < 16; i++) o[db][i] *= alpha;
            M = mt;
        }
        // p as a bf16 pair (high part + rounded remainder): the P.V product then carries ~16 significant bits of p, which keeps the batch
        // form within rounding noise of the decode kernel's fp32 probabilities (bf16 alone: up to 4 bf16 ulps on the next layer's K rows)
        // (KH = 2: a column whose keys all lie in the other half's tiles has seen nothing yet -- M = -inf; its p must be 0, not exp2(-inf + inf))
        const float Mref = (KH == 2 && M == -__builtin_inff()) ? 0.f : M;
        AP_T(3); /* rescale */
        uint32_t pw[NSUB][8], pl[NSUB][8];
#pragma unroll
        for (int u = 0; u < NSUB; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float p0 = ap_exp2((st[u][2 * i] - Mref) * LOG2E), p1 = ap_exp2((st[u][2 * i + 1] - Mref) * LOG2E); /* masked: exp2(-inf) = 0 */
                l += p0;
                l += p1;
                pw[u][i] = pack_bf16x2(p0, p1);
                if (PSPLIT) pl[u][i] = pack_bf16x2(p0 - bf_lo(pw[u][i]), p1 - bf_hi(pw[u][i]));
            }
        AP_T(4); /* exp + pack */
        // ---- O^T += V^T . P^T : step (u, s2) takes accumulator registers 8 s2 .. 8 s2 + 7 of sub-tile u = keys 32 u + 16 s2 + {0..3, 8..11} + 4 h
#pragma unroll
        for (int u = 0; u < NSUB; u++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const u32x4 B = u32x4{pw[u][4 * s2], pw[u][4 * s2 + 1], pw[u][4 * s2 + 2], pw[u][4 * s2 + 3]};
#pragma unroll
                for (int db = 0; db < NDB; db++) {
                    // lane l16 of a 16-lane group addresses key-row (l16 >> 2) of its 4, d-piece 4 (l16 & 3) of the group's 16 d; it receives the 4 keys of d = l16
                    const uint16_t* vp = vb + (32 * u + 16 * s2 + 4 * h + ((lane & 15) >> 2)) * VS + db * 32 + (lane & 16) + 4 * (lane & 3);
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * VS));
                    const bf16x8 A8 = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const u32x4 A = __builtin_bit_cast(u32x4, A8);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), o[db], 0, 0, 0);
                    if (PSPLIT) {
                        const u32x4 B2 = u32x4{pl[u][4 * s2], pl[u][4 * s2 + 1], pl[u][4 * s2 + 2], pl[u][4 * s2 + 3]};
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B2), o[db], 0, 0, 0);
                    }
                }
            }
        AP_T(5); /* P.V issued */
        } /* t < ntile_w */
        if (more) tstore(1 - cur, SN{});
        AP_T(6); /* tile stored */
        __syncthreads();
    };
    for (int j = 0; j < nstep; j += 2) {
        step(j, S0{});
        if (j + 1 < nstep) step(j + 1, S1{});
    }
#ifdef AP_STAMP
    if (stamping) {
        for (int k = 0; k < 7; k++) g_ap_stamp[k] = acc_t[k];
        g_ap_stamp[7] = nstep;
    }
#endif
    if (KH == 2) { /* the odd half hands (O, M, l) over, register-major rows of 256 lanes; the even half merges and finishes */
        float* ex = reinterpret_cast<float*>(smem_raw);
        constexpr int NR = NDB * 16;
        if (kh == 1) {
            const int et = PAIR ? tid ^ 128 : tid; /* the even half's thread that holds the same column (paired form: its wave number differs in bit 1) */
#pragma unroll
            for (int db = 0; db < NDB; db++)
#pragma unroll
                for (int i = 0; i < 16; i++) ex[(db * 16 + i) * 256 + et] = o[db][i];
            ex[NR * 256 + et] = M, ex[(NR + 1) * 256 + et] = l;
        }
        __syncthreads();
        if (kh == 1) return;
        const float M1 = ex[NR * 256 + tid], l1 = ex[(NR + 1) * 256 + tid];
        const float Mm = fmaxf(M, M1); /* the even half has seen key 0: finite */
        const float a0 = ap_exp2((M - Mm) * LOG2E), a1 = ap_exp2((M1 - Mm) * LOG2E);
        l = l * a0 + l1 * a1;
#pragma unroll
        for (int db = 0; db < NDB; db++)
#pragma unroll
            for (int i = 0; i < 16; i++) o[db][i] = o[db][i] * a0 + ex[(db * 16 + i) * 256 + tid] * a1;
    }
    // ---- out[col][d] = O^T[d][col] / l ; lane (col, h) holds d = 32 db + (i & 3) + 8 (i >> 2) + 4 h
    l += __shfl_xor(l, 32, 64);
    if (!col_ok) return;
    const float inv = 1.0f / l;
    uint16_t* orow = a.out + (seq_row + tok) * a.out_stride + (size_t)(g * GQ + hq) * HD;
#pragma unroll
    for (int db = 0; db < NDB; db++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const int d = 32 * db + 8 * gq + 4 * h;
            const uint32_t w0 = pack_bf16x2(o[db][4 * gq] * inv, o[db][4 * gq + 1] * inv), w1 = pack_bf16x2(o[db][4 * gq + 2] * inv, o[db][4 * gq + 3] * inv);
            *reinterpret_cast<u32x2*>(orow + d) = u32x2{w0, w1};
        }
}

constexpr int AP_KT_LONG = 64; /* keys per staged tile of the long-prompt form (KH = 2); 32 otherwise */
template <int HD, int GQ, int KH>
static int ap_go(hipStream_t st, const AttnPrefillArgs& a, dim3 grid) {
    constexpr int KT = KH == 2 ? AP_KT_LONG : 32;
    constexpr size_t smem = sizeof(uint16_t) * KH * (2 * (size_t)KT * (HD + 8) + 2 * (size_t)KT * (HD + AP_VPAD));
    static_assert(smem <= 160 * 1024, "LDS");
    static int attr_set = 0;
    if (!attr_set && smem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)attn_prefill_kernel<HD, GQ, KH, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return KF_HIP_CHECK;
        attr_set = 1;
    }
    hipLaunchKernelGGL((attn_prefill_kernel<HD, GQ, KH, KT>), grid, dim3(256 * KH), smem, st, a);
    return 0;
}
template <int HD>
static int ap_launch_gq(hipStream_t st, const AttnPrefillArgs& a, int GQ, dim3 grid, int kh) {
    switch (GQ) {
        case 1: return kh == 2 ? ap_go<HD, 1, 2>(st, a, grid) : ap_go<HD, 1, 1>(st, a, grid);
        case 2: return kh == 2 ? ap_go<HD, 2, 2>(st, a, grid) : ap_go<HD, 2, 1>(st, a, grid);
        case 4: return kh == 2 ? ap_go<HD, 4, 2>(st, a, grid) : ap_go<HD, 4, 1>(st, a, grid);
        case 8: return kh == 2 ? ap_go<HD, 8, 2>(st, a, grid) : ap_go<HD, 8, 1>(st, a, grid);
        default: return 1;
    }
}

// KF_OK launched; 1 = shape not covered (the caller falls back to the per-token kernel)
int attn_prefill_mfma_launch(hipStream_t st, const uint16_t* q, const uint16_t* kc, const uint16_t* vc, uint16_t* out, int pos0, int n_tok, long long q_stride,
                             int n_head, int n_kv, int hd, int kv_stride, int n_seq, long long out_stride) {
    if (out_stride <= 0) out_stride = q_stride;
    if (out_stride & 3) return 1;
    if ((hd != 64 && hd != 128) || n_kv <= 0 || n_head % n_kv != 0) return 1;
    if ((q_stride & 7) != 0 || (kv_stride & 7) != 0 || (reinterpret_cast<uintptr_t>(q) & 15) != 0 || (reinterpret_cast<uintptr_t>(out) & 7) != 0) return 1;
    const int GQ = n_head / n_kv;
    if (GQ != 1 && GQ != 2 && GQ != 4 && GQ != 8) return 1;
    AttnPrefillArgs a;
    a.q = q, a.kcache = kc, a.vcache = vc, a.out = out, a.pos0 = pos0, a.n_tok = n_tok, a.n_kv = n_kv, a.kv_stride = kv_stride, a.q_stride = q_stride, a.out_stride = out_stride;
    a.rden = 1.0f / sqrtf((float)hd);
    a.n_seq = n_seq;
    const int TQ = 128 / GQ;
    dim3 grid((n_tok + TQ - 1) / TQ, n_kv, n_seq);
    const int nsb = (n_tok + TQ / 2 - 1) / (TQ / 2); /* the paired form (kh = 2): half blocks of TQ / 2 tokens, one from the front and one from the back per workgroup */
    // about one workgroup per CU or fewer: the launch lasts as long as its last query block -- two key halves per workgroup (2047 tokens, 16 / 8 heads x 128:
    // 81 -> 67 us); with more workgroups than that the halves only compete for the CU (8 x 1024 x 25 x 64: 130 vs 143 us; 4095 tokens: 15.9 vs 16.4 ms per prompt)
    const int kh = ((long)grid.x * grid.y * grid.z <= 320 && n_tok >= g_knobs.attn_pair_min) ? 2 : 1; /* short prompts: the launch is a few microseconds either way */
    if (kh == 2) grid.x = (nsb + 1) / 2;
    const int rc = hd == 128 ? ap_launch_gq<128>(st, a, GQ, grid, kh) : ap_launch_gq<64>(st, a, GQ, grid, kh);
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
#ifdef AP_STAMP
extern "C" int kfdbg_ap_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kf::g_ap_stamp), 64); }
#endif
