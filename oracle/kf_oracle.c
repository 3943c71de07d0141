/*
 * oracle/kf_oracle.c -- CPU restatement ("oracle") of Koifish's quantized transformer forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / reported baseline.  The product path
 * (koifish_amd/) never calls into oracle/ and fails loudly when its HIP library is missing.
 *
 * PARITY PINNING.  The reference ships no runnable CPU forward (DeepSeek.cpp:160-209 is a stub,
 * Generate.cu:351-353 commented out) and its sources do not compile here (g_float.hpp:210-243
 * hard-includes CUDA headers; no nvcc) -- see SURVEY.md section 8c.  What the reference's own tree
 * pins, and this oracle is checked against in tests/test_oracle_layout.py:
 *   - the PACK_/UNPACK_ macro bit positions of src/PackedQ.hpp:99-239 (restated below AND,
 *     independently, in numpy in tests/ -- the two must agree on random and on hand-built blocks),
 *   - BIT_SET_k/BIT_GET_k (src/Utils/CLI_params.cpp:2177-2207),
 *   - NF4/NF3 tables (src/g_float.hpp:542-569), AWQ nibble order (kernel/packedN.cuh:109-116).
 *   - outputs of the reference's own Python programs, executed in the build container and committed as vectors
 *     (tests/golden/make_ref_python_vectors.py -> tests/test_oracle_ref_python.py): AutoAWQ unpack / order / dequant
 *     (src/Python/test_awq.py), causal grouped-query attention (tile_wrapper/tl_qkv.py ref_program: score scale, mask,
 *     which kv head a query head reads, softmax, PV), RMS normalisation (tile_wrapper/tl_norm.py ref_program).
 * For the quantised mat-vec ARITHMETIC (and the bf16 rounding points of attention) the reference holds no golden vector,
 * fixture or known-answer test (cases/test_lite.py needs real weights + CUDA): **parity unpinned** there.
 * Those functions follow the cited CUDA kernels line by line with every bf16 store made
 * round-to-nearest-even, and are cross-checked against an independent numpy restatement and (model
 * semantics only) HF transformers' Qwen3 at tiny random shapes (tests/golden/make_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp; contraction must stay off so that
 * mul+add pairs written below stay two roundings, as _mm256_add_ps(_mm256_mul_ps()) does in
 * src/Utils/GST_float.cpp:75-101).
 */
#include "kfo_math.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <immintrin.h> /* the vectorised mat-vec of section 4b: AVX2 + F16C, as the reference's own CPU dot products (GST_float.cpp:75-130) */

#define KFO_API __attribute__((visibility("default")))

/* typNUMBER values restated from the NUMBER_DISPATCH order, src/g_float.hpp:84-117 */
enum {
    KFO_F32 = 0, KFO_F64, KFO_F16, KFO_BF16, KFO_F8E5M2, KFO_F8E4M3, KFO_U8, KFO_I8, KFO_U16, KFO_I16,
    KFO_U32, KFO_I32, KFO_U64, KFO_I64, KFO_Q4, KFO_Q3, KFO_Q2, KFO_T_SIGN, KFO_T_SEQ, KFO_BOOL1,
    KFO_T_BINARY, KFO_T_BINARY_3, KFO_T_BINARY_TILE,
    /* oracle-internal tag (not a typNUMBER): a Q4 tensor in the vendor AutoAWQ GEMM layout, see section 3b */
    KFO_Q4_AWQ = 100,
    /* oracle-internal tag: a Q4 tensor in the row-codebook storage of GeQuant::RT_NormalF (QUANT_MODE::RTNf / LUT), see section 3c */
    KFO_Q4_LUT = 101,
    KFO_Q3_LUT = 102, KFO_Q2_LUT = 103, /* 8- / 4-entry row codebooks over a 3- / 2-bit stream (CU_Q32X_NF3 / CU_Q32X_ / CU_Q22X_) */
    KFO_Q2_ROWRTN = 104                 /* CU_Q22X_RTN: (zero, step) per row */
};

/* ------------------------------------------------------------------------------------------------
 * 1. Packed128 bit layout -- src/PackedQ.hpp:28-60 (struct{u64 low; u64 high;}), little-endian host.
 *    bytes 0..7 = low, bytes 8..15 = high.
 * ---------------------------------------------------------------------------------------------- */
static inline void put_u64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
static inline uint64_t get_u64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

/* PACK_4to128_ (PackedQ.hpp:99-141): arr[0..15] -> high, MSB-first; arr[16..31] -> low */
KFO_API void kfo_pack4_128(const int32_t* q, uint8_t* dst) {
    uint64_t high = 0, low = 0;
    for (int i = 0; i < 16; i++) {
        high |= (uint64_t)(q[i] & 0x0F) << (60 - 4 * i);
        low |= (uint64_t)(q[i + 16] & 0x0F) << (60 - 4 * i);
    }
    put_u64(dst, low);
    put_u64(dst + 8, high);
}
/* UNPACK_128to4_UNSIGNED_ (PackedQ.hpp:143-183) */
KFO_API void kfo_unpack4_128(const uint8_t* src, int32_t* q) {
    uint64_t low = get_u64(src), high = get_u64(src + 8);
    for (int i = 0; i < 16; i++) {
        q[i]      = (int32_t)((high >> (60 - 4 * i)) & 0x0F);
        q[i + 16] = (int32_t)((low >> (60 - 4 * i)) & 0x0F);
    }
}
/* PACK_2to128_ (PackedQ.hpp:185-198) */
KFO_API void kfo_pack2_128(const int32_t* q, uint8_t* dst) {
    uint64_t high = 0, low = 0;
    for (int i = 0; i < 32; i++) {
        high |= (uint64_t)(q[i] & 0x3) << (62 - 2 * i);
        low |= (uint64_t)(q[i + 32] & 0x3) << (62 - 2 * i);
    }
    put_u64(dst, low);
    put_u64(dst + 8, high);
}
/* UNPACK_128to2_UNSIGNED_ (PackedQ.hpp:214-226) */
KFO_API void kfo_unpack2_128(const uint8_t* src, int32_t* q) {
    uint64_t low = get_u64(src), high = get_u64(src + 8);
    for (int i = 0; i < 32; i++) {
        q[i]      = (int32_t)((high >> (62 - 2 * i)) & 0x3);
        q[i + 32] = (int32_t)((low >> (62 - 2 * i)) & 0x3);
    }
}
/* PACK_1to128_ (PackedQ.hpp:200-212) */
KFO_API void kfo_pack1_128(const int32_t* q, uint8_t* dst) {
    uint64_t high = 0, low = 0;
    for (int i = 0; i < 64; i++) {
        high |= (uint64_t)(q[i] & 0x1) << (63 - i);
        low |= (uint64_t)(q[i + 64] & 0x1) << (63 - i);
    }
    put_u64(dst, low);
    put_u64(dst + 8, high);
}
/* UNPACK_128to1_UNSIGNED_ (PackedQ.hpp:227-239) */
KFO_API void kfo_unpack1_128(const uint8_t* src, int32_t* q) {
    uint64_t low = get_u64(src), high = get_u64(src + 8);
    for (int i = 0; i < 64; i++) {
        q[i]      = (int32_t)((high >> (63 - i)) & 0x1);
        q[i + 64] = (int32_t)((low >> (63 - i)) & 0x1);
    }
}

/* BIT_SET_k / BIT_GET_k (src/Utils/CLI_params.cpp:2177-2207): MSB-first bit stream; 2-bit values are
 * stored biased by +1 */
KFO_API void kfo_bit_set_k(uint8_t* array, size_t offset, int elem_, int bits) {
    int elem = elem_;
    if (bits == 2) elem = (uint8_t)(elem_ + 1);
    size_t boff = offset * (size_t)bits;
    for (int i = 0; i < bits; i++, boff++) {
        size_t id = boff / 8, shift = 7 - boff % 8;
        int bit = (elem >> (bits - 1 - i)) & 0x1;
        if (bit)
            array[id] |= (uint8_t)(1u << shift);
        else
            array[id] &= (uint8_t)~(1u << shift);
    }
}
KFO_API int kfo_bit_get_k(const uint8_t* array, size_t offset, int bits) {
    int elem = 0;
    size_t boff = offset * (size_t)bits;
    for (int i = 0; i < bits; i++, boff++) {
        size_t id = boff / 8, shift = 7 - boff % 8;
        if ((array[id] >> shift) & 0x1) elem |= 1 << (bits - 1 - i);
    }
    if (bits == 2) elem = (int8_t)(elem - 1);
    return elem;
}

static void pack_block(int bits, const int32_t* q, uint8_t* dst) {
    if (bits == 4)
        kfo_pack4_128(q, dst);
    else if (bits == 2)
        kfo_pack2_128(q, dst);
    else
        kfo_pack1_128(q, dst);
}
static void unpack_block(int bits, const uint8_t* src, int32_t* q) {
    if (bits == 4)
        kfo_unpack4_128(src, q);
    else if (bits == 2)
        kfo_unpack2_128(src, q);
    else
        kfo_unpack1_128(src, q);
}

/* ------------------------------------------------------------------------------------------------
 * 2. Quantiser -- GeQuant ctor (src/Tensor/GeQuant.cpp:107-124), RTN_x (:428-533), YinYang (:536-628)
 * ---------------------------------------------------------------------------------------------- */
/* yyang: 0 = I_OFF, else on.  Returns qMin/qMax/qBias exactly as the ctor sets them. */
KFO_API void kfo_quant_range(int bits, int isSymmetric, int yyang, int* qMin, int* qMax, int* qBias) {
    *qBias = 0;
    if (yyang) {
        if (bits == 2) {
            *qMax = 1, *qMin = -1, *qBias = 1; /* ternary */
        } else {
            *qMax = 1, *qMin = 0, *qBias = 0; /* 1-bit {0,1} */
        }
    } else if (isSymmetric) {
        *qMin = -(1 << (bits - 1));
        *qMax = (1 << (bits - 1)) - 1;
        *qBias = -*qMin;
    } else {
        *qMin = 0;
        *qMax = (1 << bits) - 1;
    }
}

/*
 * RTN_x (GeQuant.cpp:428-533).  w: bf16 [nGroup*lGroup] (row-major flattened weight, groups of
 * lGroup consecutive elements, GroupShapeOfT :375-404).  packed: nGroup*lGroup*bits/8 bytes.
 * zero/step: bf16 per group (gamaZero[row] = zero is an implicit float->bf16, :474).
 * yyang != 0 selects the "vMean" step of the RTN_x yyang branch (:463-465) -- the dedicated
 * YinYang() below is what the 1-bit / ternary cards actually call.
 * Returns err_2 = sqrt(mean((a - (step*q - zero))^2)) as the reference does (:518-530).
 */
KFO_API float kfo_rtn_x(const uint16_t* w, size_t nGroup, int lGroup, int bits, int isSymmetric, int yyang, uint8_t* packed,
                        uint16_t* zero_out, uint16_t* step_out) {
    int qMin, qMax, qBias;
    kfo_quant_range(bits, isSymmetric, yyang, &qMin, &qMax, &qBias);
    const int nPer128 = 128 / bits;
    double err_2 = 0;
#pragma omp parallel for schedule(static) reduction(+ : err_2)
    for (long row = 0; row < (long)nGroup; row++) {
        float vmax = -FLT_MAX, vmin = FLT_MAX, a;
        double vSum = 0.0;
        const uint16_t* dat = w + (size_t)row * lGroup;
        for (int i = 0; i < lGroup; i++) {
            a = kfo_bf16_to_f32(dat[i]); /* sR = sC = 1 (isSinkNormal=false, :441) */
            vmax = fmaxf(vmax, a), vmin = fminf(vmin, a);
            vSum += fabs(a);
        }
        float vMean = (float)(vSum / lGroup);
        float step = (vmax - vmin) / (float)(qMax - qMin), zero = -vmin;
        if (yyang) {
            step = fmaxf(1e-5f, vMean);
            zero = 0;
        } else if (isSymmetric) {
            step = fmaxf(fabsf(vmax), fabsf(vmin)) / (float)qMax, zero = 0;
        }
        zero_out[row] = kfo_f32_to_bf16(zero), step_out[row] = kfo_f32_to_bf16(step);
        uint8_t* quanti = packed + (size_t)lGroup * row * bits / 8;
        int32_t qq[128];
        for (int i = 0; i < lGroup / nPer128; i++) {
            for (int pos = 0; pos < nPer128; pos++) {
                a = kfo_bf16_to_f32(dat[pos + i * nPer128]);
                int qid = (int)roundf((a + zero) / step); /* std::round: half away from zero (:479) */
                if (yyang) {
                    qid = qid < qMin ? qMin : qid;
                    qid = qid > qMax ? qMax : qid;
                }
                /* reference: assert(qid >= qMin && qid <= qMax) (:485) -- clamp instead of aborting so a
                 * degenerate group cannot take the test process down; never taken on finite data */
                if (qid < qMin) qid = qMin;
                if (qid > qMax) qid = qMax;
                qq[pos] = qid + qBias;
                float e = a - (step * qid - zero);
                err_2 += (double)e * e;
            }
            pack_block(bits, qq, quanti + 16 * i);
        }
    }
    return (float)sqrt(err_2 / ((double)nGroup * lGroup));
}

/* YinYang (GeQuant.cpp:536-628): step = max(1e-5, sqrt(mean(relu(a)^2))), zero = 0 */
KFO_API float kfo_yinyang(const uint16_t* w, size_t nGroup, int lGroup, int bits, uint8_t* packed, uint16_t* zero_out,
                          uint16_t* step_out) {
    int qMin, qMax, qBias;
    kfo_quant_range(bits, 0, 1, &qMin, &qMax, &qBias);
    const int nPer128 = 128 / bits;
    double err_2 = 0;
#pragma omp parallel for schedule(static) reduction(+ : err_2)
    for (long row = 0; row < (long)nGroup; row++) {
        float a;
        double vSum = 0.0;
        const uint16_t* dat = w + (size_t)row * lGroup;
        for (int i = 0; i < lGroup; i++) {
            a = kfo_bf16_to_f32(dat[i]);
            vSum += a < 0.0 ? 0.0 : a * a; /* float product promoted to double, as the reference (:572) */
        }
        float vMean = (float)sqrt(vSum / lGroup);
        float step = fmaxf(1e-5f, vMean), zero = 0;
        zero_out[row] = kfo_f32_to_bf16(zero), step_out[row] = kfo_f32_to_bf16(step);
        uint8_t* quanti = packed + (size_t)lGroup * row * bits / 8;
        int32_t qq[128];
        for (int i = 0; i < lGroup / nPer128; i++) {
            for (int pos = 0; pos < nPer128; pos++) {
                a = kfo_bf16_to_f32(dat[pos + i * nPer128]);
                int qid = (int)roundf((a + zero) / step);
                qid = qid < qMin ? qMin : qid;
                qid = qid > qMax ? qMax : qid;
                qq[pos] = qid + qBias;
                float e = a - (step * qid - zero);
                err_2 += (double)e * e;
            }
            pack_block(bits, qq, quanti + 16 * i);
        }
    }
    return (float)sqrt(err_2 / ((double)nGroup * lGroup));
}

/* ------------------------------------------------------------------------------------------------
 * 3. Dequant -- CU_Q128toX_<bf16,32/64/128> (src/Device/CUDA/T.cu:245-294), bf16-stepwise:
 *    g0 = (step * (bf16)(q - qBias) - zero) * sR   with every operator a bf16 operator (floatGama=bf16):
 *    bf16(bf16(step*q') - zero), sR = 1 (rc_normal = 0).
 * ---------------------------------------------------------------------------------------------- */
static inline float dequant_one(float step, float zero, int qm) {
    float t = kfo_round_bf16(step * kfo_round_bf16((float)qm));
    return kfo_round_bf16(t - zero);
}

KFO_API void kfo_dequant_q128(const uint8_t* packed, const uint16_t* zero, const uint16_t* step, size_t nGroup, int lGroup, int bits,
                              int qBias, uint16_t* out) {
    const int nQuant = 128 / bits;
#pragma omp parallel for schedule(static)
    for (long g = 0; g < (long)nGroup; g++) {
        float z = kfo_bf16_to_f32(zero[g]), s = kfo_bf16_to_f32(step[g]);
        const uint8_t* q128 = packed + (size_t)g * lGroup * bits / 8;
        uint16_t* x0 = out + (size_t)g * lGroup;
        int32_t qq[128];
        for (int k = 0; k < lGroup / nQuant; k++, x0 += nQuant) {
            unpack_block(bits, q128 + 16 * k, qq);
            for (int i = 0; i < nQuant; i++) x0[i] = kfo_f32_to_bf16(dequant_one(s, z, qq[i] - qBias));
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * 3b. AutoAWQ GEMM format -- CU_Q42X_awq + CU_I2Q4_unpack (kernel/quantizer.cu:131-156, kernel/packedN.cuh:109-116;
 *     the same unpack is spelled in the reference's src/Python/test_awq.py:32-66 as shifts [0,4,..,28] re-ordered by
 *     AWQ_REVERSE_ORDER = {0,4,1,5,2,6,3,7}).  qweight int32 [in, out/8], qzeros int32 [in/128, out/8], scales fp16 [in/128, out];
 *     element k of a word sits at bits 4*ORDER[k]; W^T[i, o] = bf16( (q - z) * float(scale) )  (TransA = 0: the dequantised
 *     matrix is [in, out], GeQuant.cpp:989-992).
 * ---------------------------------------------------------------------------------------------- */
static const int KFO_AWQ_ORDER[8] = {0, 4, 1, 5, 2, 6, 3, 7};
static inline int awq_nibble(uint32_t word, int k) { return (int)((word >> (KFO_AWQ_ORDER[k] * 4)) & 0x0F); }
static inline float awq_weight(const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, int n_out, int i, int o) {
    const int g = i / 128, w8 = n_out / 8;
    const int q = awq_nibble(qweight[(size_t)i * w8 + o / 8], o % 8), z = awq_nibble(qzeros[(size_t)g * w8 + o / 8], o % 8);
    const float g0 = (float)(q - z) * kfo_half_to_f32(scales[(size_t)g * n_out + o]);
    return kfo_round_bf16(g0);
}
/* mat0 [in, out] exactly as the reference's GetDataX leaves it in tmpTernary */
KFO_API void kfo_dequant_awq(const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, int n_in, int n_out, uint16_t* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_in; i++)
        for (int o = 0; o < n_out; o++) out[(size_t)i * n_out + o] = kfo_f32_to_bf16(awq_weight(qweight, qzeros, scales, n_out, i, o));
}

/* CU_F82Float (kernel/operator.cuh:536-543): dst = T(float(f8e5m2)) */
KFO_API void kfo_f8e5m2_to_bf16(const uint8_t* src, size_t n, uint16_t* dst) {
    for (size_t i = 0; i < n; i++) dst[i] = kfo_f32_to_bf16(kfo_f8e5m2_to_f32(src[i]));
}
KFO_API void kfo_bf16_to_f8e5m2(const uint16_t* src, size_t n, uint8_t* dst) {
    for (size_t i = 0; i < n; i++) dst[i] = kfo_f32_to_f8e5m2(kfo_bf16_to_f32(src[i]));
}
KFO_API float kfo_expf_export(float x) { return kfo_expf(x); }

/* ------------------------------------------------------------------------------------------------
 * 3c. Row-codebook 4-bit storage -- GeQuant::RT_NormalF / _row_lut (src/Tensor/GeQuant.cpp:696-755), Distri_PIPE::Next / Prepare /
 *     X2NormalF (GTensor.hpp:141-148, GeQuant.cpp:641-694), unpacked on the device by CU_Q42X_NF4 / CU_Q42X_lut
 *     (kernel/quantizer.cu:583-652) and CU_embed_forw_q4 / _nf4 (kernel/embed.cuh:54-121).
 *     Per ROW (params.norm = NO_NORMAL: the only entry of LowBit_worker's sweep, GeQuant.cpp:844, so sR = sC = 1, rc_normal = 0):
 *       abs_max = max(|vmin|, |vmax|) over the row's fp32 values;  scale = abs_max > 0 ? (float)(1.0f / (double)abs_max) : 1
 *       (Prepare's `scale = 1.0f / abs_max` with abs_max a double member);  codebook[i] = table[i] / scale  (fp32 division);
 *       stored table gama_[i] = bf16(codebook[i]);  element -> the FIRST index of minimal |a - codebook[i]| (strict <, fp32 codebook);
 *       nibbles streamed by BIT_SET_k(quanti, i, qid, 4): element i in byte i/2, even i in the high nibble.
 *     Dequant: lut[id] * sR with bf16 operators and sR = bf16(1) -> the table entry itself.
 *     Pinned by the reference's own literals: NF4_LUT::table / NF3_LUT::table (src/g_float.hpp:543-569) and BIT_SET_k / BIT_GET_k.
 * ---------------------------------------------------------------------------------------------- */
static const float KFO_NF4[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f, -0.28444138169288635f, -0.18477343022823334f,
                                  -0.09105003625154495f, 0.0f, 0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
                                  0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};
static const float KFO_NF3[8] = {-1.0f, -0.5350227355957031f, -0.2469314038753510f, 0.0f, 0.1833375245332718f, 0.3819939494132996f, 0.6229856610298157f, 1.0f};
KFO_API const float* kfo_nf4_table(void) { return KFO_NF4; }
KFO_API const float* kfo_nf3_table(void) { return KFO_NF3; }
static inline int lut_nibble(const uint8_t* stream, size_t i) { return kfo_bit_get_k(stream, i, 4); }
/* element i of a `bits`-wide stream, most significant bit first, as CU_Q32X_* / CU_Q22X_* extract it (quantizer.cu:672-675, 727-731) */
static inline int raw_bits(const uint8_t* stream, size_t i, int bits) {
    int v = 0;
    size_t boff = i * (size_t)bits;
    for (int b = 0; b < bits; b++, boff++) v = (v << 1) | ((stream[boff / 8] >> (7 - boff % 8)) & 1);
    return v;
}

/* w bf16 [nRow, nCol] -> packed [nRow*nCol*bits/8], lut bf16 [nRow << bits]; bits = 4 (NF4) or 3 (NF3), as RT_NormalF asserts.
 * Returns disR.err summed over rows -> sqrt(err / nRow / nCol) as RT_NormalF does. */
KFO_API float kfo_lut_quantize_nf(const uint16_t* w, int nRow, int nCol, int bits, uint8_t* packed, uint16_t* lut) {
    const int nQuant = 1 << bits;
    const float* table = bits == 4 ? KFO_NF4 : KFO_NF3;
    double err = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : err)
    for (int row = 0; row < nRow; row++) {
        const uint16_t* dat = w + (size_t)row * nCol;
        float vmin = FLT_MAX, vmax = -FLT_MAX;
        for (int i = 0; i < nCol; i++) {
            const float a = kfo_bf16_to_f32(dat[i]); /* / sR / sC with both 1 */
            vmax = a > vmax ? a : vmax, vmin = a < vmin ? a : vmin;
        }
        const double abs_max = (double)fmaxf(fabsf(vmin), fabsf(vmax));
        const float scale = abs_max > 0 ? (float)(1.0f / abs_max) : 1.0f;
        float codebook[16];
        for (int i = 0; i < nQuant; i++) {
            codebook[i] = table[i] / scale;
            lut[(size_t)row * nQuant + i] = kfo_f32_to_bf16(codebook[i]);
        }
        double e2 = 0.0;
        for (int i = 0; i < nCol; i++) {
            const float a = kfo_bf16_to_f32(dat[i]);
            float min_dist = FLT_MAX;
            int best = 0;
            for (int k = 0; k < nQuant; k++) {
                const float dist = fabsf(a - codebook[k]);
                if (dist < min_dist) min_dist = dist, best = k;
            }
            const float e = fabsf(a - codebook[best]);
            e2 += (double)(e * e);
            kfo_bit_set_k(packed, (size_t)row * nCol + i, best, bits); /* rows are whole bytes (nCol % 8 == 0): threads never share one */
        }
        err += e2;
    }
    return (float)sqrt(err / nRow / nCol);
}
KFO_API float kfo_lut_quantize_nf4(const uint16_t* w, int nRow, int nCol, uint8_t* packed, uint16_t* lut) {
    return kfo_lut_quantize_nf(w, nRow, nCol, 4, packed, lut);
}

/* ------------------------------------------------------------------------------------------------
 * 4. Weight descriptor + W.x  -- SLP::Forw -> TASKA_AxB::blasLt (NeuronFuse.cu:305-381,
 *    GTensor.hpp:703-741): rhs[OC] = W[OC,IC] . lhs[IC], bf16 in, fp32 accumulate, bf16 out
 *    (gemm.cu:126 CUBLAS_COMPUTE_32F).  cuBLASLt's summation order is unspecified; the oracle uses the
 *    order of the reference's own CPU primitive dotprod_fp16 (src/Utils/GST_float.cpp:75-101):
 *    16 strided partial sums (two 8-lane accumulators), each updated as acc = acc + x*w (mul, then add),
 *    folded 16 -> 8 -> 4 -> (s0+s1)+(s2+s3) (_mm_dp_ps).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int type;             /* KFO_BF16 / KFO_F8E5M2 / KFO_Q4 / KFO_T_SIGN / KFO_BOOL1 / KFO_T_BINARY */
    int ne0, ne1;         /* out, in  (GTensor ne[0], ne[1]) */
    const void* data;     /* bf16 elements, f8 bytes, or Packed128 stream */
    const uint16_t* zero; /* gama_T(ZERO)  bf16[nGroup] */
    const uint16_t* step; /* gama_T(STEP)  bf16[nGroup] */
    int lGroup;           /* T_group (128) */
    int qBias;
} kfo_weight;

static int bits_of(int type) {
    switch (type) {
        case KFO_Q4: case KFO_Q4_AWQ: case KFO_Q4_LUT: return 4;
        case KFO_Q3_LUT: return 3;
        case KFO_T_SIGN: case KFO_Q2: case KFO_Q2_LUT: case KFO_Q2_ROWRTN: return 2;
        case KFO_BOOL1: case KFO_T_BINARY: return 1;
        case KFO_F8E5M2: return 8;
        default: return 16;
    }
}

/* dequantise one row of W to f32 (values are exactly bf16-representable) */
static void weight_row_f32(const kfo_weight* w, long r, float* out) {
    const int K = w->ne1;
    if (w->type == KFO_BF16) {
        const uint16_t* p = (const uint16_t*)w->data + (size_t)r * K;
        for (int c = 0; c < K; c++) out[c] = kfo_bf16_to_f32(p[c]);
    } else if (w->type == KFO_F16) { /* IEEE half weights (BASELINE config 1): half_to_float = _cvtsh_ss (GST_float.cpp:60-62), exact */
        const uint16_t* p = (const uint16_t*)w->data + (size_t)r * K;
        for (int c = 0; c < K; c++) out[c] = _cvtsh_ss(p[c]);
    } else if (w->type == KFO_F8E5M2) {
        const uint8_t* p = (const uint8_t*)w->data + (size_t)r * K;
        for (int c = 0; c < K; c++) out[c] = kfo_round_bf16(kfo_f8e5m2_to_f32(p[c]));
    } else if (w->type == KFO_Q3_LUT || w->type == KFO_Q2_LUT) { /* raw ids, most significant bit first (the kernels read them without BIT_GET_k's 2-bit un-biasing) */
        const int bits = w->type == KFO_Q3_LUT ? 3 : 2, nq = 1 << bits;
        for (int c = 0; c < K; c++) out[c] = kfo_bf16_to_f32(w->zero[(size_t)r * nq + raw_bits((const uint8_t*)w->data, (size_t)r * K + c, bits)]);
    } else if (w->type == KFO_Q2_ROWRTN) { /* CU_Q22X_RTN (quantizer.cu:655-688): (zero + step * (floatGama)id) * sR in bf16 operators, sR = 1 */
        const float zero = kfo_bf16_to_f32(w->zero[(size_t)r * 2]), step = kfo_bf16_to_f32(w->zero[(size_t)r * 2 + 1]);
        for (int c = 0; c < K; c++) out[c] = kfo_round_bf16(zero + kfo_round_bf16(step * (float)raw_bits((const uint8_t*)w->data, (size_t)r * K + c, 2)));
    } else if (w->type == KFO_Q4_LUT) { /* data = BIT_SET_k nibble stream, zero = the rows' 16-entry tables (bf16) */
        for (int c = 0; c < K; c++) out[c] = kfo_bf16_to_f32(w->zero[(size_t)r * 16 + lut_nibble((const uint8_t*)w->data, (size_t)r * K + c)]);
    } else if (w->type == KFO_Q4_AWQ) { /* logical W[out = ne0, in = ne1]; data = qweight, zero = qzeros, step = fp16 scales */
        for (int c = 0; c < K; c++)
            out[c] = awq_weight((const uint32_t*)w->data, (const uint32_t*)w->zero, w->step, w->ne0, c, (int)r);
    } else {
        /* groups of lGroup CONSECUTIVE elements of the flattened tensor (GeQuant.cpp:428-533): a row need not start on a group (GPT-2's
         * n_embd = 1600 gives 12.5 groups per row), so walk the 16-byte blocks that overlap [r*K, (r+1)*K) */
        const int bits = bits_of(w->type), lG = w->lGroup, nQuant = 128 / bits, nLevel = 1 << bits;
        const size_t e0 = (size_t)r * K, e1 = e0 + (size_t)K;
        int32_t qq[128];
        float lut[16];
        size_t g_cur = (size_t)-1;
        for (size_t bl = e0 / nQuant; bl * nQuant < e1; bl++) {
            const size_t eb = bl * nQuant, g = eb / lG; /* lGroup is a multiple of the block length: a block lies in one group */
            if (g != g_cur) {
                const float z = kfo_bf16_to_f32(w->zero[g]), s = kfo_bf16_to_f32(w->step[g]);
                for (int v = 0; v < nLevel; v++) lut[v] = dequant_one(s, z, v - w->qBias);
                g_cur = g;
            }
            unpack_block(bits, (const uint8_t*)w->data + bl * 16, qq);
            for (int i = 0; i < nQuant; i++) {
                const size_t e = eb + i;
                if (e >= e0 && e < e1) out[e - e0] = lut[qq[i]];
            }
        }
    }
}

static float dot16(const float* w, const float* x, int n) {
    float s[16];
    for (int i = 0; i < 16; i++) s[i] = 0.f;
    int j = 0;
    for (; j + 16 <= n; j += 16)
        for (int i = 0; i < 16; i++) s[i] = x[j + i] * w[j + i] + s[i];
    for (int i = 0; j < n; j++, i++) s[i] = x[j] * w[j] + s[i]; /* tail (reference asserts n%16==0) */
    float s8[8], s4[4];
    for (int i = 0; i < 8; i++) s8[i] = s[i] + s[i + 8];
    for (int i = 0; i < 4; i++) s4[i] = s8[i] + s8[i + 4];
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

/* ------------------------------------------------------------------------------------------------
 * 4c. The CANONICAL summation order of the mat-vec (round 3).  cuBLASLt's order is unspecified (gemm.cu:126), so any fixed fp32 order is
 *     a faithful SLP::Forw; sections 4 / 4b use the order of the reference's CPU primitive (dotprod_fp16).  The HIP kernels cannot follow
 *     that one cheaply, and their v_dot2c_f32_bf16 has no bit-exact CPU model -- so kernels and oracle share THIS order instead, built
 *     from single fp32 fused multiply-adds only (v_fma_f32 on the GPU, fmaf here), which makes every logit and every greedy id equal
 *     bit for bit (tests/test_gpu_canonical.py; bench.py cpu_baseline.mismatches_* = 0):
 *       - a row is cut into 16-byte storage blocks of EPB elements (8 bf16 / 16 f8 / 32 four-bit / 64 two-bit / 128 one-bit);
 *       - LPR = 2^lpr_log2 "lanes" walk the row: lane l owns blocks l, l + LPR, l + 2 LPR, ... and keeps TWO accumulators through all of
 *         them (round 4; one before): e = fmaf(w[i], x[i], e) over the even-indexed elements of a block in index order, o = fmaf(w[i], x[i], o)
 *         over the odd-indexed ones -- the two halves of one v_pk_fma_f32 per weight pair on the GPU, half the dependent chain of the
 *         single-accumulator form -- and the lane's value is e + o;
 *       - 1-bit and 2-bit storage (EPB = 128 / 64; round 4): a block is cut into 32-element sub-blocks (the four dwords of a 1-bit block, the two dword pairs of a 2-bit
 *         one) and every sub-block position s keeps its OWN (e_s, o_s) pair through all of the lane's blocks; the lane's value is ((e0 + o0) + (e1 + o1)) + ((e2 + o2) +
 *         (e3 + o3)), resp. (e0 + o0) + (e1 + o1).  The sub-chains are what neighbouring hardware lanes compute side by side in the persistent engine (a fraction of the
 *         dependent chain each); the mat-vec kernel keeps them as register pairs.  The lanes-per-row figure of such a row is the rule's value for K / 32 "virtual" blocks
 *         divided by the sub-blocks per block (kfo_lpr_log2_epb);
 *       - the LPR lane values are added by a balanced binary tree (lanes 2j + 2j+1, then pairs of pairs, ...).
 *     lpr_log2 = kfo_lpr_log2(blocks per row, rows of the launch): the rule of kf::gemv_lpr_log2 (koifish_amd/csrc/kf_gemv.hip), restated.
 *     `rows` = the rows of ALL matrices a launch multiplies (Q | K | V together; gate alone for the paired gate / up launch).
 * ---------------------------------------------------------------------------------------------- */
static int g_order = 0; /* 0: the dot16 order of sections 4 / 4b; 1: canonical */
KFO_API void kfo_set_order(int o) { g_order = o; }
KFO_API int kfo_get_order(void) { return g_order; }
KFO_API int kfo_lpr_log2(int nBlk, long rows) {
    int l = 6;
    while (l > 0 && (nBlk % (1 << l)) != 0) l--;
    if ((1 << l) < 16) {
        l = 6;
        while ((1 << l) > nBlk) l--;
    }
    while (l < 6 && nBlk > (1 << l) && (rows << l) / 64 < 1024 && (nBlk + (2 << l) - 1) / (2 << l) < (nBlk + (1 << l) - 1) / (1 << l)) l++;
    return l;
}
static int epb_of_type(int type) { /* elements per 16-byte storage block; 0: the canonical order is not defined for this storage */
    switch (type) {
        case KFO_BF16: return 8;
        case KFO_F8E5M2: return 16;
        case KFO_Q4: case KFO_Q4_LUT: return 32;
        case KFO_T_SIGN: case KFO_Q2: return 64;
        case KFO_BOOL1: case KFO_T_BINARY: return 128;
        default: return 0;
    }
}
/* lanes per row for a storage with `epb` elements per block: the rule itself, except for the 1-bit blocks (see above) */
static int kfo_lpr_log2_epb(int epb, int K, long rows) {
    if (epb == 128 || epb == 64) {
        const int l = kfo_lpr_log2(K / 32, rows) - (epb == 128 ? 2 : 1);
        return l > 0 ? l : 0;
    }
    return kfo_lpr_log2(K / epb, rows);
}
static float dot_canon(const float* w, const float* x, int K, int epb, int lpr_log2) {
    const int nBlk = K / epb, LPR = 1 << lpr_log2, ns = epb == 128 ? 4 : (epb == 64 ? 2 : 1), sb = epb / ns; /* sub-chains per lane, elements per sub-block */
    float lane[64];
    for (int l = 0; l < LPR; l++) {
        float e[4] = {0.f, 0.f, 0.f, 0.f}, o[4] = {0.f, 0.f, 0.f, 0.f}; /* the even / odd chains of this lane (every sub-block length is even) */
        for (int c = l; c < nBlk; c += LPR) {
            const float *wb = w + (size_t)c * epb, *xb = x + (size_t)c * epb;
            for (int s = 0; s < ns; s++)
                for (int i = s * sb; i < (s + 1) * sb; i += 2) e[s] = fmaf(wb[i], xb[i], e[s]), o[s] = fmaf(wb[i + 1], xb[i + 1], o[s]);
        }
        lane[l] = ns == 1 ? e[0] + o[0] : (ns == 2 ? (e[0] + o[0]) + (e[1] + o[1]) : ((e[0] + o[0]) + (e[1] + o[1])) + ((e[2] + o[2]) + (e[3] + o[3])));
    }
    for (int s = 1; s < LPR; s <<= 1)
        for (int l = 0; l < LPR; l += 2 * s) lane[l] = lane[l] + lane[l + s];
    return lane[0];
}
/* row dot in the order the library is set to; `rows` only matters for the canonical order */
static float row_dot(const kfo_weight* w, const float* row, const float* xf, int c0, int c1, long rows) {
    const int epb = epb_of_type(w->type);
    if (g_order == 1 && epb > 0 && (c1 - c0) % epb == 0) return dot_canon(row + c0, xf + c0, c1 - c0, epb, kfo_lpr_log2_epb(epb, c1 - c0, rows));
    return dot16(row + c0, xf + c0, c1 - c0);
}
static long g_launch_rows = 0; /* rows of the launch the next kfo_linear* calls belong to (0: the matrix's own rows) */
KFO_API void kfo_set_launch_rows(long rows) { g_launch_rows = rows; }

/* y[r] = bf16( alpha * W[r,:].x  (+ bias[r]) (+ beta*y[r]) ),  rows [r0,r1) only (TP shards use it) */
KFO_API void kfo_linear_rows(const kfo_weight* w, const uint16_t* x, uint16_t* y, const uint16_t* bias, float alpha, float beta, int r0,
                             int r1) {
    const int K = w->ne1;
    const long lrows = g_launch_rows > 0 ? g_launch_rows : w->ne0;
    float* xf = (float*)malloc(sizeof(float) * K);
    for (int c = 0; c < K; c++) xf[c] = kfo_bf16_to_f32(x[c]);
#pragma omp parallel
    {
        float* row = (float*)malloc(sizeof(float) * K);
#pragma omp for schedule(static)
        for (long r = r0; r < r1; r++) {
            weight_row_f32(w, r, row);
            float v = row_dot(w, row, xf, 0, K, lrows);
            if (alpha != 1.0f) v = alpha * v;
            if (beta != 0.0f) v = v + beta * kfo_bf16_to_f32(y[r]);
            if (bias) v = v + kfo_bf16_to_f32(bias[r]);
            y[r] = kfo_f32_to_bf16(v);
        }
        free(row);
    }
    free(xf);
}
KFO_API void kfo_linear(const kfo_weight* w, const uint16_t* x, uint16_t* y, const uint16_t* bias, float alpha, float beta) {
    kfo_linear_rows(w, x, y, bias, alpha, beta, 0, w->ne0);
}
/* D_matmul_sparse (src/Utils/GST_float.cpp:306-318): row i is computed only when hot[i] == 1 (CS_Picker's array, SparseNeuron.cpp:20-29);
 * a cold row is val = 0, then the bias.  Hot rows: exactly kfo_linear's rows. */
KFO_API void kfo_linear_masked(const kfo_weight* w, const uint16_t* x, uint16_t* y, const uint16_t* bias, const int32_t* hot) {
    const int K = w->ne1;
    float* xf = (float*)malloc(sizeof(float) * K);
    for (int c = 0; c < K; c++) xf[c] = kfo_bf16_to_f32(x[c]);
#pragma omp parallel
    {
        float* row = (float*)malloc(sizeof(float) * K);
#pragma omp for schedule(dynamic, 16)
        for (long r = 0; r < w->ne0; r++) {
            float v = 0.f;
            if (hot[r] == 1) {
                weight_row_f32(w, r, row);
                v = row_dot(w, row, xf, 0, K, g_launch_rows > 0 ? g_launch_rows : w->ne0);
            }
            if (bias) v = v + kfo_bf16_to_f32(bias[r]);
            y[r] = kfo_f32_to_bf16(v);
        }
        free(row);
    }
    free(xf);
}
/* fp32 (un-rounded) row dots, for the TP partial-sum path and tolerance analysis */
KFO_API void kfo_linear_f32(const kfo_weight* w, const uint16_t* x, float* y, int c0, int c1) {
    const int K = w->ne1;
    float* xf = (float*)calloc(K, sizeof(float));
    for (int c = c0; c < c1; c++) xf[c] = kfo_bf16_to_f32(x[c]);
#pragma omp parallel
    {
        float* row = (float*)malloc(sizeof(float) * K);
#pragma omp for schedule(static)
        for (long r = 0; r < w->ne0; r++) {
            weight_row_f32(w, r, row);
            y[r] = row_dot(w, row, xf, c0, c1, g_launch_rows > 0 ? g_launch_rows : w->ne0);
        }
        free(row);
    }
    free(xf);
}
KFO_API void kfo_dequant_weight(const kfo_weight* w, uint16_t* out) {
    const int K = w->ne1;
#pragma omp parallel
    {
        float* row = (float*)malloc(sizeof(float) * K);
#pragma omp for schedule(static)
        for (long r = 0; r < w->ne0; r++) {
            weight_row_f32(w, r, row);
            for (int c = 0; c < K; c++) out[(size_t)r * K + c] = kfo_f32_to_bf16(row[c]);
        }
        free(row);
    }
}

/* ------------------------------------------------------------------------------------------------
 * 4b. Vectorised mat-vec: the timed CPU baseline (bench.py cpu_baseline) and BASELINE config 1 (fp16 weights).
 *     Structure of the reference's CPU dot products (dotprod_fp16 / dotprod_fp8, src/Utils/GST_float.cpp:75-130): two 8-lane accumulators over
 *     16 consecutive elements per step, acc = x * w + acc as a multiply and an add (no fma), then acc0 + acc1, low + high 128 bits, and the
 *     4-lane dot with ones -- (s0 + s1) + (s2 + s3).  That is dot16() above lane for lane, so every result bit equals the scalar path's
 *     (tests/test_oracle_fast.py).  Rows over OpenMP threads like D_matvec (GST_float.cpp:293-304).
 *     Weights are read as 16-bit values: IEEE half directly (_mm256_cvtph_ps, the reference's own conversion) or bf16 (shifted into the high
 *     half of an fp32); quantised tensors are dequantised ONCE into a bf16 copy (kfo_qwen3_prepare_fast) -- what GetDataX produces.
 * ---------------------------------------------------------------------------------------------- */
static inline __m256 load8_w16(const uint16_t* p, int is_f16) {
    const __m128i h = _mm_loadu_si128((const __m128i*)p);
    if (is_f16) return _mm256_cvtph_ps(h);
    return _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_cvtepu16_epi32(h), 16));
}
static inline float dot16_w16(const uint16_t* w, const float* x, int n, int is_f16) { /* n a multiple of 16 */
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    for (int j = 0; j < n; j += 16) {
        acc0 = _mm256_add_ps(_mm256_mul_ps(_mm256_loadu_ps(x + j), load8_w16(w + j, is_f16)), acc0);
        acc1 = _mm256_add_ps(_mm256_mul_ps(_mm256_loadu_ps(x + j + 8), load8_w16(w + j + 8, is_f16)), acc1);
    }
    const __m256 acc8 = _mm256_add_ps(acc0, acc1);
    const __m128 acc4 = _mm_add_ps(_mm256_castps256_ps128(acc8), _mm256_extractf128_ps(acc8, 1));
    float s4[4];
    _mm_storeu_ps(s4, acc4);
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}
/* y[r] = bf16(W[r,:].x) for rows [0, M) of a 16-bit weight copy; hot != NULL: D_matmul_sparse (cold rows 0) */
static void linear_w16(const uint16_t* W, int is_f16, int M, int K, const uint16_t* x, uint16_t* y, const int32_t* hot) {
    float* xf = (float*)malloc(sizeof(float) * K);
    for (int c = 0; c < K; c++) xf[c] = kfo_bf16_to_f32(x[c]);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < M; r++) {
        float v = 0.f;
        if (!hot || hot[r] == 1) v = dot16_w16(W + (size_t)r * K, xf, K, is_f16);
        y[r] = kfo_f32_to_bf16(v);
    }
    free(xf);
}
KFO_API void kfo_linear_w16(const uint16_t* W, int is_f16, int M, int K, const uint16_t* x, uint16_t* y) { linear_w16(W, is_f16, M, K, x, y, NULL); }
/* the same 16-bit view multiplied in the canonical order of section 4c (epb = the blocks of the tensor's OWN storage, not of the bf16 view) */
static void linear_canon_w16(const uint16_t* W, int M, int K, int epb, int lpr_log2, const uint16_t* x, uint16_t* y, const int32_t* hot) {
    float* xf = (float*)malloc(sizeof(float) * K);
    for (int c = 0; c < K; c++) xf[c] = kfo_bf16_to_f32(x[c]);
#pragma omp parallel
    {
        float* row = (float*)malloc(sizeof(float) * K);
#pragma omp for schedule(static)
        for (long r = 0; r < M; r++) {
            float v = 0.f;
            if (!hot || hot[r] == 1) {
                const uint16_t* p = W + (size_t)r * K;
                for (int c = 0; c < K; c++) row[c] = kfo_bf16_to_f32(p[c]);
                v = dot_canon(row, xf, K, epb, lpr_log2);
            }
            y[r] = kfo_f32_to_bf16(v);
        }
        free(row);
    }
    free(xf);
}

/* ------------------------------------------------------------------------------------------------
 * 5. Small ops
 * ---------------------------------------------------------------------------------------------- */
/* rms_norm_kernel (kernel/layernorm.cuh:800-847): y = bf16((x * rsqrt(fma(sum, 1/D, eps))) * w).
 * Deliberate: the sum of squares is accumulated in fp64 (reference: fp32 in its launch-geometry order)
 * so that the result does not depend on the reduction order -- the HIP kernel does the same. */
KFO_API void kfo_rmsnorm(const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int dim, float eps) {
    for (int r = 0; r < rows; r++) {
        const uint16_t* xr = x + (size_t)r * dim;
        double acc = 0.0;
        for (int i = 0; i < dim; i++) {
            double a = kfo_bf16_to_f32(xr[i]);
            acc += a * a;
        }
        float val = fmaf((float)acc, 1.0f / (float)dim, eps);
        float mul = 1.0f / sqrtf(val);
        for (int i = 0; i < dim; i++) {
            float v = (kfo_bf16_to_f32(xr[i]) * mul) * kfo_bf16_to_f32(w[i]);
            y[(size_t)r * dim + i] = kfo_f32_to_bf16(v);
        }
    }
}

/* q/k-norm: CU_rms_forward_v2 (layernorm.cuh:129-167): s0 = rsqrt(sum/ld + eps); s = bf16(s0);
 * out = bf16(a * float(s) * w) (Hadamard, packedN.cuh:412-419; RN instead of its stochastic store). */
KFO_API void kfo_headnorm(uint16_t* x, const uint16_t* w, int nHead, int hd, float eps) {
    for (int h = 0; h < nHead; h++) {
        uint16_t* xh = x + (size_t)h * hd;
        double acc = 0.0;
        for (int i = 0; i < hd; i++) {
            double a = kfo_bf16_to_f32(xh[i]);
            acc += a * a;
        }
        float s0 = 1.0f / sqrtf((float)acc / (float)hd + eps);
        float s = kfo_round_bf16(s0);
        for (int i = 0; i < hd; i++) {
            float res = kfo_bf16_to_f32(xh[i]) * s * kfo_bf16_to_f32(w[i]);
            xh[i] = kfo_f32_to_bf16(res);
        }
    }
}

/* RoPE table entry, CU_rope2_v0 (kernel/operator.cuh:734-772): inv_freq = 1/powf(theta, 2j/hd),
 * angle = pos*inv_freq, sincosf.  Host libm. */
KFO_API void kfo_rope_table(int pos, int hd, float theta, float* cos_out, float* sin_out) {
    for (int j = 0; j < hd / 2; j++) {
        float inv_freq = 1.0f / powf(theta, (float)(j * 2) / (float)hd);
        float angle = (float)pos * inv_freq;
        sin_out[j] = sinf(angle);
        cos_out[j] = cosf(angle);
    }
}
/* rotate-half pairs (j, j+hd/2), in place, RN bf16 stores */
KFO_API void kfo_rope(uint16_t* x, int nHead, int hd, int pos, float theta) {
    float cs[512], sn[512];
    kfo_rope_table(pos, hd, theta, cs, sn);
    for (int h = 0; h < nHead; h++) {
        uint16_t* xh = x + (size_t)h * hd;
        for (int j = 0; j < hd / 2; j++) {
            float re = kfo_bf16_to_f32(xh[j]), im = kfo_bf16_to_f32(xh[j + hd / 2]);
            float a = re * cs[j], b = im * sn[j], c = re * sn[j], d = im * cs[j];
            xh[j] = kfo_f32_to_bf16(a - b);
            xh[j + hd / 2] = kfo_f32_to_bf16(c + d);
        }
    }
}

/* CU_swiglu_v0 (Activation.cu:85-93): out = bf16((g*u)/(1+expf(-g))) */
KFO_API void kfo_swiglu(const uint16_t* gate, const uint16_t* up, uint16_t* out, int n) {
    for (int i = 0; i < n; i++) {
        float g = kfo_bf16_to_f32(gate[i]), u = kfo_bf16_to_f32(up[i]);
        out[i] = kfo_f32_to_bf16((g * u) / (1.0f + kfo_expf(-g)));
    }
}
/* CU_add3 / Add2 (packedN.cuh:446-453,866-875): out = bf16(a + b) (RN instead of stochastic) */
KFO_API void kfo_add(const uint16_t* a, const uint16_t* b, uint16_t* out, int n) {
    for (int i = 0; i < n; i++) out[i] = kfo_f32_to_bf16(kfo_bf16_to_f32(a[i]) + kfo_bf16_to_f32(b[i]));
}
/* CU_embed_forw_1 (embed.cuh:123-132): one-row gather (+dequant for quantised tables) */
KFO_API void kfo_embed(const kfo_weight* w, int token, uint16_t* out) {
    float* row = (float*)malloc(sizeof(float) * w->ne1);
    weight_row_f32(w, token, row);
    for (int c = 0; c < w->ne1; c++) out[c] = kfo_f32_to_bf16(row[c]);
    free(row);
}
/* sample_argmax (src/Manifold/GoPT.cpp:602-612): first index of the maximum */
KFO_API int kfo_argmax_bf16(const uint16_t* logits, int n) {
    int best = 0;
    float bv = kfo_bf16_to_f32(logits[0]);
    for (int i = 1; i < n; i++) {
        float v = kfo_bf16_to_f32(logits[i]);
        if (v > bv) bv = v, best = i;
    }
    return best;
}

/* GeneratOnPrompt::Sample, non-greedy branch (src/Manifold/GoPT.cpp:614-630), restated step by step:
 *   LogitsInfo::TopK -> TOPK_heap::Select (GoPT.cpp:666-704): `heap` is a std::priority_queue<int> with the DEFAULT ordering, i.e. it is
 *     ordered by token index, not by logit: heap.top() is the largest index pushed so far.  The loop therefore keeps indices
 *     0 .. k-2 for good and only ever replaces the most recent entry: the candidate set is {0, .., k-2} plus the first maximum
 *     over i >= k-1.  This is what the reference computes, so it is what is restated here (not a true top-k).
 *     Extraction pops in descending index order; ver == 1 then sorts by logit, descending, with std::sort -- whose order among
 *     EQUAL logits is unspecified: restated as a stable insertion sort of the extraction order (= libstdc++ for k <= 16).
 *   UpdateLogits (GoPT.cpp:754-769): p_i = expf((a_i - maxLogit) / temperature), then p_i /= sum (sequential fp32 sum).
 *   TopP (GoPT.cpp:729-751): nPick = 1 + first i with cumulative p > top_p; for top_p >= 1 the reference returns before
 *     setting nPick (it then reads picks[-2]): restated with the evident intent nPick = k.
 *   Qu_FlipCoin (GoPT.cpp:771-790) with random_f32 / random_u32 (GoPT.cpp:594-600; xorshift64*).
 * expf: the reference calls libm; here kfo_expf (<= 2 ulp from it, tests/test_oracle_math.py) so that the HIP kernel can agree bit
 * for bit.  Returns the token id, or -1 for arguments the reference asserts against (k < 2, k >= n/2, temperature <= 0). */
static inline uint32_t kfo_random_u32(uint64_t* state) {
    *state ^= *state >> 12;
    *state ^= *state << 25;
    *state ^= *state >> 27;
    return (uint32_t)((*state * 0x2545F4914F6CDD1Dull) >> 32);
}
KFO_API float kfo_random_f32(uint64_t* state) { return (float)(kfo_random_u32(state) >> 8) / 16777216.0f; }

/* candidates `picks` (already ordered) -> UpdateLogits, TopP, Qu_FlipCoin */
static int sample_tail(const uint16_t* logits, int* picks, int k, float temperature, float top_p, uint64_t* rng_state, float* p, int* npick_out) {
    float maxLogit = -3.402823466e+38f;
    for (int j = 0; j < k; j++) {
        const float a = kfo_bf16_to_f32(logits[picks[j]]);
        if (a > maxLogit) maxLogit = a;
    }
    float prob_sum = 0.0f;
    for (int j = 0; j < k; j++) {
        const float a = kfo_bf16_to_f32(logits[picks[j]]);
        p[j] = kfo_expf((a - maxLogit) / temperature);
        prob_sum += p[j];
    }
    for (int j = 0; j < k; j++) p[j] /= prob_sum;
    int nPick = k;
    if (top_p < 1.0f) {
        float cum = 0.0f;
        int last_idx = k - 1;
        for (int j = 0; j < k; j++) {
            cum += p[j];
            if (cum > top_p) {
                last_idx = j;
                break;
            }
        }
        nPick = last_idx + 1;
    }
    float ps = 0.0f;
    for (int j = 0; j < nPick; j++) ps += p[j];
    const float coin = kfo_random_f32(rng_state) * ps;
    float cdf = 0.0f;
    int qu = picks[nPick - 1];
    for (int j = 0; j < nPick; j++) {
        cdf += p[j];
        if (coin < cdf) {
            qu = picks[j];
            break;
        }
    }
    if (npick_out) *npick_out = nPick;
    return qu;
}

KFO_API int kfo_sample(const uint16_t* logits, int n, int top_k, float temperature, float top_p, uint64_t* rng_state, int* picks_out, float* probs_out,
                       int* npick_out) {
    const int k = top_k < n ? top_k : n;
    if (k < 2 || k >= n / 2 || !(temperature > 0.0f) || !(top_p > 0.0f)) return -1;
    int* picks = (int*)malloc(sizeof(int) * k);
    float* p = (float*)malloc(sizeof(float) * k);
    /* Select */
    int last = k - 1;
    for (int i = k; i < n; i++)
        if (kfo_bf16_to_f32(logits[i]) > kfo_bf16_to_f32(logits[last])) last = i;
    picks[0] = last;
    for (int j = 1; j < k; j++) picks[j] = k - 1 - j; /* k-2, k-3, .., 0 */
    for (int j = 1; j < k; j++) { /* stable insertion sort, descending by logit */
        const int pj = picks[j];
        const float vj = kfo_bf16_to_f32(logits[pj]);
        int q = j - 1;
        while (q >= 0 && vj > kfo_bf16_to_f32(logits[picks[q]])) picks[q + 1] = picks[q], q--;
        picks[q + 1] = pj;
    }
    const int qu = sample_tail(logits, picks, k, temperature, top_p, rng_state, p, npick_out);
    if (picks_out) memcpy(picks_out, picks, sizeof(int) * k);
    if (probs_out) memcpy(probs_out, p, sizeof(float) * k);
    free(picks);
    free(p);
    return qu;
}

/* The candidate set TopK is evidently meant to keep (not what the reference's heap keeps, see above): the k largest logits, equal logits
 * towards the lower token index, ordered by (logit descending, index ascending); then the same UpdateLogits / TopP / Qu_FlipCoin. */
KFO_API int kfo_sample_topk(const uint16_t* logits, int n, int top_k, float temperature, float top_p, uint64_t* rng_state, int* picks_out, float* probs_out,
                            int* npick_out) {
    const int k = top_k < n ? top_k : n;
    if (k < 2 || k >= n / 2 || !(temperature > 0.0f) || !(top_p > 0.0f)) return -1;
    int* picks = (int*)malloc(sizeof(int) * (k + 1));
    float* p = (float*)malloc(sizeof(float) * k);
    int cnt = 0;
    for (int i = 0; i < n; i++) { /* insertion into the running top-k list; a later equal logit never displaces an earlier one */
        const float v = kfo_bf16_to_f32(logits[i]);
        if (cnt == k && !(v > kfo_bf16_to_f32(logits[picks[k - 1]]))) continue;
        int q = cnt < k ? cnt : k - 1;
        while (q > 0 && v > kfo_bf16_to_f32(logits[picks[q - 1]])) picks[q] = picks[q - 1], q--;
        picks[q] = i;
        if (cnt < k) cnt++;
    }
    const int qu = sample_tail(logits, picks, k, temperature, top_p, rng_state, p, npick_out);
    if (picks_out) memcpy(picks_out, picks, sizeof(int) * k);
    if (probs_out) memcpy(probs_out, p, sizeof(float) * k);
    free(picks);
    free(p);
    return qu;
}

/* LayerNorm forward, CU_lm_forward (src/Device/CUDA/kernel/layernorm.cuh:226-300; GPT-2 family): m = sum(x)/C, v = sum((x-m)^2)/C,
 * s = rsqrtf(v + eps), out = bf16(s*(x-m)*w + b).  Restated with the sums in fp64 (the reference adds in warp order; fp64 makes the result
 * independent of any order, as for RMSNorm), s = 1/sqrtf(v + eps) with IEEE ops in place of the approximate rsqrtf, and the scale-and-shift
 * as a multiply then an add (no contraction).  mean / rstd (fp32 per row) are the values the backward pass caches; either may be NULL. */
KFO_API void kfo_layernorm(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* y, int rows, int C, float eps, float* mean, float* rstd) {
    for (int r = 0; r < rows; r++) {
        const uint16_t* xr = x + (size_t)r * C;
        double sum = 0.0;
        for (int c = 0; c < C; c++) sum += (double)kfo_bf16_to_f32(xr[c]);
        const float m = (float)sum / (float)C;
        double sq = 0.0;
        for (int c = 0; c < C; c++) {
            const float d = kfo_bf16_to_f32(xr[c]) - m;
            sq += (double)d * (double)d;
        }
        const float v = (float)sq / (float)C;
        const float s = 1.0f / sqrtf(v + eps);
        for (int c = 0; c < C; c++) {
            const float n = s * (kfo_bf16_to_f32(xr[c]) - m);
            const float o = n * kfo_bf16_to_f32(w[c]) + (b ? kfo_bf16_to_f32(b[c]) : 0.0f);
            y[(size_t)r * C + c] = kfo_f32_to_bf16(o);
        }
        if (mean) mean[r] = m;
        if (rstd) rstd[r] = s;
    }
}
/* Fused classifier (src/Device/CUDA/kernel/fused_classifier.cuh:68-140; prepare_softmax_blockwide3 :21-62; blockReduce_v0 utils.cuh:235-270), restated
 * thread by thread: 1024 threads per row; thread t visits the 8-element vectors i = ceil(V/8) + t - 1024, i - 1024, ... >= 0 keeping a running
 * (max, sum) -- on a new maximum sum *= exp(old - new), then sum += exp(v - max); the reference also multiplies by exp(0) = 1 when the maximum
 * stays, which changes nothing --; block max; sum_t *= exp(max_t - max); block sum; both reductions = xor-butterfly (16, 8, 4, 2, 1) inside
 * 32-thread groups, then the same butterfly over the 32 group results.  losses[row] -= log(exp(logit[target] - max) * (1/sum)); every element:
 * prob = exp(logit - max) * (1/sum), dlogit = bf16((prob - onehot) * dloss) over the logit when write_dlogits, probs = bf16(prob) when given.
 * Rows with mask bit 0x10000 (F_IGNORE_LOSS, DataLoader.hpp:78) are skipped.  exp / log = kfo_expf / kfo_logf (the reference: CUDA expf / logf).
 * The reference's V % 8 tail loop advances by 1 per thread (:124: overlapping rewrites when V % 8 >= 2); here each tail element is handled once. */
static float kfo_butterfly32(float* v, int is_max) { /* v[32]; returns lane 0's value (all lanes end equal: fp add and max commute) */
    for (int off = 16; off > 0; off >>= 1) {
        float t[32];
        for (int l = 0; l < 32; l++) t[l] = is_max ? fmaxf(v[l], v[l ^ off]) : v[l] + v[l ^ off];
        memcpy(v, t, sizeof(t));
    }
    return v[0];
}
static float kfo_block_reduce_1024(const float* val, int is_max) {
    float grp[32];
    for (int g = 0; g < 32; g++) {
        float v[32];
        memcpy(v, val + 32 * g, sizeof(v));
        grp[g] = kfo_butterfly32(v, is_max);
    }
    return kfo_butterfly32(grp, is_max);
}
KFO_API void kfo_fused_classifier(uint16_t* logits, float* losses, uint16_t* probs, float dloss, const int32_t* targets, long rows, int V, int P,
                                  const int32_t* mask, int write_dlogits) {
    float* tmax = (float*)malloc(1024 * sizeof(float));
    float* tsum = (float*)malloc(1024 * sizeof(float));
    for (long idx = 0; idx < rows; idx++) {
        if (mask && (mask[idx] & 0x10000)) continue;
        uint16_t* row = logits + idx * (long)P;
        const int ix = targets[idx];
        for (int t = 0; t < 1024; t++) {
            float mx = -INFINITY, sm = 0.0f;
            for (int i = (V + 7) / 8 + t - 1024; i >= 0; i -= 1024)
                for (int k = 0; k < 8 && i * 8 + k < V; k++) {
                    const float v = kfo_bf16_to_f32(row[i * 8 + k]);
                    if (v > mx) {
                        if (mx != -INFINITY) sm *= kfo_expf(mx - v);
                        mx = v;
                    }
                    sm += kfo_expf(v - mx);
                }
            tmax[t] = mx, tsum[t] = sm;
        }
        const float bmax = kfo_block_reduce_1024(tmax, 1);
        for (int t = 0; t < 1024; t++) tsum[t] *= kfo_expf(tmax[t] - bmax);
        const float bsum = kfo_block_reduce_1024(tsum, 0);
        const float scale = 1.0f / bsum;
        losses[idx] -= kfo_logf(kfo_expf(kfo_bf16_to_f32(row[ix]) - bmax) * scale);
        for (int e = 0; e < V; e++) {
            const float p = kfo_expf(kfo_bf16_to_f32(row[e]) - bmax) * scale;
            if (write_dlogits) row[e] = kfo_f32_to_bf16((p - (e == ix ? 1.0f : 0.0f)) * dloss);
            if (probs) probs[idx * (long)P + e] = kfo_f32_to_bf16(p);
        }
    }
    free(tmax);
    free(tsum);
}
KFO_API float kfo_logf_export(float x) { return kfo_logf(x); }
/* GELU, tanh form, gelu_forward_kernel2 (src/Device/CUDA/Activation.cu:23-40): 0.5*x*(1 + tanhf(sqrtf(2/pi)*(x + 0.044715*x*x*x))), bf16 store.
 * tanhf: the reference calls the CUDA libm; here tanh(z) = (e - 1)/(e + 1), e = kfo_expf(2z), saturated to +-1 beyond |z| = 10 -- one fixed
 * recipe shared with the HIP kernel (absolute error < 2e-7, far below the bf16 store). */
static inline float kfo_tanhf(float z) {
    if (z > 10.0f) return 1.0f;
    if (z < -10.0f) return -1.0f;
    const float e = kfo_expf(2.0f * z);
    return (e - 1.0f) / (e + 1.0f);
}
KFO_API void kfo_gelu(const uint16_t* x, uint16_t* y, size_t n) {
    const float c = 0.797884583473205566406250f; /* sqrtf(2.0f / M_PI) */
    for (size_t i = 0; i < n; i++) {
        const float xi = kfo_bf16_to_f32(x[i]);
        const float cube = 0.044715f * xi * xi * xi;
        y[i] = kfo_f32_to_bf16(0.5f * xi * (1.0f + kfo_tanhf(c * (xi + cube))));
    }
}

/* LayerNorm / RMSNorm backward (layernorm_backward_kernel10, src/Device/CUDA/kernel/layernorm.cuh:311-503; RMS form CU_rms_back_llmc, :863-1051).
 * mean == NULL: RMSNorm.  Per row: dnorm_i = w_i * dout_i; A = sum dnorm_i; B = sum dnorm_i * inp_i (both fp64 sums of terms that are exact in
 * fp32 -- the reference adds them in warp order in fp32); dnorm_mean = A / C; dnorm_norm_mean = B / C * rstd - dnorm_mean * mean * rstd;
 * norm_i = (inp_i - mean) * rstd; dval = ((w_i * dout_i - dnorm_mean) - norm_i * dnorm_norm_mean) * rstd; dinp_i = bf16(dinp_i + dval).
 * dweight_i = bf16(sum_rows norm_i * dout_i + dweight_i), dbias_i likewise with dout_i: the rows are dealt round-robin to G = min(rows, 512)
 * groups, each summed in row order, then the groups in index order, all in fp64 (the decomposition koifish_amd/csrc/kf_norm_bwd.hip uses;
 * the reference sums per block, then the blocks in index order, in fp32). */
KFO_API void kfo_norm_backward(uint16_t* dinp, uint16_t* dweight, uint16_t* dbias, const uint16_t* dout, const uint16_t* inp, const uint16_t* weight,
                               const float* mean, const float* rstd, int rows, int C) {
    const int G = rows < 512 ? rows : 512, ln = mean != NULL;
    double* pw = (double*)calloc((size_t)G * C, sizeof(double));
    double* pb = (double*)calloc((size_t)G * C, sizeof(double));
    for (int r = 0; r < rows; r++) {
        const uint16_t *dor = dout + (size_t)r * C, *inr = inp + (size_t)r * C;
        uint16_t* dir = dinp + (size_t)r * C;
        const float mean_r = ln ? mean[r] : 0.0f, rstd_r = rstd[r];
        double A = 0.0, B = 0.0;
        for (int c = 0; c < C; c++) {
            const float d = kfo_bf16_to_f32(weight[c]) * kfo_bf16_to_f32(dor[c]);
            A += (double)d;
            B += (double)(d * kfo_bf16_to_f32(inr[c]));
        }
        const float dnorm_mean = ln ? (float)A / (float)C : 0.0f;
        const float dnorm_norm_mean = ln ? (float)B / (float)C * rstd_r - dnorm_mean * mean_r * rstd_r : (float)B / (float)C * rstd_r;
        double* gw = pw + (size_t)(r % G) * C;
        double* gb = pb + (size_t)(r % G) * C;
        for (int c = 0; c < C; c++) {
            const float w_ = kfo_bf16_to_f32(weight[c]), do_ = kfo_bf16_to_f32(dor[c]), in_ = kfo_bf16_to_f32(inr[c]), di_ = kfo_bf16_to_f32(dir[c]);
            const float norm = (in_ - mean_r) * rstd_r;
            gw[c] += (double)(norm * do_);
            gb[c] += (double)do_;
            float dval = w_ * do_;
            if (ln) dval -= dnorm_mean;
            dval -= norm * dnorm_norm_mean;
            dval *= rstd_r;
            dir[c] = kfo_f32_to_bf16(di_ + dval);
        }
    }
    for (int c = 0; c < C; c++) {
        double sw = 0.0, sb = 0.0;
        const int chunk = (G + 7) / 8; /* 8 contiguous chunks of groups, each in index order, then the chunk sums in order */
        for (int q = 0; q < 8; q++) {
            double cw = 0.0, cb = 0.0;
            for (int g = q * chunk; g < (q + 1) * chunk && g < G; g++) cw += pw[(size_t)g * C + c], cb += pb[(size_t)g * C + c];
            sw += cw, sb += cb;
        }
        dweight[c] = kfo_f32_to_bf16((float)sw + kfo_bf16_to_f32(dweight[c]));
        if (ln && dbias) dbias[c] = kfo_f32_to_bf16((float)sb + kfo_bf16_to_f32(dbias[c]));
    }
    free(pw);
    free(pb);
}

/* Causal multi-head attention backward (the mathematics of the reference's cuDNN SDPA backward, QKV.cu:130-315: closed-source arithmetic, parity
 * unpinned): S = scale Q K^T, P = softmax over the keys j <= i, O = P V; D_i = dO_i . O_i (with the STORED bf16 O, as the kernels use it);
 * dV = P^T dO, dP = dO V^T, dS = P o (dP - D), dQ = scale dS K, dK = scale dS^T Q.  Everything in fp64 from the bf16 inputs, bf16 stores. */
KFO_API void kfo_attn_backward(const uint16_t* q, const uint16_t* k, const uint16_t* v, long long ld_qkv, const uint16_t* o, const uint16_t* dO, long long ld_o,
                               uint16_t* dq, uint16_t* dk, uint16_t* dv, long long ld_d, int T, int n_head, int n_kv, int hd) {
    const double scale = 1.0 / sqrt((double)hd);
    const int gq = n_head / n_kv; /* GQA: gq query heads share kv head h / gq; dk, dv sum over them */
    double* P = (double*)malloc(sizeof(double) * (size_t)T * T);
    double* dS = (double*)malloc(sizeof(double) * (size_t)T * T);
    double* ak = (double*)malloc(sizeof(double) * (size_t)T * hd);
    double* av = (double*)malloc(sizeof(double) * (size_t)T * hd);
    for (int kvh = 0; kvh < n_kv; kvh++) {
        const size_t hk = (size_t)kvh * hd;
        for (size_t i = 0; i < (size_t)T * hd; i++) ak[i] = av[i] = 0.0;
        for (int h = kvh * gq; h < (kvh + 1) * gq; h++) {
            const size_t ho = (size_t)h * hd;
            for (int i = 0; i < T; i++) {
                double mx = -INFINITY;
                for (int j = 0; j <= i; j++) {
                    double s = 0.0;
                    for (int d = 0; d < hd; d++) s += (double)kfo_bf16_to_f32(q[(size_t)i * ld_qkv + ho + d]) * (double)kfo_bf16_to_f32(k[(size_t)j * ld_qkv + hk + d]);
                    P[(size_t)i * T + j] = s * scale;
                    if (s * scale > mx) mx = s * scale;
                }
                double sum = 0.0;
                for (int j = 0; j <= i; j++) sum += (P[(size_t)i * T + j] = exp(P[(size_t)i * T + j] - mx));
                double D = 0.0;
                for (int d = 0; d < hd; d++) D += (double)kfo_bf16_to_f32(dO[(size_t)i * ld_o + ho + d]) * (double)kfo_bf16_to_f32(o[(size_t)i * ld_o + ho + d]);
                for (int j = 0; j <= i; j++) {
                    const double p = (P[(size_t)i * T + j] /= sum);
                    double dp = 0.0;
                    for (int d = 0; d < hd; d++) dp += (double)kfo_bf16_to_f32(dO[(size_t)i * ld_o + ho + d]) * (double)kfo_bf16_to_f32(v[(size_t)j * ld_qkv + hk + d]);
                    dS[(size_t)i * T + j] = p * (dp - D);
                }
                for (int d = 0; d < hd; d++) {
                    double a = 0.0;
                    for (int j = 0; j <= i; j++) a += dS[(size_t)i * T + j] * (double)kfo_bf16_to_f32(k[(size_t)j * ld_qkv + hk + d]);
                    dq[(size_t)i * ld_d + ho + d] = kfo_f32_to_bf16((float)(a * scale));
                }
            }
            for (int j = 0; j < T; j++)
                for (int d = 0; d < hd; d++)
                    for (int i = j; i < T; i++) {
                        ak[(size_t)j * hd + d] += dS[(size_t)i * T + j] * (double)kfo_bf16_to_f32(q[(size_t)i * ld_qkv + ho + d]);
                        av[(size_t)j * hd + d] += P[(size_t)i * T + j] * (double)kfo_bf16_to_f32(dO[(size_t)i * ld_o + ho + d]);
                    }
        }
        for (int j = 0; j < T; j++)
            for (int d = 0; d < hd; d++) {
                dk[(size_t)j * ld_d + hk + d] = kfo_f32_to_bf16((float)(ak[(size_t)j * hd + d] * scale));
                dv[(size_t)j * ld_d + hk + d] = kfo_f32_to_bf16((float)av[(size_t)j * hd + d]);
            }
    }
    free(P);
    free(dS);
    free(ak);
    free(av);
}

/* Embedding backward (encoder_backward, kernel/embed.cuh:380-470; wpe_backward_kernel :333-366, wte_backward_kernel :257-331): fp32 sums over the batch
 * (b ascending) / over the positions holding a token (ascending), then bf16(sum + old).  Tokens outside [0, V) are skipped. */
KFO_API void kfo_embed_backward(uint16_t* dwte, long long ldw, uint16_t* dwpe, const uint16_t* dout, const int32_t* tokens, int B, int T, int C, int V) {
    if (dwpe)
        for (int t = 0; t < T; t++)
            for (int c = 0; c < C; c++) {
                float acc = 0.0f;
                for (int b = 0; b < B; b++) acc += kfo_bf16_to_f32(dout[((size_t)b * T + t) * C + c]);
                dwpe[(size_t)t * C + c] = kfo_f32_to_bf16(acc + kfo_bf16_to_f32(dwpe[(size_t)t * C + c]));
            }
    if (dwte) {
        const int N = B * T;
        float* acc = (float*)malloc(sizeof(float) * C);
        for (int bt = 0; bt < N; bt++) {
            const int tok = tokens[bt];
            if (tok < 0 || tok >= V) continue;
            int lead = 1;
            for (int i = 0; i < bt && lead; i++) lead = tokens[i] != tok;
            if (!lead) continue;
            for (int c = 0; c < C; c++) acc[c] = 0.0f;
            for (int i = bt; i < N; i++)
                if (tokens[i] == tok)
                    for (int c = 0; c < C; c++) acc[c] += kfo_bf16_to_f32(dout[(size_t)i * C + c]);
            for (int c = 0; c < C; c++) dwte[(size_t)tok * ldw + c] = kfo_f32_to_bf16(acc[c] + kfo_bf16_to_f32(dwte[(size_t)tok * ldw + c]));
        }
        free(acc);
    }
}

/* Bias gradient of SLP::Back (matmul_backward_bias_kernel9 + reduce_add_sum_kernel, NeuronFuse.cu:511-530): dst[c] = bf16(sum_rows x[r][c] + dst[c]).
 * The sum: slabs of 256 rows in row order, then the slabs in index order, in fp64 (koifish_amd/csrc/kf_linear_bwd.hip; the reference adds per
 * block in fp32 and the blocks in index order). */
KFO_API void kfo_colsum_add(const uint16_t* x, uint16_t* dst, int n, int C) {
    for (int c = 0; c < C; c++) {
        double acc = 0.0;
        for (int s = 0; s * 256 < n; s++) {
            double a = 0.0;
            for (int r = s * 256; r < (s + 1) * 256 && r < n; r++) a += (double)kfo_bf16_to_f32(x[(size_t)r * C + c]);
            acc += a;
        }
        dst[c] = kfo_f32_to_bf16((float)acc + kfo_bf16_to_f32(dst[c]));
    }
}

/* RoPE backward, rotate-half (the transpose of kfo_rope / CU_rope2_v0, operator.cuh:734-772), in place on one row of n_head * hd gradients at `pos`:
 * (g_j, g_{j+hd/2}) -> (g_j c + g_{j+hd/2} s, g_{j+hd/2} c - g_j s), products and sum in fp32 without contraction, bf16 stores. */
KFO_API void kfo_rope_backward(uint16_t* d, int n_head, int hd, const float* cos_t, const float* sin_t) {
    const int half = hd / 2;
    for (int h = 0; h < n_head; h++)
        for (int j = 0; j < half; j++) {
            uint16_t* p = d + (size_t)h * hd + j;
            const float g0 = kfo_bf16_to_f32(p[0]), g1 = kfo_bf16_to_f32(p[half]), c = cos_t[j], sn = sin_t[j];
            const float a = g0 * c, b = g1 * sn, cc = g1 * c, dd = g0 * sn;
            p[0] = kfo_f32_to_bf16(a + b);
            p[half] = kfo_f32_to_bf16(cc - dd);
        }
}

/* GELU backward in place (gelu_backward_inplace_kernel, Activation.cu:42-60) and SwiGLU backward (CU_swiglu_back_v0, Activation.cu:245-260): the
 * reference's expressions, evaluated left to right in fp32; tanh and sech^2 from one kfo_expf(2z) (tanhf / coshf in the reference), the sigmoid from
 * kfo_expf; round-to-nearest stores. */
KFO_API void kfo_gelu_backward(uint16_t* d_in_out, const uint16_t* x, size_t n) {
    const float c = 0.797884583473205566406250f;
    for (size_t i = 0; i < n; i++) {
        const float xi = kfo_bf16_to_f32(x[i]);
        const float cube = 0.044715f * xi * xi * xi;
        const float z = c * (xi + cube);
        float th, sech2;
        if (z > 10.0f) th = 1.0f, sech2 = 0.0f;
        else if (z < -10.0f) th = -1.0f, sech2 = 0.0f;
        else {
            const float e = kfo_expf(2.0f * z), e1 = e + 1.0f;
            th = (e - 1.0f) / e1;
            sech2 = (4.0f * e) / (e1 * e1);
        }
        const float local_grad = 0.5f * (1.0f + th) + xi * 0.5f * sech2 * c * (1.0f + 3.0f * 0.044715f * xi * xi);
        d_in_out[i] = kfo_f32_to_bf16(local_grad * kfo_bf16_to_f32(d_in_out[i]));
    }
}
KFO_API void kfo_swiglu_backward(uint16_t* delta_in_out, uint16_t* delta_gate, const uint16_t* gate, const uint16_t* up, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const float xiW = kfo_bf16_to_f32(gate[i]), xiV = kfo_bf16_to_f32(up[i]), delta = kfo_bf16_to_f32(delta_in_out[i]);
        const float sigW = 1.0f / (1.0f + kfo_expf(-xiW));
        delta_gate[i] = kfo_f32_to_bf16(delta * xiV * sigW * (1.0f + xiW * (1.0f - sigW)));
        delta_in_out[i] = kfo_f32_to_bf16(delta * xiW * sigW);
    }
}

/* AdamW parameter update, CU_adamw_p (src/Device/CUDA/Optimizer.cu:393-442) launched as TASKA_1p1 (kernel/packedN.cuh:612-643): blocks of
 * 512 threads, 8 bf16 elements per thread.  Per element (all fp32): g = grad_scale * grad; m = sAtB(g, m, beta1) = fma(beta1, m, fma(-beta1, g, g))
 * (kernel/utils.cuh:26-31); v likewise with g*g and beta2; m_hat = m / beta1_correction, v_hat = v / beta2_correction;
 * step = m_hat / (sqrtf(v_hat) + eps); p = p - lr*wd*p - lr*step; grads are zeroed.  A thread that meets a non-finite parameter or step
 * stores nothing (and reports KOIFISH_ADAMW_MV through the prober: status_out here).
 * Stores go through PackedN(config, f256) -> CU_Float2T<bf16>(x, seed) (packedN.cuh:360-363, 62-72): SEEDED STOCHASTIC ROUNDING -- one
 * 16-bit threshold per thread from SquirrelNoise5 keyed on (threadIdx.x, blockIdx.x * blockDim.x + blockIdx.y, seed) (utils.cuh:296-326),
 * low 16 bits of the fp32 compared with it, then round-to-nearest of the all-ones / all-zeros padded value.  It is a pure function of the
 * launch geometry and the seed, so it is restated exactly.  m and v are stored the same way when they are bf16 (floatMV = bf16), plainly
 * when they are fp32. */
static inline uint32_t kfo_squirrel5(uint32_t pos, uint32_t seed) {
    uint32_t b = pos;
    b *= 0xd2a80a3fu;
    b += seed;
    b ^= (b >> 9);
    b += 0xa884f197u;
    b ^= (b >> 11);
    b *= 0x6C736F4Bu;
    b ^= (b >> 13);
    b += 0xB79F3ABBu;
    b ^= (b >> 15);
    b *= 0x1b56c4f5u;
    b ^= (b >> 17);
    return b;
}
KFO_API uint32_t kfo_noise2d(int x, int y, uint32_t seed) { return kfo_squirrel5((uint32_t)x + 198491317u * (uint32_t)y, seed); }
static inline uint16_t kfo_stochastic_bf16(float a, uint32_t threshold) {
    uint32_t u;
    memcpy(&u, &a, 4);
    u = ((u & 0xFFFFu) > threshold) ? (u | 0xFFFFu) : (u & ~0xFFFFu);
    float f;
    memcpy(&f, &u, 4);
    return kfo_f32_to_bf16(f);
}
KFO_API int kfo_adamw(uint16_t* params, uint16_t* grads, void* gm, void* gv, size_t n, int mv_bf16, float lr, float beta1, float beta2, float b1c, float b2c,
                      float eps, float wd, float grad_scale, uint32_t seed) {
    if (n % 8) return -1;
    int status = 0;
    const size_t nthread = n / 8;
    for (size_t t = 0; t < nthread; t++) {
        const int tx = (int)(t % 512), bx = (int)(t / 512);
        const uint32_t thr = kfo_noise2d(tx, bx * 512, seed) & 0xFFFFu;
        float pm[8], pv[8], pp[8];
        int bad = 0;
        for (int i = 0; i < 8 && !bad; i++) {
            const size_t idx = t * 8 + i;
            const float g = grad_scale * kfo_bf16_to_f32(grads[idx]);
            float m = mv_bf16 ? kfo_bf16_to_f32(((uint16_t*)gm)[idx]) : ((float*)gm)[idx];
            float v = mv_bf16 ? kfo_bf16_to_f32(((uint16_t*)gv)[idx]) : ((float*)gv)[idx];
            m = fmaf(beta1, m, fmaf(-beta1, g, g));
            const float g2 = g * g;
            v = fmaf(beta2, v, fmaf(-beta2, g2, g2));
            pm[i] = m, pv[i] = v;
            const float mh = m / b1c, vh = v / b2c;
            const float step = mh / (sqrtf(vh) + eps);
            const float old = kfo_bf16_to_f32(params[idx]);
            if (!isfinite(old) || !isfinite(step)) {
                bad = 1;
                break;
            }
            pp[i] = old - lr * wd * old - lr * step;
        }
        if (bad) {
            status = -1;
            continue;
        }
        for (int i = 0; i < 8; i++) {
            const size_t idx = t * 8 + i;
            if (mv_bf16)
                ((uint16_t*)gm)[idx] = kfo_stochastic_bf16(pm[i], thr), ((uint16_t*)gv)[idx] = kfo_stochastic_bf16(pv[i], thr);
            else
                ((float*)gm)[idx] = pm[i], ((float*)gv)[idx] = pv[i];
            params[idx] = kfo_stochastic_bf16(pp[i], thr);
            grads[idx] = 0;
        }
    }
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * 6. Decode attention (GQA), full causal over t = 0..pos.
 *    mode 0 "REF":   the reference's rounding chain -- attention_qk_kernel / CU_softmax_multihead /
 *                    attention_v_kernel (operator.cuh:572-632, 251-277, 649-668) with scores and
 *                    probabilities held in bf16 (qk_v is floatX, TGraph.cpp:124).
 *    mode 1 "FUSED": what the fused HIP kernel computes: bf16 scores (same store as REF), then an fp32
 *                    softmax, out = (sum_t e_t v_t) * (1/sum_t e_t) in fp32, one bf16 store.
 *    mode 2 "CANON": the order kernels and oracle share (round 3), free of any launch geometry:
 *                    score  s_t = bf16( d_t * (1/sqrtf(hd)) ), d_t = the q.k dot summed as hd/8 chains of 8 fused multiply-adds (elements
 *                           8j .. 8j+7 in order) joined by a balanced binary tree -- the 16 (8) lanes of a key on the GPU;
 *                    weight p_t = f_t * 2^(n_t - m), (f_t, n_t) = kfo_exp2_parts(s_t * log2 e), m = max_t n_t: an exact power-of-two scaling,
 *                           so slices / waves / lanes of the GPU may each work against a maximum of their own and rescale exactly;
 *                    sums   L = sum_t p_t and O_i = sum_t p_t v_t,i in fp64 (every term is exact in fp64; the sums are then independent of their
 *                           order far below an fp32 ulp -- the argument of the fp64 RMSNorm sum), out_i = bf16( (float)(O_i / L) ).
 *    q: bf16 [n_head*hd]; kc/vc: layer base, rows of kv_stride elements; out: bf16 [n_head*hd].
 * ---------------------------------------------------------------------------------------------- */
static void attn_decode_canon(const uint16_t* q, const uint16_t* kc, const uint16_t* vc, uint16_t* out, int pos, int n_head, int n_kv, int hd, int kv_stride) {
    const int kv_mul = n_head / n_kv, len = pos + 1, LPK = hd / 8;
    const float rden = 1.0f / sqrtf((float)hd);
#pragma omp parallel for schedule(static)
    for (int h = 0; h < n_head; h++) {
        const int kvh = h / kv_mul;
        float qf[128];
        for (int i = 0; i < hd; i++) qf[i] = kfo_bf16_to_f32(q[(size_t)h * hd + i]);
        float* ff = (float*)malloc(sizeof(float) * len);
        float* nn = (float*)malloc(sizeof(float) * len);
        double* O = (double*)calloc(hd, sizeof(double));
        float m = -INFINITY;
        for (int t = 0; t < len; t++) {
            const uint16_t* kt = kc + (size_t)t * kv_stride + (size_t)kvh * hd;
            float lane[16];
            for (int j = 0; j < LPK; j++) {
                float acc = 0.f;
                for (int i = 0; i < 8; i++) acc = fmaf(qf[8 * j + i], kfo_bf16_to_f32(kt[8 * j + i]), acc);
                lane[j] = acc;
            }
            for (int s = 1; s < LPK; s <<= 1)
                for (int j = 0; j < LPK; j += 2 * s) lane[j] = lane[j] + lane[j + s];
            const float sc = kfo_round_bf16(lane[0] * rden);
            kfo_exp2_parts(sc * 1.44269502162933349609375f, &ff[t], &nn[t]);
            if (nn[t] > m) m = nn[t];
        }
        double L = 0.0;
        for (int t = 0; t < len; t++) {
            const uint16_t* vt = vc + (size_t)t * kv_stride + (size_t)kvh * hd;
            const double p = ldexp((double)ff[t], (int)fmaxf(nn[t] - m, -1022.0f));
            L += p;
            for (int i = 0; i < hd; i++) O[i] = fma(p, (double)kfo_bf16_to_f32(vt[i]), O[i]);
        }
        for (int i = 0; i < hd; i++) out[(size_t)h * hd + i] = kfo_f32_to_bf16((float)(O[i] / L));
        free(ff), free(nn), free(O);
    }
}

KFO_API void kfo_attn_decode(const uint16_t* q, const uint16_t* kc, const uint16_t* vc, uint16_t* out, int pos, int n_head, int n_kv, int hd,
                             int kv_stride, int mode) {
    if (mode == 2) {
        attn_decode_canon(q, kc, vc, out, pos, n_head, n_kv, hd, kv_stride);
        return;
    }
    const int kv_mul = n_head / n_kv, len = pos + 1;
    const float scale_den = sqrtf((float)hd);
#pragma omp parallel for schedule(static)
    for (int h = 0; h < n_head; h++) {
        const int kvh = h / kv_mul;
        const uint16_t* qh = q + (size_t)h * hd;
        float* att = (float*)malloc(sizeof(float) * len);
        float* acc = (float*)calloc(hd, sizeof(float));
        float m = -1e9f;
        for (int t = 0; t < len; t++) {
            const uint16_t* kt = kc + (size_t)t * kv_stride + (size_t)kvh * hd;
            float score = 0.0f;
            for (int i = 0; i < hd; i++) score = fmaf(kfo_bf16_to_f32(qh[i]), kfo_bf16_to_f32(kt[i]), score);
            score /= scale_den;
            att[t] = kfo_round_bf16(score);
            if (att[t] > m) m = att[t];
        }
        if (mode == 0) {
            m = kfo_round_bf16(m); /* T max_val = -1e9f: a bf16 */
            float sum = 0.0f;
            for (int t = 0; t < len; t++) {
                float a = kfo_expf(kfo_round_bf16(att[t] - m)); /* bf16 subtract, then expf(float) */
                sum += a;
                att[t] = kfo_round_bf16(a);
            }
            float inv = kfo_round_bf16(1.0f / sum); /* scores[i] *= inv_sum: float -> bf16, bf16*bf16 */
            for (int t = 0; t < len; t++) att[t] = kfo_round_bf16(att[t] * inv);
            for (int t = 0; t < len; t++) {
                const uint16_t* vt = vc + (size_t)t * kv_stride + (size_t)kvh * hd;
                for (int i = 0; i < hd; i++) acc[i] = fmaf(att[t], kfo_bf16_to_f32(vt[i]), acc[i]);
            }
            for (int i = 0; i < hd; i++) out[(size_t)h * hd + i] = kfo_f32_to_bf16(acc[i]);
        } else {
            float sum = 0.0f;
            for (int t = 0; t < len; t++) {
                float a = kfo_expf(att[t] - m);
                sum += a;
                att[t] = a;
            }
            float inv = 1.0f / sum;
            for (int t = 0; t < len; t++) {
                const uint16_t* vt = vc + (size_t)t * kv_stride + (size_t)kvh * hd;
                for (int i = 0; i < hd; i++) acc[i] = fmaf(att[t], kfo_bf16_to_f32(vt[i]), acc[i]);
            }
            for (int i = 0; i < hd; i++) out[(size_t)h * hd + i] = kfo_f32_to_bf16(acc[i] * inv);
        }
        free(att);
        free(acc);
    }
}

/* ------------------------------------------------------------------------------------------------
 * 7. Qwen3 decoder: op order of SelfAttention::cuInfer (QKV.cu:617-702), ROPE::cuInfer
 *    (kernel/rope.cu:645-672), FFN::cuInfer (NeuronFuse.cu:615-656), Head4Token::cuInfer_1
 *    (NeuronFuse.cu:842-862); SURVEY.md section 9 lists every rounding point.
 *    Tensor-parallel emulation (tp > 1): q/k/v/gate/up split by output rows, o/down by input columns
 *    with fp32 partials summed in rank order 0..tp-1 and one bf16 store.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const uint16_t *norm_in, *norm_post, *qn, *kn;
    kfo_weight q, k, v, o, gate, up, down;
    const int32_t* hot; /* sparse forward: 1 = hot FFN row (CS_Picker::hot), NULL = dense */
} kfo_layer;

typedef struct {
    int dim, n_layer, n_head, n_kv, head_dim, ffn, vocab, max_seq;
    float rms_eps, qk_eps, theta;
    int attn_mode, tp;
    kfo_weight embed, head;
    const uint16_t* final_norm;
    kfo_layer* layers;
    uint16_t *kcache, *vcache; /* [n_layer, max_seq, kv_dim] bf16 (src/Utils/Cache.cpp:14-26) */
    /* kfo_qwen3_prepare_fast: 16-bit views of every mat-vec weight for the vectorised path (section 4b); [n_layer * 7 + 1] (last = head) */
    const uint16_t** fast;
    uint16_t** fast_own; /* the copies this model allocated (NULL where the tensor's own bf16 / f16 data is used) */
    int* fast_f16;
} kfo_qwen3;

KFO_API kfo_qwen3* kfo_qwen3_new(int dim, int n_layer, int n_head, int n_kv, int head_dim, int ffn, int vocab, int max_seq, float rms_eps,
                                 float qk_eps, float theta, int attn_mode) {
    kfo_qwen3* m = (kfo_qwen3*)calloc(1, sizeof(kfo_qwen3));
    m->dim = dim, m->n_layer = n_layer, m->n_head = n_head, m->n_kv = n_kv, m->head_dim = head_dim, m->ffn = ffn, m->vocab = vocab;
    m->max_seq = max_seq, m->rms_eps = rms_eps, m->qk_eps = qk_eps, m->theta = theta, m->attn_mode = attn_mode, m->tp = 1;
    m->layers = (kfo_layer*)calloc(n_layer, sizeof(kfo_layer));
    size_t n = (size_t)n_layer * max_seq * n_kv * head_dim;
    m->kcache = (uint16_t*)calloc(n, 2);
    m->vcache = (uint16_t*)calloc(n, 2);
    return m;
}
KFO_API void kfo_qwen3_free(kfo_qwen3* m) {
    if (!m) return;
    if (m->fast_own)
        for (int i = 0; i < m->n_layer * 7 + 1; i++) free(m->fast_own[i]);
    free(m->fast), free(m->fast_own), free(m->fast_f16);
    free(m->layers), free(m->kcache), free(m->vcache), free(m);
}
KFO_API void kfo_qwen3_set_tp(kfo_qwen3* m, int tp) { m->tp = tp; }
KFO_API void kfo_qwen3_set_attn_mode(kfo_qwen3* m, int mode) { m->attn_mode = mode; }
KFO_API uint16_t* kfo_qwen3_kcache(kfo_qwen3* m) { return m->kcache; }
KFO_API uint16_t* kfo_qwen3_vcache(kfo_qwen3* m) { return m->vcache; }

/* slot: 0 q,1 k,2 v,3 o,4 gate,5 up,6 down (layer>=0);  layer=-1: slot 0 embed, 1 head */
KFO_API int kfo_qwen3_set_weight(kfo_qwen3* m, int layer, int slot, int type, int ne0, int ne1, const void* data, const uint16_t* zero,
                                 const uint16_t* step, int lGroup, int qBias) {
    kfo_weight w = {type, ne0, ne1, data, zero, step, lGroup, qBias};
    if (layer < 0) {
        if (slot == 0)
            m->embed = w;
        else
            m->head = w;
        return 0;
    }
    if (layer >= m->n_layer || slot < 0 || slot > 6) return -1;
    kfo_weight* dst[7] = {&m->layers[layer].q, &m->layers[layer].k, &m->layers[layer].v, &m->layers[layer].o,
                          &m->layers[layer].gate, &m->layers[layer].up, &m->layers[layer].down};
    *dst[slot] = w;
    return 0;
}
/* slot: 0 input_layernorm, 1 post_attention_layernorm, 2 q_norm, 3 k_norm; layer=-1: final norm */
KFO_API int kfo_qwen3_set_norm(kfo_qwen3* m, int layer, int slot, const uint16_t* w) {
    if (layer < 0) {
        m->final_norm = w;
        return 0;
    }
    if (layer >= m->n_layer) return -1;
    kfo_layer* L = &m->layers[layer];
    if (slot == 0) L->norm_in = w;
    else if (slot == 1) L->norm_post = w;
    else if (slot == 2) L->qn = w;
    else if (slot == 3) L->kn = w;
    else return -1;
    return 0;
}

/* hot[ffn] (1 = hot) for the layer's gate / up rows, NULL = dense; the array stays owned by the caller */
KFO_API int kfo_qwen3_set_hot(kfo_qwen3* m, int layer, const int32_t* hot) {
    if (layer < 0 || layer >= m->n_layer) return -1;
    m->layers[layer].hot = hot;
    return 0;
}

/* row-split weights: plain rows.  column-split weights: fp32 partials per rank, summed in rank order. */
static void linear_colsplit(const kfo_weight* w, const uint16_t* x, uint16_t* y, int tp) {
    if (tp <= 1) {
        kfo_linear(w, x, y, NULL, 1.0f, 0.0f);
        return;
    }
    const int M = w->ne0, K = w->ne1, kc = K / tp;
    float* part = (float*)malloc(sizeof(float) * M);
    float* tot = (float*)calloc(M, sizeof(float));
    kfo_set_launch_rows(M); /* canonical order: each rank's launch multiplies its column shard of all M rows */
    for (int r = 0; r < tp; r++) {
        kfo_linear_f32(w, x, part, r * kc, (r + 1) * kc);
        for (int i = 0; i < M; i++) tot[i] = (r == 0) ? part[i] : tot[i] + part[i];
    }
    kfo_set_launch_rows(0);
    for (int i = 0; i < M; i++) y[i] = kfo_f32_to_bf16(tot[i]);
    free(part), free(tot);
}

/* 16-bit views for the vectorised mat-vec: bf16 / f16 tensors as they are, everything else dequantised once into a bf16 copy (what GetDataX
 * produces).  Returns the bytes allocated, < 0 when a tensor cannot take the path (K not a multiple of 16, AutoAWQ layout) -- then nothing changes. */
KFO_API long long kfo_qwen3_prepare_fast(kfo_qwen3* m) {
    if (m->fast) return 0;
    const int n = m->n_layer * 7 + 1;
    const kfo_weight** ws = (const kfo_weight**)malloc(sizeof(void*) * n);
    for (int l = 0; l < m->n_layer; l++) {
        kfo_layer* L = &m->layers[l];
        const kfo_weight* w7[7] = {&L->q, &L->k, &L->v, &L->o, &L->gate, &L->up, &L->down};
        for (int j = 0; j < 7; j++) ws[l * 7 + j] = w7[j];
    }
    ws[n - 1] = &m->head;
    for (int i = 0; i < n; i++)
        if (!ws[i]->data || ws[i]->ne1 % 16 != 0 || ws[i]->type == KFO_Q4_AWQ) {
            free(ws);
            return -1;
        }
    m->fast = (const uint16_t**)calloc(n, sizeof(void*));
    m->fast_own = (uint16_t**)calloc(n, sizeof(void*));
    m->fast_f16 = (int*)calloc(n, sizeof(int));
    long long bytes = 0;
    for (int i = 0; i < n; i++) {
        const kfo_weight* w = ws[i];
        if (w->type == KFO_BF16 || w->type == KFO_F16) {
            m->fast[i] = (const uint16_t*)w->data, m->fast_f16[i] = w->type == KFO_F16;
            continue;
        }
        int shared = -1; /* tied tensors: one copy */
        for (int j = 0; j < i; j++)
            if (ws[j]->data == w->data && ws[j]->type == w->type) shared = j;
        if (shared >= 0) {
            m->fast[i] = m->fast[shared];
            continue;
        }
        const size_t ne = (size_t)w->ne0 * w->ne1;
        m->fast_own[i] = (uint16_t*)malloc(ne * 2);
        kfo_dequant_weight(w, m->fast_own[i]);
        m->fast[i] = m->fast_own[i], bytes += (long long)ne * 2;
    }
    free(ws);
    return bytes;
}
static void model_linear(kfo_qwen3* m, int layer, int slot, const kfo_weight* w, const uint16_t* x, uint16_t* y, const int32_t* hot, long rows) {
    const int epb = epb_of_type(w->type);
    if (g_order == 1 && epb > 0 && w->ne1 % epb == 0) { /* canonical order: `rows` = the rows of the launch this matrix is multiplied in */
        const int i = layer < 0 ? m->n_layer * 7 : layer * 7 + slot;
        if (m->fast && !m->fast_f16[i]) {
            linear_canon_w16(m->fast[i], w->ne0, w->ne1, epb, kfo_lpr_log2_epb(epb, w->ne1, rows), x, y, hot);
        } else {
            kfo_set_launch_rows(rows);
            if (hot) kfo_linear_masked(w, x, y, NULL, hot);
            else kfo_linear(w, x, y, NULL, 1.0f, 0.0f);
            kfo_set_launch_rows(0);
        }
        return;
    }
    if (m->fast) {
        const int i = layer < 0 ? m->n_layer * 7 : layer * 7 + slot;
        linear_w16(m->fast[i], m->fast_f16[i], w->ne0, w->ne1, x, y, hot);
    } else if (hot) {
        kfo_linear_masked(w, x, y, NULL, hot);
    } else {
        kfo_linear(w, x, y, NULL, 1.0f, 0.0f);
    }
}

/* One decode step.  logits_out (bf16[vocab]) may be NULL.  hidden_out (bf16[dim], after the final
 * norm) may be NULL.  Returns the greedy token id. */
KFO_API int kfo_qwen3_decode(kfo_qwen3* m, int token, int pos, uint16_t* logits_out, uint16_t* hidden_out) {
    const int D = m->dim, hd = m->head_dim, qd = m->n_head * hd, kvd = m->n_kv * hd, F = m->ffn;
    uint16_t* x = (uint16_t*)malloc(2 * D);
    uint16_t* xb = (uint16_t*)malloc(2 * D);
    uint16_t* q = (uint16_t*)malloc(2 * qd);
    uint16_t* att = (uint16_t*)malloc(2 * qd);
    uint16_t* p = (uint16_t*)malloc(2 * D);
    uint16_t* gt = (uint16_t*)malloc(2 * F);
    uint16_t* up = (uint16_t*)malloc(2 * F);
    kfo_embed(&m->embed, token, x);
    for (int l = 0; l < m->n_layer; l++) {
        kfo_layer* L = &m->layers[l];
        uint16_t* kc = m->kcache + (size_t)l * m->max_seq * kvd;
        uint16_t* vc = m->vcache + (size_t)l * m->max_seq * kvd;
        uint16_t *krow = kc + (size_t)pos * kvd, *vrow = vc + (size_t)pos * kvd; /* _devQKV: TGraph.cpp:198-207 */
        kfo_rmsnorm(x, L->norm_in, xb, 1, D, m->rms_eps);
        const int tp = m->tp > 1 ? m->tp : 1;
        model_linear(m, l, 0, &L->q, xb, q, NULL, (qd + 2 * kvd) / tp); /* Q | K | V are one launch (per rank: its shard of each) */
        model_linear(m, l, 1, &L->k, xb, krow, NULL, (qd + 2 * kvd) / tp);
        model_linear(m, l, 2, &L->v, xb, vrow, NULL, (qd + 2 * kvd) / tp);
        if (L->qn) kfo_headnorm(q, L->qn, m->n_head, hd, m->qk_eps);
        if (L->kn) kfo_headnorm(krow, L->kn, m->n_kv, hd, m->qk_eps);
        kfo_rope(q, m->n_head, hd, pos, m->theta);
        kfo_rope(krow, m->n_kv, hd, pos, m->theta);
        kfo_attn_decode(q, kc, vc, att, pos, m->n_head, m->n_kv, hd, kvd, m->attn_mode);
        if (m->tp <= 1) model_linear(m, l, 3, &L->o, att, p, NULL, D);
        else linear_colsplit(&L->o, att, p, m->tp);
        kfo_add(x, p, x, D);
        kfo_rmsnorm(x, L->norm_post, xb, 1, D, m->rms_eps);
        model_linear(m, l, 4, &L->gate, xb, gt, L->hot, F / tp); /* L->hot: the sparse forward, D_matmul_sparse on the FFN's rows; gate | up: the paired launch counts gate's rows */
        model_linear(m, l, 5, &L->up, xb, up, L->hot, F / tp);
        kfo_swiglu(gt, up, gt, F);
        if (m->tp <= 1) model_linear(m, l, 6, &L->down, gt, p, NULL, D);
        else linear_colsplit(&L->down, gt, p, m->tp);
        kfo_add(x, p, x, D);
    }
    kfo_rmsnorm(x, m->final_norm, xb, 1, D, m->rms_eps);
    if (hidden_out) memcpy(hidden_out, xb, 2 * D);
    uint16_t* logits = logits_out ? logits_out : (uint16_t*)malloc(2 * (size_t)m->vocab);
    model_linear(m, -1, 1, &m->head, xb, logits, NULL, m->vocab / (m->tp > 1 ? m->tp : 1));
    int next = kfo_argmax_bf16(logits, m->vocab);
    if (!logits_out) free(logits);
    free(x), free(xb), free(q), free(att), free(p), free(gt), free(up);
    return next;
}

/* thread count of the OpenMP regions, and a stand-alone mat-vec probe (M x K 16-bit weights, pages first touched by the threads that read them) that
 * bench.py uses to pick the thread count the host actually sustains: containers often expose more logical CPUs than they may use, and a 2-socket host
 * loses more to remote memory and barriers than it gains from the second socket on these small matrices */
KFO_API void kfo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
KFO_API double kfo_bench_matvec(int M, int K, int reps) {
    uint16_t* W = (uint16_t*)malloc((size_t)M * K * 2);
    uint16_t* x = (uint16_t*)malloc((size_t)K * 2);
    uint16_t* y = (uint16_t*)malloc((size_t)M * 2);
    if (!W || !x || !y) return -1.0;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < M; r++)
        for (int c = 0; c < K; c++) W[(size_t)r * K + c] = (uint16_t)(0x3c00 + ((r * 31 + c * 7) & 0xff));
    for (int c = 0; c < K; c++) x[c] = 0x3f80;
    linear_w16(W, 0, M, K, x, y, NULL);
    double t0 = 0.0, t1 = 0.0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#endif
    for (int i = 0; i < reps; i++) linear_w16(W, 0, M, K, x, y, NULL);
#ifdef _OPENMP
    t1 = omp_get_wtime();
#endif
    free(W), free(x), free(y);
    return (t1 - t0) / reps;
}
KFO_API int kfo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
