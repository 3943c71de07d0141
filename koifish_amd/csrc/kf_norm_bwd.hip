// kf_norm_bwd.hip -- LayerNorm / RMSNorm backward: the input gradient accumulated over the residual-path gradient, and the weight (and bias)
// gradient accumulated over the rows.
//
// Replaces layernorm_backward_kernel10 (src/Device/CUDA/kernel/layernorm.cuh:311-503, launched by layernorm_backward from LayerNormal::cuFlow,
// T.cu:644) and its RMS form CU_rms_back_llmc (layernorm.cuh:863-1051, T.cu:634).  Per row (mean = 0 for RMS):
//   dnorm_i = w_i * dout_i;  A = sum_i dnorm_i;  B = sum_i dnorm_i * inp_i;
//   dnorm_mean = A / C (LayerNorm only);  dnorm_norm_mean = B / C * rstd - dnorm_mean * mean * rstd;
//   norm_i = (inp_i - mean) * rstd;  dval = ((w_i * dout_i - dnorm_mean) - norm_i * dnorm_norm_mean) * rstd;  dinp_i = bf16(dinp_i + dval);
//   dweight_i += sum_rows norm_i * dout_i;  dbias_i += sum_rows dout_i  (bf16(fp32 sum + old value)).
// The reference adds A and B in warp order in fp32; here both are fp64 sums of EXACT terms (a product of two or three bf16 values is exact in
// fp32), so the result does not depend on the order -- the same device as for the forward norms.  The column sums run over the rows a
// workgroup owns (rows w, w + G, ... with G = min(rows, 512): fp64, in row order) and then over the G workgroups in a second launch (8 contiguous chunks of workgroups in index order, then the 8 chunk sums in order) -- a fixed
// decomposition that oracle/kf_oracle.c kfo_norm_backward restates (the reference sums its blocks in index order too, in fp32).
// HBM-bound: 3 reads + 1 write of rows*C bf16.  One workgroup owns whole rows: thread t holds the 8-column vectors t, t + 256, ... (C <= 8192).
#include "kf_kernels.h"

namespace kf {

constexpr int NB_MAXV = 4; /* 8-column vectors per thread: C <= 8192 */

template <bool IS_LN, int NV>
__global__ void __launch_bounds__(256) norm_backward_kernel(uint16_t* __restrict__ dinp, const uint16_t* __restrict__ dout, const uint16_t* __restrict__ inp,
                                                            const uint16_t* __restrict__ weight, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            double* __restrict__ part, int rows, int C) {
    __shared__ double red[2][2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = gridDim.x;
    const int nvec = C >> 3;
    bool has[NV];
    u32x4 wv[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        has[j] = tid + 256 * j < nvec;
        wv[j] = has[j] ? *reinterpret_cast<const u32x4*>(weight + (size_t)(tid + 256 * j) * 8) : u32x4{0, 0, 0, 0};
    }
    double dw[NV][8], db[IS_LN ? NV : 1][8];
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            dw[j][k] = 0.0;
            if (IS_LN) db[j][k] = 0.0;
        }
    // the next row of this workgroup is requested (unconditionally: the last row re-reads itself) before the current one is reduced, so
    // that a row costs its arithmetic and one barrier, not a memory round trip
    u32x4 dov[NV], inv[NV], div[NV], ndo[NV], nin[NV], ndi[NV];
    auto load_row = [&](int r, u32x4* a, u32x4* b, u32x4* c_) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const size_t o = (size_t)r * C + (size_t)(has[j] ? tid + 256 * j : 0) * 8;
            a[j] = *reinterpret_cast<const u32x4*>(dout + o);
            b[j] = *reinterpret_cast<const u32x4*>(inp + o);
            c_[j] = *reinterpret_cast<const u32x4*>(dinp + o);
        }
    };
    if ((int)blockIdx.x < rows) load_row(blockIdx.x, dov, inv, div);
    int buf = 0;
    for (int r = blockIdx.x; r < rows; r += G, buf ^= 1) {
        const size_t base = (size_t)r * C;
        load_row(r + G < rows ? r + G : r, ndo, nin, ndi);
        const float mean_r = IS_LN ? mean[r] : 0.0f, rstd_r = rstd[r];
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            if (!has[j]) continue;
            const uint32_t wq[4] = {wv[j].x, wv[j].y, wv[j].z, wv[j].w}, dq_[4] = {dov[j].x, dov[j].y, dov[j].z, dov[j].w}, iq[4] = {inv[j].x, inv[j].y, inv[j].z, inv[j].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float d0 = bf_lo(wq[k]) * bf_lo(dq_[k]), d1 = bf_hi(wq[k]) * bf_hi(dq_[k]);
                if (IS_LN) sa += (double)d0 + (double)d1;
                sb += (double)(d0 * bf_lo(iq[k])) + (double)(d1 * bf_hi(iq[k]));
            }
        }
        sa = wave_sum_f64_fast(sa), sb = wave_sum_f64_fast(sb);
        if (lane == 0) red[buf][0][wave] = sa, red[buf][1][wave] = sb;
        __syncthreads(); /* one barrier per row: the two buffers alternate */
        const double A = (red[buf][0][0] + red[buf][0][1]) + (red[buf][0][2] + red[buf][0][3]);
        const double B = (red[buf][1][0] + red[buf][1][1]) + (red[buf][1][2] + red[buf][1][3]);
        const float dnorm_mean = IS_LN ? (float)A / (float)C : 0.0f;
        const float dnorm_norm_mean = IS_LN ? (float)B / (float)C * rstd_r - dnorm_mean * mean_r * rstd_r : (float)B / (float)C * rstd_r;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            if (!has[j]) continue;
            const uint32_t wq[4] = {wv[j].x, wv[j].y, wv[j].z, wv[j].w}, dq_[4] = {dov[j].x, dov[j].y, dov[j].z, dov[j].w}, iq[4] = {inv[j].x, inv[j].y, inv[j].z, inv[j].w},
                           gq[4] = {div[j].x, div[j].y, div[j].z, div[j].w};
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float res[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float w_ = h ? bf_hi(wq[k]) : bf_lo(wq[k]), do_ = h ? bf_hi(dq_[k]) : bf_lo(dq_[k]), in_ = h ? bf_hi(iq[k]) : bf_lo(iq[k]),
                                di_ = h ? bf_hi(gq[k]) : bf_lo(gq[k]);
                    const float norm = (in_ - mean_r) * rstd_r;
                    dw[j][2 * k + h] += (double)(norm * do_);
                    if (IS_LN) db[j][2 * k + h] += (double)do_;
                    float dval = w_ * do_;
                    if (IS_LN) dval -= dnorm_mean;
                    dval -= norm * dnorm_norm_mean;
                    dval *= rstd_r;
                    res[h] = di_ + dval;
                }
                o[k] = pack_bf16x2(res[0], res[1]);
            }
            *reinterpret_cast<u32x4*>(dinp + base + (size_t)(tid + 256 * j) * 8) = u32x4{o[0], o[1], o[2], o[3]};
        }
#pragma unroll
        for (int j = 0; j < NV; j++) dov[j] = ndo[j], inv[j] = nin[j], div[j] = ndi[j];
    }
    // this workgroup's column partials: part[blockIdx.x][{dw, db}][C] fp64
    double* pw = part + (size_t)blockIdx.x * (IS_LN ? 2 : 1) * C;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        if (!has[j]) continue;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            pw[(size_t)(tid + 256 * j) * 8 + k] = dw[j][k];
            if (IS_LN) pw[C + (size_t)(tid + 256 * j) * 8 + k] = db[j][k];
        }
    }
}

// Narrow rows (dim = 64 .. 512, a power of two: the per-head q/k-norm of attention, rows = tokens x heads): LPR = dim / 8 lanes per row, 64 / LPR rows
// per wave and 4 x that per workgroup at once; the row sums stay inside the row's lanes (DPP / row swaps on fp64 halves), the column partials are
// combined over the wave's rows and the 4 waves at the end.  Same group decomposition (rows r = w + G i belong to workgroup w) as the wide form; inside
// a group the fp64 sums of fp32 terms are exact, so visiting its rows 4 x 64 / LPR at a time instead of one by one changes no bit.
__device__ __forceinline__ double nb_lanes_sum(double v, int lpr_log2) { /* over the 2^lpr_log2 (8 .. 64) lanes of a row */
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    if (lpr_log2 >= 4) v += dpp_d<0x140>(v);
    if (lpr_log2 >= 5) v = xsum16_d(v);
    if (lpr_log2 >= 6) v = xsum32_d(v);
    return v;
}
template <bool IS_LN>
__global__ void __launch_bounds__(256) norm_backward_narrow_kernel(uint16_t* __restrict__ dinp, const uint16_t* __restrict__ dout, const uint16_t* __restrict__ inp,
                                                                   const uint16_t* __restrict__ weight, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   double* __restrict__ part, int rows, int C, int lpr_log2) {
    __shared__ double red[4][64][8 * (IS_LN ? 2 : 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = gridDim.x;
    const int LPR = 1 << lpr_log2, RPW = 64 >> lpr_log2; /* lanes per row, rows per wave */
    const int ll = lane & (LPR - 1), slot = wave * RPW + (lane >> lpr_log2), nslot = 4 * RPW;
    const u32x4 wv = *reinterpret_cast<const u32x4*>(weight + (size_t)ll * 8);
    const uint32_t wq[4] = {wv.x, wv.y, wv.z, wv.w};
    double dw[8], db[8];
#pragma unroll
    for (int k = 0; k < 8; k++) dw[k] = 0.0, db[k] = 0.0;
    for (long i = 0;; i++) {
        const long r0 = (long)blockIdx.x + (long)G * (i * nslot); /* the first row of this round: workgroup-uniform exit */
        if (r0 >= rows) break;
        const long r = r0 + (long)G * slot;
        const bool ok = r < rows;
        const size_t base = (size_t)(ok ? r : r0) * C + (size_t)ll * 8;
        const u32x4 dov = *reinterpret_cast<const u32x4*>(dout + base), inv = *reinterpret_cast<const u32x4*>(inp + base), div = *reinterpret_cast<const u32x4*>(dinp + base);
        const float mean_r = IS_LN ? mean[ok ? r : r0] : 0.0f, rstd_r = rstd[ok ? r : r0];
        const uint32_t dq_[4] = {dov.x, dov.y, dov.z, dov.w}, iq[4] = {inv.x, inv.y, inv.z, inv.w}, gq[4] = {div.x, div.y, div.z, div.w};
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float d0 = bf_lo(wq[k]) * bf_lo(dq_[k]), d1 = bf_hi(wq[k]) * bf_hi(dq_[k]);
            if (IS_LN) sa += (double)d0 + (double)d1;
            sb += (double)(d0 * bf_lo(iq[k])) + (double)(d1 * bf_hi(iq[k]));
        }
        if (IS_LN) sa = nb_lanes_sum(sa, lpr_log2);
        sb = nb_lanes_sum(sb, lpr_log2);
        const float dnorm_mean = IS_LN ? (float)sa / (float)C : 0.0f;
        const float dnorm_norm_mean = IS_LN ? (float)sb / (float)C * rstd_r - dnorm_mean * mean_r * rstd_r : (float)sb / (float)C * rstd_r;
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float res[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float w_ = h ? bf_hi(wq[k]) : bf_lo(wq[k]), do_ = h ? bf_hi(dq_[k]) : bf_lo(dq_[k]), in_ = h ? bf_hi(iq[k]) : bf_lo(iq[k]), di_ = h ? bf_hi(gq[k]) : bf_lo(gq[k]);
                const float norm = (in_ - mean_r) * rstd_r;
                if (ok) {
                    dw[2 * k + h] += (double)(norm * do_);
                    if (IS_LN) db[2 * k + h] += (double)do_;
                }
                float dval = w_ * do_;
                if (IS_LN) dval -= dnorm_mean;
                dval -= norm * dnorm_norm_mean;
                dval *= rstd_r;
                res[h] = di_ + dval;
            }
            o[k] = pack_bf16x2(res[0], res[1]);
        }
        if (ok) *reinterpret_cast<u32x4*>(dinp + base) = u32x4{o[0], o[1], o[2], o[3]};
    }
    // column partials: every lane to LDS, then thread c < C sums the lanes that hold column c (RPW per wave, 4 waves) in a fixed order
#pragma unroll
    for (int k = 0; k < 8; k++) {
        red[wave][lane][k] = dw[k];
        if (IS_LN) red[wave][lane][8 + k] = db[k];
    }
    __syncthreads();
    double* pw = part + (size_t)blockIdx.x * (IS_LN ? 2 : 1) * C;
    for (int c = tid; c < C; c += 256) {
        const int l0 = c >> 3, k = c & 7;
        double tw = 0.0, tb = 0.0;
        for (int w2 = 0; w2 < 4; w2++)
            for (int rr = 0; rr < RPW; rr++) {
                tw += red[w2][rr * LPR + l0][k];
                if (IS_LN) tb += red[w2][rr * LPR + l0][8 + k];
            }
        pw[c] = tw;
        if (IS_LN) pw[C + c] = tb;
    }
}

// second launch: the G partials of a column -- 8 contiguous chunks of ceil(G / 8) workgroups summed in index order by 8 threads, the 8 chunk
// sums added in chunk order -- then bf16(fp32 sum + old gradient).  (One thread walking all G partials is a chain of G dependent loads.)
__global__ void __launch_bounds__(256) norm_backward_reduce_kernel(uint16_t* __restrict__ dweight, uint16_t* __restrict__ dbias, const double* __restrict__ part,
                                                                   int G, int C, int is_ln) {
    __shared__ double sw_[8][32], sb_[8][32];
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const size_t stride = (size_t)(is_ln ? 2 : 1) * C;
    const int chunk = (G + 7) / 8, g0 = q * chunk, g1 = g0 + chunk < G ? g0 + chunk : G;
    double sw = 0.0, sbias = 0.0;
    if (c < C) { /* the same index order, eight partials' loads in flight at a time (one at a time this launch took 24 us for 512 partials of 1600 columns) */
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            double vw[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                vw[u] = part[(g + u) * stride + c];
                vb[u] = is_ln ? part[(g + u) * stride + C + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) sw += vw[u], sbias += vb[u];
        }
        for (; g < g1; g++) {
            sw += part[g * stride + c];
            if (is_ln) sbias += part[g * stride + C + c];
        }
    }
    sw_[q][cl] = sw, sb_[q][cl] = sbias;
    __syncthreads();
    if (q == 0 && c < C) {
        double tw = 0.0, tb = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) tw += sw_[k][cl], tb += sb_[k][cl];
        dweight[c] = f2bf((float)tw + bf2f(dweight[c]));
        if (is_ln && dbias) dbias[c] = f2bf((float)tb + bf2f(dbias[c]));
    }
}

int norm_backward_groups(int rows) { return rows < 512 ? rows : 512; }

int norm_backward_launch(hipStream_t st, uint16_t* dinp, uint16_t* dweight, uint16_t* dbias, const uint16_t* dout, const uint16_t* inp, const uint16_t* weight,
                         const float* mean, const float* rstd, int rows, int C, double* scratch) {
    if (rows < 1 || C < 8 || (C % 8) != 0 || C > NB_MAXV * 2048) return KF_INVALID_ARGS;
    const int G = norm_backward_groups(rows), nv = (C / 8 + 255) / 256;
    const bool ln = mean != nullptr;
    if (C >= 64 && C <= 512 && (C & (C - 1)) == 0 && rows >= 4096) { /* narrow rows, many of them: several rows per wave */
        const int lpr_log2 = __builtin_ctz(C) - 3;
        if (ln) hipLaunchKernelGGL((norm_backward_narrow_kernel<true>), dim3(G), dim3(256), 0, st, dinp, dout, inp, weight, mean, rstd, scratch, rows, C, lpr_log2);
        else hipLaunchKernelGGL((norm_backward_narrow_kernel<false>), dim3(G), dim3(256), 0, st, dinp, dout, inp, weight, mean, rstd, scratch, rows, C, lpr_log2);
        hipLaunchKernelGGL(norm_backward_reduce_kernel, dim3((C + 31) / 32), dim3(256), 0, st, dweight, dbias, scratch, G, C, ln ? 1 : 0);
        return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
    }
#define KF_NB_GO(NV)                                                                                                                             \
    do {                                                                                                                                         \
        if (ln) hipLaunchKernelGGL((norm_backward_kernel<true, NV>), dim3(G), dim3(256), 0, st, dinp, dout, inp, weight, mean, rstd, scratch, rows, C); \
        else hipLaunchKernelGGL((norm_backward_kernel<false, NV>), dim3(G), dim3(256), 0, st, dinp, dout, inp, weight, mean, rstd, scratch, rows, C);   \
    } while (0)
    switch (nv) {
        case 1: KF_NB_GO(1); break;
        case 2: KF_NB_GO(2); break;
        case 3: KF_NB_GO(3); break;
        default: KF_NB_GO(4); break;
    }
#undef KF_NB_GO
    hipLaunchKernelGGL(norm_backward_reduce_kernel, dim3((C + 31) / 32), dim3(256), 0, st, dweight, dbias, scratch, G, C, ln ? 1 : 0);
    return hipGetLastError() == hipSuccess ? KF_OK : KF_HIP_CHECK;
}

}  // namespace kf
