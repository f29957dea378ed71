// The normalisation alternates of the conv stack for gfx950: MODEL.BN = 'layer_norm1d' | 'batch_norm' of the reference's config
// (model/fp/nnfp.py:63-71, 250).  The default, 'layer_norm2d' (LayerNormalization over (F, T, C) per sample), is folded into the
// GEMM epilogues (conv.hip) and needs nothing from this file.
//
// Both alternates run on the SAME conv / tail / backward kernels, which are given
//   * told to derive IDENTITY scalars r_b = 1, c_b = -mu_b r_b = 0 from whatever the producer's statistics hold (`ident_stats` of
//     the launch arguments -> stat_ln_scalars with a negative 1/n, nafp_common.h) -- except for a POISONED sample, whose scalars
//     stay NaN: with identity scalars nothing else would carry a NaN / Inf sample through the packed ELU of the next epilogue --, and
//   * internal positional scale / offset images (the library's gamma / beta slots of shape (F, T, C)):
//
// 'batch_norm' -- keras BatchNormalization(axis=-1) as the reference CALLS it: `m_fp(feat)` without a `training` argument, in the
//   train step as in generate (trainer.py:44, 60; generate.py:88), i.e. in inference mode: a per-channel affine map built from
//   the moving statistics, x_hat = (v - mm_c) / sqrt(mv_c + 1e-3) * gamma_c + beta_c.  That folds exactly: the positional images
//   are the broadcasts of a_c = gamma_c s_c and b_c = beta_c - mm_c s_c gamma_c (bn_expand_kernel), the producers store
//   z = a . v, and the parameter gradients come from the positional sums the LayerNorm backward forms anyway
//   (bn_param_grad_kernel).  No extra pass over the activations.
//
// 'layer_norm1d' -- LayerNormalization(axis=-1): statistics over the C channels of ONE position.  A consumer sums three taps with
//   three different (mu, r), so the per-sample fold does not apply: the producers store v = ELU(t) (positional images 1 / 0), and
//   one elementwise pass per layer normalises the rows in place (ln1d_fwd_kernel: one wave per row, the row in registers); the
//   backward pass has the matching row kernel in front of the shared LayerNorm / ELU backward (ln1d_bwd_kernel).  Two extra HBM
//   crossings per layer and direction: an alternate that works, not a tuned path.
#include "nafp_common.h"

#include <algorithm>

namespace nafp {

// ---- batch_norm (inference-mode affine) -------------------------------------------------------------------------------------
__device__ __forceinline__ float nz_scale_n(float g) { return fabsf(g) < 1e-30f ? copysignf(1e-30f, g) : g; }     // as multi_copy_kernel (api.hip)

__global__ __launch_bounds__(256) void bn_expand_kernel(const BnExpandTable t) {
    const int e = blockIdx.y;
    const int C = t.C[e];
    const int64_t n = t.n[e];
    float* __restrict__ gp = t.gamma_pos[e];
    float* __restrict__ bp = gp + n;                                   // (the library's gamma | beta slots are adjacent)
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float s = 1.0f / sqrtf(t.mvar[e][c] + LN_EPS);
        const float a = t.gamma_c[e][c] * s;
        gp[i] = nz_scale_n(a);
        bp[i] = t.beta_c[e][c] - t.mmean[e][c] * a;
    }
}

int launch_bn_expand(const BnExpandTable& t, hipStream_t st) {
    if (t.count <= 0) return NAFP_OK;
    bn_expand_kernel<<<dim3(64, t.count), 256, 0, st>>>(t);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// dgamma_c += s_c (sum_p dgp[p, c] - mm_c sum_p dbp[p, c]),  dbeta_c += sum_p dbp[p, c]   (dgp / dbp: the positional sums
// sum_b D v and sum_b D of the shared LayerNorm backward, with D = dL/dx_hat)
__global__ __launch_bounds__(256) void bn_param_grad_kernel(const float* __restrict__ dgp, const float* __restrict__ dbp, int P, int C,
                                                            const float* __restrict__ mmean, const float* __restrict__ mvar,
                                                            float* __restrict__ dgamma_c, float* __restrict__ dbeta_c, int p_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int p0 = blockIdx.y * p_per_block, p1 = min(P, p0 + p_per_block);
    float sg = 0.f, sb = 0.f;
    for (int p = p0; p < p1; ++p) { sg += dgp[(int64_t)p * C + c]; sb += dbp[(int64_t)p * C + c]; }
    const float s = 1.0f / sqrtf(mvar[c] + LN_EPS);
    atomicAdd(dgamma_c + c, s * (sg - mmean[c] * sb));
    atomicAdd(dbeta_c + c, sb);
}

int launch_bn_param_grad(const float* dgp, const float* dbp, int P, int C, const float* mmean, const float* mvar, float* dgamma_c,
                         float* dbeta_c, hipStream_t st) {
    const int ppb = 64;
    bn_param_grad_kernel<<<dim3((unsigned)((C + 255) / 256), (unsigned)((P + ppb - 1) / ppb)), 256, 0, st>>>(dgp, dbp, P, C, mmean, mvar, dgamma_c,
                                                                                                           dbeta_c, ppb);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// ---- layer_norm1d -----------------------------------------------------------------------------------------------------------
// One wave per row of C = 64 NV channels; lane l holds channels 2 l, 2 l + 1 (NV = 2) or 4 l .. 4 l + 3 of every 256-channel chunk.
template <int NV>
__device__ __forceinline__ void row_load(const float* __restrict__ row, int lane, float (&v)[NV]) {
    if (NV == 2) {
        const float2 t = ((const float2*)row)[lane]; v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int k = 0; k < NV / 4; ++k) {
            const float4 t = ((const float4*)row)[k * 64 + lane];
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    }
}
template <int NV>
__device__ __forceinline__ void row_store(float* __restrict__ row, int lane, const float (&v)[NV]) {
    if (NV == 2) {
        ((float2*)row)[lane] = make_float2(v[0], v[1]);
    } else {
#pragma unroll
        for (int k = 0; k < NV / 4; ++k) ((float4*)row)[k * 64 + lane] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
}
template <int NV>
__device__ __forceinline__ int chan_of(int lane, int q) { return NV == 2 ? 2 * lane + q : (q / 4) * 256 + 4 * lane + (q % 4); }

// mean and 1 / sqrt(var + eps) of a row, two passes over the registers (tf.nn.moments: the variance is the mean of the squared
// deviations)
template <int NV>
__device__ __forceinline__ void row_moments(const float (&v)[NV], float* mean, float* rstd) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) s += v[q];
    const float m = wave_sum(s) * (1.0f / (64 * NV));
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) { const float d = v[q] - m; ss += d * d; }
    const float var = wave_sum(ss) * (1.0f / (64 * NV));
    *mean = m; *rstd = 1.0f / sqrtf(var + LN_EPS);
}

// x (rows, C) in place: x <- (x - mean_row) rstd_row gamma_c + beta_c
template <int NV>
__global__ __launch_bounds__(256) void ln1d_fwd_kernel(float* __restrict__ x, int64_t rows, const float* __restrict__ gamma_c,
                                                       const float* __restrict__ beta_c) {
    constexpr int C = 64 * NV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NV], b[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) { g[q] = gamma_c[chan_of<NV>(lane, q)]; b[q] = beta_c[chan_of<NV>(lane, q)]; }
    for (int64_t r = blockIdx.x * 4ll + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        float v[NV];
        row_load<NV>(x + r * C, lane, v);
        float mean, rstd;
        row_moments<NV>(v, &mean, &rstd);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = (v[q] - mean) * rstd * g[q] + b[q];
        row_store<NV>(x + r * C, lane, v);
    }
}

int launch_ln1d_fwd(float* x, int64_t rows, int C, const float* gamma_c, const float* beta_c, hipStream_t st) {
    const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, 16384);
    if (rows <= 0) return NAFP_OK;
    switch (C) {
        case 128: ln1d_fwd_kernel<2><<<grid, 256, 0, st>>>(x, rows, gamma_c, beta_c); break;
        case 256: ln1d_fwd_kernel<4><<<grid, 256, 0, st>>>(x, rows, gamma_c, beta_c); break;
        case 512: ln1d_fwd_kernel<8><<<grid, 256, 0, st>>>(x, rows, gamma_c, beta_c); break;
        case 1024: ln1d_fwd_kernel<16><<<grid, 256, 0, st>>>(x, rows, gamma_c, beta_c); break;
        default: return NAFP_ERR_UNSUPPORTED;
    }
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// d (rows, C) holds dL/dy of y = x_hat gamma_c + beta_c, x_hat = (v - mean) rstd over the row, v = ELU(t) with t the stored
// pre-activation.  In place: d <- dL/dv = rstd (gh - mean(gh) - x_hat mean(gh x_hat)), gh = d gamma_c;
// dgamma_c += sum_rows d x_hat, dbeta_c += sum_rows d (a wave's rows in registers, the workgroup's four waves through LDS, then one
// atomic per channel and workgroup).
template <int NV>
__global__ __launch_bounds__(256) void ln1d_bwd_kernel(float* __restrict__ d, const float* __restrict__ tpre, int64_t rows,
                                                       const float* __restrict__ gamma_c, float* __restrict__ dgamma_c,
                                                       float* __restrict__ dbeta_c) {
    constexpr int C = 64 * NV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NV], dg[NV], db[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) { g[q] = gamma_c[chan_of<NV>(lane, q)]; dg[q] = 0.f; db[q] = 0.f; }
    for (int64_t r = blockIdx.x * 4ll + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        float v[NV], dy[NV];
        row_load<NV>(tpre + r * C, lane, v);
        row_load<NV>(d + r * C, lane, dy);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = elu1(v[q]);
        float mean, rstd;
        row_moments<NV>(v, &mean, &rstd);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float xh = (v[q] - mean) * rstd;
            const float gh = dy[q] * g[q];
            dg[q] += dy[q] * xh; db[q] += dy[q];
            s1 += gh; s2 += gh * xh;
            v[q] = xh; dy[q] = gh;
        }
        const float m1 = wave_sum(s1) * (1.0f / C), m2 = wave_sum(s2) * (1.0f / C);
#pragma unroll
        for (int q = 0; q < NV; ++q) dy[q] = rstd * (dy[q] - m1 - v[q] * m2);
        row_store<NV>(d + r * C, lane, dy);
    }
    __shared__ float red[2][4][C];
#pragma unroll
    for (int q = 0; q < NV; ++q) { red[0][wave][chan_of<NV>(lane, q)] = dg[q]; red[1][wave][chan_of<NV>(lane, q)] = db[q]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(dgamma_c + c, (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
        atomicAdd(dbeta_c + c, (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
    }
}

int launch_ln1d_bwd(float* d, const float* tpre, int64_t rows, int C, const float* gamma_c, float* dgamma_c, float* dbeta_c, hipStream_t st) {
    if (rows <= 0) return NAFP_OK;
    // few, long workgroups: every workgroup ends in 2 C atomics onto the same addresses
    const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, 1024);
    switch (C) {
        case 128: ln1d_bwd_kernel<2><<<grid, 256, 0, st>>>(d, tpre, rows, gamma_c, dgamma_c, dbeta_c); break;
        case 256: ln1d_bwd_kernel<4><<<grid, 256, 0, st>>>(d, tpre, rows, gamma_c, dgamma_c, dbeta_c); break;
        case 512: ln1d_bwd_kernel<8><<<grid, 256, 0, st>>>(d, tpre, rows, gamma_c, dgamma_c, dbeta_c); break;
        case 1024: ln1d_bwd_kernel<16><<<grid, 256, 0, st>>>(d, tpre, rows, gamma_c, dgamma_c, dbeta_c); break;
        default: return NAFP_ERR_UNSUPPORTED;
    }
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

}  // namespace nafp
