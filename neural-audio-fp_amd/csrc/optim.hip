// Multi-tensor optimizer steps for gfx950: keras Adam and the reference's LAMB.
//
// Replaces `opt.apply_gradients(zip(g, m_fp.trainable_variables))` of the reference's
// train step (model/trainer.py:47-48) for
//   * tf.keras.optimizers.Adam (trainer.py:138): lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
//     w -= lr_t * m / (sqrt(v) + eps), eps = 1e-7;
//   * LAMB (model/fp/lamb_optimizer.py:123-158): m_hat/(sqrt(v_hat)+eps) + wd*w, trust ratio
//     ||w|| / ||update|| per VARIABLE, eps 1e-6, wd 1e-6 on every variable.
// The reference has 576 variables: 64 conv/LN tensors + 128 slices x (W1, b1, W2, b2).  Here the
// slice variables are stacked into 4 tensors, so a tensor carries `var_len`: the length of one
// keras variable inside it, and LAMB's norms are taken per var_len-segment -- 576 trust ratios,
// as in the reference.  All tensors of one step go through a few launches (HBM-bound
// elementwise work: 16.9 M parameters x (w, g, m, v)).
#include "nafp_common.h"

namespace nafp {

constexpr int OPT_MAX_T = 48;            // tensors per launch (kernel-argument table)

struct OptTable {
    float* p[OPT_MAX_T];
    const float* g[OPT_MAX_T];
    float* m[OPT_MAX_T];
    float* v[OPT_MAX_T];
    long long numel[OPT_MAX_T];
    long long var_len[OPT_MAX_T];
    long long norm_off[OPT_MAX_T];       // first (w2, u2) pair of this tensor in the norm buffer
    int blk0[OPT_MAX_T + 1];             // first work block of tensor i (blocks of OPT_BLK elements); blk0[n] = grid size
    int n;
};

// Work is cut into blocks of OPT_BLK elements over ALL tensors of a launch (one workgroup per block, tensor found by a
// search in the block prefix): the tensors range from 128 to 3.1 M elements, and with a fixed number of workgroups per
// tensor (round 1-2: 64) the launch lasted as long as 64 workgroups need for the largest one -- LAMB took 523 us per
// step for 475 MB of traffic.
constexpr int OPT_BLK = 4096;

__device__ __forceinline__ int opt_find_tensor(const OptTable& t, int blk) {
    int lo = 0, hi = t.n - 1;
    while (lo < hi) {                                        // largest i with blk0[i] <= blk
        const int mid = (lo + hi + 1) >> 1;
        if (t.blk0[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void adam_kernel(const OptTable t, float lr_t, float b1, float b2, float eps) {
    const int ti = opt_find_tensor(t, blockIdx.x);
    float* __restrict__ p = t.p[ti]; const float* __restrict__ g = t.g[ti];
    float* __restrict__ m = t.m[ti]; float* __restrict__ v = t.v[ti];
    const long long n = t.numel[ti];
    const long long i0 = (long long)(blockIdx.x - t.blk0[ti]) * OPT_BLK;
    const bool vec_ok = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;   // 16-B aligned tensors take float4
#pragma unroll
    for (int e = 0; e < OPT_BLK / 1024; ++e) {
        const long long i = i0 + e * 1024 + 4 * threadIdx.x;
        if (vec_ok && i + 3 < n) {
            const float4 g4 = *(const float4*)(g + i), m4 = *(const float4*)(m + i), v4 = *(const float4*)(v + i);
            float4 p4 = *(const float4*)(p + i), mo, vo;
            mo.x = b1 * m4.x + (1.f - b1) * g4.x; vo.x = b2 * v4.x + (1.f - b2) * g4.x * g4.x; p4.x -= lr_t * mo.x / (sqrtf(vo.x) + eps);
            mo.y = b1 * m4.y + (1.f - b1) * g4.y; vo.y = b2 * v4.y + (1.f - b2) * g4.y * g4.y; p4.y -= lr_t * mo.y / (sqrtf(vo.y) + eps);
            mo.z = b1 * m4.z + (1.f - b1) * g4.z; vo.z = b2 * v4.z + (1.f - b2) * g4.z * g4.z; p4.z -= lr_t * mo.z / (sqrtf(vo.z) + eps);
            mo.w = b1 * m4.w + (1.f - b1) * g4.w; vo.w = b2 * v4.w + (1.f - b2) * g4.w * g4.w; p4.w -= lr_t * mo.w / (sqrtf(vo.w) + eps);
            *(float4*)(m + i) = mo; *(float4*)(v + i) = vo; *(float4*)(p + i) = p4;
        } else {
            for (long long k = i; k < n && k < i + 4; ++k) {
                const float gi = g[k];
                const float mi = b1 * m[k] + (1.f - b1) * gi;
                const float vi = b2 * v[k] + (1.f - b2) * gi * gi;
                m[k] = mi; v[k] = vi;
                p[k] -= lr_t * mi / (sqrtf(vi) + eps);
            }
        }
    }
}

// LAMB phase 1: moments, and per-variable sum(w^2), sum(update^2) into `norms` (double pairs).
__global__ __launch_bounds__(256) void lamb_moments_kernel(const OptTable t, float b1, float b2, float eps, float wd,
                                                           float inv_bc1, float inv_bc2, double* __restrict__ norms) {
    const int ti = opt_find_tensor(t, blockIdx.x);
    const float* __restrict__ p = t.p[ti]; const float* __restrict__ g = t.g[ti];
    float* __restrict__ m = t.m[ti]; float* __restrict__ v = t.v[ti];
    const long long n = t.numel[ti], vl = t.var_len[ti];
    double* nb = norms + 2 * t.norm_off[ti];
    const bool one_var = vl == n;
    const long long i0 = (long long)(blockIdx.x - t.blk0[ti]) * OPT_BLK;
    double w2 = 0.0, u2 = 0.0;
    const bool vec_ok = one_var && i0 + OPT_BLK <= n &&
                        ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    if (vec_ok) {
        // whole block of one variable, 16-B aligned: float4 loads, all four arrays of an iteration in flight together;
        // the squares are summed in float per thread (16 terms) and in double from there on
        float w2f = 0.f, u2f = 0.f;
#pragma unroll
        for (int e = 0; e < OPT_BLK / 1024; ++e) {
            const long long i = i0 + e * 1024 + 4 * threadIdx.x;
            const float4 g4 = *(const float4*)(g + i), p4 = *(const float4*)(p + i);
            float4 m4 = *(const float4*)(m + i), v4 = *(const float4*)(v + i);
#define NAFP_LAMB1(c_)                                                                       \
            {                                                                                \
                m4.c_ = b1 * m4.c_ + (1.f - b1) * g4.c_;                                     \
                v4.c_ = b2 * v4.c_ + (1.f - b2) * g4.c_ * g4.c_;                             \
                const float u_l = (m4.c_ * inv_bc1) / (sqrtf(v4.c_ * inv_bc2) + eps) + wd * p4.c_; \
                w2f = fmaf(p4.c_, p4.c_, w2f); u2f = fmaf(u_l, u_l, u2f);                    \
            }
            NAFP_LAMB1(x) NAFP_LAMB1(y) NAFP_LAMB1(z) NAFP_LAMB1(w)
#undef NAFP_LAMB1
            *(float4*)(m + i) = m4; *(float4*)(v + i) = v4;
        }
        w2 = (double)w2f; u2 = (double)u2f;
    } else
#pragma unroll
    for (int e = 0; e < OPT_BLK / 256; ++e) {
        const long long i = i0 + e * 256 + threadIdx.x;
        if (i >= n) break;
        const float gi = g[i], wi = p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float u = (mi * inv_bc1) / (sqrtf(vi * inv_bc2) + eps) + wd * wi;
        if (one_var) { w2 += (double)wi * wi; u2 += (double)u * u; }
        else {
            // stacked variables (the divide-and-encode tensors: 128 variables of 256 / 32 / 32 / 1 elements).  One double atomic
            // per ELEMENT onto 2 addresses per variable serialised at the L2 (the launch holding these tensors ran 176 us for
            // 11 M elements, the other 30 us for 6 M): lanes of one variable are summed first -- a wave (var_len a multiple of
            // 64) or a half wave (32) holds consecutive elements of ONE variable, tensors start on block boundaries.
            const long long s = i / vl;
            double a = (double)wi * wi, c = (double)u * u;
            if (vl % 64 == 0 && n % 64 == 0) {
                a = wave_sum(a); c = wave_sum(c);
                if ((threadIdx.x & 63) == 0) { atomicAdd(nb + 2 * s, a); atomicAdd(nb + 2 * s + 1, c); }
            } else if (vl % 32 == 0 && n % 64 == 0) {
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
                if ((threadIdx.x & 31) == 0) { atomicAdd(nb + 2 * s, a); atomicAdd(nb + 2 * s + 1, c); }
            } else {
                atomicAdd(nb + 2 * s, a);
                atomicAdd(nb + 2 * s + 1, c);
            }
        }
    }
    if (one_var) {
        w2 = wave_sum(w2); u2 = wave_sum(u2);
        __shared__ double red[8];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { red[wave] = w2; red[4 + wave] = u2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(nb, red[0] + red[1] + red[2] + red[3]);
            atomicAdd(nb + 1, red[4] + red[5] + red[6] + red[7]);
        }
    }
}

// LAMB phase 2: w -= ratio * lr * update, ratio = ||w||/||update|| (1 if either norm is 0).
__global__ __launch_bounds__(256) void lamb_apply_kernel(const OptTable t, float lr, float eps, float wd,
                                                         float inv_bc1, float inv_bc2, const double* __restrict__ norms) {
    const int ti = opt_find_tensor(t, blockIdx.x);
    float* __restrict__ p = t.p[ti];
    const float* __restrict__ m = t.m[ti]; const float* __restrict__ v = t.v[ti];
    const long long n = t.numel[ti], vl = t.var_len[ti];
    const double* nb = norms + 2 * t.norm_off[ti];
    const long long i0 = (long long)(blockIdx.x - t.blk0[ti]) * OPT_BLK;
    float ratio1 = 1.f;                                        // the tensor is ONE variable: its ratio once per workgroup
    if (vl == n) {
        const float wn = (float)sqrt(nb[0]), un = (float)sqrt(nb[1]);
        ratio1 = (wn > 0.f && un > 0.f) ? wn / un : 1.f;
    }
#pragma unroll
    for (int e = 0; e < OPT_BLK / 256; ++e) {
        const long long i = i0 + e * 256 + threadIdx.x;
        if (i >= n) break;
        float ratio = ratio1;
        if (vl != n) {
            const long long s = i / vl;
            const float wn = (float)sqrt(nb[2 * s]), un = (float)sqrt(nb[2 * s + 1]);
            ratio = (wn > 0.f && un > 0.f) ? wn / un : 1.f;
        }
        const float wi = p[i];
        const float u = (m[i] * inv_bc1) / (sqrtf(v[i] * inv_bc2) + eps) + wd * wi;
        p[i] = wi - ratio * lr * u;
    }
}

static int fill_table(OptTable& tb, const nafp_opt_tensor* ts, int first, int n, long long& norm_cursor) {
    tb.n = n;
    tb.blk0[0] = 0;
    for (int i = 0; i < n; ++i) {
        const nafp_opt_tensor& t = ts[first + i];
        if (!t.param || !t.grad || !t.m || !t.v || t.numel <= 0 || t.var_len <= 0 || t.numel % t.var_len != 0)
            return NAFP_ERR_INVALID_ARG;
        tb.p[i] = t.param; tb.g[i] = t.grad; tb.m[i] = t.m; tb.v[i] = t.v;
        tb.numel[i] = t.numel; tb.var_len[i] = t.var_len; tb.norm_off[i] = norm_cursor;
        norm_cursor += t.numel / t.var_len;
        const long long nblk = (t.numel + OPT_BLK - 1) / OPT_BLK;
        if ((long long)tb.blk0[i] + nblk > 0x7fffffffll) return NAFP_ERR_UNSUPPORTED;
        tb.blk0[i + 1] = tb.blk0[i] + (int)nblk;
    }
    return NAFP_OK;
}

}  // namespace nafp

using namespace nafp;

extern "C" float nafp_cosine_decay_lr_host(float lr0, int64_t step, int64_t decay_steps, float alpha) {
    // tf.keras.experimental.CosineDecay (trainer.py:119-124): lr0 * ((1-alpha) * 0.5 (1 + cos(pi s/S)) + alpha)
    if (decay_steps <= 0) return lr0;
    const double s = (double)(step < decay_steps ? step : decay_steps) / (double)decay_steps;
    const double c = 0.5 * (1.0 + cos(M_PI * s));
    return (float)((double)lr0 * ((1.0 - (double)alpha) * c + (double)alpha));
}

extern "C" float nafp_cosine_decay_restarts_lr_host(float lr0, int64_t step, int64_t first_decay_steps, float t_mul,
                                                     float m_mul, float alpha) {
    // tf.keras.experimental.CosineDecayRestarts (SGDR; trainer.py:125-131 passes first_decay_steps = 0.1 * total
    // steps and alpha = 2e-6, t_mul / m_mul at their defaults 2 / 1)
    if (first_decay_steps <= 0) return lr0;
    double f = (double)step / (double)first_decay_steps, i_restart;
    if (t_mul != 1.0f) {
        i_restart = floor(log(1.0 - f * (1.0 - (double)t_mul)) / log((double)t_mul));
        const double sum_r = (1.0 - pow((double)t_mul, i_restart)) / (1.0 - (double)t_mul);
        f = (f - sum_r) / pow((double)t_mul, i_restart);
    } else {
        i_restart = floor(f);
        f -= i_restart;
    }
    const double c = 0.5 * pow((double)m_mul, i_restart) * (1.0 + cos(M_PI * f));
    return (float)((double)lr0 * ((1.0 - (double)alpha) * c + (double)alpha));
}

extern "C" int nafp_adam_step(const nafp_opt_tensor* tensors_host, int n, float lr, float beta1, float beta2,
                              float eps, int64_t step, void* stream) {
    if (!tensors_host || n <= 0 || step < 1) return NAFP_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float lr_t = (float)((double)lr * sqrt(bc2) / bc1);
    long long cursor = 0;
    for (int first = 0; first < n; first += OPT_MAX_T) {
        OptTable tb;
        const int cnt = n - first < OPT_MAX_T ? n - first : OPT_MAX_T;
        int rc = fill_table(tb, tensors_host, first, cnt, cursor);
        if (rc != NAFP_OK) return rc;
        adam_kernel<<<dim3((unsigned)tb.blk0[cnt]), 256, 0, st>>>(tb, lr_t, beta1, beta2, eps);
        NAFP_LAUNCH_CHECK();
    }
    return NAFP_OK;
}

extern "C" int64_t nafp_lamb_workspace_bytes(const nafp_opt_tensor* tensors_host, int n) {
    if (!tensors_host || n <= 0) return -1;
    int64_t vars = 0;
    for (int i = 0; i < n; ++i) {
        if (tensors_host[i].var_len <= 0 || tensors_host[i].numel % tensors_host[i].var_len != 0) return -1;
        vars += tensors_host[i].numel / tensors_host[i].var_len;
    }
    return vars * 2 * (int64_t)sizeof(double) + 256;
}

extern "C" int nafp_lamb_step(const nafp_opt_tensor* tensors_host, int n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t step, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    if (!tensors_host || n <= 0 || step < 1 || !workspace) return NAFP_ERR_INVALID_ARG;
    const int64_t need = nafp_lamb_workspace_bytes(tensors_host, n);
    if (need < 0) return NAFP_ERR_INVALID_ARG;
    if (workspace_bytes < need) return NAFP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* norms = (double*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
    NAFP_HIP_CHECK(hipMemsetAsync(norms, 0, (size_t)(need - 256), st));
    const float inv_bc1 = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
    const float inv_bc2 = (float)(1.0 / (1.0 - pow((double)beta2, (double)step)));
    for (int pass = 0; pass < 2; ++pass) {
        long long cursor = 0;
        for (int first = 0; first < n; first += OPT_MAX_T) {
            OptTable tb;
            const int cnt = n - first < OPT_MAX_T ? n - first : OPT_MAX_T;
            int rc = fill_table(tb, tensors_host, first, cnt, cursor);
            if (rc != NAFP_OK) return rc;
            if (pass == 0)
                lamb_moments_kernel<<<dim3((unsigned)tb.blk0[cnt]), 256, 0, st>>>(tb, beta1, beta2, eps, weight_decay, inv_bc1,
                                                                   inv_bc2, norms);
            else
                lamb_apply_kernel<<<dim3((unsigned)tb.blk0[cnt]), 256, 0, st>>>(tb, lr, eps, weight_decay, inv_bc1, inv_bc2, norms);
            NAFP_LAUNCH_CHECK();
        }
    }
    return NAFP_OK;
}
