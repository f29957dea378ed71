// Spec-augment hole masking for gfx950 (train step only).
//
// Replaces SpecNCutout.call with uniform_mask=True (model/fp/specaug_chain/layers/
// ncutout_tarray.py:214-268): one rectangle set shared by the whole batch, an optional per-sample
// activation flag, holes replaced by a fill value ('zeros': 0; 'min': the reference actually fills
// with reduce_mean(x), ncutout_tarray.py:203-204 -- the host passes that value).  The rectangles
// are drawn on the host (neural-audio-fp_amd/model/fp/specaug_chain); HBM-bound elementwise pass.
#include "nafp_common.h"

#include <algorithm>

namespace nafp {

constexpr int MAX_RECTS = 8;
struct RectTable { int f0[MAX_RECTS], f1[MAX_RECTS], t0[MAX_RECTS], t1[MAX_RECTS]; int n; };

__global__ __launch_bounds__(256) void specaug_kernel(float* __restrict__ x, int64_t n_vec4, int F, int T,
                                                      const RectTable r, const unsigned char* __restrict__ active,
                                                      float fill, const float* __restrict__ fill_dev) {
    if (fill_dev) fill = *fill_dev;
    const int per_seg4 = F * T / 4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n_vec4; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_seg4;
        if (active && !active[b]) continue;
        const int e = (int)(i - b * per_seg4) * 4;
        const int f = e / T, t = e - f * T;              // T % 4 == 0: the 4 elements share f
        bool hole[4] = {false, false, false, false};
#pragma unroll
        for (int k = 0; k < MAX_RECTS; ++k)
            if (k < r.n && f >= r.f0[k] && f <= r.f1[k]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) hole[j] |= (t + j >= r.t0[k] && t + j <= r.t1[k]);
            }
        if (hole[0] | hole[1] | hole[2] | hole[3]) {
            float4 v = ((float4*)x)[i];
            v.x = hole[0] ? fill : v.x; v.y = hole[1] ? fill : v.y;
            v.z = hole[2] ? fill : v.z; v.w = hole[3] ? fill : v.w;
            ((float4*)x)[i] = v;
        }
    }
}

// The general form (SpecNCutout with uniform_mask=False and / or a filler TENSOR, ncutout_tarray.py:106-115, 200-211, 270-276):
// rectangle sets in device memory, one per sample (n_sets == n_seg) or one for the batch (n_sets == 1), an activation flag per
// (sample, hole), and the hole value  filler[b % n_fill_seg, f, t] * so[0] + so[1]  -- the layer's fixed noise tensor scaled to the
// batch's value range ('random': so = {max - min, min}), as drawn ('random_with_range': {1, 0}), or a constant (filler == NULL
// reads as 1: 'min' = {mean, 0}, 'zeros' = {0, 0}).
constexpr int MAX_RECTS_EX = 32;
// h * scale + offset with TWO roundings, like the reference's `hf * (max - min) + min` (no contraction into an fma)
__device__ __forceinline__ float scale_offset_2r(float h, float scale, float offset) {
#pragma clang fp contract(off)
    const float p = h * scale;
    return p + offset;
}
__global__ __launch_bounds__(256) void specaug_general_kernel(float* __restrict__ x, int64_t n_vec4, int F, int T,
                                                              const int* __restrict__ rects, int n_rects, int64_t n_sets,
                                                              const unsigned char* __restrict__ active,
                                                              const float* __restrict__ filler, int64_t n_fill_seg,
                                                              const float* __restrict__ so) {
    const float scale = so[0], offset = so[1];
    const int per_seg4 = F * T / 4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n_vec4; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_seg4;
        const int e = (int)(i - b * per_seg4) * 4;
        const int f = e / T, t = e - f * T;              // T % 4 == 0: the 4 elements share f
        const int* rs = rects + (n_sets == 1 ? 0 : b * n_rects * 4);
        const unsigned char* ac = active ? active + b * n_rects : nullptr;
        bool hole[4] = {false, false, false, false};
        for (int k = 0; k < n_rects; ++k) {
            if (ac && !ac[k]) continue;
            const int f0 = rs[4 * k], f1 = rs[4 * k + 1], t0 = rs[4 * k + 2], t1 = rs[4 * k + 3];
            if (f < f0 || f > f1) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) hole[j] |= (t + j >= t0 && t + j <= t1);
        }
        if (hole[0] | hole[1] | hole[2] | hole[3]) {
            float4 v = ((float4*)x)[i];
            float4 h = make_float4(1.f, 1.f, 1.f, 1.f);
            if (filler) h = ((const float4*)filler)[(b % n_fill_seg) * per_seg4 + (i - b * per_seg4)];
            v.x = hole[0] ? scale_offset_2r(h.x, scale, offset) : v.x; v.y = hole[1] ? scale_offset_2r(h.y, scale, offset) : v.y;
            v.z = hole[2] ? scale_offset_2r(h.z, scale, offset) : v.z; v.w = hole[3] ? scale_offset_2r(h.w, scale, offset) : v.w;
            ((float4*)x)[i] = v;
        }
    }
}

// {max - min, min} of n floats (the 'random' filler's scale and offset, ncutout_tarray.py:207-208): partial extrema per
// workgroup, then one workgroup.  NaN-free inputs assumed (a NaN element is ignored by fminf / fmaxf, as by the masking itself).
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float v = x[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    for (int o = 32; o; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    __shared__ float red[8];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lo; red[4 + (threadIdx.x >> 6)] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
        part[2 * blockIdx.x + 1] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    }
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* __restrict__ part, int n_part, float* __restrict__ out) {
    float lo = threadIdx.x < n_part ? part[2 * threadIdx.x] : INFINITY, hi = threadIdx.x < n_part ? part[2 * threadIdx.x + 1] : -INFINITY;
    for (int o = 32; o; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    __shared__ float red[8];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lo; red[4 + (threadIdx.x >> 6)] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(red[0], red[1]), fminf(red[2], red[3])); hi = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        out[0] = hi - lo; out[1] = lo;
    }
}

// mean of n floats into *out, deterministic: 256 partial sums (double) in a fixed order, then one workgroup.
__global__ __launch_bounds__(256) void mean_partial_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ part) {
    double acc = 0.0;
    const int64_t n4 = n / 4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = ((const float4*)x)[i];
        acc += (double)((v.x + v.y) + (v.z + v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) acc += (double)x[4 * n4 + threadIdx.x];
    acc = wave_sum(acc);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void mean_final_kernel(const double* __restrict__ part, int n_part, int64_t n,
                                                         float* __restrict__ out) {
    double acc = threadIdx.x < n_part ? part[threadIdx.x] : 0.0;
    acc = wave_sum(acc);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
}

}  // namespace nafp

using namespace nafp;

static int specaug_launch(float* feat, int64_t n_seg, int F, int T, const nafp_rect* rects_host, int n_rects,
                          const unsigned char* active, float fill_value, const float* fill_dev, void* stream);

extern "C" int64_t nafp_specaug_mean_workspace_bytes(void) { return 256 * (int64_t)sizeof(double) + 256; }

extern "C" int nafp_specaug_mean(const float* feat, int64_t n, float* mean_out, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    if (!feat || !mean_out || !workspace || n <= 0 || ((uintptr_t)feat & 15)) return NAFP_ERR_INVALID_ARG;
    if (workspace_bytes < nafp_specaug_mean_workspace_bytes()) return NAFP_ERR_WORKSPACE;
    double* part = (double*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    mean_partial_kernel<<<256, 256, 0, (hipStream_t)stream>>>(feat, n, part);
    NAFP_LAUNCH_CHECK();
    mean_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(part, 256, n, mean_out);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

extern "C" int nafp_specaug_range(const float* feat, int64_t n, float* scale_offset_out, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
    if (!feat || !scale_offset_out || !workspace || n <= 0) return NAFP_ERR_INVALID_ARG;
    if (workspace_bytes < nafp_specaug_mean_workspace_bytes()) return NAFP_ERR_WORKSPACE;
    float* part = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    minmax_partial_kernel<<<256, 256, 0, (hipStream_t)stream>>>(feat, n, part);
    NAFP_LAUNCH_CHECK();
    minmax_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(part, 256, scale_offset_out);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

extern "C" int nafp_specaug_apply_ex(float* feat, int64_t n_seg, int F, int T, const int32_t* rects_dev, int n_rects, int64_t n_sets,
                                     const unsigned char* active_dev, const float* filler, int64_t n_fill_seg,
                                     const float* scale_offset_dev, void* stream) {
    if (!feat || n_seg < 0 || F <= 0 || T <= 0 || n_rects < 0 || (n_rects > 0 && !rects_dev) || !scale_offset_dev) return NAFP_ERR_INVALID_ARG;
    if (n_sets != 1 && n_sets != n_seg) return NAFP_ERR_INVALID_ARG;
    if (filler && (n_fill_seg <= 0 || ((uintptr_t)filler & 15))) return NAFP_ERR_INVALID_ARG;
    if (n_rects > MAX_RECTS_EX || (T % 4) != 0 || ((uintptr_t)feat & 15)) return NAFP_ERR_UNSUPPORTED;
    if (n_seg == 0 || n_rects == 0) return NAFP_OK;
    const int64_t n_vec4 = n_seg * F * T / 4;
    const int blocks = (int)std::min<int64_t>((n_vec4 + 255) / 256, 2048);
    specaug_general_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(feat, n_vec4, F, T, (const int*)rects_dev, n_rects, n_sets, active_dev,
                                                                     filler, n_fill_seg, scale_offset_dev);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

extern "C" int nafp_specaug_apply_fill_dev(float* feat, int64_t n_seg, int F, int T, const nafp_rect* rects_host,
                                           int n_rects, const unsigned char* active, const float* fill_value_dev,
                                           void* stream) {
    if (!fill_value_dev) return NAFP_ERR_INVALID_ARG;
    return specaug_launch(feat, n_seg, F, T, rects_host, n_rects, active, 0.f, fill_value_dev, stream);
}

extern "C" int nafp_specaug_apply(float* feat, int64_t n_seg, int F, int T, const nafp_rect* rects_host,
                                  int n_rects, const unsigned char* active, float fill_value, void* stream) {
    return specaug_launch(feat, n_seg, F, T, rects_host, n_rects, active, fill_value, nullptr, stream);
}

static int specaug_launch(float* feat, int64_t n_seg, int F, int T, const nafp_rect* rects_host, int n_rects,
                          const unsigned char* active, float fill_value, const float* fill_dev, void* stream) {
    if (!feat || n_seg < 0 || F <= 0 || T <= 0 || n_rects < 0 || (n_rects > 0 && !rects_host))
        return NAFP_ERR_INVALID_ARG;
    if (n_rects > MAX_RECTS || (T % 4) != 0) return NAFP_ERR_UNSUPPORTED;
    if (n_seg == 0 || n_rects == 0) return NAFP_OK;
    RectTable r; r.n = n_rects;
    for (int k = 0; k < n_rects; ++k) {
        r.f0[k] = rects_host[k].f0; r.f1[k] = rects_host[k].f1; r.t0[k] = rects_host[k].t0; r.t1[k] = rects_host[k].t1;
    }
    const int64_t n_vec4 = n_seg * F * T / 4;
    const int blocks = (int)std::min<int64_t>((n_vec4 + 255) / 256, 2048);
    specaug_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(feat, n_vec4, F, T, r, active, fill_value, fill_dev);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}
