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
