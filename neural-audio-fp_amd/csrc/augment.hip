// Time-domain augmentation of a training batch on the device, gfx950.
//
// Replaces the host loops of the reference's training loader -- load_audio per segment,
// bg_mix_batch, ir_aug_batch (model/utils/audio_utils.py:28-137, 221-264) as called from
// genUnbalSequence.__getitem__ (model/utils/dataloader_keras.py:223-311) -- which SURVEY.md names the
// true bottleneck of reference training.  One workgroup per output row; the row lives in LDS.
//   x   = event window / 2^15                      (zero tail: load_audio's padding)
//   nz  = background window (+ speech window) / 2^15
//   mix : max|x| == 0 or max|nz| == 0 ? x + nz : 10^(snr/20) x / rms(x) + nz / rms(nz)
//         -> max-normalise -> * amp                 (background_mix, bg_mix_batch)
//   ir  : circular convolution of length T with the <= 600-tap impulse response, summed directly
//         (4 outputs x 4 taps per step from three aligned LDS float4 reads) == the reference's
//         ifft(fft(ir, T) * fft(x, T)) (ir_aug_batch), then max-normalise.
// Rows without noise / IR windows (the anchors) are plain int16 -> float conversions, so ONE launch
// produces the whole (anchors | replicas) batch that feeds the log-mel kernel.
#include "nafp_common.h"

namespace nafp {

__device__ __forceinline__ float block_max(float v, float* red, int tid) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int MAX_IR = 608;       // MAX_IR_LENGTH = 600 (dataloader_keras.py:8), padded to a multiple of 4 (+4 guard)

__global__ __launch_bounds__(256) void augment_rows_kernel(const int16_t* __restrict__ pcm, const nafp_aug_row* __restrict__ rows,
                                                           float* __restrict__ out, int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // x[T] | y[T] | ir[MAX_IR]
    __shared__ float red[4];
    float* x = smem; float* y = smem + T; float* ir = smem + 2 * T;
    const int tid = threadIdx.x;
    const nafp_aug_row r = rows[blockIdx.x];
    const float sc = 1.f / 32768.f;
    const int16_t* pe = pcm + r.ev_off;
    for (int i = tid; i < T; i += 256) x[i] = i < r.ev_valid ? (float)pe[i] * sc : 0.f;
    float* cur = x;
    if (r.mix) {
        const int16_t* p1 = pcm + (r.nz_off >= 0 ? r.nz_off : 0);
        const int16_t* p2 = pcm + (r.nz2_off >= 0 ? r.nz2_off : 0);
        const int v1 = r.nz_off >= 0 ? r.nz_valid : 0, v2 = r.nz2_off >= 0 ? r.nz2_valid : 0;
        float xm = 0.f, nm = 0.f, xs = 0.f, ns = 0.f;
        for (int i = tid; i < T; i += 256) {
            const float n = (i < v1 ? (float)p1[i] * sc : 0.f) + (i < v2 ? (float)p2[i] * sc : 0.f);
            y[i] = n;
            const float xv = x[i];
            xm = fmaxf(xm, fabsf(xv)); nm = fmaxf(nm, fabsf(n));
            xs = fmaf(xv, xv, xs); ns = fmaf(n, n, ns);
        }
        xm = block_max(xm, red, tid); nm = block_max(nm, red, tid);
        xs = block_sum(xs, red, tid); ns = block_sum(ns, red, tid);
        float a = 1.f, b = 1.f;
        if (xm != 0.f && nm != 0.f) {
            a = exp10f(r.snr_db / 20.f) / sqrtf(xs / (float)T);
            b = 1.f / sqrtf(ns / (float)T);
        }
        float mm = 0.f;
        for (int i = tid; i < T; i += 256) { const float m = a * x[i] + b * y[i]; x[i] = m; mm = fmaxf(mm, fabsf(m)); }
        mm = block_max(mm, red, tid);
        const float g = (mm != 0.f ? 1.f / mm : 1.f) * r.amp;
        for (int i = tid; i < T; i += 256) x[i] *= g;
    }
    __syncthreads();
    if (r.ir_off >= 0 && r.ir_len > 0) {
        const int L = min(r.ir_len, MAX_IR - 8);
        const int L4 = (L + 3) & ~3;
        const int16_t* pi = pcm + r.ir_off;
        for (int i = tid; i < MAX_IR; i += 256) ir[i] = i < L ? (float)pi[i] * sc : 0.f;
        __syncthreads();
        // y[n] = sum_k ir[k] x[(n - k) mod T]; thread <-> 4 consecutive outputs, 4 taps per step.
        // T % 4 == 0, so (n0 - k) mod T stays 4-aligned and the two x reads are aligned float4.
        float ym = 0.f;
        for (int n0 = 4 * tid; n0 < T; n0 += 1024) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int p = n0;                                   // (n0 - k) mod T for the current k
            for (int k = 0; k < L4; k += 4) {
                const float4 h = *(const float4*)(ir + k);
                const float4 hi = *(const float4*)(x + p);               // x[p .. p+3]   = x[n0-k+j]
                int pl = p - 4; if (pl < 0) pl += T;
                const float4 lo = *(const float4*)(x + pl);              // x[p-4 .. p-1]
                // output j uses x[p + j - t] for tap k + t
                acc.x = fmaf(h.x, hi.x, fmaf(h.y, lo.w, fmaf(h.z, lo.z, fmaf(h.w, lo.y, acc.x))));
                acc.y = fmaf(h.x, hi.y, fmaf(h.y, hi.x, fmaf(h.z, lo.w, fmaf(h.w, lo.z, acc.y))));
                acc.z = fmaf(h.x, hi.z, fmaf(h.y, hi.y, fmaf(h.z, hi.x, fmaf(h.w, lo.w, acc.z))));
                acc.w = fmaf(h.x, hi.w, fmaf(h.y, hi.z, fmaf(h.z, hi.y, fmaf(h.w, hi.x, acc.w))));
                p = pl;
            }
            *(float4*)(y + n0) = acc;
            ym = fmaxf(fmaxf(ym, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
        }
        ym = block_max(ym, red, tid);
        const float g = ym != 0.f ? 1.f / ym : 1.f;
        for (int i = tid; i < T; i += 256) y[i] *= g;
        __syncthreads();
        cur = y;
    }
    float* o = out + (int64_t)blockIdx.x * T;
    for (int i = tid; i < T / 4; i += 256) ((float4*)o)[i] = ((const float4*)cur)[i];
}

}  // namespace nafp

using namespace nafp;

extern "C" int nafp_augment_rows(const int16_t* pcm, const nafp_aug_row* rows, int64_t n_rows, int seg_len, float* out,
                                 void* stream) {
    if (!pcm || !rows || !out || n_rows < 0 || seg_len <= 0) return NAFP_ERR_INVALID_ARG;
    if (seg_len % 4 != 0 || seg_len > 19000) return NAFP_ERR_UNSUPPORTED;        // x | y | ir must fit the 160 KB LDS
    if (n_rows == 0) return NAFP_OK;
    const int lds = (2 * seg_len + MAX_IR) * (int)sizeof(float);
    NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)augment_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    augment_rows_kernel<<<dim3((unsigned)n_rows), 256, lds, (hipStream_t)stream>>>(pcm, rows, out, seg_len);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}
