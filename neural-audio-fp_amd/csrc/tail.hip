// Encoder tail for gfx950: LayerNorm of the last conv, Flatten, divide-and-encode
// (Q slices x [Dense S->32, ELU, Dense 32->1]) and L2 normalisation, in one kernel.
//
// Replaces, in the reference: the last LayerNormalization + Flatten of front_conv
// (model/fp/nnfp.py:66-67, 218), DivEncLayer (nnfp.py:131-156; the BN list built at
// :119-122 is never applied, :136 is commented out) and tf.math.l2_normalize
// (nnfp.py:229).  In TF this is 256 tiny matmuls + a concat; here one workgroup of
// Q threads handles one segment, thread q owning slice q.  Slice weights are
// re-laid out (S,32,Q)/(32,Q) at set_weights time so that thread q's reads coalesce.
#include "nafp_common.h"

namespace nafp {

constexpr int MAX_S = 16;

__global__ __launch_bounds__(1024) void tail_kernel(const TailArgs a) {
    const int q = threadIdx.x, Q = a.Q, S = a.S;
    const int64_t b = blockIdx.x;
    float lnA = 1.f, lnC = 0.f;
    if (a.stats) {
        const double mean = a.stats[2 * b] / (double)a.D;
        double var = a.stats[2 * b + 1] / (double)a.D - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double rstd = 1.0 / sqrt(var + (double)LN_EPS);
        lnA = (float)rstd; lnC = (float)(-mean * rstd);
    }
    float x[MAX_S];
#pragma unroll
    for (int i = 0; i < MAX_S; ++i) {
        if (i < S) {
            const int d = q * S + i;                      // tf.reshape (B,D)->(B,Q,S): nnfp.py:155
            float v = a.x[b * a.D + d];
            if (a.stats) v = fmaf(lnA, v, fmaf(lnC, a.gamma[d], a.beta[d]));   // v holds gamma . ELU(.)
            x[i] = v;
            if (a.out_flat) a.out_flat[b * a.D + d] = v;
        } else {
            x[i] = 0.f;
        }
    }
    if (!a.out_emb) return;
    float y = a.b2[q];
    for (int j = 0; j < 32; ++j) {
        float h = a.b1p[j * Q + q];
#pragma unroll
        for (int i = 0; i < MAX_S; ++i)
            if (i < S) h = fmaf(x[i], a.w1p[(i * 32 + j) * Q + q], h);
        y = fmaf(elu1(h), a.w2p[j * Q + q], y);           // Dense(32, elu) -> Dense(1): nnfp.py:135-137
    }
    if (a.l2norm) {
        __shared__ float red[16];
        const float ss = wave_sum(y * y);
        if ((q & 63) == 0) red[q >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < (Q + 63) / 64; ++w) tot += red[w];
        y = y * rsqrtf(fmaxf(tot, 1e-12f));               // tf.math.l2_normalize: nnfp.py:229
    }
    a.out_emb[b * Q + q] = y;
}

int launch_tail(const TailArgs& a, int64_t B, hipStream_t st) {
    if (a.Q % 64 != 0 || a.Q > 1024 || a.S > MAX_S || a.S * a.Q != a.D) return NAFP_ERR_UNSUPPORTED;
    if (B == 0) return NAFP_OK;
    tail_kernel<<<dim3((unsigned)B), a.Q, 0, st>>>(a);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// (Q,S,32) -> (S,32,Q); (Q,32) -> (32,Q) for b1 and w2.
__global__ void pack_div_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                const float* __restrict__ w2, float* __restrict__ w1p,
                                float* __restrict__ b1p, float* __restrict__ w2p, int Q, int S) {
    const int n1 = Q * S * 32, n2 = Q * 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += gridDim.x * blockDim.x) {
        const int q = i / (S * 32), r = i % (S * 32);     // r = s*32 + j
        w1p[r * Q + q] = w1[i];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x) {
        const int q = i / 32, j = i % 32;
        b1p[j * Q + q] = b1[i];
        w2p[j * Q + q] = w2[i];
    }
}

int launch_pack_div(const float* w1, const float* b1, const float* w2, float* w1p, float* b1p,
                    float* w2p, int Q, int S, hipStream_t st) {
    pack_div_kernel<<<64, 256, 0, st>>>(w1, b1, w2, w1p, b1p, w2p, Q, S);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

}  // namespace nafp
