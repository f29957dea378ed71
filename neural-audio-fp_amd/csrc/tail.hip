// Encoder tail for gfx950: LayerNorm of the last conv, Flatten, divide-and-encode
// (Q slices x [Dense S->32, ELU, Dense 32->1]) and L2 normalisation, in one kernel.
//
// Replaces, in the reference: the last LayerNormalization + Flatten of front_conv
// (model/fp/nnfp.py:66-67, 218), DivEncLayer (nnfp.py:131-156; the BN list built at
// :119-122 is never applied, :136 is commented out) and tf.math.l2_normalize
// (nnfp.py:229).  In TF this is 256 tiny matmuls + a concat.
//
// A workgroup is 2*Q threads: thread (q, half) owns 16 of the 32 hidden units of slice q
// and keeps their weights in registers (S*16 + 32 floats), then loops over segments
// (grid-stride).  Slice weights are re-laid out (S,32,Q)/(32,Q) at set_weights time so
// that the one-time register fill coalesces.  Per segment: S inputs per thread (LayerNorm
// folded: the last conv stored z = gamma . v, so xhat = r*z + c*gamma + beta), 16 hidden
// units, the halves meet in LDS, then the L2 norm over the Q outputs.
#include "nafp_common.h"

#include <algorithm>
#include <cstdlib>

namespace nafp {

constexpr int MAX_S = 16;
constexpr int HALF_H = 16;

template <int S>
__global__ __launch_bounds__(512) void tail_kernel(const TailArgs a, int64_t B) {   // 2*Q <= 512 threads: 256 VGPRs each
    const int Q = a.Q;
    const int tid = threadIdx.x;
    const int q = tid % Q, half = tid / Q;            // blockDim.x == 2*Q
    extern __shared__ float sm[];                       // [Q] half-1 partials, then [16] reduction scratch
    float* s_part = sm;
    float* s_red = sm + Q;

    float w1[S][HALF_H], b1[HALF_H], w2[HALF_H];
#pragma unroll
    for (int j = 0; j < HALF_H; ++j) {
        const int jj = half * HALF_H + j;
        b1[j] = a.b1p[jj * Q + q];
        w2[j] = a.w2p[jj * Q + q];
#pragma unroll
        for (int i = 0; i < S; ++i) w1[i][j] = a.w1p[(i * 32 + jj) * Q + q];
    }
    const float b2 = a.b2[q];
    float gam[S], bet[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        gam[i] = a.stats ? a.gamma[q * S + i] : 0.f;
        bet[i] = a.stats ? a.beta[q * S + i] : 0.f;
    }

    // a parameter set with a NaN / Inf in it (set_weights found one): every row is NaN, as with keras
    const bool bad_weights = (a.nonfinite_weights && *a.nonfinite_weights != 0) || (a.launch_error && *a.launch_error != 0);
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        float lnA = 1.f, lnC = 0.f;
        if (a.stats) stat_ln_scalars(a.stats + 2 * b, a.ident_stats ? -1.0 : 1.0 / (double)a.D, &lnA, &lnC);     // NaN for a poisoned sample: its row comes out NaN
        if (bad_weights) lnA = __builtin_nanf("");
        float x[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int d = q * S + i;                    // tf.reshape (B,D)->(B,Q,S): nnfp.py:155
            float v = a.x[b * a.D + d];
            if (a.stats) v = fmaf(lnA, v, fmaf(lnC, gam[i], bet[i]));   // v held gamma . ELU(.)
            else if (bad_weights) v = lnA;
            x[i] = v;
            if (a.out_flat && half == 0) a.out_flat[b * a.D + d] = v;
        }
        if (!a.out_emb) continue;
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < HALF_H; ++j) {
            float h = b1[j];
#pragma unroll
            for (int i = 0; i < S; ++i) h = fmaf(x[i], w1[i][j], h);
            y = fmaf(elu1(h), w2[j], y);                // Dense(32, elu) -> Dense(1): nnfp.py:135-137
        }
        __syncthreads();                                // previous segment's readers are done with s_part/s_red
        if (half == 1) s_part[q] = y;
        __syncthreads();
        if (half == 0) {
            y = y + s_part[q] + b2;
            if (a.l2norm) {
                const float ss = wave_sum(y * y);
                if ((q & 63) == 0) s_red[q >> 6] = ss;
            }
        }
        if (a.l2norm) {
            __syncthreads();
            if (half == 0) {
                float tot = 0.f;
                for (int w = 0; w < (Q + 63) / 64; ++w) tot += s_red[w];
                y = y * rsqrtf(fmaxf(tot, 1e-12f));     // tf.math.l2_normalize: nnfp.py:229
            }
        }
        if (half == 0) a.out_emb[b * Q + q] = y;
    }
}

int launch_tail(const TailArgs& a, int64_t B, hipStream_t st) {
    if (a.Q % 64 != 0 || a.Q > 256 || a.S > MAX_S || a.S * a.Q != a.D) return NAFP_ERR_UNSUPPORTED;
    if (B == 0) return NAFP_OK;
    // enough workgroups to cover the CUs, few enough that the register fill (S*16+32 floats
    // per thread, from L2) is amortised over several segments
    // (NAFP_TAIL_WGS sweep at B = 640, kernel us under rocprofv3: 128 -> 17.8, 256 -> 13.1, 320 -> 14.6, 640 -> 18.0)
    static const int64_t wg_env = []() { const char* e = getenv("NAFP_TAIL_WGS"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)256; }();
    const int grid = (int)std::min<int64_t>(B, wg_env);
    const size_t lds = (size_t)(a.Q + 16) * sizeof(float);
    switch (a.S) {
        case 8: tail_kernel<8><<<grid, 2 * a.Q, lds, st>>>(a, B); break;
        case 16: tail_kernel<16><<<grid, 2 * a.Q, lds, st>>>(a, B); break;
        case 4: tail_kernel<4><<<grid, 2 * a.Q, lds, st>>>(a, B); break;
        case 2: tail_kernel<2><<<grid, 2 * a.Q, lds, st>>>(a, B); break;
        case 1: tail_kernel<1><<<grid, 2 * a.Q, lds, st>>>(a, B); break;
        default: return NAFP_ERR_UNSUPPORTED;
    }
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// tf.math.l2_normalize(x, axis=1) of a (n_rows, dim) array (trainer.py:74, 76): one wave per row.
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int64_t n_rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* src = x + row * dim;
    float ss = 0.f;
    for (int c = lane; c < dim; c += 64) ss = fmaf(src[c], src[c], ss);
    const float inv = rsqrtf(fmaxf(wave_sum(ss), 1e-12f));
    for (int c = lane; c < dim; c += 64) y[row * dim + c] = src[c] * inv;
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

namespace nafp {
__global__ __launch_bounds__(256) void pack_embedding_grads_kernel(const float* __restrict__ da, const float* __restrict__ db,
                                                                   const float* __restrict__ loss_sum, float loss_scale,
                                                                   int64_t half, float* __restrict__ send) {
    const int64_t r = blockIdx.y, chunk = 2 * half + 4;
    float* o = send + r * chunk;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < chunk; i += (int64_t)gridDim.x * 256)
        o[i] = i < half ? da[r * half + i] : i < 2 * half ? db[r * half + (i - half)] : loss_sum[0] * loss_scale;
}
}  // namespace nafp

extern "C" int nafp_pack_embedding_grads(const float* d_a_all, const float* d_b_all, const float* loss_sum, float loss_scale,
                                         int64_t world, int64_t n_anchors, int dim, float* send, void* stream) {
    if (!d_a_all || !d_b_all || !loss_sum || !send || world <= 0 || n_anchors <= 0 || dim <= 0 || world > 65535) return NAFP_ERR_INVALID_ARG;
    const int64_t half = n_anchors * dim;
    const unsigned gx = (unsigned)std::min<int64_t>((2 * half + 4 + 255) / 256, 1024);
    nafp::pack_embedding_grads_kernel<<<dim3(gx, (unsigned)world), 256, 0, (hipStream_t)stream>>>(d_a_all, d_b_all, loss_sum, loss_scale, half, send);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

extern "C" int nafp_l2_normalize_rows(const float* x, int64_t n_rows, int dim, float* out, void* stream) {
    if (!x || !out || n_rows < 0 || dim <= 0) return NAFP_ERR_INVALID_ARG;
    if (n_rows == 0) return NAFP_OK;
    nafp::l2_normalize_rows_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, out, n_rows, dim);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}
