// Online triplet loss of the now-playing baseline (model/fp/online_triplet_loss.py:185-239), forward and
// gradient, gfx950.  The problem is tiny (n_anchor x (n_pos + n_anchor) distances at d = 128: 64 x 320 in
// config/now_playing.yaml), so it is three small VALU kernels, not an MFMA kernel:
//   triplet_rows_kernel   one workgroup per anchor a: dot(a, column m) for every column of [pos ; anc],
//                         d = sqrt(2(1-dot) [2(1-dot) > 0] + 1e-9); hardest positive (max over the anchor's own
//                         replicas); per-column loss term and its coefficient C[a,m] = dLoss/d dot(a,m)
//   triplet_grad_anchor   dA[a] = sum_m C[a,m] col_m  (+ the anchor's role as a column: sum_a' C[a', nP + a] anc_a')
//   triplet_grad_pos      dP[m] = sum_a C[a,m] anc_a
#include "nafp_common.h"

namespace nafp {

constexpr float TRIPLET_EPS = 1e-9f;

__global__ __launch_bounds__(256) void triplet_rows_kernel(const float* __restrict__ anc, const float* __restrict__ pos,
                                                           float* __restrict__ dist, float* __restrict__ coef,
                                                           float* __restrict__ loss_sum, int nA, int npa, int D, int mode,
                                                           float margin) {
    extern __shared__ float sh[];                   // anchor row [D] | dist row [M] | red[8]
    const int a = blockIdx.x, tid = threadIdx.x, nP = nA * npa, M = nP + nA;
    float* arow = sh; float* drow = sh + D; float* red = drow + M;
    for (int c = tid; c < D; c += 256) arow[c] = anc[(int64_t)a * D + c];
    __syncthreads();
    for (int m = tid; m < M; m += 256) {
        const float* col = m < nP ? pos + (int64_t)m * D : anc + (int64_t)(m - nP) * D;
        float dot = 0.f;
        for (int c = 0; c < D; c += 4) {
            const float4 v = *(const float4*)(col + c);
            dot = fmaf(arow[c], v.x, fmaf(arow[c + 1], v.y, fmaf(arow[c + 2], v.z, fmaf(arow[c + 3], v.w, dot))));
        }
        const float d2 = 2.f * (1.f - dot);
        const float dd = sqrtf((d2 > 0.f ? d2 : 0.f) + TRIPLET_EPS);
        drow[m] = dd;
        if (dist) dist[(int64_t)a * M + m] = dd;
    }
    __syncthreads();
    // hardest positive of this anchor: max over its own replicas (the masked matrix is 0 elsewhere, d > 0)
    float hard = 0.f; int hard_m = a * npa;
    if (mode == 0) {
        for (int k = 0; k < npa; ++k) { const float v = drow[a * npa + k]; if (v > hard) { hard = v; hard_m = a * npa + k; } }
    }
    if (mode >= 2) {
        // 'all-balanced' (mode 2, online_triplet_loss.py:215-222): max(mean_pos d - mean_neg d + margin, 0) per anchor;
        // 'hardest' (mode 3, :223-227): max(max_pos d - min(d * an_mask) + margin, 0) per anchor.  The reference takes
        // the min over the MASKED matrix, whose entries at the anchor's own replicas and at the anchor itself are 0
        // (and every distance is > 0), so its "hardest negative" is always 0: mirrored as written.
        // Both reduce to one term per anchor, averaged over the anchors.
        float sp = 0.f, sn = 0.f, mx = 0.f;
        for (int m = tid; m < M; m += 256) {
            const bool is_pos = m >= a * npa && m < (a + 1) * npa;
            const bool is_neg = !is_pos && m != nP + a;
            const float dd = drow[m];
            if (is_pos) { sp += dd; mx = fmaxf(mx, dd); }
            if (is_neg) sn += dd;
        }
        sp = wave_sum(sp); sn = wave_sum(sn);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((tid & 63) == 0) { red[tid >> 6] = sp; red[4 + (tid >> 6)] = sn; }
        __syncthreads();
        sp = (red[0] + red[1]) + (red[2] + red[3]); sn = (red[4] + red[5]) + (red[6] + red[7]);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const int n_an = M - npa - 1;
        const float inv_a = 1.f / (float)nA;
        const float t = mode == 2 ? sp / (float)npa - sn / (float)n_an + margin : mx - 0.f + margin;
        const bool active = t > 0.f;
        // hardest positive = FIRST maximum along the row (the sub-gradient tf.reduce_max / torch.max hand out)
        int hard_m = a * npa;
        for (int k = 0; k < npa; ++k) if (drow[a * npa + k] == mx) { hard_m = a * npa + k; break; }
        for (int m = tid; m < M; m += 256) {
            const bool is_pos = m >= a * npa && m < (a + 1) * npa;
            const bool is_neg = !is_pos && m != nP + a;
            const float dd = drow[m];
            float dterm_dd = 0.f;
            if (active) {
                if (mode == 2) dterm_dd = is_pos ? 1.f / (float)npa : (is_neg ? -1.f / (float)n_an : 0.f);
                else dterm_dd = m == hard_m ? 1.f : 0.f;
            }
            const float d2pos = dd * dd - TRIPLET_EPS > 0.f ? 1.f : 0.f;
            coef[(int64_t)a * M + m] = inv_a * dterm_dd * (-d2pos / dd);
        }
        if (tid == 0 && active) atomicAdd(loss_sum, t * inv_a);
        return;
    }
    const float inv = 1.f / ((float)nA * (float)M);
    float lsum = 0.f, n_active = 0.f;
    for (int m = tid; m < M; m += 256) {
        const bool is_pos = m >= a * npa && m < (a + 1) * npa;
        const bool is_neg = !is_pos && m != nP + a;            // an_mask: everything but the own replicas and the anchor itself
        const float dd = drow[m];
        float term = 0.f, dterm_dd = 0.f;                       // d(term)/d(d[a,m]) (direct dependence)
        if (mode == 0) {                                        // semi-hard: max((hardest - d + margin) * an, 0)
            if (is_neg) { term = hard - dd + margin; if (term > 0.f) { dterm_dd = -1.f; n_active += 1.f; } else term = 0.f; }
        } else {                                                // all: max(d*ap - d*an + margin, 0) element by element
            const float t = (is_pos ? dd : 0.f) - (is_neg ? dd : 0.f) + margin;
            if (t > 0.f) { term = t; dterm_dd = is_pos ? 1.f : (is_neg ? -1.f : 0.f); }
        }
        lsum += term;
        // d(dd)/d(dot) = -[2(1-dot) > 0] / dd
        const float d2pos = dd * dd - TRIPLET_EPS > 0.f ? 1.f : 0.f;
        coef[(int64_t)a * M + m] = inv * dterm_dd * (-d2pos / dd);
    }
    lsum = wave_sum(lsum); n_active = wave_sum(n_active);
    if ((tid & 63) == 0) { red[tid >> 6] = lsum; red[4 + (tid >> 6)] = n_active; }
    __syncthreads();
    if (tid == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        const float act = (red[4] + red[5]) + (red[6] + red[7]);
        atomicAdd(loss_sum, tot * inv);
        if (mode == 0 && act > 0.f) {
            // the hardest positive carries the gradient of every active term of this row
            const float dd = drow[hard_m];
            const float d2pos = dd * dd - TRIPLET_EPS > 0.f ? 1.f : 0.f;
            coef[(int64_t)a * M + hard_m] += inv * act * (-d2pos / dd);
        }
    }
}

// out[r, :] = sum_m C[r, m] * cols[m, :]   (r over nR rows; cols = [pos ; anc])
__global__ __launch_bounds__(256) void triplet_grad_anchor_kernel(const float* __restrict__ coef, const float* __restrict__ anc,
                                                                  const float* __restrict__ pos, float* __restrict__ d_anc,
                                                                  int nA, int npa, int D) {
    const int a = blockIdx.x, c = threadIdx.x, nP = nA * npa, M = nP + nA;
    if (c >= D) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) {
        const float w = coef[(int64_t)a * M + m];
        if (w != 0.f) s = fmaf(w, m < nP ? pos[(int64_t)m * D + c] : anc[(int64_t)(m - nP) * D + c], s);
    }
    for (int r = 0; r < nA; ++r) {                                // this anchor as column nP + a of every row r
        const float w = coef[(int64_t)r * M + nP + a];
        if (w != 0.f) s = fmaf(w, anc[(int64_t)r * D + c], s);
    }
    d_anc[(int64_t)a * D + c] = s;
}

__global__ __launch_bounds__(256) void triplet_grad_pos_kernel(const float* __restrict__ coef, const float* __restrict__ anc,
                                                               float* __restrict__ d_pos, int nA, int npa, int D) {
    const int m = blockIdx.x, c = threadIdx.x, M = nA * npa + nA;
    if (c >= D) return;
    float s = 0.f;
    for (int a = 0; a < nA; ++a) {
        const float w = coef[(int64_t)a * M + m];
        if (w != 0.f) s = fmaf(w, anc[(int64_t)a * D + c], s);
    }
    d_pos[(int64_t)m * D + c] = s;
}

}  // namespace nafp

using namespace nafp;

extern "C" int64_t nafp_triplet_workspace_bytes(int64_t n_anchor, int64_t n_pos) {
    if (n_anchor <= 0 || n_pos < 0) return -1;
    return n_anchor * (n_pos + n_anchor) * (int64_t)sizeof(float) + 256;
}

extern "C" int nafp_triplet_forward(const float* emb_anchor, const float* emb_pos, int64_t n_anchor, int64_t n_pos, int dim,
                                    int mode, float margin, float* loss_out, float* pairwise_dist, float* d_anchor,
                                    float* d_pos, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!emb_anchor || !emb_pos || !loss_out || !workspace || n_anchor <= 0 || n_pos <= 0 || dim <= 0) return NAFP_ERR_INVALID_ARG;
    if (n_pos % n_anchor != 0 || dim % 4 != 0 || dim > 256 || mode < 0 || mode > 3 || n_anchor + n_pos > 8192)
        return NAFP_ERR_UNSUPPORTED;
    if (mode == 2 && n_anchor == 1) return NAFP_ERR_UNSUPPORTED;      // no negatives: the reference divides 0 by 0
    if (workspace_bytes < nafp_triplet_workspace_bytes(n_anchor, n_pos)) return NAFP_ERR_WORKSPACE;
    if ((d_anchor == nullptr) != (d_pos == nullptr)) return NAFP_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* coef = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const int nA = (int)n_anchor, npa = (int)(n_pos / n_anchor), M = (int)(n_pos + n_anchor);
    NAFP_HIP_CHECK(hipMemsetAsync(loss_out, 0, sizeof(float), st));
    const int lds = (dim + M + 8) * (int)sizeof(float);
    triplet_rows_kernel<<<nA, 256, lds, st>>>(emb_anchor, emb_pos, pairwise_dist, coef, loss_out, nA, npa, dim, mode, margin);
    NAFP_LAUNCH_CHECK();
    if (d_anchor) {
        triplet_grad_anchor_kernel<<<nA, 256, 0, st>>>(coef, emb_anchor, emb_pos, d_anchor, nA, npa, dim);
        NAFP_LAUNCH_CHECK();
        triplet_grad_pos_kernel<<<(unsigned)n_pos, 256, 0, st>>>(coef, emb_anchor, d_pos, nA, npa, dim);
        NAFP_LAUNCH_CHECK();
    }
    return NAFP_OK;
}
