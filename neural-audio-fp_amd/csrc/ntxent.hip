// NT-Xent in-batch contrastive loss for gfx950, fused similarity + log-sum-exp.
//
// Replaces NTxentLoss.compute_loss (model/fp/NTxent_loss_single_gpu.py:52-82) and,
// with n_local < n_global, one replica of NTxentLoss.loss_fn
// (model/fp/NTxent_loss_tpu.py:90-137).
//
// Rows R = [emb_org_local ; emb_rep_local] (2*n_local), columns
// C = [emb_org_all ; emb_rep_all] (2*n_global).  For row (a, i) the reference's
// logits are [ab | aa without diagonal]: every column except (a, gi); the label is
// column (b, gi).  For row (b, i): every column except (b, gi); label (a, gi).
// So per row:   CE = LSE_{c != self(r)} S[r,c] - S[r, pos(r)],   S = R C^T / tau.
// The (2N)^2 logits are never materialised (105 MB at BSZ 5120) unless the caller
// asks for the (n, 2n-1) sim_mtx the reference's trainer logs once per epoch.
//
// One workgroup = 32 rows of R; its 4 waves stride over 32-column tiles of C.
// The product is computed TRANSPOSED (S^T tile = C_tile . R_blk^T) so that in the
// 32x32 MFMA C/D layout a lane owns ONE row of R (col = lane & 31) and its 16
// accumulators are 16 different columns: the running max / sum-exp are lane-local.
// K = d = 128: lane half h multiplies k in [64h, 64h+64) (same split for both
// operands), the row fragment of R stays in 64 registers for the whole kernel.
#include "nafp_common.h"

#include <algorithm>
#include <cmath>

namespace nafp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Embedding width ND (MODEL.EMB_SZ, nnfp.py:250): 64, 128 or 256.  Lane half h multiplies
// k in [ND/2 * h, ND/2 * h + ND/2).
template <int ND>
__global__ __launch_bounds__(256) void ntxent_fwd_kernel(
        const float* __restrict__ org_l, const float* __restrict__ rep_l,
        const float* __restrict__ org_all, const float* __restrict__ rep_all,
        int n_local, int n_global, int rank_offset, float tau,
        float* __restrict__ fpart, float* __restrict__ sim_mtx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, rl = lane & 31;
    const int n_rows = 2 * n_local, n_cols = 2 * n_global;
    const int r = blockIdx.x * 32 + rl;                    // my row of R
    const bool rvalid = r < n_rows;
    const bool r_is_b = r >= n_local;
    const int ri = r_is_b ? r - n_local : r;               // index inside the local half
    const int gi = rank_offset + ri;
    const int self_col = r_is_b ? n_global + gi : gi;
    const int pos_col = r_is_b ? gi : n_global + gi;

    constexpr int HK = ND / 2, NV = ND / 8;                // k per lane half; float4 loads per lane
    float rf[HK];                                           // R[r][HK*h .. HK*h+HK-1]
    {
        const float* src = rvalid ? ((r_is_b ? rep_l : org_l) + (int64_t)ri * ND + HK * h) : nullptr;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 t = rvalid ? *(const float4*)(src + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            rf[4 * v] = t.x; rf[4 * v + 1] = t.y; rf[4 * v + 2] = t.z; rf[4 * v + 3] = t.w;
        }
    }
    float run_m = -INFINITY, run_s = 0.f, pos_v = 0.f;
    const int n_tiles = (n_cols + 31) / 32;
    // blockIdx.y = column split: this workgroup's waves take tiles t = 4 y + wave, then + 4 gridDim.y (the sharded loss
    // has few row blocks -- 20 per rank at BSZ 5120 over 8 ranks -- and 160 column tiles: without the split 20 workgroups
    // walked all of them)
    for (int t = blockIdx.y * 4 + wave; t < n_tiles; t += 4 * gridDim.y) {
        const int c = t * 32 + rl;                          // column this lane feeds into the A operand
        const bool cvalid = c < n_cols;
        const float* csrc = nullptr;
        if (cvalid) csrc = (c >= n_global ? rep_all + (int64_t)(c - n_global) * ND : org_all + (int64_t)c * ND) + HK * h;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 cv = cvalid ? *(const float4*)(csrc + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv.x, rf[4 * v], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv.y, rf[4 * v + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv.z, rf[4 * v + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv.w, rf[4 * v + 3], acc, 0, 0, 0);
        }
        // D[i = column-in-tile][j = row]: lane holds j = lane & 31, i = (q&3) + 8*(q>>2) + 4*h
        float tmax = -INFINITY;
        float sv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cc = t * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            const float s = acc[q] / tau;            // `/ self.tau` (NTxent_loss_single_gpu.py:72-77)
            const bool use = cc < n_cols && cc != self_col;
            if (cc == pos_col) pos_v = s;
            sv[q] = use ? s : -INFINITY;
            tmax = fmaxf(tmax, sv[q]);
            if (sim_mtx && rvalid && !r_is_b && cc < n_cols && cc != self_col) {
                // [ab | aa without diagonal] (NTxent_loss_single_gpu.py:78-82)
                const int j = cc >= n_global ? cc - n_global : n_global + (cc < gi ? cc : cc - 1);
                sim_mtx[(int64_t)ri * (n_cols - 1) + j] = s;
            }
        }
        if (tmax > -INFINITY) {
            const float nm = fmaxf(run_m, tmax);
            float add = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) add += __expf(sv[q] - nm);     // exp(-inf) = 0
            run_s = run_s * __expf(run_m - nm) + add;
            run_m = nm;
        }
    }
    // combine lane halves, then the 4 waves
    {
        const float om = __shfl_xor(run_m, 32, 64), os = __shfl_xor(run_s, 32, 64);
        const float op = __shfl_xor(pos_v, 32, 64);
        const float nm = fmaxf(run_m, om);
        const float a = run_m > -INFINITY ? run_s * __expf(run_m - nm) : 0.f;
        const float b = om > -INFINITY ? os * __expf(om - nm) : 0.f;
        run_s = a + b; run_m = nm; pos_v += op;               // pos_v is non-zero in exactly one place
    }
    __shared__ float sm[4][32], ss[4][32], sp[4][32];
    if (h == 0) { sm[wave][rl] = run_m; ss[wave][rl] = run_s; sp[wave][rl] = pos_v; }
    __syncthreads();
    if (tid < 32) {
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) m = fmaxf(m, sm[w][tid]);
        float s = 0.f, pv = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (sm[w][tid] > -INFINITY) s += ss[w][tid] * __expf(sm[w][tid] - m);
            pv += sp[w][tid];
        }
        const int rr = blockIdx.x * 32 + tid;
        if (rr < n_rows) {                                   // this split's (max, sum exp, positive logit): merged by ntxent_merge_kernel
            float* o = fpart + ((int64_t)blockIdx.y * n_rows + rr) * 3;
            o[0] = m; o[1] = s; o[2] = pv;
        }
    }
}

// Per row: merge the column splits' (max, sum exp, positive) in split order, lse and loss; then the sum of the row losses.
// Single workgroup, fixed order: deterministic.
__global__ void ntxent_merge_kernel(const float* __restrict__ fpart, int n_rows, int n_split,
                                    float* __restrict__ row_lse, float* __restrict__ out) {
    double acc = 0.0;
    for (int r = threadIdx.x; r < n_rows; r += blockDim.x) {
        float m = -INFINITY;
        for (int k = 0; k < n_split; ++k) m = fmaxf(m, fpart[((int64_t)k * n_rows + r) * 3]);
        float ssum = 0.f, pv = 0.f;
        for (int k = 0; k < n_split; ++k) {
            const float* q = fpart + ((int64_t)k * n_rows + r) * 3;
            if (q[0] > -INFINITY) ssum += q[1] * __expf(q[0] - m);
            pv += q[2];
        }
        const float lse = m + __logf(ssum);
        row_lse[r] = lse;
        acc += (double)(lse - pv);
    }
    acc = wave_sum(acc);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (float)(red[0] + red[1] + red[2] + red[3]);
}

// [r5] The same merge, 32 lanes per row over as many workgroups as the rows need (the single workgroup above walked 5120
// rows x 32 splits by itself: 140 us of a 370 us loss at BSZ 5120), followed by a one-wave sum of the workgroups' partial
// losses.  Fixed trees and a fixed order: deterministic.
__global__ __launch_bounds__(256) void ntxent_merge_rows_kernel(const float* __restrict__ fpart, int n_rows, int n_split,
                                                                float* __restrict__ row_lse, double* __restrict__ wg_part) {
    // 8 rows per workgroup, 32 lanes per row: a lane takes the splits k = l, l + 32, ...; the lanes of a row meet through
    // xor-shuffles inside their half wave (a fixed tree: deterministic)
    const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int r = blockIdx.x * 8 + g;
    const bool rv = r < n_rows;
    float m = -INFINITY;
    if (rv) for (int k = l; k < n_split; k += 32) m = fmaxf(m, fpart[((int64_t)k * n_rows + r) * 3]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 32));
    float ssum = 0.f, pv = 0.f;
    if (rv) for (int k = l; k < n_split; k += 32) {
        const float* q = fpart + ((int64_t)k * n_rows + r) * 3;
        const float qm = q[0], qs = q[1], qp = q[2];
        ssum += qm > -INFINITY ? qs * __expf(qm - m) : 0.f;
        pv += qp;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { ssum += __shfl_xor(ssum, o, 32); pv += __shfl_xor(pv, o, 32); }
    __shared__ double red[8];
    if (l == 0) {
        double loss = 0.0;
        if (rv) {
            const float lse = m + __logf(ssum);
            row_lse[r] = lse;
            loss = (double)(lse - pv);
        }
        red[g] = loss;
    }
    __syncthreads();
    if (threadIdx.x == 0) wg_part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}
__global__ __launch_bounds__(64) void ntxent_loss_sum_kernel(const double* __restrict__ wg_part, int n, float* __restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) acc += wg_part[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) *out = (float)acc;
}

// ---------------------------------------------------------------------------------------
// Backward.  dS[r,c] = (softmax_r[c] - [c == pos(r)]) / n_global for c != self(r); then
//   dR_r = sum_c dS[r,c] C_c / tau     (rows: this rank's embeddings)
//   dC_c = sum_r dS[r,c] R_r / tau     (columns: the gathered embeddings)
// One kernel, two roles.  The OWNER set (32 per workgroup, one per lane) is the rows (ROW
// mode, other set = columns) or the columns (COL mode, other set = rows).  Per 32-wide
// tile of the other set T the S^T tile is recomputed exactly as in the forward kernel
// (D[i = tile index][j = owner]); p = exp(s - lse_row) uses the row LSE saved by the
// forward pass.  The second product  out[owner][feature] += dS[i][owner] * T[i][feature]
// reuses the accumulator registers of the first as its B operand: MFMA step t needs
// k = i_h(t) = (t&3) + 8(t>>2) + 4h, which is exactly register t of lane half h.
// Every output row is produced by one workgroup in a fixed order: deterministic, no atomics.
// ---------------------------------------------------------------------------------------
template <int ND, bool COL>
__global__ __launch_bounds__(256) void ntxent_bwd_kernel(
        const float* __restrict__ org_l, const float* __restrict__ rep_l,
        const float* __restrict__ org_all, const float* __restrict__ rep_all,
        const float* __restrict__ row_lse, int n_local, int n_global, int rank_offset,
        float tau, float scale,                 // scale = 1 / (tau * n_global)
        float* __restrict__ part,               // [4 gridDim.y (split, wave)][n_owner][ND] partial sums over this wave's tiles
        int n_owner_pad) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, jl = lane & 31;
    const int n_rows = 2 * n_local, n_cols = 2 * n_global;
    const int n_owner = COL ? n_cols : n_rows, n_other = COL ? n_rows : n_cols;
    const int o = blockIdx.x * 32 + jl;                     // my owner index
    const bool ovalid = o < n_owner;
    // owner vector fragment (HK values: features HK*h .. HK*h+HK-1), like `rf` in the forward kernel
    constexpr int HK = ND / 2, NV = ND / 8, NB = ND / 32;
    auto row_ptr = [&](int r) { return r >= n_local ? rep_l + (int64_t)(r - n_local) * ND : org_l + (int64_t)r * ND; };
    auto col_ptr = [&](int c) { return c >= n_global ? rep_all + (int64_t)(c - n_global) * ND : org_all + (int64_t)c * ND; };
    float of[HK];
    {
        const float* src = ovalid ? (COL ? col_ptr(o) : row_ptr(o)) + HK * h : nullptr;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 t = ovalid ? *(const float4*)(src + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            of[4 * v] = t.x; of[4 * v + 1] = t.y; of[4 * v + 2] = t.z; of[4 * v + 3] = t.w;
        }
    }
    // row/column bookkeeping of the owner
    int o_self = -1, o_pos = -1;       // ROW: column indices self(r), pos(r);  COL: unused
    float o_lse = 0.f;
    if (!COL && ovalid) {
        const bool is_b = o >= n_local; const int gi = rank_offset + (is_b ? o - n_local : o);
        o_self = is_b ? n_global + gi : gi; o_pos = is_b ? gi : n_global + gi;
        o_lse = row_lse[o];
    }
    f32x16 out[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) out[nb][q] = 0.f;

    const int n_tiles = (n_other + 31) / 32;
    for (int t = blockIdx.y * 4 + wave; t < n_tiles; t += 4 * gridDim.y) {      // blockIdx.y: split over the other set's tiles
        const int ti = t * 32 + jl;                          // tile member this lane feeds (A operand)
        const bool tvalid = ti < n_other;
        const float* tsrc = tvalid ? (COL ? row_ptr(ti) : col_ptr(ti)) + HK * h : nullptr;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 cv = tvalid ? *(const float4*)(tsrc + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv.x, of[4 * v], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv.y, of[4 * v + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv.z, of[4 * v + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv.w, of[4 * v + 3], acc, 0, 0, 0);
        }
        // acc[q] = S^T[i][owner] * tau with i = t*32 + (q&3) + 8(q>>2) + 4h  -> dS in place
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = t * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            const float sv = acc[q] / tau;
            float d = 0.f;
            if (ovalid && i < n_other) {
                int r, c; float lse;
                if (COL) {
                    r = i; c = o;
                    const bool is_b = r >= n_local; const int gi = rank_offset + (is_b ? r - n_local : r);
                    const int self_c = is_b ? n_global + gi : gi, pos_c = is_b ? gi : n_global + gi;
                    lse = row_lse[r];
                    if (c != self_c) d = (__expf(sv - lse) - (c == pos_c ? 1.f : 0.f)) * scale;
                } else {
                    c = i;
                    if (c != o_self) d = (__expf(sv - o_lse) - (c == o_pos ? 1.f : 0.f)) * scale;
                }
            }
            acc[q] = d;
        }
        // out[owner][feature] += sum_i dS[i][owner] * T[i][feature]
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = t * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            const bool iv = i < n_other;
            const float* trow = iv ? (COL ? row_ptr(i) : col_ptr(i)) : nullptr;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float tv = iv ? trow[nb * 32 + jl] : 0.f;
                out[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(tv, acc[q], out[nb], 0, 0, 0);
            }
        }
    }
    // D2[i' = feature (regs)][j' = owner (lane)]: feature = nb*32 + (q&3) + 8(q>>2) + 4h
    if (ovalid) {
        float* dst = part + ((int64_t)(blockIdx.y * 4 + wave) * n_owner_pad + o) * ND;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(dst + nb * 32 + 8 * g + 4 * h) =
                    make_float4(out[nb][4 * g], out[nb][4 * g + 1], out[nb][4 * g + 2], out[nb][4 * g + 3]);
    }
}

// d_all[c] = sum_w partC[w][c] (+ sum_w partR[w][local row of c]); fixed order.
__global__ void ntxent_bwd_combine_kernel(const float* __restrict__ partC, const float* __restrict__ partR,
                                          int n_local, int n_global, int rank_offset, int padC, int padR, int ND,
                                          int n_partC, int n_partR, float* __restrict__ d_org_all, float* __restrict__ d_rep_all) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;     // float4 index over (2*n_global, ND/4)
    const int64_t total = (int64_t)2 * n_global * (ND / 4);
    if (idx >= total) return;
    const int c = (int)(idx / (ND / 4)), f4 = (int)(idx % (ND / 4));
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < n_partC; ++w) {
        const float4 t = *(const float4*)(partC + ((int64_t)w * padC + c) * ND + 4 * f4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const bool is_b = c >= n_global; const int gi = is_b ? c - n_global : c;
    const int li = gi - rank_offset;
    if (li >= 0 && li < n_local) {
        const int r = is_b ? n_local + li : li;
        for (int w = 0; w < n_partR; ++w) {
            const float4 t = *(const float4*)(partR + ((int64_t)w * padR + r) * ND + 4 * f4);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
    }
    float* dst = (is_b ? d_rep_all : d_org_all) + (int64_t)gi * ND + 4 * f4;
    *(float4*)dst = s;
}

// =======================================================================================
// [r5] Second generation of the three kernels (ND = 64 / 128; ND = 256 keeps the kernels above).
//
// What the kernels above cost (5120 x 5120, loss + both gradients: 0.80 ms = 42 TFLOP/s of executed MFMA work): every wave
// walked its OWN tiles of the other set straight from global memory -- 16 dependent float4 loads in front of the first
// product, 64 single-float loads inside the second, nothing in flight while the matrix pipe worked, each tile fetched once
// per wave -- and the single-device backward ran the whole similarity product twice (once per gradient).
//
//   * A workgroup owns 128 members of the owner set (32 per wave, their fragments in registers) and ALL FOUR WAVES work on
//     the same tile of the other set, staged once through LDS (double-buffered, rows padded by one 16-byte chunk: the first
//     product reads row-per-lane b128, the second chunk-per-lane b128 of one row, both conflict-free); the loads of tile
//     t + 1 are in flight while tile t is multiplied.  One barrier per tile.
//   * The second product takes its A operand as NB consecutive features per lane (one ds_read_b128 feeds the 4 feature
//     blocks at ND = 128), i.e. output feature = NB * m + nb instead of nb * 32 + m: only the store pattern changes.
//   * Single device (rows == columns: the BSZ-5120 and BSZ-1280 steps): S is symmetric, so
//         dZ = dS Z + dS^T Z = (dS + dS^T) Z,  (dS + dS^T)[r][c] = (e^{s - lse_r} + e^{s - lse_c} - 2 [c == pos(r)]) / n  (c != r)
//     -- ONE pass with two exponentials per logit instead of two passes with a full similarity product each (MODE SYM).
// The accumulators of the first product are the B operand of the second as before (k = tile row i(q, h) = register q of lane
// half h).  Sums are taken in a fixed order (splits -> combine kernel in split order): deterministic, no atomics.
// =======================================================================================
namespace v2 {
constexpr int OWN = 128;
enum { FWD = 0, BWD_ROW = 1, BWD_COL = 2, BWD_SYM = 3 };
struct Params {
    const float* org_l; const float* rep_l; const float* org_all; const float* rep_all;
    const float* row_lse;
    int n_local, n_global, rank_offset;
    float tau, inv_hi, inv_lo;          // 1 / tau as a float pair (s - lse = acc * (hi + lo) - lse in one rounding: the gradient is a sum with
                                        // heavy cancellation, a relative 1e-6 on every probability of a row shows)
    float scale, l2scale;               // scale = 1 / (tau * n_global) and its log2 (folded into the exponent of the backward modes)
    float* fpart; float* sim_mtx;
    float* part; int n_owner_pad, tiles_per_split;          // FWD / ROW / SYM launch (or the ROW half of the fused ROW + COL launch)
    float* partC; int n_owner_padC, tiles_per_splitC;       // the COL half of the fused launch
    int gxR, nR, gxC;                                        // fused launch: owner blocks of the ROW part, its workgroups, owner blocks of the COL part
};
constexpr float LOG2E = 1.44269504088896341f;

template <int ND, int MODE>
__device__ __forceinline__ void ntxent2_body(const Params& p, const int bx, const int by, float* __restrict__ part,
                                             const int n_owner_pad, const int tiles_per_split) {
    constexpr bool OWNER_COL = MODE == BWD_COL;
    constexpr bool NEED_LSE = MODE == BWD_COL || MODE == BWD_SYM;      // the LSE of the OTHER set's rows
    constexpr int HK = ND / 2, NV = ND / 8, NB = ND / 32, CPR = ND / 4, NCH = 32 * CPR / 256;
    static_assert(ND == 64 || ND == 128, "embedding widths of the second-generation kernels");
    // LDS: [2][32 rows of RS = ND + 4 floats] tile ring, then [2][32] row LSEs.  Rows are PADDED by one 16-byte chunk (not XOR-swizzled):
    // both read patterns -- row-per-lane b128 in the first product, chunk-per-lane b128 of one row in the second -- are conflict-free
    // AND every read address is one per-lane base plus an immediate (32 swizzled offsets kept in registers made the SYM kernel spill)
    constexpr int RS = ND + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sLse = smem + 2 * 32 * RS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, jl = lane & 31;
    const int n_local = p.n_local, n_global = p.n_global;
    const int n_rows = 2 * n_local, n_cols = 2 * n_global;
    const int n_owner = OWNER_COL ? n_cols : n_rows, n_other = OWNER_COL ? n_rows : n_cols;
    const float* const org_l = p.org_l; const float* const rep_l = p.rep_l;
    const float* const org_all = p.org_all; const float* const rep_all = p.rep_all;
    const float* const row_lse = p.row_lse;
    auto row_ptr = [=](int r) { return r >= n_local ? rep_l + (int64_t)(r - n_local) * ND : org_l + (int64_t)r * ND; };
    auto col_ptr = [=](int c) { return c >= n_global ? rep_all + (int64_t)(c - n_global) * ND : org_all + (int64_t)c * ND; };
    const int o = bx * OWN + wave * 32 + jl;                  // my owner
    const bool ovalid = o < n_owner;
    float of[HK];
    {
        const float* src = (OWNER_COL ? col_ptr(ovalid ? o : 0) : row_ptr(ovalid ? o : 0)) + HK * h;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 t = ovalid ? *(const float4*)(src + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            of[4 * v] = t.x; of[4 * v + 1] = t.y; of[4 * v + 2] = t.z; of[4 * v + 3] = t.w;
        }
    }
    // owner bookkeeping: the member of the OTHER set that is excluded for this owner (its own embedding) and its positive
    int o_self = -1, o_pos = -1, gi_o = 0;
    bool o_is_b = false;
    float o_lse = 0.f;                                         // lse of the owner (row owners of the backward modes)
    if (ovalid) {
        if (!OWNER_COL) {                                      // owner = local row o; other = columns
            o_is_b = o >= n_local; gi_o = p.rank_offset + (o_is_b ? o - n_local : o);
            o_self = o_is_b ? n_global + gi_o : gi_o; o_pos = o_is_b ? gi_o : n_global + gi_o;
            if (MODE != FWD) o_lse = row_lse[o];
        } else {                                               // owner = column o; other = local rows: the rows that exclude / reward it
            const bool is_b = o >= n_global; const int li = (is_b ? o - n_global : o) - p.rank_offset;
            if (li >= 0 && li < n_local) { o_self = is_b ? n_local + li : li; o_pos = is_b ? li : n_local + li; }
        }
    }
    const int t_self = o_self >> 5, t_pos = o_pos >> 5;       // the tiles that hold them (-1: none)
    f32x16 out[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) out[nb][q] = 0.f;
    float run_m = -INFINITY, run_s = 0.f, pos_v = 0.f;        // FWD

    const int n_tiles = (n_other + 31) / 32;
    const int t0 = by * tiles_per_split, t1 = min(n_tiles, t0 + tiles_per_split);
    const int t_ragged = (n_other & 31) ? n_tiles - 1 : -1;
    // (scalars, not an array: hipcc left a float4 array that is written under a condition in scratch)
    float4 pf0 = make_float4(0.f, 0.f, 0.f, 0.f), pf1 = pf0, pf2 = pf0, pf3 = pf0; float pf_lse = 0.f;
    static_assert(NCH == 2 || NCH == 4, "chunks per thread and tile");
#define NAFP_NT2_SRC(t_, e_)                                                                               \
    ((OWNER_COL ? row_ptr((t_) * 32 + (tid + 256 * (e_)) / CPR < n_other ? (t_) * 32 + (tid + 256 * (e_)) / CPR : 0)      \
                : col_ptr((t_) * 32 + (tid + 256 * (e_)) / CPR < n_other ? (t_) * 32 + (tid + 256 * (e_)) / CPR : 0)) +   \
     4 * ((tid + 256 * (e_)) % CPR))
    // rows beyond the set re-read row 0: finite values, their logits are masked below
#define NAFP_NT2_PREFETCH(t_)                                                                              \
    {                                                                                                      \
        pf0 = *(const float4*)NAFP_NT2_SRC(t_, 0);                                                         \
        pf1 = *(const float4*)NAFP_NT2_SRC(t_, 1);                                                         \
        if (NCH == 4) { pf2 = *(const float4*)NAFP_NT2_SRC(t_, 2); pf3 = *(const float4*)NAFP_NT2_SRC(t_, 3); } \
        if (NEED_LSE && tid < 32) pf_lse = row_lse[min((t_) * 32 + tid, n_rows - 1)];                      \
    }
#define NAFP_NT2_DST(T_, e_) ((T_) + ((tid + 256 * (e_)) / CPR) * RS + 4 * ((tid + 256 * (e_)) % CPR))
    if (t0 < t1) NAFP_NT2_PREFETCH(t0)
    const int a1off = jl * RS + HK * h;                        // first product: my row, my half of the features
    const int a2off = 4 * h * RS + NB * jl;                    // second product: row 4 h (+ the step's constant), my NB features
    for (int t = t0; t < t1; ++t) {
        float* T = smem + ((t - t0) & 1) * 32 * RS;
        *(float4*)NAFP_NT2_DST(T, 0) = pf0;
        *(float4*)NAFP_NT2_DST(T, 1) = pf1;
        if (NCH == 4) { *(float4*)NAFP_NT2_DST(T, 2) = pf2; *(float4*)NAFP_NT2_DST(T, 3) = pf3; }
        if (NEED_LSE && tid < 32) sLse[((t - t0) & 1) * 32 + tid] = pf_lse;
        __syncthreads();            // tile t is complete; the buffer written next (tile t + 1's) was last read before this barrier
        if (t + 1 < t1) NAFP_NT2_PREFETCH(t + 1)
        // ---- first product: acc[q] = tau * S^T[i(q, h)][owner], i(q, h) = (q & 3) + 8 (q >> 2) + 4 h ----
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 a = *(const float4*)(T + a1off + 4 * v);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, of[4 * v], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, of[4 * v + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, of[4 * v + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, of[4 * v + 3], acc, 0, 0, 0);
        }
        // Only the tiles that hold an owner's own embedding or its positive, and a ragged last tile, need per-logit tests
        // (wave-uniform decision; 2 of the tiles a wave walks, typically): every vector instruction here is paid in MFMA issue time.
        const bool special = __ballot(t_self == t || t_pos == t) != 0ull || t == t_ragged;
        if (MODE == FWD) {
            // online log-sum-exp over the columns of this tile (lane-local: a lane owns one row), as ntxent_fwd_kernel.
            // s = acc / tau (NTxent_loss_single_gpu.py:72-77) with 1 / tau as a float pair: within half an ulp of the quotient
            // (a division is ~10 instructions per logit)
            float tmax = -INFINITY;
            float sv[16];
            if (special || p.sim_mtx) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int cc = t * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    const float s = fmaf(acc[q], p.inv_hi, acc[q] * p.inv_lo);
                    const bool use = cc < n_cols && cc != o_self;
                    if (cc == o_pos) pos_v = s;
                    sv[q] = use ? s : -INFINITY;
                    tmax = fmaxf(tmax, sv[q]);
                    if (p.sim_mtx && ovalid && !o_is_b && use) {
                        // [ab | aa without diagonal] (NTxent_loss_single_gpu.py:78-82)
                        const int j = cc >= n_global ? cc - n_global : n_global + (cc < gi_o ? cc : cc - 1);
                        p.sim_mtx[(int64_t)o * (n_cols - 1) + j] = s;
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) { sv[q] = fmaf(acc[q], p.inv_hi, acc[q] * p.inv_lo); tmax = fmaxf(tmax, sv[q]); }
            }
            if (tmax > -INFINITY) {
                const float nm = fmaxf(run_m, tmax);
                float add = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) add += __expf(sv[q] - nm);     // exp(-inf) = 0
                run_s = run_s * __expf(run_m - nm) + add;
                run_m = nm;
            }
            continue;
        }
        // ---- dS (or dS + dS^T) in place: d = scale * (softmax - indicator), the scale folded into the exponent ----
        const float* lse_t = sLse + ((t - t0) & 1) * 32;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int il = (q & 3) + 8 * (q >> 2) + 4 * h;      // tile-local member of the other set
            const float s_lo = acc[q] * p.inv_lo;
            float e = __builtin_amdgcn_exp2f(fmaf(LOG2E, fmaf(acc[q], p.inv_hi, s_lo - (OWNER_COL ? lse_t[il] : o_lse)), p.l2scale));   // softmax of the ROW involved
            if (MODE == BWD_SYM) e += __builtin_amdgcn_exp2f(fmaf(LOG2E, fmaf(acc[q], p.inv_hi, s_lo - lse_t[il]), p.l2scale));          // + the transposed entry
            acc[q] = e;
        }
        if (special) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = t * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                const float d = acc[q] - (i == o_pos ? (MODE == BWD_SYM ? 2.f : 1.f) * p.scale : 0.f);
                acc[q] = (i == o_self || i >= n_other) ? 0.f : d;
            }
        }
        // ---- second product: out[owner][feature NB m + nb] += sum_i dS[i][owner] T[i][feature] ----
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rq = ((q & 3) + 8 * (q >> 2)) * RS;       // tile row i(q, h) = (q & 3) + 8 (q >> 2) + 4 h
            if (NB == 4) {
                const float4 a = *(const float4*)(T + a2off + rq);
                out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc[q], out[0], 0, 0, 0);
                out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc[q], out[1], 0, 0, 0);
                out[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, acc[q], out[2], 0, 0, 0);
                out[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, acc[q], out[3], 0, 0, 0);
            } else {
                const float2 a = *(const float2*)(T + a2off + rq);
                out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc[q], out[0], 0, 0, 0);
                out[NB - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc[q], out[NB - 1], 0, 0, 0);
            }
        }
    }
#undef NAFP_NT2_PREFETCH
#undef NAFP_NT2_SRC
#undef NAFP_NT2_DST
    if (MODE == FWD) {
        // the two lane halves hold different columns of the same row
        const float om = __shfl_xor(run_m, 32, 64), os = __shfl_xor(run_s, 32, 64);
        const float op = __shfl_xor(pos_v, 32, 64);
        const float nm = fmaxf(run_m, om);
        const float a = run_m > -INFINITY ? run_s * __expf(run_m - nm) : 0.f;
        const float b = om > -INFINITY ? os * __expf(om - nm) : 0.f;
        if (ovalid && h == 0) {                                // this split's (max, sum exp, positive logit): merged by ntxent_merge_rows_kernel
            float* dst = p.fpart + ((int64_t)by * n_rows + o) * 3;
            dst[0] = nm; dst[1] = a + b; dst[2] = pos_v + op;  // (pos_v is non-zero in exactly one place)
        }
        return;
    }
    // D2[m = (r & 3) + 8 (r >> 2) + 4 h][owner]: features NB m .. NB m + NB - 1 are registers r of out[0 .. NB - 1]
    if (ovalid) {
        float* dst = part + ((int64_t)by * n_owner_pad + o) * ND;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (NB == 4) *(float4*)(dst + 4 * m) = make_float4(out[0][r], out[1][r], out[2][r], out[3][r]);
            else *(float2*)(dst + 2 * m) = make_float2(out[0][r], out[NB - 1][r]);
        }
    }
}

template <int ND, int MODE>
__global__ __launch_bounds__(256, 2) void ntxent2_kernel(const Params p) {
    ntxent2_body<ND, MODE>(p, (int)blockIdx.x, (int)blockIdx.y, p.part, p.n_owner_pad, p.tiles_per_split);
}
// The sharded backward: the gradient w.r.t. the local rows (ROW) and w.r.t. all gathered columns (COL) are independent of each
// other -- ONE launch holds both (the first nR workgroups are the ROW part), so neither runs as a quarter-filled grid of its own.
template <int ND>
__global__ __launch_bounds__(256, 2) void ntxent2_rowcol_kernel(const Params p) {
    const int b = (int)blockIdx.x;
    if (b < p.nR) ntxent2_body<ND, BWD_ROW>(p, b % p.gxR, b / p.gxR, p.part, p.n_owner_pad, p.tiles_per_split);
    else ntxent2_body<ND, BWD_COL>(p, (b - p.nR) % p.gxC, (b - p.nR) / p.gxC, p.partC, p.n_owner_padC, p.tiles_per_splitC);
}

static int splits_for(int64_t owners, int64_t others, int* tiles_per_split) {
    const int64_t blocks = (owners + OWN - 1) / OWN, tiles = (others + 31) / 32;
    static const int64_t wg_target = []() { const char* e = getenv("NAFP_NTXENT2_WGS"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)512; }();
    static const int64_t max_split = []() { const char* e = getenv("NAFP_NTXENT2_MAXSPLIT"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)32; }();
    int64_t s = std::max<int64_t>(1, std::min<int64_t>(max_split, (wg_target + blocks - 1) / blocks));
    s = std::min<int64_t>(s, tiles);
    const int64_t tps = (tiles + s - 1) / s;
    *tiles_per_split = (int)tps;
    return (int)((tiles + tps - 1) / tps);
}
}  // namespace v2

}  // namespace nafp

using namespace nafp;

// Splits of the "other" set per block of 32 owners: enough workgroups for the chip, at most 16.  (NAFP_NTXENT_WGS sweeps the
// target: 320 / 640 / 1280 / 2560 workgroups -> 0.883 / 0.843 / 0.800 / 0.893 ms for loss + both gradients at 5120 x 5120,
// 0.145 / 0.129 / 0.129 / 0.129 at 1280 x 1280; beyond 8 splits the partial sums of the gradients cost more than the occupancy gains.)
static int ntxent_splits(int64_t owners, int64_t others) {
    const int64_t blocks = (owners + 31) / 32, tiles = (others + 31) / 32;
    static const int64_t wg_target = []() { const char* e = getenv("NAFP_NTXENT_WGS"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)1280; }();
    int64_t s = std::max<int64_t>(1, std::min<int64_t>(16, wg_target / std::max<int64_t>(blocks, 1)));
    s = std::min<int64_t>(s, std::max<int64_t>(1, tiles / 4));            // every wave of every split gets a tile
    return (int)s;
}

extern "C" int64_t nafp_ntxent_workspace_bytes(int64_t n_local, int64_t n_global) {
    if (n_local < 0 || n_global < n_local) return -1;
    // forward partials (splits x rows x 3) + row_lse, then the backward partials:
    // [4 splitsR][2*n_local padded][d] + [4 splitsC][2*n_global padded][d], d <= 256
    const int64_t padR = (2 * n_local + 31) / 32 * 32, padC = (2 * n_global + 31) / 32 * 32;
    const int sR = ntxent_splits(2 * n_local, 2 * n_global), sC = ntxent_splits(2 * n_global, 2 * n_local);
    const int64_t v1 = (int64_t)sizeof(float) * ((3 * sR + 1) * (2 * n_local) + 64 + 4 * (sR * padR + sC * padC) * 256) + 256;
    // second-generation kernels (d <= 128): forward partials of their own split count, one partial slab per split
    int tps;
    const int64_t pR = (2 * n_local + v2::OWN - 1) / v2::OWN * v2::OWN, pC = (2 * n_global + v2::OWN - 1) / v2::OWN * v2::OWN;
    const int64_t fR = v2::splits_for(2 * n_local, 2 * n_global, &tps), fC = v2::splits_for(2 * n_global, 2 * n_local, &tps);
    const int64_t v2b = (int64_t)sizeof(float) * ((3 * fR + 1) * (2 * n_local) + 64 + (fR * pR + fC * pC) * 128) + 8 * ((2 * n_local + 7) / 8) + 320;
    return std::max(v1, v2b);
}

namespace {
template <int ND>
int ntxent_launch(const float* emb_org_local, const float* emb_rep_local, const float* emb_org_all,
                  const float* emb_rep_all, int64_t n_local, int64_t n_global, int64_t rank_offset, float tau,
                  float* loss_sum, float* sim_mtx, float* d_org_all, float* d_rep_all, void* workspace,
                  hipStream_t st) {
    const int n_rows = (int)(2 * n_local);
    static const bool use_v2 = []() { const char* e = getenv("NAFP_NTXENT_V2"); return !e || e[0] != '0'; }();
    if constexpr (ND <= 128) {
        if (use_v2) {
            using namespace v2;
            int tpsR, tpsC;
            const int fR = splits_for(2 * n_local, 2 * n_global, &tpsR), fC = splits_for(2 * n_global, 2 * n_local, &tpsC);
            const int padR = (int)((2 * n_local + OWN - 1) / OWN * OWN), padC = (int)((2 * n_global + OWN - 1) / OWN * OWN);
            Params q{};
            q.org_l = emb_org_local; q.rep_l = emb_rep_local; q.org_all = emb_org_all; q.rep_all = emb_rep_all;
            q.n_local = (int)n_local; q.n_global = (int)n_global; q.rank_offset = (int)rank_offset;
            q.tau = tau; q.scale = 1.0f / (tau * (float)n_global); q.l2scale = (float)std::log2(1.0 / ((double)tau * (double)n_global));
            q.inv_hi = (float)(1.0 / (double)tau); q.inv_lo = (float)(1.0 / (double)tau - (double)q.inv_hi);
            float* fpart = (float*)workspace;                   // [fR][n_rows][3]
            float* row_lse = fpart + (int64_t)3 * fR * n_rows;
            const int n_mwg = (n_rows + 7) / 8;
            double* wg_part = (double*)(((uintptr_t)(row_lse + n_rows) + 15) & ~(uintptr_t)15);          // [n_mwg]
            float* partR = (float*)(((uintptr_t)(wg_part + n_mwg) + 15) & ~(uintptr_t)15);
            float* partC = partR + (int64_t)fR * padR * ND;
            q.row_lse = row_lse; q.fpart = fpart; q.sim_mtx = sim_mtx;
            const int lds = (2 * 32 * (ND + 4) + 64) * (int)sizeof(float);
            q.tiles_per_split = tpsR;
            ntxent2_kernel<ND, FWD><<<dim3(padR / OWN, fR), 256, lds, st>>>(q);
            NAFP_LAUNCH_CHECK();
            ntxent_merge_rows_kernel<<<n_mwg, 256, 0, st>>>(fpart, n_rows, fR, row_lse, wg_part);
            NAFP_LAUNCH_CHECK();
            ntxent_loss_sum_kernel<<<1, 64, 0, st>>>(wg_part, n_mwg, loss_sum);
            NAFP_LAUNCH_CHECK();
            if (!d_org_all) return NAFP_OK;
            const int64_t total = (int64_t)2 * n_global * (ND / 4);
            // rows == columns (single device): one pass over the symmetric logits gives the whole gradient
            static const bool sym_on = []() { const char* e = getenv("NAFP_NTXENT_SYM"); return !e || e[0] != '0'; }();
            if (sym_on && n_local == n_global && rank_offset == 0 && emb_org_local == emb_org_all && emb_rep_local == emb_rep_all) {
                q.part = partR; q.n_owner_pad = padR; q.tiles_per_split = tpsR;
                ntxent2_kernel<ND, BWD_SYM><<<dim3(padR / OWN, fR), 256, lds, st>>>(q);
                NAFP_LAUNCH_CHECK();
                ntxent_bwd_combine_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
                    partR, partR, (int)n_local, (int)n_global, (int)rank_offset, padR, padR, ND, fR, 0, d_org_all, d_rep_all);
                NAFP_LAUNCH_CHECK();
                return NAFP_OK;
            }
            q.part = partR; q.n_owner_pad = padR; q.tiles_per_split = tpsR;
            q.partC = partC; q.n_owner_padC = padC; q.tiles_per_splitC = tpsC;
            q.gxR = padR / OWN; q.nR = q.gxR * fR; q.gxC = padC / OWN;
            ntxent2_rowcol_kernel<ND><<<dim3((unsigned)(q.nR + q.gxC * fC)), 256, lds, st>>>(q);
            NAFP_LAUNCH_CHECK();
            ntxent_bwd_combine_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
                partC, partR, (int)n_local, (int)n_global, (int)rank_offset, padC, padR, ND, fC, fR, d_org_all, d_rep_all);
            NAFP_LAUNCH_CHECK();
            return NAFP_OK;
        }
    }
    const int sR = ntxent_splits(2 * n_local, 2 * n_global), sC = ntxent_splits(2 * n_global, 2 * n_local);
    float* fpart = (float*)workspace;                       // [sR][n_rows][3]
    float* row_lse = fpart + (int64_t)3 * sR * n_rows;
    ntxent_fwd_kernel<ND><<<dim3((n_rows + 31) / 32, sR), 256, 0, st>>>(
        emb_org_local, emb_rep_local, emb_org_all, emb_rep_all, (int)n_local, (int)n_global,
        (int)rank_offset, tau, fpart, sim_mtx);
    NAFP_LAUNCH_CHECK();
    ntxent_merge_kernel<<<1, 256, 0, st>>>(fpart, n_rows, sR, row_lse, loss_sum);
    NAFP_LAUNCH_CHECK();
    if (d_org_all) {
        const int padR = (n_rows + 31) / 32 * 32, padC = (int)((2 * n_global + 31) / 32 * 32);
        float* partR = row_lse + n_rows + 64;
        partR = (float*)(((uintptr_t)partR + 15) & ~(uintptr_t)15);
        float* partC = partR + (int64_t)4 * sR * padR * ND;
        const float scale = 1.0f / (tau * (float)n_global);
        ntxent_bwd_kernel<ND, false><<<dim3(padR / 32, sR), 256, 0, st>>>(emb_org_local, emb_rep_local, emb_org_all, emb_rep_all,
                                                                         row_lse, (int)n_local, (int)n_global, (int)rank_offset,
                                                                         tau, scale, partR, padR);
        NAFP_LAUNCH_CHECK();
        ntxent_bwd_kernel<ND, true><<<dim3(padC / 32, sC), 256, 0, st>>>(emb_org_local, emb_rep_local, emb_org_all, emb_rep_all,
                                                                        row_lse, (int)n_local, (int)n_global, (int)rank_offset,
                                                                        tau, scale, partC, padC);
        NAFP_LAUNCH_CHECK();
        const int64_t total = (int64_t)2 * n_global * (ND / 4);
        ntxent_bwd_combine_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
            partC, partR, (int)n_local, (int)n_global, (int)rank_offset, padC, padR, ND, 4 * sC, 4 * sR, d_org_all, d_rep_all);
        NAFP_LAUNCH_CHECK();
    }
    return NAFP_OK;
}
}  // namespace

extern "C" int nafp_ntxent_forward(const float* emb_org_local, const float* emb_rep_local,
                                   const float* emb_org_all, const float* emb_rep_all,
                                   int64_t n_local, int64_t n_global, int64_t rank_offset, int d,
                                   float tau, float* loss_sum, float* sim_mtx,
                                   float* d_org_all, float* d_rep_all,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    if (!emb_org_local || !emb_rep_local || !emb_org_all || !emb_rep_all || !loss_sum || !workspace)
        return NAFP_ERR_INVALID_ARG;
    if (n_local <= 0 || n_global < n_local || rank_offset < 0 || rank_offset + n_local > n_global ||
        !(tau > 0.f) || n_global > (1 << 29))
        return NAFP_ERR_INVALID_ARG;
    if (d != 64 && d != 128 && d != 256) return NAFP_ERR_UNSUPPORTED;
    if ((d_org_all == nullptr) != (d_rep_all == nullptr)) return NAFP_ERR_INVALID_ARG;
    if (workspace_bytes < nafp_ntxent_workspace_bytes(n_local, n_global)) return NAFP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
#define NAFP_NTXENT_ARGS emb_org_local, emb_rep_local, emb_org_all, emb_rep_all, n_local, n_global, rank_offset, tau, \
                         loss_sum, sim_mtx, d_org_all, d_rep_all, workspace, st
    if (d == 64) return ntxent_launch<64>(NAFP_NTXENT_ARGS);
    if (d == 256) return ntxent_launch<256>(NAFP_NTXENT_ARGS);
    return ntxent_launch<128>(NAFP_NTXENT_ARGS);
#undef NAFP_NTXENT_ARGS
}
