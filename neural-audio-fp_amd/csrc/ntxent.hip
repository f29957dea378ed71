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
    return (int64_t)sizeof(float) * ((3 * sR + 1) * (2 * n_local) + 64 + 4 * (sR * padR + sC * padC) * 256) + 256;
}

namespace {
template <int ND>
int ntxent_launch(const float* emb_org_local, const float* emb_rep_local, const float* emb_org_all,
                  const float* emb_rep_all, int64_t n_local, int64_t n_global, int64_t rank_offset, float tau,
                  float* loss_sum, float* sim_mtx, float* d_org_all, float* d_rep_all, void* workspace,
                  hipStream_t st) {
    const int n_rows = (int)(2 * n_local);
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
