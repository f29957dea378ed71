// Exact nearest-neighbour search over resident fingerprints + sequence scoring, gfx950.
//
// The consumer side of the hot path's output: eval/eval_faiss.py:115-289 with the exact index
// (faiss.IndexFlatL2, eval/utils/get_index_faiss.py:57-62).  The whole index ([dummy_db ; db], N x d
// float32: 51 GB for 100 M fingerprints) stays resident in HBM; there is no training, no
// quantisation and no host round trip per query.
//
//   search_half_norms_kernel   h[i] = |x_i|^2 / 2   (+inf padding after N)
//   search_topk_kernel<D,K>    per query the K index rows with the smallest |q - x|^2, i.e. the
//                              largest key = q.x - |x|^2/2.  fp32 MFMA (v_mfma_f32_32x32x2_f32):
//                              A = a 64-row index tile staged by LDS-DMA (XOR-swizzled chunks,
//                              2-stage ring), B = the workgroup's 128 queries held in REGISTERS for the
//                              whole scan (lane <-> one query column), so the index streams through
//                              the chip once per 128 queries.  In the 32x32 C/D layout a lane owns ONE
//                              query and 16 index rows per block: it keeps that query's running top-K
//                              as a sorted register list (threshold test per score, unrolled insertion
//                              only when a score beats the K-th best).  blockIdx.y splits the index.
//   search_merge_kernel<K>     merges the 2 x splits partial lists of a query (key desc, id asc)
//   search_seq_score_kernel<D> mean_i q[t+i] . x[c+i] for candidate sequence starts c (eval_faiss.py:221-230)
#include "nafp_common.h"

#include <algorithm>

namespace nafp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4s make_rsrc_s(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4s r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void lds_dma16_s(unsigned lds_addr, unsigned voff, u32x4s rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}

__global__ __launch_bounds__(256) void search_half_norms_kernel(const float* __restrict__ x, float* __restrict__ hn,
                                                                int64_t N, int64_t n_pad, int D) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n_pad) return;
    if (i >= N) { hn[i] = INFINITY; return; }
    const float4* r = (const float4*)(x + i * D);
    float s = 0.f;
    for (int c = 0; c < D / 4; ++c) { const float4 v = r[c]; s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
    hn[i] = 0.5f * s;
}

constexpr int TILE_ROWS = 64;

template <int D, int K>
__device__ __forceinline__ void search_topk_body(
        const float* __restrict__ Q, const float* __restrict__ X, const float* __restrict__ hn,
        float* __restrict__ out_key, int* __restrict__ out_id, int nq, int64_t N, int tiles_per_split, int n_lists) {
    constexpr int CH = D / 4;                        // 16-B chunks per row
    constexpr int SWZ = (CH < 32 ? CH : 32) - 1;      // chunk swizzle mask: the row's low bits -- the same for row rl and row 32 + rl, which share `aoff` (d = 256 has 64 chunks)
    constexpr int RPI = 64 / CH;                     // rows per DMA wave-instruction
    constexpr int NI = 16 / RPI;                     // DMA instructions per wave per tile (16 rows per wave)
    constexpr int TILE = TILE_ROWS * D;              // floats
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [2][64][D]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & 31, hh = lane >> 5;
    const int qn = blockIdx.x * 128 + wave * 32 + rl;
    const int64_t n_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
    const int64_t t0 = (int64_t)blockIdx.y * tiles_per_split;
    const int64_t t1 = std::min<int64_t>(n_tiles, t0 + tiles_per_split);

    // the query column of this lane: q[8kk + 4hh + j], j = 0..3 (the k permutation A uses too)
    float4 qr[D / 8];
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk)
        qr[kk] = qn < nq ? *(const float4*)(Q + (int64_t)qn * D + 8 * kk + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);

    float sc[K]; int id[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { sc[j] = -INFINITY; id[j] = -1; }

    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    // DMA geometry: wave w stages rows [16w, 16w+16) of the tile; instruction i covers rows
    // 16w + i*RPI + lane/CH, physical chunk lane%CH holding logical chunk pc ^ swz(row)
    unsigned voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = wave * 16 + i * RPI + lane / CH;
        const int lc = (lane % CH) ^ (row & SWZ);
        voff[i] = (unsigned)(row * D + lc * 4) * 4u;
    }
#define NAFP_S_DMA(t_, slot_)                                                                       \
    {                                                                                               \
        const int64_t row0_l = (t_) * TILE_ROWS;                                                    \
        const int64_t left_l = (N - row0_l) * D * 4;                                                \
        const u32x4s rs_l = make_rsrc_s(X + row0_l * D, (unsigned)std::min<int64_t>(left_l, TILE * 4)); \
        _Pragma("unroll") for (int i = 0; i < NI; ++i)                                              \
            lds_dma16_s(lds0 + (unsigned)(((slot_) * TILE + (wave * 16 + i * RPI) * D) * 4), voff[i], rs_l); \
    }
    (void)OOB;
    if (t0 < t1) NAFP_S_DMA(t0, 0)
    // The f32 MFMAs share the SIMD's issue time with every vector instruction (conv.hip), so the scan loop is written for
    // instruction count: operand addresses are per-lane constants (the swizzled chunk of every kk, computed once) plus
    // immediates, the accumulators start at -|x|^2/2 (the key needs no subtraction), and a block of 16 scores is tested
    // against the K-th best with one max tree before any per-score work.
    unsigned aoff[D / 8];                          // byte offset of (row rl, logical chunk 2kk + hh) inside a tile
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk) aoff[kk] = (unsigned)((rl * D + (((2 * kk + hh) ^ (rl & SWZ)) * 4)) * 4);
    const char* sbase = (const char*)smem;
#define NAFP_S_TILE(SLOT_)                                                                          \
    {                                                                                               \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __builtin_amdgcn_s_barrier();                                                               \
        asm volatile("" ::: "memory");                                                              \
        if (t + 1 < t1) NAFP_S_DMA(t + 1, (SLOT_) ^ 1)                                              \
        const float* hp = hn + t * TILE_ROWS;                  /* wave-uniform -> scalar loads */   \
        f32x16 acc[2];                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                            \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                        \
                const int o = mi * 32 + 8 * (r >> 2) + (r & 3);                                     \
                acc[mi][r] = -(hh ? hp[o + 4] : hp[o]);                                             \
            }                                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < D / 8; ++kk) {                                      \
            float4 a[2];                                                                            \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                        \
                a[mi] = *(const float4*)(sbase + aoff[kk] + ((SLOT_) * TILE + mi * 32 * D) * 4);    \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                      \
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, qr[kk].x, acc[mi], 0, 0, 0); \
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, qr[kk].y, acc[mi], 0, 0, 0); \
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, qr[kk].z, acc[mi], 0, 0, 0); \
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, qr[kk].w, acc[mi], 0, 0, 0); \
            }                                                                                       \
        }                                                                                           \
        /* selection: D[i = index row][j = query]: this lane holds rows (r&3) + 8(r>>2) + 4hh of block mi */ \
        const int base_id = (int)(t * TILE_ROWS);                                                   \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                          \
            float m = fmaxf(acc[mi][0], acc[mi][1]);                                                \
            _Pragma("unroll") for (int r = 2; r < 16; r += 2) m = fmaxf(m, fmaxf(acc[mi][r], acc[mi][r + 1])); \
            if (m > sc[K - 1]) {                                                                    \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                    \
                    const float key = acc[mi][r];                                                   \
                    if (key > sc[K - 1]) {                                                          \
                        const int nid = base_id + mi * 32 + 8 * (r >> 2) + (r & 3) + 4 * hh;        \
                        _Pragma("unroll") for (int j = K - 1; j >= 1; --j) {                        \
                            const bool cj = key > sc[j], cp = key > sc[j - 1];                      \
                            id[j] = cj ? (cp ? id[j - 1] : nid) : id[j];                            \
                            sc[j] = cj ? (cp ? sc[j - 1] : key) : sc[j];                            \
                        }                                                                           \
                        if (key > sc[0]) { sc[0] = key; id[0] = nid; }                              \
                    }                                                                               \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
        if (++t >= t1) break;                                                                       \
    }
    if (t0 < t1)
        for (int64_t t = t0;;) {
            NAFP_S_TILE(0)
            NAFP_S_TILE(1)
        }
#undef NAFP_S_TILE
#undef NAFP_S_DMA
    if (qn < nq) {
        const int64_t o = ((int64_t)qn * n_lists + blockIdx.y * 2 + hh) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) { out_key[o + j] = sc[j]; out_id[o + j] = id[j]; }
    }
}

template <int D, int K>
__global__ __launch_bounds__(256, 2) void search_topk_kernel(
        const float* __restrict__ Q, const float* __restrict__ X, const float* __restrict__ hn,
        float* __restrict__ out_key, int* __restrict__ out_id, int nq, int64_t N, int tiles_per_split, int n_lists) {
    search_topk_body<D, K>(Q, X, hn, out_key, out_id, nq, N, tiles_per_split, n_lists);
}
// d = 256 (EMB_SZ 256: the encoder and NT-Xent support it, so the exact index does too): a lane keeps 128 floats of its query
// column, and the 2-tile ring is 128 KB of LDS -- one workgroup per CU either way, so the kernel may use the whole register file
template <int K>
__global__ __launch_bounds__(256, 1) void search_topk_kernel_d256(
        const float* __restrict__ Q, const float* __restrict__ X, const float* __restrict__ hn,
        float* __restrict__ out_key, int* __restrict__ out_id, int nq, int64_t N, int tiles_per_split, int n_lists) {
    search_topk_body<256, K>(Q, X, hn, out_key, out_id, nq, N, tiles_per_split, n_lists);
}

__device__ __forceinline__ unsigned long long pack_key(float key, int id) {
    unsigned u = __float_as_uint(key);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;                 // order-preserving map
    return ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - id);      // ties: smaller id is larger
}

// one wave per query: K rounds of wave-wide arg-max over the n_lists*K partial results
template <int K>
__global__ __launch_bounds__(64) void search_merge_kernel(const float* __restrict__ in_key, const int* __restrict__ in_id,
                                                          const float* __restrict__ Q, float* __restrict__ out_dist,
                                                          int* __restrict__ out_ids, int n_lists, int D, int k_out) {
    extern __shared__ unsigned long long cand[];
    const int q = blockIdx.x, lane = threadIdx.x;
    const int M = n_lists * K;
    for (int i = lane; i < M; i += 64) {
        const int idv = in_id[(int64_t)q * M + i];
        cand[i] = idv >= 0 ? pack_key(in_key[(int64_t)q * M + i], idv) : 0ull;
    }
    float qq = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = Q[(int64_t)q * D + c]; qq += v * v; }
    qq = wave_sum(qq);
    __syncthreads();
    for (int j = 0; j < k_out; ++j) {
        unsigned long long best = 0ull; int bi = -1;
        for (int i = lane; i < M; i += 64)
            if (cand[i] > best) { best = cand[i]; bi = i; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best) { best = ob; bi = oi; }
        }
        if (lane == 0) {
            if (best == 0ull) { out_ids[(int64_t)q * k_out + j] = -1; out_dist[(int64_t)q * k_out + j] = INFINITY; }
            else {
                out_ids[(int64_t)q * k_out + j] = in_id[(int64_t)q * M + bi];
                out_dist[(int64_t)q * k_out + j] = fmaxf(qq - 2.f * in_key[(int64_t)q * M + bi], 0.f);
                cand[bi] = 0ull;
            }
        }
        __syncthreads();
    }
}

// out[task, slot] = mean_{i < min(len, N - c)} Q[q0 + i] . X[c + i]  for c = cand[task, slot] >= 0, else -inf.
// One wave per (task, slot).
template <int D>
__global__ __launch_bounds__(256) void search_seq_score_kernel(
        const float* __restrict__ Q, const float* __restrict__ X, const int* __restrict__ task_q0,
        const int* __restrict__ task_len, const int* __restrict__ cand, float* __restrict__ out, int64_t N,
        int n_slots, int64_t total) {
    const int64_t w = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= total) return;
    const int task = (int)(w / n_slots);
    const int c = cand[w];
    if (c < 0 || c >= N) { if (lane == 0) out[w] = -INFINITY; return; }
    const int q0 = task_q0[task];
    const int len = (int)std::min<int64_t>(task_len[task], N - c);
    float s = 0.f;
    for (int i = 0; i < len; ++i) {
        const float* qp = Q + (int64_t)(q0 + i) * D; const float* xp = X + (int64_t)(c + i) * D;
        for (int e = lane; e < D; e += 64) s = fmaf(qp[e], xp[e], s);
    }
    s = wave_sum(s);
    if (lane == 0) out[w] = s / (float)len;
}

template <int D, int K>
static int launch_topk(const float* Q, int nq, const float* X, const float* hn, int64_t N, float* pk, int* pi, int splits,
                       int tps, hipStream_t st) {
    const int lds = 2 * TILE_ROWS * D * (int)sizeof(float);
    const dim3 grid((unsigned)((nq + 127) / 128), (unsigned)splits);
    if constexpr (D == 256) {
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)search_topk_kernel_d256<K>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        search_topk_kernel_d256<K><<<grid, 256, lds, st>>>(Q, X, hn, pk, pi, nq, N, tps, 2 * splits);
    } else {
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)search_topk_kernel<D, K>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        search_topk_kernel<D, K><<<grid, 256, lds, st>>>(Q, X, hn, pk, pi, nq, N, tps, 2 * splits);
    }
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

static void plan_splits(int nq, int64_t N, int K, int* splits, int* tps) {
    const int64_t n_tiles = (N + TILE_ROWS - 1) / TILE_ROWS;
    const int q_tiles = (nq + 127) / 128;
    int64_t s = std::max<int64_t>(1, 2048 / q_tiles);
    s = std::min<int64_t>(s, std::max<int64_t>(1, n_tiles / 8));     // at least 8 tiles per workgroup
    s = std::min<int64_t>(s, 4096 / (2 * K));                          // merge kernel: <= 4096 partial results
    *tps = (int)((n_tiles + s - 1) / s);
    *splits = (int)((n_tiles + *tps - 1) / *tps);
}

}  // namespace nafp

using namespace nafp;

extern "C" int64_t nafp_search_index_aux_floats(int64_t n_index) {
    return n_index < 0 ? -1 : (n_index + 63) / 64 * 64 + 64;
}

extern "C" int nafp_search_index_prepare(const float* index, int64_t n_index, int dim, float* aux, void* stream) {
    if (!index || !aux || n_index <= 0) return NAFP_ERR_INVALID_ARG;
    if (dim != 64 && dim != 128 && dim != 256) return NAFP_ERR_UNSUPPORTED;
    const int64_t n_pad = nafp_search_index_aux_floats(n_index);
    search_half_norms_kernel<<<(unsigned)((n_pad + 255) / 256), 256, 0, (hipStream_t)stream>>>(index, aux, n_index, n_pad, dim);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

extern "C" int64_t nafp_search_workspace_bytes(int64_t n_query, int64_t n_index, int k) {
    if (n_query < 0 || n_index <= 0 || k <= 0 || k > 32 || n_query > (1 << 30)) return -1;
    const int K = k <= 20 ? 20 : 32;
    int splits, tps;
    plan_splits((int)n_query, n_index, K, &splits, &tps);
    return (int64_t)n_query * 2 * splits * K * 8 + 256;
}

extern "C" int nafp_search_topk_l2(const float* query, int64_t n_query, const float* index, const float* aux,
                                   int64_t n_index, int dim, int k, float* out_dist, int32_t* out_ids,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    if (!query || !index || !aux || !out_dist || !out_ids || !workspace || n_query < 0 || n_index <= 0 || k <= 0)
        return NAFP_ERR_INVALID_ARG;
    if ((dim != 64 && dim != 128 && dim != 256) || k > 32 || n_index >= ((int64_t)1 << 31) || n_query > (1 << 30)) return NAFP_ERR_UNSUPPORTED;
    if (n_query == 0) return NAFP_OK;
    if (workspace_bytes < nafp_search_workspace_bytes(n_query, n_index, k)) return NAFP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int K = k <= 20 ? 20 : 32;
    int splits, tps;
    plan_splits((int)n_query, n_index, K, &splits, &tps);
    char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* pk = (float*)ws;
    int* pi = (int*)(ws + (int64_t)n_query * 2 * splits * K * 4);
    int rc;
    if (dim == 128) rc = K == 20 ? launch_topk<128, 20>(query, (int)n_query, index, aux, n_index, pk, pi, splits, tps, st)
                                 : launch_topk<128, 32>(query, (int)n_query, index, aux, n_index, pk, pi, splits, tps, st);
    else if (dim == 256) rc = K == 20 ? launch_topk<256, 20>(query, (int)n_query, index, aux, n_index, pk, pi, splits, tps, st)
                                      : launch_topk<256, 32>(query, (int)n_query, index, aux, n_index, pk, pi, splits, tps, st);
    else            rc = K == 20 ? launch_topk<64, 20>(query, (int)n_query, index, aux, n_index, pk, pi, splits, tps, st)
                                 : launch_topk<64, 32>(query, (int)n_query, index, aux, n_index, pk, pi, splits, tps, st);
    if (rc != NAFP_OK) return rc;
    const int M = 2 * splits * K;
    if (K == 20) search_merge_kernel<20><<<(unsigned)n_query, 64, M * 8, st>>>(pk, pi, query, out_dist, out_ids, 2 * splits, dim, k);
    else         search_merge_kernel<32><<<(unsigned)n_query, 64, M * 8, st>>>(pk, pi, query, out_dist, out_ids, 2 * splits, dim, k);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

extern "C" int nafp_search_seq_scores(const float* query, const float* index, int64_t n_index, int dim,
                                      const int32_t* task_q0, const int32_t* task_len, int64_t n_tasks,
                                      const int32_t* cand, int n_slots, float* out_scores, void* stream) {
    if (!query || !index || !task_q0 || !task_len || !cand || !out_scores || n_tasks < 0 || n_slots <= 0 || n_index <= 0)
        return NAFP_ERR_INVALID_ARG;
    if (dim != 64 && dim != 128 && dim != 256) return NAFP_ERR_UNSUPPORTED;
    const int64_t total = n_tasks * n_slots;
    if (total == 0) return NAFP_OK;
    const unsigned blocks = (unsigned)((total + 3) / 4);
    if (dim == 128) search_seq_score_kernel<128><<<blocks, 256, 0, (hipStream_t)stream>>>(query, index, task_q0, task_len, cand, out_scores, n_index, n_slots, total);
    else if (dim == 256) search_seq_score_kernel<256><<<blocks, 256, 0, (hipStream_t)stream>>>(query, index, task_q0, task_len, cand, out_scores, n_index, n_slots, total);
    else            search_seq_score_kernel<64><<<blocks, 256, 0, (hipStream_t)stream>>>(query, index, task_q0, task_len, cand, out_scores, n_index, n_slots, total);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// ---- in-training mini search (model/utils/mini_search_subroutines.py:28-220) ------------------------------
namespace nafp {

// scores[q, x] = max(|q|^2 + |x|^2 - 2 q.x, 0)  (mode 0)  or  q.x  (mode 1); any dim (1024 for the un-projected
// features f(.), 128 for the fingerprints).  32 x 32 output tile per workgroup, operands staged through LDS.
__global__ __launch_bounds__(256) void pairwise_scores_kernel(const float* __restrict__ Q, const float* __restrict__ X,
                                                              float* __restrict__ out, int nQ, int nD, int D, int mode) {
    __shared__ float sq[32][33], sx[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const int q0 = blockIdx.y * 32, x0 = blockIdx.x * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f}, xx = 0.f;
    for (int k0 = 0; k0 < D; k0 += 32) {
        for (int r = ty; r < 32; r += 8) {
            sq[r][tx] = (q0 + r < nQ && k0 + tx < D) ? Q[(int64_t)(q0 + r) * D + k0 + tx] : 0.f;
            sx[r][tx] = (x0 + r < nD && k0 + tx < D) ? X[(int64_t)(x0 + r) * D + k0 + tx] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float xv = sx[tx][k];
            xx = fmaf(xv, xv, xx);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float qv = sq[ty + 8 * j][k]; acc[j] = fmaf(qv, xv, acc[j]); qq[j] = fmaf(qv, qv, qq[j]); }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = q0 + ty + 8 * j, x = x0 + tx;
        if (q < nQ && x < nD) out[(int64_t)q * nD + x] = mode == 1 ? acc[j] : fmaxf(qq[j] + xx - 2.f * acc[j], 0.f);
    }
}

// rank[t] = number of candidate starts c whose length-s diagonal sum beats the ground truth's (conv_eye_func +
// argsort + np.where(sorted == gt)): smaller sum wins in mode 0, larger in mode 1; equal sums: smaller id first.
__global__ __launch_bounds__(256) void diag_rank_kernel(const float* __restrict__ scores, int* __restrict__ rank, int nQ, int nD,
                                                        int s, int mode, int gt_offset) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int n_c = nD - s + 1, gt = t + gt_offset;
    __shared__ float ref;
    __shared__ int cnt[4];
    if (tid == 0) {
        float r = 0.f;
        for (int i = 0; i < s; ++i) r += scores[(int64_t)(t + i) * nD + gt + i];
        ref = r;
    }
    __syncthreads();
    const float r = ref;
    int mine = 0;
    for (int c = tid; c < n_c; c += 256) {
        float v = 0.f;
        for (int i = 0; i < s; ++i) v += scores[(int64_t)(t + i) * nD + c + i];
        const bool better = mode == 1 ? (v > r || (v == r && c > gt)) : (v < r || (v == r && c < gt));   // argsort order incl. the reversal
        mine += better ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((tid & 63) == 0) cnt[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) rank[t] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

}  // namespace nafp

extern "C" int nafp_minisearch_scores(const float* query, const float* db, int64_t n_query, int64_t n_db, int dim, int mode,
                                      float* out_scores, void* stream) {
    if (!query || !db || !out_scores || n_query <= 0 || n_db <= 0 || dim <= 0) return NAFP_ERR_INVALID_ARG;
    if ((mode != 0 && mode != 1) || n_query > (1 << 20) || n_db > (1 << 20)) return NAFP_ERR_UNSUPPORTED;
    pairwise_scores_kernel<<<dim3((unsigned)((n_db + 31) / 32), (unsigned)((n_query + 31) / 32)), 256, 0, (hipStream_t)stream>>>(
        query, db, out_scores, (int)n_query, (int)n_db, dim, mode);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

extern "C" int nafp_minisearch_ranks(const float* scores, int64_t n_query, int64_t n_db, int scope, int mode, int gt_id_offset,
                                     int32_t* out_rank, void* stream) {
    if (!scores || !out_rank || scope <= 0 || n_query < scope || n_db < scope) return NAFP_ERR_INVALID_ARG;
    if (mode != 0 && mode != 1) return NAFP_ERR_UNSUPPORTED;
    const int n_t = (int)(n_query - scope + 1);
    if (gt_id_offset < 0 || n_t - 1 + gt_id_offset > n_db - scope) return NAFP_ERR_INVALID_ARG;
    diag_rank_kernel<<<n_t, 256, 0, (hipStream_t)stream>>>(scores, out_rank, (int)n_query, (int)n_db, scope, mode, gt_id_offset);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}
