// Probe: K-loop designs for the exact 3-way bf16 split (float32-equivalent products on the bf16 matrix pipe) at conv1's GEMM shape
// (M = 640 * 2048 rows, K = 384, N = 128), outside the conv kernel: C (M, N) f32 = A (M, K) f32 x B^T, B given pre-split as in
// split_weights_bf16_kernel (hm: per row and group of 16 k [h(16) | m(16)] bf16; l: (N, K) bf16).
//   tile 256 x 128, 4 waves, each wave 64 rows x ALL 128 columns: the A rows of a wave are private to it (no duplicate split, no
//   barrier for A), the weights are shared through an LDS ring; A ring NA deep, B ring NB deep, 2 workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o x6_gemm_probe x6_gemm_probe.hip ;  run: ./x6_gemm_probe [rows_per_640 = 2048]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void lds_dma16(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// exact 3-way split of 8 floats (two float4) into three bf16x8
__device__ __forceinline__ void split8(const float4 f0, const float4 f1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const float x[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        const float r1 = x[e] - (float)h;
        const __bf16 m = (__bf16)r1;
        hi[e] = h; mid[e] = m; lo[e] = (__bf16)(r1 - (float)m);
    }
}

// conv1-like source geometry: X (B, 256, 16, 128) f32; GEMM row tile = 32 output positions x 8 samples (sample minor), position = (f', t'),
// t' minor, 128 x 16 output positions per sample; tap k of output (f', t') reads X[b][2 f' + k][t'][:]; K-step t = (tap t / 8, channels 16 (t % 8))
__device__ __host__ __forceinline__ long long a_row_off(int tile, int row) {
    const int sg = tile % 80, pb = tile / 80;            // 640 / 8 sample groups, 64 position blocks
    const int b = sg * 8 + (row & 7), pos = pb * 32 + (row >> 3);
    const int fo = pos >> 4, to = pos & 15;
    return ((long long)(b * 256 + 2 * fo) * 16 + to) * 128;
}
#ifndef KORD
#define KORD 0          // 1: channel-major K order (the three taps of a channel block in consecutive steps)
#endif
__device__ __host__ __forceinline__ int a_step_off(int t) { return KORD ? (t % 3) * 16 * 128 + (t / 3) * 16 : (t >> 3) * 16 * 128 + (t & 7) * 16; }      // floats
__device__ __host__ __forceinline__ int b_step(int t) { return KORD ? (t % 3) * 8 + t / 3 : t; }      // the 16-k group of the weights that step t multiplies

template <int NA, int NB, int MODE>      // MODE bits: 1 no DMA inside the loop (stale operands); 2 no split (m = l = h); 4 DMA only; 8 B in per-step contiguous blocks; 16 no A DMA; 32 no B DMA
__global__ __launch_bounds__(256, 2) void x6_gemm_256x128(const float* __restrict__ A, const unsigned short* __restrict__ Bhm,
                                                         const unsigned short* __restrict__ Bl, float* __restrict__ C, int M, int K, int N, unsigned a_bytes) {
    constexpr int BM = 256, BN = 128;
    constexpr int A_STAGE = BM * 16;                // floats
    constexpr int B_STAGE = BN * 16 + BN * 8;       // floats: hm (64 B per row) | l (32 B per row)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + NA * A_STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int S = K / 16;

    // DMA geometry
    unsigned voffA[4], voffB[2], voffL;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 64 + q * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        voffA[q] = (unsigned)(a_row_off(blockIdx.x, row) + lc * 4) * 4u;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 32 + q * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        voffB[q] = (MODE & 8) ? (unsigned)((n0 + row) * 16 + lc * 4) * 4u : (unsigned)((n0 + row) * K + lc * 4) * 4u;
    }
    {
        const int row = wave * 32 + (lane >> 1);
        const int lh = (lane & 1) ^ ((row >> 3) & 1);
        voffL = ((MODE & 8) ? (unsigned)((n0 + row) * 16) * 2u : (unsigned)((n0 + row) * K) * 2u) + (unsigned)lh * 16u;
    }
    const u32x4 rsA = make_rsrc(A, (unsigned)a_bytes);
    const u32x4 rsB = make_rsrc(Bhm, (unsigned)((size_t)N * K * 4));
    const u32x4 rsL = make_rsrc(Bl, (unsigned)((size_t)N * K * 2));
    const unsigned ldsA0 = (unsigned)(unsigned long long)(lds_ptr_t)sA + (unsigned)(wave * 64 * 64);
    const unsigned ldsB0 = (unsigned)(unsigned long long)(lds_ptr_t)sB + (unsigned)(wave * 32 * 64);
    const unsigned ldsL0 = (unsigned)(unsigned long long)(lds_ptr_t)sB + (unsigned)(BN * 64 + wave * 32 * 32);
#define DMA_A(t_)                                                                              \
    if (!(MODE & 16)) { const unsigned sl_ = (unsigned)((t_) % NA) * (A_STAGE * 4);            \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) lds_dma16(ldsA0 + sl_ + q * 16 * 64, voffA[q], rsA, (unsigned)(a_step_off(t_) * 4)); }
#define DMA_B(t_)                                                                              \
    if (!(MODE & 32)) { const unsigned sl_ = (unsigned)((t_) % NB) * (B_STAGE * 4);            \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) lds_dma16(ldsB0 + sl_ + q * 16 * 64, voffB[q], rsB, (unsigned)(b_step(t_) * ((MODE & 8) ? N * 64 : 64))); \
      lds_dma16(ldsL0 + sl_, voffL, rsL, (unsigned)(b_step(t_) * ((MODE & 8) ? N * 32 : 32))); }

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // prologue: the steady-state issue order is B(t + NB - 1), A(t + NA - 1) per step
    constexpr int PRE = NA > NB ? NA : NB;
#pragma unroll
    for (int t = 0; t < PRE - 1; ++t) {
        if (t < NB - 1 && t < S) DMA_B(t)
        if (t < NA - 1 && t < S) DMA_A(t)
    }
    const int swz = (rl >> 2) & 3;
    const int aoff = (wave * 64 + rl) * 16;
    if (MODE & 64) {
        bf16x8 ah[2], am[2], al[2], bh[4], bm[4], bl[4];
        for (int j = 0; j < 8; ++j) {
            for (int mi = 0; mi < 2; ++mi) { ah[mi][j] = (__bf16)A[(lane * 8 + j) * 7 + mi]; am[mi][j] = (__bf16)A[(lane * 8 + j) * 5 + mi + 1000]; al[mi][j] = (__bf16)A[(lane * 8 + j) * 3 + mi + 2000]; }
            for (int ni = 0; ni < 4; ++ni) { bh[ni][j] = (__bf16)A[(lane * 8 + j) * 11 + ni + 3000]; bm[ni][j] = (__bf16)A[(lane * 8 + j) * 13 + ni + 4000]; bl[ni][j] = (__bf16)A[(lane * 8 + j) * 17 + ni + 5000]; }
        }
        for (int s = 0; s < S; ++s) {
            if (!(MODE & 128)) __builtin_amdgcn_s_barrier();
            if (MODE & 2048) {
                // product-major: consecutive MFMAs go to DIFFERENT accumulators (8 independent ones between two uses of the same)
#define PROD(A_, B_) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[mi], B_[ni], acc[mi][ni], 0, 0, 0);
                PROD(al, bh) PROD(am, bm) PROD(ah, bl) PROD(am, bh) PROD(ah, bm) PROD(ah, bh)
#undef PROD
            } else
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bm[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bm[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
            asm volatile("" : "+v"(ah[0]), "+v"(bh[0]));
        }
    } else
    for (int s = 0; s < S; ++s) {
        // B(s) and A(s) have landed: what may stay in flight is what was issued after the later of the two
        constexpr int KEEP = (NA == NB) ? (NA - 2) * 7 : (NA == NB + 1 ? (NB - 2) * 7 + 4 : 0);
        if (s + PRE - 1 >= S || (MODE & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(KEEP) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!(MODE & 1)) {
            if (s + NB - 1 < S) DMA_B(s + NB - 1)
            if (s + NA - 1 < S) DMA_A(s + NA - 1)
        }
        if (MODE & 4) continue;
        const float* St = sA + (s % NA) * A_STAGE;
        const float* Sb = sB + (s % NB) * B_STAGE;
        float4 af[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (MODE & 512) { const float t_ = 0.001f * (lane + s + mi + c); af[mi][c] = make_float4(t_, t_ * 1.5f, t_ * 0.7f, -t_); }      // no A reads
                else af[mi][c] = *(const float4*)(St + aoff + mi * 32 * 16 + (((2 * hh + c) ^ swz) * 4));
            }
        bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (MODE & 1024) {                                                                                                                       // no B reads
                for (int j = 0; j < 8; ++j) { bh[ni][j] = (__bf16)(0.01f * (lane - j + s)); bm[ni][j] = (__bf16)(1e-4f * (lane + j)); bl[ni][j] = (__bf16)(1e-6f * (j + ni)); }
                continue;
            }
            bh[ni] = *(const bf16x8*)(Sb + (ni * 32 + rl) * 16 + ((hh ^ swz) * 4));
            bm[ni] = *(const bf16x8*)(Sb + (ni * 32 + rl) * 16 + (((2 + hh) ^ swz) * 4));
            bl[ni] = *(const bf16x8*)(Sb + BN * 16 + (ni * 32 + rl) * 8 + ((hh ^ ((rl >> 3) & 1)) * 4));
        }
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            split8(af[mi][0], af[mi][1], ah[mi], am[mi], al[mi]);
            if (MODE & 2) { am[mi] = ah[mi]; al[mi] = ah[mi]; }
        }
        if (MODE & 256) {                                                                                                                            // no MFMAs: the operands are summed on the vector ALU instead
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[mi][ni][j] += (float)al[mi][j] * (float)bh[ni][j] + (float)am[mi][j] * (float)bm[ni][j] + (float)ah[mi][j] * (float)bl[ni][j];
        } else
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bm[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bm[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // epilogue: plain stores
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wave * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                C[(size_t)row * N + n0 + ni * 32 + rl] = acc[mi][ni][r];
            }
}


// ---- software-pipelined form: the operand reads, the activation split of step s + 1 and the DMA issue are spread over the four
// 12-MFMA column groups of step s; the barrier of step s + 1 sits in front of the last group.
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[2][4], const int ni, const bf16x8 (&ah)[2], const bf16x8 (&am)[2], const bf16x8 (&al)[2],
                                           const bf16x8 bh, const bf16x8 bm, const bf16x8 bl) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bm, acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bh, acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bm, acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
}

template <int NA, int NB, int SCHED, int MODE = 0>
__global__ __launch_bounds__(256, 2) void x6_gemm_256x128_sp(const float* __restrict__ A, const unsigned short* __restrict__ Bhm,
                                                            const unsigned short* __restrict__ Bl, float* __restrict__ C, int M, int K, int N, unsigned a_bytes) {
    constexpr int BM = 256, BN = 128;
    constexpr int A_STAGE = BM * 16;
    constexpr int B_STAGE = BN * 16 + BN * 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + NA * A_STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int S = K / 16;
    unsigned voffA[4], voffB[2], voffL;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 64 + q * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        voffA[q] = (unsigned)(a_row_off(blockIdx.x, row) + lc * 4) * 4u;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 32 + q * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        voffB[q] = (unsigned)((n0 + row) * K + lc * 4) * 4u;
    }
    {
        const int row = wave * 32 + (lane >> 1);
        const int lh = (lane & 1) ^ ((row >> 3) & 1);
        voffL = (unsigned)((n0 + row) * K) * 2u + (unsigned)lh * 16u;
    }
    const u32x4 rsA = make_rsrc(A, (unsigned)a_bytes);
    const u32x4 rsB = make_rsrc(Bhm, (unsigned)((size_t)N * K * 4));
    const u32x4 rsL = make_rsrc(Bl, (unsigned)((size_t)N * K * 2));
    const unsigned ldsA0 = (unsigned)(unsigned long long)(lds_ptr_t)sA + (unsigned)(wave * 64 * 64);
    const unsigned ldsB0 = (unsigned)(unsigned long long)(lds_ptr_t)sB + (unsigned)(wave * 32 * 64);
    const unsigned ldsL0 = (unsigned)(unsigned long long)(lds_ptr_t)sB + (unsigned)(BN * 64 + wave * 32 * 32);

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int swz = (rl >> 2) & 3;
    const int aoff = (wave * 64 + rl) * 16;
    // per-lane LDS offsets (floats) of the operand fragments inside a stage
    const int a_c0 = aoff + (((2 * hh) ^ swz) * 4), a_c1 = aoff + (((2 * hh + 1) ^ swz) * 4);
    const int b_h = rl * 16 + ((hh ^ swz) * 4), b_m = rl * 16 + (((2 + hh) ^ swz) * 4), b_l = BN * 16 + rl * 8 + ((hh ^ ((rl >> 3) & 1)) * 4);
#define RD_A(dst_, t_)                                                                         \
    { const float* St_ = sA + ((t_) % NA) * A_STAGE;                                           \
      dst_[0][0] = *(const float4*)(St_ + a_c0); dst_[0][1] = *(const float4*)(St_ + a_c1);    \
      dst_[1][0] = *(const float4*)(St_ + a_c0 + 32 * 16); dst_[1][1] = *(const float4*)(St_ + a_c1 + 32 * 16); }
#define RD_B(h_, m_, l_, t_, ni_)                                                              \
    { const float* Sb_ = sB + ((t_) % NB) * B_STAGE;                                           \
      h_ = *(const bf16x8*)(Sb_ + b_h + (ni_) * 32 * 16); m_ = *(const bf16x8*)(Sb_ + b_m + (ni_) * 32 * 16); \
      l_ = *(const bf16x8*)(Sb_ + b_l + (ni_) * 32 * 8); }

    // prologue: A(0 .. NA - 1) private, B(0), B(1)
    // issue order: B(0), A(0), A(1) [, A(2)], B(1), then per step s (mid-step): B(s + 2), A(s + NA)
    DMA_B(0)
#pragma unroll
    for (int t = 0; t < NA; ++t) if (t < S) DMA_A(t)
    if (1 < S) DMA_B(1)
    // A(0), B(0) landed: everything but the later A's and B(1) -> conservative: wait for all but B(1)
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float4 an[2][2];
    bf16x8 ah0[2], am0[2], al0[2], ah1[2], am1[2], al1[2];
    bf16x8 bh0, bm0, bl0, bh1, bm1, bl1;
    RD_A(an, 0)
    RD_B(bh0, bm0, bl0, 0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (NA < S) DMA_A(NA)
    split8(an[0][0], an[0][1], ah0[0], am0[0], al0[0]);
    split8(an[1][0], an[1][1], ah0[1], am0[1], al0[1]);

    // one K-step: the operands CUR_ (split A(s)) are ready, (bh0, bm0, bl0) = B(s, 0); leaves NXT_ = split A(s + 1) and B(s + 1, 0)
#define SG(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0);
#define K_STEP(TAIL_, s_, ahc, amc, alc, ahn, amn, aln)                                        \
    {                                                                                          \
        /* G0: MFMAs of column group 0; reads of A(s + 1) and B(s, 1) */                        \
        RD_A(an, (s_) + 1)                                                                     \
        RD_B(bh1, bm1, bl1, (s_), 1)                                                           \
        mfma_group(acc, 0, ahc, amc, alc, bh0, bm0, bl0);                                      \
        if (SCHED & 1) { SG(0x100, 7) SG(0x008, 12) }                                              \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        /* G1: column group 1; B(s, 2); split of the first 32 rows of A(s + 1) */               \
        RD_B(bh0, bm0, bl0, (s_), 2)                                                           \
        mfma_group(acc, 1, ahc, amc, alc, bh1, bm1, bl1);                                      \
        split8(an[0][0], an[0][1], ahn[0], amn[0], aln[0]);                                    \
        if (SCHED & 1) { SG(0x100, 3) _Pragma("unroll") for (int i_ = 0; i_ < 11; ++i_) { SG(0x008, 1) SG(0x002, 4) } SG(0x008, 1) } \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        /* G2: column group 2; B(s, 3); split of the other 32 rows */                           \
        RD_B(bh1, bm1, bl1, (s_), 3)                                                           \
        mfma_group(acc, 2, ahc, amc, alc, bh0, bm0, bl0);                                      \
        split8(an[1][0], an[1][1], ahn[1], amn[1], aln[1]);                                    \
        if (SCHED & 1) { SG(0x100, 3) _Pragma("unroll") for (int i_ = 0; i_ < 11; ++i_) { SG(0x008, 1) SG(0x002, 4) } SG(0x008, 1) } \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        /* the barrier of step s + 1: B(s + 1) has landed everywhere, nobody reads slot B(s) any more (its last group is in registers) */ \
        if (TAIL_) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                 \
        else if (SCHED & 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                       \
        __builtin_amdgcn_s_barrier();                                                          \
        if (!(SCHED & 2)) { if (!(TAIL_) || (s_) + 2 < S) DMA_B((s_) + 2)                      \
        if (!(TAIL_) || (s_) + 1 + NA < S) DMA_A((s_) + 1 + NA) }                              \
        /* G3: column group 3; B(s + 1, 0) */                                                  \
        RD_B(bh0, bm0, bl0, (s_) + 1, 0)                                                       \
        mfma_group(acc, 3, ahc, amc, alc, bh1, bm1, bl1);                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        /* A(s + 2) is read at the top of the next step: my own pieces of it have landed */     \
        if (TAIL_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            \
        else if (SCHED & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(7 * (NA - 1)) : "memory");              \
    }
    int s = 0;
    for (; s + 1 + NA + 2 < S; s += 2) {          // every guard of both steps holds: branch-free body
        K_STEP(false, s, ah0, am0, al0, ah1, am1, al1)
        K_STEP(false, s + 1, ah1, am1, al1, ah0, am0, al0)
    }
    for (; s < S; s += 2) {
        K_STEP(true, s, ah0, am0, al0, ah1, am1, al1)
        K_STEP(true, s + 1, ah1, am1, al1, ah0, am0, al0)
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wave * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                C[(size_t)row * N + n0 + ni * 32 + rl] = acc[mi][ni][r];
            }
}

static unsigned short f2bf(float x) {      // round to nearest even
    unsigned u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <typename KT>
static float time_kernel(KT kern, dim3 grid, int lds, const float* A, const unsigned short* Bhm, const unsigned short* Bl, float* C, int M, int K, int N, int reps) {
    const unsigned a_bytes = 640u * 256 * 16 * 128 * 4;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) kern<<<grid, 256, lds>>>(A, Bhm, Bl, C, M, K, N, a_bytes);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) kern<<<grid, 256, lds>>>(A, Bhm, Bl, C, M, K, N, a_bytes);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int rows_per = argc > 1 ? atoi(argv[1]) : 2048;
    const int M = 640 * rows_per, K = 384, N = 128;
    std::vector<float> hA((size_t)640 * 256 * 16 * 128), hB((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = rnd() * 2.0f;
    for (auto& v : hB) v = rnd() * 0.2f;
    std::vector<unsigned short> hHm((size_t)N * K * 2), hL((size_t)N * K);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float x = hB[(size_t)n * K + k];
            const unsigned short h = f2bf(x); const float r1 = x - bf2f(h);
            const unsigned short m = f2bf(r1); const unsigned short l = f2bf(r1 - bf2f(m));
            const size_t base = (size_t)n * 2 * K + (size_t)(k >> 4) * 32;
            hHm[base + (k & 15)] = h; hHm[base + 16 + (k & 15)] = m; hL[(size_t)n * K + k] = l;
        }
    float *dA, *dC; unsigned short *dHm, *dL;
    CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMalloc(&dHm, hHm.size() * 2)); CHECK(hipMalloc(&dL, hL.size() * 2));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dHm, hHm.data(), hHm.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dL, hL.data(), hL.size() * 2, hipMemcpyHostToDevice));
    const dim3 grid(M / 256, N / 128);
    const unsigned a_bytes = 640u * 256 * 16 * 128 * 4;
    const double flop = 2.0 * M * K * N;
    auto report = [&](const char* name, float ms) {
        printf("%-34s %.3f ms  %.1f TFLOP/s f32-equivalent  (matrix pipe %.0f %% of 2.5 PF)\n", name, ms, flop / ms * 1e-9, 6.0 * flop / ms * 1e-9 / 2500.0 * 100.0);
    };
    const int ldsA = 256 * 16 * 4, ldsB = (128 * 16 + 128 * 8) * 4;
    // correctness of the default variant against a float64 host product on the first 256 rows
    CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
    {
        auto kern = x6_gemm_256x128<3, 2, 0>;
        const int lds = 3 * ldsA + 2 * ldsB;
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        kern<<<grid, 256, lds>>>(dA, dHm, dL, dC, M, K, N, a_bytes);
        CHECK(hipDeviceSynchronize());
        std::vector<float> hC((size_t)512 * N);
        CHECK(hipMemcpy(hC.data(), dC + (size_t)(M - 512) * N, hC.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, ref_max = 0;
        for (int i = 0; i < 512; ++i)
            for (int n = 0; n < N; ++n) {
                double w = 0;
                for (int k = 0; k < K; ++k) { const size_t o = (size_t)a_row_off((M - 512 + i) / 256, (M - 512 + i) % 256) + a_step_off(k >> 4) + (k & 15); w += (o < hA.size() ? (double)hA[o] : 0.0) * (double)hB[(size_t)n * K + b_step(k >> 4) * 16 + (k & 15)]; }
                emax = fmax(emax, fabs(w - hC[(size_t)i * N + n])); ref_max = fmax(ref_max, fabs(w));
            }
        printf("check: max |C - A B^T| = %.3g  (max |C| = %.3g, relative %.3g)\n", emax, ref_max, emax / ref_max);
    }
    auto check = [&](const char* name) {
        std::vector<float> hC((size_t)512 * N);
        CHECK(hipMemcpy(hC.data(), dC + (size_t)(M - 512) * N, hC.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, ref_max = 0;
        for (int i = 0; i < 512; ++i)
            for (int n = 0; n < N; ++n) {
                double w = 0;
                for (int k = 0; k < K; ++k) { const size_t o = (size_t)a_row_off((M - 512 + i) / 256, (M - 512 + i) % 256) + a_step_off(k >> 4) + (k & 15); w += (o < hA.size() ? (double)hA[o] : 0.0) * (double)hB[(size_t)n * K + b_step(k >> 4) * 16 + (k & 15)]; }
                emax = fmax(emax, fabs(w - hC[(size_t)i * N + n])); ref_max = fmax(ref_max, fabs(w));
            }
        printf("check %s: max |C - A B^T| = %.3g  (relative %.3g)\n", name, emax, emax / ref_max);
    };
    {
        CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
        time_kernel(x6_gemm_256x128_sp<2, 2, 1>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 1); check("sp<2,2,1>");
        CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
        time_kernel(x6_gemm_256x128_sp<3, 2, 0>, grid, 3 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 1); check("sp<3,2,0>");
    }
    for (int rep = 0; rep < 2; ++rep) {
        report("sp 256x128 NA=2 NB=2 sched", time_kernel(x6_gemm_256x128_sp<2, 2, 1>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("sp 256x128 NA=3 NB=2 sched", time_kernel(x6_gemm_256x128_sp<3, 2, 1>, grid, 3 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2", time_kernel(x6_gemm_256x128<2, 2, 0>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 no DMA in loop", time_kernel(x6_gemm_256x128<2, 2, 1>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 MFMA only (registers), barrier", time_kernel(x6_gemm_256x128<2, 2, 1 | 64>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 MFMA only (registers), product-major order", time_kernel(x6_gemm_256x128<2, 2, 1 | 64 | 2048>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 MFMA only (registers), no barrier", time_kernel(x6_gemm_256x128<2, 2, 1 | 64 | 128>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 DMA only", time_kernel(x6_gemm_256x128<2, 2, 4>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 DMA only, A only", time_kernel(x6_gemm_256x128<2, 2, 4 | 32>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 DMA only, B only", time_kernel(x6_gemm_256x128<2, 2, 4 | 16>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 DMA only, B only, B contiguous", time_kernel(x6_gemm_256x128<2, 2, 4 | 16 | 8>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 DMA only, B contiguous", time_kernel(x6_gemm_256x128<2, 2, 4 | 8>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 B contiguous (wrong B)", time_kernel(x6_gemm_256x128<2, 2, 8>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 no B DMA", time_kernel(x6_gemm_256x128<2, 2, 32>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
        report("256x128 NA=2 NB=2 no A DMA", time_kernel(x6_gemm_256x128<2, 2, 16>, grid, 2 * ldsA + 2 * ldsB, dA, dHm, dL, dC, M, K, N, 20));
    }
    return 0;
}
