// Encoder convolutions for gfx950 (MI355X).
//
// Replaces the 16 keras Conv2D -> ELU -> LayerNormalization groups of the reference
// (model/fp/nnfp.py:48-79, eight ConvLayer blocks at nnfp.py:210-216).
//
// Every conv is a dense channel-mixing 3-tap conv along ONE axis, i.e. an implicit
// GEMM with m = (sample, f_out, t_out), K = 3*Cin, N = Cout, computed in exact fp32
// on the matrix cores (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak).
//
// LayerNorm is folded OUT of the operand path.  With v = ELU(conv + bias) the
// activation of conv j-1, (mu_b, r_b) its per-sample mean / rstd over (F,T,C) and
// (gamma, beta) its per-element affine, the next conv needs
//     conv_j( (v - mu_b) r_b gamma + beta )
//   = r_b * conv_j(gamma . v)  -  mu_b r_b * conv_j(gamma)  +  conv_j(beta)
// (zero padding applies to every term alike).  So each conv STORES z = gamma . v
// (its own LN scale applied in the epilogue, where the data is in registers anyway)
// and the consumer multiplies raw z tiles on the MFMAs, then finishes in ITS epilogue
//     out = r_b * acc + (-mu_b r_b) * G[pos, n] + Hb[pos, n]
// with G = conv_j(gamma), Hb = conv_j(beta) + bias_j: two (positions, Cout) tensors
// that depend only on the weights and are rebuilt at set_weights time by this same
// kernel in PLAIN mode.  The K-loop therefore moves raw tiles only (no gamma/beta
// loads, no per-element math), activations cross HBM exactly once in each direction,
// and the per-sample sum / sum-of-squares of v for the NEXT conv come out of the
// epilogue as before.
#include "nafp_common.h"
#include <hip/hip_ext.h>

#include <cstdlib>

namespace nafp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
#ifndef NAFP_Z_AUX
#define NAFP_Z_AUX 0          // cache policy of the epilogue's z stores (2 = nt)
#endif
#ifndef NAFP_T_AUX
#define NAFP_T_AUX 2          // ... of the pre-activation kept for the backward pass (training epilogue): nt -- it is not read again before the backward pass (forward at BSZ 5120: 27.35 -> 27.05 ms)
#endif
// bits of a float as the int the buffer-store builtins take.  By value on purpose: hipcc (ROCm 7.2) miscompiles
// __builtin_bit_cast(int, v.y) on a vector ELEMENT expression -- it reads element 0.
__device__ __forceinline__ int f2i(float x) { return __builtin_bit_cast(int, x); }

// keras ELU(alpha = 1) on a pair: max(t, exp(min(t, 0)) - 1) -- equal to elu1() (nafp_common.h) except where exp(t) - 1
// rounds below t (|t| < 1e-7: a difference of < 6e-8).  min(exp2(m), 1) with exp2 >= 0 is v_exp_f32's clamp modifier.
__device__ __forceinline__ f32x2 elu2(f32x2 t) {
    const f32x2 m = t * 1.44269504088896341f;
    f32x2 e;
    e.x = fminf(fmaxf(__builtin_amdgcn_exp2f(m.x), 0.f), 1.f);
    e.y = fminf(fmaxf(__builtin_amdgcn_exp2f(m.y), 0.f), 1.f);
    e = e - 1.0f;
    f32x2 r;
    r.x = fmaxf(t.x, e.x); r.y = fmaxf(t.y, e.y);
    return r;
}

// ============================================================================
// conv0: b0.conv1x3, Cin = 1 (nnfp.py:48-53 on the (F,T,1) log-mel input).
// Pure store-bandwidth: 3 FMAs per output element, 2 MB written per segment.
// One workgroup = one sample x `ROWS0` frequency rows; 32 threads x float4 cover
// the Cout = 128 channels of one (f, t_out) position, a wave stores 1 KiB
// contiguous.  Stores z = gamma0 . v; statistics are those of v.
// ============================================================================


// STORE = false: statistics only (the activation is re-generated inside conv1, see FUSE0).
//
// The input rows of the workgroup (ROWS0 x Tin floats) are staged in LDS once -- the deferred tail of the log-mel
// layer is applied there, once per input element -- and a thread then works on batches of NP positions: all gamma
// loads of a batch are issued before the first store.  (The straightforward loop -- load, compute, store per position
// -- made every iteration wait for its own loads behind the previous iteration's store: 3.9 TB/s; see DESIGN.md.)
template <bool STORE, int ROWS0>
__global__ __launch_bounds__(256) void conv0_kernel(
        const float* __restrict__ feat, const float* __restrict__ w3, const float* __restrict__ bias,
        const float* __restrict__ gamma, float* __restrict__ y, float* __restrict__ v_out,
        stat_t* __restrict__ stats, int F, int Tin, int Tout, int Cout, int stride, int pad,
        const float* __restrict__ gstat, int group_size, int segment_norm, int ident_stats) {
    constexpr int NP = 4;                           // positions per thread and batch
    __shared__ float s_x[ROWS0 * 64 + 8];           // rows of the input, Tin <= 64, with one zero in front (index -1)
    __shared__ double red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blocks_per_sample = (F + ROWS0 - 1) / ROWS0;
    const int64_t b = blockIdx.x / blocks_per_sample;
    const int f0 = (blockIdx.x % blocks_per_sample) * ROWS0;
    const int cgroups = Cout / 4;                   // float4 groups per position
    const int pos_per_iter = 256 / cgroups;         // positions covered per iteration (8 for Cout=128)
    const int cg = tid % cgroups, pslot = tid / cgroups;
    const int rows = min(ROWS0, F - f0);
    const int npos = rows * Tout;
    // gstat != null: `feat` is the RAW log-mel of the front end (melspec.hip with NAFP_MELSPEC_DEFER) and the
    // batch-max subtraction, clamp and optional segment normalisation (melspectrogram.py:108-111) happen here, on
    // load -- the same float operations in the same order as melspec_finalize_kernel, so the result is bit-identical
    // and the log-mel tensor crosses HBM once less in each direction.
    {
        float gmax = 0.f, nh = 0.f, nd = 1.f;
        if (gstat) {
            const int64_t g = group_size > 0 ? b / group_size : 0;
            gmax = gstat[2 * g];
            if (segment_norm) {
                const float mn = fmaxf(gstat[2 * g + 1] - gmax, -80.f);
                nh = mn / 2.f; nd = fabsf(nh + 1e-10f);
            }
        }
        const float* xin = feat + (b * F + f0) * (int64_t)Tin;
        for (int i = tid; i < rows * Tin; i += 256) {
            float v = xin[i];
            if (gstat) {
                v = max_keep_nan(v - gmax, -80.f);
                if (segment_norm) v = (v - nh) / nd;
            }
            s_x[(i / Tin) * 64 + (i % Tin)] = v;
        }
    }
    const float4 w0 = *(const float4*)(w3 + 0 * Cout + 4 * cg);
    const float4 w1 = *(const float4*)(w3 + 1 * Cout + 4 * cg);
    const float4 w2 = *(const float4*)(w3 + 2 * Cout + 4 * cg);
    const float4 bb = *(const float4*)(bias + 4 * cg);
    float* yout = STORE ? y + ((b * F + f0) * (int64_t)Tout) * Cout : nullptr;
    const float* gin = STORE ? gamma + ((int64_t)f0 * Tout) * Cout : nullptr;
    __syncthreads();
    float s = 0.f, q = 0.f;
    // software pipeline over batches of NP positions: the gamma loads of batch i+1 are in flight while batch i computes
    float4 g[NP], gn[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = pslot + i * pos_per_iter;
        g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (STORE && p < npos) g[i] = *(const float4*)(gin + (int64_t)p * Cout + 4 * cg);
    }
    for (int p0 = pslot; p0 < npos; p0 += NP * pos_per_iter) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = p0 + (NP + i) * pos_per_iter;
            gn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (STORE && p < npos) gn[i] = *(const float4*)(gin + (int64_t)p * Cout + 4 * cg);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = p0 + i * pos_per_iter;
            if (p >= npos) continue;
            const int r = p / Tout, to = p - r * Tout;
            const int t0 = to * stride - pad;
            const float* xr = s_x + r * 64;
            const float x0 = (t0 >= 0 && t0 < Tin) ? xr[t0] : 0.f;          // conv zero padding (of the NORMALISED features)
            const float x1 = (t0 + 1 >= 0 && t0 + 1 < Tin) ? xr[t0 + 1] : 0.f;
            const float x2 = (t0 + 2 >= 0 && t0 + 2 < Tin) ? xr[t0 + 2] : 0.f;
            float4 tq, v;
            tq.x = fmaf(x2, w2.x, fmaf(x1, w1.x, fmaf(x0, w0.x, bb.x)));
            tq.y = fmaf(x2, w2.y, fmaf(x1, w1.y, fmaf(x0, w0.y, bb.y)));
            tq.z = fmaf(x2, w2.z, fmaf(x1, w1.z, fmaf(x0, w0.z, bb.z)));
            tq.w = fmaf(x2, w2.w, fmaf(x1, w1.w, fmaf(x0, w0.w, bb.w)));
            v.x = elu1(tq.x); v.y = elu1(tq.y); v.z = elu1(tq.z); v.w = elu1(tq.w);
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            if (STORE) {
#ifndef NAFP_CONV0_PLAIN_STORE      // streaming (nt) stores: 0.286 -> 0.257 ms per 640 segments on the same box; nothing re-reads z0 before it has left the L2
                {
                    typedef float f4nt __attribute__((ext_vector_type(4)));
                    const f4nt zz = {v.x * g[i].x, v.y * g[i].y, v.z * g[i].z, v.w * g[i].w};
                    __builtin_nontemporal_store(zz, (f4nt*)(yout + (int64_t)p * Cout + 4 * cg));
                }
#else
                *(float4*)(yout + (int64_t)p * Cout + 4 * cg) = make_float4(v.x * g[i].x, v.y * g[i].y, v.z * g[i].z, v.w * g[i].w);
#endif
                if (v_out) *(float4*)(v_out + (yout - y) + (int64_t)p * Cout + 4 * cg) = tq;   // training keeps the pre-activation
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) g[i] = gn[i];
    }
    double ds = wave_sum((double)s), dq = wave_sum((double)q);
    if (lane == 0) { red[wave] = ds; red[4 + wave] = dq; }
    __syncthreads();
    if (tid == 0) {
        stat_add(stats + 2 * b, red[0] + red[1] + red[2] + red[3], ident_stats != 0);
        stat_add(stats + 2 * b + 1, red[4] + red[5] + red[6] + red[7], ident_stats != 0);
    }
}

int launch_conv0(const float* feat, const float* w3, const float* bias, const float* gamma, float* y,
                 float* v_out, stat_t* stats, int64_t B, const ConvGeom& g, hipStream_t st, const float* gstat,
                 int group_size, int segment_norm, bool ident_stats) {
    if (g.Cin != 1 || g.axis != 0 || (g.Cout % 4) != 0 || 256 % (g.Cout / 4) != 0 || g.Tin > 64) return NAFP_ERR_UNSUPPORTED;
    // rows per workgroup, measured at B = 640 on one box (ms): 4 -> 0.339, 8 -> 0.291, 16 -> 0.272 (the plain per-position
    // loop of round 1 ran 0.344; a store-only kernel of this shape 0.230: tools/probes/store_probe.hip)
    static const int rows = []() { const char* e = getenv("NAFP_CONV0_ROWS"); return e ? atoi(e) : 16; }();
    if (rows == 4) {
        const int64_t blocks = B * ((g.Fin + 3) / 4);
        conv0_kernel<true, 4><<<dim3((unsigned)blocks), 256, 0, st>>>(feat, w3, bias, gamma, y, v_out, stats, g.Fin, g.Tin, g.Tout,
                                                                    g.Cout, g.stride, g.pad, gstat, group_size, segment_norm, ident_stats ? 1 : 0);
    } else if (rows == 16) {
        const int64_t blocks = B * ((g.Fin + 15) / 16);
        conv0_kernel<true, 16><<<dim3((unsigned)blocks), 256, 0, st>>>(feat, w3, bias, gamma, y, v_out, stats, g.Fin, g.Tin, g.Tout,
                                                                     g.Cout, g.stride, g.pad, gstat, group_size, segment_norm, ident_stats ? 1 : 0);
    } else if (rows == 32) {
        const int64_t blocks = B * ((g.Fin + 31) / 32);
        conv0_kernel<true, 32><<<dim3((unsigned)blocks), 256, 0, st>>>(feat, w3, bias, gamma, y, v_out, stats, g.Fin, g.Tin, g.Tout,
                                                                     g.Cout, g.stride, g.pad, gstat, group_size, segment_norm, ident_stats ? 1 : 0);
    } else {
        const int64_t blocks = B * ((g.Fin + 7) / 8);
        conv0_kernel<true, 8><<<dim3((unsigned)blocks), 256, 0, st>>>(feat, w3, bias, gamma, y, v_out, stats, g.Fin, g.Tin, g.Tout,
                                                                    g.Cout, g.stride, g.pad, gstat, group_size, segment_norm, ident_stats ? 1 : 0);
    }
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// conv0's statistics alone (the fused forms generate its activation inside conv1): one THREAD per output position, the channels in a
// loop with conv0's weights as wave-uniform scalar operands -- 3 FMAs, the ELU and two accumulations per element and nothing else (no
// LDS, no address arithmetic; conv0_kernel<false> spends more on its indexing than on the arithmetic: 177 us per 640 segments against
// ~70 us of vector issue time).  A workgroup owns 256 consecutive positions of one sample; the sums leave as ONE fixed-point partial per
// workgroup and statistic, formed in a fixed order.
__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* __restrict__ feat, const float* __restrict__ w3, const float* __restrict__ bias,
                                                         stat_t* __restrict__ stats, int F, int Tin, int Tout, int Cout, int stride, int pad,
                                                         const float* __restrict__ gstat, int group_size, int segment_norm) {
    __shared__ double red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = F * Tout, blocks_per_sample = (P + 255) / 256;
    const int64_t b = blockIdx.x / blocks_per_sample;
    const int pos = (blockIdx.x % blocks_per_sample) * 256 + tid;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    const bool live = pos < P;
    if (live) {
        float gmax = 0.f, nh = 0.f, nd = 1.f;
        if (gstat) {
            const int64_t g = group_size > 0 ? b / group_size : 0;
            gmax = gstat[2 * g];
            if (segment_norm) { const float mn = fmaxf(gstat[2 * g + 1] - gmax, -80.f); nh = mn / 2.f; nd = fabsf(nh + 1e-10f); }
        }
        const int f = pos / Tout, to = pos - f * Tout;
        const int t0 = to * stride - pad;
        const float* xr = feat + (b * F + f) * (int64_t)Tin;
        auto ld = [&](int t) {
            if (t < 0 || t >= Tin) return 0.f;
            float v = xr[t];
            if (gstat) { v = max_keep_nan(v - gmax, -80.f); if (segment_norm) v = (v - nh) / nd; }
            return v;
        };
        x0 = ld(t0); x1 = ld(t0 + 1); x2 = ld(t0 + 2);
    }
    float s = 0.f, q = 0.f;
#pragma unroll 8
    for (int c = 0; c < Cout; ++c) {
        const float v = elu1(fmaf(x2, w3[2 * Cout + c], fmaf(x1, w3[Cout + c], fmaf(x0, w3[c], bias[c]))));
        s += v; q = fmaf(v, v, q);
    }
    if (!live) { s = 0.f; q = 0.f; }
    const double ds = wave_sum((double)s), dq = wave_sum((double)q);
    if (lane == 0) { red[wave] = ds; red[4 + wave] = dq; }
    __syncthreads();
    if (tid == 0) {
        stat_add(stats + 2 * b, red[0] + red[1] + red[2] + red[3]);
        stat_add(stats + 2 * b + 1, red[4] + red[5] + red[6] + red[7]);
    }
}

int launch_conv0_stats(const float* feat, const float* w3, const float* bias, stat_t* stats, int64_t B,
                       const ConvGeom& g, hipStream_t st, const float* gstat, int group_size, int segment_norm) {
    if (g.Cin != 1 || g.axis != 0 || (g.Cout % 4) != 0 || 256 % (g.Cout / 4) != 0 || g.Tin > 64) return NAFP_ERR_UNSUPPORTED;
    static const bool old_form = []() { const char* e = getenv("NAFP_CONV0_STATS_OLD"); return e && e[0] == '1'; }();
    if (!old_form) {
        const int64_t bps = ((int64_t)g.Fin * g.Tout + 255) / 256;
        conv0_stats_kernel<<<dim3((unsigned)(B * bps)), 256, 0, st>>>(feat, w3, bias, stats, g.Fin, g.Tin, g.Tout, g.Cout, g.stride, g.pad,
                                                                      gstat, group_size, segment_norm);
        NAFP_LAUNCH_CHECK();
        return NAFP_OK;
    }
    constexpr int R = 32;
    const int64_t blocks = B * ((g.Fin + R - 1) / R);
    conv0_kernel<false, R><<<dim3((unsigned)blocks), 256, 0, st>>>(feat, w3, bias, nullptr, nullptr, nullptr, stats,
                                                                   g.Fin, g.Tin, g.Tout, g.Cout, g.stride, g.pad, gstat, group_size, segment_norm, 0);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// ============================================================================
// Implicit-GEMM conv, fp32 MFMA, operands staged by direct-to-LDS DMA.
//   tile BM x BN = 128 x 128, K-step BK (template), 256 threads = 4 waves as
//   2(M) x 2(N), each wave 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x2_f32
//   (64 accumulators).
//
//   Tile rows are PT output positions x ST samples (PT*ST = 128, sample index
//   minor, ST >= 4).  In the 32x32 C/D layout a lane's 4 consecutive registers are
//   4 consecutive rows = 4 samples at ONE position, so the position-indexed
//   epilogue operands (G, Hb, gamma_out) are fetched once per 4 outputs.
//
//   Staging: `buffer_load_dwordx4 ... lds` (LDS-DMA).  No VGPR round trip, no
//   ds_write, no per-element VALU in the K-loop.  The buffer descriptor of the A
//   operand is re-based per tile on the tile's first sample and sized to the valid
//   samples only, so rows outside the batch AND taps that fall into the conv's zero
//   padding are simply out-of-range lanes: the DMA writes zeros for them
//   (tools/probes/lds_dma_probe.hip verifies that on gfx950).
//   The LDS image of a DMA is lane-linear (16 B per lane, rows of BK floats
//   unpadded), so bank conflicts are avoided by XOR-swizzling the 16-B chunk index
//   with the row on the SOURCE side and applying the same involution when reading:
//   physical chunk = logical chunk ^ swz(row).  A lane's operand fetch is one
//   ds_read_b128 = 4 consecutive k of its row; the k permutation this implies (lane
//   half h of MFMA step j multiplies k = 8*kk + 4*h + j) is the same for A and B.
//   NSTAGE-deep LDS ring, one barrier per K-step: wait own DMA of step s ->
//   barrier -> issue DMA of step s+NSTAGE-1 into the buffer everybody just left ->
//   MFMAs of step s.  Taps that hit only zero padding for every row of the tile
//   are skipped.
//
//   What bounds the kernel (measured with nafp_conv_timeline): the f32 MFMA runs at the packed-f32
//   vector rate and shares the SIMD's issue time with every VALU instruction, so the kernel is written for
//   instruction count -- K-steps that issue nothing but MFMAs, LDS reads and DMA (skewed across the barrier, body
//   once per ring slot), an epilogue on sample pairs with v_pk_* arithmetic and straight-line buffer loads/stores
//   (one instantiation per epilogue kind), geometry once per position slot through LDS.
//   Env knobs for A/B runs (all read once): NAFP_BM256 (min 128-row tiles for the 256-row tile), NAFP_BN64 /
//   NAFP_BN64_TILES / NAFP_BN64_MIN (64-column tiles), NAFP_N64S2 (their 2-stage, 5-per-CU kernel), NAFP_SPLITK,
//   NAFP_SPLIT_INKERNEL (min tiles for the in-kernel split-K finish), NAFP_FWD_PLAN (force tile:split), NAFP_GRID3D,
//   NAFP_GEMM_PRIO, NAFP_CONV0_ROWS.
// ============================================================================
constexpr int BN = 128;          // BM (tile rows) is a template parameter: 128 (4 waves) or 256 (8 waves)

// Ablation switches for the kernel-time breakdowns of DESIGN.md (NAFP_ABL env -> ConvKernelParams::abl) exist only in a
// library built with -DNAFP_ABLATION (tools/build_variant.sh abl neural-audio-fp_amd/csrc/conv.hip -DNAFP_ABLATION): in the production build every test below is a
// compile-time 0 -- the runtime branches they put into the epilogue made hipcc serialise its loads and stores
// (an s_waitcnt vmcnt(0) per group), which cost far more than the branches themselves.
#ifdef NAFP_ABLATION
#define NAFP_ABL(p_, bits_) ((p_).abl & (bits_))
#elif defined(NAFP_ABL_CONST)      // compile-time ablation (-DNAFP_ABL_CONST=<bits>): no runtime branches, clean code
#define NAFP_ABL(p_, bits_) ((NAFP_ABL_CONST) & (bits_))
#else
#define NAFP_ABL(p_, bits_) (0)
#endif

struct ConvKernelParams {
    const float* x;           // (B, Fin, Tin, Cin)
    const float* wp;          // (Cout, 3*Cin)
    const float* wp_hm;       // PREC = 2: the same shape, every group of 16 k replaced by [h(16) | m(16)] bf16 (split_weights_multi_kernel)
    const unsigned short* wp_l;   // PREC = 2: (Cout, 3*Cin) bf16, the third term of the split
    const float* G;           // (P, Cout)      FULL
    const float* Hb;          // (P, Cout)      FULL
    const float* gamma_out;   // (P, Cout)      FULL
    const float* bias;        // (Cout) or null PLAIN
    const stat_t* stats_in;   // (B, 2)         FULL   (64-bit fixed point: stat_add() / stat_get(), nafp_common.h)
    stat_t* stats_out;        // (B, 2)         FULL
    float* y;                 // (B, P, Cout)
    float* v_out;             // (B, P, Cout) or null: the ELU output itself (kept for the backward pass)
    int Fin, Tin, Cin, Tout, Cout;
    int axis, stride, pad;
    int B, P;                 // samples; output positions per sample (Fout*Tout)
    int PT, ST, log2ST;       // tile = PT positions x ST samples
    int n_sg;                 // sample groups = ceil(B / ST)
    int n_pb, log2_ncol;      // position blocks = ceil(P / PT); log2 of the column tiles (the XCD-aware 1-D grid, opt & 8)
    int xcd_group, xcd_full;  //   ... items per group, and the number of items in full blocks of 8 groups
    int64_t sample_in;        // Fin*Tin*Cin
    int tap_stride;           // elements between consecutive taps of one output row
    double inv_n_in;          // 1 / sample_in
    int mode;                 // 0 FULL, 1 PLAIN
    int dgrad;                // 1: transposed conv (backward w.r.t. the conv input), see row_geom()
    int perm_on, perm_n0, perm_c0;   // 1 = DGRAD, stride 2: tile rows enumerate positions parity class by class; 2 = forward conv along T: the last
                                     // output frame of every line goes last (tile_pos())
    unsigned wp_bytes;
    int n_split;              // split-K factor (blockIdx.z); > 1 writes raw partial sums to `y` = slab
    int abl;                  // ablation flags for kernel-time breakdown (NAFP_ABL env; 0 in production)
    unsigned* tickets;        // split-K with the finish in-kernel (EPI 4): one arrival counter per output tile, zero between launches
    float* y_final;           //   ... and the output tensor (`y` is the slab of partial sums there)
    unsigned long long* tl;   // diagnostic phase timeline (nafp_conv_timeline), null in production: 8 u64 per wave
    int opt;                  // bit 0: the geometry prologue runs at raised wave priority, bit 1: the epilogue does (NAFP_GEMM_PRIO); bit 2: 3-D grid;
                              // bit 3: 1-D grid in XCD-aware order, bit 4: ... row-fastest (see the block-id decode of conv_gemm_body)
    // FUSE0 (conv1 only): the A operand z0 = gamma0 . ELU(conv0(feat)) is generated in-kernel
    // from the log-mel features instead of being read from memory (`x` unused).
    const float* f0_feat;     // (B, F0, T0)
    const float* f0_w;        // (3, Cin)   conv0 kernel (Cin of this conv = Cout of conv0)
    const float* f0_bias;     // (Cin)
    const float* f0_gamma;    // (F0, Tin, Cin) = LN scale of conv0 = layout of z0
    int f0_T, f0_stride, f0_pad;   // conv0: input frames, stride and pad-before along T
    const float* f0_gstat;    // (or null) f0_feat is the RAW log-mel: (max, min) per group of f0_group samples, applied on load as conv0_kernel does
    int f0_group, f0_segnorm;
    ScalarsJob sj;            // PLAIN launches of the backward pass: side job (sc == null: none)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor in SGPRs (base, num_records bytes; stride 0, no swizzle).
__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}

// One LDS-DMA wave-instruction: lane l copies 16 B from (rsrc base + voff_l + soff) to LDS
// byte address lds_addr + 16*l; out-of-range lanes write zeros.  Inline asm on purpose:
// hipcc treats the builtin form as a pending LDS store and drains it (s_waitcnt vmcnt(0))
// before the next ds_read, which would expose the whole DMA latency on every K-step.
// The kernel counts these itself with s_waitcnt vmcnt(N).
__device__ __forceinline__ void lds_dma16(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);      // wave-uniform by construction; say so to the compiler
    soff = __builtin_amdgcn_readfirstlane(soff);              // (a uniform value it keeps in a VGPR is not a legal "s" operand)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// Source geometry of one GEMM row (an output position `pos` of this launch).
//   forward conv: output (fo,to) reads input positions pos0 + tap along the tap axis;
//   DGRAD (transposed conv, backward w.r.t. the conv input): the row is an INPUT position u of the
//   conv and tap k reads dT at o = (u + pad - k) / stride when that is an integer in [0, n_out).
//   There p.Fin/p.Tin/p.Cin describe the source tensor dT, p.Tout the minor extent of the rows,
//   and p.tap_stride = -S/stride (S = elements between neighbouring dT positions along the tap
//   axis), so that  inner + k*tap_stride  is the address of tap k whenever the tap is valid.
struct RowGeom { int inner; unsigned mask; };
__device__ __forceinline__ RowGeom row_geom(const ConvKernelParams& p, int pos) {
    const int fo = pos / p.Tout, to = pos - fo * p.Tout;
    RowGeom g; g.mask = 0;
    if (!p.dgrad) {
        int pos0, lim;
        if (p.axis == 0) { pos0 = to * p.stride - p.pad; lim = p.Tin; g.inner = (fo * p.Tin + pos0) * p.Cin; }
        else             { pos0 = fo * p.stride - p.pad; lim = p.Fin; g.inner = (pos0 * p.Tin + to) * p.Cin; }
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (pos0 + t >= 0 && pos0 + t < lim) g.mask |= 1u << t;
    } else {
        const int u = (p.axis == 0 ? to : fo) + p.pad;
        const int n_out = p.axis == 0 ? p.Tin : p.Fin;
        const int S = p.axis == 0 ? p.Cin : p.Tin * p.Cin;
        const int A0 = p.axis == 0 ? (fo * p.Tin) * p.Cin : to * p.Cin;
        const int sh = p.stride == 2 ? 1 : 0;                     // the encoder's strides are 1 and 2 (checked by the launcher)
        g.inner = A0 + ((u * S) >> sh);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int d = u - k;
            if (d >= 0 && (d & sh) == 0 && (d >> sh) < n_out) g.mask |= 1u << k;
        }
    }
    return g;
}

// Position served by row-index `idx` of the launch.  Identity, except for the transposed conv of a
// stride-2 layer: there an input coordinate u (+pad) of even parity receives taps {0, 2} and an odd
// one only tap {1}, so the positions are enumerated class by class (first all positions whose tap
// coordinate is = perm_c0 mod 2, then the others).  A tile then holds rows of ONE class (up to the
// single boundary tile) and the tile-wide dead-tap skip removes the structurally zero half of the
// K-steps instead of multiplying zeros.
//
// perm_on == 2 (FORWARD conv along T, the minor coordinate): the LAST output coordinate of every line is the one whose
// window hangs over the end of the input (TF SAME puts the odd padding element behind the data, nnfp.py:48-61), i.e. the
// only positions whose tap 2 -- or, for the stride-1 two-frame layer, whose tap 0 resp. 2 -- reads zero padding.  They are
// enumerated after all the others, so that a tile of PT positions (PT divides both class sizes, launch_conv_gemm) holds
// one class and the tile-wide dead-tap skip drops that tap's K-steps: 1/3 of conv8's work, 1/6 of conv6's, 1/12 and 1/24
// of conv4's and conv2's at the 1-s input.  (Along F the natural order already ends with that class: only PT matters.)
__device__ __forceinline__ int tile_pos(const ConvKernelParams& p, int idx) {
    if (!p.perm_on || idx >= p.P) return idx;
    if (p.perm_on == 2) {
        const int L = p.Tout, lines = p.P / L, n0 = L - 1;
        if (idx < lines * n0) { const int line = idx / n0; return line * L + (idx - line * n0); }
        return (idx - lines * n0) * L + n0;
    }
    const int n0 = p.perm_n0;
    if (p.axis == 0) {            // tap axis = minor coordinate, L = p.Tout per line
        const int L = p.Tout, lines = p.P / L, n1 = L - n0;
        if (idx < lines * n0) { const int line = idx / n0; return line * L + 2 * (idx - line * n0) + p.perm_c0; }
        const int j = idx - lines * n0, line = j / n1;
        return line * L + 2 * (j - line * n1) + 1 - p.perm_c0;
    }
    if (idx < n0 * p.Tout) { const int e = idx / p.Tout; return (2 * e + p.perm_c0) * p.Tout + (idx - e * p.Tout); }
    const int j = idx - n0 * p.Tout, e = j / p.Tout;
    return (2 * e + 1 - p.perm_c0) * p.Tout + (j - e * p.Tout);
}

// Sum over each 32-lane half of a wave with DPP row shifts (VALU only, no LDS crossbar): afterwards lane 31 holds the sum
// of lanes 0..31 and lane 63 the sum of lanes 32..63.
__device__ __forceinline__ float half_wave_sum_dpp(float x) {
    int v = __builtin_bit_cast(int, x);
#define NAFP_DPP_ADD(ctrl_, rmask_)                                                                       \
    v = __builtin_bit_cast(int, __builtin_bit_cast(float, v) +                                            \
                                    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, ctrl_, rmask_, 0xf, true)));
    NAFP_DPP_ADD(0x111, 0xf)      // row_shr:1
    NAFP_DPP_ADD(0x112, 0xf)      // row_shr:2
    NAFP_DPP_ADD(0x114, 0xf)      // row_shr:4
    NAFP_DPP_ADD(0x118, 0xf)      // row_shr:8   -> lane 15 of every 16-lane row holds the row's sum
    NAFP_DPP_ADD(0x142, 0xa)      // row_bcast:15 into rows 1 and 3: lanes 31 and 63 hold the half sums
#undef NAFP_DPP_ADD
    return __builtin_bit_cast(float, v);
}

// EXPERIMENT (-DNAFP_EXP_SKIP_TAP2=1, WRONG RESULTS): tap 2's activation rows are not fetched (zero rows instead): the L2 / HBM traffic
// an input-row-staged stride-2 tile would save, for an upper bound of what that tile could gain (profiles/r05_experiments.md)
#ifndef NAFP_EXP_X6_NOSPLIT
#define NAFP_EXP_X6_NOSPLIT 0      // EXPERIMENT (wrong results): the exact-split kernels without the m / l terms of the activation split
#endif
#ifndef NAFP_EXP_SKIP_TAP2
#define NAFP_EXP_SKIP_TAP2 0
#endif
// EPI selects the epilogue the instantiation carries (one per kernel: the others' code and registers are not in it):
//   0 FULL for inference (tile = 4 or 8 samples per position: statistics in registers), 1 the same + the pre-activation
//   kept for the backward pass, 2 FULL for any tile shape (statistics through LDS; v_out by a runtime test),
//   3 PLAIN (raw sums + bias: DGRAD, weight-derived tensors, split-K partial slabs).
// PREC = 1 (experimental, inference only, NAFP_OPT_BF16X3): the products are formed on the bf16 matrix pipe from f32 operands
// split in registers into hi + lo bf16 halves, hi*hi + hi*lo + lo*hi with f32 accumulation (lo*lo, ~2^-16 of a product,
// is dropped).  Not the arithmetic of the reference: results differ from the f32 path at the 1e-6 level of a fingerprint
// component; bench.py reports it as a separate object with its measured error and never as `value`.
// PREC = 2 (experimental, NAFP_OPT_BF16X3 = 2; inference, training-forward and plain / transposed-conv epilogues): the EXACT 3-way split x = h + m + l (three bf16 terms hold the 24
// significant bits of a float32) and the six products of relative weight >= 2^-16 -- hh, hm, mh, hl, mm, lh; ml + lm + ll < 2^-25 of
// |a||b|, below half an ulp of the float32 product -- with f32 accumulation: float32-equivalent arithmetic on the bf16 matrix pipe.
// Returns whether this workgroup ran the full epilogue (true) or left after handing in a split-K part (false).
template <int BM, int BNT, int BK, int NSTAGE, bool FUSE0, int EPI, int PREC = 0>
__device__ __forceinline__ bool conv_gemm_body(const ConvKernelParams& p) {
    static_assert(!FUSE0 || (BK == 16 && BM == 128), "the in-kernel conv0 generator is written for BK = 16, BM = 128");
    static_assert(BM == 128 || BM == 256, "tile rows");
    static_assert(BNT == 128 || (BNT == 64 && BM == 128 && !FUSE0), "tile columns");
    constexpr int NIW = BNT / 64;                  // 32-column MFMA tiles per wave (a wave covers 64 rows x BNT / 2 columns)
    constexpr int NT = 2 * BM;                     // threads: BM / 32 waves as (BM/64) x 2, each wave 64 x 64
    constexpr int NW = BM / 32;
    constexpr int CH = BK / 4;                     // 16-B chunks per row
    constexpr int RPI = 64 / CH;                   // rows covered by one DMA wave-instruction
    constexpr int NI = 32 / RPI;                   // A: DMA instructions per wave per step (every wave stages 32 rows of A)
    constexpr int BROWS = BNT / NW;                 // B: rows staged per wave (32 or 16)
    constexpr int NIB = BROWS / RPI;               // B: DMA instructions per wave per step
    static_assert(NIB >= 1, "B rows per wave");
    constexpr int TILE = (FUSE0 && PREC == 2) ? BM * 24 : BM * BK;      // floats of the A tile (fused exact split: three bf16 planes of BM x 16)
    constexpr int TILEB = BNT * BK;                 // floats of the B tile
    constexpr int LROWS = BNT / NW;                // PREC = 2: rows of the weights' third bf16 plane staged per wave (16 k x 2 B = 32 B per row and step)
    constexpr int TILEL = PREC == 2 ? BNT * 8 : 0;
    constexpr int STAGE = TILE + TILEB + TILEL;    // A | B (| L)
    constexpr int NDMA = NI + NIB + (PREC == 2 ? 1 : 0);   // DMA instructions per wave and K-step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [stage 0: A | B] ... [stage NSTAGE-1] [sRB[BM]] [sCB[BM]] [sPos[32]] [sInner[32]] [sMask[32]] [FUSE0: conv0 w (3,Cin) | bias (Cin)]
    float* sRB = smem + NSTAGE * STAGE;
    float* sCB = sRB + BM;
    int* sPos = (int*)(sCB + BM);                  // per position slot of the tile (PT <= 32): the position it serves,
    int* sInner = sPos + 32;                       //   the source offset of its tap 0 inside a sample,
    unsigned* sMask = (unsigned*)(sPos + 64);      //   and which taps read real data (row_geom())
    float* sW0 = sCB + BM + 96;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // wm < BM / 64
    // grid = (sample groups, position blocks, column tiles): no division to take a block id apart
    // (split-K launches keep (sample groups x position blocks, column tiles, parts): their dispatch order matters more)
    int sg = blockIdx.x, pb = blockIdx.y, colz = blockIdx.z, zsp = 0;      // ..., column tile, split-K part
    if (p.opt & 8) {
        // XCD-aware order on a 1-D grid: workgroup b runs on XCD b % 8 and every XCD has its own L2.  The
        // work items v are laid out so that the G items which share an operand are consecutive (G = p.xcd_group), and
        // consecutive GROUPS go to the XCDs round-robin: group g = 8 * block + x is executed by the workgroups
        // b = 8 * (block * G + i) + x, i < G -- all on XCD x, dispatched within 8 G ids of each other.  (A bijection on the
        // full blocks of 8 G items; the tail keeps v = b.)  The dispatch order stays the tile order, so tiles of different
        // cost -- the live-tap classes -- spread over the XCDs as on the plain grid.
        //   column-fastest (opt & 16 clear; activations are the big operand): G = column tiles x split-K parts of one row
        //     tile: they read its A rows from HBM once instead of once per column tile;
        //   row-fastest (opt & 16; the late convs, live weights > activations): G = the row tiles of one (column tile, part):
        //     that weight slice is fetched by ONE L2 instead of by all eight.
        const unsigned lin = blockIdx.x, G = (unsigned)p.xcd_group;
        unsigned v = lin;
        if (lin < (unsigned)p.xcd_full) {
            const unsigned xcd = lin & 7u, j = lin >> 3;
            const unsigned blk = j / G, i = j - blk * G;
            v = (blk * 8u + xcd) * G + i;
        }
        unsigned rt;
        if (p.opt & 16) {
            const unsigned n_rt = (unsigned)(p.n_sg * p.n_pb);
            unsigned t = v / n_rt; rt = v - t * n_rt;
            if (p.n_split > 1) { const unsigned c = t / (unsigned)p.n_split; zsp = (int)(t - c * (unsigned)p.n_split); t = c; }
            colz = (int)t;
        } else {
            colz = (int)(v & ((1u << p.log2_ncol) - 1u)); v >>= p.log2_ncol;
            if (p.n_split > 1) { const unsigned r = v / (unsigned)p.n_split; zsp = (int)(v - r * (unsigned)p.n_split); v = r; }
            rt = v;
        }
        pb = (int)(rt / (unsigned)p.n_sg); sg = (int)(rt - (unsigned)pb * (unsigned)p.n_sg);
    } else
    if (!(p.opt & 4)) { pb = blockIdx.x / p.n_sg; sg = blockIdx.x - pb * p.n_sg; colz = blockIdx.y; zsp = blockIdx.z; }
    if ((EPI == 3 || EPI == 5) && p.sj.sc) {
        // side job of a transposed conv of the backward pass: the scalar records the LayerNorm backward of the layer below
        // starts from (their inputs are final since the LayerNorm backward of this layer; NT samples per workgroup)
        const long long wg = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
        const long long n_wg = (long long)gridDim.x * gridDim.y * gridDim.z;
        for (long long b = wg * NT + tid; b < p.sj.B; b += n_wg * NT) {
            const float mean = p.sj.mr[2 * b], rstd = p.sj.mr[2 * b + 1];
            float* o = p.sj.sc + 8 * b;
            o[0] = mean; o[1] = rstd;
            o[2] = (float)(p.sj.lnsum[2 * b] * p.sj.inv_n); o[3] = (float)(p.sj.lnsum[2 * b + 1] * p.sj.inv_n);
            o[4] = 1.f / rstd;
            o[5] = p.sj.mr_prev ? p.sj.mr_prev[2 * b + 1] : 1.f;
            o[6] = p.sj.mr_prev ? -p.sj.mr_prev[2 * b] : 0.f;
            o[7] = 0.f;
        }
    }
    const int tile_n0 = colz * BNT;
    const int K = 3 * p.Cin;
    // diagnostic timeline: lane 0 of every wave stamps the shader clock at the phase boundaries of its tile
#define NAFP_TL(k_)                                                                            \
    if (p.tl && lane == 0)                                                                                 \
        p.tl[(((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) * 8 + wave) * 8 + (k_)] = \
            __builtin_readcyclecounter();
    if (p.tl && lane == 0)
        p.tl[(((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) * 8 + wave) * 8] =
            (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) |                    // HW_REG_HW_ID: wave/simd/cu/sh/se
            ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32);             // HW_REG_XCC_ID
    NAFP_TL(1)
    // A workgroup's prologue and epilogue are short VALU / memory-issue sections; next to the older waves' MFMA streams on
    // the same SIMDs they only get the issue slots those leave over, which stretches them several-fold and keeps the
    // workgroup's LDS and registers parked.  Raised priority lets them through (the MFMA streams lose a few slots).
    if (p.opt & 1) __builtin_amdgcn_s_setprio(3);
    const int ST1 = p.ST - 1;
    const int b0 = sg * p.ST;                       // first sample of this tile
    const int nb = min(p.ST, p.B - b0);             // valid samples in this tile

    // ---- source geometry of the tile's position slots (threads 0 .. PT-1): all rows of a slot differ only by the sample,
    // so the divisions of tile_pos() / row_geom() are done once per slot, not once per row and user.  The per-sample
    // LayerNorm statistics of the INPUT (FULL mode; threads 64 ..) are requested here and used further down ----
    const bool stat_thread = tid >= 64 && tid < 64 + p.ST;
    double st_sum = 0.0, st_sq = 0.0;             // loaded here, turned into (r, c) after the first DMAs are on their way
    if (stat_thread && p.mode != 1 && b0 + tid - 64 < p.B) {
        st_sum = stat_get(p.stats_in + 2 * (int64_t)(b0 + tid - 64));
        st_sq = stat_get(p.stats_in + 2 * (int64_t)(b0 + tid - 64) + 1);          // NaN for a poisoned sample (nafp_common.h)
        if (st_sq != st_sq && p.stats_out) stat_poison(p.stats_out + 2 * (int64_t)(b0 + tid - 64));   // ... which stays poisoned
    }
    if (tid < 32) {
        int pos = p.P, inner = 0; unsigned mask = 0;
        if (tid < p.PT) {
            pos = tile_pos(p, pb * p.PT + tid);
            if (pos < p.P) { const RowGeom rg = row_geom(p, pos); inner = rg.inner; mask = rg.mask; }
        }
        sPos[tid] = pos; sInner[tid] = inner; sMask[tid] = mask;
    }
    __syncthreads();

    // ---- DMA geometry.  Wave w stages rows [32w, 32w+32) of A and of B; instruction q
    // covers rows 32w + q*RPI + lane/CH, physical chunk pc = lane % CH, which must hold
    // the logical chunk pc ^ swz(row).
    constexpr unsigned OOB = 0x80000000u;
    unsigned voffA[NI];           // byte offset of tap 0 from the tile's first sample
    unsigned vmaskA[NI];          // bit t: tap t reads real data (else zero padding -> OOB lane)
    unsigned voffB[NIB];
#pragma unroll
    for (int q = 0; q < NIB; ++q) {
        const int lr = wave * BROWS + q * RPI + lane / CH;
        const int pc = lane % CH;
        const int swz = BK == 32 ? ((lr >> 1) & 7) : ((lr >> 2) & 3);
        voffB[q] = (unsigned)((tile_n0 + lr) * K + (pc ^ swz) * 4) * 4u;
    }
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int lr = wave * 32 + q * RPI + lane / CH;
        const int pc = lane % CH;
        const int swz = BK == 32 ? ((lr >> 1) & 7) : ((lr >> 2) & 3);
        const int lc = pc ^ swz;
        const int slot_q = lr >> p.log2ST;
        const int sl = lr & ST1;
        voffA[q] = 0; vmaskA[q] = 0;
        if (sPos[slot_q] < p.P && sl < nb) {
            voffA[q] = (unsigned)(sl * (int)p.sample_in + sInner[slot_q] + lc * 4) * 4u;   // may wrap for an invalid tap: masked
            vmaskA[q] = sMask[slot_q];
        }
    }
    // ---- FUSE0 generator geometry: thread t builds row t>>1, channels 8*(t&1)..+8 of every
    // K-step of the A tile: z0[b, f, t, c] = gamma0[f,t,c] * ELU(bias0[c] + sum_k w0[k,c] feat[b, f, t*s0 - p0 + k])
    float x0[3][3];            // [conv1 tap (along F)][conv0 tap (along T)] log-mel inputs of my row
    unsigned g_valid = 0;      // bit t: the z0 row of conv1 tap t exists (else conv1 zero padding)
    int g_off = 0;             // offset of conv1 tap 0, channel 0 inside z0 / gamma0
    const int g_row = tid >> 1, g_ch = (tid & 1) * 8;
    const int g_swz = (g_row >> 2) & 3;
    if (FUSE0) {
        for (int i = tid; i < 4 * p.Cin; i += NT) sW0[i] = i < 3 * p.Cin ? p.f0_w[i] : p.f0_bias[i - 3 * p.Cin];
#pragma unroll
        for (int t = 0; t < 3; ++t) { x0[t][0] = 0.f; x0[t][1] = 0.f; x0[t][2] = 0.f; }
        const int pos = pb * p.PT + (g_row >> p.log2ST);
        const int sl = g_row & ST1;
        if (pos < p.P && sl < nb) {
            float f0_gmax = 0.f, f0_nh = 0.f, f0_nd = 1.f;
            if (p.f0_gstat) {
                const int64_t gi = p.f0_group > 0 ? (int64_t)(b0 + sl) / p.f0_group : 0;
                f0_gmax = p.f0_gstat[2 * gi];
                if (p.f0_segnorm) { const float mn = fmaxf(p.f0_gstat[2 * gi + 1] - f0_gmax, -80.f); f0_nh = mn / 2.f; f0_nd = fabsf(f0_nh + 1e-10f); }
            }
            const int fo = pos / p.Tout, to = pos - fo * p.Tout;
            const int f00 = fo * p.stride - p.pad;           // this conv's taps run along F (axis 1)
            g_off = (f00 * p.Tin + to) * p.Cin;
            const int t00 = to * p.f0_stride - p.f0_pad;      // conv0's taps run along T
            const float* fb = p.f0_feat + ((int64_t)(b0 + sl) * p.Fin) * p.f0_T;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int f = f00 + t;
                if (f >= 0 && f < p.Fin) {
                    g_valid |= 1u << t;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int tt = t00 + k;
                        float xv = 0.f;
                        if (tt >= 0 && tt < p.f0_T) {
                            xv = fb[(int64_t)f * p.f0_T + tt];
                            if (p.f0_gstat) {            // the same float operations in the same order as conv0_kernel / melspec_finalize_kernel
                                xv = max_keep_nan(xv - f0_gmax, -80.f);
                                if (p.f0_segnorm) xv = (xv - f0_nh) / f0_nd;
                            }
                        }
                        x0[t][k] = xv;
                    }
                }
            }
        }
        __syncthreads();       // sW0 is read by the generator below
    }
    // Taps that read only zero padding for EVERY row of this tile are skipped (wave-wide OR over the slots' masks;
    // slots beyond P hold 0).
    unsigned live = 0;
    {
        const unsigned m = sMask[lane & 31];
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (__ballot((m >> t) & 1u) != 0ull) live |= 1u << t;
    }
    // live taps packed 2 bits each (no runtime-indexed array: that would live in scratch)
    unsigned tap_pack = 0; int n_live = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (live & (1u << t)) { tap_pack |= (unsigned)t << (2 * n_live); ++n_live; }
    const int cpt = p.Cin / BK;                 // K-steps per tap
    const int n_steps_all = n_live * cpt;
    // split-K: this workgroup owns K-steps [s_begin, n_steps) of the tile's live steps
    int s_begin = 0, n_steps = n_steps_all;
    if (p.n_split > 1) {
        s_begin = (int)((unsigned)(n_steps_all * zsp) / (unsigned)p.n_split);
        n_steps = (int)((unsigned)(n_steps_all * (zsp + 1)) / (unsigned)p.n_split);
    }
    // (tap slot, channel) of a K-step; without split-K the steps asked for are 0, 1, 2 and a tap has >= 2 steps: no division
#define NAFP_STEP_AT(step_, tsel_, c0_)                                                        \
    const int tsel_ = s_begin == 0 ? ((step_) >= cpt ? 1 : 0) : (step_) / cpt;                 \
    const int c0_ = ((step_) - tsel_ * cpt) * BK;

    const u32x4 rsA = make_rsrc(p.x + (int64_t)b0 * p.sample_in,
                                NAFP_ABL(p, 512) ? 0u : (unsigned)nb * (unsigned)p.sample_in * 4u);   // ablation 512: every A lane out of range (zero fill, no memory traffic)
    const u32x4 rsB = make_rsrc(PREC == 2 ? p.wp_hm : p.wp, NAFP_ABL(p, 4096) ? 0u : p.wp_bytes);      // ablation 4096: every B lane out of range
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr_t)smem + (unsigned)(wave * 32 * BK * 4);
    const unsigned ldsB0 = (unsigned)(unsigned long long)(lds_ptr_t)smem + (unsigned)((TILE + wave * BROWS * BK) * 4);
    // PREC = 2: the third plane of the pre-split weights, 32 B per row and K-step: lanes 0 .. 2 LROWS - 1 of every wave stage rows
    // wave * LROWS + lane / 2 (half lane & 1) behind the B tile, row-major
    const u32x4 rsL = make_rsrc((const float*)p.wp_l, NAFP_ABL(p, 4096) ? 0u : p.wp_bytes / 2);
    const unsigned ldsL0 = (unsigned)(unsigned long long)(lds_ptr_t)smem + (unsigned)((TILE + TILEB) * 4 + wave * LROWS * 32);
    const unsigned voffL = (unsigned)((tile_n0 + wave * LROWS + (lane >> 1)) * K) * 2u + (unsigned)(lane & 1) * 16u;
#define NAFP_DMA_L(slot_, k0_)                                                                 \
    if (PREC == 2 && lane < 2 * LROWS) lds_dma16(ldsL0 + (unsigned)((slot_) * STAGE * 4), voffL, rsL, (unsigned)((k0_) * 2));

    // Issue the DMA of K-step s_ into ring slot slot_ (wave-uniform LDS bases).
#define NAFP_DMA_STEP(s_, slot_)                                                              \
    {                                                                                          \
        NAFP_STEP_AT(s_, tsel_l, c0_l)                                                         \
        const int tap_l = (int)((tap_pack >> (2 * tsel_l)) & 3u);                              \
        const unsigned la_l = lds0 + (unsigned)((slot_) * STAGE * 4);                          \
        const unsigned lb_l = ldsB0 + (unsigned)((slot_) * STAGE * 4);                         \
        const unsigned tapb_l = (unsigned)(tap_l * p.tap_stride) * 4u;                         \
        _Pragma("unroll") for (int q = 0; q < NI; ++q) {                                       \
            const unsigned va = ((vmaskA[q] >> tap_l) & 1u) ? voffA[q] + tapb_l : OOB;         \
            if (!FUSE0) lds_dma16(la_l + q * RPI * BK * 4, va, rsA, (unsigned)(c0_l * 4));     \
            if (q < NIB) lds_dma16(lb_l + q * RPI * BK * 4, voffB[q], rsB,                      \
                                   (unsigned)((tap_l * p.Cin + c0_l) * 4));                    \
        }                                                                                      \
        NAFP_DMA_L(slot_, tap_l * p.Cin + c0_l)                                                \
    }
    // FUSE0: gamma0 of my 8 channels of K-step s_ (issued early), then build + store the A rows.
    float4 gg0 = make_float4(0.f, 0.f, 0.f, 0.f), gg1 = gg0;
#define NAFP_GEN_LOAD(s_)                                                                     \
    {                                                                                          \
        const int tsel_l = (s_) / cpt;                                                         \
        const int tap_l = (int)((tap_pack >> (2 * tsel_l)) & 3u);                              \
        const int c0_l = ((s_) - tsel_l * cpt) * BK;                                           \
        const int go_l = ((g_valid >> tap_l) & 1u) ? g_off + tap_l * p.tap_stride + c0_l + g_ch : 0; \
        gg0 = *(const float4*)(p.f0_gamma + go_l);                                             \
        gg1 = *(const float4*)(p.f0_gamma + go_l + 4);                                         \
    }
// (__fmul_rn: the product is rounded to float32 HERE, as conv0_kernel's store rounds it -- left to the compiler the multiply contracts into
// the first subtraction of the exact split and the three terms then hold the unrounded product: more precise, but not conv0's z0)
#define NAFP_GEN_ONE(j_, G_) \
    __fmul_rn((ok_l ? elu1(fmaf(xc_l, wt_l[2 * p.Cin + (j_)], fmaf(xb_l, wt_l[p.Cin + (j_)], fmaf(xa_l, wt_l[(j_)], wt_l[3 * p.Cin + (j_)])))) : 0.f), (G_))
#define NAFP_GEN_STORE(s_, slot_)                                                             \
    {                                                                                          \
        const int tsel_l = (s_) / cpt;                                                         \
        const int tap_l = (int)((tap_pack >> (2 * tsel_l)) & 3u);                              \
        const int c0_l = ((s_) - tsel_l * cpt) * BK;                                           \
        const bool ok_l = (g_valid >> tap_l) & 1u;                                             \
        const float xa_l = tap_l == 0 ? x0[0][0] : (tap_l == 1 ? x0[1][0] : x0[2][0]);         \
        const float xb_l = tap_l == 0 ? x0[0][1] : (tap_l == 1 ? x0[1][1] : x0[2][1]);         \
        const float xc_l = tap_l == 0 ? x0[0][2] : (tap_l == 1 ? x0[1][2] : x0[2][2]);         \
        const float* wt_l = sW0 + c0_l + g_ch;                                                 \
        float* dst_l = smem + (slot_) * STAGE + g_row * BK;                                    \
        *(float4*)(dst_l + (((2 * (tid & 1)) ^ g_swz) * 4)) = make_float4(                      \
            NAFP_GEN_ONE(0, gg0.x), NAFP_GEN_ONE(1, gg0.y), NAFP_GEN_ONE(2, gg0.z), NAFP_GEN_ONE(3, gg0.w)); \
        *(float4*)(dst_l + (((2 * (tid & 1) + 1) ^ g_swz) * 4)) = make_float4(                  \
            NAFP_GEN_ONE(4, gg1.x), NAFP_GEN_ONE(5, gg1.y), NAFP_GEN_ONE(6, gg1.z), NAFP_GEN_ONE(7, gg1.w)); \
    }

    f32x16 acc[2][NIW];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    NAFP_TL(2)
    if (p.opt & 1) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s_begin + s < n_steps) {
            NAFP_DMA_STEP(s_begin + s, s)
            if (FUSE0 && PREC != 2) {
                NAFP_GEN_LOAD(s_begin + s)
                NAFP_GEN_STORE(s_begin + s, s)
            }
        }

    // per-sample LayerNorm scalars of the input: r_b and -mu_b r_b (first needed by the epilogue, many barriers away)
    if (stat_thread) {
        float r = 0.f, c = 0.f;
        if (p.mode != 1 && b0 + tid - 64 < p.B) {
            if (p.inv_n_in < 0.0) {                               // identity statistics (stat_ln_scalars): r = 1, c = 0, NaN if poisoned
                const double z = st_sq * 0.0;
                r = (float)(1.0 + z); c = (float)z;
            } else {
                const double mean = st_sum * p.inv_n_in;
                double var = st_sq * p.inv_n_in - mean * mean;
                var = var < 0.0 ? 0.0 : var;                      // (keeps a NaN)
                const double rstd = 1.0 / sqrt(var + (double)LN_EPS);
                r = (float)rstd; c = (float)(-mean * rstd);
            }
        }
        sRB[tid - 64] = r; sCB[tid - 64] = c;
    }

    // operand read addresses (floats): row*BK + ((lc ^ swz(row)) * 4), lc = 2*kk + (lane>>5)
    const int rl = lane & 31, hh = lane >> 5;
    const int rswz = BK == 32 ? ((rl >> 1) & 7) : ((rl >> 2) & 3);     // wm*64, 32*mi do not change swz
    const int aoff = (wm * 64 + rl) * BK, boff = TILE + (wn * (BNT / 2) + rl) * BK;     // (NIB <= NI: the B pieces ride in the A loop)
    // epilogue operands (declared here: the first 32-row block's are requested under the last MFMAs of the K-loop)
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).
    const int ncol = lane & 31;
    const int n_base = tile_n0 + wn * (BNT / 2) + ncol;
    const int g4 = p.ST >> 2;                         // sample quads per position
    constexpr int MIL = 1;            // 32-row blocks whose operands are resident at once (2 = all 48 values: spills at 168 VGPRs)
    float Gv[MIL][4][NIW], Hv[MIL][4][NIW], gv[MIL][4][NIW];
    const int pc_bytes = p.P * p.Cout * 4;
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, pc_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Hb), 0, pc_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gamma_out), 0, pc_bytes, 0x00020000);
#define NAFP_EPI_LOAD(mi_, slot_)                                                              \
    _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                         \
        const int grp_l = wm * 16 + (mi_) * 8 + 2 * rg + (lane >> 5);                          \
        const int pos_l = sPos[grp_l >> (p.log2ST - 2)];               /* = tile_pos(), P beyond the end */ \
        const int pofs_l = pos_l < p.P ? (pos_l * p.Cout + n_base) * 4 : (int)0x80000000;      \
        _Pragma("unroll") for (int ni = 0; ni < NIW; ++ni) {                                   \
            if NAFP_ABL(p, 32) { Gv[slot_][rg][ni] = 0.5f; Hv[slot_][rg][ni] = 0.25f; gv[slot_][rg][ni] = 1.5f; continue; } \
            Gv[slot_][rg][ni] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsG, pofs_l, ni * 128, 0)); \
            Hv[slot_][rg][ni] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsH, pofs_l, ni * 128, 0)); \
            gv[slot_][rg][ni] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsg, pofs_l, ni * 128, 0)); \
        }                                                                                      \
    }
    int slot = 0;
    if NAFP_ABL(p, 8) {          // ablation: prologue + pipeline fill only (wait for the prefilled stages, then leave)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (acc[0][0][0] == 12345.678f) p.y[tid] = smem[tid];
        return false;
    }
    // Operand fragments of half a K-step (8 k-values): lane (row rl, half hh) reads logical chunk 2*kk + hh of its
    // A rows (2 x 32-row blocks) and B rows (NIW x 32-column blocks); one fragment set feeds 4 * 2 * NIW MFMAs.
#define NAFP_LD_FRAG(St_, kk_, a_, b_)                                                         \
    {                                                                                          \
        const int pc4_l = ((2 * (kk_) + hh) ^ rswz) * 4;                                       \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) a_[mi] = *(const float4*)((St_) + aoff + mi * 32 * BK + pc4_l);   \
        _Pragma("unroll") for (int ni = 0; ni < NIW; ++ni) b_[ni] = *(const float4*)((St_) + boff + ni * 32 * BK + pc4_l); \
    }
#define NAFP_MM_FRAG(a_, b_)                                                                   \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                           \
        _Pragma("unroll") for (int ni = 0; ni < NIW; ++ni) {                                   \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi].x, b_[ni].x, acc[mi][ni], 0, 0, 0); \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi].y, b_[ni].y, acc[mi][ni], 0, 0, 0); \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi].z, b_[ni].z, acc[mi][ni], 0, 0, 0); \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi].w, b_[ni].w, acc[mi][ni], 0, 0, 0); \
        }
#define NAFP_WAIT_STEP(s_)                                                                     \
    if (NSTAGE == 2 || (s_) + NSTAGE - 2 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NSTAGE - 2) * NDMA) : "memory");
    // Skewed K-loop (BK = 16 = two fragment sets per step): the second half of step s is multiplied AFTER the barrier of
    // step s + 1, so that every LDS read has a block of MFMAs whose operands are already in registers in front of it --
    //   wait, barrier | read R0 = first half of s | DMA of s + 2 | MFMA R1 (second half of s - 1) | read R1 = second
    //   half of s | MFMA R0 -- and a wave issues MFMAs back to back from one barrier to the next.
    if (!FUSE0 && BK == 16 && !NAFP_ABL(p, 4 | 2048) && s_begin < n_steps) {
        float4 a0[2], b0[NIW], a1[2], b1[NIW];
        // The DMA of step s + 2 is issued in two pieces BETWEEN the MFMA groups of the step (an in-order wave issues the
        // piece's scalar address arithmetic, m0 moves and buffer_loads in the gaps while the matrix pipe works through the
        // MFMAs in front of them; ahead of the first MFMA they were ~300 idle pipe cycles per step for a wave that is alone
        // on its SIMD).  The step to stage next is tracked by running (tap slot, channel) counters -- no division per step.
        // The lane part of the A addresses is kept per CURRENT tap (recomputed when the tap changes, once per Cin / 16 steps),
        // and the loop body exists once per ring slot (3 copies), so that every LDS address is a base register plus an
        // immediate: a K-step issues no VALU instruction besides its MFMAs.
        NAFP_STEP_AT(s_begin + NSTAGE - 1, d_tsel0, d_c00)
        int d_tsel = d_tsel0, d_c0 = d_c00;
        int d_tap = 0; unsigned d_tapb = 0;
        unsigned cur_va[NI];
#define NAFP_DMA_TAP()                                                                         \
        {                                                                                      \
            d_tap = (int)((tap_pack >> (2 * d_tsel)) & 3u); d_tapb = (unsigned)(d_tap * p.tap_stride) * 4u; \
            _Pragma("unroll") for (int q = 0; q < NI; ++q) cur_va[q] = (((vmaskA[q] >> d_tap) & 1u) && !(NAFP_EXP_SKIP_TAP2 && d_tap == 2)) ? voffA[q] + d_tapb : OOB; \
        }
        NAFP_DMA_TAP()
#define NAFP_DMA_PIECE(q_, slot_)                                                              \
        {                                                                                      \
            lds_dma16(lds0 + (unsigned)((slot_) * STAGE * 4) + (q_) * RPI * BK * 4, cur_va[q_], rsA, (unsigned)(d_c0 * 4)); \
            if ((q_) < NIB) lds_dma16(ldsB0 + (unsigned)((slot_) * STAGE * 4) + (q_) * RPI * BK * 4, voffB[(q_) < NIB ? (q_) : 0], rsB, \
                                      (unsigned)((d_tap * p.Cin + d_c0) * 4));                 \
            if ((q_) == NI - 1) { d_c0 += BK; if (d_c0 == p.Cin) { d_c0 = 0; ++d_tsel; NAFP_DMA_TAP() } } \
        }
#define NAFP_MM_HALF(a_, b_, mi_)                                                              \
        _Pragma("unroll") for (int ni = 0; ni < NIW; ++ni) {                                   \
            acc[mi_][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi_].x, b_[ni].x, acc[mi_][ni], 0, 0, 0); \
            acc[mi_][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi_].y, b_[ni].y, acc[mi_][ni], 0, 0, 0); \
            acc[mi_][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi_].z, b_[ni].z, acc[mi_][ni], 0, 0, 0); \
            acc[mi_][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi_].w, b_[ni].w, acc[mi_][ni], 0, 0, 0); \
        }
        static_assert(NI == 2, "the DMA of a step is issued as two pieces");
        if (PREC >= 1) {
            // one K-step = 16 k = one v_mfma_f32_32x32x16_bf16 per product term: lane (row rl, half hh) holds k = 8 hh .. 8 hh + 7
            // of its A rows and B rows (logical chunks 2 hh and 2 hh + 1 of the stage)
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            for (int s = s_begin; s < n_steps; ++s) {
                NAFP_WAIT_STEP(s)
                __builtin_amdgcn_s_barrier();
                if (s == s_begin) { NAFP_TL(3) }
                const bool has_next = s + NSTAGE - 1 < n_steps && !NAFP_ABL(p, 1);
                int nslot = slot + NSTAGE - 1; if (nslot >= NSTAGE) nslot -= NSTAGE;
                const float* St = smem + slot * STAGE;
                float4 af[2][2], bf[PREC == 2 ? 1 : NIW][2];
                bf16x8 ah[2], al[2], bh[NIW], bl[NIW];
                bf16x8 am[PREC == 2 ? 2 : 1], bm[PREC == 2 ? NIW : 1];      // PREC = 2: the middle term of the exact 3-way split x = h + m + l
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int pc4 = ((2 * hh + c) ^ rswz) * 4;
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) af[mi][c] = *(const float4*)(St + aoff + mi * 32 * BK + pc4);
                    if (PREC != 2) {
#pragma unroll
                        for (int ni = 0; ni < NIW; ++ni) bf[ni][c] = *(const float4*)(St + boff + ni * 32 * BK + pc4);
                    }
                }
                if (PREC == 2) {
                    // the weights arrive split (split_weights_multi_kernel): logical chunks 0, 1 of a row = h[0..7], h[8..15], chunks 2, 3 = m
#pragma unroll
                    for (int ni = 0; ni < NIW; ++ni) {
                        bh[ni] = *(const bf16x8*)(St + boff + ni * 32 * BK + ((hh ^ rswz) * 4));
                        bm[ni] = *(const bf16x8*)(St + boff + ni * 32 * BK + (((2 + hh) ^ rswz) * 4));
                        bl[ni] = *(const bf16x8*)(St + TILE + TILEB + (wn * (BNT / 2) + ni * 32 + rl) * 8 + hh * 4);
                    }
                }
                if (has_next) { NAFP_DMA_PIECE(0, nslot) NAFP_DMA_L(nslot, d_tap * p.Cin + d_c0) NAFP_DMA_PIECE(1, nslot) }
#define NAFP_SPLIT8(f_, hi_, mid_, lo_)                                                        \
                {                                                                                  \
                    const float x_l[8] = {f_[0].x, f_[0].y, f_[0].z, f_[0].w, f_[1].x, f_[1].y, f_[1].z, f_[1].w}; \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                \
                        const __bf16 h_l = (__bf16)x_l[e];                                         \
                        const float r1_l = x_l[e] - (float)h_l;                                    \
                        hi_[e] = h_l;                                                              \
                        if (PREC == 2 && NAFP_EXP_X6_NOSPLIT) {                                    \
                            mid_[e] = h_l; lo_[e] = h_l;      /* experiment: what the split costs */ \
                        } else if (PREC == 2) {                                                    \
                            const __bf16 m_l = (__bf16)r1_l;                                       \
                            mid_[e] = m_l; lo_[e] = (__bf16)(r1_l - (float)m_l);                   \
                        } else {                                                                   \
                            lo_[e] = (__bf16)r1_l;                                                 \
                        }                                                                          \
                    }                                                                              \
                }
                // PREC = 2: the same split written two k at a time -- one packed conversion per term and pair, the bf16 -> f32 back-conversions as a
                // shift and a mask of the packed word: 11 vector instructions per pair where the per-element form compiles to 12 - 13
#define NAFP_SPLIT8_PAIRS(f_, hi_, mid_, lo_)                                                  \
                {                                                                                  \
                    typedef __bf16 bf16x2_l __attribute__((ext_vector_type(2)));                   \
                    typedef float f32x2_l __attribute__((ext_vector_type(2)));                     \
                    typedef unsigned u32x4_l __attribute__((ext_vector_type(4)));                  \
                    const float x_l[8] = {f_[0].x, f_[0].y, f_[0].z, f_[0].w, f_[1].x, f_[1].y, f_[1].z, f_[1].w}; \
                    u32x4_l uh_l, um_l, ul_l;                                                      \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                \
                        f32x2_l x2; x2.x = x_l[2 * e]; x2.y = x_l[2 * e + 1];                      \
                        const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(x2, bf16x2_l)); \
                        f32x2_l r2; r2.x = x2.x - __uint_as_float(h2 << 16); r2.y = x2.y - __uint_as_float(h2 & 0xffff0000u); \
                        const unsigned m2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_l)); \
                        f32x2_l t2; t2.x = r2.x - __uint_as_float(m2 << 16); t2.y = r2.y - __uint_as_float(m2 & 0xffff0000u); \
                        uh_l[e] = h2; um_l[e] = m2; ul_l[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(t2, bf16x2_l)); \
                    }                                                                              \
                    hi_ = __builtin_bit_cast(bf16x8, uh_l); mid_ = __builtin_bit_cast(bf16x8, um_l); lo_ = __builtin_bit_cast(bf16x8, ul_l); \
                }
                if (PREC == 2 && !NAFP_EXP_X6_NOSPLIT) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) NAFP_SPLIT8_PAIRS(af[mi], ah[mi], am[PREC == 2 ? mi : 0], al[mi])
                } else {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) NAFP_SPLIT8(af[mi], ah[mi], am[PREC == 2 ? mi : 0], al[mi])
                }
#undef NAFP_SPLIT8_PAIRS
                if (PREC != 2) {
#pragma unroll
                    for (int ni = 0; ni < NIW; ++ni) NAFP_SPLIT8(bf[PREC == 2 ? 0 : ni], bh[ni], bm[0], bl[ni])
                }
#undef NAFP_SPLIT8
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NIW; ++ni) {
                        if (PREC == 2) {
                            // every product of relative weight >= 2^-16 (the three dropped ones sum to < 2^-25 of |a||b|), small terms first
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[PREC == 2 ? mi : 0], bm[PREC == 2 ? ni : 0], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[PREC == 2 ? mi : 0], bh[ni], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bm[PREC == 2 ? ni : 0], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        } else {
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        }
                    }
                if (++slot == NSTAGE) slot = 0;
            }
        } else {
        {
            NAFP_WAIT_STEP(s_begin)
            __builtin_amdgcn_s_barrier();
            NAFP_TL(3)
            const float* St = smem + slot * STAGE;
            NAFP_LD_FRAG(St, 0, a0, b0)
            NAFP_LD_FRAG(St, 1, a1, b1)
            __builtin_amdgcn_sched_barrier(0);
            if (s_begin + NSTAGE - 1 < n_steps && !NAFP_ABL(p, 1)) { NAFP_DMA_PIECE(0, NSTAGE - 1) NAFP_DMA_PIECE(1, NSTAGE - 1) }
            __builtin_amdgcn_sched_barrier(0);
            NAFP_MM_FRAG(a0, b0)
            __builtin_amdgcn_sched_barrier(0);
            slot = 1;
        }
        // one K-step on ring slot SLOT_ (compile-time); leaves the loop before the last step's second MFMA group
#define NAFP_K_STEP(SLOT_, NSLOT_)                                                             \
        {                                                                                      \
            /* R1 (the last reads of the previous slot) has landed: after the barrier that slot may be overwritten */ \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
            NAFP_WAIT_STEP(s)                                                                  \
            __builtin_amdgcn_s_barrier();                                                      \
            const bool has_next = s + NSTAGE - 1 < n_steps && !NAFP_ABL(p, 1);                 \
            const int nslot = (NSLOT_);                                                        \
            const float* St = smem + (SLOT_) * STAGE;                                          \
            NAFP_LD_FRAG(St, 0, a0, b0)                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            NAFP_MM_HALF(a1, b1, 0)                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            if (has_next) { NAFP_DMA_PIECE(0, nslot) }                                         \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            NAFP_MM_HALF(a1, b1, 1)                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            if (has_next) { NAFP_DMA_PIECE(1, nslot) }                                         \
            NAFP_LD_FRAG(St, 1, a1, b1)                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            if (s == n_steps - 1) break;          /* the last step's second MFMA group runs below, behind the operand requests */ \
            NAFP_MM_FRAG(a0, b0)                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            ++s;                                                                               \
        }
        static_assert(PREC != 0 || NSTAGE == 3 || BNT == 64, "the K-loop body is written out once per ring slot (128-column tiles)");
        if (s_begin + 1 < n_steps) {
            if (BNT == 128) {
                for (int s = s_begin + 1;;) {
                    NAFP_K_STEP(1, 0)
                    NAFP_K_STEP(2, 1)
                    NAFP_K_STEP(0, 2)
                }
            } else {            // 64-column tiles: one rolled copy (the three-copy form makes hipcc spill at their register budget).
                // The last step is written out behind the loop instead of leaving it through a mid-body break: with the break
                // hipcc copies the 12 fragment registers of R1 at the end of every iteration.
                int rs = 1;
                int s = s_begin + 1;
                for (; s < n_steps - 1;) {
                    NAFP_K_STEP(rs, (rs == 0 ? NSTAGE - 1 : rs - 1))
                    if (++rs == NSTAGE) rs = 0;
                }
                do { NAFP_K_STEP(rs, (rs == 0 ? NSTAGE - 1 : rs - 1)) } while (false);     // s == n_steps - 1: leaves at its break
            }
        }
#undef NAFP_K_STEP
        // the epilogue's positional operands of the first 32-row block are requested HERE: their round trip (several
        // thousand cycles next to the other workgroups' operand and store traffic) runs under the last 16 or 32 MFMAs
        if ((EPI == 0 || EPI == 1) && BM == 128) { NAFP_EPI_LOAD(0, 0) }
        __builtin_amdgcn_sched_barrier(0);
        if (s_begin + 1 < n_steps) { NAFP_MM_FRAG(a0, b0) }
        NAFP_MM_FRAG(a1, b1)
        }
#undef NAFP_DMA_PIECE
#undef NAFP_DMA_TAP
#undef NAFP_MM_HALF
    } else if (FUSE0 && PREC == 2) {
        // conv1 with conv0 generated in-kernel, exact 3-way split (NSTAGE = 2, no split-K).  The A stream of the unfused launch -- 2 MB of z0
        // per segment written by conv0 and read back 1.5 times: what bounds conv1 at bf16-pipe speed (profiles/r06_experiments.md) -- does
        // not exist: a thread builds 8 channels of one A row of step s + 1 from the row's nine log-mel values, ALREADY SPLIT into three
        // bf16 planes, while the matrix pipe works on step s.  Branch-free: the tap of a step selects the row's inputs by scalar
        // conditions, a tap that reads this conv's zero padding is an out-of-range lane of the gamma0 descriptor (gamma = 0 -> z0 = 0),
        // conv0's weights come from LDS with immediate offsets (its channel count is a constant here).
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        static_assert(!(FUSE0 && PREC == 2) || NSTAGE == 2, "two ring slots");
        constexpr int C0 = 128;                                                    // conv0's Cout = this conv's Cin (launch_conv_gemm checks)
        const int a_rd = (wm * 64 + rl) * 8 + ((hh ^ ((rl >> 3) & 1)) * 4);          // floats: row * 32 B + 16-byte half
        const int a_wr = g_row * 8 + (((tid & 1) ^ ((g_row >> 3) & 1)) * 4);
        const __amdgpu_buffer_rsrc_t rsG0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.f0_gamma), 0, (int)((unsigned)p.sample_in * 4u), 0x00020000);
        int gvoff0, gvoff1, gvoff2;
        gvoff0 = (g_valid & 1u) ? (g_off + g_ch) * 4 : (int)0x80000000;
        gvoff1 = (g_valid & 2u) ? (g_off + p.tap_stride + g_ch) * 4 : (int)0x80000000;
        gvoff2 = (g_valid & 4u) ? (g_off + 2 * p.tap_stride + g_ch) * 4 : (int)0x80000000;
        const float* wrow = sW0 + g_ch;                                            // [w(tap 0) | w(tap 1) | w(tap 2) | bias] x C0, + channel
        int g_tsel = 0, g_c0 = 0;                                                  // (tap slot, channel block) of the step to generate next
        float4 wq[8], gq[2];
        float gxa = 0.f, gxb = 0.f, gxc = 0.f;
        // operands of the step to generate: requested early (global gamma0, LDS weights), consumed by NAFP_X6_GEN below
#define NAFP_X6_GEN_REQ()                                                                      \
        {                                                                                      \
            const int tap_l = (int)((tap_pack >> (2 * g_tsel)) & 3u);                          \
            gxa = tap_l == 0 ? x0[0][0] : (tap_l == 1 ? x0[1][0] : x0[2][0]);                  \
            gxb = tap_l == 0 ? x0[0][1] : (tap_l == 1 ? x0[1][1] : x0[2][1]);                  \
            gxc = tap_l == 0 ? x0[0][2] : (tap_l == 1 ? x0[1][2] : x0[2][2]);                  \
            const int gv_l = tap_l == 0 ? gvoff0 : (tap_l == 1 ? gvoff1 : gvoff2);             \
            gq[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG0, gv_l, g_c0 * 4, 0));      \
            gq[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsG0, gv_l + 16, g_c0 * 4, 0)); \
            const float* w_l = wrow + g_c0;                                                    \
            _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                 \
                wq[2 * r_] = *(const float4*)(w_l + r_ * C0); wq[2 * r_ + 1] = *(const float4*)(w_l + r_ * C0 + 4); \
            }                                                                                  \
            g_c0 += BK; if (g_c0 == C0) { g_c0 = 0; ++g_tsel; }                                \
        }
#define NAFP_X6_GEN_ONE(e_) \
        __fmul_rn(elu1(fmaf(gxc, ((const float*)&wq[4])[e_], fmaf(gxb, ((const float*)&wq[2])[e_], fmaf(gxa, ((const float*)&wq[0])[e_], ((const float*)&wq[6])[e_])))), ((const float*)&gq[0])[e_])
#define NAFP_X6_GEN(slot_)                                                                     \
        {                                                                                      \
            bf16x8 h_l, m_l, l_l;                                                              \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                    \
                const float v_l = NAFP_X6_GEN_ONE(e);                                          \
                const __bf16 hh_l = (__bf16)v_l;                                               \
                const float r1_l = v_l - (float)hh_l;                                          \
                const __bf16 mm_l = (__bf16)r1_l;                                              \
                h_l[e] = hh_l; m_l[e] = mm_l; l_l[e] = (__bf16)(r1_l - (float)mm_l);           \
            }                                                                                  \
            float* dst_l = smem + (slot_) * STAGE + a_wr;                                      \
            *(bf16x8*)dst_l = h_l; *(bf16x8*)(dst_l + BM * 8) = m_l; *(bf16x8*)(dst_l + 2 * BM * 8) = l_l; \
        }
        NAFP_X6_GEN_REQ()
        NAFP_X6_GEN(0)                                                             // step 0 (its weight pieces were requested by the pipeline fill above)
#define NAFP_X6_MFMAS()                                                                        \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                       \
            _Pragma("unroll") for (int ni = 0; ni < NIW; ++ni) {                               \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bm[ni], acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bh[ni], acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bm[ni], acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0); \
            }
#define NAFP_X6_READ_FRAGS(St_)                                                                \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                     \
            ah[mi] = *(const bf16x8*)((St_) + a_rd + mi * 32 * 8);                             \
            am[mi] = *(const bf16x8*)((St_) + BM * 8 + a_rd + mi * 32 * 8);                    \
            al[mi] = *(const bf16x8*)((St_) + 2 * BM * 8 + a_rd + mi * 32 * 8);                \
        }                                                                                      \
        _Pragma("unroll") for (int ni = 0; ni < NIW; ++ni) {                                   \
            bh[ni] = *(const bf16x8*)((St_) + boff + ni * 32 * BK + ((hh ^ rswz) * 4));        \
            bm[ni] = *(const bf16x8*)((St_) + boff + ni * 32 * BK + (((2 + hh) ^ rswz) * 4));  \
            bl[ni] = *(const bf16x8*)((St_) + TILE + TILEB + (wn * (BNT / 2) + ni * 32 + rl) * 8 + hh * 4); \
        }
        int s = 0;
        for (; s + 1 < n_steps; ++s) {                                             // every step but the last: generate the next one
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               // my weight pieces and my A rows of step s are in LDS
            __builtin_amdgcn_s_barrier();
            if (s == 0) { NAFP_TL(3) }
            const int nslot = slot ^ 1;
            const float* St = smem + slot * STAGE;
            bf16x8 ah[2], am[2], al[2], bh[NIW], bm[NIW], bl[NIW];
            NAFP_X6_READ_FRAGS(St)
            NAFP_DMA_STEP(s + 1, nslot)
            NAFP_X6_GEN_REQ()
            NAFP_X6_MFMAS()
            NAFP_X6_GEN(nslot)
            slot = nslot;
        }
        {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float* St = smem + slot * STAGE;
            bf16x8 ah[2], am[2], al[2], bh[NIW], bm[NIW], bl[NIW];
            NAFP_X6_READ_FRAGS(St)
            NAFP_X6_MFMAS()
        }
#undef NAFP_X6_GEN_REQ
#undef NAFP_X6_GEN_ONE
#undef NAFP_X6_GEN
#undef NAFP_X6_MFMAS
#undef NAFP_X6_READ_FRAGS
    } else
    for (int s = s_begin; s < n_steps; ++s) {
        // my DMA of step s has landed; after the barrier everybody's has, and everybody has
        // finished reading slot (s-1) % NSTAGE, which the next DMA overwrites.
        // (FUSE0: the A rows are ds_writes of this wave -> also drain lgkmcnt; the compiler's own
        // wait for the gamma0 loads has already retired every older DMA.)
        if (FUSE0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (NSTAGE == 2 || s + NSTAGE - 2 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NSTAGE - 2) * (NI + NIB)) : "memory");
        __builtin_amdgcn_s_barrier();
        const bool has_next = s + NSTAGE - 1 < n_steps && !NAFP_ABL(p, 1);
        int nslot = slot + NSTAGE - 1; if (nslot >= NSTAGE) nslot -= NSTAGE;
        const float* St = smem + slot * STAGE;
        if NAFP_ABL(p, 4) {
            if (has_next) { NAFP_DMA_STEP(s + NSTAGE - 1, nslot) }
            if (++slot == NSTAGE) slot = 0;
            continue;
        }
        // the first operand fragments are requested BEFORE the next DMA is issued: the DMA issue
        // (descriptor moves, m0 writes) then runs under the LDS read latency instead of ahead of it
        float4 a0[2], b0[NIW];
        {
            const int pc4 = ((0 + hh) ^ rswz) * 4;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a0[mi] = *(const float4*)(St + aoff + mi * 32 * BK + pc4);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) b0[ni] = *(const float4*)(St + boff + ni * 32 * BK + pc4);
        }
        if (has_next) {
            NAFP_DMA_STEP(s + NSTAGE - 1, nslot)
            if (FUSE0) NAFP_GEN_LOAD(s + NSTAGE - 1)
        }
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int pc4 = ((2 * kk + hh) ^ rswz) * 4;
            float4 a[2], b[NIW];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = kk == 0 ? a0[mi] : *(const float4*)(St + aoff + mi * 32 * BK + pc4);
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) b[ni] = kk == 0 ? b0[ni] : *(const float4*)(St + boff + ni * 32 * BK + pc4);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        if (FUSE0 && has_next) NAFP_GEN_STORE(s + NSTAGE - 1, nslot)
        if (++slot == NSTAGE) slot = 0;
    }
    NAFP_TL(4)
    if (p.opt & 2) __builtin_amdgcn_s_setprio(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // all waves are done with the operand tiles: LDS is reused below (a bare barrier: the
                                    // fence of __syncthreads() would wait for the operand loads that are in flight)
    NAFP_TL(5)

    if NAFP_ABL(p, 2) {          // ablation: no epilogue (keep the accumulators alive)
        float t = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[mi][ni][r];
        if (t == 12345.678f) p.y[tid] = t;
        return false;
    }
    // ---- epilogue ----
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).
    // Row groups of 4 (r & 3) = 4 consecutive samples at one position.
    const int ystep_b = p.P * p.Cout * 4;                                  // bytes to the same position of the next sample
    if (EPI == 3) {
        // PLAIN: y = acc (+ bias).  Stores go through a buffer descriptor re-based on the tile's first sample (of this
        // split-K slab): rows beyond the batch and position slots beyond P are out of range and dropped by the hardware,
        // so the 64 stores of a lane are straight-line code with nothing to wait for in between.
        float bv[NIW];
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) bv[ni] = p.bias ? p.bias[n_base + ni * 32] : 0.f;
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) asm volatile("" : "+v"(bv[ni]));   // the bias has landed HERE (else hipcc waits -- vmcnt(0), stores included -- in front of every use)
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
            p.y + ((int64_t)zsp * p.B + b0) * p.P * p.Cout, 0, (int)((unsigned)nb * (unsigned)ystep_b), 0x00020000);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int grp = wm * 16 + mi * 8 + 2 * rg + (lane >> 5);      // = tile row >> 2: 4 samples of one position
                const int pos = sPos[grp >> (p.log2ST - 2)];
                const int sl0 = (grp & (g4 - 1)) << 2;
                const int voff = pos < p.P ? ((sl0 * p.P + pos) * p.Cout + n_base) * 4 : (int)0x80000000;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ni = 0; ni < NIW; ++ni)
                        __builtin_amdgcn_raw_buffer_store_b32(f2i(acc[mi][ni][rg * 4 + q] + bv[ni]), rsP, voff, q * ystep_b + ni * 128, 0);
            }
        NAFP_TL(6)
        NAFP_TL(7)
        return false;
    }

    if (EPI == 4 || EPI == 5) {
        // Split-K finished in-kernel (EPI 5: of a PLAIN launch -- the transposed convs of the backward pass: the total, plus
        // bias, is stored as it is; no separate plain_finish_kernel and no second trip of the partial sums through a launch).  Every part writes its partial sums to its slab THROUGH the caches (sc0 sc1: the
        // XCDs' L2s are not coherent with each other), waits for the write acknowledgements, and draws an arrival ticket
        // for its output tile (device-scope atomic).  The workgroup that draws the last ticket re-reads all parts --
        // its own included, from memory, in part order: the sum does not depend on who arrives last -- and runs the FULL
        // epilogue below on the total; the others leave.  One launch and one round trip of the output less per conv than
        // slab + finish kernel.
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
            p.y + ((int64_t)zsp * p.B + b0) * p.P * p.Cout, 0, (int)((unsigned)nb * (unsigned)ystep_b), 0x00020000);
        int voffs[2][4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int grp = wm * 16 + mi * 8 + 2 * rg + (lane >> 5);
                const int pos = sPos[grp >> (p.log2ST - 2)];
                const int sl0 = (grp & (g4 - 1)) << 2;
                voffs[mi][rg] = pos < p.P ? ((sl0 * p.P + pos) * p.Cout + n_base) * 4 : (int)0x80000000;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ni = 0; ni < NIW; ++ni)
                        __builtin_amdgcn_raw_buffer_store_b32(f2i(acc[mi][ni][rg * 4 + q]), rsP, voffs[mi][rg], q * ystep_b + ni * 128, 17);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // my part is in memory
        __builtin_amdgcn_s_barrier();                                 // ... and so is every wave's of this workgroup
        unsigned* sTicket = (unsigned*)smem;
        const unsigned tile_id = (unsigned)((pb * p.n_sg + sg) * (p.Cout / BNT) + colz);
        if (tid == 0) sTicket[0] = __hip_atomic_fetch_add(p.tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = sTicket[0];
        if (ticket != (unsigned)(p.n_split - 1)) return false;
        __syncthreads();                                              // sTicket is read: LDS is reused by the statistics below
        if (tid == 0)                                                 // ready for the next launch; agent scope like every other access
            __hip_atomic_store(p.tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        for (int z = 0; z < p.n_split; ++z) {
            const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
                p.y + ((int64_t)z * p.B + b0) * p.P * p.Cout, 0, (int)((unsigned)nb * (unsigned)ystep_b), 0x00020000);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ni = 0; ni < NIW; ++ni)
                            acc[mi][ni][rg * 4 + q] += __builtin_bit_cast(
                                float, __builtin_amdgcn_raw_buffer_load_b32(rsZ, voffs[mi][rg], q * ystep_b + ni * 128, 17));
        }
        if (EPI == 5) {
            float bv[NIW];
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) bv[ni] = p.bias ? p.bias[n_base + ni * 32] : 0.f;
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) asm volatile("" : "+v"(bv[ni]));
            const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(
                p.y_final + (int64_t)b0 * p.P * p.Cout, 0, (int)((unsigned)nb * (unsigned)ystep_b), 0x00020000);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ni = 0; ni < NIW; ++ni)
                            __builtin_amdgcn_raw_buffer_store_b32(f2i(acc[mi][ni][rg * 4 + q] + bv[ni]), rsF, voffs[mi][rg], q * ystep_b + ni * 128, 0);
            return false;
        }
    }

    // FULL: v = ELU(r_b*acc + c_b*G + Hb); stats of v; store z = gamma_out * v
    f32x2 s2[2] = {{0.f, 0.f}, {0.f, 0.f}}, q2[2] = {{0.f, 0.f}, {0.f, 0.f}};   // per-lane sums of v and v^2 of sample slots (0, 1) and (2, 3)
    float* rowS = smem;                 // [BM] (LDS tiles are free again: last loop barrier passed)
    float* rowQ = smem + BM;
    const bool fast_stats = (EPI != 2 && EPI != 4) || p.ST == 4 || p.ST == 8;
    if (!fast_stats) {
        if (tid < BM) { rowS[tid] = 0.f; rowQ[tid] = 0.f; }           // NT >= BM
        __syncthreads();
    }
    // All position-indexed operands first (48 independent loads in flight together), then the arithmetic: issuing them
    // group by group exposed one L2 round trip per group.  (BM = 256 runs at 128 VGPRs: there the operands are fetched
    // per 32-row block, 24 at a time.)  Loads and stores go through buffer descriptors -- position slots beyond P and
    // rows beyond the batch are out of range (loads return 0, stores are dropped by the hardware), the lane part of an
    // address is one 32-bit register per 4-row group, the sample step a scalar -- and the loop body is instantiated per
    // (keep the pre-activation, statistics path) so that it is straight-line code: any runtime branch in it makes hipcc
    // put an s_waitcnt vmcnt(0) -- outstanding stores included -- in front of every block.
    constexpr bool PREF = (EPI == 0 || EPI == 1) && !FUSE0 && BK == 16 && BM == 128 && PREC == 0;   // block 0 was requested inside the K-loop (8-wave tiles run at 128 VGPRs: no room)
    if (MIL == 2) { if (!PREF) { NAFP_EPI_LOAD(0, 0) } NAFP_EPI_LOAD(1, MIL - 1) }
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        (EPI == 4 ? p.y_final : p.y) + (int64_t)b0 * p.P * p.Cout, 0, (int)((unsigned)nb * (unsigned)ystep_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        (p.v_out ? p.v_out : (EPI == 4 ? p.y_final : p.y)) + (int64_t)b0 * p.P * p.Cout, 0, (int)((unsigned)nb * (unsigned)ystep_b), 0x00020000);
    // The arithmetic runs on PAIRS of samples with the packed f32 instructions (v_pk_fma / v_pk_mul / v_pk_add: two
    // elements per issue slot): the f32 MFMAs execute at the f32 vector rate, and the timeline of this kernel shows the
    // SIMDs' issue time split between MFMAs and everything else -- every VALU instruction of the epilogue is paid for in
    // MFMA time, so the epilogue is written for instruction count.  ELU(t) = max(t, exp(min(t, 0)) - 1), the min folded
    // into v_exp's clamp modifier (exp > 1 -> 1).  No select on row validity: rows beyond the batch or beyond P have
    // zero accumulators; a slot beyond P also reads G = Hb = 0 (buffer range) and contributes ELU(0) = 0 to the sums, a
    // sample beyond the batch only touches its own sums, which are dropped; their stores are out of range.
#define NAFP_EPI_MAIN(KEEP_, FAST_)                                                            \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                         \
        const int ms = MIL == 2 ? mi : 0;                                                      \
        if (MIL == 1 && !(PREF && mi == 0)) { NAFP_EPI_LOAD(mi, 0) }                           \
        _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                     \
            const int grp = wm * 16 + mi * 8 + 2 * rg + (lane >> 5);      /* = tile row >> 2 */ \
            const int pos = sPos[grp >> (p.log2ST - 2)];                                       \
            const int sl0 = (grp & (g4 - 1)) << 2;                         /* first of the 4 samples */ \
            const int voff = pos < p.P ? ((sl0 * p.P + pos) * p.Cout + n_base) * 4 : (int)0x80000000; \
            _Pragma("unroll") for (int qp = 0; qp < 2; ++qp) {             /* samples sl0 + 2 qp, + 1 */ \
                const int r0 = rg * 4 + 2 * qp;                                                \
                const f32x2 rb2 = *(const f32x2*)(sRB + sl0 + 2 * qp), cb2 = *(const f32x2*)(sCB + sl0 + 2 * qp); \
                f32x2 rs2 = {0.f, 0.f}, rq2 = {0.f, 0.f};                                      \
                _Pragma("unroll") for (int ni = 0; ni < NIW; ++ni) {                           \
                    const f32x2 a2 = {acc[mi][ni][r0], acc[mi][ni][r0 + 1]};                   \
                    const f32x2 t2 = __builtin_elementwise_fma(rb2, a2, __builtin_elementwise_fma(cb2, (f32x2)(Gv[ms][rg][ni]), (f32x2)(Hv[ms][rg][ni]))); \
                    const f32x2 v2 = NAFP_ABL(p, 128) ? t2 : elu2(t2);    /* ablation 128: no exp */ \
                    const f32x2 z2 = v2 * gv[ms][rg][ni];                                      \
                    if (!NAFP_ABL(p, 64)) {                                /* ablation 64: no stores */ \
                        __builtin_amdgcn_raw_buffer_store_b32(f2i(z2.x), rsY, voff, (2 * qp) * ystep_b + ni * 128, NAFP_Z_AUX); \
                        __builtin_amdgcn_raw_buffer_store_b32(f2i(z2.y), rsY, voff, (2 * qp + 1) * ystep_b + ni * 128, NAFP_Z_AUX); \
                    }                                                                          \
                    if (KEEP_) {                                           /* training keeps the pre-activation */ \
                        __builtin_amdgcn_raw_buffer_store_b32(f2i(t2.x), rsV, voff, (2 * qp) * ystep_b + ni * 128, NAFP_T_AUX); \
                        __builtin_amdgcn_raw_buffer_store_b32(f2i(t2.y), rsV, voff, (2 * qp + 1) * ystep_b + ni * 128, NAFP_T_AUX); \
                    }                                                                          \
                    rs2 += v2; rq2 = __builtin_elementwise_fma(v2, v2, rq2);                   \
                }                                                                              \
                if (FAST_) { s2[qp] += rs2; q2[qp] += rq2; }                                   \
                else {                                                                         \
                    _Pragma("unroll") for (int h2 = 0; h2 < 2; ++h2) {                         \
                        float rs = rs2[h2], rq = rq2[h2];                                      \
                        _Pragma("unroll") for (int o = 16; o > 0; o >>= 1) { rs += __shfl_xor(rs, o, 64); rq += __shfl_xor(rq, o, 64); } \
                        if (ncol == 0) { atomicAdd(rowS + (grp << 2) + 2 * qp + h2, rs); atomicAdd(rowQ + (grp << 2) + 2 * qp + h2, rq); } \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
    }
    if (EPI == 0) { NAFP_EPI_MAIN(false, true) }
    else if (EPI == 1) { NAFP_EPI_MAIN(true, true) }
    else if (fast_stats) {
        if (p.v_out) { NAFP_EPI_MAIN(true, true) } else { NAFP_EPI_MAIN(false, true) }
    } else {
        if (p.v_out) { NAFP_EPI_MAIN(true, false) } else { NAFP_EPI_MAIN(false, false) }
    }
#undef NAFP_EPI_MAIN
    NAFP_TL(6)
    if NAFP_ABL(p, 16) {         // ablation: no statistics reduction (keep the sums alive)
        if (s2[0].x + s2[0].y + s2[1].x + s2[1].y + q2[0].x + q2[0].y + q2[1].x + q2[1].y == 12345.678f) p.y[tid] = 1.f;
        return false;
    }
    if (fast_stats) {
        // ST == 4: the tile's 4 samples are the 4 register slots (r & 3) of every lane.
        // ST == 8: lanes 0..31 hold samples 0..3, lanes 32..63 samples 4..7 (grp & 1 == lane >> 5 for every block).
        // Each 32-lane half is summed with DPP row operations (VALU only); lanes 31 and 63 hand the half sums to LDS
        // (no barrier in front: nothing else lives in the first bytes of the operand ring any more), one thread per
        // (statistic, sample) adds the waves' parts in a fixed order and issues the global atomic.
        double* red = (double*)smem;        // [NW waves][half][sum | sumsq][4]
        float hs[4], hq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { hs[q] = half_wave_sum_dpp(s2[q >> 1][q & 1]); hq[q] = half_wave_sum_dpp(q2[q >> 1][q & 1]); }
        if ((lane & 31) == 31) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                red[wave * 16 + (lane >> 5) * 8 + q] = (double)hs[q];
                red[wave * 16 + (lane >> 5) * 8 + 4 + q] = (double)hq[q];
            }
        }
        __syncthreads();
        if (tid < 2 * p.ST) {
            const int which = tid / p.ST, sl = tid - which * p.ST;
            const int b = sg * p.ST + sl;
            if (b < p.B) {
                double t = 0.0;
                if (p.ST == 8) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) t += red[w * 16 + (sl >> 2) * 8 + which * 4 + (sl & 3)];
                } else {
#pragma unroll
                    for (int w = 0; w < NW; ++w) t += red[w * 16 + which * 4 + sl] + red[w * 16 + 8 + which * 4 + sl];
                }
                stat_add(p.stats_out + 2 * (int64_t)b + which, t, p.inv_n_in < 0.0);
            }
        }
    } else {
        __syncthreads();
        if (tid < p.ST) {
            const int b = sg * p.ST + tid;
            if (b < p.B) {
                double ds = 0.0, dq = 0.0;
                for (int pl = 0; pl < p.PT; ++pl) { ds += (double)rowS[pl * p.ST + tid]; dq += (double)rowQ[pl * p.ST + tid]; }
                stat_add(p.stats_out + 2 * (int64_t)b, ds, p.inv_n_in < 0.0);
                stat_add(p.stats_out + 2 * (int64_t)b + 1, dq, p.inv_n_in < 0.0);
            }
        }
    }
    NAFP_TL(7)
#undef NAFP_TL
    return true;
}

// One __global__ per tile shape and epilogue (launch bounds are not template-dependent).
#define NAFP_GEMM_KERNEL(name_, BM_, BN_, BK_, NSTAGE_, MINW_, FUSE0_, EPI_)                 \
    __global__ __launch_bounds__(2 * BM_, MINW_) void name_(const ConvKernelParams p) {       \
        conv_gemm_body<BM_, BN_, BK_, NSTAGE_, FUSE0_, EPI_>(p);                              \
    }
#define NAFP_GEMM_KERNEL_BF16X3(name_, BM_, BN_, NSTAGE_, MINW_)                                \
    __global__ __launch_bounds__(2 * BM_, MINW_) void name_(const ConvKernelParams p) {       \
        conv_gemm_body<BM_, BN_, 16, NSTAGE_, false, 0, 1>(p);                                \
    }
#define NAFP_GEMM_KERNEL_BF16X6(name_, BM_, BN_, NSTAGE_, MINW_, EPI_)                          \
    __global__ __launch_bounds__(2 * BM_, MINW_) void name_(const ConvKernelParams p) {       \
        conv_gemm_body<BM_, BN_, 16, NSTAGE_, false, EPI_, 2>(p);                             \
    }
#define NAFP_GEMM_KERNELS(name_, BM_, BN_, MINW_)                                             \
    NAFP_GEMM_KERNEL(name_##_infer, BM_, BN_, 16, 3, MINW_, false, 0)                         \
    NAFP_GEMM_KERNEL(name_##_train, BM_, BN_, 16, 3, MINW_, false, 1)                         \
    NAFP_GEMM_KERNEL(name_##_any, BM_, BN_, 16, 3, MINW_, false, 2)                           \
    NAFP_GEMM_KERNEL(name_##_plain, BM_, BN_, 16, 3, MINW_, false, 3)                         \
    static void (*const name_##_tab[4])(const ConvKernelParams) = {name_##_infer, name_##_train, name_##_any, name_##_plain};
// BK = 16, 3 stages, 3 workgroups/CU.  The other staging points were built and measured on the MI355X
// (segments/s at BSZ 640, same run): k16s3 148.2 k | k32s2 (2 WG/CU) 143.6 k | k16s2 (4 WG/CU) 142.7 k |
// k16s4 (2 WG/CU) 140.9 k | k32s3 (1 WG/CU) 116.4 k; they are not compiled any more.
NAFP_GEMM_KERNELS(conv_gemm_k16s3, 128, 128, 3)
NAFP_GEMM_KERNEL(conv_gemm_k16s3_fuse0, 128, 128, 16, 3, 3, true, 0)   // conv1 with conv0 generated in-kernel (inference)
// 128 x 64 tile (each wave 64 x 32), 36 KB ring -> 4 workgroups per CU: twice the workgroups of half the size for the
// launches whose 128 x 128 tiling leaves the CUs unevenly loaded or forces a split along K (the mid and late convs)
NAFP_GEMM_KERNEL(conv_gemm_n64k16s3_infer, 128, 64, 16, 3, 4, false, 0)
NAFP_GEMM_KERNEL(conv_gemm_n64k16s3_train, 128, 64, 16, 3, 4, false, 1)
NAFP_GEMM_KERNEL(conv_gemm_n64k16s3_any, 128, 64, 16, 3, 4, false, 2)
NAFP_GEMM_KERNEL(conv_gemm_n64k16s3_plain, 128, 64, 16, 3, 4, false, 3)
NAFP_GEMM_KERNEL(conv_gemm_n64k16s3_splitfin, 128, 64, 16, 3, 4, false, 4)      // split-K part + in-kernel finish by the last arriver
NAFP_GEMM_KERNEL(conv_gemm_n64k16s3_plainfin, 128, 64, 16, 3, 4, false, 5)      // ... of a PLAIN launch (transposed conv)
NAFP_GEMM_KERNEL(conv_gemm_k16s3_plainfin, 128, 128, 16, 3, 3, false, 5)
// 128 x 64 tile on a 2-stage ring (24 KB): 5 workgroups per CU = 1280 slots -- the mid convs' 640 / 1280 tiles (x split-K) then
// fill exactly one round instead of leaving a last round with one workgroup per CU (which runs at half the pipe rate)
NAFP_GEMM_KERNEL(conv_gemm_n64k16s2_infer, 128, 64, 16, 2, 5, false, 0)
NAFP_GEMM_KERNEL(conv_gemm_n64k16s2_train, 128, 64, 16, 2, 5, false, 1)
NAFP_GEMM_KERNEL(conv_gemm_n64k16s2_splitfin, 128, 64, 16, 2, 5, false, 4)
static void (*const conv_gemm_n64k16s2_tab[5])(const ConvKernelParams) = {conv_gemm_n64k16s2_infer, conv_gemm_n64k16s2_train, nullptr, nullptr,
                                                                           conv_gemm_n64k16s2_splitfin};
static void (*const conv_gemm_n64k16s3_tab[5])(const ConvKernelParams) = {conv_gemm_n64k16s3_infer, conv_gemm_n64k16s3_train, conv_gemm_n64k16s3_any,
                                                                           conv_gemm_n64k16s3_plain, conv_gemm_n64k16s3_splitfin};
// 256 x 128 tile, 8 waves (4 x 2), 72 KB ring -> 2 workgroups = 16 waves per CU (4 per SIMD): the weight tile is staged
// once per 256 rows instead of once per 128, and a workgroup's fixed costs (geometry, pipeline fill) cover twice the output
NAFP_GEMM_KERNEL(conv_gemm_m256k16s3_infer, 256, 128, 16, 3, 4, false, 0)
NAFP_GEMM_KERNEL(conv_gemm_m256k16s3_train, 256, 128, 16, 3, 4, false, 1)
NAFP_GEMM_KERNEL(conv_gemm_m256k16s3_plain, 256, 128, 16, 3, 4, false, 3)
// (no generic-statistics instantiation: the launcher takes the 128-row tile when 256 rows are not 4 or 8 samples per position)
static void (*const conv_gemm_m256k16s3_tab[4])(const ConvKernelParams) = {conv_gemm_m256k16s3_infer, conv_gemm_m256k16s3_train, nullptr,
                                                                            conv_gemm_m256k16s3_plain};

// experimental split-bf16 products (inference epilogue only), one per tile shape
NAFP_GEMM_KERNEL_BF16X3(conv_gemm_k16s3_infer_bf16x3, 128, 128, 3, 3)
NAFP_GEMM_KERNEL_BF16X3(conv_gemm_m256k16s3_infer_bf16x3, 256, 128, 3, 4)
NAFP_GEMM_KERNEL_BF16X3(conv_gemm_n64k16s2_infer_bf16x3, 128, 64, 2, 5)
// ... and the exact 3-way split with six products (PREC = 2); the 256-row tile at 2 waves per SIMD (the third plane does not fit 128 VGPRs)
// (ring depth and occupancy bound of the two large shapes are build-time knobs: tools/build_variant.sh ... -DNAFP_X6_...)
#ifndef NAFP_X6_K16_NSTAGE
#define NAFP_X6_K16_NSTAGE 2                // 2 stages at 138 VGPRs = three workgroups per CU: conv1 0.795 ms against 0.839 with 3 stages (two per CU)
#endif
#ifndef NAFP_X6_K16_MINW
#define NAFP_X6_K16_MINW 3
#endif
#ifndef NAFP_X6_M256_NSTAGE
#define NAFP_X6_M256_NSTAGE 3
#endif
#ifndef NAFP_X6_M256_MINW
#define NAFP_X6_M256_MINW 2
#endif
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_k16s3_infer_bf16x6, 128, 128, NAFP_X6_K16_NSTAGE, NAFP_X6_K16_MINW, 0)
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_m256k16s3_infer_bf16x6, 256, 128, NAFP_X6_M256_NSTAGE, NAFP_X6_M256_MINW, 0)
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_n64k16s2_infer_bf16x6, 128, 64, 2, 4, 0)
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_k16s3_plain_bf16x6, 128, 128, NAFP_X6_K16_NSTAGE, NAFP_X6_K16_MINW, 3)      // the split-K parts of the late convs
// ... with the TRAINING epilogue (forward_train under NAFP_OPT_BF16X3 = 2: the pre-activation is kept for the backward pass)
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_k16s3_train_bf16x6, 128, 128, NAFP_X6_K16_NSTAGE, NAFP_X6_K16_MINW, 1)
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_n64k16s2_train_bf16x6, 128, 64, 2, 4, 1)
// ... with the generic-statistics epilogue (samples per position other than 4 / 8: the small late layers of a large training batch)
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_k16s3_any_bf16x6, 128, 128, NAFP_X6_K16_NSTAGE, NAFP_X6_K16_MINW, 2)
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_n64k16s2_any_bf16x6, 128, 64, 2, 4, 2)
// ... and the PLAIN epilogue on 64-column tiles (the transposed convs of the train step; their 128-column form is conv_gemm_k16s3_plain_bf16x6)
NAFP_GEMM_KERNEL_BF16X6(conv_gemm_n64k16s2_plain_bf16x6, 128, 64, 2, 4, 3)
// conv1 with conv0 generated in-kernel on the exact split: no A stream at all (see the K-loop)
__global__ __launch_bounds__(256, 3) void conv_gemm_k16s2_fuse0_bf16x6(const ConvKernelParams p) {
    conv_gemm_body<128, 128, 16, 2, true, 0, 2>(p);
}

// Optional timing events of the launch in flight (ConvGemmArgs::ev_start / ev_stop): they ride on a kernel's own dispatch
// packet (hipExtLaunchKernel: time stamps of its completion signal), so -- unlike hipEventRecord between two kernels -- they
// put nothing into the queue and cost the GPU no idle time.
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

// The exact 3-way bf16 split of a packed weight tensor (Cout, K), K = 3 Cin a multiple of 16, for the PREC = 2 kernels: x = h + m + l
// with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (each difference is exact in float32; three 8-bit significands hold the 24 bits).
// hm: the f32 tensor's shape, every group of 16 k replaced by [h(16) | m(16)] -- the B staging of the f32 kernels moves it unchanged;
// l: plain (Cout, K) bf16.
// All tensors of a parameter set in ONE launch (blockIdx.y = tensor), 8 consecutive k per thread: two 16-B loads, three 16-B stores.
// (Round 6, the train step under the option: 30 tensors per step -- 15 packed kernels and their 15 flipped forms -- were 30 launches of
// 24 us each with 2-byte stores.)
__global__ __launch_bounds__(256) void split_weights_multi_kernel(const SplitTable t) {
    const float* __restrict__ wp = t.wp[blockIdx.y];
    unsigned short* __restrict__ hm = (unsigned short*)t.hm[blockIdx.y];
    unsigned short* __restrict__ wl = (unsigned short*)t.wl[blockIdx.y];
    const int64_t n8 = t.n8[blockIdx.y];
    const int K = t.K[blockIdx.y];
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    for (int64_t i8 = blockIdx.x * 256ll + threadIdx.x; i8 < n8; i8 += (int64_t)gridDim.x * 256) {
        const int64_t i = i8 * 8;
        const float4 a = *(const float4*)(wp + i), b = *(const float4*)(wp + i + 4);
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        u16x8 vh, vm, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 h = (__bf16)x[e];
            const float r1 = x[e] - (float)h;
            const __bf16 m = (__bf16)r1;
            vh[e] = __builtin_bit_cast(unsigned short, h);
            vm[e] = __builtin_bit_cast(unsigned short, m);
            vl[e] = __builtin_bit_cast(unsigned short, (__bf16)(r1 - (float)m));
        }
        const int64_t row = i / K;
        const int k = (int)(i - row * K);
        const int64_t base = row * 2 * K + (int64_t)(k >> 4) * 32 + (k & 8);
        *(u16x8*)(hm + base) = vh;
        *(u16x8*)(hm + base + 16) = vm;
        *(u16x8*)(wl + i) = vl;
    }
}

int launch_split_weights_multi(const SplitTable& t, hipStream_t st) {
    if (t.count <= 0) return NAFP_OK;
    int64_t n8_max = 0;
    for (int j = 0; j < t.count; ++j) {
        if (t.K[j] % 16 != 0) return NAFP_ERR_UNSUPPORTED;
        n8_max = std::max(n8_max, t.n8[j]);
    }
    split_weights_multi_kernel<<<dim3((unsigned)std::min<int64_t>((n8_max + 255) / 256, 1024), (unsigned)t.count), 256, 0, st>>>(t);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

template <typename KernelT>
static int launch_variant(KernelT kernel, int BM, int BNt, int BK, int NSTAGE, const ConvKernelParams& p, dim3 grid, hipStream_t st,
                          int extra_stage_floats = 0) {
    static const int lds_pad = []() { const char* e = getenv("NAFP_LDS_PAD"); return e ? atoi(e) : 0; }();      // diagnostic: extra dynamic LDS per workgroup (changes co-residency)
    const int lds = (NSTAGE * ((BM + BNt) * BK + extra_stage_floats) + 2 * BM + 96 + (p.f0_feat ? 4 * p.Cin : 0)) * (int)sizeof(float) + lds_pad;
    NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (g_ev_start || g_ev_stop) {
        ConvKernelParams pc = p;
        void* args[] = {(void*)&pc};
        NAFP_HIP_CHECK(hipExtLaunchKernel((const void*)kernel, grid, dim3(2 * BM), args, (size_t)lds, st, g_ev_start, g_ev_stop, 0));
        g_ev_start = nullptr; g_ev_stop = nullptr;
    } else {
        kernel<<<grid, 2 * BM, lds, st>>>(p);
    }
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// Split-K finish: sums the S partial slabs in a fixed order (deterministic), then the same
// epilogue as the fused path: v = ELU(r_b*A + c_b*G + Hb), statistics of v, z = gamma_out . v.
// A workgroup owns 4096 consecutive output floats (4 ... 16 rows of Cout = 1024 ... 256): the per-sample scalars of the
// few samples it touches are formed once in LDS, a thread finishes 4 float4 with the S slab loads of each in flight
// together, and the statistics meet per sample in LDS before one pair of double atomics per (workgroup, sample).
// (Round 1 ran one workgroup per output row: 640 ... 5120 workgroups of a few hundred bytes each, latency-bound.)
constexpr int FIN_MAXS = 64;            // samples a workgroup may touch (4096 / Cout rows, Cout >= 64)
template <int FIN_F4>                   // float4 per workgroup: 1024, 512 or 256 (the launch wants >= 1024 workgroups)
__global__ __launch_bounds__(256) void splitk_finish_kernel(
        const float* __restrict__ slab, int S, const float* __restrict__ G, const float* __restrict__ Hb,
        const float* __restrict__ gamma_out, const stat_t* __restrict__ stats_in,
        stat_t* __restrict__ stats_out, float* __restrict__ y, float* __restrict__ v_out, int B, int P, int Cout,
        double inv_n_in) {
    __shared__ float sR[FIN_MAXS], sC[FIN_MAXS];
    __shared__ float sPart[4][FIN_F4 / 256][2][2];     // [wave][pass][first / last sample of the wave][sum, sumsq]
    __shared__ int sPartB[4][FIN_F4 / 256][2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int c4 = Cout / 4;
    const int64_t total = (int64_t)B * P * c4;
    const int64_t i0 = (int64_t)blockIdx.x * FIN_F4;
    const int64_t i_last = std::min<int64_t>(total, i0 + FIN_F4) - 1;
    const int b_first = (int)((i0 / c4) / P), b_last = (int)((i_last / c4) / P);
    if (tid <= b_last - b_first) {
        const int b = b_first + tid;
        stat_ln_scalars(stats_in + 2 * (int64_t)b, inv_n_in, &sR[tid], &sC[tid]);
        stat_forward_poison(stats_in + 2 * (int64_t)b, stats_out + 2 * (int64_t)b);
    }
    __syncthreads();
    const int64_t slab_stride4 = total;                 // float4 per slab
#pragma unroll 1
    for (int e = 0; e < FIN_F4 / 256; ++e) {
        const int64_t idx = i0 + e * 256 + tid;
        const bool live = idx < total;
        const int64_t ii = live ? idx : i0;
        const int64_t row = ii / c4;
        const int n = (int)(ii - row * c4) * 4;
        const int b = (int)(row / P), pos = (int)(row - (int64_t)b * P);
        const float4* src = (const float4*)slab + ii;
        float4 acc = src[0];
        for (int sp = 1; sp < S; ++sp) {
            const float4 t = src[sp * slab_stride4];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        const float4 g4 = *(const float4*)(G + (int64_t)pos * Cout + n);
        const float4 h4 = *(const float4*)(Hb + (int64_t)pos * Cout + n);
        const float4 go = *(const float4*)(gamma_out + (int64_t)pos * Cout + n);
        const float rb = sR[b - b_first], cb = sC[b - b_first];
        float4 tq, v;
        tq.x = fmaf(rb, acc.x, fmaf(cb, g4.x, h4.x));
        tq.y = fmaf(rb, acc.y, fmaf(cb, g4.y, h4.y));
        tq.z = fmaf(rb, acc.z, fmaf(cb, g4.z, h4.z));
        tq.w = fmaf(rb, acc.w, fmaf(cb, g4.w, h4.w));
        v.x = elu1(tq.x); v.y = elu1(tq.y); v.z = elu1(tq.z); v.w = elu1(tq.w);
        float s = live ? (v.x + v.y) + (v.z + v.w) : 0.f;
        float q = live ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
        if (live) {
            ((float4*)y)[idx] = make_float4(v.x * go.x, v.y * go.y, v.z * go.z, v.w * go.w);
            if (v_out) ((float4*)v_out)[idx] = tq;
        }
        // statistics, in a fixed order: a wave covers 256 consecutive floats (Cout >= 128: a wave spans at most two rows, so
        // at most two samples): one partial sum for the sample of its first lane, one for that of its last lane
        const int b_lo = __shfl(b, 0, 64), b_hi = __shfl(b, 63, 64);
        const float s_lo = wave_sum(b == b_lo ? s : 0.f), q_lo = wave_sum(b == b_lo ? q : 0.f);
        const float s_hi = wave_sum(b != b_lo ? s : 0.f), q_hi = wave_sum(b != b_lo ? q : 0.f);
        if (lane == 0) {
            const int w = tid >> 6;
            sPart[w][e][0][0] = s_lo; sPart[w][e][0][1] = q_lo; sPartB[w][e][0] = b_lo;
            sPart[w][e][1][0] = s_hi; sPart[w][e][1][1] = q_hi; sPartB[w][e][1] = b_hi != b_lo ? b_hi : -1;
        }
    }
    __syncthreads();
    if (tid <= b_last - b_first) {
        const int b = b_first + tid;
        double ds = 0.0, dq = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int e = 0; e < FIN_F4 / 256; ++e)
                for (int h = 0; h < 2; ++h)
                    if (sPartB[w][e][h] == b) { ds += (double)sPart[w][e][h][0]; dq += (double)sPart[w][e][h][1]; }
        stat_add(stats_out + 2 * (int64_t)b, ds, inv_n_in < 0.0);
        stat_add(stats_out + 2 * (int64_t)b + 1, dq, inv_n_in < 0.0);
    }
}

// Split-K policy: layers whose M x N tiling yields too few workgroups to fill 256 CUs
// (the late convs: M = 640...20480 rows at B = 640) split their K-steps across blockIdx.z.
static int live_k_steps(const ConvGeom& g) {
    // K-steps (BK = 16) of the taps that read real data for at least one output position
    const int n_in = g.axis == 0 ? g.Tin : g.Fin, n_out = g.axis == 0 ? g.Tout : g.Fout;
    int n_live = 0;
    for (int t = 0; t < 3; ++t) {
        bool live = false;
        for (int o = 0; o < n_out && !live; ++o) { const int i = o * g.stride - g.pad + t; live = i >= 0 && i < n_in; }
        n_live += live;
    }
    return n_live * g.Cin / 16;
}

// 64-column tiles of the FORWARD pass run on the 2-stage ring, 5 workgroups per CU (NAFP_N64S2=0: the 3-stage, 4-per-CU kernel)
static bool n64_two_stage() {
    static const bool on = []() { const char* e = getenv("NAFP_N64S2"); return !e || e[0] != '0'; }();
    return on;
}

static int choose_split(int64_t n_tiles, int k_steps, int64_t out_floats, double slots = 768.0) {
    // Score each split factor S by (how full the last round of workgroups is) x (share of a
    // workgroup's time spent in its K-loop rather than prologue/epilogue); 768 = 256 CUs x 3
    // resident workgroups.  Splitting costs a slab round trip, so it must win by a margin.
    static const int force = []() { const char* e = getenv("NAFP_SPLITK"); return e ? atoi(e) : -1; }();
    if (force == 0) return 1;
    if (force > 0) return force;
    if (n_tiles >= 3072) return 1;
    if (slots == 1280.0)       // forward, 64-column tiles, 5 per CU.  Measured at B = 640: 1280 tiles (convs 6, 8) fill the slots and are
        // best unsplit; 640 tiles in two parts: conv9 (96 K-steps) 0.187 -> 0.167 ms, conv7 (48 K-steps) 0.098 -> 0.096
        return (n_tiles >= 1000 || k_steps / 2 < 24 || 2 * out_floats * 4 > ((int64_t)64 << 20)) ? 1 : 2;
    if (slots == 768.0 && n_tiles <= 200) {
        // few 128 x 128 tiles (the late convs): measured at B = 640 (tools/plan_sweep.sh) a 3-way split is the best or within
        // 2 % of it for 80 and 160 tiles, 6-way for 40; keep at least 8 K-steps per part
        int S = n_tiles <= 40 ? 6 : 3;
        while (S > 1 && (k_steps / S < 8 || (int64_t)S * out_floats * 4 > ((int64_t)64 << 20))) --S;
        return S;
    }
    const double overhead_steps = 6.0;
    int best = 1; double best_score = 0.0;
    for (int S = 1; S <= 16; ++S) {
        const double k = (double)k_steps / S;
        if (S > 1 && (k < 8.0 || (int64_t)S * out_floats * 4 > ((int64_t)64 << 20))) break;
        const double w = (double)n_tiles * S;
        const double rounds = (double)((int64_t)((w + slots - 1) / slots));
        double score = (w / (rounds * slots)) * (k / (k + overhead_steps));
        if (S > 1) score *= 0.93;                      // slab write + read + finish launch
        if (score > best_score + 1e-9) { best_score = score; best = S; }
    }
    return best;
}

// Split-K finish of a PLAIN launch (transposed conv, G/Hb images): y = sum of the S slabs (+ bias).
__global__ __launch_bounds__(256) void plain_finish_kernel(const float* __restrict__ slab, int S, const float* __restrict__ bias,
                                                          float* __restrict__ y, int64_t n4, int Cout) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 acc = ((const float4*)slab)[i];
        for (int sp = 1; sp < S; ++sp) {
            const float4 t = ((const float4*)slab)[i + sp * n4];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        if (bias) {
            const float4 bv = *(const float4*)(bias + (4 * i) % Cout);
            acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
        }
        ((float4*)y)[i] = acc;
    }
}

// K-steps (BK = 16) of a tile of the transposed conv: with stride 2 the parity classes of tile_pos()
// carry 2 resp. 1 live taps (1.5 on average), with stride 1 all three.  (An estimate for the split-K policy only: the kernel skips
// dead taps per tile.  [r5] The exact per-class count -- a third of this for the layers whose forward conv has ONE live tap at the
// 1-s input -- was tried in its place: same-box A/B 11.69 vs 11.75 ms at B = 640, 22.29 vs 22.21 at 1280, 85.6 vs 85.2 at 5120:
// the split factor is not what limits those launches -- they share the chip with the side stream's weight gradients.)
static int dgrad_k_steps(const ConvGeom& g) { return (g.stride == 2 ? 3 : 6) * g.Cout / 32; }

static int tile_pt(int P) {
    int pt = 1;
    while (pt * 2 <= P && pt * 2 <= 32) pt *= 2;          // largest power of two <= min(P, 32)
    return pt;
}

// Position slots of a FORWARD tile.  Default: the largest power of two <= min(P, 32).  Where a sizeable share of the output
// positions has a tap that reads only zero padding -- TF SAME puts the odd padding element behind the data, so with stride 2
// that is the LAST output coordinate along the conv axis (tap 2), and for the stride-1 layer on two frames each of the two
// frames has its own dead tap -- the tile takes positions of ONE class (PT divides both class sizes; along T the classes
// are made contiguous by tile_pos(), perm 2) and the kernel's tile-wide dead-tap skip drops that tap's K-steps.
// Taken only when it saves >= 3 % of the launch's MACs and does not cost the launch its register-resident statistics
// (4 or 8 samples per position).  At the 1-s input and B = 640: conv8 -1/3, conv6 -1/6, conv13 -1/6, convs 4 / 11 -1/12,
// conv2 -1/24 of their K-steps.  NAFP_TAPCLASS=0 switches it off.
struct FwdTile { int pt, perm; double saved; };
static FwdTile fwd_tile(const ConvGeom& g, int BM) {
    const int P = g.Fout * g.Tout;
    FwdTile r{tile_pt(P), 0, 0.0};
    static const int mode = []() { const char* e = getenv("NAFP_TAPCLASS"); return e ? atoi(e) : 1; }();
    const int L = g.axis == 0 ? g.Tout : g.Fout, n_in = g.axis == 0 ? g.Tin : g.Fin;
    if (!mode || L < 2) return r;
    auto mask = [&](int o) { unsigned m = 0; for (int t = 0; t < 3; ++t) { const int i = o * g.stride - g.pad + t; if (i >= 0 && i < n_in) m |= 1u << t; } return m; };
    unsigned m0 = 0;
    for (int o = 0; o + 1 < L; ++o) m0 |= mask(o);
    const unsigned m1 = mask(L - 1);
    const int n_all = __builtin_popcount(m0 | m1), n0 = __builtin_popcount(m0), n1 = __builtin_popcount(m1);
    const int64_t c1 = P / L, c0 = (int64_t)c1 * (L - 1);                 // positions per class
    const double saved = n_all == 0 ? 0.0 : (double)(c0 * (n_all - n0) + c1 * (n_all - n1)) / ((double)P * n_all);
    if (saved < 0.03) return r;
    int pt = 1;
    while (pt * 2 <= 32 && c0 % (pt * 2) == 0 && c1 % (pt * 2) == 0) pt *= 2;
    const int st = BM / pt, st_def = BM / r.pt;
    const bool fast = st == 4 || st == 8, fast_def = st_def == 4 || st_def == 8;
    if (st < 4 || (fast_def && !fast)) return r;
    r.pt = pt; r.perm = g.axis == 0 ? 2 : 0; r.saved = saved;
    return r;
}

// Position slots of a DGRAD tile (rows = input positions of the conv).  With stride 2 the positions are enumerated parity
// class by parity class (tile_pos(), perm 1: one class receives taps {0, 2}, the other only tap {1}) so that a tile of ONE
// class skips the structurally zero taps; that needs PT to divide both class sizes.  The default (largest power of two
// <= min(P, 32)) straddles the classes whenever a class has fewer than 32 positions -- the transposed convs of layers 9,
// 10, 11, 13, 15 at the 1-s input then multiplied all three taps, twice the needed K-steps (BSZ 5120: dgrad_9 72
// TFLOP/s of useful work, dgrad_13 59).  Any samples-per-position count works for the PLAIN epilogue.
static int dgrad_tile_pt(const ConvGeom& g) {
    const int P = g.Fin * g.Tin;
    int pt = tile_pt(P);
    const int L = g.axis == 0 ? g.Tin : g.Fin;
    if (g.stride == 1 && g.axis == 0 && L == 2 && g.pad == 1) {        // two frames, each with its own dead tap (perm 2)
        while (pt > 1 && (P / 2) % pt != 0) pt >>= 1;
        return pt;
    }
    if (g.stride != 2 || L < 2) return pt;
    const int c0 = g.pad & 1, n0 = (L - c0 + 1) / 2;
    const int per = P / L;                                  // positions per coordinate value along the tap axis
    const int s0 = per * n0, s1 = per * (L - n0);
    while (pt > 1 && (s0 % pt != 0 || s1 % pt != 0)) pt >>= 1;
    return pt;
}

// Tile height of a launch: 256 rows (8 waves, 2 workgroups per CU) when the launch has enough 128-row tiles to fill the
// chip several times over with the larger tile as well -- those launches are never split along K; 128 rows otherwise.
// NAFP_BM256 = minimum number of 128-row tiles (0 = never; default 2000).  Measured at B = 640, same box, ms per launch
// for convs 1-5 with thresholds 0 / 2000 / 1000: 1.185 0.605 0.316 0.315 0.289 / 1.160 0.591 0.306 0.306 0.293 /
// 1.166 0.592 0.307 0.306 0.334 -- conv5 has 1280 tiles of 128 rows = 640 of 256: 1.25 rounds of 512 slots.
static int pick_bm(int64_t B, int P, int Cout) {
    static const int64_t thr = []() { const char* e = getenv("NAFP_BM256"); return e ? atoll(e) : (int64_t)2000; }();
    if (thr <= 0) return 128;
    const int pt = tile_pt(P), ST = 128 / pt;
    const int64_t n_tiles = ((B + ST - 1) / ST) * ((P + pt - 1) / pt) * (Cout / BN);
    return n_tiles >= thr ? 256 : 128;
}

// Tile width of a launch that runs 128-row tiles: 64 columns (4 workgroups per CU, each half the work) where the
// 128 x 128 tiling gives few, coarse workgroups.  NAFP_BN64: 0 never, 2 whenever Cout allows, 1 (default) by tile count.
// Diagnostic override for tile-plan sweeps: NAFP_FWD_PLAN="bn:S" forces the column width and split factor of every forward
// FULL launch on 128-row tiles (where legal); tools/plan_sweep.sh runs the combinations and prints per-conv times.
struct PlanOverride { int bn, S; };
static const PlanOverride& plan_override() {
    static const PlanOverride po = []() {
        PlanOverride r{0, 0};
        const char* e = getenv("NAFP_FWD_PLAN");
        if (e) sscanf(e, "%d:%d", &r.bn, &r.S);
        return r;
    }();
    return po;
}

// the same for the transposed convs (DGRAD launches on 128-row tiles): NAFP_DGRAD_PLAN="bn:S"
static const PlanOverride& dgrad_plan_override() {
    static const PlanOverride po = []() {
        PlanOverride r{0, 0};
        const char* e = getenv("NAFP_DGRAD_PLAN");
        if (e) sscanf(e, "%d:%d", &r.bn, &r.S);
        return r;
    }();
    return po;
}

static int pick_bn(int64_t n_tiles128, int Cout, int k_steps = 0) {
    static const int mode = []() { const char* e = getenv("NAFP_BN64"); return e ? atoi(e) : 1; }();
    static const int64_t thr = []() { const char* e = getenv("NAFP_BN64_TILES"); return e ? atoll(e) : (int64_t)1000; }();
    if (mode == 0 || Cout % 64 != 0) return 128;
    if (mode == 2) return 64;
    // plan sweep at B = 640 (tools/plan_sweep.sh): with <= 160 tiles of 128 x 128 (convs 9-15) the 128-column tile with a
    // split-K of 3-6 beats the 64-column tile by 8-10 %; with 320-640 tiles (convs 6-8) the 64-column tile wins or ties
    static const int64_t lo = []() { const char* e = getenv("NAFP_BN64_MIN"); return e ? atoll(e) : (int64_t)200; }();
    return (n_tiles128 < thr && n_tiles128 >= lo) ? 64 : 128;
}

// Tile plan of a FORWARD launch (rows x position slots x class order): ONE decision, shared by the launcher and by the
// workspace sizing below -- the slab is sized for the tile count and split factor of the plan that actually runs.
struct FwdPlan { int BM, pt, perm; };
static FwdPlan fwd_plan(int64_t B, const ConvGeom& g, bool full_epilogue, bool fuse0, bool force128 = false) {
    const int P = g.Fout * g.Tout;
    FwdPlan r{(fuse0 || force128) ? 128 : pick_bm(B, P, g.Cout), tile_pt(P), 0};
    if (r.BM == 256 && full_epilogue && 256 / r.pt != 8) r.BM = 128;      // FULL mode on 256 rows keeps its statistics in registers: 8 samples per position
    if (full_epilogue && !fuse0) {
        FwdTile ft = fwd_tile(g, r.BM);
        if (r.BM == 256 && ft.saved == 0.0) {
            // the 256-row tile cannot hold one class with 8 samples per position (conv8 at large batches: 16 positions
            // per class), the 128-row tile can: a third of the K-steps outweighs the larger tile's few per cent
            const FwdTile f2 = fwd_tile(g, 128);
            if (f2.saved >= 0.10) { r.BM = 128; ft = f2; }
        }
        r.pt = ft.pt; r.perm = ft.perm;
    }
    return r;
}

// [r5] The batch size the INFERENCE forward plans for.  Tile shape and split-K factor of a layer used to follow the launch's own
// batch, so the fp32 summation order of a segment -- the last bits of its fingerprint -- depended on how many segments shared its
// launch.  nafp_encoder_forward* now plan every launch as if it held fwd_plan_b() segments (640: the bench
// and generate launch size, so nothing changes there) and only the grid follows B: the bytes of a fingerprint no longer depend on
// TS_BATCH_SZ or on the launch size.  Training plans per launch as before.
int64_t fwd_plan_b() {
    static const int64_t b = []() { const char* e = getenv("NAFP_PLAN_B"); return e ? atoll(e) : (int64_t)640; }();
    return b;
}

int64_t conv_gemm_slab_floats(int64_t B, const ConvGeom& g, bool with_dgrad, int64_t plan_b) {
    const int P = g.Fout * g.Tout;
    const int64_t Bp = plan_b > 0 ? plan_b : B;
    const FwdPlan fp = fwd_plan(Bp, g, true, false);
    const int BM = fp.BM, pt = fp.pt, ST = BM / pt;
    const int64_t n_tiles = ((Bp + ST - 1) / ST) * ((P + pt - 1) / pt) * (g.Cout / BN);
    const int bn = BM == 256 ? 128 : pick_bn(n_tiles, g.Cout, live_k_steps(g));
    int S = BM == 256 ? 1 : choose_split(n_tiles * (BN / bn), live_k_steps(g), Bp * P * g.Cout, bn == 64 ? (n64_two_stage() ? 1280.0 : 1024.0) : 768.0);
    if (BM == 128 && plan_override().S > 0) S = std::max(S, plan_override().S);
    int64_t need = S > 1 ? (int64_t)S * B * P * g.Cout : 0;
    if (with_dgrad && g.Cin % BN == 0 && pick_bm(B, g.Fin * g.Tin, g.Cin) == 128) {
        const int Pd = g.Fin * g.Tin;
        const int ptd = dgrad_tile_pt(g), STd = 128 / ptd;
        const int64_t tiles_d = ((B + STd - 1) / STd) * ((Pd + ptd - 1) / ptd) * (g.Cin / BN);
        const int bnd = pick_bn(tiles_d, g.Cin);
        int Sd = choose_split(tiles_d * (BN / bnd), dgrad_k_steps(g), B * Pd * g.Cin, bnd == 64 ? 1024.0 : 768.0);
        if (dgrad_plan_override().S > 0) Sd = std::max(Sd, dgrad_plan_override().S);
        if (Sd > 1) need = std::max(need, (int64_t)Sd * B * Pd * g.Cin);
    }
    return need;
}

// Diagnostic (nafp_conv_timeline): the forward GEMM conv of the given shape stamps its phase boundaries into `buf`.
static struct { unsigned long long* buf; int64_t capacity; int cin, cout, positions; int last_grid[5]; } g_timeline = {};
int conv_timeline_set(unsigned long long* buf, int64_t capacity_u64, int cin, int cout, int positions) {
    g_timeline.buf = buf; g_timeline.capacity = capacity_u64; g_timeline.cin = cin; g_timeline.cout = cout; g_timeline.positions = positions;
    return NAFP_OK;
}
int conv_timeline_grid(int* out5) { for (int i = 0; i < 5; ++i) out5[i] = g_timeline.last_grid[i]; return NAFP_OK; }

int launch_conv_gemm(const ConvGemmArgs& a, int64_t B, const ConvGeom& g, hipStream_t st) {
    if (g.Cin % 32 != 0 || g.Cout % BN != 0 || B > (1 << 24)) return NAFP_ERR_UNSUPPORTED;
    ConvKernelParams p;
    p.x = a.x; p.wp = a.wp; p.G = a.G; p.Hb = a.Hb; p.gamma_out = a.gamma_out; p.bias = a.bias;
    p.stats_in = a.stats_in; p.stats_out = a.stats_out; p.y = a.y; p.v_out = a.v_out;
    p.Fin = g.Fin; p.Tin = g.Tin; p.Cin = g.Cin; p.Tout = g.Tout; p.Cout = g.Cout;
    p.axis = g.axis; p.stride = g.stride; p.pad = g.pad;
    p.B = (int)B; p.P = g.Fout * g.Tout;
    // forward launches: fwd_plan() (shared with conv_gemm_slab_floats); the transposed conv picks its rows from ITS output
    // (plan_b: the inference forward plans as if the launch held plan_b segments -- see fwd_plan_b())
    const int64_t Bp = (a.plan_b > 0 && !a.plain && !a.dgrad && !a.f0_feat) ? a.plan_b : B;
    // (the exact-split kernels, bf16x3 == 2, run best on 128-row tiles at three workgroups per CU: measured in profiles/r05_experiments.md)
    const bool x6 = a.bf16x3 == 2 && !a.plain && !a.dgrad && a.wp_hm && a.wp_l;
    const bool x6d = a.bf16x3 == 2 && a.plain && a.dgrad && a.wp_hm && a.wp_l;        // a transposed conv on the exact split: 128-row tiles as well
    const FwdPlan fp = fwd_plan(Bp, g, !a.plain && !a.dgrad, a.f0_feat != nullptr, x6);
    int BM = a.dgrad ? (x6d ? 128 : pick_bm(B, g.Fin * g.Tin, g.Cin)) : fp.BM;
    int pt = fp.pt;
    const int fwd_perm = a.dgrad ? 0 : fp.perm;
    p.PT = pt; p.ST = BM / pt;
    p.log2ST = 0;
    while ((1 << p.log2ST) < p.ST) ++p.log2ST;
    p.n_sg = (int)((B + p.ST - 1) / p.ST);
    p.sample_in = (int64_t)g.Fin * g.Tin * g.Cin;
    p.tap_stride = g.axis == 0 ? g.Cin : g.Tin * g.Cin;
    p.inv_n_in = a.ident_stats ? -1.0 : 1.0 / (double)p.sample_in;
    p.mode = a.plain ? 1 : 0;
    p.n_split = 1;
    p.dgrad = 0; p.perm_on = fwd_perm; p.perm_n0 = 0; p.perm_c0 = 0;
    int k_steps = live_k_steps(g);
    if (a.dgrad) {
        // backward w.r.t. the conv input: rows = input positions, source = dT (B,Fout,Tout,Cout),
        // weights = wp flipped to (Cin, 3*Cout); see row_geom()
        if (!a.plain || g.Cout % 32 != 0 || g.Cin % BN != 0 || (g.stride != 1 && g.stride != 2)) return NAFP_ERR_UNSUPPORTED;
        p.dgrad = 1;
        p.Fin = g.Fout; p.Tin = g.Tout; p.Cin = g.Cout; p.Cout = g.Cin; p.Tout = g.Tin;
        p.P = g.Fin * g.Tin;
        pt = dgrad_tile_pt(g);
        p.PT = pt; p.ST = BM / pt;
        p.log2ST = 0;
        while ((1 << p.log2ST) < p.ST) ++p.log2ST;
        p.n_sg = (int)((B + p.ST - 1) / p.ST);
        p.sample_in = (int64_t)g.Fout * g.Tout * g.Cout;
        const int S = g.axis == 0 ? g.Cout : g.Tout * g.Cout;
        p.tap_stride = -(S / g.stride);
        p.inv_n_in = 1.0;
        const int L = g.axis == 0 ? g.Tin : g.Fin;             // extent of the rows along the tap axis
        if (g.stride == 2 && L >= 2) {
            p.perm_on = 1; p.perm_c0 = g.pad & 1;              // class 0: coordinate + pad even -> taps {0, 2}
            p.perm_n0 = (L - p.perm_c0 + 1) / 2;
        } else if (g.stride == 1 && g.axis == 0 && L == 2 && g.pad == 1) {
            // stride 1 on two frames (conv8 at the 1-s input): input frame 0 receives taps {0, 1}, frame 1 taps {1, 2}: the
            // forward class order (last frame of every line last) separates them
            p.perm_on = 2;
        }
        k_steps = dgrad_k_steps(g);
    }
    // per-tile A descriptor covers ST samples: must stay below the 2 GiB OOB marker
    if ((int64_t)p.ST * p.sample_in * 4 >= ((int64_t)1 << 31)) return NAFP_ERR_UNSUPPORTED;
    const int64_t wbytes = (int64_t)g.Cout * 3 * g.Cin * 4;
    if (wbytes >= ((int64_t)1 << 31)) return NAFP_ERR_UNSUPPORTED;
    p.wp_bytes = (unsigned)wbytes;
    static const int abl = []() { const char* e = getenv("NAFP_ABL"); return e ? atoi(e) : 0; }();
    p.abl = a.plain ? 0 : abl;
    p.tl = nullptr;
    static const int gemm_prio = []() { const char* e = getenv("NAFP_GEMM_PRIO"); return e ? atoi(e) : 0; }();
    p.opt = gemm_prio;
    const int n_pb = (p.P + p.PT - 1) / p.PT;
    const int64_t n_tiles128 = (int64_t)p.n_sg * n_pb * (p.Cout / BN);
    const int64_t n_tiles128_plan = ((Bp + p.ST - 1) / p.ST) * n_pb * (p.Cout / BN);      // ... at the planning batch
    int bn = (BM == 256 || a.f0_feat) ? 128 : pick_bn(n_tiles128_plan, p.Cout, (a.plain || a.dgrad) ? 0 : k_steps);
    const bool plan_forced = BM == 128 && !a.plain && !a.dgrad && !a.f0_feat && plan_override().bn > 0;
    if (plan_forced && (plan_override().bn == 128 || p.Cout % 64 == 0)) bn = plan_override().bn;
    const bool dplan_forced = BM == 128 && a.dgrad && dgrad_plan_override().bn > 0;
    if (dplan_forced && (dgrad_plan_override().bn == 128 || p.Cout % 64 == 0)) bn = dgrad_plan_override().bn;
    const int64_t n_tiles = n_tiles128 * (BN / bn);
    int S = 1;
    const int64_t out_floats = B * p.P * p.Cout;
    if (a.slab && !a.f0_feat && BM == 128) {
        S = choose_split(n_tiles128_plan * (BN / bn), k_steps, Bp * p.P * p.Cout, bn == 64 ? ((n64_two_stage() && !a.plain) ? 1280.0 : 1024.0) : 768.0);
        if (plan_forced && plan_override().S > 0) { S = plan_override().S; while (S > 1 && k_steps / S < 4) --S; }
        if (dplan_forced && dgrad_plan_override().S > 0) { S = dgrad_plan_override().S; while (S > 1 && k_steps / S < 4) --S; }
        if ((int64_t)S * out_floats > a.slab_floats) S = 1;
        if (x6 && bn == 64 && S == 2 && (p.ST == 4 || p.ST == 8)) S = 1;      // convs 7, 9 at B = 640: unsplit on the bf16 pipe beats the two f32 parts + finish (only where the unsplit launch IS a split-arithmetic one: register-resident statistics)
    }
    if (S > 1) { p.mode = a.plain ? 1 : 2; p.n_split = S; p.y = a.slab; p.bias = nullptr; }
    // FULL split launches on 64-column tiles finish in-kernel (last-arriver) when the caller provides arrival counters
    // (measured per conv at B = 640: with >= 320 output tiles the last arrivers finish faster than a second launch --
    // conv7 0.119 -> 0.104 ms, conv9 0.193 -> 0.178 --, with 160 or fewer the finish kernel's finer split wins by 2-8 %)
    static const int fin_min = []() { const char* e = getenv("NAFP_SPLIT_INKERNEL"); return e ? atoi(e) : 320; }();
    // (decided at the planning batch like S itself: the two finishes group a sample's statistics into different partial sums)
    // (round-5 ADVICE: the launch's own tile count used to decide as well -- above ~4096 segments convs 7 and 9 fell back to the finish
    // kernel and a fingerprint's last bits depended on the launch size again; now a launch with more tiles than arrival counters runs as
    // several launches over sample ranges, see below)
    const bool in_kernel_finish = fin_min > 0 && S > 1 && !a.plain && bn == 64 && a.tickets && n_tiles128_plan * (BN / bn) >= fin_min &&
                                  (int64_t)n_pb * (p.Cout / bn) <= NAFP_TICKET_SLOTS;
    // PLAIN split launches (the transposed convs) with arrival counters: the last arriver adds the parts and stores the result
    // (its epilogue is a sum: nothing like the FULL epilogue's serial tail) -- NAFP_PLAIN_INKERNEL=0 restores slab + plain_finish_kernel
    static const int plain_fin = []() { const char* e = getenv("NAFP_PLAIN_INKERNEL"); return e ? atoi(e) : 1; }();
    const bool plain_in_kernel = plain_fin > 0 && S > 1 && a.plain && a.tickets && n_tiles <= NAFP_TICKET_SLOTS && !x6d;      // (the exact-split parts finish through plain_finish_kernel)
    p.tickets = (in_kernel_finish || plain_in_kernel) ? a.tickets : nullptr; p.y_final = a.y;
    if (plain_in_kernel) p.bias = a.bias;
    static const int grid3d = []() { const char* e = getenv("NAFP_GRID3D"); return e ? atoi(e) : 1; }();
    // NAFP_XCDMAP: 0 = the plain 3-D grids, 1 (default) = 1-D grid in XCD-aware order, the operand to keep inside one L2 chosen
    // by size (column-fastest unless the live weights outweigh the activations read), 2 / 3 = force column- / row-fastest
    static const int xcdmap = []() { const char* e = getenv("NAFP_XCDMAP"); return e ? atoi(e) : 1; }();
    const int n_col = p.Cout / bn;
    p.n_pb = n_pb; p.log2_ncol = 0; p.xcd_group = 1; p.xcd_full = 0;
    while ((1 << p.log2_ncol) < n_col) ++p.log2_ncol;
    const int64_t total_wg = (int64_t)p.n_sg * n_pb * n_col * S;
    // (launches that finish their split-K in-kernel keep the plain grid: measured at B = 640 the map costs convs 7 and 9 5 us each)
    bool xm = xcdmap != 0 && !a.f0_feat && !in_kernel_finish && !plain_in_kernel && (1 << p.log2_ncol) == n_col && total_wg < ((int64_t)1 << 31);
    if (xm) {
        const double a_bytes = (double)B * (double)p.sample_in * 4.0, w_bytes = (double)k_steps * 16.0 * p.Cout * 4.0;
        const bool row_fast = xcdmap == 3 || (xcdmap == 1 && w_bytes > a_bytes);
        p.xcd_group = row_fast ? p.n_sg * n_pb : n_col * S;
        p.xcd_full = (int)(total_wg / (8 * (int64_t)p.xcd_group) * (8 * (int64_t)p.xcd_group));
        if (p.xcd_group <= 1 || p.xcd_full == 0) xm = false;              // one item per group: the plain order already is this order
        else p.opt |= row_fast ? (8 | 16) : 8;
    }
    const bool g3 = !xm && S == 1 && grid3d != 0 && (grid3d == 1 || BM == 256 || bn == 128);
    if (g3) p.opt |= 4;
    const dim3 grid = xm ? dim3((unsigned)total_wg)
                    : g3 ? dim3((unsigned)p.n_sg, (unsigned)n_pb, (unsigned)(p.Cout / bn))
                         : dim3((unsigned)((int64_t)p.n_sg * n_pb), (unsigned)(p.Cout / bn), (unsigned)S);
    if (!xm && (n_pb > 65535 || p.Cout / bn > 65535)) return NAFP_ERR_UNSUPPORTED;
    // (cin < 0 in nafp_conv_timeline selects the transposed conv -- DGRAD -- of the layer with that |cin|)
    if (g_timeline.buf && ((!a.plain && !a.dgrad && g.Cin == g_timeline.cin) || (a.dgrad && g.Cin == -g_timeline.cin)) &&
        g.Cout == g_timeline.cout && g.Fout * g.Tout == g_timeline.positions &&
        (int64_t)grid.x * grid.y * grid.z * 64 <= g_timeline.capacity) {
        p.tl = g_timeline.buf;
        g_timeline.last_grid[0] = (int)grid.x; g_timeline.last_grid[1] = (int)grid.y; g_timeline.last_grid[2] = (int)grid.z;
        g_timeline.last_grid[3] = BM; g_timeline.last_grid[4] = bn;
    }
    int rc;
    const bool finish_follows = S > 1 && !in_kernel_finish && !plain_in_kernel;
    g_ev_start = a.ev_start; g_ev_stop = finish_follows ? nullptr : a.ev_stop;
    p.f0_feat = nullptr; p.f0_w = nullptr; p.f0_bias = nullptr; p.f0_gamma = nullptr;
    p.f0_T = 0; p.f0_stride = 1; p.f0_pad = 0; p.f0_gstat = nullptr; p.f0_group = 0; p.f0_segnorm = 0;
    p.sj = (a.sj && a.plain) ? *a.sj : ScalarsJob{nullptr, nullptr, nullptr, nullptr, 0, 0.0};
    if (a.f0_feat) {
        // conv0 generated in-kernel: this conv must be the 3x1 conv that consumes conv0's output
        if (a.plain || S != 1 || !a.f0_geom || g.axis != 1 || a.f0_geom->Cout != g.Cin || a.f0_geom->Tout != g.Tin ||
            a.f0_geom->Fin != g.Fin || g.Cin % 16 != 0)
            return NAFP_ERR_UNSUPPORTED;
        p.f0_feat = a.f0_feat; p.f0_w = a.f0_w; p.f0_bias = a.f0_bias; p.f0_gamma = a.f0_gamma;
        p.f0_T = a.f0_geom->Tin; p.f0_stride = a.f0_geom->stride; p.f0_pad = a.f0_geom->pad;
        if (a.v_out || (p.ST != 4 && p.ST != 8)) return NAFP_ERR_UNSUPPORTED;          // the fused kernel carries the inference epilogue only
        p.f0_gstat = a.f0_gstat; p.f0_group = a.f0_group; p.f0_segnorm = a.f0_segnorm;
        if (a.bf16x3 == 2 && a.wp_hm && a.wp_l) {
            if (g.Cin != 128) return NAFP_ERR_UNSUPPORTED;          // the exact-split generator is written for conv0's 128 channels
            p.wp_hm = a.wp_hm; p.wp_l = (const unsigned short*)a.wp_l;
            return launch_variant(conv_gemm_k16s2_fuse0_bf16x6, 128, 128, 16, 2, p, grid, st, 128 * 8 + 128 * 8);      // + A's third plane, + the weights' third plane
        }
        return launch_variant(conv_gemm_k16s3_fuse0, 128, 128, 16, 3, p, grid, st);
    }
    const bool fast_st = p.ST == 4 || p.ST == 8;
    if (plain_in_kernel)
        return bn == 64 ? launch_variant(conv_gemm_n64k16s3_plainfin, 128, 64, 16, 3, p, grid, st)
                        : launch_variant(conv_gemm_k16s3_plainfin, 128, 128, 16, 3, p, grid, st);
    const int epi = in_kernel_finish ? 4 : p.mode != 0 ? 3 : (!fast_st ? 2 : (p.v_out ? 1 : 0));
    const bool two_stage = n64_two_stage() && bn == 64 && (epi == 0 || epi == 1 || epi == 4);
    if (a.bf16x3 == 2 && epi == 0 && (bn == 128 || two_stage) && a.wp_hm && a.wp_l) {
        p.wp_hm = a.wp_hm; p.wp_l = (const unsigned short*)a.wp_l;
        return BM == 256 ? launch_variant(conv_gemm_m256k16s3_infer_bf16x6, 256, 128, 16, NAFP_X6_M256_NSTAGE, p, grid, st, 128 * 8)
             : bn == 64 ? launch_variant(conv_gemm_n64k16s2_infer_bf16x6, 128, 64, 16, 2, p, grid, st, 64 * 8)
                        : launch_variant(conv_gemm_k16s3_infer_bf16x6, 128, 128, 16, NAFP_X6_K16_NSTAGE, p, grid, st, 128 * 8);
    }
    // the training epilogue on the exact split (x6 forces 128-row tiles; launches that need the generic-statistics or the in-kernel-finish
    // epilogue -- the small late layers -- stay on the f32 kernels)
    static const bool x6_any = []() { const char* e = getenv("NAFP_X6_ANY"); return !e || e[0] != '0'; }();      // (A/B knob)
    if (x6 && epi == 2 && BM == 128 && x6_any) {
        p.wp_hm = a.wp_hm; p.wp_l = (const unsigned short*)a.wp_l;
        return bn == 64 ? launch_variant(conv_gemm_n64k16s2_any_bf16x6, 128, 64, 16, 2, p, grid, st, 64 * 8)
                        : launch_variant(conv_gemm_k16s3_any_bf16x6, 128, 128, 16, NAFP_X6_K16_NSTAGE, p, grid, st, 128 * 8);
    }
    if (x6 && epi == 1 && BM == 128 && (bn == 128 || two_stage)) {
        p.wp_hm = a.wp_hm; p.wp_l = (const unsigned short*)a.wp_l;
        return bn == 64 ? launch_variant(conv_gemm_n64k16s2_train_bf16x6, 128, 64, 16, 2, p, grid, st, 64 * 8)
                        : launch_variant(conv_gemm_k16s3_train_bf16x6, 128, 128, 16, NAFP_X6_K16_NSTAGE, p, grid, st, 128 * 8);
    }
    if (a.bf16x3 && epi == 0 && (bn == 128 || two_stage))
        return BM == 256 ? launch_variant(conv_gemm_m256k16s3_infer_bf16x3, 256, 128, 16, 3, p, grid, st)
             : bn == 64 ? launch_variant(conv_gemm_n64k16s2_infer_bf16x3, 128, 64, 16, 2, p, grid, st)
                        : launch_variant(conv_gemm_k16s3_infer_bf16x3, 128, 128, 16, 3, p, grid, st);
    if (x6 && epi == 3 && bn == 128 && BM == 128 && p.mode == 2) {
        p.wp_hm = a.wp_hm; p.wp_l = (const unsigned short*)a.wp_l;
        rc = launch_variant(conv_gemm_k16s3_plain_bf16x6, 128, 128, 16, NAFP_X6_K16_NSTAGE, p, grid, st, 128 * 8);
    } else if (x6d && epi == 3 && BM == 128) {
        p.wp_hm = a.wp_hm; p.wp_l = (const unsigned short*)a.wp_l;
        rc = bn == 64 ? launch_variant(conv_gemm_n64k16s2_plain_bf16x6, 128, 64, 16, 2, p, grid, st, 64 * 8)
                      : launch_variant(conv_gemm_k16s3_plain_bf16x6, 128, 128, 16, NAFP_X6_K16_NSTAGE, p, grid, st, 128 * 8);
    } else if (in_kernel_finish && n_tiles > NAFP_TICKET_SLOTS) {
        // more output tiles than arrival counters: sample ranges of at most NAFP_TICKET_SLOTS tiles, one launch each (the last arrivers
        // leave the counters at zero, so the ranges reuse them and the slab in stream order); every sample sees exactly the launch it
        // would have seen in a smaller batch
        const int64_t tiles_per_sg = (int64_t)n_pb * (p.Cout / bn);
        const int64_t sg_per_launch = NAFP_TICKET_SLOTS / tiles_per_sg;
        const int64_t b_step = sg_per_launch * p.ST;
        const int64_t out_per_sample = (int64_t)p.P * p.Cout;
        hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
        rc = NAFP_OK;
        for (int64_t b0 = 0; b0 < B && rc == NAFP_OK; b0 += b_step) {
            ConvKernelParams q = p;
            const int64_t bc = std::min<int64_t>(b_step, B - b0);
            q.B = (int)bc; q.n_sg = (int)((bc + p.ST - 1) / p.ST);
            q.x = p.x + b0 * p.sample_in; q.y_final = p.y_final + b0 * out_per_sample;
            if (p.v_out) q.v_out = p.v_out + b0 * out_per_sample;
            q.stats_in = p.stats_in + 2 * b0; q.stats_out = p.stats_out + 2 * b0;
            g_ev_start = b0 == 0 ? ev0 : nullptr; g_ev_stop = b0 + b_step >= B ? ev1 : nullptr;
            const dim3 gq((unsigned)((int64_t)q.n_sg * n_pb), (unsigned)(p.Cout / bn), (unsigned)S);
            rc = two_stage ? launch_variant(conv_gemm_n64k16s2_tab[epi], 128, 64, 16, 2, q, gq, st)
                           : launch_variant(conv_gemm_n64k16s3_tab[epi], 128, 64, 16, 3, q, gq, st);
        }
        return rc;
    } else
    rc = BM == 256 ? launch_variant(conv_gemm_m256k16s3_tab[epi], 256, 128, 16, 3, p, grid, st)
         : (bn == 64 && two_stage) ? launch_variant(conv_gemm_n64k16s2_tab[epi], 128, 64, 16, 2, p, grid, st)
         : bn == 64 ? launch_variant(conv_gemm_n64k16s3_tab[epi], 128, 64, 16, 3, p, grid, st)
                    : launch_variant(conv_gemm_k16s3_tab[epi], 128, 128, 16, 3, p, grid, st);
    if (rc != NAFP_OK || S == 1 || in_kernel_finish) return rc;
    if (a.plain) {
        const int64_t n4 = out_floats / 4;
        const dim3 fgrid((unsigned)std::min<int64_t>((n4 + 255) / 256, 8192));
        if (a.ev_stop) hipExtLaunchKernelGGL(plain_finish_kernel, fgrid, dim3(256), 0, st, nullptr, a.ev_stop, 0, (const float*)a.slab, S, a.bias, a.y, n4, p.Cout);
        else plain_finish_kernel<<<fgrid, 256, 0, st>>>(a.slab, S, a.bias, a.y, n4, p.Cout);
        NAFP_LAUNCH_CHECK();
        return NAFP_OK;
    }
    if (4096 / g.Cout + 1 > FIN_MAXS || g.Cout < 128) return NAFP_ERR_UNSUPPORTED;      // a wave (256 floats) spans <= 2 rows
    const int64_t f4 = out_floats / 4;
    const int64_t f4_plan = Bp * p.P * p.Cout / 4;          // the finish kernel's block size follows the planning batch (it sets the partial sums of the statistics)
#define NAFP_FIN(E_)                                                                                             \
    if (a.ev_stop) hipExtLaunchKernelGGL(splitk_finish_kernel<E_>, dim3((unsigned)((f4 + (E_) - 1) / (E_))), dim3(256), 0, st, nullptr,   \
                                         a.ev_stop, 0, (const float*)a.slab, S, a.G, a.Hb, a.gamma_out, a.stats_in, a.stats_out, a.y,   \
                                         a.v_out, p.B, p.P, g.Cout, p.inv_n_in);                                   \
    else splitk_finish_kernel<E_><<<dim3((unsigned)((f4 + (E_) - 1) / (E_))), 256, 0, st>>>(                       \
        a.slab, S, a.G, a.Hb, a.gamma_out, a.stats_in, a.stats_out, a.y, a.v_out, p.B, p.P, g.Cout, p.inv_n_in)
    if (f4_plan >= 1024 * 1024) { NAFP_FIN(1024); }
    else if (f4_plan >= 1024 * 512) { NAFP_FIN(512); }
    else { NAFP_FIN(256); }
#undef NAFP_FIN
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// ============================================================================
// Transposed conv FUSED with the LayerNorm + ELU backward of the layer it feeds (training, big layers).
//
// The plain chain per layer is: dgrad_j writes D'_{j-1} = conv_j^T(dts_j)  ->  ln_bwd_fused(j-1) reads D'_{j-1} and
// t_{j-1}, writes dts_{j-1}: the gradient crosses HBM twice for nothing.  The per-sample sums (s1, s2) that the
// LayerNorm backward needs are known BEFORE this launch (adjoint identity, backward.hip), so the whole element-wise part
// can run in the epilogue of the transposed conv, on the accumulators: read t_{j-1}, write dts_{j-1}.
// What remains are the sums over the BATCH per element (dgamma, dbeta, S1, S2; dbias).  To make those cheap a tile is
// ONE position x 128 samples (so every row of a tile adds into the same (position, channel) slots) and a workgroup walks
// over the sample groups of its position, keeping the five sums per owned column in registers: one set of atomics per
// workgroup, not per tile.  All rows of a tile share their source geometry and live taps (a stride-2 position is of one
// parity class by construction).  The per-sample sums of the layer below (adjoint identity again) are row sums: lanes of
// a row meet through 5 shuffles and one LDS atomic.
//   grid = (positions x chunks of sample groups, C / 128); 256 threads = 2 x 2 waves of 64 x 64, BK = 16, 3-stage ring.
// ============================================================================
struct DgradLnParams {
    ConvKernelParams c;            // the transposed conv (dgrad = 1): x = dts_j, wp = Wd_j, P / Tout / ... of its OUTPUT rows
    const float* t;                // (B, P, C)  pre-activation of layer j-1
    const float* gamma;            // (P, C)     LN scale of layer j-1
    const float* G; const float* Hb;   // (P, C) positional tensors of layer j-1 (sums of layer j-2), or null
    const float* sc;               // (B, 8)     per-sample scalars of layer j-1 (ln_bwd_scalars_kernel)
    float* dts;                    // (B, P, C)  out
    float* dgamma; float* dbeta; float* dbias;
    float* S1; float* S2;          // or null
    double* lnsum_below;           // (B, 2) or null
    int n_groups, groups_per_wg, n_chunks;
};

__global__ __launch_bounds__(256, 3) void dgrad_ln_kernel(const DgradLnParams q) {
    constexpr int BM = 128, BK = 16, NSTAGE = 3, CH = 4, RPI = 16, NI = 2;
    constexpr int TILE = BM * BK, STAGE = 2 * TILE;
    constexpr unsigned OOB = 0x80000000u;
    const ConvKernelParams& p = q.c;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [3 stages: A | B] [sSc[128][8]] [sRow[128][2]] [sCol[5][128]] [sDummy[64]]
    float* sSc = smem + NSTAGE * STAGE;
    float* sRow = sSc + BM * 8;
    float* sCol = sRow + BM * 2;
    float* sDummy = sCol + 5 * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int chunk = blockIdx.x % q.n_chunks;
    const int pos = tile_pos(p, blockIdx.x / q.n_chunks);
    const int tile_n0 = blockIdx.y * BN;
    const int K = 3 * p.Cin;
    const int C = p.Cout;                              // channels of the layer whose LayerNorm backward runs here
    const RowGeom rg = row_geom(p, pos);               // one position per workgroup: shared by every row
    unsigned tap_pack = 0; int n_live = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (rg.mask & (1u << t)) { tap_pack |= (unsigned)t << (2 * n_live); ++n_live; }
    const int cpt = p.Cin / BK, n_steps = n_live * cpt;

    unsigned voffA[NI], voffB[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int lr = wave * 32 + i * RPI + lane / CH;
        const int lc = (lane % CH) ^ ((lr >> 2) & 3);
        voffA[i] = (unsigned)(lr * (int)p.sample_in + rg.inner + lc * 4) * 4u;      // row lr = sample b0 + lr
        voffB[i] = (unsigned)((tile_n0 + lr) * K + lc * 4) * 4u;
    }
    const u32x4 rsB = make_rsrc(p.wp, p.wp_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr_t)smem + (unsigned)(wave * 32 * BK * 4);
    const int rl = lane & 31, hh = lane >> 5;
    const int rswz = (rl >> 2) & 3;
    const int aoff = (wm * 64 + rl) * BK, boff = TILE + (wn * 64 + rl) * BK;
    const int ncol = lane & 31;
    const int n_base = tile_n0 + wn * 64 + ncol;
    // positional operands of this lane's two columns
    float gam[2], Gq[2], Hq[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int o = pos * C + n_base + ni * 32;
        gam[ni] = q.gamma[o];
        Gq[ni] = q.lnsum_below ? q.G[o] : 0.f;
        Hq[ni] = q.lnsum_below ? q.Hb[o] : 0.f;
    }
    float ag[2] = {0.f, 0.f}, ab[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};

    const int g_begin = chunk * q.groups_per_wg, g_end = min(q.n_groups, g_begin + q.groups_per_wg);
    for (int grp = g_begin; grp < g_end; ++grp) {
        const int b0 = grp * BM;
        const int nb = min(BM, p.B - b0);
        // per-sample scalars of this group, zeroed row sums (visible after the first K-step barrier)
        {
            const int r = tid >> 1, h = tid & 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nb) v = *(const float4*)(q.sc + 8 * (int64_t)(b0 + r) + 4 * h);
            *(float4*)(sSc + 8 * r + 4 * h) = v;
            sRow[tid] = 0.f;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // landed before this wave reaches the first barrier
        }
        const u32x4 rsA = make_rsrc(p.x + (int64_t)b0 * p.sample_in, (unsigned)nb * (unsigned)p.sample_in * 4u);
        f32x16 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#define NAFP_DL_DMA(s_, slot_)                                                                 \
        {                                                                                      \
            const int tsel_l = (s_) / cpt;                                                     \
            const int tap_l = (int)((tap_pack >> (2 * tsel_l)) & 3u);                          \
            const int c0_l = ((s_) - tsel_l * cpt) * BK;                                       \
            const unsigned la_l = lds0 + (unsigned)((slot_) * STAGE * 4);                      \
            const unsigned tapb_l = (unsigned)(tap_l * p.tap_stride) * 4u;                     \
            _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                   \
                const int lr_l = wave * 32 + i * RPI + lane / CH;                              \
                lds_dma16(la_l + i * RPI * BK * 4, lr_l < nb ? voffA[i] + tapb_l : OOB, rsA, (unsigned)(c0_l * 4)); \
                lds_dma16(la_l + (TILE + i * RPI * BK) * 4, voffB[i], rsB, (unsigned)((tap_l * p.Cin + c0_l) * 4)); \
            }                                                                                  \
        }
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < n_steps) NAFP_DL_DMA(s, s)
        int slot = 0;
        for (int s = 0; s < n_steps; ++s) {
            if (s + NSTAGE - 2 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NSTAGE - 2) * 2 * NI) : "memory");
            __builtin_amdgcn_s_barrier();
            int nslot = slot + NSTAGE - 1; if (nslot >= NSTAGE) nslot -= NSTAGE;
            const float* St = smem + slot * STAGE;
            float4 a0[2], b0v[2];
            {
                const int pc4 = ((0 + hh) ^ rswz) * 4;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a0[mi] = *(const float4*)(St + aoff + mi * 32 * BK + pc4);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b0v[ni] = *(const float4*)(St + boff + ni * 32 * BK + pc4);
            }
            if (s + NSTAGE - 1 < n_steps) NAFP_DL_DMA(s + NSTAGE - 1, nslot)
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                const int pc4 = ((2 * kk + hh) ^ rswz) * 4;
                float4 a[2], b[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = kk == 0 ? a0[mi] : *(const float4*)(St + aoff + mi * 32 * BK + pc4);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[ni] = kk == 0 ? b0v[ni] : *(const float4*)(St + boff + ni * 32 * BK + pc4);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
                    }
            }
            if (++slot == NSTAGE) slot = 0;
        }
#undef NAFP_DL_DMA
        if (n_steps == 0) __syncthreads();             // (a position no tap reaches: D' = 0; still publish sSc / sRow)
        // ---- epilogue: LayerNorm + ELU backward of layer j-1 on the accumulators (D' = acc) ----
        // row = (r & 3) + 8 (r >> 2) + 4 hh inside each 32-row block = sample b0 + row.  t and dts go through buffer
        // descriptors re-based on the group's first sample: the lane part of the address is ONE 32-bit register for
        // the whole epilogue, the row part a running scalar, and rows beyond the batch are out of range (loads return
        // 0, stores are dropped) -- no 64-bit address arithmetic, no predication.
        {
            int PC = p.P * C;
            asm volatile("" : "+s"(PC));                   // keep the per-row scalars inside the loop (they were hoisted -> spills)
            const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(q.t) + (int64_t)b0 * PC, 0, (int)((unsigned)nb * (unsigned)PC * 4u), 0x00020000);
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
                q.dts + (int64_t)b0 * PC, 0, (int)((unsigned)nb * (unsigned)PC * 4u), 0x00020000);
            const int voff = ((wm * 64 + 4 * hh) * PC + pos * C + n_base) * 4;
            const int row_bytes = PC * 4;
            const float* sScL = sSc + 8 * (wm * 64 + 4 * hh);
            float* sRowL = sRow + 2 * (wm * 64 + 4 * hh);
            const int nbl = nb - (wm * 64 + 4 * hh);         // rows (r & 3) + 8 (r >> 2) + 32 mi < nbl are real samples
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {                      // 4 rows at a time: 8 loads in flight, then the arithmetic
                    const int soff0 = (mi * 32 + 8 * rq) * row_bytes;
                    float tv[4][2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        tv[k][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsT, voff, soff0 + k * row_bytes, 0));
                        tv[k][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsT, voff + 128, soff0 + k * row_bytes, 0));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = 4 * rq + k;
                        const int rl_ = mi * 32 + 8 * rq + k;          // row relative to this lane's first row
                        const bool valid = rl_ < nbl;
                        const float4 s0 = *(const float4*)(sScL + 8 * rl_), s1 = *(const float4*)(sScL + 8 * rl_ + 4);
                        const float mean = s0.x, rstd = s0.y, m1 = s0.z, m2 = s0.w, inv_r = s1.x, rprev = s1.y, cprev = s1.z;
                        float q1 = 0.f, q2 = 0.f;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const float tt = tv[k][ni], dd = acc[mi][ni][r];
                            const float vv = elu1(tt);
                            const float xt = (vv - mean) * rstd;
                            float dt = (dd * gam[ni] - m1 - xt * m2) * (vv > 0.f ? 1.f : vv + 1.f);
                            float du = dd * inv_r;
                            if (!valid) { dt = 0.f; du = 0.f; }
                            const float o = dt * rprev;
                            ag[ni] = fmaf(du, xt, ag[ni]); ab[ni] += du;
                            a1[ni] = fmaf(cprev, o, a1[ni]); a2[ni] += dt;
                            // pin the four running sums here: otherwise the compiler sinks all 64 rows' updates below the
                            // loop nest and spills their operands (xt, dt, the scalars) to scratch on the way
                            asm volatile("" : "+v"(ag[ni]), "+v"(ab[ni]), "+v"(a1[ni]), "+v"(a2[ni]));
                            q1 = fmaf(o, Gq[ni], q1); q2 = fmaf(o, tt - Hq[ni], q2);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o), rsO, voff + 128 * ni, soff0 + k * row_bytes, 0);
                        }
                        // the 32 lanes of a half hold the 64 columns of this row: (q1, q2) meet in lanes 31 / 63.
                        // Branch-free on purpose (the other lanes add into a scratch slot of their own; without a layer
                        // below the sums are simply not read): with a branch per row the compiler postponed the batch
                        // sums of all 64 rows to the end of the epilogue and spilled their operands.
                        q1 = half_wave_sum_dpp(q1); q2 = half_wave_sum_dpp(q2);
                        float* dstq = ncol == 31 ? sRowL + 2 * rl_ : sDummy + lane;
                        atomicAdd(dstq, q1);
                        atomicAdd(ncol == 31 ? dstq + 1 : dstq, q2);
                    }
                    // one 4-row block at a time: left to itself the scheduler hoists the operand reads of all 16
                    // blocks to the top of the epilogue and spills the accumulators to make room
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();                               // all reads of the ring / sSc and all sRow atomics are done
        if (q.lnsum_below && tid < 2 * nb) atomicAdd(q.lnsum_below + 2 * (int64_t)b0 + tid, (double)sRow[tid]);
        __syncthreads();                               // sRow / sSc are rewritten by the next group
    }
    // ---- the batch sums of this workgroup's (position, 128 columns): halves of a wave, then the two wm waves ----
    for (int i = tid; i < 5 * BN; i += 256) sCol[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        float v0 = ag[ni] + __shfl_xor(ag[ni], 32, 64), v1 = ab[ni] + __shfl_xor(ab[ni], 32, 64);
        float v2 = a1[ni] + __shfl_xor(a1[ni], 32, 64), v3 = a2[ni] + __shfl_xor(a2[ni], 32, 64);
        if (hh == 0) {
            const int cidx = wn * 64 + ni * 32 + ncol;
            atomicAdd(sCol + 0 * BN + cidx, v0); atomicAdd(sCol + 1 * BN + cidx, v1);
            atomicAdd(sCol + 2 * BN + cidx, v2); atomicAdd(sCol + 3 * BN + cidx, v3);
        }
    }
    __syncthreads();
    if (tid < BN) {
        const int o = pos * C + tile_n0 + tid;
        atomicAdd(q.dgamma + o, sCol[0 * BN + tid]);
        atomicAdd(q.dbeta + o, sCol[1 * BN + tid]);
        if (q.S1) { atomicAdd(q.S1 + o, sCol[2 * BN + tid]); atomicAdd(q.S2 + o, sCol[3 * BN + tid]); }
        atomicAdd(q.dbias + tile_n0 + tid, sCol[3 * BN + tid]);
    }
}

// Whether the transposed conv of layer geometry `g` at batch B takes the fused path (layer j-1 = the conv's input).
bool dgrad_ln_eligible(int64_t B, const ConvGeom& g, int mode) {
    if (mode == 0 || g.Cin % BN != 0 || g.Cout % 16 != 0) return false;
    if ((int64_t)BN * g.Fout * g.Tout * g.Cout * 4 >= ((int64_t)1 << 31)) return false;      // A descriptor of 128 samples
    if (mode == 2) return true;
    // the big layers only: that is where the LayerNorm backward is HBM traffic; small ones keep their launch count down
    return B >= 64 && (int64_t)g.Fin * g.Tin * g.Cin >= 32768;
}

int launch_dgrad_ln(const DgradLnArgs& a, int64_t B, const ConvGeom& g, hipStream_t st) {
    if (!dgrad_ln_eligible(B, g, 2)) return NAFP_ERR_UNSUPPORTED;
    DgradLnParams q;
    ConvKernelParams& p = q.c;
    p.x = a.dts_in; p.wp = a.wd; p.G = nullptr; p.Hb = nullptr; p.gamma_out = nullptr; p.bias = nullptr;
    p.stats_in = nullptr; p.stats_out = nullptr; p.y = nullptr; p.v_out = nullptr;
    // rows = input positions of the conv; source = dT (B, Fout, Tout, Cout); weights (Cin, 3*Cout): as launch_conv_gemm's dgrad
    p.axis = g.axis; p.stride = g.stride; p.pad = g.pad;
    p.dgrad = 1;
    p.Fin = g.Fout; p.Tin = g.Tout; p.Cin = g.Cout; p.Cout = g.Cin; p.Tout = g.Tin;
    p.B = (int)B; p.P = g.Fin * g.Tin;
    p.PT = 1; p.ST = BN; p.log2ST = 7;
    p.n_sg = (int)((B + 127) / 128);
    p.sample_in = (int64_t)g.Fout * g.Tout * g.Cout;
    const int S = g.axis == 0 ? g.Cout : g.Tout * g.Cout;
    p.tap_stride = -(S / g.stride);
    p.inv_n_in = 1.0; p.mode = 1; p.n_split = 1; p.abl = 0; p.tl = nullptr; p.opt = 0; p.tickets = nullptr; p.y_final = nullptr;
    p.sj = ScalarsJob{nullptr, nullptr, nullptr, nullptr, 0, 0.0};
    p.perm_on = 0; p.perm_n0 = 0; p.perm_c0 = 0;        // one position per workgroup: no class ordering needed
    p.wp_bytes = (unsigned)((int64_t)g.Cout * 3 * g.Cin * 4);
    p.f0_feat = nullptr; p.f0_w = nullptr; p.f0_bias = nullptr; p.f0_gamma = nullptr; p.f0_T = 0; p.f0_stride = 1; p.f0_pad = 0;
    q.t = a.t; q.gamma = a.gamma; q.G = a.G; q.Hb = a.Hb; q.sc = a.sc; q.dts = a.dts_out;
    q.dgamma = a.dgamma; q.dbeta = a.dbeta; q.dbias = a.dbias; q.S1 = a.S1; q.S2 = a.S2; q.lnsum_below = a.lnsum_below;
    q.n_groups = p.n_sg;
    // enough workgroups for two full rounds of 768 resident ones; a workgroup's atomics amortise over its sample groups
    const int64_t base = (int64_t)p.P * (g.Cin / BN);
    int chunks = (int)std::min<int64_t>(q.n_groups, std::max<int64_t>(1, (1536 + base - 1) / base));
    q.groups_per_wg = (q.n_groups + chunks - 1) / chunks;
    q.n_chunks = (q.n_groups + q.groups_per_wg - 1) / q.groups_per_wg;
    const int lds = (3 * 2 * 128 * 16 + 128 * 8 + 128 * 2 + 5 * BN + 64) * (int)sizeof(float);
    static bool attr = false;
    if (!attr) {
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)dgrad_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    dgrad_ln_kernel<<<dim3((unsigned)((int64_t)p.P * q.n_chunks), (unsigned)(g.Cin / BN)), 256, lds, st>>>(q);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// ============================================================================
// weight packing: keras kernel (3, Cin, Cout) -> Wp (Cout, 3*Cin), k contiguous,
// so that B-operand tiles load exactly like A-operand tiles.
// ============================================================================
__global__ void pack_conv_weight_kernel(const float* __restrict__ k3, float* __restrict__ wp, int K, int Cout) {
    __shared__ float t[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < K && n0 + tx < Cout) t[r][tx] = k3[(int64_t)(k0 + r) * Cout + n0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < Cout && k0 + tx < K) wp[(int64_t)(n0 + r) * K + k0 + tx] = t[tx][r];
}

// All convs' weight re-layouts of one set_weights call in ONE launch (training re-packs after every optimizer
// step): entry e, blockIdx.y = e; forward operand (Cout, 3*Cin) by 32 x 32 LDS transposes and -- training only --
// the transposed-conv operand (Cin, 3*Cout): Wd[c][k*Cout + n] = W[k][c][n].
// Work units of 8 tiles (8192 elements) over all entries, one workgroup each (entry found in the unit prefix): the entries
// range from 49 K to 3.1 M elements, and a fixed number of workgroups per entry made the launch as long as the largest one.
__global__ __launch_bounds__(256) void multi_pack_kernel(const PackTable t) {
    __shared__ float tile[32][33];
    int e = 0;
    while (e + 1 < t.count && t.unit0[e + 1] <= (int)blockIdx.x) ++e;
    const int u = (int)blockIdx.x - t.unit0[e];
    const float* __restrict__ k3 = t.k3[e];
    float* __restrict__ wp = t.wp[e]; float* __restrict__ wd = t.wd[e];
    const int Cin = t.cin[e], Cout = t.cout[e], K = 3 * Cin;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tiles_k = (K + 31) / 32, tiles_n = (Cout + 31) / 32;
    for (int tl = 8 * u; tl < 8 * u + 8 && tl < tiles_k * tiles_n; ++tl) {
        const int k0 = (tl % tiles_k) * 32, n0 = (tl / tiles_k) * 32;
        __syncthreads();
        for (int r = ty; r < 32; r += 8)
            if (k0 + r < K && n0 + tx < Cout) {
                const float w = k3[(int64_t)(k0 + r) * Cout + n0 + tx];
                tile[r][tx] = w;
                if (!(fabsf(w) <= 3.4028234664e38f) && t.nonfinite) atomicOr(t.nonfinite, 1);      // NaN / Inf: rare, see d_wflag
            }
        __syncthreads();
        for (int r = ty; r < 32; r += 8)
            if (n0 + r < Cout && k0 + tx < K) wp[(int64_t)(n0 + r) * K + k0 + tx] = tile[tx][r];
    }
    if (wd) {
        const int64_t total = (int64_t)3 * Cin * Cout;
        for (int64_t i = (int64_t)u * 8192 + threadIdx.x; i < (int64_t)(u + 1) * 8192 && i < total; i += 256) {
            const int n = (int)(i % Cout);
            const int64_t r = i / Cout;
            const int c = (int)(r % Cin), k = (int)(r / Cin);
            wd[((int64_t)c * 3 + k) * Cout + n] = k3[i];
        }
    }
}

int launch_multi_pack(const PackTable& t0, hipStream_t st) {
    if (t0.count <= 0) return NAFP_OK;
    PackTable t = t0;
    t.unit0[0] = 0;
    for (int e = 0; e < t.count; ++e) {
        const int64_t tiles = (int64_t)((3 * t.cin[e] + 31) / 32) * ((t.cout[e] + 31) / 32);
        // units of 8 tiles also cover the element-wise dgrad layout: 8 tiles x 1024 >= 8192 elements of the same entry
        t.unit0[e + 1] = t.unit0[e] + (int)((std::max<int64_t>(tiles * 1024, (int64_t)3 * t.cin[e] * t.cout[e]) + 8191) / 8192);
    }
    multi_pack_kernel<<<dim3((unsigned)t.unit0[t.count]), 256, 0, st>>>(t);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

// ============================================================================
// G_j / Hb_j of the small layers (set_weights; every train step re-runs it).  conv_j of the two "samples" gamma_{j-1}, beta_{j-1}
// is a GEMM of 2 P <= 16 rows against the whole weight tensor: through the tiled kernel that was 15 - 46 us per layer of pure
// latency (a handful of workgroups stream 2 - 12 MB behind a 3-stage ring); here it is a weight-streaming pass: a wave owns 4
// output columns, its lanes split K (float4 each, the rows of W are k-contiguous), the <= 16 input rows come from L1 / L2,
// the 64 partial sums per wave meet by butterfly.  One launch for all small layers.
// ============================================================================
__global__ __launch_bounds__(256) void gh_gemv_kernel(const GhTable t) {
    int e = 0;
    while (e + 1 < t.count && t.wg0[e + 1] <= (int)blockIdx.x) ++e;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Cin = t.cin[e], Cout = t.cout[e], P = t.P[e], K3 = 3 * Cin;
    const int n0 = ((int)blockIdx.x - t.wg0[e]) * 16 + wave * 4;
    const float* __restrict__ wp = t.wp[e];
    const float* __restrict__ x = t.x[e];
    float acc[4][16];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[nn][m] = 0.f;
    for (int tap = 0; tap < 3; ++tap) {
        bool any = false;
        for (int pos = 0; pos < P; ++pos) any = any || t.src[e][pos][tap] >= 0;
        if (!any) continue;                                         // a tap that only ever reads zero padding
        for (int c0 = 0; c0 < Cin; c0 += 256) {
            const int c = c0 + 4 * lane;
            float4 w4[4];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) w4[nn] = *(const float4*)(wp + (int64_t)(n0 + nn) * K3 + tap * Cin + c);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m >= 2 * P) break;                              // rows m = s * P + pos, s = 0 (gamma), 1 (beta)
                const int s = m >= P ? 1 : 0, pos = m - s * P;
                const int off = t.src[e][pos][tap];
                if (off < 0) continue;
                const float4 x4 = *(const float4*)(x + (int64_t)s * t.sample_in[e] + off + c);
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
                    acc[nn][m] = fmaf(w4[nn].x, x4.x, fmaf(w4[nn].y, x4.y, fmaf(w4[nn].z, x4.z, fmaf(w4[nn].w, x4.w, acc[nn][m]))));
            }
        }
    }
    float* __restrict__ y = t.y[e];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m >= 2 * P) break;
            float v = acc[nn][m];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == ((nn * 16 + m) & 63)) y[(int64_t)m * Cout + n0 + nn] = v;      // row m of [G ; Hb] (adjacent, P rows each)
        }
}

bool gh_gemv_eligible(const ConvGeom& g) {
    static const bool on = []() { const char* e = getenv("NAFP_GH_GEMV"); return !e || e[0] != '0'; }();
    const int P = g.Fout * g.Tout;
    return on && P <= 8 && g.Cin % 256 == 0 && g.Cout % 16 == 0 && (int64_t)g.Fin * g.Tin * g.Cin < ((int64_t)1 << 30);
}

void gh_table_add(GhTable& t, const ConvGeom& g, const float* wp, const float* x, float* y) {
    const int e = t.count++;
    if (e == 0) t.wg0[0] = 0;
    t.wp[e] = wp; t.x[e] = x; t.y[e] = y;
    t.cin[e] = g.Cin; t.cout[e] = g.Cout; t.P[e] = g.Fout * g.Tout; t.sample_in[e] = g.Fin * g.Tin * g.Cin;
    for (int pos = 0; pos < 8; ++pos)
        for (int tap = 0; tap < 3; ++tap) {
            int off = -1;
            if (pos < t.P[e]) {
                const int fo = pos / g.Tout, to = pos - fo * g.Tout;
                if (g.axis == 0) { const int ti = to * g.stride - g.pad + tap; if (ti >= 0 && ti < g.Tin) off = (fo * g.Tin + ti) * g.Cin; }
                else             { const int fi = fo * g.stride - g.pad + tap; if (fi >= 0 && fi < g.Fin) off = (fi * g.Tin + to) * g.Cin; }
            }
            t.src[e][pos][tap] = off;
        }
    t.wg0[e + 1] = t.wg0[e] + g.Cout / 16;
}

int launch_gh_gemv(const GhTable& t, hipStream_t st) {
    if (t.count <= 0) return NAFP_OK;
    gh_gemv_kernel<<<dim3((unsigned)t.wg0[t.count]), 256, 0, st>>>(t);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

int launch_pack_conv_weight(const float* k3, float* wp, int Cin, int Cout, hipStream_t st) {
    const int K = 3 * Cin;
    pack_conv_weight_kernel<<<dim3((K + 31) / 32, (Cout + 31) / 32), 256, 0, st>>>(k3, wp, K, Cout);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

}  // namespace nafp
