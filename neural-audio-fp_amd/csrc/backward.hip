// Backward pass of the encoder for gfx950 (train step, model/trainer.py:41-48: the reference
// gets these gradients from tf.GradientTape over m_fp).
//
// Per conv layer j the forward graph is  t = conv_j(xhat_{j-1}) + bias,  v = ELU(t),
// xt = (v - mu_b) r_b,  xhat_j = xt*gamma_j + beta_j  (LayerNorm over (F,T,C) per sample).
// Given dxh = dL/dxhat_j the kernels here produce
//   ln_bwd_reduce   s1_b = sum g, s2_b = sum g*xt            (g = dxh*gamma)  -- TOP layer only, see below
//   ln_bwd_fused    dt = r_b (g - s1/n - xt*s2/n) * ELU'(t); dts = r_{j-1,b} * dt [in place];
//                   dgamma = sum_b dxh*xt, dbeta = sum_b dxh, dbias = sum dt,
//                   S1 = sum_b c_{j-1,b} dt, S2 = sum_b dt   (c = -mu r of the layer below)
//                   (one pass: 2 reads + 1 write of the activation-sized arrays)
//                   + the (s1, s2) of the layer BELOW, without touching that layer's arrays: the gradient that
//                   layer j-1 will receive is D'_{j-1} = conv_j^T(dts_j), and for any image u
//                       sum_x D'_{j-1}(x) u(x) = sum_y dts_j(y) conv_j(u)(y)           (adjoint of the conv)
//                   so  s1_{j-1} = sum D' gamma_{j-1}        = sum_y dts_j(y) G_j(y)            (G_j = conv_j(gamma_{j-1}))
//                       s2_{j-1} = sum D' (xhat_{j-1} - beta) = sum_y dts_j(y) (t_j(y) - Hb_j(y))  (t_j = conv_j(xhat_{j-1}) + bias,
//                                                                                               Hb_j = conv_j(beta_{j-1}) + bias)
//                   with G_j, Hb_j the positional tensors of the forward pass and t_j the stored pre-activation.
//                   This replaces a separate reduction pass (2 reads of activation-sized arrays per layer).
//   wgrad           dW_j[tap,c,n] += sum_{b,pos} X[b, in(pos,tap), c] * D[b,pos,n]   (fp32 MFMA)
// and the transposed conv (dgrad) comes from conv_gemm in DGRAD mode.  With the LayerNorm fold
// of the forward pass, xhat_{j-1} = r z + c gamma + beta, so
//   dW_j = wgrad(z_{j-1}, dts_j) + wgrad(gamma_{j-1}, S1_j) + wgrad(beta_{j-1}, S2_j)
// -- the first term streams the stored z tiles raw (DMA), the other two are one-sample problems.
#include "nafp_common.h"

#include <algorithm>

namespace nafp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// (mean, rstd) per sample and layer from the double (sum, sumsq) statistics.
__global__ void stats_to_mr_kernel(const stat_t* __restrict__ stats, float* __restrict__ mr, int64_t n_pairs,
                                   const double* __restrict__ inv_n_per_layer, int64_t B, int identity) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    if (identity) {                       // the normalisation alternates (stat_ln_scalars): mean 0, rstd 1; NaN for a poisoned sample
        const double z = stat_get(stats + 2 * i + 1) * 0.0;
        mr[2 * i] = (float)z; mr[2 * i + 1] = (float)(1.0 + z);
        return;
    }
    const double inv_n = inv_n_per_layer[i / B];
    const double mean = stat_get(stats + 2 * i) * inv_n;
    double var = stat_get(stats + 2 * i + 1) * inv_n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    mr[2 * i] = (float)mean;
    mr[2 * i + 1] = (float)(1.0 / sqrt(var + (double)LN_EPS));
}

// FROMZ: `tpre` is the stored z = gamma . v of the layer, not its pre-activation: v = z / gamma (see ln_bwd_fused_kernel)
template <bool FROMZ>
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(
        const float* __restrict__ dxh, const float* __restrict__ tpre, const float* __restrict__ gamma,
        const float* __restrict__ mr, double* __restrict__ lnsum, int64_t n) {
    const int64_t b = blockIdx.y;
    const float mean = mr[2 * b], rstd = mr[2 * b + 1];
    const float4* d4 = (const float4*)(dxh + b * n);
    const float4* v4 = (const float4*)(tpre + b * n);
    const float4* g4 = (const float4*)gamma;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * 256) {
        const float4 d = d4[i], tq = v4[i], gg = g4[i];
        const float4 vv = FROMZ ? make_float4(__fdividef(tq.x, gg.x), __fdividef(tq.y, gg.y), __fdividef(tq.z, gg.z), __fdividef(tq.w, gg.w))
                                : make_float4(elu1(tq.x), elu1(tq.y), elu1(tq.z), elu1(tq.w));
        const float g0 = d.x * gg.x, g1 = d.y * gg.y, g2 = d.z * gg.z, g3 = d.w * gg.w;
        s1 += (g0 + g1) + (g2 + g3);
        s2 += g0 * ((vv.x - mean) * rstd) + g1 * ((vv.y - mean) * rstd) + g2 * ((vv.z - mean) * rstd) +
              g3 * ((vv.w - mean) * rstd);
    }
    const double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave] = d1; red[4 + wave] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(lnsum + 2 * b, red[0] + red[1] + red[2] + red[3]);
        atomicAdd(lnsum + 2 * b + 1, red[4] + red[5] + red[6] + red[7]);
    }
}

// Per-sample scalars of one layer's LayerNorm backward, one 32-B record per sample (read with a
// scalar load in the fused kernel): mean, rstd, s1/n, s2/n, 1/rstd, r of the layer below, -mu of
// the layer below.
__global__ void ln_bwd_scalars_kernel(const float* __restrict__ mr, const double* __restrict__ lnsum,
                                      const float* __restrict__ mr_prev, float* __restrict__ sc, int64_t B, double inv_n) {
    const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float mean = mr[2 * b], rstd = mr[2 * b + 1];
    float* o = sc + 8 * b;
    o[0] = mean; o[1] = rstd;
    o[2] = (float)(lnsum[2 * b] * inv_n); o[3] = (float)(lnsum[2 * b + 1] * inv_n);
    o[4] = 1.f / rstd;
    o[5] = mr_prev ? mr_prev[2 * b + 1] : 1.f;
    o[6] = mr_prev ? -mr_prev[2 * b] : 0.f;
    o[7] = 0.f;
}

// The fused LayerNorm + ELU backward of one layer.  Convention: the incoming gradient is
// D' = r_{j,b} * dL/dxhat_j (pre-scaled by the layer's own rstd: the transposed conv that produced
// it consumed dts of the layer above, see below), so with g' = D' gamma, s1' = sum g', s2' = sum g' xt:
//   dt  = (g' - s1'/n - xt s2'/n) * ELU'(v)                      = dL/dt_j
//   dts = r_{j-1,b} * dt     written IN PLACE: operand of wgrad (against the raw z_{j-1}) and of
//                            the transposed conv, whose output is then D' of layer j-1
//   dgamma += sum_b D' xt / r_b,  dbeta += sum_b D' / r_b,  dbias[c] += sum dt,
//   S1 += sum_b (-mu_{j-1,b}) dts = sum_b c_b dt,   S2 += sum_b dt.
// Thread <-> one float4 of the (P, C) plane, loop over a chunk of the batch (blockIdx.y): every
// per-element sum over b stays in registers and meets the other chunks through one atomic each.
// CONV0: the layer is b0.conv1x3 (Cin = 1).  Its pre-activation is not stored by the forward pass -- 2 MB per segment
// written and read back for 3 FMAs per element -- but regenerated here from the log-mel features:
// t[b, f, to, c] = bias[c] + sum_k w[k, c] feat[b, f, to * stride - pad + k]   (the same fmaf chain as conv0_kernel).
// The same kernel then also forms conv0's weight gradient dW0[k, c] = sum feat[b, f, to*s - p + k] * dt[b, f, to, c] from
// the dt it has just produced (it holds both factors), so the gradient of the first layer is not read a second time.
struct Conv0Regen { const float* feat; const float* w3; const float* bias; float* dW0; int F, Tin, Tout, stride, pad; };

// FROMZ (layers >= 1 by default, NAFP_KEEP_T=0): the forward pass keeps ONE tensor per layer, z = gamma . v (the operand of the
// next conv and of the weight gradient), and `tpre` points at it.  What this pass needs of the pre-activation t follows from
// v = z / gamma (the working gamma is never zero: multi_copy_kernel): ELU'(t) = v + 1 (t <= 0) or 1, xhat from v, and
// t = v (v > 0) or log(v + 1) for the adjoint-identity sum of the layer below -- there t only ever appears multiplied by
// dts, which carries the factor v + 1, so the saturated end (v -> -1, t -> -inf) contributes (v+1) log(v+1) -> 0.
// Saves the second store stream of every training forward conv (~10 % of its time) and 4.6 MB per segment of workspace.
// WAVE (the small layers, n <= 4096 elements per sample): a workgroup covers 64 float4 columns instead of 256, and its four
// waves take a quarter each of the workgroup's batch chunk.  Same number of threads, same samples per thread, but the four waves'
// sums meet in LDS first, so a launch ends in a quarter of the atomics (they are what such a launch mostly consists of: 32
// chunks x 5 atomics per element onto the same addresses at ~33 G/s).
// Cache policy of the pass's three streams (gradient in, stored activation in, dts out, each touched once per launch):
// -DNAFP_LNB_NT=1 loads / 2 stores / 3 both non-temporal (tools/build_variant.sh; A/B in DESIGN.md 4.6 [r5])
#ifndef NAFP_LNB_NT
#define NAFP_LNB_NT 0
#endif
typedef float f4nt_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lnb_ld_nt(const float4* p) { const f4nt_t v = __builtin_nontemporal_load((const f4nt_t*)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void lnb_st_nt(float4* p, float4 v) { const f4nt_t t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (f4nt_t*)p); }
#if NAFP_LNB_NT & 1
#define NAFP_LNB_LD(p_) lnb_ld_nt(p_)
#else
#define NAFP_LNB_LD(p_) (*(p_))
#endif
#if NAFP_LNB_NT & 2
#define NAFP_LNB_ST(p_, v_) lnb_st_nt(p_, v_)
#else
#define NAFP_LNB_ST(p_, v_) (*(p_) = (v_))
#endif
template <bool CONV0, bool FROMZ = false, bool WAVE = false>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(
        float* __restrict__ d, const float* __restrict__ tpre, const float* __restrict__ gamma,
        const float* __restrict__ sc, float* __restrict__ dgamma, float* __restrict__ dbeta,
        float* __restrict__ dbias, float* __restrict__ S1, float* __restrict__ S2, int64_t n, int64_t B, int C,
        const float* __restrict__ Gj, const float* __restrict__ Hbj, double* __restrict__ lnsum_below, const Conv0Regen c0,
        float* __restrict__ part_slab, unsigned* __restrict__ tickets) {
    extern __shared__ float s_q[];                                 // [per][2]: (s1, s2) of the layer below, this block's share
    const int lane = threadIdx.x & 63;
    const int64_t i = WAVE ? blockIdx.x * 64ll + lane : blockIdx.x * 256ll + threadIdx.x;
    const bool live = i < n / 4;
    const int64_t ii = live ? i : 0;
    const int64_t per = (B + gridDim.y - 1) / gridDim.y;
    const int64_t B0 = blockIdx.y * per, B1 = std::min<int64_t>(B, B0 + per);           // the workgroup's batch chunk
    const int64_t per_w = WAVE ? (B1 - B0 + 3) / 4 : 0;                                   // WAVE: this wave's quarter of it
    const int64_t b0 = WAVE ? std::min<int64_t>(B1, B0 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * per_w) : B0;   // (wave-uniform: scalar loads of the sample records)
    const int64_t b1 = WAVE ? std::min<int64_t>(B1, b0 + per_w) : B1;
    const float4 gg = ((const float4*)gamma)[ii];
    const float4 rgg = FROMZ ? make_float4(1.f / gg.x, 1.f / gg.y, 1.f / gg.z, 1.f / gg.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 Gq = make_float4(0.f, 0.f, 0.f, 0.f), Hq = Gq;
    if (!CONV0 && lnsum_below) {
        Gq = ((const float4*)Gj)[ii]; Hq = ((const float4*)Hbj)[ii];
        for (int64_t k = threadIdx.x; k < 2 * per; k += 256) s_q[k] = 0.f;
        __syncthreads();
    }
    // CONV0: this thread's position (f, to), its 4 channels' kernel taps and bias
    float4 k0 = make_float4(0.f, 0.f, 0.f, 0.f), k1 = k0, k2 = k0, kb = k0;
    int x_off = 0; bool x_ok[3] = {false, false, false};
    if (CONV0) {
        const int ch = (int)((4 * ii) % C), pp = (int)((4 * ii) / C);
        const int f = pp / c0.Tout, to = pp - f * c0.Tout;
        const int t0 = to * c0.stride - c0.pad;
        k0 = *(const float4*)(c0.w3 + ch); k1 = *(const float4*)(c0.w3 + C + ch); k2 = *(const float4*)(c0.w3 + 2 * C + ch);
        kb = *(const float4*)(c0.bias + ch);
        x_off = f * c0.Tin + t0;
#pragma unroll
        for (int k = 0; k < 3; ++k) x_ok[k] = t0 + k >= 0 && t0 + k < c0.Tin;
    }
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag, a1 = ag, a2 = ag;
    float4 gw0 = ag, gw1 = ag, gw2 = ag;                    // CONV0: dW0 taps 0..2 of this thread's 4 channels
    // Samples are taken UNR at a time, and the loads of the NEXT group are issued before the current group is worked on
    // (two register sets, ping-pong).  The in-place store of a sample orders every later load of `d` behind it (same
    // pointer), so a plain loop keeps ONE sample's 32 bytes per thread in flight, and a group without the look-ahead pays one
    // memory round trip per group: at a per-rank batch of 640 the mid layers (256 workgroups, 160 samples each) ran at
    // 1.5 - 2.5 TB/s on latency alone.  (More batch chunks instead cost more than they gain: every chunk ends in 5 atomics
    // per element -- NAFP_LNB_WGS 512 / 1024 / 2048 measured slower.)
    constexpr int UNR = 4;
    struct Grp { float4 dd[UNR], tt[UNR], s0[UNR], s1[UNR]; float x[UNR][3]; };
    Grp ga, gb;
#define NAFP_LN_LOAD(G_, bq_)                                                                                  \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                          \
        const int64_t b_l = std::min<int64_t>((bq_) + u, b1 - 1);        /* (a clamped duplicate is loaded, not used) */ \
        G_.s0[u] = *(const float4*)(sc + 8 * b_l); G_.s1[u] = *(const float4*)(sc + 8 * b_l + 4);             \
        G_.dd[u] = NAFP_LNB_LD((const float4*)(d + b_l * n) + ii);                                             \
        G_.x[u][0] = 0.f; G_.x[u][1] = 0.f; G_.x[u][2] = 0.f;                                                  \
        if (CONV0) {                                                                                           \
            const float* xr = c0.feat + b_l * (int64_t)c0.F * c0.Tin + x_off;                                  \
            G_.x[u][0] = x_ok[0] ? xr[0] : 0.f; G_.x[u][1] = x_ok[1] ? xr[1] : 0.f; G_.x[u][2] = x_ok[2] ? xr[2] : 0.f; \
            G_.tt[u] = make_float4(0.f, 0.f, 0.f, 0.f);                                                        \
        } else {                                                                                               \
            G_.tt[u] = NAFP_LNB_LD((const float4*)(tpre + b_l * n) + ii);                                      \
        }                                                                                                      \
    }
#define NAFP_LN_ONE(c_)                                                                     \
        {                                                                                   \
            const float vv_l = FROMZ ? tt.c_ * rgg.c_ : elu1(tt.c_);         /* 1 / gamma is per element: formed once per thread */ \
            const float w_l = fmaxf(vv_l + 1.f, 0.f);                      /* = exp(t) where t <= 0 */ \
            /* t; where w = 0 its weight below is 0 and the clamp keeps the product finite */          \
            const float tv_l = !FROMZ ? tt.c_ : (vv_l > 0.f ? vv_l : 0.69314718056f * __builtin_amdgcn_logf(fmaxf(w_l, 1.17549435e-38f))); \
            const float xt = (vv_l - mean) * rstd;                                          \
            const float dt = (dd.c_ * gg.c_ - m1 - xt * m2) * (vv_l > 0.f ? 1.f : w_l);     /* ELU'(t) = v + 1, t <= 0 */ \
            const float du = dd.c_ * inv_r;                                                 \
            ag.c_ = fmaf(du, xt, ag.c_); ab.c_ += du;                                       \
            o.c_ = dt * rprev;                                                              \
            a2.c_ += dt;                                                                    \
            if constexpr (!CONV0) {                    /* (layer 0 has no layer below: S1, the sums below and G / Hb are not formed) */ \
                a1.c_ = fmaf(cprev, o.c_, a1.c_);                                           \
                q1 = fmaf(o.c_, Gq.c_, q1); q2 = fmaf(o.c_, tv_l - Hq.c_, q2);              \
            }                                                                               \
        }
#define NAFP_LN_WORK(G_, bq_)                                                                                  \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                          \
        const int64_t b = (bq_) + u;                                                                           \
        if (b >= b1) break;                                                                                    \
        const float4 s0 = G_.s0[u], s1 = G_.s1[u];                                                             \
        const float mean = s0.x, rstd = s0.y, m1 = s0.z, m2 = s0.w, inv_r = s1.x, rprev = s1.y, cprev = s1.z;  \
        float4* dp = (float4*)(d + b * n) + ii;                                                                \
        const float4 dd = G_.dd[u];                                                                            \
        float4 tt = G_.tt[u];                                                                                  \
        const float x0 = G_.x[u][0], x1 = G_.x[u][1], x2 = G_.x[u][2];                                         \
        if (CONV0) {                                                                                           \
            tt.x = fmaf(x2, k2.x, fmaf(x1, k1.x, fmaf(x0, k0.x, kb.x)));                                       \
            tt.y = fmaf(x2, k2.y, fmaf(x1, k1.y, fmaf(x0, k0.y, kb.y)));                                       \
            tt.z = fmaf(x2, k2.z, fmaf(x1, k1.z, fmaf(x0, k0.z, kb.z)));                                       \
            tt.w = fmaf(x2, k2.w, fmaf(x1, k1.w, fmaf(x0, k0.w, kb.w)));                                       \
        }                                                                                                      \
        float4 o;                                                                                              \
        float q1 = 0.f, q2 = 0.f;                                                                              \
        NAFP_LN_ONE(x) NAFP_LN_ONE(y) NAFP_LN_ONE(z) NAFP_LN_ONE(w)                                            \
        if (CONV0 && live) {                                   /* o = dt here (no layer below: rprev = 1) */   \
            gw0.x = fmaf(x0, o.x, gw0.x); gw0.y = fmaf(x0, o.y, gw0.y); gw0.z = fmaf(x0, o.z, gw0.z); gw0.w = fmaf(x0, o.w, gw0.w); \
            gw1.x = fmaf(x1, o.x, gw1.x); gw1.y = fmaf(x1, o.y, gw1.y); gw1.z = fmaf(x1, o.z, gw1.z); gw1.w = fmaf(x1, o.w, gw1.w); \
            gw2.x = fmaf(x2, o.x, gw2.x); gw2.y = fmaf(x2, o.y, gw2.y); gw2.z = fmaf(x2, o.z, gw2.z); gw2.w = fmaf(x2, o.w, gw2.w); \
        }                                                                                                      \
        if (live && !(CONV0 && c0.dW0)) NAFP_LNB_ST(dp, o);    /* (nothing reads dts_0 once dW0 is formed here) */ \
        if (!CONV0 && lnsum_below) {                                                                           \
            /* wave sums of (q1, q2) in 7 shuffles: fold the halves, then q1 lives in lanes 0..31 and q2 in 32..63 */ \
            if (!live) { q1 = 0.f; q2 = 0.f; }                                                                 \
            q1 += __shfl_xor(q1, 32, 64); q2 += __shfl_xor(q2, 32, 64);                                        \
            float x = lane < 32 ? q1 : q2;                                                                     \
            _Pragma("unroll") for (int o2 = 16; o2 > 0; o2 >>= 1) x += __shfl_xor(x, o2, 64);                  \
            if ((lane & 31) == 0) atomicAdd(s_q + 2 * (b - B0) + (lane >> 5), x);                              \
        }                                                                                                      \
    }
    if (CONV0) {
        // (layer 0 regenerates its pre-activation and forms dW0: at 229 registers the look-ahead cost it its occupancy --
        // 385 -> 467 us at a batch of 640 -- so it keeps the plain groups)
        for (int64_t bq = b0; bq < b1; bq += UNR) {
            NAFP_LN_LOAD(ga, bq)
            NAFP_LN_WORK(ga, bq)
        }
    } else {
        if (b0 < b1) { NAFP_LN_LOAD(ga, b0) }
        for (int64_t bq = b0; bq < b1; bq += 2 * UNR) {
            if (bq + UNR < b1) { NAFP_LN_LOAD(gb, bq + UNR) }
            NAFP_LN_WORK(ga, bq)
            if (bq + UNR >= b1) break;
            if (bq + 2 * UNR < b1) { NAFP_LN_LOAD(ga, bq + 2 * UNR) }
            NAFP_LN_WORK(gb, bq + UNR)
        }
    }
#undef NAFP_LN_ONE
#undef NAFP_LN_WORK
#undef NAFP_LN_LOAD
    if (!CONV0 && lnsum_below) {
        __syncthreads();
        for (int64_t k = threadIdx.x; k < 2 * (B1 - B0); k += 256)
            atomicAdd(lnsum_below + 2 * B0 + k, (double)s_q[k]);
    }
    if constexpr (WAVE) {
        // wave w adds up array w (dgamma, dbeta, S1, S2) of this block's 64 columns over the four waves; the S2 sums are also the
        // block's share of dbias (a block of 256 consecutive elements lies in one position: C % 256 == 0, distinct channels per lane)
        __shared__ float4 wred[4][256];
        wred[0][threadIdx.x] = ag; wred[1][threadIdx.x] = ab; wred[2][threadIdx.x] = a1; wred[3][threadIdx.x] = a2;
        __syncthreads();
        const int wv = threadIdx.x >> 6;
        if (live) {
            float4 t = wred[wv][lane];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float4 u = wred[wv][64 * k + lane];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            float* o = (wv == 0 ? dgamma : (wv == 1 ? dbeta : (wv == 2 ? S1 : S2))) + 4 * i;
            if (!part_slab) {
                atomicAdd(o, t.x); atomicAdd(o + 1, t.y); atomicAdd(o + 2, t.z); atomicAdd(o + 3, t.w);
                if (wv == 3) {
                    float* ob = dbias + (4 * i) % C;
                    atomicAdd(ob, t.x); atomicAdd(ob + 1, t.y); atomicAdd(ob + 2, t.z); atomicAdd(ob + 3, t.w);
                }
                return;
            }
            // The larger layers at a small batch (n > 4096, B <= 2048): the batch chunks of a 256-element block meet through the slab
            // -- [block][chunk][array][256], written through the caches -- and an arrival ticket per block; in the LAST arriver wave w
            // adds up array w over the chunks, in chunk order, and stores it once.  A quarter of the chunks per element of the
            // 1024-element blocks above for the same samples per thread, and the last arriver's pass runs four arrays abreast.
            typedef int v4i __attribute__((ext_vector_type(4)));
            const int ny = (int)gridDim.y;
            float* base = part_slab + (int64_t)blockIdx.x * ny * 1024;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, ny * 4096, 0x00020000);
            const int mine = (wv * 256 + 4 * lane) * 4;                                 // byte offset inside a chunk's 4 x 256 floats
            if (ny > 1) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, t), rs, (int)blockIdx.y * 4096 + mine, 0, 17);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __shared__ unsigned s_wtk;
            __syncthreads();
            if (threadIdx.x == 0) s_wtk = ny > 1 ? __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            __syncthreads();
            if ((int)s_wtk != ny - 1) return;
            if (ny > 1) {
                if (threadIdx.x == 0) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int y0 = 0; y0 < ny; y0 += 4) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, std::min(y0 + u, ny - 1) * 4096 + mine, 0, 17));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (y0 + u >= ny) break;
                        t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w;
                    }
                }
            }
            *(float4*)o = t;
            if (wv == 3) {
                // dbias: with C = 128 the block spans two positions and lanes l, l + 32 hold the same channels
                if (C == 128) { t.x += __shfl_xor(t.x, 32, 64); t.y += __shfl_xor(t.y, 32, 64); t.z += __shfl_xor(t.z, 32, 64); t.w += __shfl_xor(t.w, 32, 64); }
                if (C != 128 || lane < 32) {
                    float* ob = dbias + (4 * i) % C;
                    atomicAdd(ob, t.x); atomicAdd(ob + 1, t.y); atomicAdd(ob + 2, t.z); atomicAdd(ob + 3, t.w);
                }
            }
        }
        return;
    }
    if (part_slab) {
        // The batch chunks (blockIdx.y) of a block of 1024 elements meet through a slab instead of 4 fp32 atomics per element
        // and chunk onto the same addresses (~33 G contended atomics/s: 15 - 28 % of this kernel at a batch of 640): every
        // chunk writes its (dgamma, dbeta, S1, S2) partials through the caches, draws the block's arrival ticket, and the
        // LAST arriver adds the parts in chunk order and stores the four tensors once -- deterministic as a by-product.
        typedef int v4i __attribute__((ext_vector_type(4)));
        const int n_arr = (!CONV0 && S1) ? 4 : 2;
        const int64_t blk_floats = (int64_t)gridDim.y * n_arr * 1024;                   // per x-block: [chunk][array][1024]
        float* base = part_slab + (int64_t)blockIdx.x * blk_floats;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(blk_floats * 4), 0x00020000);
        const int my = ((int)blockIdx.y * n_arr * 1024 + 4 * (int)threadIdx.x) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, ag), rs, my, 0, 17);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, ab), rs, my + 4096, 0, 17);
        if (!CONV0 && S1) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, a1), rs, my + 8192, 0, 17);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, a2), rs, my + 12288, 0, 17);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ unsigned s_tk;
        if (threadIdx.x == 0) s_tk = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_tk == gridDim.y - 1) {
            if (threadIdx.x == 0) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float4 t4[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) t4[a] = make_float4(0.f, 0.f, 0.f, 0.f);
            // (chunks four at a time: the 16 loads of a group are issued before the first add -- the adds keep the chunk order)
            const int ny = (int)gridDim.y;
            for (int y0 = 0; y0 < ny; y0 += 4) {
                float4 v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int off = (std::min(y0 + u, ny - 1) * n_arr * 1024 + 4 * (int)threadIdx.x) * 4;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        v[u][a] = a < n_arr ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + a * 4096, 0, 17))
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (y0 + u >= ny) break;
#pragma unroll
                    for (int a = 0; a < 4; ++a) { t4[a].x += v[u][a].x; t4[a].y += v[u][a].y; t4[a].z += v[u][a].z; t4[a].w += v[u][a].w; }
                }
            }
            *(float4*)(dgamma + 4 * i) = t4[0]; *(float4*)(dbeta + 4 * i) = t4[1];
            if (!CONV0 && S1) { *(float4*)(S1 + 4 * i) = t4[2]; *(float4*)(S2 + 4 * i) = t4[3]; }
        }
    } else if (live) {
        float* g = dgamma + 4 * i; float* bt = dbeta + 4 * i;
        atomicAdd(g, ag.x); atomicAdd(g + 1, ag.y); atomicAdd(g + 2, ag.z); atomicAdd(g + 3, ag.w);
        atomicAdd(bt, ab.x); atomicAdd(bt + 1, ab.y); atomicAdd(bt + 2, ab.z); atomicAdd(bt + 3, ab.w);
        if (!CONV0 && S1) {
            float* o1 = S1 + 4 * i; float* o2 = S2 + 4 * i;
            atomicAdd(o1, a1.x); atomicAdd(o1 + 1, a1.y); atomicAdd(o1 + 2, a1.z); atomicAdd(o1 + 3, a1.w);
            atomicAdd(o2, a2.x); atomicAdd(o2 + 1, a2.y); atomicAdd(o2 + 2, a2.z); atomicAdd(o2 + 3, a2.w);
        }
    }
    // dbias[c] = sum over positions and samples of dt = the a2 sums of every thread on channel c
    __shared__ float4 red[256];
    red[threadIdx.x] = live ? a2 : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int cg = C / 4;                                   // float4 channel groups; 256 % cg == 0 or cg % 256 == 0
    if (cg >= 256) {
        if (live) {
            float* o = dbias + (4 * i) % C;
            atomicAdd(o, a2.x); atomicAdd(o + 1, a2.y); atomicAdd(o + 2, a2.z); atomicAdd(o + 3, a2.w);
        }
    } else if ((int)threadIdx.x < cg) {
        float4 t = red[threadIdx.x];
        for (int r = threadIdx.x + cg; r < 256; r += cg) {
            const float4 u = red[r];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        float* o = dbias + 4 * threadIdx.x;                 // block start is a multiple of 1024 elements, C | 1024
        atomicAdd(o, t.x); atomicAdd(o + 1, t.y); atomicAdd(o + 2, t.z); atomicAdd(o + 3, t.w);
    }
    if (CONV0 && c0.dW0) {
        // dW0[k][c]: the same reduction over the threads of a channel group, once per tap (C / 4 <= 256 here)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            __syncthreads();
            red[threadIdx.x] = k == 0 ? gw0 : (k == 1 ? gw1 : gw2);
            __syncthreads();
            if ((int)threadIdx.x < cg) {
                float4 t = red[threadIdx.x];
                for (int r = threadIdx.x + cg; r < 256; r += cg) {
                    const float4 u = red[r];
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                float* o = c0.dW0 + k * C + 4 * threadIdx.x;
                atomicAdd(o, t.x); atomicAdd(o + 1, t.y); atomicAdd(o + 2, t.z); atomicAdd(o + 3, t.w);
            }
        }
    }
}

// ============================================================================
// wgrad: dW[tap, c, n] += sum_{rows m = (b,pos)} X[b, in(pos,tap), c] * D[m, n]      fp32 MFMA
//   workgroup tile = 128 c x 128 n for ONE tap, over a chunk of rows (split-K over the rows);
//   K-step = 16 rows: X rows (128 floats) and D rows (128 floats) are staged by LDS-DMA exactly
//   as they lie in memory (row-contiguous), and the MFMA operands are read ALONG the rows
//   (lane <-> c resp. n), so no transpose and no swizzle is needed.  Padding taps and rows
//   beyond the batch are out-of-range DMA lanes (zeros).  Partial sums of the row chunks meet in
//   dW (keras layout, n contiguous across lanes) through fp32 atomics.
// ============================================================================
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4b make_rsrc_b(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4b r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void lds_dma16_b(unsigned lds_addr, unsigned voff, u32x4b rsrc, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// (ScalarsJob, nafp_common.h: side job of a wgrad launch -- the per-sample scalar records of the LayerNorm backward of the
// layer BELOW, what ln_bwd_scalars_kernel computes: its inputs are final once ln_bwd_fused of this layer has run, its consumer
// is the next ln_bwd_fused launch; 15 three-workgroup launches of ~10 us each on the serial chain of the backward pass otherwise.)
struct WgradParams {
    const float* X; const float* D; float* dW;
    int B, P, Tout, Fin, Tin, Cin, Cout, axis, stride, pad;
    long long sample_in;
    int rows_per_wg;
    int tap_pack, n_live;          // live taps, 2 bits each: taps that read real data for at least one output position
    // n_aux = 2 "samples" appended BEHIND the batch (rows B*P ... (B+2)*P - 1 of the same GEMM): X2 = [gamma_{j-1} ; beta_{j-1}]
    // (2 x sample_in, adjacent), D2 = [S1_j ; S2_j] (2 x P x Cout, adjacent): the two rank-one terms of dW_j,
    //   dW_j = wgrad(z_{j-1}, r dt) + wgrad(gamma_{j-1}, sum_b c_b dt) + wgrad(beta_{j-1}, sum_b dt),
    // ride in the main launch instead of two launches of their own (30 launches of ~13 us per backward pass).
    const float* X2; const float* D2; int n_aux; long long M_main;
    // small-P kernel: the row chunks of an output tile meet through a slab + an arrival ticket per tile; the LAST arriver adds
    // the parts in chunk order and stores dW (no atomics: deterministic, and dW is written once)
    float* slab; unsigned* tickets; int n_chunks;
    ScalarsJob sj;
};

__device__ __forceinline__ void wgrad_scalars_side_job(const ScalarsJob& j) {
    if (!j.sc) return;
    const long long wg = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
    const long long n_wg = (long long)gridDim.x * gridDim.y * gridDim.z;
    for (long long b = wg * 256 + threadIdx.x; b < j.B; b += n_wg * 256) {
        const float mean = j.mr[2 * b], rstd = j.mr[2 * b + 1];
        float* o = j.sc + 8 * b;
        o[0] = mean; o[1] = rstd;
        o[2] = (float)(j.lnsum[2 * b] * j.inv_n); o[3] = (float)(j.lnsum[2 * b + 1] * j.inv_n);
        o[4] = 1.f / rstd;
        o[5] = j.mr_prev ? j.mr_prev[2 * b + 1] : 1.f;
        o[6] = j.mr_prev ? -j.mr_prev[2 * b] : 0.f;
        o[7] = 0.f;
    }
}

__global__ __launch_bounds__(256, 3) void wgrad_kernel(const WgradParams p) {
    constexpr int NST = 3, KR = 16;
    constexpr int TILE = KR * 128, STAGE = 2 * TILE;           // X rows | D rows
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wn = wave & 1;
    const int ctiles = p.Cin / 128;
    // blockIdx.z enumerates (live tap, c-tile): a tap that only ever sees zero padding (b5-b7 at the 1-s input) has a
    // zero gradient, which the zeroed dW already holds
    const int tap = (p.tap_pack >> (2 * (blockIdx.z / ctiles))) & 3, c0 = (blockIdx.z % ctiles) * 128, n0 = blockIdx.y * 128;
    wgrad_scalars_side_job(p.sj);
    const long long M = (long long)p.B * p.P;
    const long long m0 = (long long)blockIdx.x * p.rows_per_wg;
    const long long m_end = std::min<long long>(M, m0 + p.rows_per_wg);
    const int n_steps = (int)((m_end - m0 + KR - 1) / KR);
    const int b_first = (int)(m0 / p.P);

    // this lane's DMA rows: instruction i (0,1) of wave w covers rows 4w + 2i + (lane>>5).
    // (b, pos) of those rows are carried from step to step (no division in the loop); scalars and
    // a macro rather than arrays captured by a lambda, which hipcc would place in scratch.
    const int q16 = KR / p.P, r16 = KR % p.P;
    const long long mA = m0 + 4 * wave + (lane >> 5), mB = mA + 2;
    int rbA = (int)(mA / p.P), rposA = (int)(mA - (long long)rbA * p.P);
    int rbB = (int)(mB / p.P), rposB = (int)(mB - (long long)rbB * p.P);
    const int chunk = lane & 31;
    const long long x_bytes = ((long long)(m_end - 1) / p.P - b_first + 1) * p.sample_in * 4;
    const u32x4b rsX = make_rsrc_b(p.X + (long long)b_first * p.sample_in, (unsigned)std::min<long long>(x_bytes, 0x7fffffffll));
    const u32x4b rsD = make_rsrc_b(p.D + m0 * p.Cout, (unsigned)((m_end - m0) * p.Cout * 4));
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;

#define NAFP_WG_DMA_ROW(i_, RB, RPOS, s_, slot_)                                                       \
    {                                                                                                   \
        const int r_l = 4 * wave + 2 * (i_);                   /* wave-uniform first row */             \
        const long long m_l = m0 + (long long)(s_) * KR + r_l + (lane >> 5);                            \
        unsigned vx_l = OOB, vd_l = OOB;                                                                \
        if (m_l < m_end) {                                                                              \
            const int fo_l = RPOS / p.Tout, to_l = RPOS - fo_l * p.Tout;                                \
            int src_l; bool ok_l;                                                                       \
            if (p.axis == 0) { const int t_l = to_l * p.stride - p.pad + tap; ok_l = t_l >= 0 && t_l < p.Tin; src_l = (fo_l * p.Tin + t_l) * p.Cin; } \
            else             { const int f_l = fo_l * p.stride - p.pad + tap; ok_l = f_l >= 0 && f_l < p.Fin; src_l = (f_l * p.Tin + to_l) * p.Cin; } \
            if (ok_l) vx_l = (unsigned)(((long long)(RB - b_first) * p.sample_in + src_l + c0 + 4 * chunk) * 4); \
            vd_l = (unsigned)((((long long)(s_) * KR + r_l + (lane >> 5)) * p.Cout + n0 + 4 * chunk) * 4); \
        }                                                                                               \
        const unsigned la_l = lds0 + (unsigned)(((slot_) * STAGE + r_l * 128) * 4);                     \
        lds_dma16_b(la_l, vx_l, rsX, 0u);                                                               \
        lds_dma16_b(la_l + TILE * 4, vd_l, rsD, 0u);                                                    \
        RB += q16; RPOS += r16;                                /* advance by KR rows */                 \
        if (RPOS >= p.P) { RPOS -= p.P; RB += 1; }                                                      \
    }
#define NAFP_WG_DMA_STEP(s_, slot_) { NAFP_WG_DMA_ROW(0, rbA, rposA, s_, slot_) NAFP_WG_DMA_ROW(1, rbB, rposB, s_, slot_) }

    f32x16 acc[2][2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ci][ni][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < n_steps) NAFP_WG_DMA_STEP(s, s)

    const int rl = lane & 31, hh = lane >> 5;
    int slot = 0;
    for (int s = 0; s < n_steps; ++s) {
        if (s + NST - 2 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // one younger step (4 DMA) may stay in flight
        __builtin_amdgcn_s_barrier();
        if (s + NST - 1 < n_steps) {
            int nslot = slot + NST - 1; if (nslot >= NST) nslot -= NST;
            NAFP_WG_DMA_STEP(s + NST - 1, nslot)
        }
        const float* Xt = smem + slot * STAGE;
        const float* Dt = Xt + TILE;
#pragma unroll
        for (int kp = 0; kp < KR / 2; ++kp) {
            float a[2], bq[2];
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) a[ci] = Xt[(2 * kp + hh) * 128 + wc * 64 + ci * 32 + rl];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bq[ni] = Dt[(2 * kp + hh) * 128 + wn * 64 + ni * 32 + rl];
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ci], bq[ni], acc[ci][ni], 0, 0, 0);
        }
        if (++slot == NST) slot = 0;
    }
    // D[i = c][j = n]: lane holds n = lane & 31, rows c = (r&3) + 8(r>>2) + 4*hh
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + wc * 64 + ci * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int n = n0 + wn * 64 + ni * 32 + rl;
                atomicAdd(p.dW + ((long long)tap * p.Cin + c) * p.Cout + n, acc[ci][ni][r]);
            }
}

// wgrad, regular shapes (every layer of the 1-s encoder down to P = 16): Tout | 16, 16 | P, Fout a power of two and the
// source extent along the tap axis = stride x the output extent.  Then a lane's DMA rows move through X in a straight
// line -- row m = m0 + 16 s + r sits at (line = m >> log2 Tout, to = r mod Tout) and the source line index is linear
// in `line` ACROSS sample boundaries -- so a K-step's addresses are a per-lane base (computed once) plus a scalar
// offset, rows beyond the chunk are cut by the D descriptor's range (X may read on: it multiplies zeros), and the only
// per-step vector work is the padding test of the rows at a sample's edge when the taps run along F.
// The f32 MFMAs share the SIMD's issue time with every VALU instruction (see conv.hip): the generic kernel above spends
// ~160 VALU instructions per 32 MFMAs on row bookkeeping, this one at most 12.  The loop body exists once per ring slot
// so that LDS addresses are immediates.
// PREC = 2 (the train step under NAFP_OPT_BF16X3 = 2, experimental): the products on the bf16 matrix pipe from the exact 3-way split of BOTH operands
// (each is an activation: nothing to pre-split), six v_mfma_f32_32x32x16_bf16 per 16 rows and 32 x 32 block instead of eight f32 MFMAs;
// lane (rl, hh) then supplies k = 8 hh + j = rows 8 hh .. 8 hh + 7 of the K-step for its column pair.
template <int PREC>
__device__ __forceinline__ void wgrad_fast_body(const WgradParams& p, const int lt, const int lfo) {
    constexpr int NST = 3, KR = 16;
    constexpr int TILE = KR * 128, STAGE = 2 * TILE;           // X rows | D rows
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wn = wave & 1;
    const int ctiles = p.Cin / 128;
    const int tap = (p.tap_pack >> (2 * (blockIdx.z / ctiles))) & 3, c0 = (blockIdx.z % ctiles) * 128, n0 = blockIdx.y * 128;
    wgrad_scalars_side_job(p.sj);
    // rows [0, M_main) are the batch, rows [M_main, M) the aux samples (X2 / D2): the launcher makes M_main and every chunk
    // start multiples of the rows a K-step consumes, so a K-step lies entirely on one side and only its buffer descriptors
    // and scalar offsets change (the per-lane address parts are linear in the row index across sample boundaries)
    const long long M = (long long)(p.B + p.n_aux) * p.P;
    const long long m0 = (long long)blockIdx.x * p.rows_per_wg;               // multiple of 16 (32 with aux rows)
    const long long m_end = std::min<long long>(M, m0 + p.rows_per_wg);
    const long long main_end = std::min<long long>(m_end, p.M_main);
    // Two output frames per line (Tout = 2) and a tap that reads real data for only ONE of them -- conv8's taps 0 and 2
    // (stride 1 on two frames), conv6's tap 2 (its last frame hangs over the end): every second row of this tap's GEMM is
    // zero padding.  Such a tap walks only the live rows: a K-step takes 16 LINES at the live frame instead of 8 lines x 2
    // frames, half the K-steps for the same sum (wgrad_8 was 87 TFLOP/s of useful work, a third of its rows zeros).
    int to_sel = -1;
    if (p.axis == 0 && lt == 1) {
        const int t0 = -p.pad + tap, t1 = p.stride - p.pad + tap;
        const bool v0 = t0 >= 0 && t0 < p.Tin, v1 = t1 >= 0 && t1 < p.Tin;
        if (v0 != v1) to_sel = v1 ? 1 : 0;
    }
    const int rows_per_step = to_sel >= 0 ? 2 * KR : KR;        // rows of the launch's row order consumed per K-step
    const int n_steps = (int)((m_end - m0 + rows_per_step - 1) / rows_per_step);
    const int b_first = (int)(m0 / p.P);
    const int chunk = lane & 31, hh = lane >> 5;
    const long long x_bytes = main_end > m0 ? ((long long)(main_end - 1) / p.P - b_first + 1) * p.sample_in * 4 : 0;
    const u32x4b rsX = make_rsrc_b(p.X + (long long)b_first * p.sample_in, (unsigned)std::min<long long>(x_bytes, 0x7fffffffll));
    const u32x4b rsD = make_rsrc_b(p.D + m0 * p.Cout, (unsigned)(std::max<long long>(main_end - m0, 0) * p.Cout * 4));
    const u32x4b rsX2 = make_rsrc_b(p.X2, (unsigned)(p.n_aux * p.sample_in * 4));
    const u32x4b rsD2 = make_rsrc_b(p.D2, (unsigned)(std::max<long long>(m_end - p.M_main, 0) * p.Cout * 4));   // rows beyond this chunk: out of range
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;

    // per-lane bases of the two DMA instructions of this wave (rows r = 4 wave + 2 i + hh of every step)
    const int dl = to_sel >= 0 ? KR : (KR >> lt);              // source/output lines per step
    const int fmask = (1 << lfo) - 1;
    // The aux rows start at K-step s_aux of this chunk (INT_MAX: none).  From there on the lane parts are re-based on X2 / D2
    // (vxa / vda = the lane's offsets at step s_aux) and the scalar part counts steps from s_aux: both parts stay
    // non-negative -- the hardware adds voffset and soffset without wrapping, so a "negative" scalar part is out of range.
    const unsigned dX = (unsigned)((p.axis == 0 ? dl : dl * p.stride) * p.Tin * p.Cin * 4);   // bytes per step
    const unsigned dD = (unsigned)(rows_per_step * p.Cout * 4);
    const int s_aux = (p.n_aux && m_end > p.M_main) ? (int)(std::max<long long>(p.M_main - m0, 0) / rows_per_step) : 0x7fffffff;
    const long long x_rebase = s_aux == 0x7fffffff ? 0 : (long long)s_aux * dX - ((long long)p.B - b_first) * p.sample_in * 4;
    const long long d_rebase = s_aux == 0x7fffffff ? 0 : (long long)s_aux * dD - (p.M_main - m0) * p.Cout * 4;
    unsigned vxb[2], vdb[2], vxa[2], vda[2]; int fo0[2];
    bool okc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 4 * wave + 2 * i + hh;
        if (to_sel >= 0) {                                      // r-th LIVE row of the step: line (m0 >> 1) + r, frame to_sel
            const long long line = (m0 >> 1) + r;
            const int t = to_sel * p.stride - p.pad + tap;
            const long long src = (line * p.Tin + t) * p.Cin;
            okc[i] = true; fo0[i] = (int)(line & fmask);
            const long long vx = (src - (long long)b_first * p.sample_in + c0 + 4 * chunk) * 4;
            const long long vd = ((long long)(2 * r + to_sel) * p.Cout + n0 + 4 * chunk) * 4;
            vxb[i] = (unsigned)vx; vdb[i] = (unsigned)vd;
            vxa[i] = (unsigned)(vx + x_rebase); vda[i] = (unsigned)(vd + d_rebase);
            continue;
        }
        const long long line = ((m0 + r) >> lt);               // = b * Fout + fo of step 0
        const int to = r & ((1 << lt) - 1);
        long long src;                                         // floats from the start of X
        if (p.axis == 0) {
            const int t = to * p.stride - p.pad + tap;
            okc[i] = t >= 0 && t < p.Tin;
            src = (line * p.Tin + t) * p.Cin;
        } else {
            okc[i] = true;
            src = (((long long)p.stride * line - p.pad + tap) * p.Tin + to) * p.Cin;
        }
        fo0[i] = (int)(line & fmask);
        const long long vx = (src - (long long)b_first * p.sample_in + c0 + 4 * chunk) * 4;
        const long long vd = ((long long)r * p.Cout + n0 + 4 * chunk) * 4;
        vxb[i] = (unsigned)vx; vdb[i] = (unsigned)vd;
        vxa[i] = (unsigned)(vx + x_rebase); vda[i] = (unsigned)(vd + d_rebase);      // (a padding row may wrap: it is masked)
    }
    const int Fout = 1 << lfo;

#define NAFP_WGF_DMA(s_, slot_)                                                                        \
    {                                                                                                  \
        const bool aux_l = (s_) >= s_aux;                                                  /* wave-uniform */ \
        const u32x4b rx_l = aux_l ? rsX2 : rsX, rd_l = aux_l ? rsD2 : rsD;                             \
        const unsigned srel_l = (unsigned)(aux_l ? (s_) - s_aux : (s_));                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
            bool ok_l = okc[i];                                                                        \
            if (p.axis == 1) {                                 /* padding rows at the edge of a sample */ \
                const int f_l = ((fo0[i] + (s_) * dl) & fmask) * p.stride - p.pad + tap;               \
                ok_l = f_l >= 0 && f_l < p.stride * Fout;                                              \
            }                                                                                          \
            const unsigned la_l = lds0 + (unsigned)(((slot_) * STAGE + (4 * wave + 2 * i) * 128) * 4); \
            lds_dma16_b(la_l, ok_l ? (aux_l ? vxa[i] : vxb[i]) : OOB, rx_l, srel_l * dX);              \
            lds_dma16_b(la_l + TILE * 4, aux_l ? vda[i] : vdb[i], rd_l, srel_l * dD);                  \
        }                                                                                              \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ci][ni][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < n_steps) NAFP_WGF_DMA(s, s)

    // MFMA operands: lane (rl, hh) owns the column PAIR 2 rl, 2 rl + 1 of its wave's 64 columns (tiles ci / ni = 0, 1 are
    // the even / odd columns), so that the two values of a k-row are one 8-byte LDS read at base + immediate.
    const int rl = lane & 31;
    typedef float f32x2w __attribute__((ext_vector_type(2)));
    const float* Xl = smem + (PREC == 2 ? 8 : 1) * hh * 128 + wc * 64 + 2 * rl;
    const float* Dl = smem + TILE + (PREC == 2 ? 8 : 1) * hh * 128 + wn * 64 + 2 * rl;
    int slot = 0;
    for (int s = 0; s < n_steps; ++s) {
        if (s + NST - 2 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // one younger step (4 DMA) may stay in flight
        __builtin_amdgcn_s_barrier();
        const float* Xs = Xl + slot * STAGE;
        const float* Ds = Dl + slot * STAGE;
        if (PREC == 2) {
            typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
            f32x2w a[8], bq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] = *(const f32x2w*)(Xs + j * 128);
                bq[j] = *(const f32x2w*)(Ds + j * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            int nslot6 = slot + NST - 1; if (nslot6 >= NST) nslot6 -= NST;
            if (s + NST - 1 < n_steps) { NAFP_WGF_DMA(s + NST - 1, nslot6) }
            __builtin_amdgcn_sched_barrier(0);
            // the split, two k at a time: one packed conversion per term and pair, the bf16 -> f32 back-conversions as a shift and a mask of the
            // packed word (11 vector instructions per pair; written per element the compiler spent 19 incl. the moves that assemble the operands)
            typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
            u32x4w uh[4], um[4], ul[4];            // [0, 1]: A columns ci = 0, 1; [2, 3]: B columns ni = 0, 1
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    f32x2w x;
                    x.x = q < 2 ? a[2 * jp][q] : bq[2 * jp][q - 2];
                    x.y = q < 2 ? a[2 * jp + 1][q] : bq[2 * jp + 1][q - 2];
                    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2w));
                    f32x2w r;
                    r.x = x.x - __uint_as_float(h2 << 16);
                    r.y = x.y - __uint_as_float(h2 & 0xffff0000u);
                    const unsigned m2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2w));
                    f32x2w t;
                    t.x = r.x - __uint_as_float(m2 << 16);
                    t.y = r.y - __uint_as_float(m2 & 0xffff0000u);
                    uh[q][jp] = h2; um[q][jp] = m2; ul[q][jp] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2w));
                }
            }
            bf16x8w ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ah[q] = __builtin_bit_cast(bf16x8w, uh[q]); am[q] = __builtin_bit_cast(bf16x8w, um[q]); al[q] = __builtin_bit_cast(bf16x8w, ul[q]);
                bh[q] = __builtin_bit_cast(bf16x8w, uh[2 + q]); bm[q] = __builtin_bit_cast(bf16x8w, um[2 + q]); bl[q] = __builtin_bit_cast(bf16x8w, ul[2 + q]);
            }
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ci], bh[ni], acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ci], bm[ni], acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ci], bl[ni], acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ci], bh[ni], acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ci], bm[ni], acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ci], bh[ni], acc[ci][ni], 0, 0, 0);
                }
            if (++slot == NST) slot = 0;
            continue;
        }
        f32x2w a[KR / 2], bq[KR / 2];
#pragma unroll
        for (int kp = 0; kp < KR / 2; ++kp) {
            a[kp] = *(const f32x2w*)(Xs + 2 * kp * 128);
            bq[kp] = *(const f32x2w*)(Ds + 2 * kp * 128);
        }
        __builtin_amdgcn_sched_barrier(0);
        int nslot = slot + NST - 1; if (nslot >= NST) nslot -= NST;
#pragma unroll
        for (int kp = 0; kp < KR / 2; ++kp) {
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp][ci], bq[kp][ni], acc[ci][ni], 0, 0, 0);
            if (kp == 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + NST - 1 < n_steps) { NAFP_WGF_DMA(s + NST - 1, nslot) }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++slot == NST) slot = 0;
    }
#undef NAFP_WGF_DMA
    // D[i = c][j = n]: lane holds n = 2 (lane & 31) + ni, rows c = 2 ((r&3) + 8(r>>2) + 4*hh) + ci
    if (p.slab) {
        // the row chunks of a tile meet in wgrad_reduce_kernel (next launch): partial tile -> slab[tile][chunk], row-major
        // 128 x 128, plain coalesced 8-byte stores.  (fp32 atomics straight into dW run at ONE element per clock and L2
        // channel -- 768 workgroups x 16 K of them were ~47 us of every layer, whatever its size -- and made the sum
        // depend on the arrival order.)
        float* part = p.slab + ((long long)(blockIdx.y + gridDim.y * blockIdx.z) * gridDim.x + blockIdx.x) * (128 * 128);
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c_l = wc * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hh) + ci;
                f32x2w v2; v2.x = acc[ci][0][r]; v2.y = acc[ci][1][r];
                *(f32x2w*)(part + c_l * 128 + wn * 64 + 2 * rl) = v2;
            }
        return;
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + wc * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hh) + ci;
                const int n = n0 + wn * 64 + 2 * rl + ni;
                atomicAdd(p.dW + ((long long)tap * p.Cin + c) * p.Cout + n, acc[ci][ni][r]);
            }
}

__global__ __launch_bounds__(256, 3) void wgrad_fast_kernel(const WgradParams p, const int lt, const int lfo) { wgrad_fast_body<0>(p, lt, lfo); }
__global__ __launch_bounds__(256, 3) void wgrad_fast_bf16x6_kernel(const WgradParams p, const int lt, const int lfo) { wgrad_fast_body<2>(p, lt, lfo); }

// dW tile = sum over the row chunks of its partial tiles, in a fixed order (deterministic), stored once.
// grid = (128 slices of 128 elements, n-tiles, live taps x c-tiles) -- the wgrad grid's y / z.  A lane owns 2 elements of the slice
// and each of the four waves a quarter of the chunks (contiguous, in order); the four partial sums meet in LDS and are added in
// wave order.  (32 slices of 512 elements with one thread walking ALL chunks of its 2 elements were 96 workgroups and 32 dependent
// memory round trips for layer 1's 3 tiles x 256 chunks.)
constexpr int WGRAD_REDUCE_SLICES = 128;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dW, int n_chunks,
                                                           int Cin, int Cout, int tap_pack) {
    typedef float f32x2w __attribute__((ext_vector_type(2)));
    const int ctiles = Cin / 128;
    const int tap = (tap_pack >> (2 * (blockIdx.z / ctiles))) & 3, c0 = (blockIdx.z % ctiles) * 128, n0 = blockIdx.y * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 128 + lane * 2;                      // this lane's 2 elements of the 128 x 128 tile (one row per slice)
    const f32x2w* src = (const f32x2w*)(slab + ((long long)(blockIdx.y + gridDim.y * blockIdx.z) * n_chunks) * (128 * 128) + e);
    const int per = (n_chunks + 3) / 4, k0 = wave * per, k1 = min(n_chunks, k0 + per);
    f32x2w t = {0.f, 0.f};
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        f32x2w v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (long long)(k + u) * (128 * 128 / 2));
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; k < k1; ++k) t += __builtin_nontemporal_load(src + (long long)k * (128 * 128 / 2));
    __shared__ f32x2w red[3][64];
    if (wave > 0) red[wave - 1][lane] = t;
    __syncthreads();
    if (wave == 0) {
        t += red[0][lane]; t += red[1][lane]; t += red[2][lane];
        const int c = c0 + (e >> 7), n = n0 + (e & 127);
        *(f32x2w*)(dW + ((long long)tap * Cin + c) * Cout + n) = t;
    }
}

// wgrad for the SMALL layers: P = Fout * Tout a power of two below 16 (every layer from b5 on at the 1-s input: P = 8, 4, 4, 2,
// 2, 1).  There the GEMM's reduction dimension -- rows m = (sample, position) -- is short (640 ... 5120 rows at a batch of
// 640) and its output, the weight tensor, is the large side (0.5 ... 3.1 M elements): what the generic kernel paid for was
// (i) ~160 vector instructions of row bookkeeping per K-step, (ii) 16 K fp32 atomics per workgroup -- 6 ... 23 row chunks
// each adding a full 128 x 128 tile into dW --, (iii) the two rank-one terms as launches of their own.  Here:
//   * 16 rows of a K-step are 16 / P whole samples, so a lane's DMA row keeps its position (hence its tap geometry and
//     padding test) for the whole loop and moves 16 / P samples per step: per-lane base + scalar offset, no vector
//     instruction in the loop besides the MFMAs (as wgrad_fast_kernel);
//   * the tile is 128 c x BNT n (BNT = 64 doubles the tiles of the layers with few of them), the row chunks are as few as
//     fill the chip (n_chunks, often 1 or 2), and they meet through a slab: every chunk writes its partial tile through
//     the caches, draws an arrival ticket for the tile, and the LAST arriver adds the parts in chunk order and stores dW
//     once (n_chunks == 1: straight from the accumulators).  No atomics: the gradient is bit-reproducible;
//   * the aux samples (gamma | beta against S1 | S2) are rows B*P ... of the same loop.
// PREC = 2: as wgrad_fast_body<2> -- the exact 3-way bf16 split of both operands in registers, six bf16 MFMAs per 16 rows and 32 x 32 block.
template <int BNT, int PREC = 0>
__global__ __launch_bounds__(256, 3) void wgrad_smallp_kernel(const WgradParams p, const int lp) {
    constexpr int NST = 3, KR = 16;
    constexpr int TX = KR * 128, TD = KR * BNT, STAGE = TX + TD;   // X rows | D rows
    constexpr int NIW = BNT / 64;                                  // n-tiles (32 columns) per wave: wave = 64 c x BNT / 2 n
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wn = wave & 1;
    const int ctiles = p.Cin / 128;
    const int tap = (p.tap_pack >> (2 * (blockIdx.z / ctiles))) & 3, c0 = (blockIdx.z % ctiles) * 128, n0 = blockIdx.y * BNT;
    wgrad_scalars_side_job(p.sj);
    const long long M = (long long)(p.B + p.n_aux) * p.P;
    const long long m0 = (long long)blockIdx.x * p.rows_per_wg;               // multiple of 16
    const long long m_end = std::min<long long>(M, m0 + p.rows_per_wg);
    const long long main_end = std::min<long long>(m_end, p.M_main);
    const int n_steps = (int)((m_end - m0 + KR - 1) / KR);
    const int b_first = (int)(m0 >> lp);
    const int chunk = lane & 31, hh = lane >> 5;
    const long long x_bytes = main_end > m0 ? (((main_end - 1) >> lp) - b_first + 1) * p.sample_in * 4 : 0;
    const u32x4b rsX = make_rsrc_b(p.X + (long long)b_first * p.sample_in, (unsigned)std::min<long long>(x_bytes, 0x7fffffffll));
    const u32x4b rsD = make_rsrc_b(p.D + m0 * p.Cout, (unsigned)(std::max<long long>(main_end - m0, 0) * p.Cout * 4));
    const u32x4b rsX2 = make_rsrc_b(p.X2, (unsigned)(p.n_aux * p.sample_in * 4));
    const u32x4b rsD2 = make_rsrc_b(p.D2, (unsigned)(std::max<long long>(m_end - p.M_main, 0) * p.Cout * 4));
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    const unsigned dX = (unsigned)((KR >> lp) * p.sample_in * 4);             // bytes per K-step: 16 / P samples
    const unsigned dD = (unsigned)(KR * p.Cout * 4);
    // aux rows from K-step s_aux on: lane parts re-based on X2 / D2, scalar part counted from s_aux (see wgrad_fast_kernel)
    const int s_aux = (p.n_aux && m_end > p.M_main) ? (int)(std::max<long long>(p.M_main - m0, 0) / KR) : 0x7fffffff;
    const long long x_rebase = s_aux == 0x7fffffff ? 0 : (long long)s_aux * dX - ((long long)p.B - b_first) * p.sample_in * 4;
    const long long d_rebase = s_aux == 0x7fffffff ? 0 : (long long)s_aux * dD - (p.M_main - m0) * p.Cout * 4;

    // X: instruction i of wave w stages rows r = 4 w + 2 i + hh (32 lanes x 16 B = the 128 channels of a row)
    unsigned vxb[2], vxa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 4 * wave + 2 * i + hh;
        const int pos = r & (p.P - 1);
        const int fo = pos / p.Tout, to = pos - fo * p.Tout;
        int src; bool ok;
        if (p.axis == 0) { const int t = to * p.stride - p.pad + tap; ok = t >= 0 && t < p.Tin; src = (fo * p.Tin + t) * p.Cin; }
        else             { const int f = fo * p.stride - p.pad + tap; ok = f >= 0 && f < p.Fin; src = (f * p.Tin + to) * p.Cin; }
        const long long vx = ((long long)(r >> lp) * p.sample_in + src + c0 + 4 * chunk) * 4;
        vxb[i] = ok ? (unsigned)vx : OOB;
        vxa[i] = ok ? (unsigned)(vx + x_rebase) : OOB;
    }
    // D: BNT = 128: the same row mapping; BNT = 64: one instruction covers 4 rows of 16 lanes
    unsigned vdb[NIW], vda[NIW];
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int r = BNT == 128 ? 4 * wave + 2 * i + hh : 4 * wave + (lane >> 4);
        const int ch = BNT == 128 ? chunk : (lane & 15);
        const long long vd = ((long long)r * p.Cout + n0 + 4 * ch) * 4;
        vdb[i] = (unsigned)vd; vda[i] = (unsigned)(vd + d_rebase);
    }

#define NAFP_WSP_DMA(s_, slot_)                                                                        \
    {                                                                                                  \
        const bool aux_l = (s_) >= s_aux;                                                  /* wave-uniform */ \
        const u32x4b rx_l = aux_l ? rsX2 : rsX, rd_l = aux_l ? rsD2 : rsD;                             \
        const unsigned srel_l = (unsigned)(aux_l ? (s_) - s_aux : (s_));                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
            lds_dma16_b(lds0 + (unsigned)(((slot_) * STAGE + (4 * wave + 2 * i) * 128) * 4), aux_l ? vxa[i] : vxb[i], rx_l, srel_l * dX); \
        _Pragma("unroll") for (int i = 0; i < NIW; ++i)                                                \
            lds_dma16_b(lds0 + (unsigned)(((slot_) * STAGE + TX + (4 * wave + 2 * i) * BNT) * 4), aux_l ? vda[i] : vdb[i], rd_l, srel_l * dD); \
    }

    f32x16 acc[2][NIW];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ci][ni][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < n_steps) NAFP_WSP_DMA(s, s)

    // MFMA operands: lane (rl, hh) owns the channel pair 2 rl, 2 rl + 1 of its wave's 64 channels (tiles ci = 0, 1 = even /
    // odd channels) and -- BNT = 128 -- the column pair 2 rl, 2 rl + 1 of its wave's 64 columns, or -- BNT = 64 -- column rl of 32
    const int rl = lane & 31;
    typedef float f32x2w __attribute__((ext_vector_type(2)));
    const float* Xl = smem + (PREC == 2 ? 8 : 1) * hh * 128 + wc * 64 + 2 * rl;
    const float* Dl = smem + TX + (PREC == 2 ? 8 : 1) * hh * BNT + wn * (BNT / 2) + (BNT == 128 ? 2 * rl : rl);
    int slot = 0;
    for (int s = 0; s < n_steps; ++s) {
        if (s + NST - 2 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 + NIW) : "memory");      // one younger step may stay in flight
        __builtin_amdgcn_s_barrier();
        const float* Xs = Xl + slot * STAGE;
        const float* Ds = Dl + slot * STAGE;
        if (PREC == 2) {
            // lane (rl, hh) supplies k = 8 hh + j = rows 8 hh .. 8 hh + 7 of the K-step; the split two k at a time (wgrad_fast_body)
            typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
            typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
            f32x2w a[8], bq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] = *(const f32x2w*)(Xs + j * 128);
                if (BNT == 128) bq[j] = *(const f32x2w*)(Ds + j * BNT);
                else { bq[j].x = Ds[j * BNT]; bq[j].y = 0.f; }
            }
            __builtin_amdgcn_sched_barrier(0);
            int nslot6 = slot + NST - 1; if (nslot6 >= NST) nslot6 -= NST;
            if (s + NST - 1 < n_steps) { NAFP_WSP_DMA(s + NST - 1, nslot6) }
            __builtin_amdgcn_sched_barrier(0);
            u32x4w uh[2 + NIW], um[2 + NIW], ul[2 + NIW];          // [0, 1]: X channels ci = 0, 1; [2 ..]: D columns
#pragma unroll
            for (int q = 0; q < 2 + NIW; ++q) {
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    f32x2w x;
                    x.x = q < 2 ? a[2 * jp][q] : bq[2 * jp][q - 2];
                    x.y = q < 2 ? a[2 * jp + 1][q] : bq[2 * jp + 1][q - 2];
                    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2w));
                    f32x2w r;
                    r.x = x.x - __uint_as_float(h2 << 16);
                    r.y = x.y - __uint_as_float(h2 & 0xffff0000u);
                    const unsigned m2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2w));
                    f32x2w t;
                    t.x = r.x - __uint_as_float(m2 << 16);
                    t.y = r.y - __uint_as_float(m2 & 0xffff0000u);
                    uh[q][jp] = h2; um[q][jp] = m2; ul[q][jp] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2w));
                }
            }
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni) {
                    const bf16x8w xh = __builtin_bit_cast(bf16x8w, uh[ci]), xm = __builtin_bit_cast(bf16x8w, um[ci]), xl = __builtin_bit_cast(bf16x8w, ul[ci]);
                    const bf16x8w dh = __builtin_bit_cast(bf16x8w, uh[2 + ni]), dm = __builtin_bit_cast(bf16x8w, um[2 + ni]), dl = __builtin_bit_cast(bf16x8w, ul[2 + ni]);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, dh, acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, dm, acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, dl, acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, dh, acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, dm, acc[ci][ni], 0, 0, 0);
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, dh, acc[ci][ni], 0, 0, 0);
                }
            if (++slot == NST) slot = 0;
            continue;
        }
        f32x2w a[KR / 2], bq[KR / 2];
#pragma unroll
        for (int kp = 0; kp < KR / 2; ++kp) {
            a[kp] = *(const f32x2w*)(Xs + 2 * kp * 128);
            if (BNT == 128) bq[kp] = *(const f32x2w*)(Ds + 2 * kp * BNT);
            else { bq[kp].x = Ds[2 * kp * BNT]; bq[kp].y = 0.f; }
        }
        __builtin_amdgcn_sched_barrier(0);
        int nslot = slot + NST - 1; if (nslot >= NST) nslot -= NST;
#pragma unroll
        for (int kp = 0; kp < KR / 2; ++kp) {
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
                    acc[ci][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp][ci], bq[kp][ni], acc[ci][ni], 0, 0, 0);
            if (kp == 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + NST - 1 < n_steps) { NAFP_WSP_DMA(s + NST - 1, nslot) }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++slot == NST) slot = 0;
    }
#undef NAFP_WSP_DMA
    // C/D layout: lane holds column n_l (+ ni for BNT = 128), rows c_l(r) = 2 ((r&3) + 8 (r>>2) + 4 hh) + ci of its wave's block
    const int n_l = wn * (BNT / 2) + (BNT == 128 ? 2 * rl : rl);
    const int tile_id = blockIdx.y + gridDim.y * blockIdx.z;
    if (p.n_chunks > 1) {
        // my partial tile -> slab[tile][chunk] (row-major 128 x BNT) THROUGH the caches (sc0 sc1: the XCDs' L2s are not
        // coherent with each other), wait for the write acknowledgements, then draw the tile's arrival ticket (device-scope
        // atomic) -- the protocol of conv_gemm's in-kernel split-K finish (conv.hip, EPI 4)
        const int part_bytes = 128 * BNT * 4;
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
            p.slab + ((long long)tile_id * p.n_chunks + blockIdx.x) * (128 * BNT), 0, part_bytes, 0x00020000);
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c_l = wc * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hh) + ci;
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)acc[ci][ni][r]), rsP, (c_l * BNT + n_l + ni) * 4, 0, 17);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // my part is in memory
        __syncthreads();                                              // ... and so is every wave's of this workgroup
        __shared__ unsigned s_ticket;
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != (unsigned)(p.n_chunks - 1)) return;
        if (tid == 0) __hip_atomic_store(p.tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        // last arriver: all parts -- its own included -- from memory, in chunk order (the sum does not depend on who arrives last)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ci][ni][r] = 0.f;
        for (int k = 0; k < p.n_chunks; ++k) {
            const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
                p.slab + ((long long)tile_id * p.n_chunks + k) * (128 * BNT), 0, part_bytes, 0x00020000);
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c_l = wc * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hh) + ci;
#pragma unroll
                    for (int ni = 0; ni < NIW; ++ni)
                        acc[ci][ni][r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsK, (c_l * BNT + n_l + ni) * 4, 0, 17));
                }
        }
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + wc * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hh) + ci;
            float* o = p.dW + ((long long)tap * p.Cin + c) * p.Cout + n0 + n_l;
            if (BNT == 128) { f32x2w v2; v2.x = acc[ci][0][r]; v2.y = acc[ci][NIW - 1][r]; *(f32x2w*)o = v2; }
            else *o = acc[ci][0][r];
        }
}

static int wgrad_live_taps(const ConvGeom& g, int* tap_pack) {
    const int n_in = g.axis == 0 ? g.Tin : g.Fin, n_out = g.axis == 0 ? g.Tout : g.Fout;
    int n_live = 0; *tap_pack = 0;
    for (int t = 0; t < 3; ++t) {
        bool live = false;
        for (int o = 0; o < n_out && !live; ++o) { const int i = o * g.stride - g.pad + t; live = i >= 0 && i < n_in; }
        if (live) { *tap_pack |= t << (2 * n_live); ++n_live; }
    }
    return n_live;
}

// Plan of the small-P kernel for a batch of B samples (+ 2 aux samples): tile width, row chunks, slab need.
struct WgradSmallPlan { bool ok; int bnt, chunks, rows_per_wg, lp; int64_t tiles, slab_floats; };
static WgradSmallPlan wgrad_small_plan(int64_t B, const ConvGeom& g) {
    WgradSmallPlan r{false, 128, 1, 0, 0, 0, 0};
    static const int on = []() { const char* e = getenv("NAFP_WGRAD_SMALLP"); return e ? atoi(e) : 1; }();
    const int P = g.Fout * g.Tout;
    if (!on || g.Cin % 128 != 0 || g.Cout % 128 != 0 || P >= 16 || (P & (P - 1)) != 0 || (B * P) % 16 != 0) return r;
    while ((1 << r.lp) < P) ++r.lp;
    int tp; const int n_live = wgrad_live_taps(g, &tp);
    if (n_live == 0) return r;
    const int64_t M = (B + 2) * P, steps = (M + 15) / 16;
    // Tile width and row chunks by a makespan model: W = tiles x chunks workgroups go round-robin over 256 CUs, a CU works
    // through ceil(W / 256) of them (co-resident workgroups share its SIMDs), each steps / chunks K-steps long plus ~`ov`
    // K-steps of prologue / epilogue; a 64-column tile's K-step costs half a 128-column one's; a CU holding a single
    // workgroup (one wave per SIMD) leaves the LDS round trip of every K-step exposed (`eff1`).  Sweep knobs:
    // NAFP_WGRAD_SP_OV, NAFP_WGRAD_SP_EFF1 (per cent), NAFP_WGRAD_SP_FORCE="bnt:chunks".
    static const double ov = []() { const char* e = getenv("NAFP_WGRAD_SP_OV"); return e ? atof(e) : 5.0; }();
    static const double eff1 = []() { const char* e = getenv("NAFP_WGRAD_SP_EFF1"); return e ? atof(e) / 100.0 : 0.8; }();
    struct Force { int bnt, chunks; };
    static const Force force = []() {
        Force f{0, 0}; const char* e = getenv("NAFP_WGRAD_SP_FORCE");
        if (e) sscanf(e, "%d:%d", &f.bnt, &f.chunks);
        return f;
    }();
    double best = 1e30;
    for (int bnt = 64; bnt <= 128; bnt += 64) {
        if (force.bnt && bnt != force.bnt) continue;
        const int64_t tiles = (int64_t)(g.Cout / bnt) * (n_live * g.Cin / 128);
        if (tiles > NAFP_TICKET_SLOTS) continue;
        for (int64_t c = 1; c <= std::max<int64_t>(1, steps / 8) && c <= 64; ++c) {
            if (force.chunks && c != std::min<int64_t>(force.chunks, std::max<int64_t>(1, steps / 8))) continue;
            const int64_t spw = (steps + c - 1) / c, cc = (steps + spw - 1) / spw;      // chunks actually used
            const int64_t per_cu = (tiles * cc + 255) / 256;
            double cost = (double)per_cu * ((double)spw + ov) * (bnt == 64 ? 0.5 : 1.0) / (per_cu == 1 ? eff1 : 1.0);
            if (cc > 1) cost += 1.0;                                                   // slab write + last-arriver pass
            if (cost < best - 1e-9) { best = cost; r.bnt = bnt; r.chunks = (int)cc; r.rows_per_wg = (int)(spw * 16); r.tiles = tiles; }
        }
    }
    if (best >= 1e30) return r;
    r.slab_floats = r.chunks > 1 ? r.tiles * r.chunks * 128 * r.bnt : 0;
    r.ok = true;
    return r;
}

int64_t wgrad_slab_floats(int64_t B, const ConvGeom& g) {
    const WgradSmallPlan pl = wgrad_small_plan(B, g);
    if (pl.ok) return pl.slab_floats;
    // fast kernel: <= 768 workgroups (+ the rounding of the chunk count) x one 128 x 128 partial tile each
    int tp; const int n_live = wgrad_live_taps(g, &tp);
    const int64_t col_tiles = (int64_t)(g.Cout / 128) * (n_live * g.Cin / 128);
    return col_tiles == 0 ? 0 : (1536 + 2 * col_tiles) * (int64_t)(128 * 128);      // (room for NAFP_WGRAD_WGS up to 1536)
}

// dW (keras (3,Cin,Cout)) of one layer.  X2 / D2 (both or neither): the two aux samples [gamma | beta] resp. [S1 | S2],
// each pair adjacent in memory, folded into the main launch where the shape allows (else two more launches, as before).
// slab / tickets (or null): workspace of the small-P kernel (null: it is not used).  sj: optional side job (see ScalarsJob).
int launch_wgrad(const float* X, const float* D, float* dW, int64_t B, const ConvGeom& g, hipStream_t st,
                 const float* X2, const float* D2, float* slab, int64_t slab_floats, unsigned* tickets, const ScalarsJob* sj, int prec) {
    if (g.Cin % 128 != 0 || g.Cout % 128 != 0) return NAFP_ERR_UNSUPPORTED;
    WgradParams p;
    p.X = X; p.D = D; p.dW = dW; p.B = (int)B; p.P = g.Fout * g.Tout; p.Tout = g.Tout;
    p.Fin = g.Fin; p.Tin = g.Tin; p.Cin = g.Cin; p.Cout = g.Cout; p.axis = g.axis; p.stride = g.stride; p.pad = g.pad;
    p.sample_in = (long long)g.Fin * g.Tin * g.Cin;
    p.X2 = X; p.D2 = D; p.n_aux = 0; p.M_main = (long long)B * p.P;
    p.slab = nullptr; p.tickets = nullptr; p.n_chunks = 1;
    p.sj = sj ? *sj : ScalarsJob{nullptr, nullptr, nullptr, nullptr, 0, 0.0};
    p.n_live = wgrad_live_taps(g, &p.tap_pack);
    if (p.n_live == 0) {                                   // nothing to add; the side job still has to run
        if (sj && sj->sc) {
            ln_bwd_scalars_kernel<<<(unsigned)((sj->B + 255) / 256), 256, 0, st>>>(sj->mr, sj->lnsum, sj->mr_prev, sj->sc, sj->B, sj->inv_n);
            NAFP_LAUNCH_CHECK();
        }
        return NAFP_OK;
    }
    static bool attr = false;
    const int lds = 3 * 2 * 16 * 128 * (int)sizeof(float);
    if (!attr) {
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_fast_bf16x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_smallp_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_smallp_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_smallp_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        NAFP_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_smallp_kernel<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    const bool have_aux = X2 && D2;
    // ---- small layers (P < 16) ----
    const WgradSmallPlan sp = wgrad_small_plan(B, g);
    if (sp.ok && have_aux && (sp.chunks == 1 || (slab && tickets && sp.slab_floats <= slab_floats))) {
        p.X2 = X2; p.D2 = D2; p.n_aux = 2;
        p.rows_per_wg = sp.rows_per_wg; p.n_chunks = sp.chunks; p.slab = slab; p.tickets = tickets;
        const dim3 grid((unsigned)sp.chunks, (unsigned)(g.Cout / sp.bnt), (unsigned)(p.n_live * g.Cin / 128));
        static const bool x6_small = []() { const char* e = getenv("NAFP_X6_WGRAD_SMALL"); return !e || e[0] != '0'; }();      // (A/B knob of the split arithmetic on the small layers)
        if (prec == 2 && x6_small) {
            if (sp.bnt == 64) wgrad_smallp_kernel<64, 2><<<grid, 256, 3 * (16 * 128 + 16 * 64) * sizeof(float), st>>>(p, sp.lp);
            else wgrad_smallp_kernel<128, 2><<<grid, 256, lds, st>>>(p, sp.lp);
        } else if (sp.bnt == 64) wgrad_smallp_kernel<64><<<grid, 256, 3 * (16 * 128 + 16 * 64) * sizeof(float), st>>>(p, sp.lp);
        else wgrad_smallp_kernel<128><<<grid, 256, lds, st>>>(p, sp.lp);
        NAFP_LAUNCH_CHECK();
        return NAFP_OK;
    }
    const int col_tiles = (g.Cout / 128) * (p.n_live * g.Cin / 128);
    // regular shapes take the kernel whose K-steps are (nearly) free of vector instructions
    auto log2_exact = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    const int lt = log2_exact(g.Tout), lfo = log2_exact(g.Fout);
    static const bool fast_on = []() { const char* e = getenv("NAFP_WGRAD_FAST"); return !e || e[0] != '0'; }();
    const bool fast = fast_on && lt >= 0 && lt <= 4 && lfo >= 0 && p.P % 16 == 0 && (g.stride == 1 || g.stride == 2) &&
                      (g.axis == 0 ? (g.Fin == g.Fout) : (g.Fin == g.stride * g.Fout && g.Tin == g.Tout));
    static const bool aux_fold = []() { const char* e = getenv("NAFP_WGRAD_AUXFOLD"); return !e || e[0] != '0'; }();
    // aux rows ride in the fast kernel when every K-step stays on one side of row B * P (a K-step takes 16 rows, 32 for the
    // half-dead taps of the two-frame layers)
    // (16-row K-steps everywhere except the convs along T with two output frames, whose half-dead taps take 32: wgrad_fast_kernel's to_sel)
    const int rows_step = (g.axis == 0 && g.Tout == 2) ? 32 : 16;
    const bool fold = fast && have_aux && aux_fold && p.P % rows_step == 0;
    auto launch_one = [&](const float* Xq, const float* Dq, int64_t Bq, bool with_aux, const ScalarsJob* job, bool is_main) -> int {
        WgradParams q = p;
        q.X = Xq; q.D = Dq; q.B = (int)Bq; q.M_main = (long long)Bq * q.P;
        q.n_aux = with_aux ? 2 : 0; q.X2 = with_aux ? X2 : Xq; q.D2 = with_aux ? D2 : Dq;
        q.sj = job ? *job : ScalarsJob{nullptr, nullptr, nullptr, nullptr, 0, 0.0};
        const long long M = (long long)(Bq + q.n_aux) * q.P;
        // one resident round: 256 CUs x 3 workgroups (each workgroup ends in 16 K atomics, so fewer is better;
        // measured at B = 1280: 768 / 1536 / 3072 / 6144 workgroups -> backward 22.3 / 22.5 / 22.9 / 24.3 ms)
        static const long long wg_target = []() { const char* e = getenv("NAFP_WGRAD_WGS"); return e ? atoll(e) : 768ll; }();
        long long chunks = std::max<long long>(1, wg_target / col_tiles);
        long long rpw = (M + chunks - 1) / chunks;
        rpw = std::max<long long>(64, (rpw + 31) / 32 * 32);
        // the X descriptor of a workgroup spans (rows/P + 2) samples: keep it below 2 GiB
        while (rpw > 64 && (rpw / q.P + 2) * q.sample_in * 4 >= (1ll << 31)) rpw = std::max<long long>(64, rpw / 2 / 32 * 32);
        q.rows_per_wg = (int)rpw;
        const unsigned gx = (unsigned)((M + rpw - 1) / rpw);
        // the main launch of the fast kernel leaves its partial tiles in the slab; wgrad_reduce_kernel STORES their sum into dW
        // (is_main: the separate rank-one launches of the non-fold fallback accumulate into that sum by atomics -- also at B = 1)
        static const bool slab_on = []() { const char* e = getenv("NAFP_WGRAD_SLAB"); return !e || e[0] != '0'; }();
        const bool use_slab = slab_on && fast && is_main && slab && (int64_t)col_tiles * gx * 128 * 128 <= slab_floats;
        q.slab = use_slab ? slab : nullptr; q.n_chunks = (int)gx;
        if (fast && prec == 2) wgrad_fast_bf16x6_kernel<<<dim3(gx, g.Cout / 128, q.n_live * g.Cin / 128), 256, lds, st>>>(q, lt, lfo);
        else if (fast) wgrad_fast_kernel<<<dim3(gx, g.Cout / 128, q.n_live * g.Cin / 128), 256, lds, st>>>(q, lt, lfo);
        else wgrad_kernel<<<dim3(gx, g.Cout / 128, q.n_live * g.Cin / 128), 256, lds, st>>>(q);
        NAFP_LAUNCH_CHECK();
        if (use_slab) {
            wgrad_reduce_kernel<<<dim3(WGRAD_REDUCE_SLICES, g.Cout / 128, q.n_live * g.Cin / 128), 256, 0, st>>>(slab, dW, (int)gx, g.Cin, g.Cout, q.tap_pack);
            NAFP_LAUNCH_CHECK();
        }
        return NAFP_OK;
    };
    int rc = launch_one(X, D, B, fold, sj, true);
    if (rc != NAFP_OK || !have_aux || fold) return rc;
    rc = launch_one(X2, D2, 1, false, nullptr, false);                                   // gamma_{j-1} against S1_j
    if (rc != NAFP_OK) return rc;
    return launch_one(X2 + p.sample_in, D2 + (long long)p.P * p.Cout, 1, false, nullptr, false);   // beta_{j-1} against S2_j
}

// ============================================================================
// conv0 backward: dW0[k][c] = sum feat[b, f, t*s - p + k] * dt0[b,f,t,c], dbias0[c] = sum dt0.
// ============================================================================
__global__ __launch_bounds__(256) void conv0_bwd_kernel(
        const float* __restrict__ feat, const float* __restrict__ dt, float* __restrict__ dW0,
        float* __restrict__ dbias0, int F, int Tin, int Tout, int Cout, int stride, int pad, int rows) {
    const int tid = threadIdx.x;
    const int blocks_per_sample = (F + rows - 1) / rows;
    const int64_t b = blockIdx.x / blocks_per_sample;
    const int f0 = (blockIdx.x % blocks_per_sample) * rows;
    const int cgroups = Cout / 4, pos_per_iter = 256 / cgroups;
    const int cg = tid % cgroups, pslot = tid / cgroups;
    const int nrows = min(rows, F - f0), npos = nrows * Tout;
    const float* xin = feat + (b * F + f0) * (int64_t)Tin;
    const float* din = dt + ((b * F + f0) * (int64_t)Tout) * Cout;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, ab = a0;
    for (int pp = pslot; pp < npos; pp += pos_per_iter) {
        const int r = pp / Tout, to = pp % Tout;
        const int t0 = to * stride - pad;
        const float* xr = xin + r * Tin;
        const float x0 = (t0 >= 0 && t0 < Tin) ? xr[t0] : 0.f;
        const float x1 = (t0 + 1 >= 0 && t0 + 1 < Tin) ? xr[t0 + 1] : 0.f;
        const float x2 = (t0 + 2 >= 0 && t0 + 2 < Tin) ? xr[t0 + 2] : 0.f;
        const float4 d = *(const float4*)(din + (int64_t)pp * Cout + 4 * cg);
        a0.x += x0 * d.x; a0.y += x0 * d.y; a0.z += x0 * d.z; a0.w += x0 * d.w;
        a1.x += x1 * d.x; a1.y += x1 * d.y; a1.z += x1 * d.z; a1.w += x1 * d.w;
        a2.x += x2 * d.x; a2.y += x2 * d.y; a2.z += x2 * d.z; a2.w += x2 * d.w;
        ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
    }
    __shared__ float4 red[4][256];
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2; red[3][tid] = ab;
    __syncthreads();
    if (tid < cgroups) {
        for (int k = 0; k < 4; ++k) {
            float4 t = red[k][tid];
            for (int r = 1; r < pos_per_iter; ++r) {
                const float4 u = red[k][tid + r * cgroups];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            if (k == 3 && !dbias0) continue;
            float* o = (k < 3 ? dW0 + k * Cout : dbias0) + 4 * tid;
            atomicAdd(o, t.x); atomicAdd(o + 1, t.y); atomicAdd(o + 2, t.z); atomicAdd(o + 3, t.w);
        }
    }
}

// ============================================================================
// tail backward: L2 normalisation, divide-and-encode, flatten (nnfp.py:141-156, 224-231).
//   kernel A (one workgroup per segment, thread <-> slice q): dy, then dxhat of the last conv;
//   kernel B (workgroup = slice x batch chunk, thread <-> (i, j)): batch-reduced weight gradients.
// ============================================================================
template <int S>
__global__ __launch_bounds__(256) void tail_bwd_a_kernel(const TailBwdArgs a) {
    const int q = threadIdx.x, Q = a.Q;
    const int64_t b = blockIdx.x;
    float lnA, lnC;
    stat_ln_scalars(a.stats + 2 * b, a.ident_stats ? -1.0 : 1.0 / (double)a.D, &lnA, &lnC);
    if (q == 0) { a.ln[2 * b] = lnA; a.ln[2 * b + 1] = lnC; }     // kernel B reads them instead of redoing the double arithmetic per thread
    float x[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const int d = q * S + i;
        x[i] = fmaf(lnA, a.z[b * a.D + d], fmaf(lnC, a.gamma[d], a.beta[d]));
    }
    // thread <-> slice q: the packed layouts (S,32,Q) / (32,Q) put the Q slices of one (i, j) side by side, so every load is one
    // 256-B row per wave (the keras layout (Q,S,32) touched 64 cache lines per load).  The hidden pre-activation h_j is formed
    // twice -- for y, then again for the gradient -- instead of kept: 32 live values per thread plus the weights the compiler then
    // hoists spilled to scratch.
    const float* w1q = a.w1p + q; const float* b1q = a.b1p + q; const float* w2q = a.w2p + q;
    float y = a.b2[q];
#pragma unroll 4
    for (int j = 0; j < 32; ++j) {
        float h = b1q[j * Q];
#pragma unroll
        for (int i = 0; i < S; ++i) h = fmaf(x[i], w1q[(i * 32 + j) * Q], h);
        y = fmaf(elu1(h), w2q[j * Q], y);
    }
    float dyq = a.d_emb[b * Q + q];
    if (a.l2norm) {
        __shared__ float red[2][8];
        const float ss = wave_sum(y * y), sd = wave_sum(y * dyq);
        if ((q & 63) == 0) { red[0][q >> 6] = ss; red[1][q >> 6] = sd; }
        __syncthreads();
        float n2 = 0.f, yd = 0.f;
        for (int w = 0; w < (Q + 63) / 64; ++w) { n2 += red[0][w]; yd += red[1][w]; }
        // e = y * rn, rn = rsqrt(max(n2, eps)); d/dy: rn*de - (n2 > eps) * y * rn^3 * (y.de)
        const float rn = rsqrtf(fmaxf(n2, 1e-12f));
        dyq = rn * dyq - (n2 > 1e-12f ? y * rn * rn * rn * yd : 0.f);
    }
    a.dy[b * Q + q] = dyq;
    float dx[S];
#pragma unroll
    for (int i = 0; i < S; ++i) dx[i] = 0.f;
#pragma unroll 4
    for (int j = 0; j < 32; ++j) {
        float w[S];
        float h = b1q[j * Q];
#pragma unroll
        for (int i = 0; i < S; ++i) { w[i] = w1q[(i * 32 + j) * Q]; h = fmaf(x[i], w[i], h); }
        const float da = dyq * w2q[j * Q] * (h > 0.f ? 1.f : __expf(h));
#pragma unroll
        for (int i = 0; i < S; ++i) dx[i] = fmaf(w[i], da, dx[i]);
    }
#pragma unroll
    for (int i = 0; i < S; ++i) a.dxh[b * a.D + q * S + i] = lnA * dx[i];      // r_b * dL/dxhat: ln_bwd_fused's convention
}

template <int S>
__global__ __launch_bounds__(256) void tail_bwd_b_kernel(const TailBwdArgs a, int64_t B) {
    const int q = blockIdx.x;
    const int j = threadIdx.x & 31, i = threadIdx.x >> 5;         // 32 x 8 threads; row i computes dw1[q][i + 8t][j]
    constexpr int NI = (S + 7) / 8;
    constexpr int SB = 128;                                       // samples staged at a time
    const int64_t per = (B + gridDim.y - 1) / gridDim.y;
    const int64_t bb0 = blockIdx.y * per, bb1 = std::min<int64_t>(B, bb0 + per);
    // The slice's inputs of SB samples at a time go through LDS: x (the LayerNorm output of the slice's S flatten elements) and
    // dL/dy.  Every thread of the workgroup needs every sample's S + 1 numbers; loading them per sample inside the loop was one
    // dependent memory round trip per sample (40 of them at a batch of 640: the whole kernel).
    __shared__ float s_x[SB][S + 1];
    float aw1[NI], ab1 = 0.f, aw2 = 0.f, ab2 = 0.f;
#pragma unroll
    for (int t = 0; t < NI; ++t) aw1[t] = 0.f;
    const float b1v = a.b1[q * 32 + j], w2v = a.w2[q * 32 + j];
    float w1v[S];
#pragma unroll
    for (int k = 0; k < S; ++k) w1v[k] = a.w1[(q * S + k) * 32 + j];
    for (int64_t s0 = bb0; s0 < bb1; s0 += SB) {
        const int ns = (int)std::min<int64_t>(SB, bb1 - s0);
        __syncthreads();
        for (int e = threadIdx.x; e < ns * (S + 1); e += 256) {
            const int sb = e / (S + 1), k = e - sb * (S + 1);
            const int64_t b = s0 + sb;
            float v;
            if (k < S) {
                const int d = q * S + k;
                v = fmaf(a.ln[2 * b], a.z[b * a.D + d], fmaf(a.ln[2 * b + 1], a.gamma[d], a.beta[d]));
            } else {
                v = a.dy[b * a.Q + q];
            }
            s_x[sb][k] = v;
        }
        __syncthreads();
        for (int sb = 0; sb < ns; ++sb) {
            float x[S];
#pragma unroll
            for (int k = 0; k < S; ++k) x[k] = s_x[sb][k];
            const float dyq = s_x[sb][S];
            float h = b1v;
#pragma unroll
            for (int k = 0; k < S; ++k) h = fmaf(x[k], w1v[k], h);
            const float da = dyq * w2v * (h > 0.f ? 1.f : __expf(h));
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                float xi = 0.f;
#pragma unroll
                for (int k = 0; k < S; ++k) xi = (k == i + 8 * t) ? x[k] : xi;
                aw1[t] += xi * da;
            }
            ab1 += da;
            aw2 += dyq * elu1(h);
            ab2 += dyq;
        }
    }
    // the batch chunks (blockIdx.y) meet through atomics; the gradients are zeroed by the caller
#pragma unroll
    for (int t = 0; t < NI; ++t)
        if (i + 8 * t < S) atomicAdd(a.dw1 + (q * S + i + 8 * t) * 32 + j, aw1[t]);
    if (i == 0) { atomicAdd(a.db1 + q * 32 + j, ab1); atomicAdd(a.dw2 + q * 32 + j, aw2); }
    if (threadIdx.x == 0) atomicAdd(a.db2 + q, ab2);
}

template <int S>
static int launch_tail_bwd_s(const TailBwdArgs& a, int64_t B, hipStream_t st) {
    tail_bwd_a_kernel<S><<<dim3((unsigned)B), a.Q, 0, st>>>(a);
    NAFP_LAUNCH_CHECK();
    const unsigned chunks = (unsigned)std::min<int64_t>(16, std::max<int64_t>(1, B / 16));
    tail_bwd_b_kernel<S><<<dim3((unsigned)a.Q, chunks), 256, 0, st>>>(a, B);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

int launch_tail_bwd(const TailBwdArgs& a, int64_t B, hipStream_t st) {
    if (a.Q % 64 != 0 || a.Q > 256 || a.S * a.Q != a.D) return NAFP_ERR_UNSUPPORTED;
    switch (a.S) {                       // emb_sz 256 / 128 / 64 on the 1024-wide flatten
        case 4: return launch_tail_bwd_s<4>(a, B, st);
        case 8: return launch_tail_bwd_s<8>(a, B, st);
        case 16: return launch_tail_bwd_s<16>(a, B, st);
        default: return NAFP_ERR_UNSUPPORTED;
    }
}

// keras kernel (3, Cin, Cout) -> dgrad operand (Cin, 3*Cout): Wd[c][k*Cout + n] = W[k][c][n]
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ k3, float* __restrict__ wd, int Cin, int Cout) {
    const int64_t total = (int64_t)3 * Cin * Cout;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i % Cout);
        const int64_t r = i / Cout;
        const int c = (int)(r % Cin), k = (int)(r / Cin);
        wd[((int64_t)c * 3 + k) * Cout + n] = k3[i];
    }
}

// ---- host launch helpers used by api.hip -------------------------------------------------
int launch_stats_to_mr(const stat_t* stats, float* mr, const double* inv_n_dev, int64_t B, int n_layers, hipStream_t st, bool identity) {
    const int64_t n = (int64_t)n_layers * B;
    stats_to_mr_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(stats, mr, n, inv_n_dev, B, identity ? 1 : 0);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

int launch_ln_bwd(float* d, const float* tpre, const float* gamma, const float* mr, const float* mr_prev,
                  double* lnsum, float* sc, float* dgamma, float* dbeta, float* dbias, float* S1, float* S2,
                  int64_t B, int P, int C, hipStream_t st, bool reduce_here, const float* Gj, const float* Hbj,
                  double* lnsum_below, const float* feat0, const float* w0, const float* bias0, const ConvGeom* g0, float* dW0,
                  bool scalars_done, float* part_slab, int64_t part_slab_floats, unsigned* tickets, bool tpre_is_z) {
    const int64_t n = (int64_t)P * C;
    if (C % 4 != 0 || (256 % (C / 4) != 0 && (C / 4) % 256 != 0) || n % 1024 != 0 || 1024 % C != 0) return NAFP_ERR_UNSUPPORTED;
    if (reduce_here) {
        // top layer: its gradient comes from the divide-and-encode tail, not from a transposed conv
        const int chunks = (int)std::min<int64_t>(std::max<int64_t>(1, n / 4 / 2048), 64);
        if (tpre_is_z) ln_bwd_reduce_kernel<true><<<dim3(chunks, (unsigned)B), 256, 0, st>>>(d, tpre, gamma, mr, lnsum, n);
        else ln_bwd_reduce_kernel<false><<<dim3(chunks, (unsigned)B), 256, 0, st>>>(d, tpre, gamma, mr, lnsum, n);
        NAFP_LAUNCH_CHECK();
    }
    if (!scalars_done) {                  // (else: written by the side job of the wgrad launch of the layer above)
        ln_bwd_scalars_kernel<<<(unsigned)((B + 255) / 256), 256, 0, st>>>(mr, lnsum, mr_prev, sc, B, 1.0 / (double)n);
        NAFP_LAUNCH_CHECK();
    }
    // batch chunks: every chunk ends in 5 atomics per element (~33 G atomics/s measured), which is what a
    // small layer pays for; 256..1024 workgroups keep the streaming layers at HBM speed
    const int64_t bx = n / 1024;
    // (B >= 4096: the chunks are >= 64 samples long whatever their number, and layers 2-5 at 256-512 workgroups ran one wave per
    // SIMD: 2048 measured 86.5 -> 86.2 ms per step at B = 5120; at B = 640 / 1280 more chunks cost more than they gain)
    static const int64_t wg_env = []() { const char* e = getenv("NAFP_LNB_WGS"); return e ? atoll(e) : (int64_t)0; }();
    const int64_t wg_target = wg_env > 0 ? wg_env : (B >= 4096 ? 2048 : 256);
    // at least 4 batch chunks -- 2 for the two largest layers at a small batch (>= 512 workgroups either way; measured at
    // B = 640: layer 0 377 -> 321 us, layer 1 422 -> 402; at B = 5120 the halved chunk count costs layer 0 4 %): NAFP_LNB_MINBY overrides
    static const int64_t min_by_env = []() { const char* e = getenv("NAFP_LNB_MINBY"); return e ? atoll(e) : (int64_t)0; }();
    const int64_t min_by = min_by_env > 0 ? min_by_env : ((bx >= 256 && B <= 1024) ? 2 : 4);
    // at most 64 chunks, and at least ~20 samples per chunk: every chunk ends in 5 atomics per element onto the same addresses,
    // which is what the small layers' launches consist of (B = 640: 64 -> 32 chunks takes layers 10-15 from 221 to 174 us;
    // at B = 5120 fewer chunks than 64 only lengthen the sample loops).  NAFP_LNB_MAXBY overrides.
    static const int64_t max_by_env = []() { const char* e = getenv("NAFP_LNB_MAXBY"); return e ? atoll(e) : (int64_t)0; }();
    const int64_t max_by = max_by_env > 0 ? max_by_env : std::min<int64_t>(64, std::max<int64_t>(8, B / 20));
    int by = (int)std::min<int64_t>(B, std::min<int64_t>(max_by, std::max<int64_t>(min_by, wg_target / bx)));
    while (lnsum_below && (B + by - 1) / by * 8 > 32768 && by < B) by *= 2;      // LDS share of the sums below: 8 B per sample
    const size_t lds = lnsum_below ? (size_t)((B + by - 1) / by) * 2 * sizeof(float) : 0;
    // slab + last-arriver in place of the per-element atomics (needs whole 1024-element blocks: n % 1024 == 0, checked above)
    static const bool slab_on = []() { const char* e = getenv("NAFP_LNB_SLAB"); return !e || e[0] != '0'; }();
    // (taken for the layers with >= 8 blocks of 1024 elements: there it is faster -- B = 640: layers 0-9 lose 5-25 us each --; on the
    // small layers the 1-4 last arrivers add 32 chunks each on their own, +8-12 us per launch, so those keep the atomics)
    float* const slab_in = slab_on ? part_slab : nullptr; unsigned* const tickets_in = tickets;     // (the wave-split variant below sizes its own layout)
    if (!(slab_on && bx >= 8 && part_slab && tickets && bx <= NAFP_TICKET_SLOTS && bx * by * 4 * 1024 <= part_slab_floats &&
          bx * by * 4 * 1024 * 4 < ((int64_t)1 << 31))) { part_slab = nullptr; tickets = nullptr; }
    Conv0Regen c0{};
    // the small layers: four waves of a workgroup share its batch chunk (ln_bwd_fused_kernel<.., WAVE>); NAFP_LNB_WAVE_MAXN=0 turns it off
    static const int64_t wave_maxn = []() { const char* e = getenv("NAFP_LNB_WAVE_MAXN"); return e ? atoll(e) : (int64_t)4096; }();
    // samples per wave: 10 for the small layers, 20 for the slab variant (B = 640, us per layer with 10 / 20 / 40: layers 15-10
    // 14-27 / 19-52 / 30-86; layers 8, 6, 5 with 10 / 20: 36 38 67 / 29 34 59).  NAFP_LNB_WAVE_SPW overrides both.
    static const int64_t wave_spw_env = []() { const char* e = getenv("NAFP_LNB_WAVE_SPW"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)0; }();
    static const int64_t wslab_maxn = []() { const char* e = getenv("NAFP_LNB_WSLAB_MAXN"); return e ? atoll(e) : (int64_t)65536; }();
    const int64_t wave_spw = wave_spw_env > 0 ? wave_spw_env : (n <= wave_maxn ? 10 : 20);
    // ... and, through a slab instead of atomics, the larger layers at a small batch (NAFP_LNB_WSLAB_MAXB, default 2048; 0 = off)
    static const int64_t wslab_maxb = []() { const char* e = getenv("NAFP_LNB_WSLAB_MAXB"); return e ? atoll(e) : (int64_t)2048; }();
    if (!feat0 && tpre_is_z && S1 && S2 && n % 256 == 0) {
        int byw = (int)std::max<int64_t>(1, std::min<int64_t>(64, B / (4 * wave_spw)));
        while (lnsum_below && (B + byw - 1) / byw * 8 > 32768 && byw < B) byw *= 2;
        const size_t ldsw = lnsum_below ? (size_t)((B + byw - 1) / byw) * 2 * sizeof(float) : 0;
        const int64_t nblk = n / 256;
        const bool small = n <= wave_maxn && C % 256 == 0;
        const bool wslab = !small && n > wave_maxn && n <= wslab_maxn && B <= wslab_maxb && (C == 128 || C % 256 == 0) && slab_in && tickets_in &&
                           nblk <= NAFP_TICKET_SLOTS && nblk * byw * 1024 <= part_slab_floats && (int64_t)byw * 4096 < ((int64_t)1 << 31);
        if (small || wslab) {
            ln_bwd_fused_kernel<false, true, true><<<dim3((unsigned)nblk, byw), 256, ldsw, st>>>(d, tpre, gamma, sc, dgamma, dbeta, dbias, S1, S2, n, B,
                                                                                           C, Gj, Hbj, lnsum_below, c0, wslab ? slab_in : nullptr,
                                                                                           wslab ? tickets_in : nullptr);
            NAFP_LAUNCH_CHECK();
            return NAFP_OK;
        }
    }
    if (feat0) {
        if (!g0 || g0->Cin != 1 || g0->axis != 0 || g0->Cout != C || g0->Fout * g0->Tout != P || reduce_here) return NAFP_ERR_INVALID_ARG;
        if (dW0 && C / 4 > 256) return NAFP_ERR_UNSUPPORTED;
        c0.feat = feat0; c0.w3 = w0; c0.bias = bias0; c0.dW0 = dW0; c0.F = g0->Fin; c0.Tin = g0->Tin; c0.Tout = g0->Tout;
        c0.stride = g0->stride; c0.pad = g0->pad;
        ln_bwd_fused_kernel<true><<<dim3((unsigned)bx, by), 256, lds, st>>>(d, nullptr, gamma, sc, dgamma, dbeta, dbias, S1, S2, n, B,
                                                                          C, Gj, Hbj, lnsum_below, c0, part_slab, tickets);
    } else if (tpre_is_z) {
        ln_bwd_fused_kernel<false, true><<<dim3((unsigned)bx, by), 256, lds, st>>>(d, tpre, gamma, sc, dgamma, dbeta, dbias, S1, S2, n, B,
                                                                                 C, Gj, Hbj, lnsum_below, c0, part_slab, tickets);
    } else {
        ln_bwd_fused_kernel<false><<<dim3((unsigned)bx, by), 256, lds, st>>>(d, tpre, gamma, sc, dgamma, dbeta, dbias, S1, S2, n, B,
                                                                           C, Gj, Hbj, lnsum_below, c0, part_slab, tickets);
    }
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

int launch_ln_bwd_scalars(const float* mr, const double* lnsum, const float* mr_prev, float* sc, int64_t B, int64_t n, hipStream_t st) {
    ln_bwd_scalars_kernel<<<(unsigned)((B + 255) / 256), 256, 0, st>>>(mr, lnsum, mr_prev, sc, B, 1.0 / (double)n);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

int launch_conv0_bwd(const float* feat, const float* dt, float* dW0, float* dbias0, int64_t B, const ConvGeom& g,
                     hipStream_t st) {
    if (g.Cin != 1 || g.axis != 0 || (g.Cout % 4) != 0 || 256 % (g.Cout / 4) != 0) return NAFP_ERR_UNSUPPORTED;
    // 64 rows per workgroup: every workgroup ends in 512 atomics onto the same 512 addresses (dW0, dbias0), so
    // fewer, longer workgroups win (measured at B = 1280: rows 16/32/64/128/256 -> backward 24.4/23.5/22.7/23.0/22.8 ms)
    const int rows = 64;
    const int64_t blocks = B * ((g.Fin + rows - 1) / rows);
    conv0_bwd_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(feat, dt, dW0, dbias0, g.Fin, g.Tin, g.Tout, g.Cout,
                                                            g.stride, g.pad, rows);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

int launch_pack_dgrad_weight(const float* k3, float* wd, int Cin, int Cout, hipStream_t st) {
    pack_dgrad_weight_kernel<<<256, 256, 0, st>>>(k3, wd, Cin, Cout);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

}  // namespace nafp
