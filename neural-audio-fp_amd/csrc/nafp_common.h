// Shared declarations for libnafp (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "../../include/nafp.h"

namespace nafp {

extern thread_local int g_last_hip_error;

#define NAFP_HIP_CHECK(expr)                                   \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) {                                \
            nafp::g_last_hip_error = (int)_e;                  \
            return NAFP_ERR_HIP;                               \
        }                                                      \
    } while (0)

#define NAFP_LAUNCH_CHECK() NAFP_HIP_CHECK(hipGetLastError())

constexpr float LN_EPS = 1e-3f;   // keras LayerNormalization default (nnfp.py:66-67)

// TF 'SAME' geometry along one axis.
struct SamePad { int n_out, before, after; };
inline SamePad same_pad(int n_in, int k, int s) {
    int n_out = (n_in + s - 1) / s;
    int total = (n_out - 1) * s + k - n_in;
    if (total < 0) total = 0;
    return {n_out, total / 2, total - total / 2};
}

// One conv of the 16 in front_conv (nnfp.py:48-59): 3 taps along T (axis 0, "1x3")
// or along F (axis 1, "3x1"), dense over channels, TF SAME padding.
struct ConvGeom {
    int axis;            // 0: taps along T, 1: taps along F
    int Fin, Tin, Cin;
    int Fout, Tout, Cout;
    int stride;          // along the tap axis (the other axis has stride 1)
    int pad;             // pad before, along the tap axis
};

std::vector<ConvGeom> encoder_geometry(int in_f, int in_t);

// keras ELU(alpha=1) (nnfp.py:74,77).  TF computes exp(x) - 1 for x < 0; branch-free here
// (v_exp_f32 + select) because it sits in every conv epilogue.  A NaN stays a NaN (fminf would hand back the 0).
__device__ __forceinline__ float elu1(float v) {
    const float e = __expf(fminf(v, 0.f)) - 1.f;
    return v > 0.f ? v : (v != v ? v : e);
}

// max(v, lo) that keeps a NaN (fmaxf hands back `lo`): the -80 dB clamp of the log-mel layer (melspectrogram.py:109) must not
// turn a NaN segment into a finite one.  Bit-identical to fmaxf for every other input.
__device__ __forceinline__ float max_keep_nan(float v, float lo) { return v < lo ? lo : v; }

// Per-sample LayerNorm statistics (sum, sum of squares of a conv's ELU output) are accumulated as 64-bit FIXED-POINT
// integers with STAT_FRAC_BITS fractional bits.  Integer addition is associative, so the totals do not depend on the
// order in which tiles, workgroups or split-K finishers arrive: the forward pass -- and with it every fingerprint -- is
// BIT-REPRODUCIBLE run to run (the double atomicAdd used before made two generate runs differ by ~1e-6).  Each partial is
// a double formed in a fixed order inside its workgroup and rounded once to the grid: <= 2^-21 absolute per partial
// (at most a few thousand partials per sample), against sums that are divided by n >= 1024 and compared with the
// LayerNorm epsilon 1e-3 -- far below float32 rounding of the activations themselves.
//
// NON-FINITE VALUES AND RANGE [r5].  A partial that is NaN, infinite or >= 2^34 in magnitude (a workgroup's share of one
// sample -- at most 32 K elements, conv0's: an RMS activation above ~700; LayerNorm keeps activations O(1)) is not added: it
// POISONS the sample instead -- atomicOr of STAT_POISON into the sample's sum-of-squares slot.  That slot only ever receives
// non-negative partials below 2^54 (fixed point), at most 2^7 of them per sample and layer (conv1 of a 2-s input: 128 position
// blocks), so its two top bits are clear unless poisoned, an OR is idempotent, and later adds cannot carry into them: the
// flag is sticky and order-free.  stat_get() of a poisoned slot
// is NaN, hence r_b = c_b = NaN for that sample in every consumer; and because the packed ELU of the GEMM epilogues
// (max(t, exp(min(t, 0)) - 1): IEEE maxNum drops a NaN operand) would wash a NaN row back to finite values one layer
// later, every consumer hands the poison of its input statistics on to its output statistics (stat_forward_poison: one
// thread per sample and workgroup, in the prologue).  The tail writes NaN rows for poisoned samples: what keras
// LayerNormalization does with such a sample (nnfp.py:73-79), while the other samples of the launch are untouched.
typedef long long stat_t;
constexpr int STAT_FRAC_BITS = 20;
constexpr double STAT_PARTIAL_LIMIT = 17179869184.0;                // 2^34
constexpr unsigned long long STAT_POISON = 1ull << 62;
#ifdef __HIPCC__
__device__ __forceinline__ void stat_poison(stat_t* sample_slots) {      // sample_slots = the sample's (sum, sumsq) pair
    atomicOr((unsigned long long*)(sample_slots + 1), STAT_POISON);
}
// `slot` = &stats[2 b + which]; the pair of a sample is 16-byte aligned, so (slot | 8) is its sum-of-squares slot
// `range_is_benign` (the normalisation alternates, where nothing is normalised by these sums -- they only carry the poison): a FINITE
// partial beyond the range is dropped instead of poisoning the sample.  keras gives finite rows for finite, large activations under
// inference-mode batch_norm / layer_norm1d (a restored checkpoint with a large scale or a small moving variance); so does this.
__device__ __forceinline__ void stat_add(stat_t* slot, double v, bool range_is_benign = false) {
    if (fabs(v) < STAT_PARTIAL_LIMIT)
        atomicAdd((unsigned long long*)slot, (unsigned long long)__double2ll_rn(v * (double)(1 << STAT_FRAC_BITS)));
    else if (!(range_is_benign && fabs(v) < 1.0e300))          // (NaN and infinity fail the comparison: they poison either way)
        atomicOr((unsigned long long*)((uintptr_t)slot | 8), STAT_POISON);
}
__device__ __forceinline__ double stat_get(const stat_t* slot) {
    const long long x = *slot;
    if (((uintptr_t)slot & 8) && (unsigned long long)x >= STAT_POISON) return __builtin_nan("");
    return (double)x * (1.0 / (double)(1 << STAT_FRAC_BITS));
}
// (r_b, c_b) = (rstd, -mean * rstd) of one sample from its statistics; NaN for a poisoned sample (the variance clamp keeps a NaN)
// inv_n < 0: IDENTITY statistics (the normalisation alternates, norm.hip): r = 1, c = 0 exactly -- NaN for a poisoned sample all the
// same (NaN * 0), so that its row still comes out NaN
__device__ __forceinline__ void stat_ln_scalars(const stat_t* sample_slots, double inv_n, float* r, float* c) {
    if (inv_n < 0.0) { const double z = stat_get(sample_slots + 1) * 0.0; *r = (float)(1.0 + z); *c = (float)z; return; }
    const double mean = stat_get(sample_slots) * inv_n;
    double var = stat_get(sample_slots + 1) * inv_n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + (double)LN_EPS);
    *r = (float)rstd; *c = (float)(-mean * rstd);
}
__device__ __forceinline__ bool stat_is_poisoned(const stat_t* sample_slots) {
    return (unsigned long long)sample_slots[1] >= STAT_POISON;
}
__device__ __forceinline__ void stat_forward_poison(const stat_t* in_slots, stat_t* out_slots) {
    if (stat_is_poisoned(in_slots)) stat_poison(out_slots);
}
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// atomic max / min on a float stored in global memory (any sign).
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
    else          atomicMin((unsigned int*)addr, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_min_float(float* addr, float v) {
    if (v >= 0.f) atomicMin((int*)addr, __float_as_int(v));
    else          atomicMax((unsigned int*)addr, __float_as_uint(v));
}

// ---- kernels' host launchers (defined in the .hip files) -------------------

// conv0: (B,F,T) x kernel(3,Cout) -> z0 = gamma0 . ELU(conv + bias), (B,F,Tout,Cout),
// + per-sample sum/sumsq of the ELU output.
// gstat != nullptr: `feat` is the raw log-mel and (max, min) per group of `group_size` samples; the batch-max
// subtraction / clamp / segment normalisation of melspectrogram.py:108-111 is applied on load.
int launch_conv0(const float* feat, const float* w3, const float* bias, const float* gamma, float* y,
                 float* v_out, stat_t* stats, int64_t B, const ConvGeom& g, hipStream_t st,
                 const float* gstat = nullptr, int group_size = 0, int segment_norm = 0, bool ident_stats = false);

// implicit-GEMM conv (see conv.hip for the LayerNorm folding).
constexpr int NAFP_TICKET_SLOTS = 4096;      // output tiles of a split-K launch that finishes in-kernel
// Side job a launch of the backward pass can carry: the per-sample scalar records of the LayerNorm backward of the layer below
// (what ln_bwd_scalars_kernel computes; backward.hip).  sc == null: none.
struct ScalarsJob { const float* mr; const double* lnsum; const float* mr_prev; float* sc; long long B; double inv_n; };   // sc == null: none
struct ConvGemmArgs {
    const float* x;          // (B,Fin,Tin,Cin): z of the previous conv (FULL) or a raw image (PLAIN)
    const float* wp;         // packed (Cout, 3*Cin), k = tap*Cin + cin
    const float* G;          // (P,Cout) conv(gamma_in)            FULL
    const float* Hb;         // (P,Cout) conv(beta_in) + bias      FULL
    const float* gamma_out;  // (P,Cout) LN scale of THIS conv     FULL
    const float* bias;       // (Cout) or nullptr                  PLAIN
    const stat_t* stats_in;  // (B,2): sum, sumsq of the previous conv's ELU output (fixed point, stat_get())   FULL
    bool ident_stats;        // FULL: derive r_b = 1, c_b = 0 from stats_in whatever it holds (a poisoned sample: NaN) -- the alternates of norm.hip
    stat_t* stats_out;       // (B,2), must be zero on entry                          FULL
    float* y;                // (B,Fout,Tout,Cout): z = gamma_out . v (FULL) or acc + bias (PLAIN)
    float* v_out;            // optional (FULL): the pre-activation t (v = ELU(t)), kept for the backward pass
    bool plain;
    bool dgrad;              // PLAIN only: y (B,Fin,Tin,Cin) = transposed conv of x = dT (B,Fout,Tout,Cout) with wp = (Cin, 3*Cout)
    float* slab;             // split-K partial sums workspace (or nullptr: never split)
    int64_t slab_floats;
    hipEvent_t ev_start, ev_stop;   // optional: time stamps attached to this conv's first / last kernel dispatch (no queue entry)
    const float* wp_hm; const void* wp_l;   // bf16x3 == 2: the weights pre-split into three bf16 terms (launch_split_weights_multi)
    int bf16x3;              // experimental: split-bf16 products for the unsplit FULL launches (NAFP_OPT_BF16X3): 1 = hi / lo, 3 products; 2 = exact 3-way split, 6 products
    unsigned* tickets;       // NAFP_TICKET_SLOTS arrival counters, zero on entry and on exit (or nullptr: split launches use slab + finish kernel)
    // optional: generate the A operand from the log-mel features (conv0 fused into conv1);
    // `x` is then unused.  f0_geom = geometry of conv0.
    const float* f0_feat; const float* f0_w; const float* f0_bias; const float* f0_gamma;
    const ConvGeom* f0_geom;
    const float* f0_gstat; int f0_group, f0_segnorm;   // (or null) f0_feat is the RAW log-mel of the front end: the layer's tail is applied on load, as launch_conv0 does
    const ScalarsJob* sj;    // optional (PLAIN launches of the backward pass): side job, see ScalarsJob
    int64_t plan_b;          // > 0 (inference forward): take the tile shape and the split-K factor the launch would have at THIS batch size,
                             // whatever B is -- the fp32 summation order of a segment then does not depend on the launch size (see fwd_plan_b())
};
// statistics (sum, sum of squares of ELU(conv0 + bias) per sample) without storing the activation
int launch_conv0_stats(const float* feat, const float* w3, const float* bias, stat_t* stats, int64_t B,
                       const ConvGeom& g, hipStream_t st, const float* gstat = nullptr, int group_size = 0, int segment_norm = 0);
int64_t conv_gemm_slab_floats(int64_t B, const ConvGeom& g, bool with_dgrad = false, int64_t plan_b = 0);   // workspace the split-K policy wants
int64_t fwd_plan_b();      // the batch size the inference forward plans its tiles for (NAFP_PLAN_B, default 640; 0 = the launch's own size)
int launch_conv_gemm(const ConvGemmArgs& a, int64_t B, const ConvGeom& g, hipStream_t st);
int conv_timeline_set(unsigned long long* buf, int64_t capacity_u64, int cin, int cout, int positions);
int conv_timeline_grid(int* out5);

// Transposed conv of layer j fused with the LayerNorm + ELU backward of layer j-1 (conv.hip, dgrad_ln_kernel): reads
// dts_j, t_{j-1} and the per-sample scalars of layer j-1, writes dts_{j-1} and accumulates dgamma / dbeta / dbias / S1 /
// S2 of layer j-1 and (G, Hb, lnsum_below given) the LayerNorm sums of layer j-2.
struct DgradLnArgs {
    const float* dts_in; const float* wd;
    const float* t; const float* gamma; const float* G; const float* Hb; const float* sc;
    float* dts_out; float* dgamma; float* dbeta; float* dbias; float* S1; float* S2; double* lnsum_below;
};
bool dgrad_ln_eligible(int64_t B, const ConvGeom& g, int mode);      // mode: 0 never, 1 big layers at B >= 64, 2 wherever the geometry permits
int launch_dgrad_ln(const DgradLnArgs& a, int64_t B, const ConvGeom& g, hipStream_t st);
// per-sample scalar records (B, 8) of one layer's LayerNorm backward (see backward.hip)
int launch_ln_bwd_scalars(const float* mr, const double* lnsum, const float* mr_prev, float* sc, int64_t B, int64_t n, hipStream_t st);

// tail: LN of the last conv + flatten + divide-and-encode + optional L2 norm.
struct TailArgs {
    const float* x;          // (B, D): z = gamma . v of the last conv, or already-normalised flat
    const stat_t* stats;     // (B,2) or nullptr when x is already the LN output
    const float* gamma;      // (D) or nullptr
    const float* beta;       // (D) or nullptr
    const float* w1p;        // (S,32,Q)
    const float* b1p;        // (32,Q)
    const float* w2p;        // (32,Q)
    const float* b2;         // (Q)
    float* out_flat;         // (B,D) or nullptr
    float* out_emb;          // (B,Q) or nullptr
    int D, Q, S, l2norm;
    bool ident_stats;        // r_b = 1, c_b = 0 whatever `stats` holds (a poisoned sample: NaN): the alternates of norm.hip
    const int* nonfinite_weights;   // (or null) != 0: the parameter set holds a NaN / Inf -> every output row is NaN
    const unsigned* launch_error;   // (or null) != 0: a launch of this pass raised its error word -> NaN rows
};
int launch_tail(const TailArgs& a, int64_t B, hipStream_t st);

// ---- backward pass (backward.hip) ----------------------------------------------------------
struct TailBwdArgs {
    const float* z; const stat_t* stats; const float* gamma; const float* beta;     // last conv (z = gamma . v)
    bool ident_stats;                                                               // identity statistics (the alternates of norm.hip)
    const float* w1; const float* b1; const float* w2; const float* b2;             // keras layouts
    const float* w1p; const float* b1p; const float* w2p;                           // the forward tail's layouts (S,32,Q) / (32,Q): kernel A, thread <-> q, reads them coalesced
    const float* d_emb;                                                             // (B,Q)
    float* dy;            // (B,Q) scratch
    float* ln;            // (B,2) scratch: the last conv's LayerNorm scalars (r_b, -mu_b r_b) per sample, written by kernel A for kernel B
    float* dxh;           // (B,D) out: dL/dxhat of the last conv
    float* dw1; float* db1; float* dw2; float* db2;                                 // keras layouts, out
    int D, Q, S, l2norm;
};
int launch_tail_bwd(const TailBwdArgs& a, int64_t B, hipStream_t st);
// identity: mean 0, rstd 1 for every sample (NaN for a poisoned one), see stat_ln_scalars
int launch_stats_to_mr(const stat_t* stats, float* mr, const double* inv_n_dev, int64_t B, int n_layers, hipStream_t st, bool identity = false);
// LayerNorm + ELU backward of one layer (backward.hip): d = r_j * dL/dxhat_j -> dts = r_{j-1} * dL/dt_j in place;
// dgamma/dbeta/dbias/S1/S2 accumulate (zeroed by the caller, like lnsum); sc = (B, 8) scratch.
// tpre = the layer's stored PRE-activation t (v = ELU(t) is recomputed).  reduce_here: compute this layer's (s1, s2)
// from d by a reduction pass (top layer); otherwise lnsum already holds them.  Gj / Hbj / lnsum_below (all or none):
// also accumulate the (s1, s2) of the layer below through the adjoint identity (see backward.hip).
int launch_ln_bwd(float* d, const float* tpre, const float* gamma, const float* mr, const float* mr_prev,
                  double* lnsum, float* sc, float* dgamma, float* dbeta, float* dbias, float* S1, float* S2,
                  int64_t B, int P, int C, hipStream_t st, bool reduce_here, const float* Gj, const float* Hbj,
                  double* lnsum_below, const float* feat0 = nullptr, const float* w0 = nullptr, const float* bias0 = nullptr,
                  const ConvGeom* g0 = nullptr, float* dW0 = nullptr, bool scalars_done = false,
                  float* part_slab = nullptr, int64_t part_slab_floats = 0, unsigned* tickets = nullptr, bool tpre_is_z = false);
// tpre_is_z: `tpre` is the layer's stored z = gamma . v instead of its pre-activation (v = z / gamma; gamma = the working copy)
// part_slab / tickets (or null): the batch chunks' (dgamma, dbeta, S1, S2) partials meet through the slab, summed in chunk
// order by the last arriver of each 1024-element block, instead of through atomics
// feat0 (layer 0 only): regenerate the pre-activation instead of reading tpre; dW0: also accumulate conv0's weight gradient
// (keras (1,3,1,C) layout) and leave `d` unwritten
// dW (keras (3,Cin,Cout), accumulated) += sum_rows X[b, in(pos,tap), :] (x) D[b,pos,:]
// X2 / D2 (both or neither): the aux samples [gamma_{j-1} | beta_{j-1}] resp. [S1_j | S2_j] (each pair adjacent), i.e. the two
// rank-one terms of dW_j; slab / tickets: workspace of the small-layer kernel (or null); sj: optional side job.
int launch_wgrad(const float* X, const float* D, float* dW, int64_t B, const ConvGeom& g, hipStream_t st,
                 const float* X2 = nullptr, const float* D2 = nullptr, float* slab = nullptr, int64_t slab_floats = 0,
                 unsigned* tickets = nullptr, const ScalarsJob* sj = nullptr, int prec = 0);      // prec 2: the regular-shape kernel on the exact 3-way bf16 split (the train step under NAFP_OPT_BF16X3 = 2)
int64_t wgrad_slab_floats(int64_t B, const ConvGeom& g);
int launch_conv0_bwd(const float* feat, const float* dt, float* dW0, float* dbias0, int64_t B, const ConvGeom& g,
                     hipStream_t st);
int launch_pack_dgrad_weight(const float* k3, float* wd, int Cin, int Cout, hipStream_t st);

// ---- normalisation alternates (norm.hip): MODEL.BN = 'layer_norm1d' | 'batch_norm' (nnfp.py:63-71) -----------------------------
// batch_norm: the positional scale / offset images (gamma_pos | beta_pos adjacent, n = P * C floats each) from the per-channel
// parameters and moving statistics
struct BnExpandTable { const float* gamma_c[16]; const float* beta_c[16]; const float* mmean[16]; const float* mvar[16];
                       float* gamma_pos[16]; int64_t n[16]; int C[16]; int count; };
int launch_bn_expand(const BnExpandTable& t, hipStream_t st);
int launch_bn_param_grad(const float* dgp, const float* dbp, int P, int C, const float* mmean, const float* mvar, float* dgamma_c,
                         float* dbeta_c, hipStream_t st);
// layer_norm1d: rows of C channels normalised in place; the backward of that row map in front of the shared LayerNorm / ELU backward
int launch_ln1d_fwd(float* x, int64_t rows, int C, const float* gamma_c, const float* beta_c, hipStream_t st);
int launch_ln1d_bwd(float* d, const float* tpre, int64_t rows, int C, const float* gamma_c, float* dgamma_c, float* dbeta_c, hipStream_t st);

// weight packing
struct PackTable { const float* k3[16]; float* wp[16]; float* wd[16]; int cin[16], cout[16]; int unit0[17]; int count;
                   int* nonfinite; };      // nonfinite (or null): set to 1 when a conv kernel holds a NaN / Inf (see nafp_encoder::d_wflag)
int launch_multi_pack(const PackTable& t, hipStream_t st);
// exact 3-way bf16 split of a packed (Cout, K) weight tensor for the bf16x3 == 2 launches: hm (Cout * K floats), wl (Cout * K bf16)
// every tensor of a parameter set in one launch: (Cout_j, K_j) f32 -> hm_j ([h(16) | m(16)] bf16 per group of 16 k, the f32 tensor's shape), wl_j ((Cout_j, K_j) bf16); n8 = Cout_j * K_j / 8
struct SplitTable { const float* wp[32]; float* hm[32]; void* wl[32]; int64_t n8[32]; int K[32]; int count; };
int launch_split_weights_multi(const SplitTable& t, hipStream_t st);
int launch_pack_conv_weight(const float* k3, float* wp, int Cin, int Cout, hipStream_t st);
// Positional epilogue tensors G_j = conv_j(gamma_{j-1}), Hb_j = conv_j(beta_{j-1}) (bias added later) of the SMALL layers
// (P <= 8 output positions: 2 P <= 16 GEMM rows against a 2 - 12 MB weight tensor) in ONE weight-streaming launch.
constexpr int GH_MAX_LAYERS = 8;
struct GhTable {
    const float* wp[GH_MAX_LAYERS];      // (Cout, 3 * Cin) packed weights
    const float* x[GH_MAX_LAYERS];       // gamma_{j-1} (Fin, Tin, Cin); beta_{j-1} follows at + sample_in
    float* y[GH_MAX_LAYERS];             // G_j (P, Cout); Hb_j follows at + P * Cout
    int cin[GH_MAX_LAYERS], cout[GH_MAX_LAYERS], P[GH_MAX_LAYERS], sample_in[GH_MAX_LAYERS];
    int src[GH_MAX_LAYERS][8][3];        // per output position and tap: float offset of the source row inside a sample, -1 = zero padding
    int wg0[GH_MAX_LAYERS + 1];          // first workgroup of each layer (16 output columns per workgroup)
    int count;
};
bool gh_gemv_eligible(const ConvGeom& g);
void gh_table_add(GhTable& t, const ConvGeom& g, const float* wp, const float* x, float* y);
int launch_gh_gemv(const GhTable& t, hipStream_t st);
int launch_pack_div(const float* w1, const float* b1, const float* w2, float* w1p, float* b1p,
                    float* w2p, int Q, int S, hipStream_t st);

}  // namespace nafp
