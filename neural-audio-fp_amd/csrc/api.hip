// C-ABI glue of libnafp: status strings and the encoder handle (geometry, packed
// weights, the 16-conv + tail launch sequence).  See include/nafp.h.
#include "nafp_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace nafp {

thread_local int g_last_hip_error = 0;

// Channel and stride tables of FingerPrinter (model/fp/nnfp.py:193-197).
static const int kHiddenCh[8] = {128, 128, 256, 256, 512, 512, 1024, 1024};
// stride of the 1x3 conv along T; the 3x1 conv always has stride 2 along F.
static const int kStrideT[8] = {2, 2, 2, 2, 1, 2, 1, 2};

std::vector<ConvGeom> encoder_geometry(int in_f, int in_t) {
    std::vector<ConvGeom> g;
    int F = in_f, T = in_t, C = 1;
    for (int i = 0; i < 8; ++i) {
        SamePad pt = same_pad(T, 3, kStrideT[i]);
        g.push_back(ConvGeom{0, F, T, C, F, pt.n_out, kHiddenCh[i], kStrideT[i], pt.before});
        T = pt.n_out; C = kHiddenCh[i];
        SamePad pf = same_pad(F, 3, 2);
        g.push_back(ConvGeom{1, F, T, C, pf.n_out, T, kHiddenCh[i], 2, pf.before});
        F = pf.n_out;
    }
    return g;
}

// set_weights helpers: all plain tensor copies in ONE launch, and bias_j added to every Hb_j in one launch.
// `nz` entries (the LayerNorm scales): the library's WORKING copy never holds a value of magnitude below 1e-30 -- an exact zero
// becomes +1e-30.  Every conv stores z = gamma . v; the backward pass recovers v = z / gamma from it instead of keeping a second
// tensor (the pre-activation) per layer, which needs gamma != 0.  Against activations of O(1) a scale of 1e-30 IS zero in
// float32 (its contribution vanishes in the first addition it meets), so the forward result does not change; the variable
// the caller sees is not touched.
struct CopyTable { const float* src[96]; float* dst[96]; int64_t n[96]; int nz[96]; int count; int* nonfinite; };
__device__ __forceinline__ float nz_scale(float g) { return fabsf(g) < 1e-30f ? copysignf(1e-30f, g) : g; }
__device__ __forceinline__ bool nonfinite4(float4 v) {
    return !(fabsf(v.x) <= 3.4028234664e38f) || !(fabsf(v.y) <= 3.4028234664e38f) || !(fabsf(v.z) <= 3.4028234664e38f) || !(fabsf(v.w) <= 3.4028234664e38f);
}
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyTable t) {
    const int e = blockIdx.y;
    const float* __restrict__ s = t.src[e]; float* __restrict__ d = t.dst[e];
    const int64_t n = t.n[e];
    const bool nz = t.nz[e] != 0;
    bool bad = false;                 // a NaN / Inf among the parameters (nafp_encoder::d_wflag)
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * 256) {
            float4 v = ((const float4*)s)[i];
            bad = bad || nonfinite4(v);
            if (nz) { v.x = nz_scale(v.x); v.y = nz_scale(v.y); v.z = nz_scale(v.z); v.w = nz_scale(v.w); }
            ((float4*)d)[i] = v;
        }
        for (int64_t i = n / 4 * 4 + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            bad = bad || !(fabsf(s[i]) <= 3.4028234664e38f);
            d[i] = nz ? nz_scale(s[i]) : s[i];
        }
    } else {
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            bad = bad || !(fabsf(s[i]) <= 3.4028234664e38f);
            d[i] = nz ? nz_scale(s[i]) : s[i];
        }
    }
    if (bad && t.nonfinite) atomicOr(t.nonfinite, 1);
}
#define NAFP_OPT_DEBUG_SIDE_DELAY 6
// TEST HOOK (NAFP_OPT_DEBUG_SIDE_DELAY): holds a stream for `us` microseconds (100 MHz real-time counter), so that a test can make the
// weight-gradient stream lag the main stream deterministically and see whether every event is ordered behind what it stands for
__global__ void debug_delay_kernel(long long us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < us * 100) __builtin_amdgcn_s_sleep(32);
}
struct BiasTable { float* hb[16]; const float* bias[16]; int64_t n[16]; int cout[16]; int count; };
__global__ __launch_bounds__(256) void add_bias_kernel(const BiasTable t) {
    const int e = blockIdx.y;
    float* __restrict__ h = t.hb[e]; const float* __restrict__ b = t.bias[e];
    const int64_t n = t.n[e]; const int C = t.cout[e];
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) h[i] += b[i % C];
}

}  // namespace nafp

using namespace nafp;

constexpr int NAFP_PROF_EV = 34;
struct nafp_encoder {
    int in_f, in_t, emb_sz;
    // NAFP_NORM_*: with the alternates (norm.hip) d_gamma / d_beta below are INTERNAL positional images (1 / 0 for layer_norm1d, the
    // broadcast affine map for batch_norm) and the caller's per-channel tensors live in d_gc / d_bc (/ d_mm / d_mv)
    int norm = NAFP_NORM_LAYER2D;
    int n_trainable = 68;
    std::vector<float*> d_gc, d_bc, d_mm, d_mv;
    bool ln1d_images_set = false;
    std::vector<ConvGeom> geom;           // 16
    int64_t flat_dim; int S;
    // tensor table (keras shapes)
    std::vector<std::vector<int64_t>> shapes;
    // device storage (one allocation)
    float* d_blob = nullptr;
    int64_t blob_floats = 0;
    std::vector<float*> d_w;              // per conv: conv0 -> raw (3,Cout); others packed (Cout,3Cin)
    std::vector<float*> d_bias, d_gamma, d_beta;
    std::vector<float*> d_G, d_Hb;        // per conv j >= 1: conv_j(gamma_{j-1}), conv_j(beta_{j-1}) + bias_j
    float *d_w1p = nullptr, *d_b1p = nullptr, *d_w2p = nullptr, *d_b2 = nullptr;
    // training only: dgrad operand layout (Cin, 3*Cout) per conv j >= 1, keras-layout divide-and-encode
    // weights for the tail backward, 1/n per layer for the statistics
    std::vector<float*> d_wd;
    float *d_w1k = nullptr, *d_b1k = nullptr, *d_w2k = nullptr;
    double* d_inv_n = nullptr;
    // != 0: the parameter set of the last set_weights holds a NaN or an Inf.  The packed ELU of the GEMM epilogues (max(t, exp(min(t, 0)) - 1))
    // drops a NaN operand, so NaN weights would come out as FINITE garbage; keras gives NaN fingerprints (nnfp.py:73-79).  The kernels
    // of set_weights that touch every parameter anyway raise this word, and the tail writes NaN rows when it is set.
    int* d_wflag = nullptr;
    float* d_sw_slab = nullptr; int64_t sw_slab_floats = 0;      // split-K slab of the G/Hb launches in set_weights
    bool has_weights = false;
    // Ordering of set_weights against the passes that read what it writes, when they run on DIFFERENT streams (a trainer
    // starts the re-pack of the updated variables on a stream of its own while the next batch's front end runs, nnfp.py
    // prefetch_weights): `sw_copied` = the plain copies are done (conv0 reads nothing else: its kernel, bias and gamma_0),
    // `sw_done` = everything is (packed kernels, G / Hb, the divide-and-encode re-layout).  A pass on another stream waits for
    // the event it needs right before the first launch that needs it (wait_weights): the training forward starts conv0 behind
    // `sw_copied`, conv1 behind `sw_l1` (set_weights forms layer 1's G / Hb first, on its own stream) and conv2 behind `sw_done`,
    // so the ~0.3 ms chain of small G / Hb launches runs under conv0 and conv1 instead of in front of them (B = 640: the main
    // stream sat idle for 0.19 ms per step waiting for it; next to conv0, which saturates the HBM write path, the chain stretches
    // to 0.4 ms, hence the third event).  Same stream: no wait is enqueued.
    hipEvent_t sw_copied = nullptr, sw_l1 = nullptr, sw_done = nullptr;      // (sw_l1: conv1's packed kernel and G / Hb are done too)
    hipStream_t sw_stream = nullptr;
    bool sw_recorded = false;
    // NAFP_OPT_FUSE_CONV0 (default: NAFP_FUSE0 env, else off).  Re-measured in round 4 on the current kernels (B = 640, same
    // box, tools/ab_bench.sh, two rounds): materialised conv0 0.257 ms + conv1 1.118 ms = 1.375 ms (187.2 k segments/s) vs
    // fused 0.172 ms (statistics pass) + 1.436 ms = 1.608 ms (179.3 k): the f32 MFMAs share the SIMDs' issue time with the
    // vector ALU, so re-generating z0 (3 FMAs + an exponential per element, per K-step) inside conv1 costs the matrix pipe
    // more than the 1.34 GB store + re-read costs the bus.  (Round 1: 0.35 + 1.17 vs 0.17 + 1.38.)
    bool opt_fuse_conv0 = []() { const char* v = getenv("NAFP_FUSE0"); return v && v[0] == '1'; }();
    // NAFP_OPT_FUSED_LN_BWD (default NAFP_DGRAD_LN env, else 1)
    // Measured (B = 1280, same box, ms per backward pass): 20.9 with the separate LayerNorm-backward pass, 23.3 with the
    // fused kernel: its one-position x 128-sample tiles read 128 different samples per K-step and its epilogue (row sums
    // for the layer below, five running batch sums) outweighs the two passes it saves.  Kept as an option, off by default.
    int opt_fused_ln_bwd = []() { const char* v = getenv("NAFP_DGRAD_LN"); return v ? atoi(v) : 0; }();
    // per-segment workspace layout (floats)
    int64_t bufA_per_seg = 0, bufB_per_seg = 0;
    // optional per-kernel event timing (nafp_encoder_profile_*)
    std::vector<hipEvent_t> prof_events;  // (max_forwards, NAFP_PROF_EV): [conv0 a, b | conv j start, stop (j = 1..15) | tail a, b]
    int prof_max = 0, prof_count = 0;
    int opt_bf16x3 = 0;                   // NAFP_OPT_BF16X3 (experimental): 0 off, 1 hi / lo split with 3 products, 2 exact 3-way split with 6 products
    // value 2: the packed conv kernels split into three bf16 terms (conv.hip, split_weights_multi_kernel), refreshed by the first forward after a
    // set_weights; one allocation of its own, made when the option is first switched on
    float* d_x6_blob = nullptr; std::vector<float*> d_whm; std::vector<void*> d_wl; bool x6_dirty = true;
    std::vector<float*> d_wdhm; std::vector<void*> d_wdl;      // ... and of the flipped kernels (Cin, 3 Cout) of the transposed convs (the train step under the option)
    int prof_coarse = 0;                  // 1: stamp only around conv0, the 15 GEMM convs as a group, and the tail; 2: only around the GEMM convs
    // nafp_encoder_backward records one event per gradient group (layers complete last to first), so that a
    // communication stream can start reducing a group while the rest of the backward pass still runs
    hipEvent_t grad_events[NAFP_GRAD_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    bool grad_events_valid = false;
    // NAFP_OPT_BWD_OVERLAP (default NAFP_BWD_OVERLAP env, else 2): weight gradients on a second stream of the handle.  The chain
    // LayerNorm backward(j) -> transposed conv(j) -> LayerNorm backward(j-1) ... is the critical path; wgrad(j) only needs dts_j
    // and is needed at the very end.
    //   1 = every layer's: MEASURED SLOWER.  Side by side with the HBM-bound LayerNorm pass both kernels stretch to the sum of
    //       their solo times or beyond (BSZ 5120: wgrad_2 3.96 -> 6.67 ms next to ln_bwd_1 3.0 -> 6.6 ms; step 93.3 -> 96.7 ms):
    //       the streaming pass starves wgrad's operand ring and its VALU work shares the SIMDs' issue slots with the f32 MFMAs.
    //   2 = only the SMALL layers' (P < 16 output positions: b5 ... b7).  Their launches -- weight gradient, LayerNorm backward and
    //       transposed conv alike -- are 250-750 short workgroups that fill a quarter of the chip, so wgrad(j) runs next to the
    //       LayerNorm backward of layer j-1 for free; it has a slab and arrival counters of its own (TrainLayout::slab2).
    //       Same-box A/B, ms per step: B = 640 13.04 -> 12.83, 1280 24.11 -> 24.07, 5120 88.1 -> 87.4.
    // The side stream has the DEFAULT priority: with a lowest- or highest-priority stream every hand-over between the two
    // queues cost ~0.4 ms (B = 640: 13.0 -> 18.0-18.2 ms per step for twelve of them).
    int opt_bwd_overlap = []() { const char* v = getenv("NAFP_BWD_OVERLAP"); return v ? atoi(v) : 2; }();
    // NAFP_KEEP_T=1: the training forward also stores every conv's pre-activation t (rounds 1-3).  Default 0: only z = gamma . v
    // is kept and the LayerNorm backward recovers what it needs of t from v = z / gamma (ln_bwd_fused_kernel<.., FROMZ>): the
    // forward convs lose their second store stream and the workspace 4.6 MB per segment.  The fused transposed-conv +
    // LayerNorm-backward path (opt_fused_ln_bwd) reads t and therefore implies keeping it.
    bool keep_t_env = []() { const char* v = getenv("NAFP_KEEP_T"); return v && v[0] == '1'; }();
    bool keep_t() const { return keep_t_env || (opt_fused_ln_bwd != 0 && norm == NAFP_NORM_LAYER2D) || norm == NAFP_NORM_LAYER1D; }   // (layer_norm1d overwrites z with the normalised rows: v = z / gamma is gone)
    // what forward_train laid a training workspace out with, PER WORKSPACE (the last 16 distinct ones): the backward pass re-derives the
    // layout from (B, keep_t()) and must find the one that forward pass wrote -- an option changed in between would make it read
    // activations at other offsets.  Keyed by the workspace pointer, so two forward passes into two workspaces (micro-batches in
    // flight) can both be followed by their backward passes, and a backward pass on a workspace no forward pass of this handle
    // wrote is refused (round-5 ADVICE: the record used to be one per handle)
    struct TrainRec { const void* ws; int64_t B; bool keep_t; };
    std::vector<TrainRec> train_recs;
    void train_rec_put(const void* ws, int64_t B, bool kt) {
        for (auto& r : train_recs) if (r.ws == ws) { r.B = B; r.keep_t = kt; return; }
        if (train_recs.size() >= 16) train_recs.erase(train_recs.begin());
        train_recs.push_back({ws, B, kt});
    }
    bool train_rec_ok(const void* ws, int64_t B, bool kt) const {
        for (const auto& r : train_recs) if (r.ws == ws) return r.B == B && r.keep_t == kt;
        return false;
    }
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_main[16] = {}, ev_side[16] = {};
    int debug_side_delay_us = 0;          // NAFP_OPT_DEBUG_SIDE_DELAY (tests only): the weight-gradient stream idles this long in front of its first launch of a pass
    // set_weights: the 15 G / Hb images are small, latency-bound launches (2 "samples"; the late ones stream 6-12 MB of
    // weights through a handful of workgroups): they run NAFP_SW_STREAMS abreast on helper streams of the handle, each
    // with its own split-K slab, between a fork and a join on the caller's stream.
    static constexpr int NAFP_SW_STREAMS = 4;
    hipStream_t sw_streams[NAFP_SW_STREAMS] = {};
    hipEvent_t sw_fork = nullptr, sw_join[NAFP_SW_STREAMS] = {};
};

// gradient group k = parameter tensors [kGroupFirst[k], kGroupLast[k]] in COMPLETION order of the backward pass
static const int kGroupFirst[NAFP_GRAD_GROUPS] = {48, 32, 16, 0};
static const int kGroupLast[NAFP_GRAD_GROUPS] = {67, 47, 31, 15};

static void profile_free(nafp_encoder* e) {
    for (auto ev : e->prof_events) (void)hipEventDestroy(ev);
    e->prof_events.clear();
    e->prof_max = 0; e->prof_count = 0;
}

static int64_t numel(const std::vector<int64_t>& s) {
    int64_t n = 1;
    for (auto d : s) n *= d;
    return n;
}

extern "C" int nafp_abi_version(void) { return NAFP_ABI_VERSION; }

extern "C" const char* nafp_status_string(int status) {
    switch (status) {
        case NAFP_OK: return "ok";
        case NAFP_ERR_INVALID_ARG: return "invalid argument";
        case NAFP_ERR_UNSUPPORTED: return "unsupported geometry or option";
        case NAFP_ERR_HIP: return "HIP runtime error";
        case NAFP_ERR_WORKSPACE: return "workspace too small";
        case NAFP_ERR_NO_WEIGHTS: return "encoder weights not set";
        default: return "unknown status";
    }
}

extern "C" int nafp_last_hip_error(void) { return g_last_hip_error; }

// CRC-32C (Castagnoli, reflected 0x82F63B78), slicing-by-8, host only: checksums of TensorFlow checkpoint files
// (model/utils/tf_checkpoint.py).  crc = running value (0 to start).
extern "C" uint32_t nafp_crc32c_host(const void* data, int64_t n, uint32_t crc) {
    static uint32_t T[8][256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xffu];
        init = true;
    }
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = ~crc;
    while (n > 0 && ((uintptr_t)p & 7)) { c = (c >> 8) ^ T[0][(c ^ *p++) & 0xffu]; --n; }
    while (n >= 8) {
        uint64_t w; memcpy(&w, p, 8);
        w ^= c;
        c = T[7][w & 0xff] ^ T[6][(w >> 8) & 0xff] ^ T[5][(w >> 16) & 0xff] ^ T[4][(w >> 24) & 0xff] ^
            T[3][(w >> 32) & 0xff] ^ T[2][(w >> 40) & 0xff] ^ T[1][(w >> 48) & 0xff] ^ T[0][(w >> 56) & 0xff];
        p += 8; n -= 8;
    }
    while (n-- > 0) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xffu];
    return ~c;
}

extern "C" int nafp_encoder_create(nafp_encoder** out, int in_f, int in_t, int emb_sz) {
    return nafp_encoder_create_ex(out, in_f, in_t, emb_sz, NAFP_NORM_LAYER2D);
}
extern "C" int nafp_encoder_norm(const nafp_encoder* e) { return e ? e->norm : -1; }
extern "C" int nafp_encoder_n_trainable(const nafp_encoder* e) { return e ? e->n_trainable : -1; }

extern "C" int nafp_encoder_create_ex(nafp_encoder** out, int in_f, int in_t, int emb_sz, int norm) {
    if (!out || in_f <= 0 || in_t <= 0 || emb_sz <= 0) return NAFP_ERR_INVALID_ARG;
    if (norm != NAFP_NORM_LAYER2D && norm != NAFP_NORM_LAYER1D && norm != NAFP_NORM_BATCH) return NAFP_ERR_INVALID_ARG;
    nafp_encoder* e = new nafp_encoder();
    e->in_f = in_f; e->in_t = in_t; e->emb_sz = emb_sz; e->norm = norm;
    e->geom = encoder_geometry(in_f, in_t);
    const ConvGeom& last = e->geom.back();
    e->flat_dim = (int64_t)last.Fout * last.Tout * last.Cout;
    if (e->flat_dim % emb_sz != 0 || emb_sz % 64 != 0 || emb_sz > 256 || e->flat_dim / emb_sz > 16) {
        delete e; return NAFP_ERR_UNSUPPORTED;
    }
    e->S = (int)(e->flat_dim / emb_sz);
    for (int j = 0; j < 16; ++j) {
        const ConvGeom& g = e->geom[j];
        if (g.axis == 0) e->shapes.push_back({1, 3, g.Cin, g.Cout});
        else             e->shapes.push_back({3, 1, g.Cin, g.Cout});
        e->shapes.push_back({g.Cout});
        e->shapes.push_back({g.Fout, g.Tout, g.Cout});
        e->shapes.push_back({g.Fout, g.Tout, g.Cout});
    }
    e->shapes.push_back({emb_sz, e->S, 32});
    e->shapes.push_back({emb_sz, 32});
    e->shapes.push_back({emb_sz, 32, 1});
    e->shapes.push_back({emb_sz, 1});
    // (the device blob is laid out from the layer_norm2d shapes -- the alternates keep positional images of that size --; the shapes
    // the CALLER sees are put in place at the end)
    int64_t total = 0;
    for (auto& s : e->shapes) total += (numel(s) + 63) / 64 * 64;      // 256-B aligned slots
    if (norm != NAFP_NORM_LAYER2D) for (int j = 0; j < 16; ++j) total += 4 * ((e->geom[j].Cout + 63) / 64 * 64);   // gamma_c, beta_c, moving mean / variance
    for (int j = 1; j < 16; ++j) total += 2 * ((numel(e->shapes[4 * j + 2]) + 63) / 64 * 64);   // G, Hb
    for (int j = 1; j < 16; ++j) total += (numel(e->shapes[4 * j]) + 63) / 64 * 64;               // dgrad weights
    total += (numel(e->shapes[64]) + 63) / 64 * 64 + 2 * ((numel(e->shapes[65]) + 63) / 64 * 64) + 64 + 64;   // keras div copies, inv_n, d_wflag
    for (int j = 1; j < 16; ++j) e->sw_slab_floats = std::max(e->sw_slab_floats, conv_gemm_slab_floats(2, e->geom[j]));
    total += nafp_encoder::NAFP_SW_STREAMS * (e->sw_slab_floats + 64);
    e->blob_floats = total;
    hipError_t err = hipMalloc(&e->d_blob, sizeof(float) * total);
    if (err != hipSuccess) { g_last_hip_error = (int)err; delete e; return NAFP_ERR_HIP; }
    float* p = e->d_blob;
    auto take = [&](int idx) { float* r = p; p += (numel(e->shapes[idx]) + 63) / 64 * 64; return r; };
    for (int j = 0; j < 16; ++j) {
        e->d_w.push_back(take(4 * j));
        e->d_bias.push_back(take(4 * j + 1));
        e->d_gamma.push_back(take(4 * j + 2));
        e->d_beta.push_back(take(4 * j + 3));
    }
    e->d_w1p = take(64); e->d_b1p = take(65); e->d_w2p = take(66); e->d_b2 = take(67);
    e->d_G.push_back(nullptr); e->d_Hb.push_back(nullptr);
    for (int j = 1; j < 16; ++j) { e->d_G.push_back(take(4 * j + 2)); e->d_Hb.push_back(take(4 * j + 2)); }
    e->d_wd.push_back(nullptr);
    for (int j = 1; j < 16; ++j) e->d_wd.push_back(take(4 * j));
    e->d_w1k = take(64); e->d_b1k = take(65); e->d_w2k = take(65);
    e->d_inv_n = (double*)p; p += 64;
    e->d_wflag = (int*)p; p += 64;
    e->d_sw_slab = p; p += nafp_encoder::NAFP_SW_STREAMS * (e->sw_slab_floats + 64);      // one slab per helper stream, (sw_slab_floats + 64) apart
    if (norm != NAFP_NORM_LAYER2D)
        for (int j = 0; j < 16; ++j) {
            const int64_t c = (e->geom[j].Cout + 63) / 64 * 64;
            e->d_gc.push_back(p); e->d_bc.push_back(p + c); e->d_mm.push_back(p + 2 * c); e->d_mv.push_back(p + 3 * c); p += 4 * c;
        }
    // set_weights runs G_j and Hb_j as the two "samples" of one launch: the pairs must be adjacent
    for (int j = 0; j < 16; ++j) {
        const int64_t n = numel(e->shapes[4 * j + 2]);
        if (e->d_beta[j] != e->d_gamma[j] + n || (j >= 1 && e->d_Hb[j] != e->d_G[j] + n)) {
            (void)hipFree(e->d_blob); delete e; return NAFP_ERR_UNSUPPORTED;
        }
    }
    {
        double inv_n[16];
        for (int j = 0; j < 16; ++j) inv_n[j] = 1.0 / ((double)e->geom[j].Fout * e->geom[j].Tout * e->geom[j].Cout);
        err = hipMemcpy(e->d_inv_n, inv_n, sizeof(inv_n), hipMemcpyHostToDevice);
        if (err != hipSuccess) { g_last_hip_error = (int)err; (void)hipFree(e->d_blob); delete e; return NAFP_ERR_HIP; }
    }
    for (int j = 0; j < 16; ++j) {
        const ConvGeom& g = e->geom[j];
        const int64_t n = (int64_t)g.Fout * g.Tout * g.Cout;
        if (j % 2 == 0) e->bufA_per_seg = std::max(e->bufA_per_seg, n);
        else            e->bufB_per_seg = std::max(e->bufB_per_seg, n);
    }
    if (norm != NAFP_NORM_LAYER2D) {
        for (int j = 0; j < 16; ++j) { e->shapes[4 * j + 2] = {e->geom[j].Cout}; e->shapes[4 * j + 3] = {e->geom[j].Cout}; }
        if (norm == NAFP_NORM_BATCH)
            for (int j = 0; j < 16; ++j) { e->shapes.push_back({e->geom[j].Cout}); e->shapes.push_back({e->geom[j].Cout}); }
    }
    *out = e;
    return NAFP_OK;
}

extern "C" int nafp_encoder_destroy(nafp_encoder* e) {
    if (!e) return NAFP_OK;
    if (e->d_blob) (void)hipFree(e->d_blob);
    if (e->d_x6_blob) (void)hipFree(e->d_x6_blob);
    profile_free(e);
    for (auto& ev : e->grad_events) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_main) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_side) if (ev) (void)hipEventDestroy(ev);
    if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
    for (auto& q : e->sw_streams) if (q) (void)hipStreamDestroy(q);
    for (auto& ev : e->sw_join) if (ev) (void)hipEventDestroy(ev);
    if (e->sw_fork) (void)hipEventDestroy(e->sw_fork);
    if (e->sw_copied) (void)hipEventDestroy(e->sw_copied);
    if (e->sw_l1) (void)hipEventDestroy(e->sw_l1);
    if (e->sw_done) (void)hipEventDestroy(e->sw_done);
    delete e;
    return NAFP_OK;
}

// NAFP_OPT_BF16X3 = 2: the three bf16 terms of every packed conv kernel and of its flipped form (conv.hip split_weights_multi_kernel)
static int split_all_weights(nafp_encoder* e, hipStream_t st) {
    SplitTable t; t.count = 0;
    for (int j = 1; j < 16; ++j) {
        const int64_t n = (int64_t)e->geom[j].Cout * 3 * e->geom[j].Cin;
        t.wp[t.count] = e->d_w[j]; t.hm[t.count] = e->d_whm[j]; t.wl[t.count] = e->d_wl[j]; t.n8[t.count] = n / 8; t.K[t.count] = 3 * e->geom[j].Cin; ++t.count;
        if (e->geom[j].Cout % 16 == 0 && e->d_wd[j]) {          // the flipped kernel (Cin, 3 Cout): K = 3 Cout must be whole 16-k groups
            t.wp[t.count] = e->d_wd[j]; t.hm[t.count] = e->d_wdhm[j]; t.wl[t.count] = e->d_wdl[j]; t.n8[t.count] = n / 8; t.K[t.count] = 3 * e->geom[j].Cout; ++t.count;
        }
    }
    { int rcs = launch_split_weights_multi(t, st); if (rcs != NAFP_OK) return rcs; }
    e->x6_dirty = false;
    return NAFP_OK;
}

extern "C" int nafp_encoder_set_option(nafp_encoder* e, int option, int value) {
    if (!e) return NAFP_ERR_INVALID_ARG;
    switch (option) {
        case NAFP_OPT_FUSE_CONV0: e->opt_fuse_conv0 = value != 0; return NAFP_OK;
        case NAFP_OPT_BF16X3:
            if (value < 0 || value > 2) return NAFP_ERR_INVALID_ARG;
            if (value == 2 && !e->d_x6_blob) {
                int64_t tot = 0;
                for (int j = 1; j < 16; ++j) tot += 2 * (((int64_t)e->geom[j].Cout * 3 * e->geom[j].Cin * 3 / 2 + 63) / 64 * 64);
                hipError_t er = hipMalloc(&e->d_x6_blob, sizeof(float) * tot);
                if (er != hipSuccess) { g_last_hip_error = (int)er; e->d_x6_blob = nullptr; return NAFP_ERR_HIP; }
                float* q = e->d_x6_blob;
                e->d_whm.assign(16, nullptr); e->d_wl.assign(16, nullptr); e->d_wdhm.assign(16, nullptr); e->d_wdl.assign(16, nullptr);
                for (int j = 1; j < 16; ++j) {
                    const int64_t n = (int64_t)e->geom[j].Cout * 3 * e->geom[j].Cin;
                    e->d_whm[j] = q; e->d_wl[j] = q + n; q += (n * 3 / 2 + 63) / 64 * 64;
                    e->d_wdhm[j] = q; e->d_wdl[j] = q + n; q += (n * 3 / 2 + 63) / 64 * 64;
                }
                e->x6_dirty = true;
            }
            e->opt_bf16x3 = value; return NAFP_OK;
        case NAFP_OPT_FUSED_LN_BWD:
            if (value < 0 || value > 2) return NAFP_ERR_INVALID_ARG;
            e->opt_fused_ln_bwd = value; return NAFP_OK;
        case NAFP_OPT_BWD_OVERLAP: e->opt_bwd_overlap = value < 0 ? 0 : (value > 2 ? 1 : value); return NAFP_OK;
        case NAFP_OPT_SMALLNET: return NAFP_OK;          // retired in round 6 (the persistent small-layer launch lost to the per-layer launches at every setting): accepted, ignored
        case NAFP_OPT_DEBUG_SIDE_DELAY: {     // test hook, not in include/nafp.h: refused outside a test process (round-5 ADVICE)
            static const bool hooks = []() { const char* v = getenv("NAFP_TEST_HOOKS"); return v && v[0] == '1'; }();
            if (!hooks) return NAFP_ERR_UNSUPPORTED;
            e->debug_side_delay_us = value < 0 ? 0 : value; return NAFP_OK;
        }
        default: return NAFP_ERR_INVALID_ARG;
    }
}

extern "C" int nafp_encoder_n_tensors(const nafp_encoder* e) { return e ? (int)e->shapes.size() : -1; }

extern "C" int64_t nafp_encoder_tensor_numel(const nafp_encoder* e, int index) {
    if (!e || index < 0 || index >= (int)e->shapes.size()) return -1;
    return numel(e->shapes[index]);
}

extern "C" int nafp_encoder_tensor_shape(const nafp_encoder* e, int index, int64_t dims_out[4]) {
    if (!e || !dims_out || index < 0 || index >= (int)e->shapes.size()) return -1;
    const auto& s = e->shapes[index];
    for (size_t i = 0; i < s.size(); ++i) dims_out[i] = s[i];
    return (int)s.size();
}

extern "C" int64_t nafp_encoder_flat_dim(const nafp_encoder* e) { return e ? e->flat_dim : -1; }

extern "C" int nafp_encoder_set_weights(nafp_encoder* e, const float* const* t, void* stream) {
    if (!e || !t) return NAFP_ERR_INVALID_ARG;
    for (size_t i = 0; i < e->shapes.size(); ++i)
        if (!t[i]) return NAFP_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    CopyTable ct; ct.count = 0;
    auto add_copy = [&](const float* src, float* dst, int64_t n, int nz = 0) { ct.src[ct.count] = src; ct.dst[ct.count] = dst; ct.n[ct.count] = n; ct.nz[ct.count] = nz; ++ct.count; };
    add_copy(t[0], e->d_w[0], 3 * e->geom[0].Cout);
    for (int j = 0; j < 16; ++j) {
        const ConvGeom& g = e->geom[j];
        const int64_t nln = (int64_t)g.Fout * g.Tout * g.Cout;
        add_copy(t[4 * j + 1], e->d_bias[j], g.Cout);
        if (e->norm == NAFP_NORM_LAYER2D) {
            add_copy(t[4 * j + 2], e->d_gamma[j], nln, 1);          // LayerNorm scale: kept away from exact zero (multi_copy_kernel)
            add_copy(t[4 * j + 3], e->d_beta[j], nln);
        } else {
            add_copy(t[4 * j + 2], e->d_gc[j], g.Cout);
            add_copy(t[4 * j + 3], e->d_bc[j], g.Cout);
            if (e->norm == NAFP_NORM_BATCH) { add_copy(t[68 + 2 * j], e->d_mm[j], g.Cout); add_copy(t[69 + 2 * j], e->d_mv[j], g.Cout); }
        }
    }
    add_copy(t[64], e->d_w1k, numel(e->shapes[64]));
    add_copy(t[65], e->d_b1k, numel(e->shapes[65]));
    add_copy(t[66], e->d_w2k, numel(e->shapes[66]));
    add_copy(t[67], e->d_b2, e->emb_sz);
    if (!e->sw_copied) {
        NAFP_HIP_CHECK(hipEventCreateWithFlags(&e->sw_copied, hipEventDisableTiming));
        NAFP_HIP_CHECK(hipEventCreateWithFlags(&e->sw_done, hipEventDisableTiming));
        NAFP_HIP_CHECK(hipEventCreateWithFlags(&e->sw_l1, hipEventDisableTiming));
    }
    e->sw_recorded = false;                 // (an early return below leaves the passes without events to wait for: the caller got an error)
    NAFP_HIP_CHECK(hipMemsetAsync(e->d_wflag, 0, sizeof(int), st));
    ct.nonfinite = e->d_wflag;
    multi_copy_kernel<<<dim3(32, ct.count), 256, 0, st>>>(ct);
    NAFP_LAUNCH_CHECK();
    // the alternates' positional images (norm.hip): 1 / 0 for layer_norm1d (the row pass applies gamma_c / beta_c), the broadcast
    // affine map of the moving statistics for batch_norm -- in front of `sw_copied`: conv0 reads gamma_pos of layer 0
    if (e->norm == NAFP_NORM_LAYER1D && !e->ln1d_images_set) {      // constants: written by the first call only
        e->ln1d_images_set = true;
        for (int j = 0; j < 16; ++j) {
            const int64_t nln = (int64_t)e->geom[j].Fout * e->geom[j].Tout * e->geom[j].Cout;
            NAFP_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)e->d_gamma[j], 0x3f800000, nln, st));
            NAFP_HIP_CHECK(hipMemsetAsync(e->d_beta[j], 0, sizeof(float) * nln, st));
        }
    } else if (e->norm == NAFP_NORM_BATCH) {
        BnExpandTable bx; bx.count = 16;
        for (int j = 0; j < 16; ++j) {
            bx.gamma_c[j] = e->d_gc[j]; bx.beta_c[j] = e->d_bc[j]; bx.mmean[j] = e->d_mm[j]; bx.mvar[j] = e->d_mv[j];
            bx.gamma_pos[j] = e->d_gamma[j]; bx.n[j] = (int64_t)e->geom[j].Fout * e->geom[j].Tout * e->geom[j].Cout; bx.C[j] = e->geom[j].Cout;
        }
        int rcx = launch_bn_expand(bx, st);
        if (rcx != NAFP_OK) return rcx;
    }
    NAFP_HIP_CHECK(hipEventRecord(e->sw_copied, st));
    {
        PackTable pt; pt.count = 0; pt.nonfinite = e->d_wflag;
        for (int j = 1; j < 16; ++j) {
            pt.k3[pt.count] = t[4 * j]; pt.wp[pt.count] = e->d_w[j]; pt.wd[pt.count] = e->d_wd[j];
            pt.cin[pt.count] = e->geom[j].Cin; pt.cout[pt.count] = e->geom[j].Cout; ++pt.count;
        }
        int rc = launch_multi_pack(pt, st);
        if (rc != NAFP_OK) return rc;
    }
    // positional epilogue terms of conv j: G = conv_j(gamma_{j-1}), Hb = conv_j(beta_{j-1}) + bias_j:
    // one 2-"sample" PLAIN launch per conv (gamma | beta adjacent in, G | Hb adjacent out), then all biases at once
    BiasTable bt; bt.count = 0;
    // (NAFP_SW_NSTREAMS: how many of the helper streams are used, 1..4; NAFP_SW_GEMV_OWN=1: the weight-streaming launch on the
    // caller's stream itself instead of the last helper)
    static const int ns_env = []() { const char* v = getenv("NAFP_SW_NSTREAMS"); return v ? atoi(v) : nafp_encoder::NAFP_SW_STREAMS; }();
    static const bool gemv_own = []() { const char* v = getenv("NAFP_SW_GEMV_OWN"); return v && v[0] == '1'; }();
    const int NS = std::max(1, std::min<int>(nafp_encoder::NAFP_SW_STREAMS, ns_env));
    if (!e->sw_fork) {
        NAFP_HIP_CHECK(hipEventCreateWithFlags(&e->sw_fork, hipEventDisableTiming));
        for (int k = 0; k < NS; ++k) {
            NAFP_HIP_CHECK(hipStreamCreateWithFlags(&e->sw_streams[k], hipStreamNonBlocking));
            NAFP_HIP_CHECK(hipEventCreateWithFlags(&e->sw_join[k], hipEventDisableTiming));
        }
    }
    // layer 1 first, on the caller's stream: the training forward's conv1 waits for this much only (sw_l1)
    bool l1_done = false;
    if (!gh_gemv_eligible(e->geom[1])) {
        ConvGemmArgs a{};
        a.wp = e->d_w[1]; a.plain = true; a.x = e->d_gamma[0]; a.bias = nullptr; a.y = e->d_G[1];
        a.slab = e->sw_slab_floats ? e->d_sw_slab : nullptr; a.slab_floats = e->sw_slab_floats;
        int rc1 = launch_conv_gemm(a, 2, e->geom[1], st);
        if (rc1 != NAFP_OK) return rc1;
        BiasTable b1; b1.count = 1;
        b1.hb[0] = e->d_Hb[1]; b1.bias[0] = e->d_bias[1]; b1.n[0] = (int64_t)e->geom[1].Fout * e->geom[1].Tout * e->geom[1].Cout; b1.cout[0] = e->geom[1].Cout;
        add_bias_kernel<<<dim3(32, 1), 256, 0, st>>>(b1);
        NAFP_LAUNCH_CHECK();
        l1_done = true;
        NAFP_HIP_CHECK(hipEventRecord(e->sw_l1, st));
    }
    NAFP_HIP_CHECK(hipEventRecord(e->sw_fork, st));                         // copies and re-packs above are done
    // Between the fork and the join nothing returns early: whatever fails, the helper streams are joined back into the
    // caller's stream first -- otherwise later work on `st` would race G / Hb launches still running on the helpers, and the
    // next call would reuse their slabs under them.  The first error is kept and returned after the join.
    int fork_rc = NAFP_OK;
    int forked = 0;
    for (int k = 0; k < NS && fork_rc == NAFP_OK; ++k) {
        if (hipStreamWaitEvent(e->sw_streams[k], e->sw_fork, 0) != hipSuccess) fork_rc = NAFP_ERR_HIP; else ++forked;
    }
    // the small layers (P <= 8): ONE weight-streaming launch (conv.hip, gh_gemv_kernel), first in line on the last helper stream
    GhTable gh; gh.count = 0;
    bool by_gemv[16] = {};
    for (int j = 1; j < 16; ++j)
        if (gh.count < GH_MAX_LAYERS && gh_gemv_eligible(e->geom[j])) {
            gh_table_add(gh, e->geom[j], e->d_w[j], e->d_gamma[j - 1], e->d_G[j]);
            by_gemv[j] = true;
            bt.hb[bt.count] = e->d_Hb[j]; bt.bias[bt.count] = e->d_bias[j];
            bt.n[bt.count] = (int64_t)e->geom[j].Fout * e->geom[j].Tout * e->geom[j].Cout; bt.cout[bt.count] = e->geom[j].Cout; ++bt.count;
        }
    if (fork_rc == NAFP_OK && forked == NS) fork_rc = launch_gh_gemv(gh, gemv_own ? st : e->sw_streams[NS - 1]);
    else for (int j = 1; j < 16; ++j) by_gemv[j] = false;              // (could not fork: everything through the tiled launches, bt rebuilt below)
    if (forked != NS) bt.count = 0;
    int n_tiled = 0;
    for (int j = 1; j < 16 && fork_rc == NAFP_OK; ++j) {
        if (by_gemv[j] || (j == 1 && l1_done)) continue;
        const int k = (n_tiled++) % NS;
        ConvGemmArgs a{};
        a.wp = e->d_w[j]; a.plain = true;
        a.x = e->d_gamma[j - 1]; a.bias = nullptr; a.y = e->d_G[j];
        a.slab = e->sw_slab_floats ? e->d_sw_slab + (int64_t)k * (e->sw_slab_floats + 64) : nullptr; a.slab_floats = e->sw_slab_floats;
        fork_rc = launch_conv_gemm(a, 2, e->geom[j], e->sw_streams[k]);
        bt.hb[bt.count] = e->d_Hb[j]; bt.bias[bt.count] = e->d_bias[j];
        bt.n[bt.count] = (int64_t)e->geom[j].Fout * e->geom[j].Tout * e->geom[j].Cout; bt.cout[bt.count] = e->geom[j].Cout; ++bt.count;
    }
    for (int k = 0; k < forked; ++k) {
        if (hipEventRecord(e->sw_join[k], e->sw_streams[k]) != hipSuccess || hipStreamWaitEvent(st, e->sw_join[k], 0) != hipSuccess) {
            (void)hipStreamSynchronize(e->sw_streams[k]);                   // cannot order it on the device: drain it on the host
            if (fork_rc == NAFP_OK) fork_rc = NAFP_ERR_HIP;
        }
    }
    if (fork_rc != NAFP_OK) return fork_rc;
    add_bias_kernel<<<dim3(32, bt.count), 256, 0, st>>>(bt);
    NAFP_LAUNCH_CHECK();
    int rc = launch_pack_div(t[64], t[65], t[66], e->d_w1p, e->d_b1p, e->d_w2p, e->emb_sz, e->S, st);
    if (rc != NAFP_OK) return rc;
    if (!l1_done) NAFP_HIP_CHECK(hipEventRecord(e->sw_l1, st));
    e->x6_dirty = true;
    if (e->opt_bf16x3 == 2) {                 // (experimental) the exact-split kernels' three bf16 terms of every packed conv kernel: part of this call,
        int rcs = split_all_weights(e, st);   //  so that `sw_done` covers them for passes on other streams
        if (rcs != NAFP_OK) return rcs;
    }
    NAFP_HIP_CHECK(hipEventRecord(e->sw_done, st));
    e->sw_stream = st; e->sw_recorded = true;
    e->has_weights = true;
    return NAFP_OK;
}

// A pass on stream `st` is about to read what the last set_weights wrote (all of it, or only the plain copies): see sw_copied / sw_done.
static int wait_weights(nafp_encoder* e, hipStream_t st, int stage = 2) {          // 0: the plain copies, 1: + layer 1, 2: everything
    if (!e->sw_recorded || st == e->sw_stream) return NAFP_OK;
    NAFP_HIP_CHECK(hipStreamWaitEvent(st, stage == 0 ? e->sw_copied : (stage == 1 ? e->sw_l1 : e->sw_done), 0));
    return NAFP_OK;
}

static int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// split-K slab of a forward pass: the largest per-layer need
static int64_t forward_slab_floats(const nafp_encoder* e, int64_t n_seg) {
    int64_t slab = 0;
    for (int j = 1; j < 16; ++j) slab = std::max(slab, conv_gemm_slab_floats(n_seg, e->geom[j], false, fwd_plan_b()));
    return slab;
}

extern "C" int64_t nafp_encoder_workspace_bytes(const nafp_encoder* e, int64_t n_seg) {
    if (!e || n_seg < 0) return -1;
    const int64_t stats = align_up((int64_t)sizeof(double) * 2 * 16 * n_seg, 256) + NAFP_TICKET_SLOTS * (int64_t)sizeof(unsigned);
    const int64_t a = align_up((int64_t)sizeof(float) * e->bufA_per_seg * n_seg, 256);
    const int64_t b = align_up((int64_t)sizeof(float) * e->bufB_per_seg * n_seg, 256);
    return stats + a + b + align_up(forward_slab_floats(e, n_seg) * (int64_t)sizeof(float), 256) + 256;
}

static int encoder_forward_impl(nafp_encoder* e, const float* feat, const float* gstat, int group_size, int segment_norm,
                                int64_t n_seg, void* workspace, int64_t workspace_bytes, float* out_flat,
                                float* out_emb, int l2norm, void* stream);

extern "C" int nafp_encoder_forward(nafp_encoder* e, const float* feat, int64_t n_seg,
                                    void* workspace, int64_t workspace_bytes,
                                    float* out_flat, float* out_emb, int l2norm, void* stream) {
    return encoder_forward_impl(e, feat, nullptr, 0, 0, n_seg, workspace, workspace_bytes, out_flat, out_emb, l2norm, stream);
}

extern "C" int nafp_encoder_forward_raw(nafp_encoder* e, const float* raw_feat, const float* group_stat, int group_size,
                                        int segment_norm, int64_t n_seg, void* workspace, int64_t workspace_bytes,
                                        float* out_flat, float* out_emb, int l2norm, void* stream) {
    if (!group_stat) return NAFP_ERR_INVALID_ARG;
    if (group_size <= 0 || group_size > n_seg) group_size = (int)std::min<int64_t>(n_seg, INT32_MAX);   // as the front end
    return encoder_forward_impl(e, raw_feat, group_stat, group_size, segment_norm & 1, n_seg, workspace, workspace_bytes,
                                out_flat, out_emb, l2norm, stream);
}

static int encoder_forward_impl(nafp_encoder* e, const float* feat, const float* gstat, int group_size, int segment_norm,
                                int64_t n_seg, void* workspace, int64_t workspace_bytes, float* out_flat,
                                float* out_emb, int l2norm, void* stream) {
    if (!e || !feat || !workspace || n_seg < 0) return NAFP_ERR_INVALID_ARG;
    if (!e->has_weights) return NAFP_ERR_NO_WEIGHTS;
    if (n_seg == 0) return NAFP_OK;
    if (workspace_bytes < nafp_encoder_workspace_bytes(e, n_seg)) return NAFP_ERR_WORKSPACE;
    if (n_seg * e->bufA_per_seg >= ((int64_t)1 << 40)) return NAFP_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)align_up((int64_t)(uintptr_t)workspace, 256);
    const int64_t stats_only = align_up((int64_t)sizeof(double) * 2 * 16 * n_seg, 256);
    const int64_t stats_bytes = stats_only + NAFP_TICKET_SLOTS * (int64_t)sizeof(unsigned);
    stat_t* stats = (stat_t*)ws;
    unsigned* tickets = (unsigned*)(ws + stats_only);       // arrival counters of the split-K launches (zero between launches)
    float* bufA = (float*)(ws + stats_bytes);
    float* bufB = (float*)(ws + stats_bytes + align_up((int64_t)sizeof(float) * e->bufA_per_seg * n_seg, 256));
    float* slab = (float*)((char*)bufB + align_up((int64_t)sizeof(float) * e->bufB_per_seg * n_seg, 256));
    const int64_t slab_floats = forward_slab_floats(e, n_seg);
    { int wrc = wait_weights(e, st); if (wrc != NAFP_OK) return wrc; }
    if (e->opt_bf16x3 == 2 && e->x6_dirty) {                          // (experimental) the weights' three bf16 terms, once per parameter set
        { int rcs = split_all_weights(e, st); if (rcs != NAFP_OK) return rcs; }
        // (the option was switched on after the last set_weights: later passes on OTHER streams wait for this split like for a re-pack)
        if (!e->sw_done) NAFP_HIP_CHECK(hipEventCreateWithFlags(&e->sw_done, hipEventDisableTiming));
        NAFP_HIP_CHECK(hipEventRecord(e->sw_done, st));
        e->sw_stream = st; e->sw_recorded = true;
    }
    NAFP_HIP_CHECK(hipMemsetAsync(stats, 0, stats_bytes, st));        // statistics + counters, one fill
    hipEvent_t* ev = nullptr;
    if (e->prof_max > 0 && e->prof_count < e->prof_max) ev = e->prof_events.data() + (size_t)NAFP_PROF_EV * e->prof_count++;
    if (ev && e->prof_coarse < 2) NAFP_HIP_CHECK(hipEventRecord(ev[0], st));

    // conv0 is either materialised (z0 written to bufA) or -- default -- only its statistics are
    // computed here and conv1 re-generates z0 tiles in-kernel from the log-mel features
    // (NAFP_FUSE0=0 selects the materialised path).
    const bool alt = e->norm != NAFP_NORM_LAYER2D;             // layer_norm1d / batch_norm (norm.hip)
    // The alternates: every consumer derives identity scalars (r = 1, c = 0; NaN for a poisoned sample) from the producer's
    // statistics (`ident_stats`); layer_norm1d normalises the rows of every layer's output in place.
    auto stats_of = [&](int j) { return stats + 2 * n_seg * j; };
    auto row_pass = [&](float* z, int j) -> int {
        if (e->norm != NAFP_NORM_LAYER1D) return NAFP_OK;
        return launch_ln1d_fwd(z, n_seg * e->geom[j].Fout * e->geom[j].Tout, e->geom[j].Cout, e->d_gc[j], e->d_bc[j], st);
    };
    // (the exact-split forward takes the fused form by default: at bf16-pipe speed conv1 is bound by its A stream -- conv0's 2 MB per
    // segment written and read back --, the generator removes that stream and conv0's own launch; NAFP_X6_FUSE0=0 for the A/B)
    static const bool x6_fuse0 = []() { const char* v = getenv("NAFP_X6_FUSE0"); return !v || v[0] != '0'; }();
    const bool fuse0 = !alt && (e->opt_fuse_conv0 || (e->opt_bf16x3 == 2 && x6_fuse0)) && e->geom[1].axis == 1 && e->geom[1].Cin % 16 == 0 &&
                       e->geom[0].Tin <= 64 && (e->opt_bf16x3 != 2 || e->geom[1].Cin == 128);
    int rc = fuse0 ? launch_conv0_stats(feat, e->d_w[0], e->d_bias[0], stats, n_seg, e->geom[0], st, gstat, group_size, segment_norm)
                   : launch_conv0(feat, e->d_w[0], e->d_bias[0], e->d_gamma[0], bufA, nullptr, stats, n_seg, e->geom[0], st,
                                  gstat, group_size, segment_norm, alt);
    if (rc != NAFP_OK) return rc;
    if (!fuse0) { rc = row_pass(bufA, 0); if (rc != NAFP_OK) return rc; }
    if (ev && e->prof_coarse < 2) NAFP_HIP_CHECK(hipEventRecord(ev[1], st));
    float* cur = bufA;
    for (int j = 1; j < 16; ++j) {
        float* nxt = (j % 2 == 0) ? bufA : bufB;
        ConvGemmArgs a{};
        a.x = cur; a.stats_in = stats_of(j - 1); a.ident_stats = alt;
        a.wp = e->d_w[j]; a.G = e->d_G[j]; a.Hb = e->d_Hb[j]; a.gamma_out = e->d_gamma[j];
        a.y = nxt; a.stats_out = stats + 2 * n_seg * j; a.plain = false;
        a.slab = slab_floats ? slab : nullptr; a.slab_floats = slab_floats; a.tickets = tickets; a.bf16x3 = e->opt_bf16x3;
        static const int x6_layers = []() { const char* v = getenv("NAFP_X6_LAYERS"); return v ? (int)strtol(v, nullptr, 0) : 0xffff; }();     // diagnostic: bit j = layer j takes the split arithmetic
        if (!((x6_layers >> j) & 1)) a.bf16x3 = 0;
        if (a.bf16x3 == 2) { a.wp_hm = e->d_whm[j]; a.wp_l = e->d_wl[j]; }
        a.plan_b = fwd_plan_b();              // tile shape and split-K factor as at the reference launch size: results do not depend on n_seg
        if (j == 1 && fuse0) {
            a.x = nullptr; a.f0_feat = feat; a.f0_w = e->d_w[0]; a.f0_bias = e->d_bias[0];
            a.f0_gamma = e->d_gamma[0]; a.f0_geom = &e->geom[0];
            a.f0_gstat = gstat; a.f0_group = group_size; a.f0_segnorm = segment_norm;
        }
        // time stamps that ride on the kernels' own dispatch packets (no queue entry, no idle time): every conv's first and
        // last kernel (all stamps), or only conv1's start and conv15's stop (the GEMM span of a timed region)
        if (ev && e->prof_coarse == 0) { a.ev_start = ev[2 * j]; a.ev_stop = ev[2 * j + 1]; }
        if (ev && e->prof_coarse == 2) {
            if (j == 1) a.ev_start = ev[2];
            if (j == 15) a.ev_stop = ev[31];
        }
        rc = launch_conv_gemm(a, n_seg, e->geom[j], st);
        if (rc != NAFP_OK) return rc;
        rc = row_pass(nxt, j);
        if (rc != NAFP_OK) return rc;
        cur = nxt;
    }
    TailArgs t;
    t.x = cur; t.stats = stats_of(15); t.ident_stats = alt; t.gamma = e->d_gamma[15]; t.beta = e->d_beta[15];
    t.w1p = e->d_w1p; t.b1p = e->d_b1p; t.w2p = e->d_w2p; t.b2 = e->d_b2;
    t.out_flat = out_flat; t.out_emb = out_emb;
    t.D = (int)e->flat_dim; t.Q = e->emb_sz; t.S = e->S; t.l2norm = l2norm; t.nonfinite_weights = e->d_wflag; t.launch_error = nullptr;
    if (ev && e->prof_coarse < 2) NAFP_HIP_CHECK(hipEventRecord(ev[32], st));
    rc = launch_tail(t, n_seg, st);
    if (rc != NAFP_OK) return rc;
    if (ev && e->prof_coarse < 2) NAFP_HIP_CHECK(hipEventRecord(ev[33], st));
    return NAFP_OK;
}

extern "C" int nafp_encoder_profile_enable(nafp_encoder* e, int max_forwards) {
    if (!e || max_forwards < 0 || max_forwards > 4096) return NAFP_ERR_INVALID_ARG;
    profile_free(e);
    for (int i = 0; i < max_forwards * NAFP_PROF_EV; ++i) {
        hipEvent_t ev;
        // no system-scope fence at the stamp: with the default flags every record made the GPU write its caches back
        // (measured: 15 stamps between the GEMM convs cost 0.35 ms of a 3.6 ms forward)
        static const int ev_mode = []() { const char* v = getenv("NAFP_PROF_EVENT_FENCE"); return v ? atoi(v) : 0; }();
        hipError_t err = ev_mode == 1 ? hipEventCreate(&ev)
                         : hipEventCreateWithFlags(&ev, ev_mode == 2 ? hipEventReleaseToDevice : hipEventDisableSystemFence);
        if (err != hipSuccess) { g_last_hip_error = (int)err; profile_free(e); return NAFP_ERR_HIP; }
        e->prof_events.push_back(ev);
    }
    e->prof_max = max_forwards;
    return NAFP_OK;
}

extern "C" int nafp_encoder_profile_coarse(nafp_encoder* e, int coarse) {
    if (!e) return NAFP_ERR_INVALID_ARG;
    e->prof_coarse = coarse < 0 ? 0 : (coarse > 2 ? 2 : coarse);
    return NAFP_OK;
}

extern "C" int nafp_encoder_profile_count(const nafp_encoder* e) { return e ? e->prof_count : -1; }

extern "C" int nafp_encoder_profile_read(nafp_encoder* e, int slot, float* ms_out_host) {
    if (!e || !ms_out_host || slot < 0 || slot >= e->prof_count) return NAFP_ERR_INVALID_ARG;
    hipEvent_t* ev = e->prof_events.data() + (size_t)NAFP_PROF_EV * slot;
    NAFP_HIP_CHECK(hipEventSynchronize(e->prof_coarse == 2 ? ev[31] : ev[33]));
    for (int k = 0; k < 17; ++k) ms_out_host[k] = 0.f;
    if (e->prof_coarse == 2) {  // only the span of the 15 GEMM convs (first dispatch start -> last dispatch end)
        NAFP_HIP_CHECK(hipEventElapsedTime(ms_out_host + 1, ev[2], ev[31]));
        return NAFP_OK;
    }
    NAFP_HIP_CHECK(hipEventElapsedTime(ms_out_host + 0, ev[0], ev[1]));
    NAFP_HIP_CHECK(hipEventElapsedTime(ms_out_host + 16, ev[32], ev[33]));
    if (e->prof_coarse) {       // conv0 | the 15 GEMM convs (incl. split-K finishes) as ONE span in slot 1 | zeros | tail
        NAFP_HIP_CHECK(hipEventElapsedTime(ms_out_host + 1, ev[1], ev[32]));
        return NAFP_OK;
    }
    // every conv: start of its first kernel -> end of its last (a split-K finish kernel included)
    for (int j = 1; j < 16; ++j) NAFP_HIP_CHECK(hipEventElapsedTime(ms_out_host + j, ev[2 * j], ev[2 * j + 1]));
    return NAFP_OK;
}

extern "C" int nafp_conv_timeline(void* dev_buf, int64_t capacity_u64, int cin, int cout, int positions) {
    if (dev_buf && capacity_u64 <= 0) return NAFP_ERR_INVALID_ARG;
    return conv_timeline_set((unsigned long long*)dev_buf, capacity_u64, cin, cout, positions);
}
extern "C" int nafp_conv_timeline_grid(int* out5_host) {
    if (!out5_host) return NAFP_ERR_INVALID_ARG;
    return conv_timeline_grid(out5_host);
}

extern "C" int nafp_encoder_div_enc(nafp_encoder* e, const float* flat, int64_t n_seg,
                                    float* out_emb, int l2norm, void* stream) {
    if (!e || !flat || !out_emb || n_seg < 0) return NAFP_ERR_INVALID_ARG;
    if (!e->has_weights) return NAFP_ERR_NO_WEIGHTS;
    TailArgs t;
    t.x = flat; t.stats = nullptr; t.ident_stats = false; t.gamma = nullptr; t.beta = nullptr;
    t.w1p = e->d_w1p; t.b1p = e->d_b1p; t.w2p = e->d_w2p; t.b2 = e->d_b2;
    t.out_flat = nullptr; t.out_emb = out_emb;
    t.D = (int)e->flat_dim; t.Q = e->emb_sz; t.S = e->S; t.l2norm = l2norm; t.nonfinite_weights = e->d_wflag; t.launch_error = nullptr;
    { int wrc = wait_weights(e, (hipStream_t)stream); if (wrc != NAFP_OK) return wrc; }
    return launch_tail(t, n_seg, (hipStream_t)stream);
}


// ============================================================================================
// Training: forward that keeps every activation, and the backward pass (model/trainer.py:41-47)
// ============================================================================================
namespace {
struct TrainLayout {
    int64_t B;
    stat_t* stats; unsigned* tickets; float* mr; float* sc;
    // zeroed together at the start of the backward pass: per-layer LN sums, S1/S2 of every layer
    char* zero_begin; int64_t zero_bytes;
    double* lnsum[16]; float* S1[16]; float* S2[16];
    float* z[16]; float* v[16];          // z = gamma . ELU(t) (operand of the next conv), v = the pre-activation t
    float* slab; int64_t slab_floats;
    float* slab2; int64_t slab2_floats; unsigned* tickets2;     // the weight-gradient stream's own slab and arrival counters (opt_bwd_overlap 2)
    float* dA; float* dB; float* dy;
    float* dts[16];                      // the small layers' gradients in buffers of their own (opt_bwd_overlap 2), else nullptr
    float* dgp; float* dbp;              // the alternates (norm.hip): the positional (dgamma | dbeta) sums of the shared LayerNorm backward (max_n floats each)
    int64_t bytes;
};

TrainLayout train_layout(const nafp_encoder* e, int64_t B, void* ws) {
    TrainLayout L; L.B = B;
    char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    char* p0 = p;
    auto take = [&](int64_t bytes) { char* r = p; p += (bytes + 255) / 256 * 256; return r; };
    L.stats = (stat_t*)take((int64_t)sizeof(stat_t) * 2 * 16 * B);
    L.tickets = (unsigned*)take(2 * NAFP_TICKET_SLOTS * (int64_t)sizeof(unsigned));  // directly behind the statistics: one fill covers all
    L.tickets2 = L.tickets + NAFP_TICKET_SLOTS;                                       // (second half: the weight-gradient stream's counters)
    L.mr = (float*)take((int64_t)sizeof(float) * 2 * 16 * B);
    L.sc = (float*)take((int64_t)sizeof(float) * 8 * B);
    L.zero_begin = p;
    for (int j = 0; j < 16; ++j) L.lnsum[j] = (double*)take((int64_t)sizeof(double) * 2 * B);
    for (int j = 0; j < 16; ++j) {
        const int64_t n = (int64_t)e->geom[j].Fout * e->geom[j].Tout * e->geom[j].Cout;
        L.S1[j] = (float*)take((int64_t)sizeof(float) * n);
        L.S2[j] = (float*)take((int64_t)sizeof(float) * n);
    }
    L.zero_bytes = p - L.zero_begin;
    int64_t max_n = 0;
    for (int j = 0; j < 16; ++j) {
        const int64_t n = (int64_t)e->geom[j].Fout * e->geom[j].Tout * e->geom[j].Cout;
        max_n = std::max(max_n, n);
        L.z[j] = (float*)take((int64_t)sizeof(float) * n * B);
        // the pre-activation: only with keep_t(); layer 0's is regenerated by the backward pass -- except for layer_norm1d, whose row
        // kernel reads it
        L.v[j] = ((j == 0 && e->norm != NAFP_NORM_LAYER1D) || !e->keep_t()) ? nullptr : (float*)take((int64_t)sizeof(float) * n * B);
    }
    L.slab_floats = 0;
    for (int j = 1; j < 16; ++j)
        L.slab_floats = std::max(L.slab_floats, std::max(conv_gemm_slab_floats(B, e->geom[j], true), wgrad_slab_floats(B, e->geom[j])));
    L.slab = (float*)take((int64_t)sizeof(float) * L.slab_floats);
    L.slab2_floats = 0;
    for (int j = 1; j < 16; ++j)
        if (e->geom[j].Fout * e->geom[j].Tout < 16) L.slab2_floats = std::max(L.slab2_floats, wgrad_slab_floats(B, e->geom[j]));
    L.slab2 = (float*)take((int64_t)sizeof(float) * L.slab2_floats);
    L.dA = (float*)take((int64_t)sizeof(float) * max_n * B);
    L.dB = (float*)take((int64_t)sizeof(float) * max_n * B);
    L.dy = (float*)take((int64_t)sizeof(float) * e->emb_sz * B);
    // The layers whose weight gradient runs on the second stream (P < 16: 61 KB per sample together) keep their incoming gradient /
    // dts in a buffer of their own instead of the dA / dB pair: nothing overwrites it during the pass, so the main stream never
    // has to wait for the weight-gradient stream before a transposed conv (a wait is a barrier packet in its queue: ~14 us of
    // idle each, six per step).
    for (int j = 0; j < 16; ++j) {
        const int64_t n = (int64_t)e->geom[j].Fout * e->geom[j].Tout * e->geom[j].Cout;
        L.dts[j] = (j >= 1 && e->geom[j].Fout * e->geom[j].Tout < 16) ? (float*)take((int64_t)sizeof(float) * n * B) : nullptr;
    }
    L.dgp = nullptr; L.dbp = nullptr;
    if (e->norm != NAFP_NORM_LAYER2D) {
        L.dgp = (float*)take((int64_t)sizeof(float) * 2 * max_n);
        L.dbp = L.dgp + max_n;
    }
    L.bytes = (p - p0) + 256;
    return L;
}
}  // namespace

extern "C" int64_t nafp_encoder_train_workspace_bytes(const nafp_encoder* e, int64_t n_seg) {
    if (!e || n_seg < 0) return -1;
    return train_layout(e, n_seg, nullptr).bytes;
}

extern "C" int nafp_encoder_forward_train(nafp_encoder* e, const float* feat, int64_t n_seg, void* workspace,
                                          int64_t workspace_bytes, float* out_emb, int l2norm, void* stream) {
    if (!e || !feat || !workspace || !out_emb || n_seg <= 0) return NAFP_ERR_INVALID_ARG;
    if (!e->has_weights) return NAFP_ERR_NO_WEIGHTS;
    if (workspace_bytes < nafp_encoder_train_workspace_bytes(e, n_seg)) return NAFP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    TrainLayout L = train_layout(e, n_seg, workspace);
    e->train_rec_put(workspace, n_seg, e->keep_t());
    NAFP_HIP_CHECK(hipMemsetAsync(L.stats, 0, (char*)(L.tickets + 2 * NAFP_TICKET_SLOTS) - (char*)L.stats, st));
    static const bool split_wait = []() { const char* v = getenv("NAFP_SW_SPLIT_WAIT"); return !v || v[0] != '0'; }();
    { int wrc = wait_weights(e, st, split_wait ? 0 : 2); if (wrc != NAFP_OK) return wrc; }
    if (e->opt_bf16x3 == 2 && e->x6_dirty) {      // the option was switched on after the last set_weights (which splits by itself otherwise)
        int wrc = wait_weights(e, st);
        if (wrc != NAFP_OK) return wrc;
        { int rcs = split_all_weights(e, st); if (rcs != NAFP_OK) return rcs; }
        if (!e->sw_done) NAFP_HIP_CHECK(hipEventCreateWithFlags(&e->sw_done, hipEventDisableTiming));
        NAFP_HIP_CHECK(hipEventRecord(e->sw_done, st));
        e->sw_stream = st; e->sw_recorded = true;
    }
    // (layer 0 keeps no pre-activation: the backward pass regenerates it from `feat`, 3 FMAs per element)
    const bool alt = e->norm != NAFP_NORM_LAYER2D;             // layer_norm1d / batch_norm (norm.hip), as in encoder_forward_impl
    auto stats_of = [&](int j) { return L.stats + 2 * n_seg * j; };
    auto row_pass = [&](float* z, int j) -> int {
        if (e->norm != NAFP_NORM_LAYER1D) return NAFP_OK;
        return launch_ln1d_fwd(z, n_seg * e->geom[j].Fout * e->geom[j].Tout, e->geom[j].Cout, e->d_gc[j], e->d_bc[j], st);
    };
    int rc = launch_conv0(feat, e->d_w[0], e->d_bias[0], e->d_gamma[0], L.z[0], L.v[0], L.stats, n_seg, e->geom[0], st, nullptr, 0, 0, alt);
    if (rc != NAFP_OK) return rc;
    rc = row_pass(L.z[0], 0);
    if (rc != NAFP_OK) return rc;
    for (int j = 1; j < 16; ++j) {
        if (split_wait && j <= 2) { int wrc = wait_weights(e, st, j); if (wrc != NAFP_OK) return wrc; }
        ConvGemmArgs a{};
        a.x = L.z[j - 1]; a.stats_in = stats_of(j - 1); a.ident_stats = alt;
        a.wp = e->d_w[j]; a.G = e->d_G[j]; a.Hb = e->d_Hb[j]; a.gamma_out = e->d_gamma[j];
        a.y = L.z[j]; a.v_out = L.v[j]; a.stats_out = L.stats + 2 * n_seg * j; a.plain = false;
        a.slab = L.slab_floats ? L.slab : nullptr; a.slab_floats = L.slab_floats; a.tickets = L.tickets;
        // NAFP_OPT_BF16X3 = 2: the GEMM products of the training forward on the exact 3-way bf16 split too (same kernels, training epilogue)
        if (e->opt_bf16x3 == 2 && !alt) { a.bf16x3 = 2; a.wp_hm = e->d_whm[j]; a.wp_l = e->d_wl[j]; }
        rc = launch_conv_gemm(a, n_seg, e->geom[j], st);
        if (rc != NAFP_OK) return rc;
        rc = row_pass(L.z[j], j);
        if (rc != NAFP_OK) return rc;
    }
    TailArgs t;
    t.x = L.z[15]; t.stats = stats_of(15); t.ident_stats = alt; t.gamma = e->d_gamma[15]; t.beta = e->d_beta[15];
    t.w1p = e->d_w1p; t.b1p = e->d_b1p; t.w2p = e->d_w2p; t.b2 = e->d_b2;
    t.out_flat = nullptr; t.out_emb = out_emb;
    t.D = (int)e->flat_dim; t.Q = e->emb_sz; t.S = e->S; t.l2norm = l2norm; t.nonfinite_weights = e->d_wflag; t.launch_error = nullptr;
    return launch_tail(t, n_seg, st);
}

extern "C" int nafp_encoder_backward(nafp_encoder* e, const float* feat, const float* d_emb, int64_t n_seg,
                                     void* workspace, int64_t workspace_bytes, float* const* grads, int l2norm,
                                     void* stream) {
    if (!e || !feat || !d_emb || !workspace || !grads || n_seg <= 0) return NAFP_ERR_INVALID_ARG;
    if (!e->has_weights) return NAFP_ERR_NO_WEIGHTS;
    if (workspace_bytes < nafp_encoder_train_workspace_bytes(e, n_seg)) return NAFP_ERR_WORKSPACE;
    if (!e->train_rec_ok(workspace, n_seg, e->keep_t())) return NAFP_ERR_INVALID_ARG;     // not a layout forward_train wrote into THIS workspace
    hipStream_t st = (hipStream_t)stream;
    const int64_t B = n_seg;
    TrainLayout L = train_layout(e, B, workspace);
    const size_t n_tr = (size_t)e->n_trainable;                // (batch_norm: the moving statistics behind tensor 67 have no gradient)
    for (size_t i = 0; i < n_tr; ++i)
        if (!grads[i]) return NAFP_ERR_INVALID_ARG;
    { int wrc = wait_weights(e, st); if (wrc != NAFP_OK) return wrc; }
    const bool alt = e->norm != NAFP_NORM_LAYER2D;             // layer_norm1d / batch_norm (norm.hip)
    const stat_t* const stats_r = L.stats;
    // every gradient accumulates through atomics: zero them (adjacent tensors -- e.g. views into one flat
    // all-reduce bucket -- in one memset) together with the LN sums and S1/S2
    for (size_t i = 0; i < n_tr;) {
        size_t k = i; int64_t run = numel(e->shapes[i]);
        while (k + 1 < n_tr && grads[k + 1] == grads[i] + run) { ++k; run += numel(e->shapes[k]); }
        NAFP_HIP_CHECK(hipMemsetAsync(grads[i], 0, sizeof(float) * run, st));
        i = k + 1;
    }
    NAFP_HIP_CHECK(hipMemsetAsync(L.zero_begin, 0, L.zero_bytes, st));
    for (auto& ev : e->grad_events)
        if (!ev) NAFP_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    e->grad_events_valid = false;
    // the weight-gradient stream (see opt_bwd_overlap)
    // opt_bwd_overlap: 1 = every layer's weight gradient on the side stream (measured slower, see the struct), 2 = only the SMALL
    // layers' (P < 16 positions)
    const int ov_mode = e->opt_bwd_overlap;
    const bool overlap = ov_mode == 1;
    static const int side_max_p = []() { const char* v = getenv("NAFP_SIDE_MAXP"); return v ? atoi(v) : 16; }();      // (sweep knob: 16 = the small layers)
    auto side_layer = [&](int j) { return ov_mode == 1 || (ov_mode == 2 && j >= 1 && j <= 15 && e->geom[j].Fout * e->geom[j].Tout < side_max_p); };
    if (ov_mode != 0 && !e->side_stream) {
        int pr_least = 0, pr_greatest = 0;
        NAFP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
        static const int side_prio = []() { const char* v = getenv("NAFP_SIDE_PRIO"); return v ? atoi(v) : 1; }();   // 0: lowest, 1: default (see opt_bwd_overlap), 2: highest
        if (side_prio == 1) NAFP_HIP_CHECK(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
        else NAFP_HIP_CHECK(hipStreamCreateWithPriority(&e->side_stream, hipStreamNonBlocking, side_prio == 2 ? pr_greatest : pr_least));
        for (auto& ev : e->ev_main) NAFP_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (auto& ev : e->ev_side) NAFP_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    int rc = launch_stats_to_mr(stats_r, L.mr, e->d_inv_n, B, 16, st, alt);      // (the alternates: identity scalars, as the forward pass's consumers derived)
    if (rc != NAFP_OK) return rc;
    // tail: d_emb -> r * dxhat of the last conv + divide-and-encode gradients
    TailBwdArgs tb;
    tb.z = L.z[15]; tb.stats = stats_r + 2 * B * 15; tb.ident_stats = alt; tb.gamma = e->d_gamma[15]; tb.beta = e->d_beta[15];
    tb.w1 = e->d_w1k; tb.b1 = e->d_b1k; tb.w2 = e->d_w2k; tb.b2 = e->d_b2;
    tb.w1p = e->d_w1p; tb.b1p = e->d_b1p; tb.w2p = e->d_w2p;
    // gradient buffers: layer j's incoming gradient (then dts_j, in place) lives in its own buffer when its weight gradient runs on
    // the second stream (mode 2), else in dA / dB by parity
    static const bool own_bufs = []() { const char* v = getenv("NAFP_SIDE_OWNBUF"); return !v || v[0] != '0'; }();
    auto own = [&](int j) { return own_bufs && ov_mode == 2 && j >= 1 && side_layer(j) && L.dts[j] != nullptr; };
    auto gbuf = [&](int j) { return own(j) ? L.dts[j] : ((j & 1) ? L.dA : L.dB); };
    tb.d_emb = d_emb; tb.dy = L.dy; tb.dxh = gbuf(15); tb.ln = L.sc;      // (L.sc is rewritten by the first ln_bwd_scalars launch, after the tail)
    tb.dw1 = grads[64]; tb.db1 = grads[65]; tb.dw2 = grads[66]; tb.db2 = grads[67];
    tb.D = (int)e->flat_dim; tb.Q = e->emb_sz; tb.S = e->S; tb.l2norm = l2norm;
    rc = launch_tail_bwd(tb, B, st);
    if (rc != NAFP_OK) return rc;
    // `cur` holds r_j * dL/dxhat_j on entry of iteration j and r_{j-1} * dL/dt_j (dts) after launch_ln_bwd
    float* cur = gbuf(15); float* other = nullptr;
    static const bool ride = []() { const char* v = getenv("NAFP_SIDE_RIDE"); return !v || v[0] != '0'; }();
    bool ln_done = false;       // `cur` already holds dts_j (the LayerNorm backward of layer j ran inside dgrad_{j+1})
    bool sc_ready = false;      // L.sc already holds the scalar records of layer j (side job of wgrad_{j+1})
    bool side_pending = false;  // a weight gradient has been enqueued on the weight-gradient stream (never joined before the end of the pass)
    for (int j = 15; j >= 1; --j) {
        const ConvGeom& g = e->geom[j];
        const int P = g.Fout * g.Tout;
        const float* mr_j = L.mr + 2 * B * j; const float* mr_p = L.mr + 2 * B * (j - 1);
        other = gbuf(j - 1);
        if (!ln_done && !alt) {
            rc = launch_ln_bwd(cur, L.v[j] ? L.v[j] : L.z[j], e->d_gamma[j], mr_j, mr_p, L.lnsum[j], L.sc, grads[4 * j + 2], grads[4 * j + 3],
                               grads[4 * j + 1], L.S1[j], L.S2[j], B, P, g.Cout, st, j == 15, e->d_G[j], e->d_Hb[j], L.lnsum[j - 1],
                               nullptr, nullptr, nullptr, nullptr, nullptr, sc_ready, L.slab_floats ? L.slab : nullptr, L.slab_floats,
                               L.tickets, L.v[j] == nullptr);
            if (rc != NAFP_OK) return rc;
        } else if (alt) {
            // The alternates: `cur` holds dL/d(the stored, normalised activation of layer j).  layer_norm1d first takes it through the
            // row map's backward (-> dL/dv, dgamma_c, dbeta_c).  Then the shared kernel with identity statistics and NO LayerNorm sums
            // (lnsum stays at the zeros of the start of the pass: no reduction here, none handed down) is the plain
            // dts = cur . gamma_pos . ELU'(t) with dbias and S2 = sum_b dt (S1 = 0); its positional (dgamma | dbeta) sums go to
            // scratch, from which batch_norm forms its per-channel gradients.
            if (e->norm == NAFP_NORM_LAYER1D) {
                rc = launch_ln1d_bwd(cur, L.v[j], B * P, g.Cout, e->d_gc[j], grads[4 * j + 2], grads[4 * j + 3], st);
                if (rc != NAFP_OK) return rc;
            }
            NAFP_HIP_CHECK(hipMemsetAsync(L.dgp, 0, sizeof(float) * 2 * (int64_t)P * g.Cout, st));
            rc = launch_ln_bwd(cur, L.v[j] ? L.v[j] : L.z[j], e->d_gamma[j], mr_j, mr_p, L.lnsum[j], L.sc, L.dgp, L.dgp + (int64_t)P * g.Cout,
                               grads[4 * j + 1], L.S1[j], L.S2[j], B, P, g.Cout, st, false, nullptr, nullptr, nullptr,
                               nullptr, nullptr, nullptr, nullptr, nullptr, sc_ready, nullptr, 0, nullptr, L.v[j] == nullptr);
            if (rc != NAFP_OK) return rc;
            if (e->norm == NAFP_NORM_BATCH) {
                rc = launch_bn_param_grad(L.dgp, L.dgp + (int64_t)P * g.Cout, P, g.Cout, e->d_mm[j], e->d_mv[j], grads[4 * j + 2], grads[4 * j + 3], st);
                if (rc != NAFP_OK) return rc;
            }
        }
        // the transposed conv below writes `other`; where that is the dA / dB buffer that still holds dts_{j+1}, wgrad(j+1), on the
        // weight-gradient stream, must be done with it (not with buffers of their own: see TrainLayout::dts)
        if (j < 15 && side_layer(j + 1) && !own(j + 1)) NAFP_HIP_CHECK(hipStreamWaitEvent(st, e->ev_side[j + 1], 0));
        const bool on_side = side_layer(j);
        hipStream_t sw = on_side ? e->side_stream : st;
        const ConvGeom& gp1 = e->geom[j - 1];
        ScalarsJob sj{L.mr + 2 * B * (j - 1), L.lnsum[j - 1], j >= 2 ? L.mr + 2 * B * (j - 2) : nullptr, L.sc, (long long)B,
                      1.0 / ((double)gp1.Fout * gp1.Tout * gp1.Cout)};
        sc_ready = false;
        ln_done = !alt && j >= 2 && dgrad_ln_eligible(B, g, e->opt_fused_ln_bwd);   // (layer 0 keeps no pre-activation: see conv0 below)
        if (ln_done) {
            // transposed conv of dts_j with the LayerNorm + ELU backward of layer j-1 in its epilogue: `other` <- dts_{j-1}
            const ConvGeom& gp = e->geom[j - 1];
            const int64_t n_prev = (int64_t)gp.Fout * gp.Tout * gp.Cout;
            rc = launch_ln_bwd_scalars(mr_p, L.lnsum[j - 1], j >= 2 ? L.mr + 2 * B * (j - 2) : nullptr, L.sc, B, n_prev, st);
            if (rc != NAFP_OK) return rc;
            DgradLnArgs d{};
            d.dts_in = cur; d.wd = e->d_wd[j]; d.t = L.v[j - 1]; d.gamma = e->d_gamma[j - 1]; d.sc = L.sc; d.dts_out = other;
            d.dgamma = grads[4 * (j - 1) + 2]; d.dbeta = grads[4 * (j - 1) + 3]; d.dbias = grads[4 * (j - 1) + 1];
            if (j >= 2) { d.G = e->d_G[j - 1]; d.Hb = e->d_Hb[j - 1]; d.S1 = L.S1[j - 1]; d.S2 = L.S2[j - 1]; d.lnsum_below = L.lnsum[j - 2]; }
            rc = launch_dgrad_ln(d, B, g, st);
            if (rc != NAFP_OK) return rc;
        } else {
            // r_{j-1} * dxhat_{j-1} = transposed conv of dts_j
            ConvGemmArgs a{};
            a.x = cur; a.wp = e->d_wd[j]; a.y = other; a.plain = true; a.dgrad = true;
            // NAFP_OPT_BF16X3 = 2: the transposed conv on the exact 3-way bf16 split as well (dts split in registers like the forward's activations)
            if (e->opt_bf16x3 == 2 && !alt && !e->x6_dirty && g.Cout % 16 == 0) { a.bf16x3 = 2; a.wp_hm = e->d_wdhm[j]; a.wp_l = e->d_wdl[j]; }
            a.slab = L.slab_floats ? L.slab : nullptr; a.slab_floats = L.slab_floats;
            a.tickets = overlap ? nullptr : L.tickets;      // split-K finished in-kernel (mode 1: the side stream's wgrad shares the counters)
            a.sj = &sj;                                     // side job: the scalar records of layer j - 1
            // the weight-gradient stream starts behind this launch: its event rides on the launch's last dispatch packet
            // (hipExtLaunchKernel) instead of a record of its own in the queue (~7 us of idle each)
            if (on_side && ride) a.ev_stop = e->ev_main[j];
            rc = launch_conv_gemm(a, B, g, st);
            if (rc != NAFP_OK) return rc;
            sc_ready = true;
        }
        // dW_j = wgrad(z_{j-1}, r dt) + wgrad(gamma_{j-1}, sum_b c_b dt) + wgrad(beta_{j-1}, sum_b dt).  On the weight-gradient
        // stream it starts BEHIND the transposed conv of this layer (two MFMA-bound kernels side by side only share the
        // pipes) and so runs next to the LayerNorm backward of layer j-1, which is HBM-bound and next on the main stream.
        if (on_side) {
            if (ln_done || !ride) NAFP_HIP_CHECK(hipEventRecord(e->ev_main[j], st));
            NAFP_HIP_CHECK(hipStreamWaitEvent(sw, e->ev_main[j], 0));
            if (e->debug_side_delay_us > 0 && !side_pending) {                  // test hook: the weight-gradient stream starts late
                debug_delay_kernel<<<1, 64, 0, sw>>>((long long)e->debug_side_delay_us);
                NAFP_LAUNCH_CHECK();
            }
        }
        float* w_slab = !on_side ? L.slab : (ov_mode == 2 ? L.slab2 : nullptr);
        const int64_t w_slab_floats = !on_side ? L.slab_floats : (ov_mode == 2 ? L.slab2_floats : 0);
        unsigned* w_tickets = !on_side ? L.tickets : (ov_mode == 2 ? L.tickets2 : nullptr);
        // (the two rank-one terms ride in the main launch as two aux samples: [gamma | beta] and [S1 | S2] are adjacent pairs.)
        // Side job of that launch, single-stream mode only: the scalar records the LayerNorm backward of layer j - 1 starts from
        // (its sums are final since ln_bwd of layer j; the fused dgrad path and the side stream keep the separate launch)
        const bool pairs = L.S2[j] == L.S1[j] + (int64_t)P * g.Cout;
        if (pairs) {
            // NAFP_OPT_BF16X3 = 2: the weight gradients of the regular shapes (wgrad_fast_kernel's) on the split arithmetic as well -- both operands are
            // activations, split in registers (NAFP_X6_WGRAD=0 keeps them on the f32 pipe for an A/B: 76.8 -> 72.9 ms per step at BSZ 5120)
            static const bool x6_wgrad = []() { const char* v = getenv("NAFP_X6_WGRAD"); return !v || v[0] != '0'; }();
            rc = launch_wgrad(L.z[j - 1], cur, grads[4 * j], B, g, sw, e->d_gamma[j - 1], L.S1[j], w_slab, w_slab_floats, w_tickets, nullptr,
                              (x6_wgrad && e->opt_bf16x3 == 2 && !alt) ? 2 : 0);
            if (rc != NAFP_OK) return rc;
        } else {
            rc = launch_wgrad(L.z[j - 1], cur, grads[4 * j], B, g, sw, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
            if (rc != NAFP_OK) return rc;
            rc = launch_wgrad(e->d_gamma[j - 1], L.S1[j], grads[4 * j], 1, g, sw);
            if (rc != NAFP_OK) return rc;
            rc = launch_wgrad(e->d_beta[j - 1], L.S2[j], grads[4 * j], 1, g, sw);
            if (rc != NAFP_OK) return rc;
        }
        if (on_side) NAFP_HIP_CHECK(hipEventRecord(e->ev_side[j], sw));
        cur = other;
        // layers j .. 15 (and the divide-and-encode tensors) are final from here on (when the LayerNorm backward of layer
        // j-1 ran fused, its dgamma / dbeta / dbias are final too: they belong to the next group or are simply early)
        // (with the weight-gradient stream the event is recorded THERE: behind wgrad(j), and -- through ev_main[j] -- behind
        // everything the main stream had enqueued up to the LayerNorm backward of layer j)
        // A group whose boundary layer runs on the MAIN stream may still hold layers whose weight gradient is in flight on the
        // weight-gradient stream (mode 2: group 1 = layers 8..11, boundary 8 on the main stream, wgrad(10) / wgrad(11) over there):
        // its event is then recorded on the weight-gradient stream too, behind a main-stream event -- that stream is in order, so
        // every earlier wgrad is done, and nothing is put in front of the main stream's next kernel but one record.
        if (on_side) side_pending = true;
        for (int k = 0; k < NAFP_GRAD_GROUPS - 1; ++k)
            if (4 * j == kGroupFirst[k]) {
                if (sw == st && side_pending) {
                    NAFP_HIP_CHECK(hipEventRecord(e->ev_main[j], st));
                    NAFP_HIP_CHECK(hipStreamWaitEvent(e->side_stream, e->ev_main[j], 0));
                    NAFP_HIP_CHECK(hipEventRecord(e->grad_events[k], e->side_stream));
                } else {
                    NAFP_HIP_CHECK(hipEventRecord(e->grad_events[k], sw));
                }
            }
    }
    {
        const ConvGeom& g = e->geom[0];
        if (alt) {
            if (e->norm == NAFP_NORM_LAYER1D) {
                rc = launch_ln1d_bwd(cur, L.v[0], B * g.Fout * g.Tout, g.Cout, e->d_gc[0], grads[2], grads[3], st);
                if (rc != NAFP_OK) return rc;
            }
            NAFP_HIP_CHECK(hipMemsetAsync(L.dgp, 0, sizeof(float) * 2 * (int64_t)g.Fout * g.Tout * g.Cout, st));
            rc = launch_ln_bwd(cur, nullptr, e->d_gamma[0], L.mr, nullptr, L.lnsum[0], L.sc, L.dgp, L.dgp + (int64_t)g.Fout * g.Tout * g.Cout, grads[1],
                               nullptr, nullptr, B, g.Fout * g.Tout, g.Cout, st, false, nullptr, nullptr, nullptr,
                               feat, e->d_w[0], e->d_bias[0], &g, grads[0], sc_ready, nullptr, 0, nullptr);
            if (rc != NAFP_OK) return rc;
            if (e->norm == NAFP_NORM_BATCH) {
                rc = launch_bn_param_grad(L.dgp, L.dgp + (int64_t)g.Fout * g.Tout * g.Cout, g.Fout * g.Tout, g.Cout, e->d_mm[0], e->d_mv[0], grads[2], grads[3], st);
                if (rc != NAFP_OK) return rc;
            }
        } else if (!ln_done) {
            rc = launch_ln_bwd(cur, nullptr, e->d_gamma[0], L.mr, nullptr, L.lnsum[0], L.sc, grads[2], grads[3], grads[1],
                               nullptr, nullptr, B, g.Fout * g.Tout, g.Cout, st, false, nullptr, nullptr, nullptr,
                               feat, e->d_w[0], e->d_bias[0], &g, grads[0], sc_ready,     // ... and dW0 in the same pass
                               L.slab_floats ? L.slab : nullptr, L.slab_floats, L.tickets);
            if (rc != NAFP_OK) return rc;
        } else {
            rc = launch_conv0_bwd(feat, cur, grads[0], nullptr, B, g, st);
            if (rc != NAFP_OK) return rc;
        }
    }
    for (int j = 1; j <= 15; ++j)                       // every side-stream wgrad is done (in order: the last is the lowest side layer)
        if (side_layer(j)) { NAFP_HIP_CHECK(hipStreamWaitEvent(st, e->ev_side[j], 0)); break; }
    NAFP_HIP_CHECK(hipEventRecord(e->grad_events[NAFP_GRAD_GROUPS - 1], st));
    e->grad_events_valid = true;
    return NAFP_OK;
}

extern "C" int nafp_encoder_grad_group_range(const nafp_encoder* e, int group, int* first_tensor, int* last_tensor) {
    if (!e || group < 0 || group >= NAFP_GRAD_GROUPS || !first_tensor || !last_tensor) return NAFP_ERR_INVALID_ARG;
    *first_tensor = kGroupFirst[group]; *last_tensor = kGroupLast[group];
    return NAFP_OK;
}

extern "C" int nafp_encoder_grad_group_wait(nafp_encoder* e, int group, void* stream) {
    if (!e || group < 0 || group >= NAFP_GRAD_GROUPS) return NAFP_ERR_INVALID_ARG;
    if (!e->grad_events_valid) return NAFP_ERR_INVALID_ARG;          // no backward pass has been enqueued
    NAFP_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, e->grad_events[group], 0));
    return NAFP_OK;
}
