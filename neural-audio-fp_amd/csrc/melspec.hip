// Front end: 1-s PCM segment -> padded STFT (Hann, 1024/256) -> |.| -> sparse mel
// -> log10(.+0.06) -> (minus group max, clamp) .  gfx950.
//
// Replaces Melspec_layer.call of the reference (model/fp/melspec/melspectrogram.py:
// 59-65 pad, 82-98 kapre STFT/Magnitude/ApplyFilterbank, 102-112 log/max/clamp).
//
// One 256-thread workgroup per segment.  The zero-padded segment sits in LDS once
// (every sample is reused by four overlapping frames).  Two real frames are packed
// into one complex 1024-point FFT (frame t -> re, frame t+1 -> im); each of the
// 4 waves runs one such FFT per round as an in-place radix-4 decimation-in-frequency FFT in LDS
// (5 passes), un-packs the two spectra, and the 256 threads then apply the mel
// bank as a <=8-tap gather from the LDS-resident magnitudes (941 non-zeros in
// total: never a dense 513x256 GEMM).  The (n_mels, 32) tile is staged in LDS and
// leaves with 16-byte coalesced stores.
#include "nafp_common.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace nafp {

constexpr int NFFT = 1024;
constexpr int HOP = 256;
constexpr int NBIN = NFFT / 2 + 1;          // 513
constexpr int MAX_TAPS = 8;
constexpr int MAX_SEG = 1 << 22;             // LDS use no longer depends on the segment length
#ifndef NAFP_MEL_CHUNK
#define NAFP_MEL_CHUNK 16                    // (8 = twice the workgroups of half the frames: measured, DESIGN.md 4.1 [r4])
#endif
constexpr int CHUNK_FRAMES = NAFP_MEL_CHUNK; // frames per LDS-resident chunk
constexpr int TILE_LD = CHUNK_FRAMES + 4;   // LDS leading dimension of the (n_mels, <= CHUNK_FRAMES) tile chunk
constexpr int SIG_CHUNK = (CHUNK_FRAMES - 1) * HOP + NFFT;  // 4864 samples feed 16 consecutive frames
constexpr int N_TW = 768;                   // twiddles used by the radix-4 passes (index < 3 * 256)

}  // namespace nafp

struct nafp_melspec {
    int fs, seg_len, n_fft, hop, n_mels, n_frames;
    float f_min, f_max;
    float2* d_twiddle;     // 1024: exp(-2 pi i n / 1024)
    float* d_window;       // 1024 periodic Hann
    int* d_mel_start;      // n_mels
    float* d_mel_w;        // n_mels * 8
};

namespace nafp {

// ---- host: librosa-0.8.1 Slaney mel bank (restated from the published formula) --
static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

static void mel_bank_host(int fs, int n_fft, int n_mels, double fmin, double fmax,
                          std::vector<float>& out) {
    const int n_freq = n_fft / 2 + 1;
    out.assign((size_t)n_mels * n_freq, 0.f);
    std::vector<double> mel_f(n_mels + 2);
    const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
    for (int i = 0; i < n_mels + 2; ++i) {
        // numpy.linspace: start + i*step, last point forced to stop
        double m = (i == n_mels + 1) ? m1 : m0 + i * ((m1 - m0) / (n_mels + 1));
        mel_f[i] = mel_to_hz(m);
    }
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < n_freq; ++k) {
            double fk = (k == n_freq - 1) ? fs / 2.0 : k * ((fs / 2.0) / (n_freq - 1));
            double lower = -(mel_f[i] - fk) / fd0;
            double upper = (mel_f[i + 2] - fk) / fd1;
            // librosa keeps `weights` in float32: round the triangle, then scale in f64, round again
            const float tri = (float)std::fmax(0.0, std::fmin(lower, upper));
            out[(size_t)i * n_freq + k] = (float)((double)tri * enorm);
        }
    }
}

// ---- device ------------------------------------------------------------------

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// One radix-4 decimation-in-frequency pass, IN PLACE, over a 1024-point complex sequence held in LDS, executed by
// one wave: 256 butterflies of span S, 4 per lane.  After the five passes (S = 256, 64, 16, 4, 1) X[k] sits at the
// base-4 digit reversal of k (rev4 below).  In place = one 8 KB buffer per wave instead of a ping-pong pair: that is
// what lets two workgroups share a CU.
template <int S>
__device__ __forceinline__ void dif_pass(float2* __restrict__ x, const float2* __restrict__ tw, int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int b = it * 64 + lane;
        const int off = b & (S - 1);
        const int base = ((b - off) << 2) + off;
        const float2 u0 = x[base], u1 = x[base + S], u2 = x[base + 2 * S], u3 = x[base + 3 * S];
        const float2 t0 = make_float2(u0.x + u2.x, u0.y + u2.y);
        const float2 t1 = make_float2(u0.x - u2.x, u0.y - u2.y);
        const float2 t2 = make_float2(u1.x + u3.x, u1.y + u3.y);
        const float2 t3 = make_float2(u1.y - u3.y, -(u1.x - u3.x));   // (u1-u3) * (-i)
        float2 y1 = make_float2(t1.x + t3.x, t1.y + t3.y);
        float2 y2 = make_float2(t0.x - t2.x, t0.y - t2.y);
        float2 y3 = make_float2(t1.x - t3.x, t1.y - t3.y);
        if (S > 1) {
            const int m = off * (256 / S);
            y1 = cmul(y1, tw[m]); y2 = cmul(y2, tw[2 * m]); y3 = cmul(y3, tw[3 * m]);
        }
        x[base] = make_float2(t0.x + t2.x, t0.y + t2.y);
        x[base + S] = y1; x[base + 2 * S] = y2; x[base + 3 * S] = y3;
    }
}

// position of X[k] after the in-place passes: the five base-4 digits of k reversed
__device__ __forceinline__ int rev4(int k) {
    const unsigned r = __brev((unsigned)k) >> 22;                   // 10-bit reversal
    return (int)(((r & 0x155u) << 1) | ((r >> 1) & 0x155u));        // swap the bits of every pair
}

// order this wave's LDS writes before its later LDS reads (no workgroup barrier)
#define NAFP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

template <typename T> __device__ __forceinline__ float pcm_to_float(T v);
template <> __device__ __forceinline__ float pcm_to_float<float>(float v) { return v; }
template <> __device__ __forceinline__ float pcm_to_float<int16_t>(int16_t v) {
    return (float)v * (1.0f / 32768.0f);     // audio_utils.py:245-246 (exact in f32)
}

__global__ void melspec_init_stats(float* group_stat, int n_groups) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_groups) {
        group_stat[2 * i] = -INFINITY;
        group_stat[2 * i + 1] = INFINITY;
    }
}

// LDS carve (80,896 B, so that TWO workgroups share a CU):
//   sig  [SIG_CHUNK]          zero-padded samples of the current 16-frame chunk (19 KB)
//   fft  [4][1024] float2     one in-place buffer per wave (32 KB); re-used for the two magnitude spectra
//   tile [n_mels][TILE_LD]    log-mel tile of the chunk (20 KB)
//   stw  [768] float2         twiddles exp(-2 pi i n / 1024), n < 768 (6 KB)
template <typename TIn>
__global__ __launch_bounds__(256, 2) void melspec_kernel(
        const TIn* __restrict__ audio, const int64_t* __restrict__ seg_offset, const int* __restrict__ seg_valid,
        float* __restrict__ feat, float* __restrict__ group_stat,
        const float2* __restrict__ tw, const float* __restrict__ window,
        const int* __restrict__ mel_start, const float* __restrict__ mel_w,
        int seg_len, int n_frames, int n_mels, int group_size) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t seg = blockIdx.x;
    float* sig = smem;
    float2* fft = (float2*)(smem + SIG_CHUNK) + wave * NFFT;
    float* tile = smem + SIG_CHUNK + 4 * 2 * NFFT;
    float2* stw = (float2*)(tile + n_mels * TILE_LD);
    for (int i = tid; i < N_TW; i += 256) stw[i] = tw[i];

    // rows of a (n_seg, seg_len) array, or -- window mode -- segment `seg` starts at sample
    // seg_offset[seg] of one PCM arena and has seg_valid[seg] real samples (the rest is the zero
    // tail load_audio pads, audio_utils.py:261-263)
    const TIn* a = audio + (seg_offset ? seg_offset[seg] : seg * seg_len);
    const int n_valid = seg_valid ? min(seg_valid[seg], seg_len) : seg_len;
    // per-thread mel filter (thread m <-> mel bin m)
    const bool has_mel = tid < n_mels;
    int mstart = 0;
    float mw[MAX_TAPS];
#pragma unroll
    for (int j = 0; j < MAX_TAPS; ++j) mw[j] = 0.f;
    if (has_mel) {
        mstart = mel_start[tid];
#pragma unroll
        for (int j = 0; j < MAX_TAPS; ++j) mw[j] = mel_w[tid * MAX_TAPS + j];
    }
    // the window values of this lane's 16 points of pass 0 (n = it*64 + lane + q*256): registers, read once
    float wreg[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) wreg[it][q] = window[it * 64 + lane + q * 256];
    __syncthreads();

    float lmax = -INFINITY, lmin = INFINITY;
    const int n_pairs = (n_frames + 1) / 2;
    float* out_seg = feat + seg * (int64_t)n_mels * n_frames;

    // one workgroup per (segment, 16-frame chunk): 640 segments are 1280 workgroups of half the length, which fill the
    // 512 resident slots (2 per CU) in 2.5 rounds of half a segment instead of 2 rounds of a whole one (the second
    // three quarters empty)
    {
        const int chunk0 = blockIdx.y * CHUNK_FRAMES;
        const int chunk_frames = min(CHUNK_FRAMES, n_frames - chunk0);
        // ---- zero-padded samples of this chunk into LDS (melspectrogram.py:59-65): padded index
        // chunk0*256 + i  <->  sample index chunk0*256 + i - 512
        __syncthreads();
        {
            // all loads of the chunk are issued before the first LDS write: 4 samples per lane and step.
            // Vector loads need the segment start 4-sample aligned (always true for rows of a (n_seg, seg_len)
            // array with seg_len % 4 == 0; window mode checks the offset).
            constexpr int STEPS = (SIG_CHUNK / 4 + 255) / 256;            // 5
            const int s_base = chunk0 * HOP - NFFT / 2;                    // multiple of 4
            const bool vec_ok = ((uintptr_t)a % (4 * sizeof(TIn))) == 0;
            float4 v[STEPS];
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int i4 = st * 256 + tid;                             // vector index inside the chunk
                const int s = s_base + 4 * i4;
                v[st] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i4 < SIG_CHUNK / 4 && s + 3 >= 0 && s < n_valid) {
                    if (vec_ok && s >= 0 && s + 3 < n_valid) {
                        if (sizeof(TIn) == 4) {
                            v[st] = *(const float4*)((const float*)a + s);
                        } else {
                            const short4 q = *(const short4*)((const int16_t*)a + s);
                            v[st] = make_float4(pcm_to_float<int16_t>(q.x), pcm_to_float<int16_t>(q.y),
                                                pcm_to_float<int16_t>(q.z), pcm_to_float<int16_t>(q.w));
                        }
                    } else {
                        v[st].x = (s >= 0 && s < n_valid) ? pcm_to_float<TIn>(a[s]) : 0.f;
                        v[st].y = (s + 1 >= 0 && s + 1 < n_valid) ? pcm_to_float<TIn>(a[s + 1]) : 0.f;
                        v[st].z = (s + 2 >= 0 && s + 2 < n_valid) ? pcm_to_float<TIn>(a[s + 2]) : 0.f;
                        v[st].w = (s + 3 >= 0 && s + 3 < n_valid) ? pcm_to_float<TIn>(a[s + 3]) : 0.f;
                    }
                }
            }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int i4 = st * 256 + tid;
                if (i4 < SIG_CHUNK / 4) *(float4*)(sig + 4 * i4) = v[st];
            }
        }
        __syncthreads();
        for (int round = 0; round < CHUNK_FRAMES / 8; ++round) {        // 4 waves x 2 frames per round
            const int pair = chunk0 / 2 + round * 4 + wave;
            const int f0 = 2 * pair, f1 = 2 * pair + 1;
            const bool live = pair < n_pairs && f0 < chunk0 + chunk_frames;
            if (live) {
                // pass 0 (S = 256) reads the windowed frames straight from `sig`: z = w*(x_f0 + i x_f1)
                const float* s0 = sig + (f0 - chunk0) * HOP;
                const bool has1 = f1 < n_frames;
                const float* s1 = sig + ((has1 ? f1 : f0) - chunk0) * HOP;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int i = it * 64 + lane;
                    float2 u[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = i + q * 256;
                        const float w = wreg[it][q];
                        u[q] = make_float2(w * s0[n], has1 ? w * s1[n] : 0.f);
                    }
                    const float2 t0 = make_float2(u[0].x + u[2].x, u[0].y + u[2].y);
                    const float2 t1 = make_float2(u[0].x - u[2].x, u[0].y - u[2].y);
                    const float2 t2 = make_float2(u[1].x + u[3].x, u[1].y + u[3].y);
                    const float2 t3 = make_float2(u[1].y - u[3].y, -(u[1].x - u[3].x));
                    fft[i] = make_float2(t0.x + t2.x, t0.y + t2.y);
                    fft[i + 256] = cmul(make_float2(t1.x + t3.x, t1.y + t3.y), stw[i]);
                    fft[i + 512] = cmul(make_float2(t0.x - t2.x, t0.y - t2.y), stw[2 * i]);
                    fft[i + 768] = cmul(make_float2(t1.x - t3.x, t1.y - t3.y), stw[3 * i]);
                }
            }
            // the passes of a frame pair touch only this wave's buffer: wave-level ordering is enough
            // (LDS operations of one wave execute in order), so the four waves are not forced into lockstep
            NAFP_WAVE_SYNC();
            if (live) dif_pass<64>(fft, stw, lane);
            NAFP_WAVE_SYNC();
            if (live) dif_pass<16>(fft, stw, lane);
            NAFP_WAVE_SYNC();
            if (live) dif_pass<4>(fft, stw, lane);
            NAFP_WAVE_SYNC();
            if (live) dif_pass<1>(fft, stw, lane);
            NAFP_WAVE_SYNC();
            if (live) {
                // un-pack: X0[k] = (Z[k]+conj Z[N-k])/2, X1[k] = (Z[k]-conj Z[N-k])/(2i); keep |.|.
                // Z[k] lives at rev4(k); everything is read into registers first, then the two magnitude
                // spectra overwrite the buffer (floats [0..512] and [520..1032]).
                float m0[9], m1[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const int k = lane + 64 * j;
                    m0[j] = 0.f; m1[j] = 0.f;
                    if (k < NBIN) {
                        const float2 z = fft[rev4(k)];
                        const float2 zc = fft[rev4((NFFT - k) & (NFFT - 1))];
                        const float ar = 0.5f * (z.x + zc.x), ai = 0.5f * (z.y - zc.y);
                        const float br = 0.5f * (z.y + zc.y), bi = -0.5f * (z.x - zc.x);
                        m0[j] = sqrtf(ar * ar + ai * ai);            // kapre Magnitude = tf.abs
                        m1[j] = sqrtf(br * br + bi * bi);
                    }
                }
                NAFP_WAVE_SYNC();
                float* mag0 = (float*)fft;             // [0..512]
                float* mag1 = mag0 + 520;              // [0..512]
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const int k = lane + 64 * j;
                    if (k < NBIN) { mag0[k] = m0[j]; mag1[k] = m1[j]; }
                }
            }
            __syncthreads();
            // mel gather for the 8 frames of this round (thread <-> mel bin)
            if (has_mel) {
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) {
                    const int p = chunk0 / 2 + round * 4 + wv;
                    const float* mg = (const float*)((float2*)(smem + SIG_CHUNK) + wv * NFFT);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int f = 2 * p + h;
                        if (f < chunk0 + chunk_frames && f < n_frames) {
                            const float* mm = mg + h * 520 + mstart;
                            float acc = 0.f;
#pragma unroll
                            for (int j = 0; j < MAX_TAPS; ++j) acc += mw[j] * mm[j];
                            // melspectrogram.py:104,107
                            const float v = logf(max_keep_nan(acc + 0.06f, 1e-10f)) / 2.302585092994046f;
                            tile[tid * TILE_LD + (f - chunk0)] = v;
                            lmax = fmaxf(lmax, v);
                            lmin = fminf(lmin, v);
                        }
                    }
                }
            }
            __syncthreads();
        }
        // ---- tile chunk out: (n_mels, chunk_frames) -> feat[seg][m][chunk0 + t] ----
        if (chunk_frames == CHUNK_FRAMES && (n_frames & 3) == 0) {
            for (int idx = tid; idx < n_mels * (CHUNK_FRAMES / 4); idx += 256) {
                const int m = idx / (CHUNK_FRAMES / 4), q = idx % (CHUNK_FRAMES / 4);
                const float4 v = *(const float4*)(tile + m * TILE_LD + 4 * q);
                *(float4*)(out_seg + (int64_t)m * n_frames + chunk0 + 4 * q) = v;
            }
        } else {
            for (int idx = tid; idx < n_mels * chunk_frames; idx += 256) {
                const int m = idx / chunk_frames, t = idx % chunk_frames;
                out_seg[(int64_t)m * n_frames + chunk0 + t] = tile[m * TILE_LD + t];
            }
        }
        __syncthreads();
    }

    // ---- group max / min (melspectrogram.py:108, 110) ----
    lmax = wave_max(lmax);
    lmin = wave_min(lmin);
    float* red = tile;                     // tile is dead after the last chunk's barrier
    if (lane == 0) { red[wave] = lmax; red[4 + wave] = lmin; }
    __syncthreads();
    if (tid == 0) {
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float mn = fminf(fminf(red[4], red[5]), fminf(red[6], red[7]));
        const int64_t g = group_size > 0 ? seg / group_size : 0;
        atomic_max_float(group_stat + 2 * g, mx);
        atomic_min_float(group_stat + 2 * g + 1, mn);
    }
}

// ---- radix-16 front end -----------------------------------------------------------------------------------------------
// The same workgroup layout as melspec_kernel, with the 1024-point FFT of a frame pair factored 16 x 16 x 4 and the two
// radix-16 stages held in REGISTERS (16 points per lane): three LDS exchanges per FFT instead of five read-modify-write
// passes, and every exchange conflict-free by construction.  With n = 64 n1 + l and k = k1 + 16 k2:
//   A  lane l:            Y[k1, l] = W1024^(l k1) . DFT16_{n1}( w[n] z[n] )                 -> buf[68 k1 + l]
//   B  lane (k1, r):      T[k1, r, q] = W64^(r q) . DFT16_{m}( Y[k1, 4 m + r] )             -> buf[68 k1 + 17 r + q]
//   C  lane (k1, g):      X[k1 + 16 q + 256 p] = DFT4_{r}( T[k1, r, q] ),  q = 4 j + g      -> buf[nat(k)]
// nat(k): see nat_addr() (keeps the C stores and the un-pack reads free of bank conflicts).
// The level-1 twiddles W1024^(l k1) depend on the lane only: 15 registers, loaded once per workgroup.
constexpr int R16_BUF = 16 * 68;             // float2 per wave

__device__ __forceinline__ void dft4(float2 u0, float2 u1, float2 u2, float2 u3, float2& o0, float2& o1, float2& o2, float2& o3) {
    const float2 t0 = make_float2(u0.x + u2.x, u0.y + u2.y);
    const float2 t1 = make_float2(u0.x - u2.x, u0.y - u2.y);
    const float2 t2 = make_float2(u1.x + u3.x, u1.y + u3.y);
    const float2 t3 = make_float2(u1.y - u3.y, -(u1.x - u3.x));   // (u1 - u3) * (-i)
    o0 = make_float2(t0.x + t2.x, t0.y + t2.y);
    o1 = make_float2(t1.x + t3.x, t1.y + t3.y);
    o2 = make_float2(t0.x - t2.x, t0.y - t2.y);
    o3 = make_float2(t1.x - t3.x, t1.y - t3.y);
}

// exp(-2 pi i m / 16) for the products b * c of the two radix-4 stages (b, c in 0..3)
template <int M> __device__ __forceinline__ float2 w16() {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508978f, H = 0.70710678118654752f;
    return M == 0 ? make_float2(1.f, 0.f) : M == 1 ? make_float2(C1, -S1) : M == 2 ? make_float2(H, -H)
         : M == 3 ? make_float2(S1, -C1) : M == 4 ? make_float2(0.f, -1.f) : M == 6 ? make_float2(-H, -H)
         : make_float2(-C1, S1);            // M == 9
}

// x[k] <- sum_n x[n] exp(-2 pi i n k / 16), natural order in and out, all indices compile-time
__device__ __forceinline__ void dft16(float2 (&x)[16]) {
    float2 y[4][4];
#define NAFP_S1(b_)                                                                            \
    {                                                                                          \
        float2 o0, o1, o2, o3;                                                                 \
        dft4(x[b_], x[4 + b_], x[8 + b_], x[12 + b_], o0, o1, o2, o3);                         \
        y[0][b_] = o0;                                                                         \
        y[1][b_] = (b_) == 0 ? o1 : cmul(o1, w16<(b_) * 1>());                                 \
        y[2][b_] = (b_) == 0 ? o2 : cmul(o2, w16<(b_) * 2>());                                 \
        y[3][b_] = (b_) == 0 ? o3 : cmul(o3, w16<(b_) * 3>());                                 \
    }
    NAFP_S1(0) NAFP_S1(1) NAFP_S1(2) NAFP_S1(3)
#undef NAFP_S1
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(y[c][0], y[c][1], y[c][2], y[c][3], x[c], x[c + 4], x[c + 8], x[c + 12]);
}

// [r5] k with bits 2-3 flipped by bits 4-5: the 16 lanes a ds_write_b64 serves per cycle (k1 = 4 a .. 4 a + 3 in bits 0-1, r in
// bits 4-5 of k) then cover 32 distinct banks -- the round-3 form (bit 3 flipped by bit 5) left r = 0 / 1 and 2 / 3 on the same
// banks (a 2-way conflict on every store of stage C); the un-pack reads (64 consecutive k per half wave) stay a permutation of
// 64 consecutive float2.
__device__ __forceinline__ int nat_addr(int k) { return k ^ (((k >> 4) & 3) << 2); }

// LDS carve (75,328 B: two workgroups per CU):
//   sig  [SIG_CHUNK]            zero-padded samples of the 16-frame chunk (19 KB)
//   buf  [4][R16_BUF] float2    one exchange buffer per wave (34 KB); re-used for the two magnitude spectra
//   tile [n_mels][TILE_LD]      log-mel tile of the chunk (20 KB)
//   tw64 [64] float2            exp(-2 pi i n / 64)
template <typename TIn>
__global__ __launch_bounds__(256, 2) void melspec_r16_kernel(
        const TIn* __restrict__ audio, const int64_t* __restrict__ seg_offset, const int* __restrict__ seg_valid,
        float* __restrict__ feat, float* __restrict__ group_stat,
        const float2* __restrict__ tw, const float* __restrict__ window,
        const int* __restrict__ mel_start, const float* __restrict__ mel_w,
        int seg_len, int n_frames, int n_mels, int group_size) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t seg = blockIdx.x;
    float* sig = smem;
    float2* buf = (float2*)(smem + SIG_CHUNK) + wave * R16_BUF;
    float* tile = smem + SIG_CHUNK + 4 * 2 * R16_BUF;
    float2* tw64 = (float2*)(tile + n_mels * TILE_LD);
    if (tid < 64) tw64[tid] = tw[16 * tid];

    const TIn* a = audio + (seg_offset ? seg_offset[seg] : seg * seg_len);
    const int n_valid = seg_valid ? min(seg_valid[seg], seg_len) : seg_len;
    const bool has_mel = tid < n_mels;
    int mstart = 0;
    float mw[MAX_TAPS];
#pragma unroll
    for (int j = 0; j < MAX_TAPS; ++j) mw[j] = 0.f;
    if (has_mel) {
        mstart = mel_start[tid];
#pragma unroll
        for (int j = 0; j < MAX_TAPS; ++j) mw[j] = mel_w[tid * MAX_TAPS + j];
    }
    // [r5] taps this WAVE needs: the Slaney bank has 2-3 taps per filter at the bottom and 8 at the top (mean 3.7), and a wave holds 64
    // neighbouring filters -- the gather below runs as many tap rounds as the widest filter of the wave has, not always 8
    int ntap = 0;
#pragma unroll
    for (int j = 0; j < MAX_TAPS; ++j) ntap = mw[j] != 0.f ? j + 1 : ntap;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ntap = max(ntap, __shfl_xor(ntap, o, 64));
    ntap = __builtin_amdgcn_readfirstlane(ntap);
    // tile[mel][frame] with TILE_LD = 20: lanes t, t + 8, t + 16, t + 24 of a half wave write the same bank; the low two bits of the
    // frame are XOR-ed with (mel >> 3) & 3 on the way in and undone on the way out (a float4 of the row holds the same four frames)
    const int tsw = (tid >> 3) & 3;
    // per lane, once: the window of its 16 points n = 64 n1 + lane and the level-1 twiddles W1024^(lane k1)
    float wreg[16];
    float2 tw1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { wreg[i] = window[64 * i + lane]; tw1[i] = tw[lane * i]; }
    const int k1 = lane >> 2, r = lane & 3;

    float lmax = -INFINITY, lmin = INFINITY;
    const int n_pairs = (n_frames + 1) / 2;
    float* out_seg = feat + seg * (int64_t)n_mels * n_frames;
    const int chunk0 = blockIdx.y * CHUNK_FRAMES;
    const int chunk_frames = min(CHUNK_FRAMES, n_frames - chunk0);
    {
        // zero-padded samples of this chunk into LDS (melspectrogram.py:59-65), all loads issued before the first LDS write
        constexpr int STEPS = (SIG_CHUNK / 4 + 255) / 256;            // 5
        const int s_base = chunk0 * HOP - NFFT / 2;                    // multiple of 4
        const bool vec_ok = ((uintptr_t)a % (4 * sizeof(TIn))) == 0;
        float4 v[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int i4 = st * 256 + tid;
            const int s = s_base + 4 * i4;
            v[st] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i4 < SIG_CHUNK / 4 && s + 3 >= 0 && s < n_valid) {
                if (vec_ok && s >= 0 && s + 3 < n_valid) {
                    if (sizeof(TIn) == 4) {
                        v[st] = *(const float4*)((const float*)a + s);
                    } else {
                        const short4 q = *(const short4*)((const int16_t*)a + s);
                        v[st] = make_float4(pcm_to_float<int16_t>(q.x), pcm_to_float<int16_t>(q.y),
                                            pcm_to_float<int16_t>(q.z), pcm_to_float<int16_t>(q.w));
                    }
                } else {
                    v[st].x = (s >= 0 && s < n_valid) ? pcm_to_float<TIn>(a[s]) : 0.f;
                    v[st].y = (s + 1 >= 0 && s + 1 < n_valid) ? pcm_to_float<TIn>(a[s + 1]) : 0.f;
                    v[st].z = (s + 2 >= 0 && s + 2 < n_valid) ? pcm_to_float<TIn>(a[s + 2]) : 0.f;
                    v[st].w = (s + 3 >= 0 && s + 3 < n_valid) ? pcm_to_float<TIn>(a[s + 3]) : 0.f;
                }
            }
        }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int i4 = st * 256 + tid;
            if (i4 < SIG_CHUNK / 4) *(float4*)(sig + 4 * i4) = v[st];
        }
    }
    __syncthreads();
    for (int round = 0; round < CHUNK_FRAMES / 8; ++round) {        // 4 waves x 2 frames per round
        const int pair = chunk0 / 2 + round * 4 + wave;
        const int f0 = 2 * pair, f1 = 2 * pair + 1;
        const bool live = pair < n_pairs && f0 < chunk0 + chunk_frames;
        float2 x[16];
        if (live) {
            // A: z = w (x_f0 + i x_f1) at n = 64 n1 + lane, DFT16 over n1, level-1 twiddle
            const float* s0 = sig + (f0 - chunk0) * HOP + lane;
            const bool has1 = f1 < n_frames;
            const float* s1 = sig + ((has1 ? f1 : f0) - chunk0) * HOP + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = make_float2(wreg[i] * s0[64 * i], has1 ? wreg[i] * s1[64 * i] : 0.f);
            dft16(x);
            buf[lane] = x[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) buf[68 * i + lane] = cmul(x[i], tw1[i]);
        }
        NAFP_WAVE_SYNC();
        if (live) {
            // B: DFT16 over m of Y[k1, 4 m + r], level-2 twiddle W64^(r q)
#pragma unroll
            for (int m = 0; m < 16; ++m) x[m] = buf[68 * k1 + 4 * m + r];
            dft16(x);
#pragma unroll
            for (int q = 1; q < 16; ++q) x[q] = cmul(x[q], tw64[r * q]);
        }
        NAFP_WAVE_SYNC();
        if (live) {
#pragma unroll
            for (int q = 0; q < 16; ++q) buf[68 * k1 + 17 * r + q] = x[q];
        }
        NAFP_WAVE_SYNC();
        if (live) {
            // C: DFT4 over r of T[k1, r, q] for q = 4 j + g (g = this lane's r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2* t = buf + 68 * k1 + 4 * j + r;
                dft4(t[0], t[17], t[34], t[51], x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]);
            }
        }
        NAFP_WAVE_SYNC();
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) buf[nat_addr(k1 + 16 * (4 * j + r) + 256 * p)] = x[4 * j + p];
        }
        NAFP_WAVE_SYNC();
        if (live) {
            // un-pack: X0[k] = (Z[k]+conj Z[N-k])/2, X1[k] = (Z[k]-conj Z[N-k])/(2i); keep |.|
            float m0[9], m1[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int k = lane + 64 * j;
                m0[j] = 0.f; m1[j] = 0.f;
                if (k < NBIN) {
                    const float2 z = buf[nat_addr(k)];
                    const float2 zc = buf[nat_addr((NFFT - k) & (NFFT - 1))];
                    const float ar = 0.5f * (z.x + zc.x), ai = 0.5f * (z.y - zc.y);
                    const float br = 0.5f * (z.y + zc.y), bi = -0.5f * (z.x - zc.x);
                    m0[j] = sqrtf(ar * ar + ai * ai);            // kapre Magnitude = tf.abs
                    m1[j] = sqrtf(br * br + bi * bi);
                }
            }
            NAFP_WAVE_SYNC();
            float* mag0 = (float*)buf;             // [0..512]
            float* mag1 = mag0 + 520;              // [0..512]
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int k = lane + 64 * j;
                if (k < NBIN) { mag0[k] = m0[j]; mag1[k] = m1[j]; }
            }
        }
        __syncthreads();
        if (has_mel) {
            // tap-major: one wave-uniform test per tap round (the taps beyond a filter's own are zero weights: the shorter rounds
            // drop only x * 0 terms, added in the same order -- bit-identical to the 8-tap loop), 8 gathers + FMAs per round, all
            // addresses one per-lane base (mstart) plus immediates
            const float* mg0 = (const float*)((float2*)(smem + SIG_CHUNK)) + mstart;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int j = 0; j < MAX_TAPS; ++j) {
                if (j < ntap) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += mw[j] * mg0[(e >> 1) * (2 * R16_BUF) + (e & 1) * 520 + j];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = 2 * (chunk0 / 2 + round * 4 + (e >> 1)) + (e & 1);
                if (f < chunk0 + chunk_frames && f < n_frames) {
                    const float v = logf(max_keep_nan(acc[e] + 0.06f, 1e-10f)) / 2.302585092994046f;     // melspectrogram.py:104,107
                    tile[tid * TILE_LD + ((f - chunk0) ^ tsw)] = v;
                    lmax = fmaxf(lmax, v);
                    lmin = fminf(lmin, v);
                }
            }
        }
        __syncthreads();
    }
    if (chunk_frames == CHUNK_FRAMES && (n_frames & 3) == 0) {
        for (int idx = tid; idx < n_mels * (CHUNK_FRAMES / 4); idx += 256) {
            const int m = idx / (CHUNK_FRAMES / 4), q = idx % (CHUNK_FRAMES / 4);
            float4 v = *(const float4*)(tile + m * TILE_LD + 4 * q);
            const int sw = (m >> 3) & 3;                  // undo the frame swizzle of the row: element e holds frame e ^ sw
            if (sw & 1) { float t = v.x; v.x = v.y; v.y = t; t = v.z; v.z = v.w; v.w = t; }
            if (sw & 2) { float t = v.x; v.x = v.z; v.z = t; t = v.y; v.y = v.w; v.w = t; }
            *(float4*)(out_seg + (int64_t)m * n_frames + chunk0 + 4 * q) = v;
        }
    } else {
        for (int idx = tid; idx < n_mels * chunk_frames; idx += 256) {
            const int m = idx / chunk_frames, t = idx % chunk_frames;
            out_seg[(int64_t)m * n_frames + chunk0 + t] = tile[m * TILE_LD + (t ^ ((m >> 3) & 3))];
        }
    }
    __syncthreads();
    lmax = wave_max(lmax);
    lmin = wave_min(lmin);
    float* red = tile;
    if (lane == 0) { red[wave] = lmax; red[4 + wave] = lmin; }
    __syncthreads();
    if (tid == 0) {
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float mn = fminf(fminf(red[4], red[5]), fminf(red[6], red[7]));
        const int64_t g = group_size > 0 ? seg / group_size : 0;
        atomic_max_float(group_stat + 2 * g, mx);
        atomic_min_float(group_stat + 2 * g + 1, mn);
    }
}

// feat <- max(raw - group_max, -80) [; segment_norm]   (melspectrogram.py:108-111)
__global__ __launch_bounds__(256) void melspec_finalize_kernel(
        float* __restrict__ feat, const float* __restrict__ group_stat, int64_t n_vec4,
        int vec4_per_seg, int group_size, int segment_norm) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_vec4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t seg = i / vec4_per_seg;
        const int64_t g = group_size > 0 ? seg / group_size : 0;
        const float gmax = group_stat[2 * g];
        float4 v = ((float4*)feat)[i];
        v.x = max_keep_nan(v.x - gmax, -80.f);
        v.y = max_keep_nan(v.y - gmax, -80.f);
        v.z = max_keep_nan(v.z - gmax, -80.f);
        v.w = max_keep_nan(v.w - gmax, -80.f);
        if (segment_norm) {
            const float mn = fmaxf(group_stat[2 * g + 1] - gmax, -80.f);
            const float h = mn / 2.f, d = fabsf(h + 1e-10f);
            v.x = (v.x - h) / d; v.y = (v.y - h) / d; v.z = (v.z - h) / d; v.w = (v.w - h) / d;
        }
        ((float4*)feat)[i] = v;
    }
}

template <typename TIn>
static int melspec_forward(nafp_melspec* p, const TIn* audio, int64_t n_seg, int group_size,
                           int segment_norm, float* feat, float* group_stat, void* stream,
                           const int64_t* seg_offset = nullptr, const int* seg_valid = nullptr) {
    if (!p || !audio || !feat || !group_stat || n_seg < 0) return NAFP_ERR_INVALID_ARG;
    if (n_seg == 0) return NAFP_OK;
    hipStream_t st = (hipStream_t)stream;
    if (group_size <= 0 || group_size > n_seg) group_size = (int)std::min<int64_t>(n_seg, INT32_MAX);
    const int n_groups = (int)((n_seg + group_size - 1) / group_size);
    melspec_init_stats<<<(n_groups + 255) / 256, 256, 0, st>>>(group_stat, n_groups);
    NAFP_LAUNCH_CHECK();
    // NAFP_MELSPEC_R4=1: the round-1/2 kernel (five in-place radix-4 passes through LDS) for A/B runs
    static const bool r4 = []() { const char* e = getenv("NAFP_MELSPEC_R4"); return e && e[0] == '1'; }();
    const dim3 grid((unsigned)n_seg, (unsigned)((p->n_frames + CHUNK_FRAMES - 1) / CHUNK_FRAMES));
    if (r4) {
        const size_t lds = (size_t)(SIG_CHUNK + 4 * 2 * NFFT + p->n_mels * TILE_LD + 2 * N_TW) * sizeof(float);
        melspec_kernel<TIn><<<grid, 256, lds, st>>>(audio, seg_offset, seg_valid, feat, group_stat, p->d_twiddle, p->d_window,
                                                    p->d_mel_start, p->d_mel_w, p->seg_len, p->n_frames, p->n_mels, group_size);
    } else {
        const size_t lds = (size_t)(SIG_CHUNK + 4 * 2 * R16_BUF + p->n_mels * TILE_LD + 2 * 64) * sizeof(float);
        melspec_r16_kernel<TIn><<<grid, 256, lds, st>>>(audio, seg_offset, seg_valid, feat, group_stat, p->d_twiddle, p->d_window,
                                                        p->d_mel_start, p->d_mel_w, p->seg_len, p->n_frames, p->n_mels, group_size);
    }
    NAFP_LAUNCH_CHECK();
    if (segment_norm & NAFP_MELSPEC_DEFER) return NAFP_OK;   // raw log-mel + group_stat: the consumer finishes
    segment_norm &= 1;
    const int per_seg = p->n_mels * p->n_frames;     // multiple of 4 (n_mels % 64 == 0)
    const int64_t n_vec4 = n_seg * per_seg / 4;
    const int blocks = (int)std::min<int64_t>((n_vec4 + 255) / 256, 2048);
    melspec_finalize_kernel<<<blocks, 256, 0, st>>>(feat, group_stat, n_vec4, per_seg / 4,
                                                    group_size, segment_norm);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

}  // namespace nafp

using namespace nafp;

extern "C" int nafp_mel_filterbank_host(int fs, int n_fft, int n_mels, float f_min, float f_max,
                                        float* out_host) {
    if (!out_host || fs <= 0 || n_fft <= 0 || n_mels <= 0 || !(f_max > f_min)) return NAFP_ERR_INVALID_ARG;
    std::vector<float> fb;
    mel_bank_host(fs, n_fft, n_mels, f_min, f_max, fb);
    std::copy(fb.begin(), fb.end(), out_host);
    return NAFP_OK;
}

extern "C" int nafp_melspec_create(nafp_melspec** plan, int fs, int seg_len, int n_fft, int hop,
                                   int n_mels, float f_min, float f_max) {
    if (!plan || fs <= 0 || seg_len <= 0 || !(f_max > f_min)) return NAFP_ERR_INVALID_ARG;
    if (n_fft != NFFT || hop != HOP || n_mels <= 0 || n_mels > 256 || (n_mels % 64) != 0 ||
        seg_len > MAX_SEG)
        return NAFP_ERR_UNSUPPORTED;
    std::vector<float> fb;
    mel_bank_host(fs, n_fft, n_mels, f_min, f_max, fb);
    std::vector<int> start(n_mels, 0);
    std::vector<float> w((size_t)n_mels * MAX_TAPS, 0.f);
    for (int m = 0; m < n_mels; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < NBIN; ++k)
            if (fb[(size_t)m * NBIN + k] != 0.f) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) { start[m] = 0; continue; }
        if (hi - lo + 1 > MAX_TAPS) return NAFP_ERR_UNSUPPORTED;
        if (lo + MAX_TAPS > NBIN) lo = NBIN - MAX_TAPS;       // keep the 8-wide window in range
        start[m] = lo;
        for (int j = 0; j < MAX_TAPS; ++j) w[(size_t)m * MAX_TAPS + j] = fb[(size_t)m * NBIN + lo + j];
    }
    std::vector<float2> tw(NFFT);
    std::vector<float> win(NFFT);
    for (int n = 0; n < NFFT; ++n) {
        const double a = -2.0 * M_PI * n / NFFT;
        tw[n] = make_float2((float)std::cos(a), (float)std::sin(a));
        win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / NFFT));   // tf.signal.hann_window, periodic
    }
    nafp_melspec* p = new nafp_melspec();
    p->fs = fs; p->seg_len = seg_len; p->n_fft = n_fft; p->hop = hop; p->n_mels = n_mels;
    p->f_min = f_min; p->f_max = f_max;
    p->n_frames = 1 + seg_len / hop;          // 1 + ((seg_len + n_fft) - n_fft) // hop
    p->d_twiddle = nullptr; p->d_window = nullptr; p->d_mel_start = nullptr; p->d_mel_w = nullptr;
    auto fail = [&](hipError_t e) { g_last_hip_error = (int)e; nafp_melspec_destroy(p); return NAFP_ERR_HIP; };
    hipError_t e;
    if ((e = hipMalloc(&p->d_twiddle, sizeof(float2) * NFFT)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&p->d_window, sizeof(float) * NFFT)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&p->d_mel_start, sizeof(int) * n_mels)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&p->d_mel_w, sizeof(float) * n_mels * MAX_TAPS)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(p->d_twiddle, tw.data(), sizeof(float2) * NFFT, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(p->d_window, win.data(), sizeof(float) * NFFT, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(p->d_mel_start, start.data(), sizeof(int) * n_mels, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(p->d_mel_w, w.data(), sizeof(float) * n_mels * MAX_TAPS, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    // the kernel needs > 64 KiB of dynamic LDS (80,896 B: two workgroups per CU)
    const int lds = (SIG_CHUNK + 4 * 2 * NFFT + n_mels * TILE_LD + 2 * N_TW) * (int)sizeof(float);
    if ((e = hipFuncSetAttribute((const void*)melspec_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute((const void*)melspec_kernel<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return fail(e);
    const int lds16 = (SIG_CHUNK + 4 * 2 * R16_BUF + n_mels * TILE_LD + 2 * 64) * (int)sizeof(float);
    if ((e = hipFuncSetAttribute((const void*)melspec_r16_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16)) != hipSuccess) return fail(e);
    if ((e = hipFuncSetAttribute((const void*)melspec_r16_kernel<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16)) != hipSuccess) return fail(e);
    *plan = p;
    return NAFP_OK;
}

extern "C" int nafp_melspec_destroy(nafp_melspec* p) {
    if (!p) return NAFP_OK;
    if (p->d_twiddle) (void)hipFree(p->d_twiddle);
    if (p->d_window) (void)hipFree(p->d_window);
    if (p->d_mel_start) (void)hipFree(p->d_mel_start);
    if (p->d_mel_w) (void)hipFree(p->d_mel_w);
    delete p;
    return NAFP_OK;
}

extern "C" int nafp_melspec_n_frames(const nafp_melspec* p) { return p ? p->n_frames : -1; }
extern "C" int nafp_melspec_n_mels(const nafp_melspec* p) { return p ? p->n_mels : -1; }

extern "C" int nafp_melspec_forward_f32(nafp_melspec* plan, const float* audio, int64_t n_seg,
                                        int group_size, int segment_norm, float* feat,
                                        float* group_stat, void* stream) {
    return melspec_forward<float>(plan, audio, n_seg, group_size, segment_norm, feat, group_stat, stream);
}
extern "C" int nafp_melspec_forward_i16(nafp_melspec* plan, const int16_t* audio, int64_t n_seg,
                                        int group_size, int segment_norm, float* feat,
                                        float* group_stat, void* stream) {
    return melspec_forward<int16_t>(plan, audio, n_seg, group_size, segment_norm, feat, group_stat, stream);
}

extern "C" int nafp_melspec_finish(nafp_melspec* p, float* feat, const float* group_stat, int64_t n_seg, int group_size,
                                   int segment_norm, void* stream) {
    if (!p || !feat || !group_stat || n_seg < 0) return NAFP_ERR_INVALID_ARG;
    if (n_seg == 0) return NAFP_OK;
    if (group_size <= 0 || group_size > n_seg) group_size = (int)std::min<int64_t>(n_seg, INT32_MAX);
    const int per_seg = p->n_mels * p->n_frames;
    const int64_t n_vec4 = n_seg * per_seg / 4;
    const int blocks = (int)std::min<int64_t>((n_vec4 + 255) / 256, 2048);
    melspec_finalize_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(feat, group_stat, n_vec4, per_seg / 4, group_size,
                                                                     segment_norm & 1);
    NAFP_LAUNCH_CHECK();
    return NAFP_OK;
}

extern "C" int nafp_melspec_forward_windows_i16(nafp_melspec* plan, const int16_t* pcm, const int64_t* seg_offset,
                                                const int32_t* seg_valid, int64_t n_seg, int group_size,
                                                int segment_norm, float* feat, float* group_stat, void* stream) {
    if (!seg_offset || !seg_valid) return NAFP_ERR_INVALID_ARG;
    return melspec_forward<int16_t>(plan, pcm, n_seg, group_size, segment_norm, feat, group_stat, stream, seg_offset,
                                    seg_valid);
}
