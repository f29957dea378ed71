// Which f32 MFMA shape does the chip sustain at its power plateau -- v_mfma_f32_32x32x2_f32 (16 accumulator registers read and
// written per 4096 FLOP) or v_mfma_f32_16x16x4_f32 (4 per 2048 FLOP: half the accumulator traffic per FLOP, twice the operand
// traffic)?  Both have the same nominal rate (64 FLOP / clk / SIMD).  Operands are RANDOM and change every instruction (a chip
// multiplying constants runs 19 % faster than one multiplying data: MI355X_MICROARCH.md, DVFS), every CU is busy, each variant runs
// for ~2 s so that the clock has settled, variants are interleaved.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shape_power_probe.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ rnd, float* out, int iters) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = rnd[(blockIdx.x * 256 + threadIdx.x) * 16 + i]; b[i] = rnd[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + i]; }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};                     // a 64 x 64 wave tile: 2 x 2 blocks, 64 accumulators
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {                                    // 8 k-pairs = one BK = 16 step of the conv kernel
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 3) & 7], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 5) & 7], b[u], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 5) & 7], b[(u + 3) & 7], c3, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    } else {
        f32x4 c[16];                                                          // the same 64 x 64 wave tile: 4 x 4 blocks, 64 accumulators
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)                                      // 4 k-quads = the same BK = 16 step
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + (i >> 2)) & 7], b[(u + 4 + (i & 3)) & 7], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 2;                                              // 2 workgroups of 4 waves per CU: 2 waves per SIMD (as the 256-row conv tile)
    std::vector<float> h((size_t)blocks * 256 * 16);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *rnd, *d;
    hipMalloc(&rnd, h.size() * 4); hipMalloc(&d, 4096);
    hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 60000;                                                  // x 32 (or 64) MFMAs: ~0.5 s per launch
    for (int rep = 0; rep < 4; ++rep)
        for (int shape : {32, 16}) {
            float ms_tot = 0.f;
            for (int k = 0; k < 4; ++k) {                                     // ~2 s per variant and repetition
                hipEventRecord(e0);
                if (shape == 32) mfma_loop<32><<<blocks, 256>>>(rnd, d, iters); else mfma_loop<16><<<blocks, 256>>>(rnd, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms_tot += ms;
            }
            const double flops = 4.0 * blocks * 4 /*waves*/ * (double)iters * 32 * 4096.0;     // 32 x 4096 = 64 x 2048 FLOP per iteration
            printf("rep %d  %s: %.1f ms  %.1f TFLOP/s sustained (nominal 157.3 at 2.4 GHz)\n", rep, shape == 32 ? "32x32x2 " : "16x16x4 ", ms_tot,
                   flops / ms_tot / 1e9);
        }
    return 0;
}
