// What fp32 MFMA rate does this chip sustain?  256 CUs x 4 SIMDs x v_mfma_f32_32x32x2_f32 back to back,
// W waves per SIMD, 4 independent accumulators per wave, no memory traffic at all.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak_probe.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
        const int blocks = 256 * wgs_per_cu * 8, iters = 4000 / wgs_per_cu;
        mfma_loop<<<blocks, 256>>>(d, 100); hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); mfma_loop<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 /*waves*/ * iters * 32 /*mfma per iter*/ * 4096.0;
            printf("%d workgroup(s)/CU resident, %d blocks: %.2f ms  %.1f TFLOP/s (nominal peak 157.3 at 2.4 GHz)\n", wgs_per_cu, blocks, ms,
                   flops / ms / 1e9);
        }
    }
    return 0;
}
