// Probe: how v_mfma_f32_32x32x16_bf16 sums its 16 products and the accumulator: does a term below half an ulp of the largest
// survive next to its 15 neighbours (a wide adder, one rounding), or is every product aligned to the largest and cut?
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_bf16_sum_probe mfma_bf16_sum_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float c0, float* C) {       // A (32,16), B (32,16) row-major, C (32,32) = A B^T + c0
    const int lane = threadIdx.x, rl = lane & 31, hh = lane >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[rl * 16 + 8 * hh + j]; b[j] = (__bf16)B[rl * 16 + 8 * hh + j]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + rl] = acc[r];
}
static float run(const float* hA, const float* hB, float c0) {
    static float *dA = nullptr, *dB, *dC; float hC[1024];
    if (!dA) { hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096); }
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, c0, dC); hipDeviceSynchronize();
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    return hC[0];
}
int main() {
    float hA[512] = {0}, hB[512] = {0};
    // row 0 of A times row 0 of B: term 0 = 1 * 1, terms 1..15 = 2^-12 * 2^-12 = 2^-24 each (half an ulp of 1)
    for (int n_small = 1; n_small <= 15; n_small += 2) {
        for (int j = 0; j < 16; ++j) { hA[j] = 0; hB[j] = 0; }
        hA[0] = 1.f; hB[0] = 1.f;
        for (int j = 1; j <= n_small; ++j) { hA[j] = ldexpf(1.f, -12); hB[j] = ldexpf(1.f, -12); }
        const float got = run(hA, hB, 0.f);
        printf("1 + %2d x 2^-24 : got 1 + %.2f ulp   (exact %.1f ulp)\n", n_small, (got - 1.f) / ldexpf(1.f, -23), n_small * 0.5);
    }
    // the same with the large term in the ACCUMULATOR
    for (int n_small = 1; n_small <= 16; n_small += 3) {
        for (int j = 0; j < 16; ++j) { hA[j] = 0; hB[j] = 0; }
        for (int j = 0; j < n_small; ++j) { hA[j] = ldexpf(1.f, -12); hB[j] = ldexpf(1.f, -12); }
        const float got = run(hA, hB, 1.f);
        printf("C = 1, + %2d x 2^-24 : got 1 + %.2f ulp   (exact %.1f ulp)\n", n_small, (got - 1.f) / ldexpf(1.f, -23), n_small * 0.5);
    }
    // smaller terms: 2^-26 each (an eighth of an ulp), 16 of them = 2 ulp
    for (int e = 25; e <= 30; ++e) {
        for (int j = 0; j < 16; ++j) { hA[j] = ldexpf(1.f, -(e / 2)); hB[j] = ldexpf(1.f, -(e - e / 2)); }
        const float got = run(hA, hB, 1.f);
        printf("C = 1, + 16 x 2^-%d : got 1 + %.3f ulp   (exact %.3f ulp)\n", e, (got - 1.f) / ldexpf(1.f, -23), 16.0 * ldexp(1.0, -e) / ldexp(1.0, -23));
    }
    // negative small terms (is the cut a truncation toward zero, toward -inf, or a rounding?)
    for (int e = 24; e <= 27; ++e) {
        for (int j = 0; j < 16; ++j) { hA[j] = -ldexpf(1.f, -(e / 2)); hB[j] = ldexpf(1.f, -(e - e / 2)); }
        const float got = run(hA, hB, 1.f);
        printf("C = 1, - 16 x 2^-%d : got 1 + %.3f ulp(1-)   (exact %.3f)\n", e, (got - 1.f) / ldexpf(1.f, -24), -16.0 * ldexp(1.0, -e) / ldexp(1.0, -24));
    }
    return 0;
}
