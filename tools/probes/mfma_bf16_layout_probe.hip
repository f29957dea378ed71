// Probe: operand layout of v_mfma_f32_32x32x16_bf16 on gfx950.  Assumption under test: lane l holds row (column) l & 31 and the
// 8 consecutive k = 8 (l >> 5) .. + 7 of A (B); C/D as the f32 32x32 forms.  Prints the max error against a host product.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_bf16_layout_probe mfma_bf16_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C) {       // A (32,16), B (32,16) row-major, C (32,32) = A B^T
    const int lane = threadIdx.x, rl = lane & 31, hh = lane >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[rl * 16 + 8 * hh + j]; b[j] = (__bf16)B[rl * 16 + 8 * hh + j]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + rl] = acc[r];
}
int main() {
    float hA[512], hB[512], hC[1024], *dA, *dB, *dC;
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 17) - 8.f; hB[i] = (float)((i * 5 + 1) % 13) - 6.f; }
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC); hipDeviceSynchronize();
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double w = 0; for (int kk = 0; kk < 16; ++kk) w += (double)hA[i * 16 + kk] * hB[j * 16 + kk];
        e1 = fmax(e1, fabs(w - hC[i * 32 + j])); e2 = fmax(e2, fabs(w - hC[j * 32 + i]));
    }
    printf("max |C - A B^T| = %g   (transposed reading: %g)\n", e1, e2);
    return 0;
}
