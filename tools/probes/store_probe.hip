// What bounds conv0_kernel (1.34 GB written per 640 segments in 0.36 ms = 3.7 TB/s)?  Same grid and store pattern
// (40,960 workgroups x 32 KB, float4 per lane, 1 KiB per wave-store), with the other ingredients added one by one.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int ITER>
__global__ __launch_bounds__(256) void k(float4* __restrict__ out, const float4* __restrict__ gamma, const float* __restrict__ x) {
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * (256 * ITER);
    const float4* g = gamma + (size_t)(blockIdx.x % 64) * (256 * ITER);
    float4 gv[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) gv[i] = MODE >= 1 ? g[i * 256 + tid] : make_float4(1.f, 2.f, 3.f, 4.f);
    float xs = MODE >= 2 ? x[blockIdx.x & 1023] : 0.5f;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        float4 v = gv[i];
        if (MODE >= 2) {
            v.x = v.x * (__expf(fminf(xs * v.x, 0.f)) - 1.f); v.y = v.y * (__expf(fminf(xs * v.y, 0.f)) - 1.f);
            v.z = v.z * (__expf(fminf(xs * v.z, 0.f)) - 1.f); v.w = v.w * (__expf(fminf(xs * v.w, 0.f)) - 1.f);
        }
        out[base + i * 256 + tid] = v;
    }
}

template <int MODE, int ITER>
void run(const char* name, float4* out, float4* gamma, float* x, size_t n4) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = (int)(n4 / (256 * ITER));
    k<MODE, ITER><<<blocks, 256>>>(out, gamma, x); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); k<MODE, ITER><<<blocks, 256>>>(out, gamma, x); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-44s %d WGs  %.3f ms  %.2f TB/s written\n", name, blocks, best, n4 * 16.0 / best / 1e9);
}

int main() {
    const size_t n4 = (size_t)640 * 524288 / 4;            // conv0's output at B = 640
    float4 *out, *gamma; float* x;
    hipMalloc(&out, n4 * 16); hipMalloc(&gamma, 8 << 20); hipMalloc(&x, 4096);
    hipMemset(gamma, 0, 8 << 20); hipMemset(x, 0, 4096);
    run<0, 8>("store only, 32 KB / WG", out, gamma, x, n4);
    run<0, 16>("store only, 64 KB / WG", out, gamma, x, n4);
    run<0, 2>("store only, 8 KB / WG", out, gamma, x, n4);
    run<1, 8>("+ 2 MB L2-resident operand read", out, gamma, x, n4);
    run<2, 8>("+ exp per element", out, gamma, x, n4);
    run<2, 16>("+ exp per element, 64 KB / WG", out, gamma, x, n4);
    return 0;
}
