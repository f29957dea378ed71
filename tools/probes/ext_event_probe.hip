// Probe: what does timing every kernel cost?  N back-to-back launches of a ~50 us kernel on one stream:
//   (a) plain launches, (b) hipEventRecord after every launch, (c) hipExtLaunchKernel with a start/stop event pair per launch
// (the events ride on the dispatch packet's completion signal).  Prints wall time per variant and, for (c), the mean of the
// per-kernel durations.
// build: hipcc --offload-arch=gfx950 -O2 -o ext_event_probe ext_event_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int iters) {
    float x = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    p[blockIdx.x * 256 + threadIdx.x] = x;
}
int main() {
    const int N = 200, iters = 20000;
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    hipStream_t st; hipStreamCreate(&st);
    std::vector<hipEvent_t> ev(2 * N);
    for (auto& e : ev) hipEventCreate(&e);
    auto wall = [&](int mode) {
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            if (mode == 2) {
                void* args[] = {(void*)&d, (void*)&iters};
                hipExtLaunchKernel((const void*)spin, dim3(1024), dim3(256), args, 0, st, ev[2 * i], ev[2 * i + 1], 0);
            } else {
                spin<<<1024, 256, 0, st>>>(d, iters);
                if (mode == 1) hipEventRecord(ev[i], st);
            }
        }
        hipStreamSynchronize(st);
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    wall(0);
    for (int r = 0; r < 2; ++r) {
        const double a = wall(0), b = wall(1), c = wall(2);
        double sum = 0; float ms;
        for (int i = 0; i < N; ++i) { hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); sum += ms; }
        printf("plain %.3f ms | event after every launch %.3f ms | ext start/stop events %.3f ms (sum of kernel times %.3f ms)\n", a, b, c, sum);
    }
    return 0;
}
