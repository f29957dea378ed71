// Ceiling of the conv kernel's K-loop STRUCTURE without any global traffic: per K-step (BK = 16) a wave does
// 2 x (4 ds_read_b128 + 16 MFMA 32x32x2 f32) and one workgroup barrier; 4 waves per workgroup, W workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_probe.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <bool BARRIER, bool LDS>
__global__ __launch_bounds__(256, 3) void kloop(float* out, int steps) {
    __shared__ __attribute__((aligned(16))) float smem[3 * 2 * 128 * 16];       // the kernel's 48 KB ring
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 3 * 2 * 128 * 16; i += 256) smem[i] = 1e-3f * (i & 255);
    __syncthreads();
    const int rl = lane & 31, hh = lane >> 5, rswz = (rl >> 2) & 3;
    const int aoff = (wm * 64 + rl) * 16, boff = 128 * 16 + (wn * 64 + rl) * 16;
    f32x16 acc[2][2] = {};
    int slot = 0;
    float4 ra = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int s = 0; s < steps; ++s) {
        if (BARRIER) __builtin_amdgcn_s_barrier();
        const float* St = smem + slot * 2 * 128 * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int pc4 = ((2 * kk + hh) ^ rswz) * 4;
            float4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = LDS ? *(const float4*)(St + aoff + mi * 32 * 16 + pc4) : ra;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = LDS ? *(const float4*)(St + boff + ni * 32 * 16 + pc4) : ra;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        if (++slot == 3) slot = 0;
    }
    float t = 0.f;
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int r = 0; r < 16; ++r) t += acc[mi][ni][r];
    if (t == 12345.678f) out[tid] = t;
}

template <bool BARRIER, bool LDS>
void run(const char* name, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 3 * 4, steps = 2000;
    kloop<BARRIER, LDS><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); kloop<BARRIER, LDS><<<blocks, 256>>>(d, steps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-28s %.2f ms  %.1f TFLOP/s\n", name, best, (double)blocks * 4 * steps * 32 * 4096.0 / best / 1e9);
}

int main() {
    float* d; hipMalloc(&d, 4096);
    run<false, false>("mfma only", d);
    run<false, true>("mfma + lds reads", d);
    run<true, false>("mfma + barrier", d);
    run<true, true>("mfma + lds reads + barrier", d);
    return 0;
}
