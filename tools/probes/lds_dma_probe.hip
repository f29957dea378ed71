// Probe: buffer_load_dwordx4 ... lds (direct-to-LDS DMA) on gfx950.
//  - does an out-of-range lane write ZERO into its LDS slot?
//  - is the LDS image lane-linear (base + lane*16)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, float* out, int n_bytes) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * 64 * 4; i += 64) lds[i] = -7.f;      // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_bytes, 0x00020000);
    // lanes 0..31 in range (reversed order to show per-lane source), 32..63 out of range
    unsigned voff = lane < 32 ? (unsigned)(31 - lane) * 16u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    // second instruction with a scalar offset, all lanes in range, into the next 1 KiB
    unsigned voff2 = (unsigned)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 256), 16, voff2, 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 2 * 64 * 4; i += 64) out[i] = lds[i];
}
int main() {
    const int n = 1024;   // floats
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, o, n * 4);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    printf("lane0 slot: %g %g %g %g (expect 124..127)\n", r[0], r[1], r[2], r[3]);
    printf("lane31 slot: %g %g %g %g (expect 0..3)\n", r[124], r[125], r[126], r[127]);
    printf("lane32 slot (OOB): %g %g %g %g (0 = zero fill, -7 = untouched)\n", r[128], r[129], r[130], r[131]);
    printf("lane63 slot (OOB): %g %g %g %g\n", r[252], r[253], r[254], r[255]);
    printf("2nd instr lane0: %g %g %g %g (expect 256..259)\n", r[256], r[257], r[258], r[259]);
    printf("2nd instr lane63: %g %g %g %g (expect 508..511)\n", r[508], r[509], r[510], r[511]);
    return 0;
}
