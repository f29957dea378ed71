// Probe: which parts of a raw buffer address take part in the hardware range check on gfx950?
//   case 0: voffset in range, soffset pushes the address beyond num_records
//   case 1: voffset itself beyond num_records
//   case 2: voffset in range, immediate offset pushes the address beyond num_records
// prints, per case, how many of the 64 floats behind the descriptor's range were overwritten.
// build: hipcc --offload-arch=gfx950 -O2 -o buffer_oob_probe buffer_oob_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float* base, int which, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 256, 0x00020000);
    const int lane = threadIdx.x;
    if (which == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 1.0f), rs, lane * 4, soff, 0);
    if (which == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 2.0f), rs, lane * 4 + 512, 0, 0);
    if (which == 2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 3.0f), rs, lane * 4 + 512 - 512 + 0, 0, 0);
    if (which == 3) {   // load side: soffset beyond the range -> zero or data?
        const int v = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, soff, 0);
        base[512 + lane] = __builtin_bit_cast(float, v);
    }
}
int main() {
    float* d; float h[1024];
    hipMalloc(&d, sizeof(h));
    for (int which = 0; which < 4; ++which) {
        for (int i = 0; i < 1024; ++i) h[i] = (float)(100 + i);
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d, which, 512);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int in_range = 0, beyond = 0;
        for (int i = 0; i < 64; ++i) in_range += h[i] != (float)(100 + i);
        for (int i = 128; i < 192; ++i) beyond += h[i] != (float)(100 + i);
        printf("case %d: changed inside the range %d, at +512 B (beyond num_records = 256) %d", which, in_range, beyond);
        if (which == 3) printf("  loaded[0] = %g (memory there holds %g)", h[512], (float)(100 + 128));
        printf("\n");
    }
    return 0;
}
