// Probe: what does a co-resident kernel lose next to the exact-split GEMM kernels?  `victim` fills LDS and registers with patterns, idles,
// verifies both and counts mismatches; tools/corrupt_probe.py runs it on one stream while encoder forwards run on another.
// (`spin`: a register-only MFMA spinner, for the control experiment.)
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcorrupt_probe.so corrupt_probe.hip
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NACC>
__global__ __launch_bounds__(256, 3) void spin_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(0.001f * lane, 0.002f, acc[i], 0, 0, 0);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
extern "C" int spin(int mode, int nacc, int blocks, int iters, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0 && nacc == 4) spin_kernel<0, 4><<<blocks, 256, 0, st>>>(out, iters);
    else if (mode == 0 && nacc == 2) spin_kernel<0, 2><<<blocks, 256, 0, st>>>(out, iters);
    else if (mode == 0) spin_kernel<0, 8><<<blocks, 256, 0, st>>>(out, iters);
    else spin_kernel<1, 4><<<blocks, 256, 0, st>>>(out, iters);
    return (int)hipGetLastError();
}

// counts[0]: LDS words found changed, counts[1]: register values found changed, counts[2]: workgroups run
__global__ __launch_bounds__(256, 2) void victim_kernel(unsigned* counts, int lds_words, long long wait_ticks) {
    extern __shared__ unsigned vs[];
    const unsigned tid = threadIdx.x, key = blockIdx.x * 2654435761u;
    for (int i = tid; i < lds_words; i += 256) vs[i] = key ^ (unsigned)(i * 40503u);
    unsigned r[96];
#pragma unroll
    for (int j = 0; j < 96; ++j) { r[j] = key + tid * 131u + j * 7919u; asm volatile("" : "+v"(r[j])); }
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned bad_l = 0, bad_r = 0, bad_w = 0;
    if (wait_ticks < 0 && lds_words < 0) {}
    if (wait_ticks < -1000000) {
        // EXCHANGE form: every round each wave writes a round-dependent pattern into its quarter of LDS, barrier, reads the quarter of
        // the NEXT wave and checks it, barrier -- what a kernel that stages data for other waves through LDS relies on
        const int q = lds_words / 4, wave = tid >> 6, lane = tid & 63;
        unsigned round = 0;
        for (int it = 0; it < 150; ++it) {            // (a fixed count: every thread runs the same number of barriers)
            ++round;
            for (int i = lane; i < q; i += 64) vs[wave * q + i] = (key + round * 977u) ^ (unsigned)((wave * q + i) * 40503u);
            __syncthreads();
            const int w2 = (wave + 1) & 3;
            for (int i = lane; i < q; i += 64) bad_l += vs[w2 * q + i] != ((key + round * 977u) ^ (unsigned)((w2 * q + i) * 40503u));
            __syncthreads();
        }
        for (int i = tid; i < lds_words; i += 256) vs[i] = key ^ (unsigned)(i * 40503u);
        __syncthreads();
    } else
    if (wait_ticks < 0) {
        // ACTIVE form: keep reading (and re-writing) LDS and doing arithmetic while the other kernel runs
        float f = 1.0f + tid * 1e-3f, f_ref = f;
        while (wall_clock64() - t0 < -wait_ticks) {
            for (int i = tid; i < lds_words; i += 256) { const unsigned v = vs[i]; bad_l += v != (key ^ (unsigned)(i * 40503u)); vs[i] = v; }
            for (int k = 0; k < 64; ++k) { f = f * 1.0009765625f; f_ref = __fmul_rn(f_ref, 1.0009765625f); }
            __syncthreads();
        }
        bad_w = __float_as_uint(f) != __float_as_uint(f_ref);
    } else
    while (wall_clock64() - t0 < wait_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (bad_w) atomicAdd(counts + 3, 1u);
    for (int i = tid; i < lds_words; i += 256) bad_l += vs[i] != (key ^ (unsigned)(i * 40503u));
#pragma unroll
    for (int j = 0; j < 96; ++j) { asm volatile("" : "+v"(r[j])); bad_r += r[j] != key + tid * 131u + j * 7919u; }
    if (bad_l) atomicAdd(counts, bad_l);
    if (bad_r) atomicAdd(counts + 1, bad_r);
    if (tid == 0) atomicAdd(counts + 2, 1u);
}
extern "C" int victim(unsigned* counts, int blocks, int lds_bytes, long long wait_ticks, void* stream) {
    hipFuncSetAttribute((const void*)victim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    victim_kernel<<<blocks, 256, lds_bytes, (hipStream_t)stream>>>(counts, lds_bytes / 4, wait_ticks);
    return (int)hipGetLastError();
}

// A victim with NO shared state at all: every thread recomputes the same chain of float operations (FMAs, packed FMAs, sqrt, exp2, log2,
// a DPP add) from fixed inputs over and over and compares with its own first result.  counts[4 + which]: mismatching recomputations.
__global__ __launch_bounds__(256, 2) void selfcheck_kernel(unsigned* counts, int rounds, int which) {
    const unsigned tid = threadIdx.x;
    float x[16], first[16];
    unsigned bad = 0;
    for (int rd = 0; rd < rounds; ++rd) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = 0.37f + 0.011f * j + 0.0007f * tid + 0.05f * (blockIdx.x & 7);
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#pragma unroll 1
        for (int it = 0; it < 24; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v = x[j];
                if (which == 0) v = fmaf(v, 1.0009765625f, 0.03125f) - 0.03f * v;                                  // plain FMAs
                else if (which == 1) v = sqrtf(v * v + 0.5f) * 0.9f;                                                 // v_sqrt_f32
                else if (which == 2) v = __builtin_amdgcn_exp2f(v * 0.25f) - 0.4f;                                   // v_exp_f32
                else if (which == 3) v = __builtin_amdgcn_logf(v + 1.5f) + 0.3f;                                     // v_log_f32
                else v = v * 0.75f + __shfl_xor(v, 1 + (j & 3), 64) * 0.25f;                                         // cross-lane
                x[j] = v;
            }
        }
        if (rd == 0) { for (int j = 0; j < 16; ++j) first[j] = x[j]; }
        else { for (int j = 0; j < 16; ++j) bad += __float_as_uint(x[j]) != __float_as_uint(first[j]); }
    }
    if (bad) atomicAdd(counts + 4 + which, bad);
    if (tid == 0) atomicAdd(counts + 2, 1u);
}
extern "C" int selfcheck(unsigned* counts, int blocks, int rounds, int which, void* stream) {
    selfcheck_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(counts, rounds, which);
    return (int)hipGetLastError();
}

// Packed-f32 victims (the front end, rocFFT and the GEMM epilogues are full of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32; the victims above
// hardly hold any): chains of exactly one kind of instruction, written in assembly so that the compiler cannot choose another, recomputed and
// compared with the thread's own first result.  counts[10 + which]: mismatching recomputations.
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 2) void pkcheck_kernel(unsigned* counts, int rounds, int which) {
    const unsigned tid = threadIdx.x;
    pk_f32x2 x[8], first[8];
    pk_f32x2 ca = {1.0009765625f, 0.99951171875f}, cb = {0.03125f, -0.015625f};
    asm volatile("" : "+v"(ca), "+v"(cb));
    unsigned bad = 0;
    for (int rd = 0; rd < rounds; ++rd) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j][0] = 0.37f + 0.011f * j + 0.0007f * tid + 0.05f * (blockIdx.x & 7); x[j][1] = 0.41f + 0.013f * j + 0.0005f * tid; }
#pragma unroll 1
        for (int it = 0; it < 48; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (which == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[j]) : "v"(x[j]), "v"(ca), "v"(cb));
                else if (which == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(ca)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(cb)); }
                else if (which == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(x[j]) : "v"(x[j]), "v"(ca), "v"(cb));
                else if (which == 3) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j][0]) : "v"(x[j][0]), "v"(ca[0]), "v"(cb[0])); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j][1]) : "v"(x[j][1]), "v"(ca[1]), "v"(cb[1])); }
                else { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(x[j]) : "v"(x[j]), "v"(ca)); asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x[j]) : "v"(x[j]), "v"(cb)); }
            }
        }
        if (rd == 0) { for (int j = 0; j < 8; ++j) first[j] = x[j]; }
        else {
            for (int j = 0; j < 8; ++j)
                for (int h = 0; h < 2; ++h)
                    if (__float_as_uint(x[j][h]) != __float_as_uint(first[j][h])) {
                        ++bad;
                        const unsigned slot = atomicAdd(counts + 15, 1u);          // the first 64 mismatches are recorded: expected bits, found bits, (round, j, half, lane)
                        if (slot < 64) { counts[16 + slot * 4] = __float_as_uint(first[j][h]); counts[17 + slot * 4] = __float_as_uint(x[j][h]); counts[18 + slot * 4] = (rd << 16) | (j << 8) | (h << 7) | (tid & 63);
                                         counts[19 + slot * 4] = blockIdx.x; }
                    }
        }
    }
    if (bad) atomicAdd(counts + 10 + which, bad);
    if (tid == 0) atomicAdd(counts + 2, 1u);
}
extern "C" int pkcheck(unsigned* counts, int blocks, int rounds, int which, void* stream) {
    pkcheck_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(counts, rounds, which);
    return (int)hipGetLastError();
}
