// Stand-alone reproducer (no library, no torch), gfx950 / MI355X, ROCm 7.2: a packed-f32 vector instruction that carries an op_sel modifier
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[..]) returns wrong values in a wave that shares a compute unit with waves issuing the
// 128-bit-operand matrix instructions (v_mfma_f32_32x32x16_bf16 / _f16, 16x16x32, i32_32x32x32_i8, f8f6f4), worst with vector work between them.
//
//   victim  : every thread runs fixed chains of ONE kind of instruction (written in assembly) and compares each recomputation with its own
//             first result -- no memory, no LDS, no cross-lane traffic, nothing shared with anybody;
//   culprit : a loop of matrix instructions on another stream.  First table: which ingredient of the loop is needed (KIND below);
//             second table: which matrix instruction, on static registers / with independent vector work in the loop.
// Measured (profiles/r06_experiments.md section 5c): zero mismatches without a culprit, with f32 matrix instructions, with the 64-bit-operand
// bf16_1k instruction, and for victims without op_sel (plain packed, neg only, scalar); 10^6 - 10^8 mismatches otherwise.
//
// build: hipcc --offload-arch=gfx950 -O3 -o pk_opsel_hazard_probe pk_opsel_hazard_probe.hip ;  run: ./pk_opsel_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- victim -------------------------------------------------------------------------------------------------------------------------------
// which: 0 v_pk_fma_f32 | 1 v_pk_fma_f32 op_sel / neg | 2 v_pk_mul + v_pk_add op_sel / neg | 3 v_fma_f32 | 4 v_pk_fma_f32 op_sel only | 5 v_pk_fma_f32 neg only
//        6 v_pk_mov_b32 op_sel (a half swap) + v_pk_fma_f32 plain | 7 v_pk_add_f32 op_sel_hi:[1,0] (the complex-arithmetic broadcast form)
__global__ __launch_bounds__(256, 2) void victim_kernel(unsigned* counts, int rounds, int which) {
    const unsigned tid = threadIdx.x;
    f32x2 x[8], first[8];
    f32x2 ca = {1.0009765625f, 0.99951171875f}, cb = {0.03125f, -0.015625f};
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
                else if (which == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(x[j]) : "v"(x[j]), "v"(ca), "v"(cb));
                else if (which == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(x[j]) : "v"(x[j]), "v"(ca));
                                       asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x[j]) : "v"(x[j]), "v"(cb)); }
                else if (which == 3) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j][0]) : "v"(x[j][0]), "v"(ca[0]), "v"(cb[0]));
                                       asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j][1]) : "v"(x[j][1]), "v"(ca[1]), "v"(cb[1])); }
                else if (which == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(x[j]) : "v"(x[j]), "v"(ca), "v"(cb));
                else if (which == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,1,0]" : "=v"(x[j]) : "v"(x[j]), "v"(ca), "v"(cb));
                else if (which == 6) { asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(x[j]) : "v"(x[j]));
                                       asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[j]) : "v"(x[j]), "v"(ca), "v"(cb)); }
                else { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(ca));
                       asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x[j]) : "v"(x[j]), "v"(cb)); }
            }
        }
        if (rd == 0) { for (int j = 0; j < 8; ++j) first[j] = x[j]; }
        else for (int j = 0; j < 8; ++j) bad += (__float_as_uint(x[j][0]) != __float_as_uint(first[j][0])) + (__float_as_uint(x[j][1]) != __float_as_uint(first[j][1]));
    }
    if (bad) atomicAdd(counts, bad);
}

// ---- culprit ------------------------------------------------------------------------------------------------------------------------------
// KIND 0: bf16 32x32x16 matrix instructions on STATIC registers (no vector instruction in the loop)
//      1: one operand re-made by vector instructions (f32 -> bf16 conversions) in front of every group of 8 matrix instructions
//      2: the same vector instructions in the loop, but the matrix instructions keep reading STATIC registers (no dependency)
//      3: as 1 with 4 x s_nop 15 between the vector instructions and the first matrix instruction
//      4: as 1 with the f32 matrix instruction (v_mfma_f32_32x32x2_f32)
//      5: as 1 with v_mfma_f32_16x16x32_bf16
//      6: as 1, the operand re-made by integer instructions (v_xor_b32) instead of conversions
//      7: as 1 with the 64-bit-operand bf16 instruction (v_mfma_f32_32x32x8_bf16_1k)
//      8: as 1 with v_mfma_i32_32x32x32_i8 (128-bit operands, integer)
//      9: as 1, but ONE matrix instruction per re-made operand (the densest dependency)
template <int KIND>
__global__ __launch_bounds__(256, 2) void culprit_kernel(float* out, const float* in, int iters) {
    const int lane = threadIdx.x & 63;
    float src[8];
    for (int j = 0; j < 8; ++j) src[j] = in[lane * 8 + j];
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)src[j]; b[j] = (__bf16)in[512 + lane * 8 + j]; }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    f32x4 acc4[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) acc4[q][r] = 0.f;
    i32x16 acci[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acci[q][r] = 0;
    bf16x8 dummy = a;
    for (int it = 0; it < iters; ++it) {
        if (KIND != 0) {
            if (KIND == 6) { i32x4 ai = __builtin_bit_cast(i32x4, a); for (int j = 0; j < 4; ++j) ai[j] ^= (it & 1) << 7; asm volatile("" : "+v"(ai)); bf16x8 t = __builtin_bit_cast(bf16x8, ai); if (KIND == 2) dummy = t; else a = t; }
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) src[j] = src[j] * 1.0001f + 1e-3f;
                bf16x8 t;
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = (__bf16)src[j];
                if (KIND == 2) { asm volatile("" : "+v"(t)); dummy = t; } else a = t;
            }
            if (KIND == 3) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        }
        constexpr int NM = KIND == 9 ? 1 : 8;
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            if (KIND == 4) { acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[q & 7], (float)b[q & 7], acc[q & 3], 0, 0, 0); }
            else if (KIND == 5) acc4[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[q & 3], 0, 0, 0);
            else if (KIND == 7) { s16x4 a4 = __builtin_bit_cast(s16x4, __builtin_shufflevector(a, a, 0, 1, 2, 3)), b4 = __builtin_bit_cast(s16x4, __builtin_shufflevector(b, b, 0, 1, 2, 3));
                                  acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[q & 3], 0, 0, 0); }
            else if (KIND == 8) acci[q & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), acci[q & 3], 0, 0, 0);
            else acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q & 3], 0, 0, 0);
        }
    }
    float s = (float)dummy[0];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r] + (float)acci[q][r];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) s += acc4[q][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}


// ---- second culprit family: every matrix instruction type, (FORM 0) on static registers with no vector instruction in the loop,
// (FORM 1) with independent vector work (f32 FMAs + conversions whose results nobody reads) between the matrix instructions.
// MF: 0 v_mfma_f32_32x32x16_bf16 | 1 v_mfma_f32_32x32x2_f32 | 2 v_mfma_f32_16x16x32_bf16 | 3 v_mfma_f32_32x32x8_bf16_1k | 4 v_mfma_i32_32x32x32_i8
//     5 v_mfma_f32_32x32x16_f16 | 6 v_mfma_f32_16x16x32_f16 | 7 v_mfma_f32_16x16x4_f32 | 8 v_mfma_f32_32x32x64_f8f6f4 (fp8) | 9 v_mfma_f32_16x16x128_f8f6f4 (fp8)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int MF, int FORM>
__global__ __launch_bounds__(256, 2) void culprit2_kernel(float* out, const float* in, int iters) {
    const int lane = threadIdx.x & 63;
    float src[8];
    for (int j = 0; j < 8; ++j) src[j] = in[lane * 8 + j];
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)src[j]; b[j] = (__bf16)in[512 + lane * 8 + j]; }
    asm volatile("" : "+v"(a), "+v"(b));
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    f32x4 acc4[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) acc4[q][r] = 0.f;
    i32x16 acci[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acci[q][r] = 0;
    i32x8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = __float_as_int(src[j]) & 0x3f3f3f3f; b8[j] = (__float_as_int(src[j]) >> 3) & 0x3f3f3f3f; }
    asm volatile("" : "+v"(a8), "+v"(b8));
    bf16x8 dummy = a;
    for (int it = 0; it < iters; ++it) {
        if (FORM == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) src[j] = src[j] * 1.0001f + 1e-3f;
            bf16x8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (__bf16)src[j];
            asm volatile("" : "+v"(t));
            dummy = t;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MF == 0) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q & 3], 0, 0, 0);
            else if (MF == 1) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, a)[q & 3], __builtin_bit_cast(f32x4, b)[q & 3], acc[q & 3], 0, 0, 0);
            else if (MF == 2) acc4[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[q & 3], 0, 0, 0);
            else if (MF == 3) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, __builtin_shufflevector(a, a, 0, 1, 2, 3)), __builtin_bit_cast(s16x4, __builtin_shufflevector(b, b, 0, 1, 2, 3)), acc[q & 3], 0, 0, 0);
            else if (MF == 4) acci[q & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), acci[q & 3], 0, 0, 0);
            else if (MF == 5) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[q & 3], 0, 0, 0);
            else if (MF == 6) acc4[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc4[q & 3], 0, 0, 0);
            else if (MF == 7) acc4[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, a)[q & 3], __builtin_bit_cast(f32x4, b)[q & 3], acc4[q & 3], 0, 0, 0);
            else if (MF == 8) acc[q & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[q & 3], 0, 0, 0, 0, 0, 0);
            else acc4[q & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc4[q & 3], 0, 0, 0, 0, 0, 0);
        }
    }
    float s = (float)dummy[0];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r] + (float)acci[q][r];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) s += acc4[q][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MF, int FORM>
static void launch_culprit2(float* out, const float* in, int iters, hipStream_t st) { culprit2_kernel<MF, FORM><<<2048, 256, 0, st>>>(out, in, iters); }
typedef void (*launch_fn)(float*, const float*, int, hipStream_t);

template <int KIND>
static void launch_culprit(float* out, const float* in, int iters, hipStream_t st) { culprit_kernel<KIND><<<2048, 256, 0, st>>>(out, in, iters); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;
    float *d_in, *d_out;
    unsigned* d_counts;
    std::vector<float> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = 0.25f + 0.001f * (i % 97) - 0.003f * (i % 13);
    CHECK(hipMalloc(&d_in, 2048 * 4)); CHECK(hipMalloc(&d_out, 4096)); CHECK(hipMalloc(&d_counts, 64));
    CHECK(hipMemcpy(d_in, h.data(), 2048 * 4, hipMemcpyHostToDevice));
    hipStream_t sc, sv;
    CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    const char* vname[8] = {"v_pk_fma_f32", "v_pk_fma_f32 op_sel+neg", "v_pk_mul/add op_sel+neg", "v_fma_f32", "v_pk_fma_f32 op_sel", "v_pk_fma_f32 neg", "v_pk_mov_b32 op_sel", "v_pk_add_f32 op_sel_hi"};
    const char* cname[11] = {"bf16 MFMAs on static registers", "operand re-made by conversions, 8 MFMAs", "same vector work, MFMAs on static registers", "re-made + 4 x s_nop 15", "re-made, f32 MFMA 32x32x2",
                             "re-made, bf16 MFMA 16x16x32", "re-made by v_xor", "re-made, bf16_1k 32x32x8 (64-bit operands)", "re-made, i8 MFMA 32x32x32", "re-made, 1 MFMA each", "no culprit"};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int kind = 0; kind <= 10; ++kind) {
        printf("%-48s", cname[kind]);
        for (int which = 0; which < 8; ++which) {
            unsigned total = 0;
            float ms_c = 0.f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemsetAsync(d_counts, 0, 64, sv)); CHECK(hipStreamSynchronize(sv));
                CHECK(hipEventRecord(e0, sc));
                switch (kind) {
                    case 0: launch_culprit<0>(d_out, d_in, iters, sc); break; case 1: launch_culprit<1>(d_out, d_in, iters, sc); break;
                    case 2: launch_culprit<2>(d_out, d_in, iters, sc); break; case 3: launch_culprit<3>(d_out, d_in, iters / 4, sc); break;
                    case 4: launch_culprit<4>(d_out, d_in, iters / 2, sc); break; case 5: launch_culprit<5>(d_out, d_in, iters * 2, sc); break;
                    case 6: launch_culprit<6>(d_out, d_in, iters, sc); break; case 7: launch_culprit<7>(d_out, d_in, iters, sc); break;
                    case 8: launch_culprit<8>(d_out, d_in, iters, sc); break; case 9: launch_culprit<9>(d_out, d_in, iters * 4, sc); break;
                    default: break;
                }
                CHECK(hipEventRecord(e1, sc));
                for (int k = 0; k < 6; ++k) victim_kernel<<<1024, 256, 0, sv>>>(d_counts, 200, which);
                CHECK(hipDeviceSynchronize());
                unsigned c;
                CHECK(hipMemcpy(&c, d_counts, 4, hipMemcpyDeviceToHost));
                total += c;
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms_c += ms / 3;
            }
            if (which == 0) printf(" [%5.1f ms]", ms_c);
            printf(" %u", total);
        }
        printf("\n");
        fflush(stdout);
    }
    printf("columns:");
    for (int w = 0; w < 8; ++w) printf(" | %s", vname[w]);
    printf("\n");
    {
        const char* mname[10] = {"f32_32x32x16_bf16", "f32_32x32x2_f32", "f32_16x16x32_bf16", "f32_32x32x8_bf16_1k", "i32_32x32x32_i8", "f32_32x32x16_f16", "f32_16x16x32_f16", "f32_16x16x4_f32",
                                 "f32_32x32x64_f8f6f4 (fp8)", "f32_16x16x128_f8f6f4 (fp8)"};
        const launch_fn fn[10][2] = {{launch_culprit2<0, 0>, launch_culprit2<0, 1>}, {launch_culprit2<1, 0>, launch_culprit2<1, 1>}, {launch_culprit2<2, 0>, launch_culprit2<2, 1>},
                                     {launch_culprit2<3, 0>, launch_culprit2<3, 1>}, {launch_culprit2<4, 0>, launch_culprit2<4, 1>}, {launch_culprit2<5, 0>, launch_culprit2<5, 1>},
                                     {launch_culprit2<6, 0>, launch_culprit2<6, 1>}, {launch_culprit2<7, 0>, launch_culprit2<7, 1>}, {launch_culprit2<8, 0>, launch_culprit2<8, 1>},
                                     {launch_culprit2<9, 0>, launch_culprit2<9, 1>}};
        const int scale_num[10] = {2, 1, 4, 2, 2, 2, 4, 2, 1, 2};      // iterations scaled so that every culprit runs for several ms
        const int wsel[4] = {0, 1, 2, 4};
        printf("\nmatrix instruction x form -> mismatches of the victims (%s | %s | %s | %s)\n", vname[0], vname[1], vname[2], vname[4]);
        for (int mf = 0; mf < 10; ++mf)
            for (int form = 0; form < 2; ++form) {
                printf("v_mfma_%-28s %-28s", mname[mf], form ? "+ independent vector work" : "static registers only");
                for (int wi = 0; wi < 4; ++wi) {
                    unsigned total = 0;
                    float ms_c = 0.f;
                    for (int rep = 0; rep < 3; ++rep) {
                        CHECK(hipMemsetAsync(d_counts, 0, 64, sv)); CHECK(hipStreamSynchronize(sv));
                        CHECK(hipEventRecord(e0, sc));
                        fn[mf][form](d_out, d_in, iters * scale_num[mf] / 2, sc);
                        CHECK(hipEventRecord(e1, sc));
                        for (int k = 0; k < 6; ++k) victim_kernel<<<1024, 256, 0, sv>>>(d_counts, 200, wsel[wi]);
                        CHECK(hipDeviceSynchronize());
                        unsigned c;
                        CHECK(hipMemcpy(&c, d_counts, 4, hipMemcpyDeviceToHost));
                        total += c;
                        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms_c += ms / 3;
                    }
                    if (wi == 0) printf(" [%5.1f ms]", ms_c);
                    printf(" %u", total);
                }
                printf("\n");
                fflush(stdout);
            }
    }
    return 0;
}
