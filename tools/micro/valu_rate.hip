// Micro-benchmark: issue cost (shader cycles per wave-instruction, s_memtime around an unrolled loop) of the VALU / transcendental
// / MFMA instructions the attention softmax is made of, for 1, 2 and 4 waves per SIMD -- alone and interleaved with MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate      Run on the GPU box.
// Output: CSV  op, waves_per_simd, cycles_per_instruction (per wave), instructions_per_clk_per_simd
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

enum Op { EXP32 = 0, EXP16, FMA32, PKFMA, PKADD, PKMUL, MAX3, CVTPK, LDEXP, PERM32, MFMA32, MFMA_EXP2, MFMA_EXP1_FMA4, MFMA_FMA6, MFMA32_FP8, MFMA32_MXFP8, CVT_FP8, NOPS };
static const char* NAMES[] = {"v_exp_f32", "v_exp_f16", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_max3_f32",
                              "v_cvt_pk_bf16_f32", "v_ldexp_f32", "v_permlane32_swap", "mfma_32x32x16_bf16", "mfma + 2 v_exp_f32",
                              "mfma + 1 v_exp_f32 + 4 v_fma_f32", "mfma + 6 v_fma_f32", "mfma_32x32x16_fp8_fp8 (non-scaled)",
                              "mfma_scale_32x32x64_f8f6f4 (MX fp8, K = 64)", "v_cvt_pk_fp8_f32"};

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed, int iters) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-6f + i * 0.01f;
    f2 y[4] = {{seed, seed}, {seed, seed}, {seed, seed}, {seed, seed}};
    f16v acc = {};
    bf8 a = {}, b = {};
    long la = threadIdx.x, lb = 3;
    typedef __attribute__((ext_vector_type(8))) int i8v;
    i8v wa = {1, 2, 3, 4, 5, 6, 7, 8}, wb = {1, 2, 3, 4, 5, 6, 7, 8};
    unsigned u = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (OP == EXP16) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
                if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
                if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
                if (OP == LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(u));
                if (OP == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
                if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y[i & 3]));
                if (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(y[i & 3]));
                if (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(y[i & 3]));
                if (OP == PERM32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[(i + 4) & 7]));
                if (OP == MFMA32) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                if (OP == MFMA_EXP2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(x[i]), "+v"(x[(i + 1) & 7]));
                }
                if (OP == MFMA_EXP1_FMA4) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                    asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\tv_fma_f32 %4, %4, %4, %4"
                                 : "+v"(x[i]), "+v"(x[(i + 1) & 7]), "+v"(x[(i + 2) & 7]), "+v"(x[(i + 3) & 7]), "+v"(x[(i + 4) & 7]));
                }
                if (OP == MFMA32_FP8) acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la, lb, acc, 0, 0, 0);
                if (OP == MFMA32_MXFP8) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, wb, acc, 0, 0, 0, 127, 0, 127);
                if (OP == CVT_FP8) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(u) : "v"(x[i]), "v"(x[(i + 1) & 7]));
                if (OP == MFMA_FMA6) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                    asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5"
                                 : "+v"(x[i]), "+v"(x[(i + 1) & 7]), "+v"(x[(i + 2) & 7]), "+v"(x[(i + 3) & 7]), "+v"(x[(i + 4) & 7]), "+v"(x[(i + 5) & 7]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    s += y[0][0] + y[1][1] + y[2][0] + y[3][1] + acc[0] + acc[15] + (float)u;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = (t1 - t0) + (s == 1234.5f ? 1 : 0);
}

template <int OP>
void run(unsigned long long* d_out) {
    const int iters = 256;
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        hipMemset(d_out, 0, 16 * 8);
        k<OP><<<1, threads>>>(d_out, 0.5f, iters);
        k<OP><<<1, threads>>>(d_out, 0.5f, iters);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        double mx = 0;
        for (int w = 0; w < 4 * wps; ++w) mx = h[w] > mx ? (double)h[w] : mx;
        const double groups = (double)iters * 32;      // (op groups) per wave
        const double cyc = mx / groups;
        printf("%s,%d,%.2f,%.3f\n", NAMES[OP], wps, cyc, wps / cyc);
    }
}

int main() {
    unsigned long long* d_out;
    hipMalloc(&d_out, 16 * 8);
    printf("op (group),waves_per_simd,cycles_per_group_per_wave,groups_per_clk_per_simd\n");
    run<EXP32>(d_out); run<EXP16>(d_out); run<FMA32>(d_out); run<PKFMA>(d_out); run<PKADD>(d_out); run<PKMUL>(d_out);
    run<MAX3>(d_out); run<CVTPK>(d_out); run<LDEXP>(d_out); run<PERM32>(d_out); run<MFMA32>(d_out); run<MFMA_EXP2>(d_out);
    run<MFMA_EXP1_FMA4>(d_out); run<MFMA_FMA6>(d_out); run<MFMA32_FP8>(d_out); run<MFMA32_MXFP8>(d_out); run<CVT_FP8>(d_out);
    return 0;
}
