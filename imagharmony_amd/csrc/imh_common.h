// Common device helpers for the gfx950 (CDNA4 / MI355X) kernels.  gfx950 only: no
// compatibility macros, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "imh_layout.h"

namespace imh {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <typename T> struct Pk2;
template <> struct Pk2<__bf16> { typedef __attribute__((ext_vector_type(2))) __bf16 t; };
template <> struct Pk2<_Float16> { typedef __attribute__((ext_vector_type(2))) _Float16 t; };
template <typename T> struct Vec;
template <> struct Vec<bf16_t> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct Vec<f16_t> { typedef f16x8 v8; typedef f16x4 v4; };

// Kernel qualifier of the gemm.hip kernels (the ones that carry the folded-LayerNorm epilogue): packed-fp32 VALU code generation OFF (scalar v_fma_f32 instead of
// compiler-formed v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Measured on MI355X (tools/pk_fma_probe.py reproduces it
// against a build without this attribute): with two or more workgroups resident per CU and the epilogue's operands
// arriving from cold memory, SLP-formed `v_pk_fma_f32 ... op_sel:[0,1,0]` (the LOW lane reading the ODD register of a
// pair) returned wrong low halves for one 16-lane row at a time -- 300-800 wrong outputs per 8192 x 1280 x 640
// folded-LayerNorm GEMM, non-deterministic; gone with scalar fp32 math, with one workgroup per CU, or with this attribute
// (0 wrong in 108 cold runs).  The epilogues are not VALU-bound, so scalar fp32 costs nothing measurable.
// gemm_ring.hip / norm.hip / elementwise.hip contain no odd-register low-lane selects (checked in the ISA) and stay as they are (the
// attribute would also stop hipcc from inlining the HIP headers' own device functions there).  The attention kernels use
// packed math deliberately (float2 builtins with broadcast operands); the one place where the SLP vectoriser formed the
// same pattern there (the folded LayerNorm of the projected query) is written with fma_nopk().
#define IMH_KERNEL __attribute__((target("no-packed-fp32-ops"))) __global__
// a*b + c as a scalar v_fma_f32 the SLP vectoriser cannot pair into v_pk_fma_f32
__device__ __forceinline__ float fma_nopk(float a, float b, float c) {
    float r = __builtin_fmaf(a, b, c);
    asm volatile("" : "+v"(r));
    return r;
}

// ---- MFMA wrappers (cdna_hip_programming.md section 3) ----
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ---- packed dot product with fp32 accumulate (v_dot2c_f32_bf16 / v_dot2c_f32_f16): c + a.x*b.x + a.y*b.y ----
__device__ __forceinline__ float dot2(bf16x2 a, bf16x2 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
__device__ __forceinline__ float dot2(f16x2 a, f16x2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
// ---- async global -> LDS copy, 16 B per lane, destination = lds_base + lane*16 ----
// lds_base must be wave-uniform (it travels in M0).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a literal): counted waits of rings whose in-flight depth varies at the ends
__device__ __forceinline__ void wait_vmcnt_dyn(const int n) {
#define IMH_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {       // wave-uniform; the immediate must be a literal
        IMH_VMC(0) IMH_VMC(1) IMH_VMC(2) IMH_VMC(3) IMH_VMC(4) IMH_VMC(5) IMH_VMC(6) IMH_VMC(7) IMH_VMC(8) IMH_VMC(9) IMH_VMC(10) IMH_VMC(11)
        IMH_VMC(12) IMH_VMC(13) IMH_VMC(14) IMH_VMC(15) IMH_VMC(16) IMH_VMC(17) IMH_VMC(18) IMH_VMC(19) IMH_VMC(20) IMH_VMC(21) IMH_VMC(22)
        IMH_VMC(23) IMH_VMC(24) IMH_VMC(25) IMH_VMC(26) IMH_VMC(27) IMH_VMC(28) IMH_VMC(29) IMH_VMC(30)
        default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    }
#undef IMH_VMC
}

// ... and the same wait where n is one of a few compile-time candidates in the steady state: an if-chain of one to four scalar compares.
// (The 32-way switch above lowers to a tree of ~6 dependent scalar branches through the structurizer's flow blocks: 230-400 cycles PER CALL,
// measured as a constant "vmcnt" / "weight wait" segment in every per-step probe of the LDS-halo convs -- tools/hws_phase_probe.py,
// profiles/r05_halo_phase_probe.txt.  Keep wait_vmcnt_dyn for prologues and the ragged ends of a ring.)
template <int N0, int... Ns>
__device__ __forceinline__ void wait_vmcnt_of(const int n) {
    if (n == N0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N0) : "memory");
    else if constexpr (sizeof...(Ns) > 0) wait_vmcnt_of<Ns...>(n);
    else wait_vmcnt_dyn(n);
}

// a 256-B page of zeros: out-of-range tile rows / conv padding taps fetch from here, so
// the main loops carry no bounds branches
static __device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];   // one copy per translation unit

// Tail prefetch: before a workgroup exits it touches its slice (one dword per 128-B line) of the NEXT kernel's
// weight matrix, pulling it from HBM into L2 / Infinity Cache while the rest of this kernel is still running.
// Every layer's weights are read exactly once per UNet forward (5 GB per forward), so without this each GEMM
// starts on cold HBM lines (+2..13 us per launch measured in round 2).  Purely a cache hint.
__device__ __forceinline__ void tail_prefetch(const void* ptr, unsigned bytes, unsigned bid, unsigned nblocks,
                                              unsigned tid, unsigned nthreads) {
    if (!ptr) return;
    const unsigned per = (((bytes + nblocks - 1) / nblocks) + 127u) & ~127u;
    const unsigned s0 = bid * per;
    const unsigned s1 = min(bytes, s0 + per);
    for (unsigned o = s0 + tid * 128u; o < s1; o += nthreads * 128u)
        (void)*(const volatile unsigned*)((const unsigned char*)ptr + o);
}

// A 16-B output store at agent scope (`sc1`): WRITE-THROUGH -- the line leaves for memory (the memory-side cache) now instead of staying dirty
// in this XCD's L2 until the end-of-kernel write-back, which the next kernel waits for (dirty bytes / ~6 TB/s: 0.9 us per 5 MB).  Only for
// stores that cover whole 32-B sectors with neighbouring lanes (rows written as full lines): scattered 4- / 8- / 16-B pieces measure SLOWER
// this way, staging the GEMM epilogues through LDS to make them whole rows gains nothing at the wall, and the non-temporal hint costs the
// consumer 2-8 us (profiles/r05_forward_ab_write_through.txt).
template <typename V>
__device__ __forceinline__ void st16_wt(V* p, const V& t) {
    static_assert(sizeof(V) == 16, "one dwordx4");
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const u4 d = __builtin_bit_cast(u4, t);
    // (s_nop: a VMEM store of more than 64 bits still reads its data registers for a few cycles after issue, and hipcc's hazard recognizer
    // does not look inside inline asm -- without it the next VALU write to `d` corrupts the stored line)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(d) : "memory");
}

__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// x * sigmoid(x) as 1 mul + v_exp + add + v_rcp + mul (an IEEE divide here costs ~10 more instructions; the result is rounded to
// bf16 / fp16 anyway); exp2 overflow for x << 0 gives rcp(inf) = 0 -> -0, the limit
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
// erf-GELU (diffusers GEGLU / nn.GELU(): 0.5 x (1 + erf(x / sqrt 2))) WITHOUT transcendentals: erf(x / sqrt 2) = x Q(x^2) on
// |x| <= 3.3 sqrt 2 (clamped; 1 - erf(3.3) = 3e-6), Q a degree-10 minimax fit evaluated by Horner in u = 2 x^2 / xmax^2 - 1 (in [-1, 1]:
// well conditioned in fp32).  |erf error| <= 5.1e-6, |GELU error| <= 1.2e-5 over all x in fp32 (beyond the fit range erf is set to +-1
// exactly, so GELU(x) is exactly x or 0 there) -- two orders below bf16 / fp16
// output resolution.  Only FMAs, so two gates go through one v_pk_fma_f32 stream (gelu2): ~8.5 VALU issues per gate instead of
// 14 + v_rcp + v_exp (Abramowitz-Stegun 7.1.26, rounds 1-3) -- the GEGLU epilogue of ff.net.0 evaluates 160 gates per lane per tile.
// (fit: tools/fit_gelu_poly.py)
typedef __attribute__((ext_vector_type(2))) float f32x2v;
#define GELU_Q0 0.3027370870113373f
#define GELU_Q1 -0.1496594101190567f
#define GELU_Q2 0.10756179690361023f
#define GELU_Q3 -0.08084674924612045f
#define GELU_Q4 0.05918922647833824f
#define GELU_Q5 -0.04226003959774971f
#define GELU_Q6 0.02678486704826355f
#define GELU_Q7 -0.012068570591509342f
#define GELU_Q8 0.00669495714828372f
#define GELU_Q9 -0.006969131994992495f
#define GELU_Q10 0.003111163154244423f
constexpr float GELU_XMAX = 4.666904755831214f;              // 3.3 * sqrt(2)
constexpr float GELU_UA = 2.0f / (GELU_XMAX * GELU_XMAX);
__device__ __forceinline__ constexpr float gelu_coef(int i) {
    // Q(u) = sum_i c_i u^i, highest degree first
    constexpr float c[11] = {GELU_Q10, GELU_Q9, GELU_Q8, GELU_Q7, GELU_Q6, GELU_Q5, GELU_Q4, GELU_Q3, GELU_Q2, GELU_Q1, GELU_Q0};
    return c[i];
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -GELU_XMAX, GELU_XMAX);
    const float u = __builtin_fmaf(xc * xc, GELU_UA, -1.0f);
    float q = gelu_coef(0);
#pragma unroll
    for (int i = 1; i <= 10; ++i) q = __builtin_fmaf(q, u, gelu_coef(i));
    const float h = 0.5f * x;
    // beyond the fit range erf is exactly +-1 (a clamped fit value of 1 - 3e-6 would leave x * 1.5e-6 instead of 0 for x << 0: an error
    // that grows with |x|)
    const float e = __builtin_fabsf(x) >= GELU_XMAX ? __builtin_copysignf(1.0f, x) : xc * q;
    return __builtin_fmaf(h, e, h);
}
// two gates at once on packed fp32 (plain even-aligned register pairs, broadcast constants: no op_sel low-lane selects)
__device__ __forceinline__ f32x2v gelu2(f32x2v x) {
    f32x2v xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -GELU_XMAX, GELU_XMAX);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -GELU_XMAX, GELU_XMAX);
    const f32x2v ua = {GELU_UA, GELU_UA}, m1 = {-1.0f, -1.0f};
    const f32x2v u = __builtin_elementwise_fma(xc * xc, ua, m1);
    f32x2v q = {gelu_coef(0), gelu_coef(0)};
#pragma unroll
    for (int i = 1; i <= 10; ++i) { const f32x2v c = {gelu_coef(i), gelu_coef(i)}; q = __builtin_elementwise_fma(q, u, c); }
    const f32x2v hf = {0.5f, 0.5f};
    const f32x2v h = x * hf;
    f32x2v e = xc * q;
    e[0] = __builtin_fabsf(x[0]) >= GELU_XMAX ? __builtin_copysignf(1.0f, x[0]) : e[0];      // exactly +-1 beyond the fit range (see gelu_erf_f)
    e[1] = __builtin_fabsf(x[1]) >= GELU_XMAX ? __builtin_copysignf(1.0f, x[1]) : e[1];
    return __builtin_elementwise_fma(h, e, h);
}
// GEGLU over a lane's NV consecutive pre-activation columns, interleaved in QUADS (value_2k, value_2k+1, gate_2k, gate_2k+1) -- the two
// gates of a quad sit in one even-aligned accumulator pair: o[2k], o[2k+1] = v[4k], v[4k+1] * gelu(v[4k+2], v[4k+3])
template <int NV>
__device__ __forceinline__ void geglu_quads(const float* v, float* o) {
    static_assert(NV % 4 == 0, "whole quads per lane");
#pragma unroll
    for (int k = 0; k < NV / 4; ++k) {
        const f32x2v val = {v[4 * k], v[4 * k + 1]}, gate = {v[4 * k + 2], v[4 * k + 3]};
        const f32x2v r = val * gelu2(gate);
        o[2 * k] = r[0]; o[2 * k + 1] = r[1];
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// butterfly steps between lanes 16 / 32 apart on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap; no LDS
// round trip like ds_bpermute): swapping x with itself leaves {r0,r0,r2,r2} / {r1,r1,r3,r3} (rows of 16 lanes),
// resp. {lo,lo} / {hi,hi}, whose sum is the pair sum in every lane.
__device__ __forceinline__ float xor16_sum(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float xor32_sum(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace imh

// ---- status codes of the C ABI (include/imh.h) ----
#define IMH_OK 0
#define IMH_ERR_ARG (-1)
#define IMH_ERR_SHAPE (-2)
#define IMH_ERR_DTYPE (-3)
#define IMH_ERR_LAUNCH (-4)
#define IMH_ERR_WORKSPACE (-5)

#define IMH_DT_BF16 0
#define IMH_DT_F16 1

namespace imh {
// Dynamic-LDS opt-in (> 64 KB) is a per-DEVICE function attribute: remember it per device ordinal, so a process that drives a
// second GPU through the C ABI sets it there as well (a per-process flag would leave every such launch failing on device 1).
struct DynLdsOnce {
    unsigned long long done = 0;       // bit d: attribute set on device d (ordinals >= 64: set every time)
    void ensure(const void* kern, int bytes) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return;
        (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (dev >= 0 && dev < 64) done |= 1ull << dev;
    }
};
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// Kernel variants that no tuning.json entry and no default mode reaches (measured, not selected: profiles/NEGATIVE_RESULTS.md) are compiled
// only with -DIMH_EXPERIMENTAL (IMH_EXPERIMENTAL=1 python -m imagharmony_amd.build); the default library refuses their codes / modes.
// imh_debug_set(1, 0) returns 1 from an experimental build, 0 otherwise (the tests filter on it).
#ifdef IMH_EXPERIMENTAL
#define IMH_EXP_ONLY(...) __VA_ARGS__
#else
#define IMH_EXP_ONLY(...)
#endif
int experimental_refused(const char* what);
}
