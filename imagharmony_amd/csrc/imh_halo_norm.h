// GroupNorm (+ SiLU) of staged halo pieces, in place in LDS (conv_halo.hip / conv_hws.hip: the ResnetBlock2D front end norm -> SiLU -> conv
// inside the conv's halo staging).  One piece = 8 halo pixels x 64 channels = 64 lanes x 16 B; every piece of a lane holds the same logical
// 16-B chunk, i.e. the same 8 channels (gch), so the lane's (scale, shift) pairs are fetched from the LDS table ONCE PER CHUNK and the SiLU
// switch is a wave-uniform branch around the whole piece (round 6: 78 -> 60 instructions and four LDS reads fewer per piece; the pieces of one
// call are read together before the first is transformed).
#pragma once
#include "imh_common.h"

namespace imh {

template <typename T>
struct HaloNorm {
    typedef typename Vec<T>::v8 v8;
    float sc[8], sh[8];
    int ct_loaded = -1;
    __device__ __forceinline__ void load(const float* gtab, const int ct, const int gch) {
        if (ct == ct_loaded) return;                 // (wave-uniform)
        ct_loaded = ct;
        const f32x4* tb = (const f32x4*)(gtab + (ct * GEMM_BK + gch) * 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const f32x4 v = tb[e]; sc[2 * e] = v[0]; sh[2 * e] = v[1]; sc[2 * e + 1] = v[2]; sh[2 * e + 1] = v[3]; }
    }
    template <bool SILU>
    __device__ __forceinline__ v8 apply(const v8 t) const {
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = __builtin_fmaf(to_f32(t[e]), sc[e], sh[e]);
            if (SILU) f = silu_f(f);
            // (opaque to the optimiser: with the SiLU switch a compile-time constant hipcc would contract the last multiply / the fma with the
            // conversion into v_fma_mixlo_f16 -- ONE rounding where the apply pass of norm.hip, whose switch is a run-time select, rounds to
            // fp32 and then to fp16; the in-kernel transform and the apply pass must give the same bits, tests/test_gpu_gnstats.py)
            asm("" : "+v"(f));
            o[e] = from_f32<T>(f);
        }
        return o;
    }
    // pieces at addr[0 .. N): valid[k] = transform it (wave-uniform), real[k] = this lane's pixel is not padding (padding stays zero: the conv
    // pads the NORMALISED tensor)
    template <int N>
    __device__ __forceinline__ void run(unsigned char* const (&addr)[N], const bool (&valid)[N], const bool (&real)[N], const bool silu) const {
        v8 tv[N];
#pragma unroll
        for (int k = 0; k < N; ++k) if (valid[k]) tv[k] = *(const v8*)addr[k];
        if (silu) {
#pragma unroll
            for (int k = 0; k < N; ++k) if (valid[k]) { const v8 o = apply<true>(tv[k]); if (real[k]) *(v8*)addr[k] = o; }
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) if (valid[k]) { const v8 o = apply<false>(tv[k]); if (real[k]) *(v8*)addr[k] = o; }
        }
    }
};

}  // namespace imh
